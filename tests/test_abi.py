"""CPU-side checks of the C ABI: the library loads and exports every symbol include/ste.h declares; argument
validation works without a GPU (no compute calls)."""
import ctypes as C
import os
import re

import pytest
from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from track_estimators._hip import binding

    lib = binding.load()
    hdr = open(os.path.join(ROOT, "include", "ste.h")).read()
    declared = set(re.findall(r"^(?:int|size_t|const char\*)\s+(ste_\w+)\s*\(", hdr, flags=re.M))
    assert declared, "no declarations parsed from include/ste.h"
    assert declared == set(binding.SYMBOLS), declared ^ set(binding.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ste_version() == int(re.search(r"#define STE_VERSION (\d+)", hdr).group(1))


def test_struct_layout_matches_header():
    """Field order of the ctypes mirror == field order in the header (a silent mismatch would scramble pointers)."""
    from track_estimators._hip import binding

    hdr = open(os.path.join(ROOT, "include", "ste.h")).read()
    for cname, mirror in (("ste_ukf_batch_f64", binding.SteUkfBatchF64), ("ste_gp_batch_f64", binding.SteGpBatchF64),
                          ("ste_prep_batch_f64", binding.StePrepBatchF64), ("ste_fwd_sched_f64", binding.SteFwdSchedF64),
                          ("ste_bwd_sched_f64", binding.SteBwdSchedF64)):
        body = hdr[hdr.index("typedef struct %s {" % cname): hdr.index("} %s;" % cname)]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = re.findall(r"(?:const\s+)?(?:int32_t|uint32_t|int64_t|double|size_t|void|ste_ukf_batch_f64)\s*\*?\s*(\w+)\s*;", body)
        assert fields == [f[0] for f in mirror._fields_], cname
    assert C.sizeof(binding.SteUkfBatchF64) == 24 + 3 * 8 + 22 * 8 + 16 + 8 + 8 + 8  # 0.3.1: track_stride, sm_pos, 2 x int32
    assert C.sizeof(binding.SteFwdSchedF64) == 8 + 8 + 16 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8  # 0.3.2


def _minimal_batch(binding, keep):
    """A batch struct that passes every NULL check (the pointers are never dereferenced: argument errors come first)."""
    import numpy as np

    s = binding.SteUkfBatchF64()
    s.B, s.Nmax, s.Tmax, s.n = 8, 256, 4, 4
    s.w0, s.wi, s.fan_scale = -1.0 / 3.0, 1.0 / 6.0, 3.0
    m = np.eye(4)
    keep.append(m)
    s.H = s.Q = s.R = m.ctypes.data
    for name in ("x0", "P0", "dt", "sog_rate", "cog_rate", "upd_idx", "z", "fwd_mean", "fwd_cov", "status"):
        setattr(s, name, 0x1000)
    return s


def test_window_and_slice_argument_validation():
    """include/ste.h 0.3.1: a window's stride must hold it, time slices start and end on multiples of STE_SLICE_ALIGN,
    quad-per-track slices need full covariances -- all refused before any launch (no GPU needed)."""
    from track_estimators._hip import binding

    lib, keep = binding.load(), []
    s = _minimal_batch(binding, keep)
    s.track_stride = 4
    assert lib.ste_ukf_forward_f64(C.byref(s), None) == -1 and b"track_stride" in lib.ste_last_error()
    s.track_stride = 0
    s.step_begin, s.step_end = 10, 128
    assert lib.ste_ukf_forward_f64(C.byref(s), None) == -1 and b"STE_SLICE_ALIGN" in lib.ste_last_error()
    s.step_begin, s.step_end = 64, 100
    assert lib.ste_ukf_forward_f64(C.byref(s), None) == -1 and b"STE_SLICE_ALIGN" in lib.ste_last_error()
    s.step_begin, s.step_end = 128, 64
    assert lib.ste_ukf_forward_f64(C.byref(s), None) == -1 and b"step_begin" in lib.ste_last_error()
    s.step_begin, s.step_end = 64, 300
    assert lib.ste_ukf_forward_f64(C.byref(s), None) == -1
    s.step_begin, s.step_end = 64, 128
    s.flags = binding.STE_FLAG_PACKED_COV | binding.STE_FLAG_LANES_4
    assert lib.ste_ukf_forward_f64(C.byref(s), None) == -1 and b"lane-per-track" in lib.ste_last_error()
    s.flags = 0
    s.sm_mean = s.sm_cov = 0x1000
    assert lib.ste_ukf_urtss_f64(C.byref(s), None) == -1 and b"whole passes" in lib.ste_last_error()
    assert binding.STE_SLICE_ALIGN == 64


def test_fleet_windows_and_slice_bounds():
    from track_estimators import batch

    for n, chunk in ((100_000, 10_000), (35_000, 10_000), (116, 10_000), (10_001, 10_000), (64, 64), (65, 64)):
        w = batch.fleet_windows(n, chunk)
        assert w[0][0] == 0 and w[-1][1] == n and all(a[1] == b[0] for a, b in zip(w, w[1:]))
        assert all((hi - lo) % 64 == 0 for lo, hi in w[:-1]) and len(w) == -(-n // chunk)
    for n, k in ((500, 4), (500, 1), (500, 3), (17_084, 8), (63, 4), (64, 2), (129, 2), (0, 2)):
        b = batch.DeviceBatch.slice_bounds(n, k)
        assert b[0][0] == 0 and b[-1][1] == n and all(a[1] == c[0] for a, c in zip(b, b[1:])) and len(b) <= max(k, 1)
        assert all(lo % 64 == 0 for lo, _ in b)


def test_argument_validation_without_gpu():
    from track_estimators._hip import binding

    lib = binding.load()
    s = binding.SteUkfBatchF64()
    s.n = 3
    assert lib.ste_ukf_forward_f64(C.byref(s), None) == -1
    assert b"n must be 4" in lib.ste_last_error()
    s.n, s.B = 4, 0
    assert lib.ste_ukf_urtss_f64(C.byref(s), None) == -1
    assert lib.ste_ukf_forward_f64(None, None) == -1
    assert lib.ste_geodetic_dynamics_f64(-1, None, None, None, None, None, None) == -1
    assert lib.ste_geodetic_dynamics_f64(0, None, None, None, None, None, None) == 0
    # observation preparation: NULL batch, empty batch, unknown model, missing arrays
    assert lib.ste_track_prep_f64(None, None) == -1
    pb = binding.StePrepBatchF64()
    assert lib.ste_track_prep_f64(C.byref(pb), None) == -1
    pb.B, pb.Tmax, pb.model = 4, 8, 7
    assert lib.ste_track_prep_f64(C.byref(pb), None) == -1 and b"model" in lib.ste_last_error()
    pb.model = binding.STE_PREP_WGS84
    assert lib.ste_track_prep_f64(C.byref(pb), None) == -1 and b"required" in lib.ste_last_error()
    # GP: NULL batch, subset launch without an index list
    assert lib.ste_gp_lml_f64(None, None) == -1
    assert lib.ste_gp_lml_subset_f64(None, 1, None, None) == -1 and b"active" in lib.ste_gp_last_error()


def test_no_gpu_fails_loudly():
    from track_estimators._hip import binding

    lib = binding.load()
    if lib.ste_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(binding.SteError):
        binding.require_gpu()
