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
    declared = set(re.findall(r"^(?:int|const char\*)\s+(ste_\w+)\s*\(", hdr, flags=re.M))
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
                          ("ste_prep_batch_f64", binding.StePrepBatchF64)):
        body = hdr[hdr.index("typedef struct %s {" % cname): hdr.index("} %s;" % cname)]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = re.findall(r"(?:const\s+)?(?:int32_t|uint32_t|double)\s*\*?\s*(\w+)\s*;", body)
        assert fields == [f[0] for f in mirror._fields_], cname
    assert C.sizeof(binding.SteUkfBatchF64) == 24 + 3 * 8 + 22 * 8 + 16


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
