#!/usr/bin/env python3
"""
bench.py — UKF + URTSS track-steps/s on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (forward UKF + unscented RTS smoother) over one batch of synthetic tracks that
is already resident in HBM.

Workload
  N = 1   BASELINE.json configs[1]: 10 000 synthetic dim-4 geodetic tracks x 500 steps, fp64 (126 observations at 1 h
          gaps, 4 sub-steps).
  N > 1   BASELINE.json configs[2]: ``--total-tracks`` (default 12 500 x N, i.e. 100 000 at N = 8) cut into contiguous
          track shards, one rank per GPU (launched by torch.distributed.run over RCCL); tracks are independent, so
          nothing is exchanged while filtering, and every step ends with the one exchange the path has: an all-gather
          of the smoothed lon/lat.  Per-GPU work is fixed as N grows ("weak").

Prints ONE JSON line on rank 0.  Extra objects:
  roofline      SURVEY.md section 8(d)'s figure: `achieved` = 512 algorithmic HBM bytes per track-step (192 forward + 320
                smoother) x the track-steps per second of the timed region, per GPU, against 8 TB/s (the forward and smoother
                kernels of different steps share the chip, so the path's rate is the job's rate); `traffic` = HBM bytes of one
                step (one forward + one smoother launch) from the committed rocprofv3 PMC summary under profiles/ (same kernel
                generation, same command); `per_launch` = one forward launch's 192 B per track-step over its own HIP-event
                duration, with the launches in flight beside it; `fp64_valu` = executed fp64 VALU work from the same summary,
                beside the nominal SURVEY.md estimate (labelled)
  cpu_baseline  the NumPy oracle timed on this box's host cores on bounded samples of the same batch: the vectorised
                restatement ("port"; one process, and one forked worker per core of the box's CPU share) and
                `reference_call_sequence`, the per-track restatement that issues the reference's own calls in its order
  serial        one batch at a time, forward then smoother on one unrestricted stream -- the latency figure

Steps are pipelined by default (track_estimators.batch.SmootherPipeline): every step runs the complete forward pass
and smoother of one batch, but several steps are in flight -- forward passes and the smoothers of the steps before them
share every compute unit (one forward wave per SIMD by construction, smoother waves beside them), each stream on a
hardware queue of its own, with one set of history buffers per step in flight plus one.  A short run on one GPU
(``--steps`` <= 40, ``--sequence auto``) is mostly fill and drain, and three launch forms of the same arithmetic (same bits)
are compared untimed first: one launch pair per step; the forward passes of the run as two scheduled launches of resident
waves over (tile, time slice) items with a smoother per step; ONE scheduled forward launch with ONE smoother launch of a
wave per tile -- the line says which form it timed and what each cost (``config.sequence_auto``).
"""
import argparse
import contextlib
import csv
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.join(ROOT, "ship-track-estimators_amd")
for p in (ROOT, PKG_ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

TRACKS_CONFIG1 = 10_000  # BASELINE.json configs[1]
TRACKS_PER_GPU_CONFIG2 = 12_500  # BASELINE.json configs[2]: 100 000 tracks over 8 GPUs
NOBS = 126
SUBSTEPS = 4
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6
# measured on MI355X this round (profiles/r03_fp64_clock_probe.log, profiles/r03_pipeline_ceilings.log): what a SIMD sustains
# in wave64 fp64 FMAs with the whole chip busy, and the counter traffic rate of the smoother kernel with the chip to itself
VALU_SUSTAINED_INSTS_PER_SIMD = 537e6
HBM_MEASURED_GBS = 4900.0
BYTES_FWD = 192  # algorithmic, per track-step: 32 B inputs + 160 B filtered mean/cov written
BYTES_BWD = 320  # algorithmic, per track-step: 160 B filtered history re-read + 160 B smoothed written
FLOPS_NOMINAL = 2.0e4  # SURVEY.md §8d estimate of the REFERENCE algorithm (fp64 flop-equivalents, forward + backward)
# rocprofv3 PMC summary of this command for the current kernel generation (see profiles/README.md for the passes)
COUNTERS_CSV = os.path.join(ROOT, "profiles", "r05_counters_per_track_step.csv")
EVENT_EVERY = 4  # steps between two steps whose kernels are bracketed by HIP events (see the timed loop)


def load_counters():
    """kernel-name prefix -> dict(hbm_read, hbm_write, fp64_flops) per track-step, from the committed PMC summary."""
    out = {}
    try:
        with open(COUNTERS_CSV) as f:
            for row in csv.DictReader(ln for ln in f if not ln.startswith("#")):
                vals = {}
                for k, v in row.items():
                    try:
                        vals[k] = float(v)
                    except (TypeError, ValueError):  # the kernel's device name and the like
                        pass
                out[row["kernel"]] = vals
    except OSError:
        pass
    return out


def cpu_baseline(ntracks: int, seed0: int = 0):
    """Time the oracle's vectorised restatement on `ntracks` tracks of the same workload (single process)."""
    from track_estimators import batch, synthetic

    from oracle import ukf_oracle as orc

    H, Q, R, P0 = synthetic.example_matrices()
    sb = synthetic.make_batch(ntracks, nobs=NOBS, gap_h=1.0, seed0=seed0)
    hb = batch.pack_uniform(sb, SUBSTEPS, H, Q, R, P0)
    fires = hb.upd_idx.T >= 0
    zidx = np.where(fires, hb.upd_idx.T, 0)
    ridx = np.cumsum(fires, axis=1) - fires
    rr = np.broadcast_to(batch.rts_rate_index(hb.Nmax + 1, NOBS - 1, NOBS), (hb.B, hb.Nmax))
    t0 = time.perf_counter()
    m, P = orc.forward_batch(hb.x0.T, P0, H, Q, R, hb.dt.T, fires, zidx, ridx, sb.z, sb.sog_rate, sb.cog_rate)
    sm, sP = orc.backward_batch(m, P, Q, hb.dt.T, rr, sb.sog_rate, sb.cog_rate)
    dt = time.perf_counter() - t0
    return hb.track_steps / dt, dt, (m, P, sm, sP), hb


def cpu_reference_call_sequence(ntracks: int, seed0: int = 0):
    """Time the oracle's per-track restatement -- the one that issues the reference's own call sequence
    (kalman_filter.py:61-117 and unscented.py:285-351: ``scipy.linalg.sqrtm`` per fan, a Python loop over the nine sigma
    points, ``np.linalg.pinv`` per gain; bit-exact against the reference on every golden case, tests/test_oracle_golden.py)
    -- on the first ``ntracks`` tracks of the bench batch, one track after the other like the loop of
    examples/example_ukf_rts_smoother_batch.py:19-90.  One process, one core."""
    from track_estimators import synthetic
    from track_estimators.utils import generate_dts

    from oracle import ukf_oracle as orc

    H, Q, R, P0 = synthetic.example_matrices()
    sb = synthetic.make_batch(ntracks, nobs=NOBS, gap_h=1.0, seed0=seed0)
    steps, out = 0, []
    t0 = time.perf_counter()
    for b in range(ntracks):
        dt = generate_dts(sb.dts[b], SUBSTEPS)
        m, P = orc.forward_track(sb.z[b][:, 0], P0, H, Q, R, dt, sb.dts[b], sb.z[b], sb.sog_rate[b], sb.cog_rate[b])
        sm, sP = orc.backward_track(m, P, Q, dt, len(sb.dts[b]), sb.sog_rate[b], sb.cog_rate[b])
        steps += len(dt)
        out.append((m, P, sm, sP))
    secs = time.perf_counter() - t0
    return steps / secs, secs, out


def _pool_worker(job):
    seed0, ntracks = job
    v, secs, _, hb = cpu_baseline(ntracks, seed0=seed0)
    return hb.track_steps, secs


def host_core_share() -> int:
    """Cores this process may really use: the affinity mask, cut by the cgroup CPU quota when there is one, and by 16 --
    the CPU share of a one-GPU box -- when the quota is not visible."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 16)


def cpu_baseline_pool(procs: int, tracks_per_proc: int):
    """The same oracle on every host core this process may use: `procs` forked workers, each timing its own
    `tracks_per_proc`-track sample (SURVEY.md §8d asks for the single-process and the all-cores figure).  Must run
    before anything initialises the GPU (fork after HIP init is not safe)."""
    import multiprocessing as mp

    ctx = mp.get_context("fork")
    with ctx.Pool(procs) as pool:
        res = pool.map(_pool_worker, [(10_000_000 + i * tracks_per_proc, tracks_per_proc) for i in range(procs)])
    steps = sum(r[0] for r in res)
    secs = max(r[1] for r in res)
    return steps / secs, secs


def _free_port() -> int:
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(ngpus: int) -> int:
    """Run this script under ``python -m torch.distributed.run`` with ``ngpus`` ranks on 127.0.0.1 and return its exit
    code; the child's stdout (rank 0's one JSON line) and stderr pass straight through."""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    print("[bench] --gpus %d without a launcher: starting %s" % (ngpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform

    return platform.processor() or platform.machine()


def compare_histories(gpu: dict, ref: dict):
    """The parity figures of the run: for each of the four histories the largest error in the north-star's measure (means:
    |d| / max(|ref|, 1e-12) element-wise; covariances: max|dP| / max|P| per matrix), where it sits (track, step, component)
    and the values there; headings additionally as the largest wrap-aware absolute difference in degrees, which does
    not blow up when a heading happens to pass through 0."""
    out = {}
    for name in ("means", "means_smoothed"):
        g, r = gpu[name], ref[name]
        rel = np.abs(g - r) / np.maximum(np.abs(r), 1e-12)
        i = np.unravel_index(int(np.argmax(rel)), rel.shape)
        dh = np.abs((g[..., 3] - r[..., 3] + 180.0) % 360.0 - 180.0)
        out[name] = {"max_rel_err": float(rel[i]), "at_track_step_component": [int(v) for v in i],
                     "gpu": float(g[i]), "oracle": float(r[i]),
                     "max_abs_err_position_deg": float(np.max(np.abs(g[..., :2] - r[..., :2]))),
                     "max_abs_err_heading_deg_wrap_aware": float(dh.max())}
    for name in ("covs", "covs_smoothed"):
        g, r = gpu[name], ref[name]
        rel = np.max(np.abs(g - r), axis=(-1, -2)) / np.max(np.abs(r), axis=(-1, -2))
        i = np.unravel_index(int(np.argmax(rel)), rel.shape)
        out[name] = {"max_rel_err_per_matrix": float(rel[i]), "at_track_step": [int(v) for v in i]}
    return out


def measure_config3(dev, reps: int = 3):
    """BASELINE.json configs[3]: every ship id of data/modern_ships (7 ragged tracks, 13 274 .. 19 236 filter steps at two
    sub-steps, variable dt, duplicate timestamps in five of them), Mahalanobis outlier rejection ON, one launch on one
    GPU.  Seven tracks are seven quads of ONE wave: the run is a latency figure (a chain of ~19 000 steps on a single
    SIMD), not a throughput one -- reported as such.  Observation preparation runs on the device (sphere pair)."""
    import gzip
    import shutil
    import tempfile

    import pandas as pd
    import torch
    from track_estimators import batch
    from track_estimators.ship_track import ShipTrack
    from track_estimators.utils import generate_dts, haversine_formula, heading

    src = os.path.join(ROOT, "tests", "golden", "data", "modern_ship_data.csv.gz")
    if not os.path.exists(src):
        return {"error": "tests/golden/data/modern_ship_data.csv.gz not present"}
    with tempfile.TemporaryDirectory() as td:
        csv_path = os.path.join(td, "modern.csv")
        with gzip.open(src, "rb") as a, open(csv_path, "wb") as b:
            shutil.copyfileobj(a, b)
        ids = [str(v) for v in pd.read_csv(csv_path)["id"].unique().tolist()]
        tracks = []
        with np.errstate(all="ignore"):
            for sid in ids:
                st = ShipTrack(calc_distance_func=haversine_formula, calc_heading_func=heading)
                st.read_csv(csv_path, ship_id=sid, id_col="id", lat_col="lat", lon_col="lon")
                tracks.append(st)
            t0 = time.perf_counter()
            batch.prepare_ship_tracks(tracks, device=dev)  # sog, cog, rates, z for all 58 693 observations in one launch
            t_prep = time.perf_counter() - t0
    H = np.diag([1.0, 1, 0, 0])
    R = np.diag([0.25, 0.25, 0, 0])
    Q = np.diag([1e-4, 1e-4, 1e-6, 1e-6])
    dts = [generate_dts(st.dts, 2) for st in tracks]
    hb = batch.pack_tracks(tracks, dts, [st.z[:, 0] for st in tracks], H, Q, R, np.eye(4))
    hb.robust = True
    db = batch.DeviceBatch(hb, device=dev)
    db.run()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        db.run()
    torch.cuda.synchronize(dev)
    secs = (time.perf_counter() - t0) / reps
    status = db.status_host()
    return {"workload": f"data/modern_ships, all {len(ids)} ship ids, {int(hb.nsteps.min())}..{int(hb.nsteps.max())} filter steps "
                        "(variable dt, 2 sub-steps), Mahalanobis outlier rejection on (BASELINE.json configs[3])",
            "seconds": secs, "track_steps": hb.track_steps, "value": hb.track_steps / secs, "unit": "track-steps/s",
            "prep_seconds_device": t_prep,
            "tracks_flagged_nan_or_index": int(((status | (hb.host_status if hb.host_status is not None else 0)) & 0x11 != 0).sum()),
            "note": "7 tracks = one quad per wave on 7 SIMDs: a latency chain of ~17 000 sequential steps (about "
                    f"{secs / max(int(hb.nsteps.max()), 1) * 1e6:.1f} us per step of the longest track, forward + smoother), not a "
                    "throughput figure; five of the seven ships carry "
                    "duplicate timestamps and end non-finite exactly where the reference raises LinAlgError"}


def measure_fleet(dev, ntracks: int = 100_000, chunk=None, reps: int = 3, sample: int = 48):
    """BASELINE.json configs[2]'s job on ONE GPU through the product's entry point: `ntracks` DISTINCT synthetic tracks x 500
    steps, resident in HBM, through batch.run_fleet (windows of one resident fleet through the pipelined kernels; results
    stay in the fleet's tensors) -- beside the same job as one DeviceBatch.run() (one forward launch of 1 563 waves on
    1 024 SIMDs, then one smoother launch).  The two must leave the same bits; a sample is checked against the oracle."""
    import torch
    from track_estimators import batch, synthetic

    from oracle import ukf_oracle as orc

    chunk = chunk or batch.FLEET_CHUNK
    H, Q, R, P0 = synthetic.example_matrices()
    t0 = time.perf_counter()
    sb = synthetic.make_batch(ntracks, nobs=NOBS, gap_h=1.0, seed0=50_000_000)
    hb = batch.pack_uniform(sb, SUBSTEPS, H, Q, R, P0)
    t_host = time.perf_counter() - t0
    hb.lanes = 1
    db = batch.DeviceBatch(hb, device=dev)
    torch.cuda.synchronize(dev)
    # one launch of everything
    db.run()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        db.run()
    torch.cuda.synchronize(dev)
    one_s = (time.perf_counter() - t0) / reps
    ref = [t.clone() for t in (db.fwd_mean, db.fwd_cov, db.sm_mean, db.sm_cov)]
    for t in (db.fwd_mean, db.fwd_cov, db.sm_mean, db.sm_cov):
        t.zero_()
    with batch.SmootherPipeline(dev, ntracks=chunk) as pipe:
        batch.run_fleet(db, chunk=chunk, pipeline=pipe)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            res = batch.run_fleet(db, chunk=chunk, pipeline=pipe)
        torch.cuda.synchronize(dev)
        fleet_s = (time.perf_counter() - t0) / reps
        desc = f"{len(pipe.fwd_streams)} forward + {len(pipe.bwd_streams)} smoother streams, {pipe.slices} time slice(s) per forward pass"
    same = all(torch.equal(a, b) for a, b in zip(ref, (db.fwd_mean, db.fwd_cov, db.sm_mean, db.sm_cov)))
    del ref
    n = min(sample, ntracks)
    fires = hb.upd_idx.T[:n] >= 0
    zidx = np.where(fires, hb.upd_idx.T[:n], 0)
    ridx = np.cumsum(fires, axis=1) - fires
    rr = np.broadcast_to(batch.rts_rate_index(hb.Nmax + 1, NOBS - 1, NOBS), (n, hb.Nmax))
    m, P = orc.forward_batch(hb.x0.T[:n], P0, H, Q, R, hb.dt.T[:n], fires, zidx, ridx, sb.z[:n], sb.sog_rate[:n], sb.cog_rate[:n])
    sm, sP = orc.backward_batch(m, P, Q, hb.dt.T[:n], rr, sb.sog_rate[:n], sb.cog_rate[:n])
    got = db.download(("means", "covs", "means_smoothed", "covs_smoothed"), torch.arange(n, device=dev))
    parity = compare_histories(got, {"means": m, "covs": P, "means_smoothed": sm, "covs_smoothed": sP})
    return {"workload": f"{ntracks} distinct synthetic tracks x {hb.Nmax} steps resident in HBM (BASELINE.json configs[2]'s job on one "
                        f"GPU) through batch.run_fleet: {len(batch.fleet_windows(ntracks, chunk))} windows of one resident fleet, {desc}; "
                        "results stay in the fleet's tensors",
            "seconds": fleet_s, "track_steps": hb.track_steps, "value": hb.track_steps / fleet_s, "unit": "track-steps/s",
            "one_launch": {"seconds": one_s, "value": hb.track_steps / one_s, "unit": "track-steps/s",
                           "note": "the same resident fleet as ONE DeviceBatch.run(): a forward launch of "
                                   f"{-(-ntracks // 64)} lane-per-track waves on 1 024 SIMDs, then one smoother launch"},
            "bit_identical_to_one_launch": bool(same),
            "flagged_tracks": int((res["status"] != 0).sum()),
            "host_synthesis_seconds": t_host,
            "gpu_vs_oracle": {"tracks": n, "means_max_rel_err": parity["means"]["max_rel_err"],
                              "means_smoothed_max_rel_err": parity["means_smoothed"]["max_rel_err"],
                              "covs_max_rel_err_per_matrix": parity["covs"]["max_rel_err_per_matrix"],
                              "covs_smoothed_max_rel_err_per_matrix": parity["covs_smoothed"]["max_rel_err_per_matrix"]}}


def main():
    # the pipeline's CU-masked streams are destroyed before the interpreter goes down, whatever happens in between
    with contextlib.ExitStack() as stack:
        _main(stack)


def _main(stack):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--total-tracks", type=int, default=None,
                    help="tracks of the whole job, cut into one contiguous shard per GPU (default: 10 000 at --gpus 1 = "
                         "BASELINE configs[1]; 12 500 x N otherwise = configs[2]'s shard size, 100 000 at --gpus 8)")
    ap.add_argument("--tracks", type=int, default=None, help="tracks per GPU (overrides --total-tracks; tests)")
    ap.add_argument("--cpu-tracks", type=int, default=3072, help="tracks in the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-ref-tracks", type=int, default=8,
                    help="tracks of the batch timed through the oracle's per-track form, the reference's own call sequence "
                         "(about 0.4 s per 500-step track; 0 = skip)")
    ap.add_argument("--lanes", type=int, default=0, help="forward-kernel lanes per track (0 = by batch size, 1, 4)")
    ap.add_argument("--no-gather", action="store_true", help="skip the all-gather of smoothed lon/lat when N>1")
    ap.add_argument("--gather-every", type=int, default=1,
                    help="N>1: run the all-gather of the smoothed lon/lat once per this many steps (default 1 = BASELINE "
                         "configs[2] as written, every step's batch is exchanged; K = --steps is one exchange per K-batch "
                         "fleet, what a real 100 000-track job needs)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run forward and smoother of every step back to back on one stream instead of overlapping the "
                         "smoothers of earlier steps with the forward passes of later ones on disjoint CU partitions")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group and run the gather even with one rank (rehearsal of the N>1 path)")
    ap.add_argument("--cpu-pool-tracks", type=int, default=768,
                    help="tracks per worker in the all-cores CPU-baseline leg (0 = skip)")
    # experiments (defaults are the measured best)
    ap.add_argument("--forward-cus", type=int, default=None)
    ap.add_argument("--forward-streams", type=int, default=None)
    ap.add_argument("--smoother-streams", type=int, default=None)
    ap.add_argument("--forward-lanes", type=int, default=None, help="lane mapping of the pipelined forward passes (1, 4, 0)")
    ap.add_argument("--reserve-cus", type=int, default=None,
                    help="compute units the pipeline's streams leave free (for the collective's kernels; default 0)")
    ap.add_argument("--partition", action="store_true",
                    help="round-2 pipeline: forward passes and smoothers on disjoint CU partitions (default: they share every CU)")
    ap.add_argument("--tuning", type=lambda v: int(v, 0), default=0, help="ste_ukf_batch_f64.tuning")
    ap.add_argument("--full-cov", action="store_true",
                    help="keep the covariance histories as full 4x4 matrices in HBM (default: packed upper triangles)")
    ap.add_argument("--slices", type=int, default=None,
                    help="time slices per pipelined forward pass (default: batch.DEFAULT_SLICES)")
    ap.add_argument("--sequence", default="auto",
                    help="steps per SCHEDULED forward launch (SmootherPipeline.submit_sequence: the forward passes of that many "
                         "steps as one launch of resident waves over (tile, time slice) items): an integer, 0 = one launch per "
                         "step, or 'auto'.  auto, one GPU and --steps <= 40 (a short run is all fill and drain, which a schedule "
                         "shortens): two scheduled launches -- the first of as many steps as fill the chip once --, one scheduled "
                         "launch with one tile-smoother launch, and one launch per step are all run untimed, each on the pipeline it "
                         "needs (2 + 6 + 1, 1 + 1 + 1 and 7 + 6 + 1 streams; three runs, the slowest counts), and the fastest form is used (config.sequence_auto says which: 0.74-0.77 "
                         "against 0.80 ms per step at the driver's 20 steps); longer runs and runs with an exchange: one launch "
                         "per step (in the steady state per-step launches re-balance by themselves and are 1-2 %% faster; "
                         "DESIGN.md section 5)")
    ap.add_argument("--sequence-form", default="auto", choices=("auto", "per_step", "split", "single"),
                    help="with --sequence auto: run the untimed comparison of the three launch forms (auto), or pin one of them "
                         "without comparing (profiling passes)")
    ap.add_argument("--no-fleet", action="store_true", help="skip the `extra.fleet_100k` entry (100 000 distinct tracks through batch.run_fleet)")
    ap.add_argument("--fleet-tracks", type=int, default=100_000)
    ap.add_argument("--fleet-chunk", type=int, default=None, help="window size of the fleet entry (default: batch.FLEET_CHUNK)")
    ap.add_argument("--no-gp", action="store_true",
                    help="skip the `extra.gp_config4` entry (BASELINE configs[4]: one batched GP objective at 1000 x 2000, "
                         "measured after the timed region at --gpus 1)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks as a child job (one process per GPU) before anything in
        # this process touches the GPU, relay rank 0's JSON line and leave with the child's exit code.  (Never an exec:
        # the launcher is an ordinary child, this process stays a plain CPU parent.)
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    pool_result = None
    if world == 1 and args.gpus == 1 and args.cpu_tracks > 0 and args.cpu_pool_tracks > 0:
        # all-cores leg of the CPU baseline: forked workers, so it runs before this process touches the GPU
        procs = host_core_share()
        pool_result = (procs,) + cpu_baseline_pool(procs, args.cpu_pool_tracks)

    import torch

    from track_estimators import batch, distributed, synthetic
    from track_estimators._hip import binding

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    dist = nccl_log = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # STE_BENCH_BACKEND=gloo lets several ranks share one GPU for a rehearsal of the N>1 control flow (RCCL refuses
        # two ranks on one device); the driver's multi-GPU runs use the default, nccl = RCCL over xGMI.
        backend = os.environ.get("STE_BENCH_BACKEND", "nccl")
        if backend == "nccl" and os.environ.get("STE_BENCH_KEEP_NCCL_ENV") != "1":
            # which algorithm / protocol RCCL picks for the exchange: its TUNING log lines, into a file per rank (stdout
            # carries the one JSON line), read back after the untimed all_gather_alone leg.  A caller's own NCCL_DEBUG settings
            # are replaced for this process (STE_BENCH_KEEP_NCCL_ENV=1 keeps them and drops the `rccl` object)
            import tempfile

            nccl_log = os.path.join(tempfile.gettempdir(), f"ste_bench_rccl_{os.getpid()}.log")
            os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="TUNING,COLL", NCCL_DEBUG_FILE=nccl_log)
            stack.callback(lambda: os.path.exists(nccl_log) and os.unlink(nccl_log))
        ndev = torch.cuda.device_count()
        local_rank = local_rank % max(ndev, 1)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        stack.callback(dist.destroy_process_group)
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    binding.require_gpu()

    # --- this rank's shard of the synthetic job, resident in HBM before the timed region ------------------------
    if args.tracks is not None:
        total = args.tracks * world
    elif args.total_tracks is not None:
        total = args.total_tracks
    else:
        total = TRACKS_CONFIG1 if world == 1 else TRACKS_PER_GPU_CONFIG2 * world
    lo, hi = distributed.shard_bounds(total, rank, world)
    B = hi - lo
    bmax = -(-total // world)  # largest shard: what the gathered tensor is padded to
    H, Q, R, P0 = synthetic.example_matrices()
    sb = synthetic.make_batch(B, nobs=NOBS, gap_h=1.0, seed0=lo)  # seed = global track index
    hb = batch.pack_uniform(sb, SUBSTEPS, H, Q, R, P0)
    hb.lanes = args.lanes
    want_gather = (world > 1 or args.force_dist) and not args.no_gather
    mk = lambda: batch.DeviceBatch(hb, device=dev, tuning=args.tuning, packed_cov=not args.full_cov, sm_pos=want_gather)  # noqa: E731
    db = mk()
    pipe = None
    # With a collective in the step its kernels (RCCL's, and the snapshot of the send buffer) need somewhere to run: on CUs they
    # share with forward and smoother waves they displace those (one RCCL rank on this box, 12 500 tracks: 6.2e9 track-steps/s);
    # with 32 of the 256 CUs -- four per XCD -- kept free of the pipeline's streams: 6.8e9 (24: 6.7, 40: 6.4, 48: 6.1).
    reserve_cus = args.reserve_cus
    if reserve_cus is None and dist is not None and not args.no_gather and not args.partition:
        reserve_cus = 32
    if not args.no_pipeline:
        try:
            kw = {k: v for k, v in (("forward_cus", args.forward_cus), ("forward_streams", args.forward_streams),
                                    ("smoother_streams", args.smoother_streams),
                                    ("forward_lanes", args.forward_lanes), ("shared", False if args.partition else None),
                                    ("reserve_cus", reserve_cus), ("slices", args.slices)) if v is not None}
            pipe = stack.enter_context(batch.SmootherPipeline(dev, ntracks=bmax, **kw))
        except (binding.SteError, ValueError) as exc:  # no CU-masked streams here, or a batch too large to partition
            print(f"[bench] pipelining disabled, steps run back to back: {exc}", file=sys.stderr, flush=True)
    seq = 0
    if pipe is not None and args.lanes != 4 and (args.forward_lanes in (None, 1)):
        # auto: single-GPU runs only (with an exchange in the step the per-step launches are the measured configuration)
        seq = (10 if (args.steps <= 40 and dist is None) else 0) if args.sequence == "auto" else int(args.sequence)
    # how a run of n steps is cut into scheduled launches.  auto: a first launch of as many steps as fill the chip's SIMDs once
    # (7 at 10 000 tracks: its windows finish together after one pass and their smoothers start while the second launch, the
    # rest, runs) -- measured at the driver's 20 steps: (7, 13) 14.8 ms, (8, 12) 15.0, (10, 10) 15.3, (20) 15.3, three launches
    # 16.5+, one launch per step 16.1 (profiles/r05_scheduled_forward.txt)
    first_fill = max(1, -(-1024 // max(1, -(-B // 64))))

    plan_single = False  # auto's second scheduled form: ONE launch for the run, its smoothers as one launch of waves per tile

    def seq_plan(n):
        if not seq or n <= 0:
            return []
        if args.sequence == "auto":
            return [n] if (n <= first_fill + 1 or plan_single) else [first_fill, n - first_fill]
        return [min(seq, n - c0) for c0 in range(0, n, seq)]

    # per-step launches rotate through buffers_needed sets; a scheduled launch needs a set per step of the sequence, and two
    # sequences' worth so that the next sequence does not wait for the smoothers of the one before it
    nsets = 1 if pipe is None else max(pipe.buffers_needed, (max(args.steps, args.warmup) if args.sequence == "auto" else
                                                             min(2 * seq, max(args.steps, args.warmup))) if seq else 0)
    dbs = [db] + [mk() for _ in range(nsets - 1)]
    gathered = None
    if dist is not None and not args.no_gather:
        # the one exchange of the path: all-gather of the smoothed lon/lat, overlapped with the next step's kernels
        gathered = distributed.OverlappedGather(hb.Nmax + 1, bmax, dev)

    stream = torch.cuda.current_stream(dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731

    def serial_step(d, events=None, exchange=True):
        if events is not None:
            events[0].record(stream)
        d.forward(stream)
        if events is not None:
            events[1].record(stream)
            events[2].record(stream)
        d.backward(stream)
        if events is not None:
            events[3].record(stream)
        if gathered is not None and exchange:
            gathered.launch(d.sm_pos)

    if args.gather_every < 1:
        raise SystemExit("--gather-every must be >= 1")

    def one_step(k, events=None, final=False):
        d = dbs[k % len(dbs)]
        exchange = gathered is not None and (k + 1) % args.gather_every == 0
        if pipe is None:
            serial_step(d, events, exchange)
        else:
            # the exchange sends the smoother's own output (sm_pos): no snapshot; the event it returns keeps the next use of
            # this buffer set behind the collective that reads it
            pipe.submit(d, after_smoother=(lambda _s: gathered.launch_for_pipeline(d.sm_pos)) if exchange else None,
                        timing=events, final=final)

    seq_timings = []

    def run_steps(nsteps, events=None):
        """`nsteps` steps of the hot path: one launch pair per step, or -- `seq` > 0 -- the forward passes of every `seq`
        consecutive steps as one scheduled launch (each step still filters and smooths its own batch into its own buffers)."""
        if not seq:
            for k in range(nsteps):
                one_step(k, None if events is None else events[k], final=(k == nsteps - 1))
            return
        c0 = 0
        for n in seq_plan(nsteps):
            ks = list(range(c0, c0 + n))
            c0 += n
            tm = {"every": EVENT_EVERY} if events is not None else None
            hook = None
            if gathered is not None:
                def hook(i, _s, ks=ks):  # noqa: E306
                    if (ks[i] + 1) % args.gather_every == 0:
                        return gathered.launch_for_pipeline(dbs[ks[i] % len(dbs)].sm_pos)
                    return None
            h0 = time.perf_counter()
            pipe.submit_sequence([dbs[k % len(dbs)] for k in ks], after_smoother=hook, timing=tm, final=(ks[-1] == nsteps - 1))
            if tm is not None:
                tm["host_ms"] = (time.perf_counter() - h0) * 1e3  # the call queues work and returns: this is host time only
                seq_timings.append((len(ks), tm))

    def drain():
        if pipe is not None:
            pipe.synchronize()
        if gathered is not None:
            gathered.finish()
        torch.cuda.synchronize(dev)

    # one batch at a time (not the timed region): forward + smoother of one batch on one unrestricted stream
    serial_ms = serial_kernels = None
    if pipe is not None and rank == 0:
        saved, gathered = gathered, None
        serial_step(db)
        torch.cuda.synchronize(dev)
        sev = [[ev(), ev(), ev(), ev()] for _ in range(3)]
        ts = time.perf_counter()
        for i in range(3):
            serial_step(db, sev[i])
        torch.cuda.synchronize(dev)
        serial_ms = (time.perf_counter() - ts) / 3 * 1e3
        serial_kernels = {"ukf_forward": float(np.mean([e[0].elapsed_time(e[1]) for e in sev])),
                          "urtss_backward": float(np.mean([e[2].elapsed_time(e[3]) for e in sev]))}
        gathered = saved

    # Untimed pre-pass: every set of history buffers of the rotation goes through the path once, so that nothing the
    # timed region uses is touched for the first time inside it (the driver's run has fewer warm-up steps than sets).
    prepass = len(dbs) if pipe is not None else 0
    try:
        run_steps(prepass)
        drain()
    except binding.SteError as exc:
        # harness guard, untimed part only: a scheduled launch that could not progress on this box (its waits are bounded and
        # report instead of hanging) must not cost the run its line -- fall back to one launch per step and say so
        if not seq:
            raise
        print(f"[bench] scheduled forward launches disabled after an error in the untimed pre-pass: {exc}", file=sys.stderr, flush=True)
        seq = 0
        torch.cuda.synchronize(dev)
        run_steps(prepass)
        drain()
    # auto: the two launch forms give the same bits, so the choice between them is made by the clock, untimed, on this box now
    # (three runs each): scheduled launches -- a long-lived grid beside one-wave gates that never leave their
    # queues idle -- are the first to suffer when the device runs out of hardware queues (this pipeline holds 14 of about 24;
    # on three boxes of round 5 something else on the device held the rest for a while, and scheduled launches took twice
    # their time while per-step launches kept theirs: DESIGN.md section 5)
    auto_choice = None

    def wall(n):
        torch.cuda.synchronize(dev)
        tw = time.perf_counter()
        run_steps(n)
        drain()
        return (time.perf_counter() - tw) * 1e3

    if seq and args.sequence == "auto" and args.sequence_form != "auto":
        # the comparison pinned to one form (profiling passes: kernel statistics of one form)
        if args.sequence_form == "per_step":
            seq = 0
        else:
            plan_single = args.sequence_form == "single" and len(seq_plan(args.steps)) > 1
            pipe.shrink(1, 1) if plan_single else pipe.shrink(min(2, len(pipe.fwd_streams)), len(pipe.bwd_streams))
        wall(prepass)
        auto_choice = {"chosen": ("single_launch" if plan_single else "scheduled") if seq else "per_step", "steps": args.steps,
                       "streams": f"{len(pipe.fwd_streams)} forward + {len(pipe.bwd_streams)} smoother + 1",
                       "note": f"--sequence-form {args.sequence_form}: no comparison was run"}
    elif seq and args.sequence == "auto":
        # Each form on the streams it needs: per-step launches on the 7 + 6 + 1 streams built above; then the pipeline is
        # shrunk -- streams destroyed, none created -- to 2 + 6 + 1 for the split scheduled form and to 1 + 1 + 1 for the single
        # one, so that each holds only the hardware queues it uses while it runs.
        # (decided by the SLOWEST of three runs of either form: at the edge of the device's queue slots a form is fast in
        #  one run and twice as slow in the next, and the timed region is one run)
        keep, seq = seq, 0
        wall(args.steps)  # (this form's first launches on these streams)
        step_all = sorted(wall(args.steps) for _ in range(3))
        step_ms = step_all[-1]
        pipe.shrink(min(2, len(pipe.fwd_streams)), len(pipe.bwd_streams))  # (queues handed back, none created)
        seq = keep
        try:
            wall(args.steps)  # (this pipeline's first launches: schedules, workspaces)
            sched_all = sorted(wall(args.steps) for _ in range(3))
            sched_ms = sched_all[-1]
        except binding.SteError as exc:
            print(f"[bench] scheduled forward launches disabled after an error in the untimed comparison: {exc}", file=sys.stderr, flush=True)
            torch.cuda.synchronize(dev)
            sched_all, sched_ms = None, float("inf")
        # third form: the whole run as ONE scheduled launch with its smoothers as one launch of waves per tile -- within 1 % of
        # the split form on a normal box, and it runs on ONE forward and ONE smoother stream: three hardware queues
        single_all, single_ms = None, float("inf")
        if len(seq_plan(args.steps)) > 1:
            pipe.shrink(1, 1)
            plan_single = True
            try:
                wall(args.steps)
                single_all = sorted(wall(args.steps) for _ in range(3))
                single_ms = single_all[-1]
            except binding.SteError as exc:
                print(f"[bench] single scheduled launch disabled after an error in the untimed comparison: {exc}", file=sys.stderr, flush=True)
                torch.cuda.synchronize(dev)
                single_all, single_ms = None, float("inf")
        best = min(step_ms, sched_ms, single_ms)
        # the pipeline is down to the single form's three streams by now: going back to another form means building streams --
        # new hardware queues -- again, and a pipeline built behind two destroyed ones ran 12 % slower than the one it replaces
        # (0.83-0.86 against 0.734-0.741 ms per step, gpurun_out r5_bb); the single form stays unless another is 2 % faster
        if np.isfinite(single_ms) and single_ms <= 1.02 * best:
            best = single_ms
        if best == single_ms and np.isfinite(single_ms):
            pass
        elif best == sched_ms and np.isfinite(sched_ms):
            if plan_single:  # back to the split form's pipeline
                plan_single = False
                pipe.close()
                pipe = stack.enter_context(batch.SmootherPipeline(dev, ntracks=bmax, **dict(kw, sequence_only=True)))
                for _ in range(3):
                    wall(args.steps)
        elif best == step_ms:
            seq, plan_single = 0, False
            pipe.close()
            pipe = stack.enter_context(batch.SmootherPipeline(dev, ntracks=bmax, **kw))
            for _ in range(3):
                wall(prepass)
        auto_choice = {"scheduled_launches_ms": sched_ms if np.isfinite(sched_ms) else None, "per_step_launches_ms": step_ms,
                       "single_launch_ms": single_ms if np.isfinite(single_ms) else None,
                       "scheduled_launches_ms_all": sched_all, "per_step_launches_ms_all": step_all, "single_launch_ms_all": single_all,
                       "steps": args.steps, "chosen": ("single_launch" if plan_single else "scheduled") if seq else "per_step",
                       "streams": f"{len(pipe.fwd_streams)} forward + {len(pipe.bwd_streams)} smoother + 1",
                       "note": "untimed, before the warm-up: wall time of --steps steps in each launch form -- one launch per "
                               "step (7 + 6 + 1 streams), scheduled launches of first-fill + rest with a smoother per step "
                               "(2 + 6 + 1), one scheduled launch with one tile-smoother launch (1 + 1 + 1) --, each on the "
                               "pipeline it needs, three runs each, the slowest counts; the timed region uses the fastest "
                               "form -- all give the same histories bit for bit"}
    run_steps(args.warmup)
    drain()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    # HIP events around the kernels of every fourth step (launch durations for the roofline object, completion marks for the
    # steady state): each record is a packet on the stream's hardware queue between two launches, and with four of them on
    # every step the pipeline ran 2 % slower (7.67 against 7.83e9 track-steps/s; no events at all: 7.84e9)
    every = EVENT_EVERY
    evs = [[ev(), ev(), ev(), ev()] if k % every == 0 else None for k in range(args.steps)]
    t0 = time.perf_counter()
    run_steps(args.steps, evs)  # (the last smoother has nothing to hide behind: it gets the whole chip)
    drain()  # every step's smoother (and gathered result) has landed before the clock stops
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # The exchange on its own (not the timed region): the same all-gather with nothing beside it, so that a step time can be
    # read as max(filter, exchange) -- at the filter's speed the 16 B per track-step every rank sends to every other rank
    # is the larger term on xGMI.
    gather_alone = None
    if gathered is not None:
        torch.cuda.synchronize(dev)
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(3):
            gathered.launch(db.sm_pos)
            gathered.finish()
            torch.cuda.synchronize(dev)
        tg = torch.tensor([(time.perf_counter() - t1) / 3], dtype=torch.float64, device=dev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        out_bytes = int(np.prod(gathered.shape)) * 8
        gather_alone = {"ms": float(tg.item()) * 1e3, "bytes_sent_per_rank": out_bytes,
                        "bytes_received_per_rank": out_bytes * (world - 1),
                        "busbw_GBps": out_bytes * (world - 1) / max(float(tg.item()), 1e-12) / 1e9,
                        "note": "one all-gather of the smoothed lon / lat of a step ([N+1][2][tracks per rank] fp64 from every "
                                "rank to every rank) with no kernels beside it, blocking, slowest rank; in the timed region "
                                "it is asynchronous and double-buffered under the following steps"}

    rccl_choice = None
    if gather_alone is not None and nccl_log is not None and rank == 0:
        # RCCL's own words about the collective it just ran (TUNING: "... Bytes -> Algo ... proto ..."; COLL: the call)
        try:
            with open(nccl_log, errors="replace") as f:
                lines = [ln.strip() for ln in f if "AllGather" in ln or "Algo" in ln]
            tun = [ln.split("NCCL INFO", 1)[-1].strip() for ln in lines if "Algo" in ln]
            rccl_choice = {"tuning_lines": sorted(set(tun))[-4:] or None, "allgather_calls_logged": sum("AllGather" in ln for ln in lines),
                           "note": "NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=TUNING,COLL of this process, rank 0 (a communicator of one "
                                   "rank copies locally and logs no algorithm)"}
        except OSError as exc:
            rccl_choice = {"error": str(exc)}
    if gather_alone is not None:
        gather_alone["rccl"] = rccl_choice

    # The same timed loop with the exchange switched off (not the timed region): one run then reads as filter against
    # exchange -- `value` has the gather in every step, `filter_only` does not, `all_gather_alone` is the gather by itself.
    filter_only = None
    if gathered is not None:
        saved, gathered = gathered, None
        torch.cuda.synchronize(dev)
        dist.barrier()
        t1 = time.perf_counter()
        run_steps(args.steps)
        drain()
        dist.barrier()
        torch.cuda.synchronize(dev)
        tf = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        per_rank = [torch.zeros_like(tf) for _ in range(world)]
        dist.all_gather(per_rank, tf)
        dist.all_reduce(tf, op=dist.ReduceOp.MAX)
        gathered = saved
        filter_only = {"ms_per_step": float(tf.item()) / args.steps * 1e3,
                       "value": total * int(hb.Nmax) * args.steps / float(tf.item()), "unit": "track-steps/s",
                       "per_rank_ms_per_step": [float(t.item()) / args.steps * 1e3 for t in per_rank],
                       "note": "the timed loop again, same pipeline (same CUs reserved), without the all-gather: what the "
                               "exchange costs is value against this"}

    if seq_timings:  # scheduled launches: one forward kernel per sequence, a smoother per step
        fwd_ms = float(np.mean([tm["forward"][0].elapsed_time(tm["forward"][1]) for _n, tm in seq_timings]))
        bwd_ms = float(np.mean([a.elapsed_time(b) for _n, tm in seq_timings for a, b in tm["smoothers"]]))
        steps_per_launch = float(np.mean([n for n, _tm in seq_timings]))
    else:
        fwd_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs if e is not None]))
        bwd_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in evs if e is not None]))
        steps_per_launch = 1.0
    # Steady state of the pipeline: completed steps between two completion events (end of a step's smoother), leaving out
    # the steps during which the pipeline fills (no smoother has anything to do yet) and drains (no forward pass left).
    steady = None
    if pipe is not None and not seq:
        skip = len(pipe.fwd_streams) + 1
        # completions come in bursts (forward passes run in generations of as many launches as fill the chip), so the
        # window has to span several of them: at the driver's K = 20 there is no such window and the object is null
        if args.steps - 1 - 2 * skip >= 3 * len(pipe.fwd_streams) + 2 * every:
            a, b = -(-skip // every) * every, (args.steps - 1 - skip) // every * every  # steps that carry events
            span_ms = evs[a][3].elapsed_time(evs[b][3])
            steady = {"ms_per_step": span_ms / (b - a), "value": hb.track_steps * world * (b - a) / (span_ms * 1e-3),
                      "unit": "track-steps/s", "steps": b - a,
                      "note": f"steps {a + 1}..{b} of the timed region, between the completion events of steps {a} and {b} "
                              "(HIP events after each step's smoother): the rate the pipeline sustains once it is full; "
                              "`value` above includes its fill and drain"}
    status = db.status_host()
    for d in dbs[1:]:
        status = status | d.status_host()
    steps_per_track = int(hb.Nmax)
    total_units = total * steps_per_track * args.steps  # every rank's shard, every timed step
    value = total_units / elapsed

    if rank == 0:
        track_steps_rank = hb.track_steps
        counters = load_counters()
        cf = counters.get("ukf_forward", {})
        cb = counters.get("urtss_backward", {})
        # dominant kernel: the forward filter (it holds its partition for the whole step; the smoother hides behind it)
        achieved = BYTES_FWD * track_steps_rank * steps_per_launch / (fwd_ms * 1e-3) / 1e9
        per_s = track_steps_rank * args.steps / elapsed  # this rank's rate over the timed region
        traffic_fwd = (cf["hbm_read"] + cf["hbm_write"]) * track_steps_rank if "hbm_read" in cf else None
        traffic_all = None
        if "hbm_read" in cf and "hbm_read" in cb:
            traffic_all = cf["hbm_read"] + cf["hbm_write"] + cb["hbm_read"] + cb["hbm_write"]
        flops_exec = (cf.get("fp64_flops", 0.0) + cb.get("fp64_flops", 0.0)) or None
        valu_insts = (cf.get("valu_insts_per_wave_step", 0.0) + cb.get("valu_insts_per_wave_step", 0.0)) or None
        out = {
            "metric": "UKF+URTSS track-steps/sec (dim=4, 500-step tracks)",
            "value": value,
            "unit": "track-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (f"{total} synthetic dim-4 geodetic tracks x {steps_per_track} steps, UKF+URTSS, fp64 "
                             + ("(BASELINE.json configs[1])" if world == 1 and total == TRACKS_CONFIG1 else
                                f"({'BASELINE.json configs[2]' if total == 100_000 and world == 8 else 'configs[2] shard size'}: "
                                f"{bmax} tracks per GPU on {world} GPU(s))")
                             + "; zero injected noise; inputs resident in HBM"),
                "total_tracks": total, "tracks_per_gpu": bmax, "steps_per_track": steps_per_track, "observations": NOBS,
                "substeps": SUBSTEPS,
                "parallelism": f"track-sharded x{world}" + (f", {'RCCL' if os.environ.get('STE_BENCH_BACKEND', 'nccl') == 'nccl' else os.environ['STE_BENCH_BACKEND']} all-gather of smoothed lon/lat {'of every step' if args.gather_every == 1 else f'once per {args.gather_every} steps'}, overlapped with the following steps" if gathered is not None else ""),
                "pipeline": ("none: forward and smoother of a step back to back on one stream" if pipe is None else
                             (f"the forward passes of {' + '.join(str(n) for n in seq_plan(args.steps))} steps as ONE scheduled launch each of {4 * (pipe.forward_cus - pipe.reserve_cus)} resident "
                              "waves over (64-track tile, 64-step slice) items (SmootherPipeline.submit_sequence; bit-identical to "
                              "per-step launches), " + ("the smoothers of all steps as ONE launch of a wave per 64-track tile that waits for "
                                                        "its own tile's forward pass (ste_urtss_backward_sched_f64), " if plan_single else
                                                        f"each step's smoother behind a gate on one of {len(pipe.bwd_streams)} smoother streams, ")
                              + f"{len(dbs)} sets of histories in rotation") if seq else
                             f"{len(pipe.fwd_streams)} forward passes ({'lane' if pipe.forward_lanes == 1 and not args.lanes else 'quad' if (pipe.forward_lanes == 4 or args.lanes == 4) else 'lane' if args.lanes == 1 else 'auto'}-per-track) in flight "
                             + (f"beside {len(pipe.bwd_streams)} smoothers, all sharing {pipe.forward_cus - pipe.reserve_cus} of the {pipe.forward_cus} CUs "
                                + (f"({pipe.reserve_cus} left to the collective's kernels) " if pipe.reserve_cus else "") +
                                "(one forward wave per SIMD by construction, smoother waves beside them; one hardware queue per stream, "
                                if pipe.shared else
                                f"on {pipe.forward_cus} CUs beside {len(pipe.bwd_streams)} smoothers on the other {pipe.smoother_cus} CUs (CU-masked streams, ")
                             + f"{len(dbs)} sets of histories in rotation)"),
                "lanes_per_track": args.lanes, "tuning": args.tuning, "untimed_prepass_steps": prepass,
                "steps_per_scheduled_forward_launch": seq_plan(args.steps) or 0,
                "sequence_auto": auto_choice,
                "gather_every": args.gather_every if gathered is not None else None,
            },
            "kernels_ms": {"ukf_forward": fwd_ms, "urtss_backward": bwd_ms,
                           "timed_launches": len(seq_timings) if seq else len([e for e in evs if e is not None]),
                           "steps_per_forward_launch": steps_per_launch,
                           "scheduled_launches": ([{"steps": n, "forward_ms": tm["forward"][0].elapsed_time(tm["forward"][1]),
                                                    "host_submit_ms": tm["host_ms"]} for n, tm in seq_timings] or None),
                           "note": ("HIP events around every scheduled forward launch (the forward passes of "
                                    f"{steps_per_launch:g} steps each) and around "
                                    + ("the ONE smoother launch of the sequence (its duration includes its waves' waits for their tiles)"
                                       if any(tm.get("tile_smoothers") for _n, tm in seq_timings) else f"the smoother of every {EVENT_EVERY}th step") if seq else
                                    f"HIP events around the forward and smoother kernels of every {EVENT_EVERY}th step of the timed region")},
            "steady_state": steady,
            "all_gather_alone": gather_alone,
            "filter_only": filter_only,
            "serial": None if serial_ms is None else {
                "ms_per_step": serial_ms, "value": track_steps_rank / (serial_ms * 1e-3), "unit": "track-steps/s",
                "kernels_ms": serial_kernels,
                "note": "one batch at a time, forward then smoother on one unrestricted stream (per GPU, no overlap "
                        "between steps): the latency figure"},
            "status_flagged_tracks": int((status != 0).sum()),
            "roofline": {
                # achieved = algorithmic bytes per track-step (192 forward + 320 smoother) x the
                # track-steps per second of the timed region, per GPU, against the 8 TB/s HBM peak.  The forward and smoother
                # kernels of different steps share the chip in pipelined mode, so the path's rate is the whole-job rate.
                "bound": "hbm", "kernel": "ukf_forward + urtss_recur (the path; they overlap in pipelined mode)",
                "achieved": (BYTES_FWD + BYTES_BWD) * per_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (BYTES_FWD + BYTES_BWD) * per_s / 1e9 / HBM_PEAK_GBS,
                "traffic": None if traffic_all is None else traffic_all * track_steps_rank,
                "traffic_unit": "bytes per step (one forward + one smoother launch)",
                "traffic_source": (os.path.relpath(COUNTERS_CSV, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in "
                                   "separate passes of this command; FETCH_SIZE doubled per the gfx950 calibration in "
                                   "profiles/README.md)") if traffic_all is not None else None,
                "algorithmic_bytes_per_track_step": BYTES_FWD + BYTES_BWD,
                "algorithmic_bytes_per_step": (BYTES_FWD + BYTES_BWD) * track_steps_rank,
                "traffic_bytes_per_track_step": traffic_all,
                "traffic_rate": None if traffic_all is None else traffic_all * per_s / 1e9,
                "frac_of_measured_roof": None if traffic_all is None else traffic_all * per_s / 1e9 / HBM_MEASURED_GBS,
                "measured_roof": HBM_MEASURED_GBS,
                "measured_roof_note": "counter traffic of the smoother kernel alone on five streams (profiles/r03_pipeline_ceilings.log)",
                # one launch of the dominant kernel over ITS duration (HIP events on its stream): in the pipelined run
                # `launches_in_flight` forward launches and the smoothers of earlier steps share every CU for that duration
                "per_launch": {
                    "kernel": "ukf_forward", "algorithmic_bytes_per_track_step": BYTES_FWD, "launch_ms": fwd_ms,
                    "time_slices_per_launch": 1 if pipe is None else pipe.slices,
                    "achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "unit": "GB/s",
                    "traffic": traffic_fwd, "traffic_unit": "bytes per launch",
                    "steps_per_launch": steps_per_launch,
                    "launches_in_flight": 1 if (pipe is None or seq) else len(pipe.fwd_streams),
                    "aggregate": None if (pipe is None or seq) else {
                        "achieved": achieved * len(pipe.fwd_streams), "frac": achieved * len(pipe.fwd_streams) / HBM_PEAK_GBS,
                        "unit": "GB/s", "note": "algorithmic bytes of the forward launches in flight together over one launch duration"},
                    "alone": None if "alone_ms" not in cf else {
                        "launch_ms": cf["alone_ms"], "achieved": BYTES_FWD * track_steps_rank / (cf["alone_ms"] * 1e-3) / 1e9,
                        "frac": BYTES_FWD * track_steps_rank / (cf["alone_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "note": "the same kernel with the chip to itself (rocprofv3 counter passes serialise kernels; "
                                + os.path.relpath(COUNTERS_CSV, ROOT) + ")"}},
                "note": "the path is bound by fp64 issue and by HBM together (DESIGN.md section 5): frac_of_measured_roof "
                        "and fp64_valu.executed.frac_of_sustained are the two utilisations",
                "fp64_valu": {
                    "executed": None if flops_exec is None else {
                        "flops_per_track_step": flops_exec, "achieved": flops_exec * per_s / 1e12,
                        "frac": flops_exec * per_s / 1e12 / FP64_VALU_PEAK_TFLOPS,
                        "valu_insts_per_wave_step": valu_insts,
                        "frac_of_sustained": None if valu_insts is None else
                        valu_insts * per_s / 64.0 / (1024 * VALU_SUSTAINED_INSTS_PER_SIMD),
                        "sustained_note": "vector instructions issued / what 1 024 SIMDs sustain on fp64 FMA chains "
                                          "(537 M wave-instructions/s each = 70.4 TFLOP/s, profiles/r03_fp64_clock_probe.log)",
                        "source": os.path.relpath(COUNTERS_CSV, ROOT) + " (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 x 64 "
                                  "lanes, FMA counted as 2 flops)"},
                    "nominal": {"flops_per_track_step": FLOPS_NOMINAL, "achieved": FLOPS_NOMINAL * per_s / 1e12,
                                "frac": FLOPS_NOMINAL * per_s / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                "note": "SURVEY.md §8d estimate of the reference algorithm's flops, NOT what these "
                                        "kernels execute: an estimate of useful work delivered, not a utilisation"},
                    "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s"},
            },
        }
        if world == 1:
            # PCIe-inclusive rate of the host-buffer boundary (batch.run_batch: upload inputs, filter + smooth, download
            # the four history tensors).  Reported for DESIGN.md; it is never `value`.
            del dbs[1:]
            torch.cuda.synchronize(dev)
            batch.run_batch(hb, device=dev, smooth=True)  # first call pins its staging buffers
            t1 = time.perf_counter()
            res = batch.run_batch(hb, device=dev, smooth=True)
            t_host = time.perf_counter() - t1
            out["pcie_inclusive"] = {"value": hb.track_steps / t_host, "unit": "track-steps/s", "seconds": t_host,
                                     "bytes_down": int(sum(res[k].nbytes for k in ("means", "covs", "means_smoothed",
                                                                                     "covs_smoothed")))}
            del res
        if args.cpu_tracks > 0 and world == 1:  # the CPU baseline is a rank-0, N = 1 leg
            v, secs, ref, chb = cpu_baseline(args.cpu_tracks)
            # the CPU sample is the first cpu_tracks tracks of rank 0's shard: cross-check the GPU result on it -- filtered
            # and smoothed, means and covariances (north_star: means 1e-6 relative, covariances 1e-5)
            n = min(args.cpu_tracks, B)
            idx = torch.arange(n, device=dev)
            got = db.download(("means", "covs", "means_smoothed", "covs_smoothed"), idx)
            parity = compare_histories(got, {"means": ref[0][:n], "covs": ref[1][:n], "means_smoothed": ref[2][:n],
                                             "covs_smoothed": ref[3][:n]})
            del got
            out["cpu_baseline"] = {
                "value": v, "unit": "track-steps/s", "cores": 1, "kind": "port",
                "sample": f"{args.cpu_tracks} tracks x {chb.Nmax} steps of the same synthetic workload "
                          f"({secs:.1f} s, oracle/ukf_oracle.py vectorised NumPy, single process)",
                "host_cpus": os.cpu_count(), "host_cpu_model": cpu_model(),
                "reference_itself": {
                    "value": 1482.0, "unit": "track-steps/s", "cores": 1,
                    "note": "NOC-OI/ship-track-estimators as shipped (per-ship Python loop), measured in the survey container "
                            "on one 2.1 GHz Xeon core (BASELINE.md section 2); the reference cannot travel to this box, so "
                            "the figure timed here is this repo's NumPy restatement of it (`kind: port`, ~94x faster per core)"},
                "gpu_vs_oracle_max_rel_err_smoothed_means": parity["means_smoothed"]["max_rel_err"],
                "gpu_vs_oracle": {"tracks": n, "tolerance": {"means_rel": 1e-6, "covs_rel_per_matrix": 1e-5}, **parity},
            }
            if args.cpu_ref_tracks > 0:
                nr = min(args.cpu_ref_tracks, n)
                vr, sr, per_track = cpu_reference_call_sequence(nr)
                got = db.download(("means", "covs", "means_smoothed", "covs_smoothed"), torch.arange(nr, device=dev))
                pr = compare_histories(got, {k: np.stack([t[i] for t in per_track]) for i, k in
                                             enumerate(("means", "covs", "means_smoothed", "covs_smoothed"))})
                del got
                out["cpu_baseline"]["reference_call_sequence"] = {
                    "value": vr, "unit": "track-steps/s", "cores": 1, "kind": "port",
                    "sample": f"{nr} tracks x {chb.Nmax} steps of the same batch, one after the other ({sr:.1f} s): "
                              "oracle/ukf_oracle.py forward_track + backward_track, the restatement that issues the reference's "
                              "own calls in its order (scipy.linalg.sqrtm per fan, a Python loop over the sigma points, "
                              "np.linalg.pinv per gain; kalman_filter.py:61-117, unscented.py:285-351) and reproduces the "
                              "reference bit for bit on every golden case -- the cost model of the reference's per-ship loop "
                              "(examples/example_ukf_rts_smoother_batch.py:19-90) on this box's cores",
                    "gpu_vs_this": {"tracks": nr, "means_max_rel_err": pr["means"]["max_rel_err"],
                                    "means_smoothed_max_rel_err": pr["means_smoothed"]["max_rel_err"],
                                    "covs_max_rel_err_per_matrix": pr["covs"]["max_rel_err_per_matrix"],
                                    "covs_smoothed_max_rel_err_per_matrix": pr["covs_smoothed"]["max_rel_err_per_matrix"]}}
            if pool_result is not None:
                out["cpu_baseline"]["all_cores"] = {
                    "value": pool_result[1], "unit": "track-steps/s", "cores": pool_result[0],
                    "sample": f"{pool_result[0]} forked workers x {args.cpu_pool_tracks} tracks x {chb.Nmax} steps "
                              f"({pool_result[2]:.1f} s, slowest worker)"}
        if world == 1 and args.cpu_tracks > 0:
            # The other BASELINE configs, reported beside the headline so that the driver's default run measures them too.
            if pipe is not None:
                pipe.close()  # its streams own hardware queues, and the fleet entry below builds a pipeline of its own
            dbs, db, pipe = [], None, None
            torch.cuda.empty_cache()
            out["extra"] = {}
            if not args.no_gp:
                # configs[4], the second kernel set (bench_gp.py alone prints the same object, with --fit for a whole fit)
                import bench_gp

                try:
                    out["extra"]["gp_config4"] = bench_gp.measure(1000, 2000, evals=3, cpu_evals=1)
                except Exception as exc:  # the headline line must not depend on the second workload
                    out["extra"]["gp_config4"] = {"error": f"{type(exc).__name__}: {exc}"}
            try:
                out["extra"]["config3_modern_ships_robust"] = measure_config3(dev)
            except Exception as exc:
                out["extra"]["config3_modern_ships_robust"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and not args.no_fleet and args.cpu_tracks > 0:
            dbs, db, pipe = [], None, None
            torch.cuda.empty_cache()
            try:
                out.setdefault("extra", {})["fleet_100k"] = measure_fleet(dev, args.fleet_tracks, chunk=args.fleet_chunk)
            except Exception as exc:
                out.setdefault("extra", {})["fleet_100k"] = {"error": f"{type(exc).__name__}: {exc}"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()


if __name__ == "__main__":
    main()
