#!/usr/bin/env python3
"""
bench.py — UKF + URTSS track-steps/s on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (forward UKF + unscented RTS smoother) over one batch of synthetic tracks that
is already resident in HBM.  Workload at N=1: BASELINE.json configs[1] — 10 000 synthetic dim-4 geodetic tracks x 500
steps, fp64 (126 observations at 1 h gaps, 4 sub-steps).  With N>1 (launched by torch.distributed.run,
one rank per GPU over RCCL) every rank filters its own 10 000-track shard (weak scaling; tracks are independent) and
the step ends with the one exchange the path has: an all-gather of the smoothed lon/lat (BASELINE.json configs[2]).

Prints ONE JSON line on rank 0.  Extra objects:
  roofline      dominant kernel, algorithmic HBM bytes / HIP-event duration vs 8 TB/s (512 B per
                track-step for the pair of kernels = 192 B forward + 320 B backward)
  cpu_baseline  the NumPy oracle ("port" of the reference arithmetic) timed on this box's host cores on a bounded sample
                (one process, and one forked worker per core of the box's CPU share)
  serial        the same work without overlapping consecutive steps (see --no-pipeline), for the record

Steps are pipelined by default: every step runs the complete forward pass and smoother of one 10 000-track batch, but up
to four steps are in flight -- two forward passes (625 long-running waves each, two waves per SIMD) on 160 CUs and the two
smoothers before them (latency-bound) on the other 96, on CU-masked streams, with five sets of history buffers in rotation
(track_estimators.batch.SmootherPipeline).
"""
import argparse
import contextlib
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

TRACKS_PER_GPU = 10_000
NOBS = 126
SUBSTEPS = 4
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6
BYTES_FWD = 192  # algorithmic, per track-step: 32 B inputs + 160 B filtered mean/cov written
BYTES_BWD = 320  # algorithmic, per track-step: 160 B filtered history re-read + 160 B smoothed written
# Measured HBM bytes per track-step (rocprofv3 PMC, FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc_hbm_traffic_v6_quad.csv).
# Above the algorithmic figure by the rts_work rows the forward pass hands to the smoother.
TRAFFIC_FWD = 441
TRAFFIC_BWD = 337 + 560  # urtss_gain_kernel + urtss_combine_l1
FLOPS_PER_TRACK_STEP = 2.0e4  # SURVEY.md §8d estimate (fp64 flop-equivalents, forward + backward)


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


def main():
    # the pipeline's CU-masked streams are destroyed before the interpreter goes down, whatever happens in between
    with contextlib.ExitStack() as stack:
        _main(stack)


def _main(stack):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--tracks", type=int, default=TRACKS_PER_GPU, help="tracks per GPU")
    ap.add_argument("--cpu-tracks", type=int, default=3072, help="tracks in the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per track (0 = library default)")
    ap.add_argument("--no-gather", action="store_true", help="skip the all-gather of smoothed lon/lat when N>1")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run forward and smoother of every step back to back on one stream instead of overlapping the "
                         "smoother of step i with the forward pass of step i+1 on disjoint CU partitions")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group and run the gather even with one rank (rehearsal of the N>1 path)")
    ap.add_argument("--cpu-pool-tracks", type=int, default=768,
                    help="tracks per worker in the all-cores CPU-baseline leg (0 = skip)")
    args = ap.parse_args()

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
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # STE_BENCH_BACKEND=gloo lets several ranks share one GPU for a rehearsal of the N>1 control flow (RCCL refuses
        # two ranks on one device); the driver's multi-GPU runs use the default, nccl = RCCL over xGMI.
        backend = os.environ.get("STE_BENCH_BACKEND", "nccl")
        ndev = torch.cuda.device_count()
        local_rank = local_rank % max(ndev, 1)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    lib = binding.require_gpu()
    if args.lanes:
        lib.ste_set_lanes_per_track(args.lanes)

    # --- synthetic shard of this rank, resident in HBM before the timed region ---------------------------------
    H, Q, R, P0 = synthetic.example_matrices()
    B = args.tracks
    sb = synthetic.make_batch(B, nobs=NOBS, gap_h=1.0, seed0=rank * B)
    hb = batch.pack_uniform(sb, SUBSTEPS, H, Q, R, P0)
    db = batch.DeviceBatch(hb, device=dev)
    # Pipelined mode (default): five sets of histories in rotation; two forward passes at a time share the first 160 CUs
    # (two waves per SIMD) while the smoothers of the two steps before them share the other 96 (batch.SmootherPipeline).
    # Every step still does the whole forward + smoother of one batch; nothing is skipped, the steps overlap.
    pipe = None
    if not args.no_pipeline:
        try:
            # lane-per-track recurrence on the smoother partition: 1.95 instead of 2.39 ms there, which keeps the smoother
            # off the critical path now that the forward pass takes 2.38 ms (results differ from the quad recurrence
            # by rounding, ~1e-13; the oracle cross-check below covers them)
            pipe = stack.enter_context(batch.SmootherPipeline(dev, ntracks=B, smoother_lane_per_track=True))
        except (binding.SteError, ValueError) as exc:  # no CU-masked streams here, or a batch too large to partition
            print(f"[bench] pipelining disabled, steps run back to back: {exc}", file=sys.stderr, flush=True)
    dbs = [db] if pipe is None else [db] + [batch.DeviceBatch(hb, device=dev) for _ in range(pipe.buffers_needed - 1)]
    gathered = None
    if dist is not None and not args.no_gather:
        # the one exchange of the path: all-gather of the smoothed lon/lat, overlapped with the next step's kernels
        gathered = distributed.OverlappedGather(hb.Nmax + 1, B, dev)

    stream = torch.cuda.current_stream(dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731

    def serial_step(d, events=None):
        if events is not None:
            events[0].record(stream)
        d.forward(stream)
        if events is not None:
            events[1].record(stream)
            events[2].record(stream)
        d.backward(stream)
        if events is not None:
            events[3].record(stream)
        if gathered is not None:
            gathered.launch(d.sm_mean)

    def one_step(k, events=None, final=False):
        d = dbs[k % len(dbs)]
        if pipe is None:
            serial_step(d, events)
        else:
            pipe.submit(d, after_smoother=(lambda _s: gathered.launch(d.sm_mean)) if gathered is not None else None,
                        timing=events, final=final)

    def drain():
        if pipe is not None:
            pipe.synchronize()
        if gathered is not None:
            gathered.finish()
        torch.cuda.synchronize(dev)

    # back-to-back figure for the record (not the timed region): forward + smoother of one batch on one stream
    serial_ms = None
    if pipe is not None and rank == 0:
        saved, gathered = gathered, None
        serial_step(db)
        torch.cuda.synchronize(dev)
        ts = time.perf_counter()
        for _ in range(3):
            serial_step(db)
        torch.cuda.synchronize(dev)
        serial_ms = (time.perf_counter() - ts) / 3 * 1e3
        gathered = saved

    for k in range(args.warmup):
        one_step(k, final=(k == args.warmup - 1))
    drain()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    evs = [[ev(), ev(), ev(), ev()] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(k, evs[k], final=(k == args.steps - 1))  # the last smoother has nothing to hide behind: whole chip
    drain()  # every step's smoother (and gathered result) has landed before the clock stops
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    fwd_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))
    bwd_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in evs]))
    status = db.status_host()
    for d in dbs[1:]:
        status = status | d.status_host()
    track_steps_rank = hb.track_steps
    total_units = track_steps_rank * world * args.steps
    value = total_units / elapsed

    if rank == 0:
        dom, dom_ms, dom_bytes, dom_traffic = ("urtss_backward", bwd_ms, BYTES_BWD, TRAFFIC_BWD) if bwd_ms >= fwd_ms else (
            "ukf_forward", fwd_ms, BYTES_FWD, TRAFFIC_FWD)
        achieved = dom_bytes * track_steps_rank / (dom_ms * 1e-3) / 1e9
        # whole-job rates of this rank over the timed region (the two kernels overlap in pipelined mode, so their event
        # durations no longer add up to the step)
        per_s = track_steps_rank * args.steps / elapsed
        pair_gbs = (BYTES_FWD + BYTES_BWD) * per_s / 1e9
        valu_tf = FLOPS_PER_TRACK_STEP * per_s / 1e12
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
                "workload": f"{B} synthetic dim-4 geodetic tracks x {hb.Nmax} steps per GPU, UKF+URTSS, fp64 "
                            "(BASELINE.json configs[1]); zero injected noise; inputs resident in HBM",
                "tracks_per_gpu": B, "steps_per_track": int(hb.Nmax), "observations": NOBS, "substeps": SUBSTEPS,
                "parallelism": f"track-sharded x{world}" + (", RCCL all-gather of smoothed lon/lat overlapped with the next step" if gathered is not None else ""),
                "pipeline": ("none: forward and smoother of a step back to back on one stream" if pipe is None else
                             f"{len(pipe.fwd_streams)} forward passes in flight on {pipe.forward_cus} CUs (two waves per "
                             f"SIMD) beside {len(pipe.bwd_streams)} smoothers on the other {pipe.smoother_cus} CUs (CU-masked "
                             f"streams, {len(dbs)} sets of histories in rotation, lane-per-track recurrence on the smoother "
                             "partition)"),
                "lanes_per_track": int(lib.ste_set_lanes_per_track(args.lanes)),
            },
            "kernels_ms": {"ukf_forward": fwd_ms, "urtss_backward": bwd_ms},
            "serial": None if serial_ms is None else {
                "ms_per_step": serial_ms, "value": track_steps_rank / (serial_ms * 1e-3), "unit": "track-steps/s",
                "note": "one batch, forward then smoother on one unrestricted stream (per GPU, no overlap between steps)"},
            "status_flagged_tracks": int((status != 0).sum()),
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": dom_traffic * track_steps_rank if db.rts_work is not None else None,
                "traffic_unit": "bytes per launch",
                "traffic_source": "profiles/r01_pmc_hbm_traffic_v6_quad.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate "
                                  "passes of this command; FETCH_SIZE doubled per the gfx950 calibration in profiles/README.md)",
                "algorithmic_bytes_per_track_step": dom_bytes,
                "pair_achieved": pair_gbs,
                "note": "path is fp64-VALU/latency bound, not HBM bound (SURVEY.md headline 6)",
                "fp64_valu": {"achieved": valu_tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": valu_tf / FP64_VALU_PEAK_TFLOPS, "flops_per_track_step": FLOPS_PER_TRACK_STEP},
            },
        }
        if world == 1:
            # PCIe-inclusive rate of the host-buffer boundary (batch.run_batch: upload inputs, filter + smooth, download
            # the four history tensors).  Reported for DESIGN.md; it is never `value`.
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            res = batch.run_batch(hb, device=dev, smooth=True)
            t_host = time.perf_counter() - t1
            out["pcie_inclusive"] = {"value": hb.track_steps / t_host, "unit": "track-steps/s", "seconds": t_host,
                                     "bytes_down": int(sum(res[k].nbytes for k in ("means", "covs", "means_smoothed",
                                                                                     "covs_smoothed")))}
            del res
        if args.cpu_tracks > 0:
            v, secs, ref, chb = cpu_baseline(args.cpu_tracks)
            # the CPU sample is the first cpu_tracks tracks of rank 0's shard: cross-check the GPU result on it
            n = min(args.cpu_tracks, B)
            gm = db.sm_mean[:, :, :n].permute(2, 0, 1).cpu().numpy()
            err = float(np.max(np.abs(gm - ref[2][:n]) / np.maximum(np.abs(ref[2][:n]), 1e-12)))
            out["cpu_baseline"] = {
                "value": v, "unit": "track-steps/s", "cores": 1, "kind": "port",
                "sample": f"{args.cpu_tracks} tracks x {chb.Nmax} steps of the same synthetic workload "
                          f"({secs:.1f} s, oracle/ukf_oracle.py vectorised NumPy, single process)",
                "host_cpus": os.cpu_count(),
                "gpu_vs_oracle_max_rel_err_smoothed_means": err,
            }
            if pool_result is not None:
                out["cpu_baseline"]["all_cores"] = {
                    "value": pool_result[1], "unit": "track-steps/s", "cores": pool_result[0],
                    "sample": f"{pool_result[0]} forked workers x {args.cpu_pool_tracks} tracks x {chb.Nmax} steps "
                              f"({pool_result[2]:.1f} s, slowest worker)"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
