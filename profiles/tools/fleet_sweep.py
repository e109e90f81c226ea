#!/usr/bin/env python3
"""How a resident fleet of 100 000 x 500 goes through the chip fastest: batch.run_fleet under different window sizes,
stream counts and time slices per forward pass, beside the same fleet as one launch (profiles/r04_fleet_sweep.txt).
usage: python3 profiles/tools/fleet_sweep.py [ntracks=100000]"""
import itertools
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "ship-track-estimators_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from track_estimators import batch, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
H, Q, R, P0 = synthetic.example_matrices()
hb = batch.pack_uniform(synthetic.make_batch(n, nobs=126, gap_h=1.0, seed0=50_000_000), 4, H, Q, R, P0)
hb.lanes = 1
db = batch.DeviceBatch(hb)
ts = hb.track_steps


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


t = timed(db.run)
print(f"one launch                                              {t * 1e3:7.2f} ms  {ts / t:.3e} track-steps/s", flush=True)
# host side of a fleet call alone: the same windows over a fleet of 2 steps
configs = [(c, f, s, sl) for c, (f, s), sl in itertools.product((10_000, 16_384, 8_192), ((7, 6), (10, 6), (13, 8), (16, 8)), (1, 2, 4))]
for chunk, fs, ss, sl in configs:
    nwin = len(batch.fleet_windows(n, chunk))
    if fs > nwin + 3:
        continue
    with batch.SmootherPipeline("cuda:0", ntracks=chunk, forward_streams=fs, smoother_streams=ss, slices=sl) as pipe:
        t = timed(lambda: batch.run_fleet(db, chunk=chunk, pipeline=pipe))
    print(f"chunk {chunk:6d} ({nwin:2d} windows) fwd {fs:2d} bwd {ss} slices {sl}   {t * 1e3:7.2f} ms  {ts / t:.3e}", flush=True)
