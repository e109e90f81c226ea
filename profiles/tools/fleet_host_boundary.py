#!/usr/bin/env python3
"""The host-buffer boundary of batch.run_fleet on a 100 000 x 500 fleet: HostBatch in, NumPy histories out (uploads of window
k + 1 and downloads of window k - 1 overlapped with window k's kernels).  profiles/r04_fleet_host_boundary.txt."""
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
t0 = time.perf_counter()
hb = batch.pack_uniform(synthetic.make_batch(n, nobs=126, gap_h=1.0, seed0=50_000_000), 4, H, Q, R, P0)
print(f"host synthesis + packing {time.perf_counter() - t0:.1f} s; inputs {sum(a.nbytes for a in (hb.dt, hb.sog_rate, hb.cog_rate, hb.upd_idx, hb.z, hb.x0)) / 1e9:.2f} GB", flush=True)
ts = hb.track_steps
for outputs in (("means_smoothed",), ("means", "means_smoothed"), None):
    for rep in range(2):
        t0 = time.perf_counter()
        out = batch.run_fleet(hb, outputs=outputs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        nbytes = sum(v.nbytes for k, v in out.items() if k not in ("status", "nsteps", "device_batch"))
        print(f"outputs={outputs or 'all four'} run {rep}: {dt:.3f} s  {ts / dt:.3e} track-steps/s  {nbytes / 1e9:.1f} GB down", flush=True)
        del out
