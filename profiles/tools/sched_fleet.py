import sys, time, os
sys.path[:0] = ['/root/repo', '/root/repo/ship-track-estimators_amd']
import numpy as np, torch
from track_estimators import batch, synthetic
H, Q, R, P0 = synthetic.example_matrices()
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sb = synthetic.make_batch(B, nobs=126, gap_h=1.0, seed0=50_000_000)
hb = batch.pack_uniform(sb, 4, H, Q, R, P0)
hb.lanes = 1
db = batch.DeviceBatch(hb, device=dev)
db.run(); torch.cuda.synchronize()
ref = [t.clone() for t in (db.fwd_mean, db.fwd_cov, db.sm_mean, db.sm_cov, db.status)]
for chunk in (16384, 10048, 25088, 33344, 50048):
    wins = batch.fleet_windows(B, chunk)
    with batch.SmootherPipeline(dev, ntracks=wins[0][1] - wins[0][0]) as pipe:
        ws = [db.window(lo, hi) for lo, hi in wins]
        for stag in (0.0, 1.0):
            for t in (db.fwd_mean, db.fwd_cov, db.sm_mean, db.sm_cov): t.zero_()
            pipe.submit_sequence(ws, stagger=stag); pipe.synchronize()
            same = all(torch.equal(a, b) for a, b in zip(ref, (db.fwd_mean, db.fwd_cov, db.sm_mean, db.sm_cov, db.status)))
            ts = []
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                pipe.submit_sequence(ws, stagger=stag); pipe.synchronize()
                ts.append(time.perf_counter() - t0)
            print(f'chunk {chunk}: {len(wins)} windows, stagger {stag}: scheduled {min(ts)*1e3:.3f} ms (bit-identical {same}), {len(pipe.fwd_streams)}+{len(pipe.bwd_streams)} streams', flush=True)
        ts = []
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i, w in enumerate(ws):
                pipe.submit(w, final=(i == len(ws) - 1))
            pipe.synchronize()
            ts.append(time.perf_counter() - t0)
        print(f'chunk {chunk}: per-window launches {min(ts)*1e3:.3f} ms', flush=True)
