import sys, time, os
sys.path[:0] = ['/root/repo', '/root/repo/ship-track-estimators_amd']
import numpy as np, torch
from track_estimators import batch, synthetic
H, Q, R, P0 = synthetic.example_matrices()
dev = torch.device('cuda:0')
B = 100000
sb = synthetic.make_batch(B, nobs=126, gap_h=1.0, seed0=50_000_000)
hb = batch.pack_uniform(sb, 4, H, Q, R, P0)
hb.lanes = 1
db = batch.DeviceBatch(hb, device=dev)
db.run(); torch.cuda.synchronize()
for chunk in (16384, 10048):
    wins = batch.fleet_windows(B, chunk)
    with batch.SmootherPipeline(dev, ntracks=wins[0][1] - wins[0][0]) as pipe:
        ws = [db.window(lo, hi) for lo, hi in wins]
        pipe.submit_sequence(ws); pipe.synchronize()
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pipe.submit_sequence(ws, smooth=False); pipe.synchronize()
            t1 = time.perf_counter() - t0
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for w in ws: pipe.submit(w, smooth=False)
            pipe.synchronize()
            t2 = time.perf_counter() - t0
            torch.cuda.synchronize(); t0 = time.perf_counter()
            db.forward(); torch.cuda.synchronize()
            t3 = time.perf_counter() - t0
            print(f'chunk {chunk} forward only: scheduled {t1*1e3:.3f}  per-window {t2*1e3:.3f}  one launch {t3*1e3:.3f} ms', flush=True)
        for rep in range(2):
            tm = {"every": 1}
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pipe.submit_sequence(ws, timing=tm); pipe.synchronize()
            t1 = time.perf_counter() - t0
            f0, f1 = tm['forward']
            print(f'scheduled total {t1*1e3:.3f} ms; forward kernel {f0.elapsed_time(f1):.3f} ms', flush=True)
            print('smoothers:', ' '.join(f'{f0.elapsed_time(a):.2f}-{f0.elapsed_time(b):.2f}' for a, b in tm['smoothers']), flush=True)
        it = pipe._schedules[list(pipe._schedules)[0]]
        print('rounds', it.shape, 'idle', int((it[..., 0] < 0).sum()))
