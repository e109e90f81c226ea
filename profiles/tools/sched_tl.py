import sys, time, os
sys.path[:0] = ['/root/repo', '/root/repo/ship-track-estimators_amd']
import numpy as np, torch
from track_estimators import batch, synthetic
H, Q, R, P0 = synthetic.example_matrices()
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
stag = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
sb = synthetic.make_batch(B, nobs=126, gap_h=1.0, seed0=7)
hb = batch.pack_uniform(sb, 4, H, Q, R, P0)
hb.lanes = 1
dbs = [batch.DeviceBatch(hb, device=dev) for _ in range(K)]
with batch.SmootherPipeline(dev, ntracks=B) as pipe:
    pipe.submit_sequence(dbs, stagger=stag); pipe.synchronize()
    # forward only
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pipe.submit_sequence(dbs, smooth=False, stagger=stag); pipe.synchronize()
        t1 = time.perf_counter() - t0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i, d in enumerate(dbs):
            pipe.submit(d, smooth=False)
        pipe.synchronize()
        t2 = time.perf_counter() - t0
        print(f'forward only K={K}: scheduled {t1*1e3:.3f} ms  per-batch {t2*1e3:.3f} ms', flush=True)
    for rep in range(2):
        tm = {"every": 1}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pipe.submit_sequence(dbs, timing=tm, stagger=stag); pipe.synchronize()
        t1 = time.perf_counter() - t0
        f0, f1 = tm['forward']
        print(f'scheduled total {t1*1e3:.3f} ms; forward kernel {f0.elapsed_time(f1):.3f} ms', flush=True)
        print('smoother start/end rel. forward start:', ' '.join(f'{f0.elapsed_time(a):.2f}-{f0.elapsed_time(b):.2f}' for a, b in tm['smoothers']), flush=True)
