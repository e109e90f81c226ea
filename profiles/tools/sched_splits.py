import sys, time, os
sys.path[:0] = ['/root/repo', '/root/repo/ship-track-estimators_amd']
import numpy as np, torch
from track_estimators import batch, synthetic
H, Q, R, P0 = synthetic.example_matrices()
dev = torch.device('cuda:0')
B = 10000
sb = synthetic.make_batch(B, nobs=126, gap_h=1.0, seed0=7)
hb = batch.pack_uniform(sb, 4, H, Q, R, P0)
hb.lanes = 1
dbs = [batch.DeviceBatch(hb, device=dev) for _ in range(20)]
with batch.SmootherPipeline(dev, ntracks=B) as pipe:
    def run(splits):
        k = 0
        for i, n in enumerate(splits):
            pipe.submit_sequence(dbs[k:k + n], final=(i == len(splits) - 1))
            k += n
        pipe.synchronize()
    for splits in ((10, 10), (20,), (12, 8), (13, 7), (14, 6), (8, 12), (7, 13), (7, 7, 6), (6, 7, 7), (5, 5, 5, 5), (13, 4, 3), (16, 4)):
        run(splits)
        ts = []
        for rep in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); run(splits); ts.append((time.perf_counter() - t0) * 1e3)
        print(splits, ' '.join(f'{t:.2f}' for t in sorted(ts)), 'ms', flush=True)
    ts = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i, d in enumerate(dbs): pipe.submit(d, final=(i == 19))
        pipe.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print('per-batch', ' '.join(f'{t:.2f}' for t in sorted(ts)), 'ms', flush=True)
