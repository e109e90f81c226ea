import sys, time, os
sys.path[:0] = ['/root/repo', '/root/repo/ship-track-estimators_amd']
import numpy as np, torch
from track_estimators import batch, synthetic
H, Q, R, P0 = synthetic.example_matrices()
dev = torch.device('cuda:0')
B, K = 10000, 100
sb = synthetic.make_batch(B, nobs=126, gap_h=1.0, seed0=7)
hb = batch.pack_uniform(sb, 4, H, Q, R, P0)
hb.lanes = 1
dbs = [batch.DeviceBatch(hb, device=dev) for _ in range(40)]
with batch.SmootherPipeline(dev, ntracks=B) as pipe:
    for G, nset in ((20, 40), (10, 20), (10, 30), (14, 28), (7, 14)):
        def run(K):
            for c0 in range(0, K, G):
                ks = range(c0, min(c0 + G, K))
                pipe.submit_sequence([dbs[k % nset] for k in ks], final=(ks[-1] == K - 1))
            pipe.synchronize()
        run(2 * G)
        for K in (20, 100):
            ts = []
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter(); run(K); ts.append(time.perf_counter() - t0)
            print(f'sequences of {G} on {nset} sets, K={K}: {min(ts)/K*1e3:.4f} ms/step', flush=True)
    for K in (20, 100):
        ts = []
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(K):
                pipe.submit(dbs[k % 14], final=(k == K - 1))
            pipe.synchronize(); ts.append(time.perf_counter() - t0)
        print(f'per-batch launches on 14 sets, K={K}: {min(ts)/K*1e3:.4f} ms/step', flush=True)
