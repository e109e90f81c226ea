import sys, time, os
sys.path[:0] = ['/root/repo', '/root/repo/ship-track-estimators_amd']
import numpy as np, torch
from track_estimators import batch, synthetic
H, Q, R, P0 = synthetic.example_matrices()
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sb = synthetic.make_batch(B, nobs=126, gap_h=1.0, seed0=7)
hb = batch.pack_uniform(sb, 4, H, Q, R, P0)
hb.lanes = 1
ref = batch.DeviceBatch(hb, device=dev)
ref.run(); torch.cuda.synchronize()
dbs = [batch.DeviceBatch(hb, device=dev) for _ in range(K)]
with batch.SmootherPipeline(dev, ntracks=B) as pipe:
    t0 = time.perf_counter()
    pipe.submit_sequence(dbs)
    pipe.synchronize()
    print('first sequence', time.perf_counter() - t0, flush=True)
    for d in dbs:
        for name in ('fwd_mean', 'fwd_cov', 'sm_mean', 'sm_cov', 'status', 'rts_work'):
            a, b = getattr(d, name), getattr(ref, name)
            if not torch.equal(a, b):
                bad = (a != b) & ~(torch.isnan(a) & torch.isnan(b)) if a.dtype.is_floating_point else (a != b)
                print('MISMATCH', name, int(bad.sum()), flush=True)
    print('bit-identity checked', flush=True)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pipe.submit_sequence(dbs); pipe.synchronize()
        t1 = time.perf_counter() - t0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i, d in enumerate(dbs):
            pipe.submit(d, final=(i == K - 1))
        pipe.synchronize()
        t2 = time.perf_counter() - t0
        print(f'K={K} B={B}: scheduled {t1*1e3:.3f} ms  per-batch launches {t2*1e3:.3f} ms', flush=True)
