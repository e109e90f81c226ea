"""Ceilings of the shared pipeline's two kernels: forward-only, smoother-only, and both without dependencies."""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, "ship-track-estimators_amd")
from track_estimators import batch, synthetic
dev = torch.device("cuda:0")
H, Q, R, P0 = synthetic.example_matrices()
B = 10000
sb = synthetic.make_batch(B, nobs=126, gap_h=1.0, seed0=0)
hb = batch.pack_uniform(sb, 4, H, Q, R, P0)
hb.lanes = 1
tuning = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
nb = 14
dbs = [batch.DeviceBatch(hb, device=dev, tuning=tuning) for _ in range(nb)]
for d in dbs:
    d.run()
torch.cuda.synchronize()
def timed(fn, K):
    fn(8); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(K); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
out = {}
for f, s in ((7, 5), (8, 5), (8, 6), (10, 8)):
    with batch.SmootherPipeline(dev, ntracks=B, forward_streams=f, smoother_streams=s) as pipe:
        fs, bs = pipe.fwd_streams, pipe.bwd_streams
        def fwd(K):
            for k in range(K): dbs[k % nb].forward(fs[k % f])
        def bwd(K):
            for k in range(K): dbs[k % nb].backward(bs[k % s])
        def both(K):
            for k in range(K):
                dbs[k % nb].forward(fs[k % f]); dbs[(k + 7) % nb].backward(bs[k % s])
        out[f"f{f}_s{s}"] = dict(fwd_only=timed(fwd, 140), bwd_only=timed(bwd, 140), both_nodep=timed(both, 140))
        print(f, s, out[f"f{f}_s{s}"], flush=True)
print(json.dumps(out))
