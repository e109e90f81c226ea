"""Per-step timeline of a K-step pipelined run after a drained warm-up (the driver's --steps 20 --warmup 5 shape)."""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, "ship-track-estimators_amd")
from track_estimators import batch, synthetic
dev = torch.device("cuda:0")
H, Q, R, P0 = synthetic.example_matrices()
B = 10000
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kw = {}
quads = set(int(v) for v in sys.argv[2].split(",") if v) if len(sys.argv) > 2 else set()
nbuf = int(sys.argv[3]) if len(sys.argv) > 3 else None
verbose = len(sys.argv) > 4
sb = synthetic.make_batch(B, nobs=126, gap_h=1.0, seed0=0)
hb = batch.pack_uniform(sb, 4, H, Q, R, P0)
with batch.SmootherPipeline(dev, ntracks=B, **kw) as pipe:
    dbs = [batch.DeviceBatch(hb, device=dev) for _ in range(nbuf or pipe.buffers_needed)]
    for k in range(len(dbs)):
        pipe.submit(dbs[k]); 
    pipe.synchronize(); torch.cuda.synchronize()
    for rep in range(2):
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(K)]
        t0 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        w0 = time.perf_counter()
        t0.record(pipe.fwd_streams[0])
        for k in range(K):
            pipe.submit(dbs[k % len(dbs)], timing=evs[k], final=(k == K - 1), lanes=(4 if k in quads else None))
        w1 = time.perf_counter()
        pipe.synchronize(); torch.cuda.synchronize()
        w2 = time.perf_counter()
    print(f"submit {1e3*(w1-w0):.2f} ms, total {1e3*(w2-w0):.2f} ms, per step {1e3*(w2-w0)/K:.3f}")
    print("quads", sorted(quads), "buffers", len(dbs))
    for k in range(K if verbose else 0):
        a = [t0.elapsed_time(e) for e in evs[k]]
        print(f"step {k:2d} fwd {a[0]:6.2f} -> {a[1]:6.2f} ({a[1]-a[0]:.2f})  bwd {a[2]:6.2f} -> {a[3]:6.2f} ({a[3]-a[2]:.2f})")
