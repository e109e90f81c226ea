# one long-lived process: every ~8 s time (7,13), (10,10), (20) and per-step launches of 20 batches, with the host time spent
# inside each submit_sequence call -- does the slow mode switch on inside a process, and is the host blocked in it?
import sys, time, ctypes as C, subprocess
sys.path[:0] = ['/root/repo', '/root/repo/ship-track-estimators_amd']
import numpy as np, torch
from track_estimators import batch, synthetic
from track_estimators._hip import binding
H, Q, R, P0 = synthetic.example_matrices()
dev = torch.device('cuda:0')
B = 10000
sb = synthetic.make_batch(B, nobs=126, gap_h=1.0, seed0=7)
hb = batch.pack_uniform(sb, 4, H, Q, R, P0)
hb.lanes = 0
dbs = [batch.DeviceBatch(hb, device=dev) for _ in range(20)]
total = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
with batch.SmootherPipeline(dev, ntracks=B) as pipe:
    def seq(splits, host=None):
        k = 0
        for i, n in enumerate(splits):
            h0 = time.perf_counter()
            pipe.submit_sequence(dbs[k:k + n], final=(i == len(splits) - 1)); k += n
            if host is not None: host.append((time.perf_counter() - h0) * 1e3)
        h0 = time.perf_counter(); pipe.synchronize()
        if host is not None: host.append((time.perf_counter() - h0) * 1e3)
    def per_step(host=None):
        for k in range(20):
            pipe.submit(dbs[k], final=(k == 19))
        pipe.synchronize()
    def t(fn):
        torch.cuda.synchronize(); host = []; t0 = time.perf_counter(); fn(host); return (time.perf_counter() - t0) * 1e3, host
    seq((7, 13)); seq((10, 10)); seq((20,)); per_step()
    T0 = time.time()
    while time.time() - T0 < total:
        a, ha = t(lambda h: seq((7, 13), h)); b, hb_ = t(lambda h: seq((10, 10), h)); c, hc = t(lambda h: seq((20,), h)); d, _ = t(per_step)
        print(f't={time.time()-T0:6.1f}s (7,13) {a:6.2f} host {" ".join(f"{v:.2f}" for v in ha)} | (10,10) {b:6.2f} host {" ".join(f"{v:.2f}" for v in hb_)} | (20) {c:6.2f} host {" ".join(f"{v:.2f}" for v in hc)} | per-step {d:6.2f}', flush=True)
        time.sleep(6)
