"""ms per step in windows of a long pipelined run: does the rate change as the streams drift out of lockstep?"""
import sys, time
import numpy as np, torch
sys.path.insert(0, "ship-track-estimators_amd")
from track_estimators import batch, synthetic
dev = torch.device("cuda:0")
H, Q, R, P0 = synthetic.example_matrices()
sb = synthetic.make_batch(10000, nobs=126, gap_h=1.0, seed0=0)
hb = batch.pack_uniform(sb, 4, H, Q, R, P0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
W = 100
with batch.SmootherPipeline(dev, ntracks=hb.B) as pipe:
    dbs = [batch.DeviceBatch(hb, device=dev) for _ in range(pipe.buffers_needed)]
    for k in range(len(dbs)):
        pipe.submit(dbs[k])
    pipe.synchronize(); torch.cuda.synchronize()
    marks = []
    for k in range(K):
        done = pipe.submit(dbs[k % len(dbs)])
        if k % W == W - 1:
            e = torch.cuda.Event(enable_timing=True)
            e.record(pipe.bwd_streams[k % len(pipe.bwd_streams)])
            marks.append(e)
    pipe.synchronize(); torch.cuda.synchronize()
    ms = [marks[i].elapsed_time(marks[i + 1]) / W for i in range(len(marks) - 1)]
    print("windows of %d steps, ms per step:" % W)
    print(" ".join("%.3f" % v for v in ms))
