# the pipeline's own queue footprint against the device's hardware-queue slots: scheduled launches on a pipeline with
# 1 / 2 / 7 forward streams (x 6 smoother streams + 1), with k extra idle queues in the process
import sys, time, ctypes as C
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
lib = binding.load()
extra = []
def add_queues(n):
    for _ in range(n):
        h = C.c_void_p()
        binding.check(lib.ste_stream_create_cu_range(0, 256, C.byref(h)), "create")
        s = torch.cuda.ExternalStream(h.value, device=dev)
        with torch.cuda.stream(s):
            torch.zeros(16, device=dev).add_(1)
        extra.append((h, s))
    torch.cuda.synchronize()
def drop_queues():
    global extra
    torch.cuda.synchronize()
    for h, s in extra:
        del s
        lib.ste_stream_destroy(h)
    extra = []
for fs, ss in ((7, 6), (2, 6), (1, 6), (2, 3)):
    for nq in (0, 8, 12, 16, 20):
        drop_queues(); add_queues(nq)
        with batch.SmootherPipeline(dev, ntracks=B, forward_streams=fs, smoother_streams=ss) as pipe:
            def seq(splits):
                k = 0
                for i, n in enumerate(splits):
                    pipe.submit_sequence(dbs[k:k + n], final=(i == len(splits) - 1)); k += n
                pipe.synchronize()
            def t(fn, reps=5):
                fn(); out = []
                for _ in range(reps):
                    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); out.append((time.perf_counter() - t0) * 1e3)
                return ' '.join(f'{v:.1f}' for v in sorted(out))
            print(f'pipeline {fs}+{ss}+1 streams, extra queues {nq}: (7,13) {t(lambda: seq((7,13)))} | (20) {t(lambda: seq((20,)))}', flush=True)
