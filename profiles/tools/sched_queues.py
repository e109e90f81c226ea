# does the number of live HSA queues in the process (or beside it) change how scheduled launches overlap?
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
with batch.SmootherPipeline(dev, ntracks=B) as pipe:
    def seq(splits):
        k = 0
        for i, n in enumerate(splits):
            pipe.submit_sequence(dbs[k:k + n], final=(i == len(splits) - 1)); k += n
        pipe.synchronize()
    def per_step():
        for k in range(20):
            pipe.submit(dbs[k], final=(k == 19))
        pipe.synchronize()
    def t(fn, reps=4):
        fn(); out = []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); out.append((time.perf_counter() - t0) * 1e3)
        return ' '.join(f'{v:.2f}' for v in out)
    for nq in (8, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2):
        add_queues(nq)
        print(f'extra queues {len(extra)}: per-step {t(per_step)} | (7,13) {t(lambda: seq((7,13)))} | (10,10) {t(lambda: seq((10,10)))} | (20) {t(lambda: seq((20,)))}', flush=True)
