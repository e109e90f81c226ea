# a resident 100 000-track fleet: its windows as ONE scheduled launch, or as two (the first filling the chip's SIMDs about once)
import sys, time, os
sys.path[:0] = ['/root/repo', '/root/repo/ship-track-estimators_amd']
import numpy as np, torch
from track_estimators import batch, synthetic
H, Q, R, P0 = synthetic.example_matrices()
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sb = synthetic.make_batch(B, nobs=126, gap_h=1.0, seed0=50_000_000)
hb = batch.pack_uniform(sb, 4, H, Q, R, P0)
hb.lanes = 1
db = batch.DeviceBatch(hb, device=dev)
db.run(); torch.cuda.synchronize()
def t5(fn):
    fn(); ts = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return ' '.join(f'{v:.2f}' for v in sorted(ts))
for chunk in (8192, 10048, 12544, 16384, 25088, 33344):
    wins = batch.fleet_windows(B, chunk)
    with batch.SmootherPipeline(dev, ntracks=wins[0][1] - wins[0][0]) as pipe:
        ws = [db.window(lo, hi) for lo, hi in wins]
        def one():
            pipe.submit_sequence(ws); pipe.synchronize()
        print(f'chunk {chunk}: {len(wins)} windows of {-(-(wins[0][1]-wins[0][0])//64)} tiles, one launch: {t5(one)} ms   ({len(pipe.fwd_streams)}+{len(pipe.bwd_streams)} streams)', flush=True)
        for k in range(1, len(ws)):
            tiles = sum(-(-(hi - lo) // 64) for lo, hi in wins[:k])
            if not 700 <= tiles <= 1300: continue
            def two():
                pipe.submit_sequence(ws[:k], final=False); pipe.submit_sequence(ws[k:]); pipe.synchronize()
            print(f'    two launches, {k} + {len(ws)-k} windows ({tiles} tiles first): {t5(two)} ms', flush=True)
