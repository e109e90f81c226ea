#!/usr/bin/env python3
"""How many Jacobi sweeps a fan's square root costs on the bench batch, and how many of their rotations any lane needed:
the instrumented build of csrc/ste_lane.h (-DSTE_DEBUG_SWEEPS: device counters g_dbg, read through ste_dbg_counters, which is
not part of the ABI).  The figures quoted in ste_lane.h / DESIGN.md section 5 (2.2 sweeps per solve, 99.6 % of the solves need two,
15 % of the executed rotations are identities) come from this script.

  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=fast-honor-pragmas -mllvm -disable-machine-licm \
        -DSTE_DEBUG_SWEEPS ship-track-estimators_amd/csrc/*.hip -o /tmp/libste_dbg.so
  STE_LIB_PATH=/tmp/libste_dbg.so python3 profiles/tools/jacobi_sweep_probe.py
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ship-track-estimators_amd")]
assert os.environ.get("STE_LIB_PATH"), "point STE_LIB_PATH at the -DSTE_DEBUG_SWEEPS build (see the header)"
import numpy as np, torch
from track_estimators import batch, synthetic
from track_estimators._hip import binding
lib = binding.require_gpu()
dbg = C.CDLL(binding.LIB_PATH).ste_dbg_counters
H, Q, R, P0 = synthetic.example_matrices()
sb = synthetic.make_batch(10000, nobs=126, gap_h=1.0, seed0=0)
hb = batch.pack_uniform(sb, 4, H, Q, R, P0)
hb.lanes = 1
db = batch.DeviceBatch(hb)
out = (C.c_ulonglong * 64)()
dbg(None, 1)
db.forward(); torch.cuda.synchronize()
dbg(out, 1)
o = list(out)
solves = o[0]
print("eigen-solves (wave level):", solves, "per wave-step", solves / 157 / 500)
print("sweeps per solve:", o[1] / solves, " rotations some lane needed per solve:", o[2] / solves, " of", 6 * o[1] / solves, "executed")
print("lane-level rotations needed per lane-solve:", o[3] / (solves * 64))
print("sweeps-per-solve histogram:", {i: o[8 + i] for i in range(16) if o[8 + i]})
print("-log10(offdiag) histogram over probes:", {i: o[32 + i] for i in range(21) if o[32 + i]})
