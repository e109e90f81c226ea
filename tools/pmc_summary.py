#!/usr/bin/env python3
"""Average every counter of every ste kernel over its launches, from the counter_collection.csv files of one or more
`rocprofv3 --pmc ... --kernel-trace --output-format csv` passes, and derive the per-track-step / per-wave-step figures
DESIGN.md and bench.py quote.

usage: tools/pmc_summary.py <dir-with-pass-subdirs> [track_steps_per_launch=5e6] [steps=500]
Prints a table; with --csv writes the two profiles/ files' rows to stdout instead.
"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
track_steps = float(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 5.0e6
steps = float(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith("-") else 500.0


def short(name):
    name = name.replace("void ", "").replace("(ste::KParams)", "")
    return name if name.startswith("ste::") else None


acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if not k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (path, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)

for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    print(f"== {k}  ({len(dur[k])} launches, {sum(dur[k]) / len(dur[k]):.3f} ms average while counting)")
    for n in sorted(c):
        print(f"   {n:28s} {c[n]:16.1f}")
    w = c.get("SQ_WAVES")
    if w:
        for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"):
            if n in c:
                print(f"   {n + ' / wave / step':36s} {c[n] / w / steps:10.1f}")
    f64 = [c.get("SQ_INSTS_VALU_" + x + "_F64") for x in ("ADD", "MUL", "FMA", "TRANS")]
    if all(v is not None for v in f64):
        flops = (f64[0] + f64[1] + f64[3] + 2 * f64[2]) * 64 / track_steps
        print(f"   fp64 wave-instructions per launch {sum(f64):.0f}; flop / track-step {flops:.1f}")
    if "FETCH_SIZE" in c:
        print(f"   HBM read  B / track-step (x2 gfx950) {c['FETCH_SIZE'] * 1024 * 2 / track_steps:10.1f}")
    if "WRITE_SIZE" in c:
        print(f"   HBM write B / track-step             {c['WRITE_SIZE'] * 1024 / track_steps:10.1f}")
    if "SQ_WAVE_CYCLES" in c:
        wc = c["SQ_WAVE_CYCLES"]
        for n in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            if n in c:
                print(f"   {n + ' / SQ_WAVE_CYCLES':36s} {c[n] / wc:10.3f}")
