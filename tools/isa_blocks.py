#!/usr/bin/env python3
"""Basic-block map of one kernel in a hipcc -save-temps .s file: label, instruction count, VALU / SALU / VMEM / scratch
counts, and where each block branches -- enough to find the step loop and read off its per-iteration instruction mix.

usage: tools/isa_blocks.py <file.s> <mangled-name-substring>
"""
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
funcs = re.split(r"\n\s*\.globl\s+", s)
for f in funcs:
    name = f.split("\n", 1)[0].strip()
    if key not in name:
        continue
    print(name)
    blocks, cur = [], {"label": "entry", "n": 0, "valu": 0, "salu": 0, "vmem": 0, "scratch": 0, "br": [], "f64": 0, "mov": 0}
    for ln in f.split("\n")[1:]:
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith("."):
            m = re.match(r"^(\.LBB[0-9_]+):", t)
            if m:
                blocks.append(cur)
                cur = {"label": m.group(1), "n": 0, "valu": 0, "salu": 0, "vmem": 0, "scratch": 0, "br": [], "f64": 0, "mov": 0}
            continue
        m = re.match(r"^(\.LBB[0-9_]+):", t)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "n": 0, "valu": 0, "salu": 0, "vmem": 0, "scratch": 0, "br": [], "f64": 0, "mov": 0}
            continue
        op = t.split()[0]
        if op.startswith("s_endpgm"):
            cur["n"] += 1
            continue
        cur["n"] += 1
        if op.startswith("v_"):
            cur["valu"] += 1
            if "f64" in op:
                cur["f64"] += 1
            if op.startswith(("v_mov", "v_accvgpr", "v_cndmask", "v_readlane", "v_writelane", "v_readfirstlane")):
                cur["mov"] += 1
        elif op.startswith("scratch_"):
            cur["scratch"] += 1
        elif op.startswith(("global_", "flat_", "buffer_")):
            cur["vmem"] += 1
        elif op.startswith("s_"):
            cur["salu"] += 1
            if op.startswith(("s_cbranch", "s_branch")):
                cur["br"].append(t.split()[-1])
    blocks.append(cur)
    print(f"{'label':14s} {'n':>5s} {'valu':>5s} {'f64':>5s} {'mov':>5s} {'salu':>5s} {'vmem':>5s} {'scr':>4s}  branches")
    for b in blocks:
        print(f"{b['label']:14s} {b['n']:5d} {b['valu']:5d} {b['f64']:5d} {b['mov']:5d} {b['salu']:5d} {b['vmem']:5d} {b['scratch']:4d}  {' '.join(b['br'])}")
