#!/usr/bin/env python3
"""Opcode histogram of the step loop of one kernel in a hipcc -save-temps .s file.

The step loop is taken to be the strongly connected component of the control-flow graph that contains the largest basic
block (the sigma-fan propagation).  Inner loops (the Jacobi sweep loop) are counted once; blocks are listed so that
cold ones can be told apart.

usage: tools/isa_loop_hist.py <file.s> <mangled-name-substring>
"""
import collections
import re
import sys

sys.setrecursionlimit(10000)
s = open(sys.argv[1]).read()
key = sys.argv[2]
for f in re.split(r"\n\s*\.globl\s+", s):
    name = f.split("\n", 1)[0].strip()
    if key not in name:
        continue
    blocks, order = {}, []
    cur = "entry"
    blocks[cur] = []
    order.append(cur)
    for ln in f.split("\n")[1:]:
        t = ln.strip()
        m = re.match(r"^(\.LBB[0-9_]+):", t)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if t.startswith("s_endpgm"):
            blocks[cur].append(t)
            break
        blocks[cur].append(t)
    succ = {b: set() for b in order}
    for i, b in enumerate(order):
        ins = blocks[b]
        fall = True
        for t in ins:
            op = t.split()[0]
            if op.startswith("s_cbranch"):
                succ[b].add(t.split()[-1])
            elif op == "s_branch":
                succ[b].add(t.split()[-1])
                fall = False
            elif op.startswith("s_endpgm") or op.startswith("s_setpc"):
                fall = False
        if fall and i + 1 < len(order):
            succ[b].add(order[i + 1])
    # Tarjan SCC
    index, low, stack, on, sccs, idx = {}, {}, [], set(), [], [0]

    def strong(v):
        index[v] = low[v] = idx[0]
        idx[0] += 1
        stack.append(v)
        on.add(v)
        for w in succ[v]:
            if w not in succ:
                continue
            if w not in index:
                strong(w)
                low[v] = min(low[v], low[w])
            elif w in on:
                low[v] = min(low[v], index[w])
        if low[v] == index[v]:
            comp = []
            while True:
                w = stack.pop()
                on.discard(w)
                comp.append(w)
                if w == v:
                    break
            sccs.append(comp)

    for b in order:
        if b not in index:
            strong(b)
    # the SCC that contains the LAST of the two largest blocks (the loop's fan, not the prologue's copy)
    big = sorted(order, key=lambda b: len(blocks[b]))[-2:]
    big = max(big, key=order.index)
    comp = next(c for c in sccs if big in c)
    comp = sorted(comp, key=order.index)
    print(name, "loop blocks:", len(comp), "instructions (static):", sum(len(blocks[b]) for b in comp))
    hist = collections.Counter()
    for b in comp:
        for t in blocks[b]:
            hist[t.split()[0]] += 1
    for op, n in hist.most_common(60):
        print(f"  {op:28s} {n}")
    print("blocks:", " ".join(f"{b}({len(blocks[b])})" for b in comp))
