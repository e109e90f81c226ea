#!/usr/bin/env python3
"""Timeline of the ste kernels in a `rocprofv3 --kernel-trace --output-format csv` file.

usage: tools/trace_timeline.py <kernel_trace.csv> [last_n]
Prints start / end (ms, relative to the first of the last_n ste dispatches), the queue and the duration of each dispatch,
and per queue the idle gaps between consecutive dispatches -- what VERDICT r02 item 1 asked for: where the time of a
`bench.py --steps 20 --warmup 5` run goes.
"""
import csv
import sys


def short(name):
    for k in ("ukf_forward_sched", "urtss_recur_sched", "sched_gate", "sched_upload", "ukf_forward_l1", "ukf_forward_q4", "urtss_smooth_wg", "urtss_recur_l1", "urtss_backward_l1"):
        if k in name:
            return k
    return None


def main():
    path = sys.argv[1]
    last_n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            s = short(r["Kernel_Name"])
            if s:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), s))
    rows.sort()
    if last_n:
        rows = rows[-last_n:]
    t0 = rows[0][0]
    print(f"{'kernel':18s} {'queue':>5s} {'start':>9s} {'end':>9s} {'ms':>7s}")
    last_end = {}
    for s, e, q, k in rows:
        gap = "" if q not in last_end else f"  gap {1e-6 * (s - last_end[q]):7.3f}"
        print(f"{k:18s} {q:5d} {1e-6 * (s - t0):9.3f} {1e-6 * (e - t0):9.3f} {1e-6 * (e - s):7.3f}{gap}")
        last_end[q] = e
    print(f"span {1e-6 * (max(r[1] for r in rows) - t0):.3f} ms over {len(rows)} dispatches")


if __name__ == "__main__":
    main()
