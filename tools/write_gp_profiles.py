#!/usr/bin/env python3
"""profiles/rNN_gp_kernel_stats_1000x2000.csv and rNN_gp_counters_1000x2000.csv from the rocprofv3 passes of
profiles/tools/pmc_passes_gp.sh (usage: tools/write_gp_profiles.py gpurun_out/<pmc dir> [round, default 04])."""
import collections
import csv
import glob
import os
import shutil
import sys

root = sys.argv[1]
RN = "r" + (sys.argv[2] if len(sys.argv) > 2 else "04")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("stegp::"):
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (path, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
if not acc:
    sys.exit(f"no counter files of the GP kernels under {root}")
names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
         "SQ_ACTIVE_INST_ANY", "TCC_HIT_sum", "TCC_MISS_sum"]
stats = glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True)
ms = {}
if stats:
    shutil.copy(stats[0], f"profiles/{RN}_gp_kernel_stats_1000x2000.csv")
    for r in csv.DictReader(open(stats[0])):
        ms[r["Name"].split("(")[0].replace("void ", "")] = float(r["AverageNs"]) * 1e-6
with open(f"profiles/{RN}_gp_counters_1000x2000.csv", "w") as f:
    f.write("# rocprofv3 --pmc <group> --kernel-trace -- python3 bench_gp.py --tracks 1000 --nobs 2000 --evals 2 --cpu-evals 0 (profiles/tools/pmc_passes_gp.sh);\n")
    f.write(f"# GP kernels of round {RN}; averages per launch; avg_ms from the --stats pass of the same script (no counters).\n")
    f.write("# hbm_read_GB = FETCH_SIZE KiB x 2048 / 1e9 (gfx950 half-count), hbm_write_GB = WRITE_SIZE KiB x 1024 / 1e9; l2_hit = TCC_HIT / (HIT + MISS)\n")
    f.write("kernel,avg_ms,hbm_read_GB,hbm_write_GB,hbm_TBps,l2_hit," + ",".join(names) + "\n")
    for k in sorted(acc, key=lambda k: -ms.get(k, 0.0)):
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        rd, wr = c.get("FETCH_SIZE", 0.0) * 2048 / 1e9, c.get("WRITE_SIZE", 0.0) * 1024 / 1e9
        t = ms.get(k, sum(dur[k]) / len(dur[k]))
        hit = c.get("TCC_HIT_sum", 0.0) / max(c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0), 1.0)
        f.write('"%s",%.3f,%.2f,%.2f,%.3f,%.3f,' % (k, t, rd, wr, (rd + wr) / t, hit) + ",".join(("%.6g" % c[n]) if n in c else "" for n in names) + "\n")
print(open(f"profiles/{RN}_gp_counters_1000x2000.csv").read())
