#!/usr/bin/env python3
"""profiles/rNN_counters_per_track_step.csv and rNN_pmc_counters_per_launch.csv from the rocprofv3 --pmc passes of
profiles/tools/pmc_passes_rNN.sh (usage: tools/write_profiles.py gpurun_out/<pmc dir> [round, default 05]).

The --stats pass and the --pmc passes must be of the same build: the script refuses to write anything when the filter kernels
named in the kernel-stats file differ from the ones in the counter files (round 4 committed a stats file taken before the
forward kernel gained a template parameter)."""
import collections
import csv
import glob
import os
import shutil
import sys

root = sys.argv[1]
RN = "r" + (sys.argv[2] if len(sys.argv) > 2 else "05")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("void ", "").replace("(ste::KParams)", "")
        if not k.startswith("ste::"):
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (path, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
names = sorted({n for k in acc for n in acc[k]})
if not acc or not all(k in acc for _, k in [("f", "ste::ukf_forward_l1<true, true, false>")]):
    sys.exit(f"no counter files of the filter kernels under {root}: nothing written")
# the --stats pass must name the same filter kernels as the counter passes (same build)
stats_files = glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats_files:
    norm = lambda n: n.replace("void ", "").replace("(ste::KParams)", "")  # noqa: E731
    in_stats = {norm(r["Name"]) for r in csv.DictReader(open(stats_files[0])) if norm(r["Name"]).startswith("ste::")}
    filt = lambda names_: {n for n in names_ if n.startswith(("ste::ukf_forward", "ste::urtss_"))}  # noqa: E731
    if filt(in_stats) != filt(acc):
        sys.exit("kernel-stats pass and counter passes are of different builds:\n  stats only:    %s\n  counters only: %s\nnothing written"
                 % (sorted(filt(in_stats) - filt(acc)), sorted(filt(acc) - filt(in_stats))))
TS = 5.0e6
with open(f"profiles/{RN}_pmc_counters_per_launch.csv", "w") as f:
    f.write("# rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 bench.py --steps 10 --warmup 2 --cpu-tracks 0 --no-gp\n")
    f.write("# (profiles/tools/pmc_passes_" + RN + ".sh: one pass per counter group, never combined with other trace domains); kernels of round " + RN + ",\n")
    f.write("# 10 000 tracks x 500 steps = 5.0e6 track-steps per launch; average of every counter over the launches of each kernel;\n")
    f.write("# ms = average dispatch duration while counting (counter runs serialise kernels: these are alone-on-the-chip times)\n")
    f.write("kernel,launches,ms," + ",".join(names) + "\n")
    for k in sorted(acc):
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        f.write('"%s",%d,%.3f,' % (k, len(dur[k]), sum(dur[k]) / len(dur[k])) + ",".join(("%.1f" % c[n]) if n in c else "" for n in names) + "\n")
rows = [("ukf_forward", "ste::ukf_forward_l1<true, true, false>"), ("urtss_backward", "ste::urtss_recur_l1<false>"),
        ("ukf_forward_q4", "ste::ukf_forward_q4<true, false, true>")]
with open(f"profiles/{RN}_counters_per_track_step.csv", "w") as f:
    f.write("# rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --steps 10 --warmup 2 --cpu-tracks 0 --no-gp   (kernels of round " + RN + ", 10 000 tracks x 500 steps\n")
    f.write("# = 5.0e6 track-steps per launch; one --pmc pass per counter group (profiles/tools/pmc_passes_" + RN + ".sh), averages over the launches of each kernel; raw\n")
    f.write("# averages: profiles/" + RN + "_pmc_counters_per_launch.csv).  hbm_read = FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count, profiles/README.md),\n")
    f.write("# hbm_write = WRITE_SIZE KiB x 1024, fp64_flops = (ADD_F64 + MUL_F64 + TRANS_F64 + 2 FMA_F64) wave-instructions x 64 lanes, all / 5.0e6.\n")
    f.write("# ukf_forward = the lane-per-track kernel the default (pipelined) run launches; ukf_forward_q4 = the quad-per-track kernel a batch on its\n")
    f.write("# own gets at this size (bench.py's `serial` leg); urtss_backward = the recurrence smoother (one lane per track, gain solve + recurrence).\n")
    f.write("kernel,device_kernel,hbm_read,hbm_write,fp64_flops,valu_insts_per_wave_step,salu_insts_per_wave_step,fp64_insts_per_wave_step,waves_per_launch,alone_ms\n")
    for short, k in rows:
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        w = c["SQ_WAVES"]
        f64 = [c["SQ_INSTS_VALU_%s_F64" % x] for x in ("ADD", "MUL", "FMA", "TRANS")]
        flops = (f64[0] + f64[1] + f64[3] + 2 * f64[2]) * 64 / TS
        f.write('%s,"%s",%.2f,%.2f,%.1f,%.1f,%.1f,%.1f,%d,%.3f\n' % (
            short, k, c["FETCH_SIZE"] * 2048 / TS, c["WRITE_SIZE"] * 1024 / TS, flops, c["SQ_INSTS_VALU"] / w / 500,
            c["SQ_INSTS_SALU"] / w / 500, sum(f64) / w / 500, w, sum(dur[k]) / len(dur[k])))
for src, dst in (("stats/**/*kernel_stats.csv", f"profiles/{RN}_pipelined_kernel_stats.csv"), ("bench_k100.json", f"profiles/{RN}_bench_default.json"),
                 ("bench_driver_form.json", f"profiles/{RN}_bench_driver_form.json"),
                 ("stats_driver_form/**/*kernel_stats.csv", f"profiles/{RN}_driver_form_kernel_stats.csv")):
    m = glob.glob(os.path.join(root, src), recursive=True)
    if m:
        shutil.copy(m[0], dst)
print(open(f"profiles/{RN}_counters_per_track_step.csv").read())
