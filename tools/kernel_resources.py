#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` remarks.

usage: tools/kernel_resources.py <file.hip> [filter]
"""
import os
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # the flags the library is built with, per file

cmd = [entry._hipcc()] + entry.HIPCC_FLAGS + entry.HIPCC_FILE_FLAGS.get(os.path.basename(src), []) + [
    "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for ln in err.splitlines():
    m = re.search(r"remark: (.*?): (.*?)\s*\[-Rpass", ln) or re.search(r"remark: \s*(.*?): (.*?)\s*\[-Rpass", ln)
    if not m:
        if "error" in ln:
            print(ln)
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print(f"{'kernel':60s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'occ':>4s} {'vspill':>6s} {'sspill':>6s} {'LDS':>7s}")
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(ste::KParams\)", "", name).replace("void ", "")
    if flt and flt not in name:
        continue
    print(f"{name[:60]:60s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} "
          f"{r.get('ScratchSize [bytes/lane]', '?'):>8s} {r.get('Occupancy [waves/SIMD]', '?'):>4s} "
          f"{r.get('VGPRs Spill', '?'):>6s} {r.get('SGPRs Spill', '?'):>6s} {r.get('LDS Size [bytes/block]', '?'):>7s}")
