#!/usr/bin/env python3
"""One line per bench.py JSON file: value, ms/step, steady state, kernel times, serial figure, parity."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(path, "ERR", e)
        continue
    st = d.get("steady_state") or {}
    se = d.get("serial") or {}
    cb = (d.get("cpu_baseline") or {}).get("gpu_vs_oracle") or {}
    par = ""
    if cb:
        par = " | err m %.1e ms %.1e c %.1e cs %.1e" % (cb["means"]["max_rel_err"], cb["means_smoothed"]["max_rel_err"],
                                                         cb["covs"]["max_rel_err_per_matrix"], cb["covs_smoothed"]["max_rel_err_per_matrix"])
    print("%-40s %.3e  %.3f ms/step  steady %s  fwd %.2f bwd %.2f  serial %s (%s)  flagged %s%s" % (
        path.split("/")[-1], d["value"], d["ms_per_step"], ("%.3f" % st["ms_per_step"]) if st else "-",
        d["kernels_ms"]["ukf_forward"], d["kernels_ms"]["urtss_backward"],
        ("%.2f" % se["ms_per_step"]) if se else "-",
        "/".join("%.2f" % v for v in (se.get("kernels_ms") or {}).values()), d.get("status_flagged_tracks"), par))
