#!/usr/bin/env python3
"""tools/show_bench.py FILE... -- the headline and the side configurations of bench.py's JSON line(s), one row each."""
import json
import sys

for path in sys.argv[1:]:
    p = json.loads(open(path).read().strip().splitlines()[-1])
    r = p["roofline"]
    print("%s: %.0f scans/s (min %.0f max %.0f) %.4f ms/step frac %.4f whole %.4f box %s" % (
        path, p["value"], p.get("value_min", 0), p.get("value_max", 0), p["ms_per_step"], r["frac"], r["whole_path_frac"],
        (p.get("box") or {}).get("copy_gbs")))
    print("   ", {k: v for k, v in r["kernel_us_per_launch"].items() if v > 0}, r.get("ceiling"))
    for c in p.get("configs") or []:
        if "error" in c:
            print("   ERROR", c)
            continue
        print("  %-58s %10.0f %8.4f ms frac %.3f whole %.3f %s %s" % (
            (c["workload"][:24] + ".." + c["workload"][-30:]) if len(c["workload"]) > 58 else c["workload"], c["value"], c["ms_per_step"], c["frac"], c["whole_path_frac"],
            {k.replace("_kernel", ""): v for k, v in c["kernel_us_per_launch"].items()}, c["parity_spot_check"]))
