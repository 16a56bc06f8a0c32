#!/usr/bin/env python3
"""tools/merge_pmc.py PMC_TRAFFIC_JSON... -- put the pmc_traffic.json files tools/profile.sh wrote (one per workload) into
profiles/pmc_traffic.json as sections keyed "<rings>x<cols>x<batch>[+zeros]": what bench.py's roofline.traffic reads."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
out = {"note": "one section per workload of bench.py; each as tools/profile.sh wrote it (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
       "sections": {}}
for path in sys.argv[1:]:
    t = json.load(open(path))
    key = "%dx%dx%d%s" % (t["rings"], t["cols"], t["batch"], "+zeros" if t.get("drop_zero") else "")
    t["from"] = os.path.relpath(os.path.abspath(path), ROOT)
    out["sections"][key] = t
json.dump(out, open(out_path, "w"), indent=1)
print("wrote", out_path, list(out["sections"]))
