#!/usr/bin/env python3
"""Merge a respond_traffic.json produced by scripts/profile_gpu.sh (one record) into profiles/respond_traffic.json ({"records": [...]}),
replacing the record of the same config + packing.   usage: merge_traffic.py gpurun_out/prof_<tag>/respond_traffic.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles", "respond_traffic.json")
new = json.load(open(sys.argv[1]))
doc = json.load(open(dst)) if os.path.exists(dst) else {"records": []}
doc["records"] = [r for r in doc.get("records", []) if (r.get("config"), r.get("packing")) != (new.get("config"), new.get("packing"))] + [new]
json.dump(doc, open(dst, "w"), indent=1)
print("records:", [(r["config"], r["packing"], r.get("git_head")) for r in doc["records"]])
