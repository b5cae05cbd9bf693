#!/usr/bin/env python3
"""local: merge gpurun_out/pmc_<tag>_<workload>.json (tools/pmc_collect.sh) and gpurun_out/wide_stats_<scene>.json
(tools/tools_wide_stats.py, diagnostic build) into profiles/<tag>_pmc.json, stamped with the hash of the kernel sources the
counters were collected from (bench.py marks the file stale when the sources have changed since)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
out = {"source_hash": bench.source_hash(), "collected_with": "tools/pmc_collect.sh (rocprofv3 --pmc, seven separate passes, one uncounted frame) "
       "and tools/tools_wide_stats.py (-DJTX_PROFILE_WIDE build)", "workloads": {}}
ws = {"atrium_1920x1080_64spp_d8": "atrium", "mixed_1920x1080_128spp_d8": "mixed"}
for wl in bench.WORKLOADS:
    p = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{wl}.json")
    if not os.path.exists(p):
        continue
    j = json.load(open(p))
    e = {"kernel": j["kernel"], "counters": j["counters_per_dispatch"]}
    w = os.path.join(ROOT, "gpurun_out", f"wide_stats_{ws.get(wl, '')}.json")
    if wl in ws and os.path.exists(w):
        e["wide_stats"] = json.load(open(w))
    out["workloads"][wl] = e
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
print(f"profiles/{tag}_pmc.json:", list(out["workloads"]))
