#!/bin/bash
# ON THE GPU BOX: like pmc_collect.sh, with the counter lists given on the command line (each list its own rocprofv3 pass).
# usage: tools/pmc_sets.sh <tag> <workload> <kernel-regex> "<list 1>" ["<list 2>" ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; wl=$2; kern=$3; shift 3
i=0
for P in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcs_${tag}_p$i
  timeout -k 10 ${PMC_TIMEOUT:-400} rocprofv3 --pmc $P --output-format csv -d gpurun_out/pmcs_${tag}_p$i -- python3 tools/run_frames.py --workload $wl --frames 1 > gpurun_out/pmcs_${tag}_p$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/pmcs_${tag}_p$i.log; }
done
python3 - <<PY
import csv, glob, json, re, collections
agg = collections.defaultdict(float); n = collections.defaultdict(int); names = set()
for f in glob.glob("gpurun_out/pmcs_${tag}_p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if re.search(r"""$kern""", r["Kernel_Name"]):
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1; names.add(r["Kernel_Name"])
json.dump({"tag": "$tag", "workload": "$wl", "kernel": sorted(names), "counters_per_dispatch": {k: agg[k] / n[k] for k in sorted(agg)}},
          open("gpurun_out/pmcs_${tag}.json", "w"), indent=1)
print("$tag", sorted(names))
for k in sorted(agg): print("%-40s %20.0f" % (k, agg[k] / n[k]))
PY
rm -rf gpurun_out/pmcs_${tag}_p?/
