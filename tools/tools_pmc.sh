#!/bin/bash
# usage: tools_pmc.sh <tag> <kernel-substring> <counters...>   (runs on the GPU box; one PMC pass; sums over dispatches)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift; kern=$1; shift
rocprofv3 --pmc $@ --output-format csv -d gpurun_out/pmc_$tag -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_$tag.log 2>&1
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob("gpurun_out/pmc_$tag/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$kern" in r["Kernel_Name"]:
            agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
for k in sorted(agg): print("%-28s %18.0f  dispatches %d" % (k, agg[k], n[k]))
PY
