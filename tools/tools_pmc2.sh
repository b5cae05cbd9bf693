#!/bin/bash
# usage: tools_pmc2.sh <tag> <workload> <integrator> <kernel-substring> <counters...>  -> sums per kernel over one bench step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; wl=$2; integ=$3; kern=$4; shift 4
JTX_INTEGRATOR=$integ rocprofv3 --pmc $@ --output-format csv -d gpurun_out/pmc_$tag -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload $wl > gpurun_out/pmc_$tag.log 2>&1
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob("gpurun_out/pmc_$tag/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        import re
        if re.search(r"""$kern""", r["Kernel_Name"]):
            agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
for k in sorted(agg): print("$tag %-24s %18.0f  dispatches %d" % (k, agg[k], n[k]))
PY
