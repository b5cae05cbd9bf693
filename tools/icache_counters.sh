cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for wl in cornell_1920x1080_64spp_d8 atrium_1920x1080_64spp_d8 mixed_1920x1080_128spp_d8; do
  rm -rf gpurun_out/ic_$wl
  timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES --output-format csv -d gpurun_out/ic_$wl -- python3 tools/run_frames.py --workload $wl --frames 1 > gpurun_out/ic_$wl.log 2>&1 || { echo failed $wl; tail -3 gpurun_out/ic_$wl.log; continue; }
  python3 - <<PY
import csv, glob, collections, re
agg=collections.defaultdict(float)
for f in glob.glob("gpurun_out/ic_$wl/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_render_paths" in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"])
print("$wl", {k: "%.3g" % v for k, v in agg.items()}, "icache hit rate %.4f" % (agg["SQC_ICACHE_HITS"] / max(1, agg["SQC_ICACHE_REQ"])), "misses per wave-quad-cycle %.5f" % (agg["SQC_ICACHE_MISSES"] / max(1, agg["SQ_WAVE_CYCLES"])))
PY
  rm -rf gpurun_out/ic_$wl
done
