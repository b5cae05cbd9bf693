#!/bin/bash
# ON THE GPU BOX: rocprofv3 counter passes (each --pmc list its own run, never combined with a trace domain) over one
# uncounted frame of a workload, summed for the dominant kernel; writes gpurun_out/pmc_<tag>_<workload>.json.
# usage: tools/pmc_collect.sh <tag> <workload> [kernel-regex]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; wl=$2; kern=${3:-k_render_paths}
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM"
P3="FETCH_SIZE GRBM_GUI_ACTIVE"
P4="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
# round 3: the dynamic instruction mix (VALU classes issue at different rates on gfx950: tools/micro/rate4.hip) and the vector L1
P5="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32"
P6="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum"
P7="TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum"      # the vector-memory pipeline (address / data-return units)
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6" "$P7"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_${tag}_${wl}_p$i
  timeout -k 10 400 rocprofv3 --pmc $P --output-format csv -d gpurun_out/pmc_${tag}_${wl}_p$i -- python3 tools/run_frames.py --workload $wl --frames 1 > gpurun_out/pmc_${tag}_${wl}_p$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/pmc_${tag}_${wl}_p$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, json, re, collections
agg = collections.defaultdict(float); n = collections.defaultdict(int); names = set()
for f in glob.glob("gpurun_out/pmc_${tag}_${wl}_p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if re.search(r"""$kern""", r["Kernel_Name"]):
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1; names.add(r["Kernel_Name"])
out = {"tag": "$tag", "workload": "$wl", "kernel": sorted(names), "dispatches": max(n.values()) if n else 0,
       "counters_per_dispatch": {k: agg[k] / n[k] for k in sorted(agg)},
       "method": "rocprofv3 --pmc, seven separate passes over tools/run_frames.py --frames 1 (one uncounted frame); sums over all XCDs/SEs"}
json.dump(out, open("gpurun_out/pmc_${tag}_${wl}.json", "w"), indent=1)
for k in sorted(agg): print("%-26s %20.0f" % (k, agg[k] / n[k]))
PY
rm -rf gpurun_out/pmc_${tag}_${wl}_p?/
