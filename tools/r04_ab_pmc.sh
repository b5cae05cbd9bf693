#!/bin/bash
# ON THE GPU BOX: A/B counters of variant libraries on one workload ("each with A/B counters committed as profiles/r04_*").
# (FETCH_SIZE, WRITE_SIZE and GRBM_GUI_ACTIVE in ONE pass crash rocprofv3 on this pool: split as in pmc_collect.sh)
# usage: tools/r04_ab_pmc.sh <workload> <kernel-regex> v0 v1 ...     -> gpurun_out/pmcs_ab_<v>_<workload>.json
wl=$1; kern=$2; shift 2
for v in "$@"; do
  lib=$PWD/jtx-pathtracer_amd/libjtx_mi_$v.so; [ "$v" = product ] && lib=$PWD/jtx-pathtracer_amd/libjtx_mi.so
  JTX_MI_LIB=$lib PMC_TIMEOUT=300 tools/pmc_sets.sh ab_${v}_$wl $wl "$kern" \
    "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" \
    "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum" \
    "TD_TD_BUSY_sum SQ_THREAD_CYCLES_VALU" \
    "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" > gpurun_out/pmcs_ab_${v}_$wl.txt 2>&1 || echo "$v failed"
  echo "== $v"; tail -n +2 gpurun_out/pmcs_ab_${v}_$wl.txt
done
