#!/bin/bash
# ON THE GPU BOX: everything profiles/r05_* is made of -- counters, kernel statistics and bench lines of the three timed workloads.
# The kernel statistics come twice: from `JTX_FRAMES_IN_FLIGHT=1 bench.py` (one frame in flight: the launches the roofline's kernel_ms is
# measured on -- the averages must agree) and from the default command (three frames in flight: launches overlap, a launch's own
# duration then exceeds the time per frame).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for wl in cornell_1920x1080_64spp_d8 atrium_1920x1080_64spp_d8 mixed_1920x1080_128spp_d8; do
  tools/pmc_collect.sh r05 $wl > gpurun_out/pmc_r05_$wl.txt 2>&1 || { echo "pmc $wl failed"; tail -5 gpurun_out/pmc_r05_$wl.txt; exit 1; }
  echo "pmc $wl done"
done
declare -A STEPS=([cornell_1920x1080_64spp_d8]=10 [atrium_1920x1080_64spp_d8]=5 [mixed_1920x1080_128spp_d8]=5)
for wl in cornell_1920x1080_64spp_d8 atrium_1920x1080_64spp_d8 mixed_1920x1080_128spp_d8; do
  for mode in serial inflight; do
    rm -rf gpurun_out/ks_r05_${mode}_$wl
    if [ $mode = serial ]; then export JTX_FRAMES_IN_FLIGHT=1; else unset JTX_FRAMES_IN_FLIGHT; fi
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_r05_${mode}_$wl -- python3 bench.py --steps ${STEPS[$wl]} --warmup 2 --no-cpu-baseline --headline-only --workload $wl > gpurun_out/ks_r05_${mode}_$wl.log 2>&1 || { echo "kstats $mode $wl failed"; tail -5 gpurun_out/ks_r05_${mode}_$wl.log; exit 1; }
    cp $(ls gpurun_out/ks_r05_${mode}_$wl/*/*kernel_stats.csv | head -1) gpurun_out/ks_r05_${mode}_$wl.csv
    rm -rf gpurun_out/ks_r05_${mode}_$wl
    echo "kstats $mode $wl done"
  done
done
unset JTX_FRAMES_IN_FLIGHT
