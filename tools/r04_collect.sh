#!/bin/bash
# ON THE GPU BOX: everything profiles/r04_* is made of -- counters, kernel statistics and bench lines of the three timed workloads.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for wl in cornell_1920x1080_64spp_d8 atrium_1920x1080_64spp_d8 mixed_1920x1080_128spp_d8; do
  tools/pmc_collect.sh r04 $wl > gpurun_out/pmc_r04_$wl.txt 2>&1 || { echo "pmc $wl failed"; tail -5 gpurun_out/pmc_r04_$wl.txt; exit 1; }
  echo "pmc $wl done"
done
declare -A STEPS=([cornell_1920x1080_64spp_d8]=10 [atrium_1920x1080_64spp_d8]=5 [mixed_1920x1080_128spp_d8]=5)
for wl in cornell_1920x1080_64spp_d8 atrium_1920x1080_64spp_d8 mixed_1920x1080_128spp_d8; do
  rm -rf gpurun_out/ks_r04_$wl
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_r04_$wl -- python3 bench.py --steps ${STEPS[$wl]} --warmup 2 --no-cpu-baseline --headline-only --workload $wl > gpurun_out/ks_r04_$wl.log 2>&1 || { echo "kstats $wl failed"; tail -5 gpurun_out/ks_r04_$wl.log; exit 1; }
  cp $(ls gpurun_out/ks_r04_$wl/*/*kernel_stats.csv | head -1) gpurun_out/ks_r04_$wl.csv
  rm -rf gpurun_out/ks_r04_$wl
  echo "kstats $wl done"
done
