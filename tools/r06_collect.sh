#!/bin/bash
# ON THE GPU BOX: everything profiles/r06_* is made of -- counters, kernel statistics and bench lines of the three timed workloads.
# The kernel statistics come twice: from `JTX_FRAMES_IN_FLIGHT=1 bench.py` (one frame in flight: the launches the roofline's kernel_ms is
# measured on -- the averages must agree) and from the default command (three frames in flight: launches overlap, a launch's own
# duration then exceeds the time per frame).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
# STAGE=pmc | kstats | progressive | all (a gpurun call is limited to 20 minutes: one stage per call)
STAGE=${STAGE:-all}
if [ $STAGE = pmc ] || [ $STAGE = all ]; then
for wl in cornell_1920x1080_64spp_d8 atrium_1920x1080_64spp_d8 mixed_1920x1080_128spp_d8; do
  tools/pmc_collect.sh r06 $wl > gpurun_out/pmc_r06_$wl.txt 2>&1 || { echo "pmc $wl failed"; tail -5 gpurun_out/pmc_r06_$wl.txt; exit 1; }
  echo "pmc $wl done"
done
fi
if [ $STAGE = kstats ] || [ $STAGE = all ]; then
declare -A STEPS=([cornell_1920x1080_64spp_d8]=10 [atrium_1920x1080_64spp_d8]=5 [mixed_1920x1080_128spp_d8]=5)
for wl in cornell_1920x1080_64spp_d8 atrium_1920x1080_64spp_d8 mixed_1920x1080_128spp_d8; do
  for mode in serial inflight; do
    rm -rf gpurun_out/ks_r06_${mode}_$wl
    if [ $mode = serial ]; then export JTX_FRAMES_IN_FLIGHT=1; else unset JTX_FRAMES_IN_FLIGHT; fi
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_r06_${mode}_$wl -- python3 bench.py --steps ${STEPS[$wl]} --warmup 2 --no-cpu-baseline --headline-only --workload $wl > gpurun_out/ks_r06_${mode}_$wl.log 2>&1 || { echo "kstats $mode $wl failed"; tail -5 gpurun_out/ks_r06_${mode}_$wl.log; exit 1; }
    cp $(ls gpurun_out/ks_r06_${mode}_$wl/*/*kernel_stats.csv | head -1) gpurun_out/ks_r06_${mode}_$wl.csv
    rm -rf gpurun_out/ks_r06_${mode}_$wl
    echo "kstats $mode $wl done"
  done
done
unset JTX_FRAMES_IN_FLIGHT
fi
if [ $STAGE = progressive ] || [ $STAGE = all ]; then
# round 6: the progressive launch (jtx_mi_render with a callback per pass): k_render_paths<.., PROG> beside k_resolve_progressive
export JTX_PROBE_CHECK=0
for spp in 1 8; do
  rm -rf gpurun_out/ks_r06_progressive_$spp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_r06_progressive_$spp -- python3 tools/progressive_probe.py cornell $spp > gpurun_out/ks_r06_progressive_$spp.log 2>&1 || { echo "kstats progressive $spp failed"; tail -5 gpurun_out/ks_r06_progressive_$spp.log; exit 1; }
  cp $(ls gpurun_out/ks_r06_progressive_$spp/*/*kernel_stats.csv | head -1) gpurun_out/ks_r06_progressive_$spp.csv
  rm -rf gpurun_out/ks_r06_progressive_$spp
  echo "kstats progressive $spp done"
done
fi
