#!/bin/bash
# ON THE GPU BOX: time the three workloads with each variant library  (usage: tools/sweep.sh v0 v1 ...)
for v in "$@"; do
  for wl in cornell_1920x1080_64spp_d8 atrium_1920x1080_64spp_d8 mixed_1920x1080_128spp_d8; do
    r=$(JTX_MI_LIB=$PWD/jtx-pathtracer_amd/libjtx_mi_$v.so timeout -k 10 120 python3 tools/run_frames.py --workload $wl --frames ${FRAMES:-3} --warmup 1 2>&1 | grep "ms/frame") || { echo "$v $wl FAILED"; continue; }
    echo "$v $r"
  done
done
