#!/bin/bash
# ON THE GPU BOX: one workload, several variant libraries.  usage: WL=<workload> tools/sweep1.sh v0 v1 ...
WL=${WL:-cornell_1920x1080_64spp_d8}
for v in "$@"; do
  r=$(JTX_MI_LIB=$PWD/jtx-pathtracer_amd/libjtx_mi_$v.so timeout -k 10 120 python3 tools/run_frames.py --workload $WL --frames ${FRAMES:-4} --warmup 1 2>&1 | grep "ms/frame") || { echo "$v FAILED"; continue; }
  echo "$v $r"
done
