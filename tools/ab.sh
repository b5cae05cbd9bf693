#!/bin/bash
# ON THE GPU BOX: kernel ms per frame of several variant libraries on several workloads.
# usage: WLS="c2 c3 c5" FRAMES=4 tools/ab.sh v0 v1 ...      (c2 / c3 / c5 / c1 = the bench workloads)
declare -A NAME=([c2]=cornell_1920x1080_64spp_d8 [c3]=atrium_1920x1080_64spp_d8 [c5]=mixed_1920x1080_128spp_d8 [c1]=cornell_512x512_16spp_d4)
for wl in ${WLS:-c2}; do
  for v in "$@"; do
    lib=$PWD/jtx-pathtracer_amd/libjtx_mi_$v.so; [ "$v" = product ] && lib=$PWD/jtx-pathtracer_amd/libjtx_mi.so
    r=$(JTX_MI_LIB=$lib timeout -k 10 180 python3 tools/run_frames.py --workload ${NAME[$wl]} --frames ${FRAMES:-4} --warmup 1 2>&1 | grep "ms/frame") || { echo "$wl $v FAILED"; continue; }
    echo "$wl $v ${r#*: }"
  done
done
