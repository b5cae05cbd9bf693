#!/bin/bash
# ON THE GPU BOX: kernel ms per frame, round-5 library (in _base_r5/, its own Python) against the tree's (and variants libjtx_mi_<v>.so)
# usage: WLS="c2 c3 c5" FRAMES=6 tools/ab6.sh base product v1 ...
declare -A NAME=([c2]=cornell_1920x1080_64spp_d8 [c3]=atrium_1920x1080_64spp_d8 [c5]=mixed_1920x1080_128spp_d8 [c1]=cornell_512x512_16spp_d4)
ROOT=$PWD
for wl in ${WLS:-c2}; do
  for v in "$@"; do
    if [ "$v" = base ]; then
      r=$(cd $ROOT/_base_r5 && timeout -k 10 240 python3 tools/run_frames.py --workload ${NAME[$wl]} --frames ${FRAMES:-4} --warmup 1 2>&1 | grep "ms/frame") || { echo "$wl $v FAILED"; continue; }
    else
      lib=$ROOT/jtx-pathtracer_amd/libjtx_mi_$v.so; [ "$v" = product ] && lib=$ROOT/jtx-pathtracer_amd/libjtx_mi.so
      r=$(JTX_MI_LIB=$lib timeout -k 10 240 python3 tools/run_frames.py --workload ${NAME[$wl]} --frames ${FRAMES:-4} --warmup 1 2>&1 | grep "ms/frame") || { echo "$wl $v FAILED"; continue; }
    fi
    echo "$wl $v ${r#*: }"
  done
done
