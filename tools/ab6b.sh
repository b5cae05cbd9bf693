#!/bin/bash
# ON THE GPU BOX: variants built inside _base_r5 (round-5 tree + experiments): kernel ms per frame.  usage: WLS="c2 c3" tools/ab6b.sh product minpush ...
declare -A NAME=([c2]=cornell_1920x1080_64spp_d8 [c3]=atrium_1920x1080_64spp_d8 [c5]=mixed_1920x1080_128spp_d8 [c1]=cornell_512x512_16spp_d4)
ROOT=$PWD
cd $ROOT/_base_r5
for wl in ${WLS:-c2}; do
  for v in "$@"; do
    lib=$PWD/jtx-pathtracer_amd/libjtx_mi_$v.so; [ "$v" = product ] && lib=$PWD/jtx-pathtracer_amd/libjtx_mi.so
    r=$(JTX_MI_LIB=$lib timeout -k 10 240 python3 tools/run_frames.py --workload ${NAME[$wl]} --frames ${FRAMES:-4} --warmup 1 2>&1 | grep "ms/frame") || { echo "$wl $v FAILED"; continue; }
    echo "$wl $v ${r#*: }"
  done
done
