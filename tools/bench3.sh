#!/bin/bash
# ON THE GPU BOX: the three timed workloads, short runs -> one line each
for wl in cornell_1920x1080_64spp_d8 atrium_1920x1080_64spp_d8 mixed_1920x1080_128spp_d8; do
  timeout -k 10 300 python3 tools/run_frames.py --workload $wl --frames ${FRAMES:-5} --warmup 2 || exit 1
done
