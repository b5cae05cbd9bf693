#!/bin/bash
# on the GPU box: integrator 2 (HBM wavefront) on the HBM-resident workloads, wide vs binary trace kernels
for wl in atrium_1920x1080_64spp_d8 mixed_1920x1080_128spp_d8; do
  for nw in 0 1; do
    JTX_NO_WIDE=$nw JTX_INTEGRATOR=2 timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $wl | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$wl JTX_NO_WIDE=$nw', j['value'],'Mrays/s', j['ms_per_step'],'ms')" || exit 1
  done
done
