#!/bin/bash
# on the GPU box: every workload x integrator, short runs
for wl in cornell_1920x1080_64spp_d8 atrium_1920x1080_64spp_d8 mixed_1920x1080_128spp_d8; do
  for integ in 1 2; do
    echo "== $wl integrator $integ"
    JTX_INTEGRATOR=$integ timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $wl | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'],'Mrays/s', j['ms_per_step'],'ms', 'rays/frame', j['config']['rays_per_frame'], 'rps', j['config']['rays_per_sample'], 'lds', j['config']['lds_resident_bvh'])" || exit 1
  done
done
