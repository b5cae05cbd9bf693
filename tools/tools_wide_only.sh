#!/bin/bash
# on the GPU box: integrator 1 with the wide traversal on the HBM-resident workloads
for wl in atrium_1920x1080_64spp_d8 mixed_1920x1080_128spp_d8; do
    JTX_INTEGRATOR=1 timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $wl | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$wl', j['value'],'Mrays/s', j['ms_per_step'],'ms')" || exit 1
done
