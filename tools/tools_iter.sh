#!/bin/bash
# local: rebuild the HIP library, then run GPU parity tests + a short bench on the MI355X box
# env: INTEG=<1|2|3> integrator for the bench, NOTEST=1 skips pytest, STEPS
cd "$(dirname "$0")/.."
python -c "import jtx_pathtracer_amd as j; j.build_all(force=True)" || exit 1
T="timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 &&"
[ -n "$NOTEST" ] && T=""
/usr/local/graft/bin/gpurun --timeout 900 -- "$T JTX_INTEGRATOR=${INTEG:-0} timeout -k 10 300 python bench.py --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline $@ | cut -c1-330" 2>&1 | grep -v "^\[gpurun\] sending\|amdgpu.ids\|stderr (tail)"
