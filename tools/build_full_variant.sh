#!/bin/bash
# local: libjtx_mi_<tag>.so with extra flags on EVERY source (layout switches such as -DJTX_WIDE_TAILS=2 reach the builders, the refit
# and the kernels alike).  usage: tools/build_full_variant.sh <tag> "<flags>"
cd "$(dirname "$0")/../jtx-pathtracer_amd/csrc" || exit 1
tag=$1; flags=$2
SRCS=$(python3 -c "import re;print(' '.join(re.findall(r'\"(jtx_[a-z_]+\.(?:hip|cpp))\"', open('../build.py').read().split('SOURCES')[1].split(']')[0])))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 -Wall -Wno-unused-function $flags -o ../libjtx_mi_$tag.so $SRCS || exit 1
echo built libjtx_mi_$tag.so
