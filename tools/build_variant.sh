#!/bin/bash
# local: build jtx-pathtracer_amd/libjtx_mi_<tag>.so with extra -D flags applied to ONE source (default jtx_kernels.hip);
# the other objects are compiled once into /tmp/jtxobj.  usage: tools/build_variant.sh <tag> "<flags>" [source.hip]
cd "$(dirname "$0")/../jtx-pathtracer_amd/csrc" || exit 1
tag=$1; flags=$2; vsrc=${3:-jtx_kernels.hip}
OBJ=/tmp/jtxobj; mkdir -p $OBJ
CF="--offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 -Wall -Wno-unused-function"
SRCS=$(python3 -c "import re;print(' '.join(re.findall(r'\"(jtx_[a-z_]+\.(?:hip|cpp))\"', open('../build.py').read().split('SOURCES')[1].split(']')[0])))")
for f in $SRCS; do
  o=$OBJ/${f%.*}.o
  if [ "$f" != "$vsrc" ] && { [ ! -f $o ] || [ $f -nt $o ] || [ -n "$(find . ../../include -name '*.h*' -newer $o | head -1)" ]; }; then
    /opt/rocm/bin/hipcc $CF -x hip -c $f -o $o || exit 1
  fi
done
/opt/rocm/bin/hipcc $CF $flags -x hip -c $vsrc -o $OBJ/variant_$tag.o || exit 1
objs=""
for f in $SRCS; do
  if [ "$f" = "$vsrc" ]; then objs="$objs $OBJ/variant_$tag.o"; else objs="$objs $OBJ/${f%.*}.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libjtx_mi_$tag.so $objs || exit 1
echo built libjtx_mi_$tag.so
