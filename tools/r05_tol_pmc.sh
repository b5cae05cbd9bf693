#!/bin/bash
# ON THE GPU BOX: SQ_INSTS_VALU of the tolerance variants -- one rocprofv3 --pmc pass per library and workload
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
declare -A NAME=([c2]=cornell_1920x1080_64spp_d8 [c3]=atrium_1920x1080_64spp_d8 [c5]=mixed_1920x1080_128spp_d8)
for wl in c2 c3 c5; do
  for v in product tolA tolB tolAB; do
    lib=$PWD/jtx-pathtracer_amd/libjtx_mi_$v.so; [ "$v" = product ] && lib=$PWD/jtx-pathtracer_amd/libjtx_mi.so
    JTX_MI_LIB=$lib PMC_TIMEOUT=200 tools/pmc_sets.sh tol_${v}_$wl ${NAME[$wl]} k_render_paths "SQ_INSTS_VALU SQ_INSTS_VMEM_RD" > gpurun_out/pmcs_tol_${v}_$wl.txt 2>&1 || echo "$v $wl failed"
    echo "$wl $v $(grep SQ_INSTS_VALU gpurun_out/pmcs_tol_${v}_$wl.txt | awk '{print $2}') VALU wave-instructions per frame"
  done
done
