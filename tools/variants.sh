#!/bin/bash
# local: several variant libraries at once.  usage: tools/variants.sh "tag:<-D flags>" "tag2:<flags>" ...   (source: jtx_kernels.hip;
# VSRC=<file> overrides).  The other objects are compiled once (tools/build_variant.sh), the variants in parallel.
cd "$(dirname "$0")/.." || exit 1
vsrc=${VSRC:-jtx_kernels.hip}
first=1
pids=""
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  if [ $first = 1 ]; then tools/build_variant.sh "$tag" "$flags" "$vsrc" > /tmp/variant_$tag.log 2>&1 || { cat /tmp/variant_$tag.log; exit 1; }; first=0
  else tools/build_variant.sh "$tag" "$flags" "$vsrc" > /tmp/variant_$tag.log 2>&1 & pids="$pids $!"; fi
done
for p in $pids; do wait $p || { echo "a variant failed"; tail -5 /tmp/variant_*.log; exit 1; }; done
for spec in "$@"; do tail -n1 /tmp/variant_${spec%%:*}.log; done
