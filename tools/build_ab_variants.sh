#!/bin/bash
# usage (build container): tools/build_ab_variants.sh "tag:-DFLAG=1 -DOTHER=2" ...  -> lagomorph_amd/_lib/ab_<tag>.so each
# (full rebuilds with LAGO_HIPCC_EXTRA); restores the default library at the end.
cd "$(dirname "$0")/.."
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  LAGO_HIPCC_EXTRA="$flags" python -m lagomorph_amd.build -f > /dev/null 2>/tmp/ab_build_$tag.err || { echo "build $tag failed"; tail -5 /tmp/ab_build_$tag.err; }
  cp lagomorph_amd/_lib/liblagomorph_hip.so lagomorph_amd/_lib/ab_$tag.so
  echo "built ab_$tag.so ($flags)"
done
python -m lagomorph_amd.build -f > /dev/null
