#!/bin/bash
# usage (build container): tools/build_ab_one.sh <file.hip> "tag:-DFLAG=1 ..." ...  -> lagomorph_amd/_lib/ab_<tag>.so each:
# ONE source recompiled with the flags and linked against the default build's other objects (seconds instead of the
# full rebuild of build_ab_variants.sh).  The default library is not touched.
cd "$(dirname "$0")/.."
src=$1; shift
python -m lagomorph_amd.build > /dev/null || exit 1
B=lagomorph_amd/_build
base=$(basename "$src" .hip)
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-pass-failed $flags -c lagomorph_amd/csrc/$base.hip -o /tmp/ab_${tag}_$base.o || { echo "build $tag failed"; continue; }
  objs=$(ls $B/*.o | grep -v "/$base.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lagomorph_amd/_lib/ab_$tag.so $objs /tmp/ab_${tag}_$base.o -L/opt/rocm/lib -lhipfft && echo "built ab_$tag.so ($flags)"
done
