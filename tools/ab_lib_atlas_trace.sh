#!/bin/bash
# usage (GPU box): tools/ab_lib_atlas_trace.sh <batch> <size> tagA tagB ... -- rocprofv3 kernel stats of the atlas step per library build
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
B=$1; S=$2; shift 2
mkdir -p gpurun_out/abatlas
for v in "$@"; do
  out=gpurun_out/abatlas/$v
  rm -rf $out
  LAGO_HIP_LIBRARY=$PWD/lagomorph_amd/_lib/ab_$v.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/run_atlas_step.py $B $S > $out.txt 2> $out.err < /dev/null
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "== $v  $(tail -1 $out.txt)"
  if [ -n "$f" ]; then python3 tools/rocprof_summary.py "$f" | sed -n 3,11p | cut -c1-110; fi
done
