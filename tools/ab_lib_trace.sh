#!/bin/bash
# usage (GPU box): tools/ab_lib_trace.sh tagA tagB ...   -- rocprofv3 kernel stats of the headline shoot per library build
# (lagomorph_amd/_lib/ab_<tag>.so), the five kernels of the Euler step side by side
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/abtrace
for v in "$@"; do
  out=gpurun_out/abtrace/$v
  rm -rf $out
  LAGO_HIP_LIBRARY=$PWD/lagomorph_amd/_lib/ab_$v.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-micro --no-atlas --no-extras > $out.json 2> $out.err < /dev/null
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "== $v  $(python3 -c "import json,sys; d=json.loads(open('$out.json').read().strip().splitlines()[-1]); print('ms_per_step', round(d['ms_per_step'],3))" 2>/dev/null)"
  if [ -n "$f" ]; then python3 tools/rocprof_summary.py "$f" | sed -n 3,7p | cut -c1-110; fi
done
