#!/bin/bash
# Kernel-trace stats of the headline shoot with the LDS-window gathers off and on, same box (run via gpurun).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for w in 0 1 0 1; do
  out=gpurun_out/prof_w$w
  rm -rf $out && mkdir -p $out
  LAGO_GATHER_WINDOW=$w timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/expmap_trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-micro --no-atlas > $out/bench_trace.json 2> $out/bench_trace.err
  echo "window $w"; python3 tools/rocprof_summary.py $out/expmap_trace/*/*_kernel_stats.csv | sed -n 3,7p | cut -c1-100
  rm -rf $out/expmap_trace
done
