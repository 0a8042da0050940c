#!/bin/bash
# usage: tools/gpu_stats.sh <tag> <python script> [args]  -> rocprofv3 kernel-trace stats summary of the script
tag=$1; shift
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$repo" || exit 1
rm -rf gpurun_out/stats_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_$tag -- python3 "$@" > gpurun_out/stats_$tag.out 2> gpurun_out/stats_$tag.err
cat gpurun_out/stats_$tag.out | grep -v amdgpu.ids | tail -5
python3 tools/rocprof_summary.py gpurun_out/stats_$tag/*/*_kernel_stats.csv | head -45
rm -f gpurun_out/stats_$tag/*/*_kernel_trace.csv
