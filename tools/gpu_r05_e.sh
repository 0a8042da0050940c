#!/bin/bash
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$repo" || exit 1
mkdir -p gpurun_out
timeout 600 python tools/ab_compose_epilogue.py 32 128 > gpurun_out/r05_compose_epilogue.txt 2>&1
timeout 300 python tools/ab_compose_epilogue.py 8 128 >> gpurun_out/r05_compose_epilogue.txt 2>&1
timeout 300 python tools/ab_compose_epilogue.py 16 96 >> gpurun_out/r05_compose_epilogue.txt 2>&1
cat gpurun_out/r05_compose_epilogue.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_ce_prof -- python3 tools/ab_compose_epilogue.py 32 128 > /dev/null 2>&1
python3 tools/rocprof_summary.py gpurun_out/r05_ce_prof/*/*_kernel_stats.csv 2>/dev/null | head -20 > gpurun_out/r05_compose_epilogue_kernels.txt
cat gpurun_out/r05_compose_epilogue_kernels.txt
rm -rf gpurun_out/r05_ce_prof
