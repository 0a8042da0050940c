#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: kernel-trace stats, the two PMC passes and a plain bench run.
# usage: tools/gpu_profile.sh <tag>
tag=${1:-X}
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$repo" || exit 1
mkdir -p gpurun_out
rm -rf gpurun_out/prof_$tag gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof$tag.json 2> gpurun_out/bench_prof$tag.err
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-micro > /dev/null 2> gpurun_out/pmc_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-micro > /dev/null 2> gpurun_out/pmc_write.err
python3 tools/pmc_traffic.py gpurun_out/pmc_fetch/*/*_counter_collection.csv gpurun_out/pmc_write/*/*_counter_collection.csv gpurun_out/traffic_$tag.json > gpurun_out/traffic_$tag.md
python3 tools/rocprof_summary.py gpurun_out/prof_$tag/*/*_kernel_stats.csv > gpurun_out/kernel_stats_$tag.md
if [ -s gpurun_out/traffic_$tag.json ]; then cp gpurun_out/traffic_$tag.json profiles/r01_traffic.json; fi   # bench.py reads it for roofline.traffic
timeout 600 python3 bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
tail -c 400 gpurun_out/bench_$tag.json
