#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root.  usage: tools/gpu_profile.sh <tag>
#  1. headline workload only (bench.py --no-micro --no-atlas): kernel-trace stats + the two PMC passes -> traffic per launch
#     of the kernels of lddmm.expmap at batch 32 x 3 x 128^3 (what bench.py's roofline.traffic reads);
#  2. tools/run_micro.py: the same three passes for the configs[1] pair and the backward kernels at batch 8;
#  3. tools/run_atlas_step.py 8 160: kernel-trace stats of the configs[4] atlas step.
# Counters are collected in runs of their own (no trace domains beside --pmc), program directly after `--`.
tag=${1:-X}
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$repo" || exit 1
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
# --streams 1: whole-batch launches on one stream, the launch bench.py's roofline object describes (the product default
# cuts a forward shoot into two sub-batches on two streams: half-size launches that overlap)
B="python3 bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-micro --no-atlas --no-epoch --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/expmap_trace -- $B > $out/bench_trace.json 2> $out/bench_trace.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/expmap_fetch -- $B > /dev/null 2> $out/expmap_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/expmap_write -- $B > /dev/null 2> $out/expmap_write.err
M="python3 tools/run_micro.py"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/micro_trace -- $M > /dev/null 2> $out/micro_trace.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/micro_fetch -- $M > /dev/null 2> $out/micro_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/micro_write -- $M > /dev/null 2> $out/micro_write.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/atlas_trace -- python3 tools/run_atlas_step.py 8 160 > $out/atlas160.txt 2> $out/atlas_trace.err
python3 tools/rocprof_summary.py $out/expmap_trace/*/*_kernel_stats.csv > gpurun_out/${tag}_expmap_kernel_stats.md
python3 tools/rocprof_summary.py $out/micro_trace/*/*_kernel_stats.csv > gpurun_out/${tag}_micro_kernel_stats.md
python3 tools/rocprof_summary.py $out/atlas_trace/*/*_kernel_stats.csv > gpurun_out/${tag}_atlas160_kernel_stats.md
python3 tools/pmc_traffic.py $out/expmap_fetch/*/*_counter_collection.csv $out/expmap_write/*/*_counter_collection.csv gpurun_out/${tag}_traffic_expmap.json > gpurun_out/${tag}_traffic_expmap.md
python3 tools/pmc_traffic.py $out/micro_fetch/*/*_counter_collection.csv $out/micro_write/*/*_counter_collection.csv gpurun_out/${tag}_traffic_micro.json > gpurun_out/${tag}_traffic_micro.md
tail -1 $out/atlas160.txt
head -14 gpurun_out/${tag}_expmap_kernel_stats.md
cat gpurun_out/${tag}_traffic_micro.md
