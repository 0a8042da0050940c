#!/bin/bash
# usage (GPU box): tools/gpu_traffic_atlas.sh <batch> <size> -- HBM traffic per launch of the kernels of one atlas step (two --pmc passes)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/atlas_traffic_$1_$2
rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -- python3 tools/run_atlas_step.py $1 $2 > /dev/null 2> $out/f.err < /dev/null
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w -- python3 tools/run_atlas_step.py $1 $2 > /dev/null 2> $out/w.err < /dev/null
python3 tools/pmc_traffic.py $out/f/*/*_counter_collection.csv $out/w/*/*_counter_collection.csv $out/traffic.json > $out/traffic.md
head -16 $out/traffic.md
