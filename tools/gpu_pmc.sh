#!/bin/bash
# usage: tools/gpu_pmc.sh <tag> "<COUNTER1 COUNTER2 ...>" <python script> [args]   (one rocprofv3 --pmc pass)
tag=$1; ctrs=$2; shift 2
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$repo" || exit 1
rm -rf gpurun_out/pmc_$tag
timeout 600 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_$tag -- python3 "$@" > gpurun_out/pmc_$tag.out 2> gpurun_out/pmc_$tag.err
python3 tools/pmc_table.py gpurun_out/pmc_$tag/*/*_counter_collection.csv | head -60
