#!/bin/bash
# usage: tools/gpu_pmc_multi.sh <tag> <script+args quoted> "<set1>" "<set2>" ...   (one rocprofv3 --pmc pass per set)
tag=$1; cmd=$2; shift 2
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$repo" || exit 1
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_${tag}_$i
  timeout 600 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_${tag}_$i -- python3 $cmd > gpurun_out/pmc_${tag}_$i.out 2> gpurun_out/pmc_${tag}_$i.err || tail -3 gpurun_out/pmc_${tag}_$i.err
  python3 tools/pmc_table.py gpurun_out/pmc_${tag}_$i/*/*_counter_collection.csv | grep -A1 "ad_star\|compose3\|jtv_fwd\|zy_forward\|xpass2"
  rm -rf gpurun_out/pmc_${tag}_$i
done
