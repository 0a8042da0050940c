#!/bin/bash
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$repo" || exit 1
mkdir -p gpurun_out
for cfg in "128 4" "128 6" "128 8" "128 16" "128 32" "160 4" "160 8"; do
  set -- $cfg
  S=$1 B=$2 timeout 300 python tools/ab_streams.py 2>/dev/null
done > gpurun_out/r05_ab_streams.txt
cat gpurun_out/r05_ab_streams.txt
