#!/bin/bash
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$repo" || exit 1
mkdir -p gpurun_out
for cfg in "160 4" "160 8" "160 16" "160 32" "128 4" "128 8" "128 32"; do
  set -- $cfg
  S=$1 B=$2 timeout 600 python tools/ab_step_streams.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05_ab_step_streams.txt
cat gpurun_out/r05_ab_step_streams.txt
