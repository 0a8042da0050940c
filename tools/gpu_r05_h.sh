#!/bin/bash
# round 5: full GPU suite + smoke + bench lines + profiles
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$repo" || exit 1
LAGO_ROUND_TAG=r05 bash tools/gpu_final.sh
bash tools/gpu_profile.sh r05 > gpurun_out/r05_profile.log 2>&1
tail -30 gpurun_out/r05_profile.log
