#!/bin/bash
# The round-end GPU run (on the box, through gpurun, from the repo root):  tools/gpu_round.sh <tag, e.g. r06> [nobench]
#   full `pytest -m gpu` with the observed-error reports, smoke(), bench.py at global batch 32 / 8 / 4 (gpu_final.sh),
#   then the rocprofv3 kernel-trace / PMC passes whose summaries are copied to profiles/ (gpu_profile.sh).
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$repo" || exit 1
tag=${1:-rXX}
LAGO_ROUND_TAG=$tag bash tools/gpu_final.sh $2
bash tools/gpu_profile.sh $tag > gpurun_out/${tag}_profile.log 2>&1
tail -30 gpurun_out/${tag}_profile.log
