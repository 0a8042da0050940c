#!/bin/bash
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$repo" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_lddmm_step.py tests/test_atlas_golden.py tests/test_affine_atlas_golden.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r05_tests_j.log 2>&1
tail -5 gpurun_out/r05_tests_j.log
