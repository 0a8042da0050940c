#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for v in v1 v2 v4 v6 v8 v4nt v1 v4; do
  LAGO_HIP_LIBRARY=$PWD/lagomorph_amd/_lib/ab_$v.so timeout 300 python tools/ab_forward.py $v 2>/dev/null
done > gpurun_out/r05_ab_forward.txt
cat gpurun_out/r05_ab_forward.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "regrid or compose or affine" > gpurun_out/r05_tests_b.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_dispatch.py -x -q -m gpu >> gpurun_out/r05_tests_b.log 2>&1
grep -E "passed|failed" gpurun_out/r05_tests_b.log
