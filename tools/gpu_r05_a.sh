#!/bin/bash
# round 5, first GPU call: the colour-phase probe, the tests touched so far, one bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 300 tools/probes/atomic_overlap > gpurun_out/r05_atomic_overlap.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_dispatch.py tests/test_gpu_lddmm_step.py -x -q -m gpu > gpurun_out/r05_tests_a.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "regrid or compose" >> gpurun_out/r05_tests_a.log 2>&1
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "lddmm_step_160" -s >> gpurun_out/r05_tests_a.log 2>&1
timeout 1200 python bench.py > gpurun_out/r05_bench_a.json 2> gpurun_out/r05_bench_a.err
tail -5 gpurun_out/r05_tests_a.log
cat gpurun_out/r05_atomic_overlap.txt
