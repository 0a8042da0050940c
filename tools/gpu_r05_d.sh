#!/bin/bash
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$repo" || exit 1
mkdir -p gpurun_out
for v in old new2 new4 old new2 new4; do
  LAGO_HIP_LIBRARY=$PWD/lagomorph_amd/_lib/ab_$v.so timeout 300 python tools/ab_forward.py $v 2>/dev/null
done > gpurun_out/r05_ab_forward3.txt
cat gpurun_out/r05_ab_forward3.txt
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf gpurun_out/fw_$i
  timeout 600 rocprofv3 --pmc $set --output-format csv -d gpurun_out/fw_$i -- python3 tools/run_forward_ops.py > /dev/null 2> gpurun_out/fw_$i.err
done
python3 tools/pmc_table.py gpurun_out/fw_*/*/*_counter_collection.csv > gpurun_out/r05_forward_counters.txt
rm -rf gpurun_out/fw_?
tail -4 gpurun_out/r05_forward_counters.txt | cut -c1-600
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "regrid or compose or affine" > gpurun_out/r05_tests_d.log 2>&1
grep -E "passed|failed" gpurun_out/r05_tests_d.log
