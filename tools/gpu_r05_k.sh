#!/bin/bash
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$repo" || exit 1
mkdir -p gpurun_out
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/ab_$i
  timeout 240 rocprofv3 --pmc $set --output-format csv -d gpurun_out/ab_$i -- python3 tools/run_affine_bwd.py > /dev/null 2> gpurun_out/ab_$i.err
done
python3 tools/pmc_table.py gpurun_out/ab_*/*/*_counter_collection.csv > gpurun_out/r05_affine_box_counters.txt
rm -rf gpurun_out/ab_?
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abt -- python3 tools/run_affine_bwd.py > /dev/null 2>&1
python3 tools/rocprof_summary.py gpurun_out/abt/*/*_kernel_stats.csv > gpurun_out/r05_affine_box_kernels.md
rm -rf gpurun_out/abt
cat gpurun_out/r05_affine_box_kernels.md | head -12
tail -5 gpurun_out/r05_affine_box_counters.txt | cut -c1-700
