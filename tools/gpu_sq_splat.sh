#!/bin/bash
# rocprofv3 --pmc passes of SQ counters over tools/run_splat_modes.py; per-dispatch table -> gpurun_out/r04/sq_splat.txt
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$repo" || exit 1
mkdir -p gpurun_out/r04
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/sqs_$i
  timeout 600 rocprofv3 --pmc $set --output-format csv -d gpurun_out/sqs_$i -- python3 tools/run_splat_modes.py > /dev/null 2> gpurun_out/r04/sqs_$i.err
done
python3 tools/pmc_table.py gpurun_out/sqs_1/*/*_counter_collection.csv gpurun_out/sqs_2/*/*_counter_collection.csv > gpurun_out/r04/sq_splat${TAG}.txt
rm -rf gpurun_out/sqs_1 gpurun_out/sqs_2
grep -c . gpurun_out/r04/sq_splat${TAG}.txt
