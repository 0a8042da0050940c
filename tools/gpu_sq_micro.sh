#!/bin/bash
# Two rocprofv3 --pmc passes of SQ counters over tools/run_micro.py (every kernel at ONE batch size); per-dispatch table.
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$repo" || exit 1
mkdir -p gpurun_out
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/sqm_$i
  timeout 600 rocprofv3 --pmc $set --output-format csv -d gpurun_out/sqm_$i -- python3 tools/run_micro.py > /dev/null 2> gpurun_out/sqm_$i.err
done
python3 tools/pmc_table.py gpurun_out/sqm_1/*/*_counter_collection.csv gpurun_out/sqm_2/*/*_counter_collection.csv > gpurun_out/sq_micro.txt
rm -rf gpurun_out/sqm_1 gpurun_out/sqm_2
wc -l gpurun_out/sq_micro.txt
