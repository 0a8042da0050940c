#!/bin/bash
# PMC passes over tools/run_compose.py: HBM traffic, SQ issue mix, TCP / TCC behaviour of the pair and window composes.
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$repo" || exit 1
mkdir -p gpurun_out
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/cmp_$i
  timeout 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/cmp_$i -- python3 tools/run_compose.py "$@" > /dev/null 2> gpurun_out/cmp_$i.err
done
python3 tools/pmc_table.py gpurun_out/cmp_*/*/*_counter_collection.csv > gpurun_out/pmc_compose.txt
rm -rf gpurun_out/cmp_?
grep -i "compose" gpurun_out/pmc_compose.txt | head -40
