#!/bin/bash
# SQ counters of the headline workload's kernels (one stream), two passes
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$repo" || exit 1
mkdir -p gpurun_out
B="python3 bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-micro --no-atlas --no-extras"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/sqh_$i
  timeout 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/sqh_$i -- $B > /dev/null 2> gpurun_out/sqh_$i.err
done
python3 tools/pmc_table.py gpurun_out/sqh_*/*/*_counter_collection.csv > gpurun_out/r05_sq_expmap_all.txt
rm -rf gpurun_out/sqh_?
grep -m1 "dispatch" gpurun_out/r05_sq_expmap_all.txt > gpurun_out/r05_sq_expmap.txt
for k in ad_star3 compose3 zy_forward zy_inverse fluid_xpass2; do grep "$k" gpurun_out/r05_sq_expmap_all.txt | tail -1 >> gpurun_out/r05_sq_expmap.txt; done
rm -f gpurun_out/r05_sq_expmap_all.txt
cat gpurun_out/r05_sq_expmap.txt | cut -c1-400
