#!/bin/bash
# Two rocprofv3 --pmc passes of SQ counters over bench.py (expmap + micro-benchmarks); prints per-kernel
# averages.  Only counters that have been collected on this pool before: exotic TA/TD counters hung a box.
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$repo" || exit 1
mkdir -p gpurun_out
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/sq_$i
  timeout 600 rocprofv3 --pmc $set --output-format csv -d gpurun_out/sq_$i -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/sq_$i.err
done
python3 tools/pmc_table.py gpurun_out/sq_1/*/*_counter_collection.csv gpurun_out/sq_2/*/*_counter_collection.csv > gpurun_out/sq_counters.txt
rm -rf gpurun_out/sq_1 gpurun_out/sq_2
wc -l gpurun_out/sq_counters.txt
