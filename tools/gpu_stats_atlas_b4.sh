#!/bin/bash
# rocprofv3 kernel stats of the 160^3 atlas step at batch 4 (what each of 8 ranks sees) and batch 32, for the per-kernel small-batch accounting
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
for b in 4 32; do
rm -rf gpurun_out/stats_atlas_b$b
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_atlas_b$b -- python3 tools/run_atlas_step.py $b 160 > gpurun_out/stats_atlas_b$b.out 2> gpurun_out/stats_atlas_b$b.err
tail -1 gpurun_out/stats_atlas_b$b.out
python3 tools/rocprof_summary.py gpurun_out/stats_atlas_b$b/*/*_kernel_stats.csv | head -16
rm -f gpurun_out/stats_atlas_b$b/*/*_kernel_trace.csv
done
