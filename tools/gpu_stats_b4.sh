cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/stats_b4
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_b4 -- python3 bench.py --batch 4 --steps 4 --warmup 2 --no-cpu-baseline --no-micro --no-atlas > gpurun_out/stats_b4.out 2> gpurun_out/stats_b4.err
python3 tools/rocprof_summary.py gpurun_out/stats_b4/*/*_kernel_stats.csv | head -12
rm -f gpurun_out/stats_b4/*/*_kernel_trace.csv
