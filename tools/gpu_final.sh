cd /root/repo
python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest7.log 2>&1; echo "pytest exit $?" 
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02_smoke.log 2>&1; echo "smoke exit $?"
python bench.py > gpurun_out/r02_bench6.json 2> gpurun_out/r02_bench6.err; echo "bench exit $?"
python bench.py --batch 8 --atlas-batch 8 --no-micro --no-cpu-baseline > gpurun_out/r02_bench6_b8.json 2>/dev/null
python bench.py --batch 4 --atlas-batch 4 --no-micro --no-cpu-baseline > gpurun_out/r02_bench6_b4.json 2>/dev/null
tail -3 gpurun_out/r02_pytest7.log; tail -2 gpurun_out/r02_smoke.log
