#!/bin/bash
# Round-end check on the GPU box: full GPU suite (with the observed-error reports), smoke(), bench at three batch sizes.
cd "$(dirname "$0")/.."
tag=${LAGO_ROUND_TAG:-rXX}
mkdir -p gpurun_out
LAGO_TOL_REPORT=gpurun_out/${tag}_tolerances.json LAGO_TOL_REPORT_GOLDEN=gpurun_out/${tag}_tolerances_golden.json python -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest exit $?"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_smoke.log 2>&1; echo "smoke exit $?"
if [ "$1" != "nobench" ]; then
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench exit $?"
python bench.py --batch 8 --atlas-batch 8 --no-epoch --no-micro --no-cpu-baseline > gpurun_out/${tag}_bench_b8.json 2>/dev/null
python bench.py --batch 4 --atlas-batch 4 --no-epoch --no-micro --no-cpu-baseline > gpurun_out/${tag}_bench_b4.json 2>/dev/null
# the N-rank code path (spawned ranks, barriers, the all-reduce hook, max over ranks) on ONE GPU over gloo: a plumbing check
LAGO_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 2 --warmup 1 --atlas-steps 2 --no-epoch --no-micro --no-cpu-baseline > gpurun_out/${tag}_bench_share2.json 2> gpurun_out/${tag}_bench_share2.err; echo "share-gpu 2-rank bench exit $?"
fi
tail -3 gpurun_out/${tag}_pytest.log; tail -2 gpurun_out/${tag}_smoke.log
