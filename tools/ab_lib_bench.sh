#!/bin/bash
# usage (GPU box): tools/ab_lib_bench.sh tagA tagB ...  -- bench.py (micro ops, atlas steps) per library build ab_<tag>.so, one line each
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/abbench
for v in "$@"; do
  LAGO_HIP_LIBRARY=$PWD/lagomorph_amd/_lib/ab_$v.so timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --atlas-steps 3 > gpurun_out/abbench/$v.json 2> gpurun_out/abbench/$v.err < /dev/null
  python3 - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads(open(f"gpurun_out/abbench/{v}.json").read().strip().splitlines()[-1])
o = d["other_ops"]["ops"]
us = lambda k: round(o[k]["ms"] * 1e3, 1)
print(f"{v:>6s}: shoot {d['ms_per_step']:.2f} atlas160 {d['atlas_step']['ms_per_step']:.2f} atlas128 {d['atlas_step_128']['ms']:.3f} | pair {d['interp_splat']['smooth']['pair_ms']*1e3:.1f} fwd {d['interp_splat']['smooth']['fwd_ms']*1e3:.1f} bwd {d['interp_splat']['smooth']['bwd_lds_ms']*1e3:.1f} | "
      f"jtv {us('jtv_forward(disp)')} jtvT {us('jtv_forward(transpose)')} jtvb {us('jtv_backward')} adjf {us('jtv_adjoint_forward')} adjb {us('jtv_adjoint_backward')} "
      f"if3 {us('interp_forward(C=3)')} ib3 {us('interp_backward(C=3)')} comp {us('compose')} ad {us('ad_star(fused interp+jtv)')}", flush=True)
PY
done
