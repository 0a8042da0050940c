#!/usr/bin/env python3
"""`compose` as the epilogue of the fluid metric's last pass (zy_inverse_compose_kernel), as a fusion.
Times sharp(m) + compose(v, phi, -dt, 1) (four launches) against lago_fluid_metric_compose (three launches) on the
headline workload's shapes, checks bit equality, and runs a whole 10-step shoot both ways.
usage: python tools/ab_compose_epilogue.py [batch] [size]   (needs the library of commit 3b31818: lago_fluid_metric_compose_f32)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import lagomorph_amd as lm

ext = lm.lagomorph_ext
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda")
torch.manual_seed(1234)
metric = lm.FluidMetric([0.1, 0.0, 0.01])
with torch.no_grad():
    m = bench.gaussian_blur(torch.randn((B, 3, S, S, S), device=dev), 4.0)
    m *= 5.0 / metric.sharp(m).abs().max()
    phi = lm.expmap(metric, m, num_steps=10) * 0.5      # a realistic phi^-1 halfway through the shoot
    luts, params = metric.luts, metric.params
    gen = luts.get("gen", 0)
    dt = 0.1

    def two():
        v = metric.sharp(m)
        return ext.compose(v, phi, -dt, 1.0)

    def fused():
        return ext.fluid_metric_compose(m, phi, True, luts["cos"], luts["sin"], *params, gen, -dt, 1.0)

    a, b = two(), fused()
    torch.cuda.synchronize()
    print(f"batch {B} x 3 x {S}^3: same bits: {bool(torch.equal(a, b))}  max |diff| {float((a - b).abs().max()):.3g}")
    for r in range(3):
        t2, _ = bench.time_op(two, reps=20, warm=10)
        tf, _ = bench.time_op(fused, reps=20, warm=10)
        ts, _ = bench.time_op(lambda: metric.sharp(m), reps=20, warm=10)
        v = metric.sharp(m)
        tc, _ = bench.time_op(lambda: ext.compose(v, phi, -dt, 1.0), reps=20, warm=10)
        print(f"  round {r}: sharp {ts * 1e3:.1f} us + compose {tc * 1e3:.1f} us; the two together {t2 * 1e3:.1f} us;"
              f" fused {tf * 1e3:.1f} us ({100 * (tf / t2 - 1):+.1f} %)")
