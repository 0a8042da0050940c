#!/usr/bin/env python3
"""`lddmm.expmap` (10 Euler steps) on one 3D shape, 3 calls: for rocprofv3 --kernel-trace --stats.
usage: run_expmap_shape.py B n0 n1 n2   (e.g. 8 176 208 176)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import lagomorph_amd as lm

B = int(sys.argv[1])
sp = tuple(int(a) for a in sys.argv[2:5])
g = torch.Generator(device="cuda").manual_seed(3)
met = lm.FluidMetric([0.1, 0.0, 0.01])
m = bench.gaussian_blur(torch.randn((B, 3) + sp, device="cuda", generator=g), 4.0)
with torch.no_grad():
    m *= 2.5 / met.sharp(m).abs().max()
    lm.lddmm.EXPMAP_STREAMS = 1   # whole-batch launches on one stream: per-kernel durations without overlap
    for _ in range(3):
        lm.expmap(met, m, num_steps=10)
torch.cuda.synchronize()
