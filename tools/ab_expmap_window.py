#!/usr/bin/env python3
"""In-process A/B of lago_set_gather_window on the headline shoot (bench.py's workload: expmap, 10 Euler steps,
batch 32 x 3 x 128^3, momentum scaled to a 5-voxel deformation) and on the configs[4] shoot (8 x 160^3)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur

ext = lm.lagomorph_ext
dev = torch.device("cuda")
metric = lm.FluidMetric([0.1, 0.0, 0.01])
for B, S, E, amp in ((32, 128, 10, 5.0), (32, 128, 10, 15.0), (8, 160, 10, 5.0)):
    torch.manual_seed(1234)
    with torch.no_grad():
        m = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev), 4.0)
        m *= amp / metric.sharp(m).abs().max()
        res = {0: [], 1: []}
        outs = {}
        for r in range(3):
            for mode in (0, 1):
                ext.set_gather_window(mode)
                for _ in range(2):
                    h = lm.expmap(metric, m, num_steps=E)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    h = lm.expmap(metric, m, num_steps=E)
                torch.cuda.synchronize()
                res[mode].append((time.perf_counter() - t0) / 5 * 1e3)
                outs[mode] = h
        ext.set_gather_window(1)
        print(f"expmap {B} x {S}^3, {E} steps, |h|max {float(outs[1].abs().max()):.2f}: pair "
              + " ".join(f"{x:.2f}" for x in res[0]) + " ms | window " + " ".join(f"{x:.2f}" for x in res[1])
              + f" ms | same bits {torch.equal(outs[0], outs[1])}", flush=True)
    del m, h, outs
