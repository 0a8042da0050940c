#!/usr/bin/env python3
"""A/B in one process of the launch-direction alternation (lago_set_launch_order): expmap (10 Euler steps, 128^3) at
several batch sizes and the 160^3 atlas step at batch 8; results of the shoot compared bit for bit."""
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


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for B in [int(x) for x in os.environ.get("BATCHES", "32,8,4").split(",")]:
    with torch.no_grad():
        m = gaussian_blur(torch.randn((B, 3, 128, 128, 128), device=dev), 4.0)
        m *= 5.0 / metric.sharp(m).abs().max()
        res = {}
        for rep in range(2):
            for alt in (0, 1):
                ext.set_launch_order(alt)
                h = lm.expmap(metric, m, num_steps=10)
                same = "first" if alt not in res and not res else ("same" if torch.equal(h, next(iter(res.values()))) else "DIFFER")
                res[alt] = h
                t = timed(lambda: lm.expmap(metric, m, num_steps=10), 8 if B >= 16 else 20)
                print(f"expmap B={B:2d} alternate={alt}: {t:8.3f} ms  {B*128**3*10/t/1e6:6.2f} Gvox-step/s  bits {same}", flush=True)
        del m, res, h
        torch.cuda.empty_cache()

S, B = 160, 8
g = torch.Generator(device=dev).manual_seed(4321)
I = gaussian_blur(torch.randn((1, 1, S, S, S), device=dev, generator=g), 3.0)
I = (I / I.std()).requires_grad_(True)
img = gaussian_blur(torch.randn((B, 1, S, S, S), device=dev, generator=g), 3.0)
img = img / img.std()
with torch.no_grad():
    m = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 4.0)
    m *= 3.0 / metric.sharp(m).abs().max()
for rep in range(2):
    for alt in (0, 1):
        ext.set_launch_order(alt)
        t = timed(lambda: lm.lddmm_step(I, m, img, metric, dataset_size=B, integration_steps=5, learning_rate_pose=0.0), 5)
        print(f"lddmm_step 8 x 160^3 alternate={alt}: {t:8.3f} ms", flush=True)
ext.set_launch_order(1)
