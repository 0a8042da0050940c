#!/usr/bin/env python3
"""The atlas matching step (`lddmm_step`: 5-step shoot, match, backward through every operator, momentum update) on ONE
stream against the sub-batch split over 2 / 4 HIP streams (`lddmm.LDDMM_STEP_STREAMS`): alternating rounds in one
process, results compared (loss, regulariser, updated momenta, atlas gradient).  env: S (160), B (8)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur
from lagomorph_amd import lddmm

S, B = int(os.environ.get("S", 160)), int(os.environ.get("B", 8))
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(4321)
metric = lm.FluidMetric([0.1, 0.0, 0.01])
I0 = gaussian_blur(torch.randn((1, 1, S, S, S), device=dev, generator=g), 3.0)
I0 = I0 / I0.std()
img = gaussian_blur(torch.randn((B, 1, S, S, S), device=dev, generator=g), 3.0)
img = img / img.std()
with torch.no_grad():
    m0 = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 4.0)
    m0 *= 3.0 / metric.sharp(m0).abs().max()


def run(parts, lr=1e-3):
    lddmm.LDDMM_STEP_STREAMS = parts
    I = I0.clone().requires_grad_(True)
    m = m0.clone()
    out, loss, reg = lm.lddmm_step(I, m, img, metric, dataset_size=B, integration_steps=5, learning_rate_pose=lr)
    return out, loss, reg, I.grad


default = lddmm.LDDMM_STEP_STREAMS
ref = run(1)
for parts in (1, 2, 4, 1, 2, 4):
    if B < 2 * parts:
        continue
    for _ in range(2):
        run(parts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        r = run(parts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 4
    errs = [float((a.double() - b.double()).abs().max() / b.double().abs().max()) for a, b in zip(r, ref)]
    print(f"S={S} B={B} streams={parts}: {dt * 1e3:7.2f} ms per step  ({B * S**3 / dt / 1e9:.3f} Gvoxel/s)   vs one stream: "
          f"m {errs[0]:.1e} loss {errs[1]:.1e} reg {errs[2]:.1e} I.grad {errs[3]:.1e}", flush=True)
lddmm.LDDMM_STEP_STREAMS = default
