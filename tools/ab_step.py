#!/usr/bin/env python3
"""A/B helper (run once per library build, tools/ab_run.sh style): lddmm_step at 8 x S^3, several rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lagomorph_amd as lm
from bench import gaussian_blur
dev = torch.device("cuda")
metric = lm.FluidMetric([0.1, 0.0, 0.01])
S, B = int(os.environ.get("S", 160)), int(os.environ.get("B", 8))
g = torch.Generator(device=dev).manual_seed(4321)
I = gaussian_blur(torch.randn((1, 1, S, S, S), device=dev, generator=g), 3.0)
I = (I / I.std()).requires_grad_(True)
img = gaussian_blur(torch.randn((B, 1, S, S, S), device=dev, generator=g), 3.0)
img = img / img.std()
with torch.no_grad():
    m = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 4.0)
    m *= 3.0 / metric.sharp(m).abs().max()
out = []
for r in range(3):
    for _ in range(2):
        lm.lddmm_step(I, m, img, metric, dataset_size=B, integration_steps=5, learning_rate_pose=0.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6):
        lm.lddmm_step(I, m, img, metric, dataset_size=B, integration_steps=5, learning_rate_pose=0.0)
    torch.cuda.synchronize()
    out.append(f"{(time.perf_counter()-t0)/6*1e3:7.3f}")
print(sys.argv[1] if len(sys.argv) > 1 else "", f"lddmm_step {B} x {S}^3 ms:", " ".join(out))
