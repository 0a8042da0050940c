#!/usr/bin/env python3
"""lddmm_step (8 x 160^3, 5 integration steps) under the multi-channel splat forms (lago_set_splat_shear_mc 1 / 2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lagomorph_amd as lm
from bench import gaussian_blur
ext = lm.lagomorph_ext
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
def step():
    lm.lddmm_step(I, m, img, metric, dataset_size=B, integration_steps=5, learning_rate_pose=0.0)
cfgs = [tuple(int(x) for x in c.split("x")) for c in os.environ.get("SHEARS", "8x6x0x1x1x4").split(",")]
for rep in range(2):
    for cfg in cfgs:
        ext.set_splat_shear(1, *cfg, 1024)
        for mc in [int(x) for x in os.environ.get("MCS", "1,2").split(",")]:
            ext.set_splat_shear_mc(mc)
            for _ in range(2): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(6): step()
            torch.cuda.synchronize()
            print(f"{S}^3 B={B} shear={cfg} shear_mc={mc}: {(time.perf_counter()-t0)/6*1e3:8.3f} ms", flush=True)
ext.set_splat_shear_mc(2)
ext.set_splat_shear(1, 8, 6, 0, 1, 1, 4, 1024)
