#!/usr/bin/env python3
"""Peak allocated / reserved memory of a loop that alternates a forward-only expmap (two sub-batches on side streams,
lddmm.EXPMAP_STREAMS = 2, or one stream) with a matching step on the caller's stream (the side streams'
allocator pools are not shared with the caller's).   env: S (128), B (32)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur
from lagomorph_amd import lddmm

S, B = int(os.environ.get("S", 128)), int(os.environ.get("B", 32))
dev = torch.device("cuda")
met = lm.FluidMetric([0.1, 0.0, 0.01])
g = torch.Generator(device=dev).manual_seed(3)
m = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 4.0)
with torch.no_grad():
    m *= 3.0 / met.sharp(m).abs().max()
I = gaussian_blur(torch.randn((1, 1, S, S, S), device=dev, generator=g), 3.0).requires_grad_(True)
img = gaussian_blur(torch.randn((B, 1, S, S, S), device=dev, generator=g), 3.0)
base = torch.cuda.memory_allocated() / 1e9
print(f"resident inputs {base:.2f} GB ({B} x {S}^3)")
for streams in (1, 2, 1, 2):
    lddmm.EXPMAP_STREAMS = streams
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    for _ in range(3):
        with torch.no_grad():
            h = lm.expmap(met, m, num_steps=10)
        del h
        lm.lddmm_step(I, m.clone(), img, met, B, integration_steps=5, learning_rate_pose=0.0)
    torch.cuda.synchronize()
    print(f"EXPMAP_STREAMS {streams}: peak allocated {torch.cuda.max_memory_allocated()/1e9:.2f} GB, peak reserved "
          f"{torch.cuda.max_memory_reserved()/1e9:.2f} GB, reserved at rest {torch.cuda.memory_reserved()/1e9:.2f} GB", flush=True)
lddmm.EXPMAP_STREAMS = 2
