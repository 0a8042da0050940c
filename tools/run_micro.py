#!/usr/bin/env python3
"""Run the configs[1] kernels a few times (for rocprofv3 counter passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lagomorph_amd as lm
from bench import gaussian_blur
ext = lm.lagomorph_ext
dev = torch.device("cuda")
S, B = 128, 8
g = torch.Generator(device=dev).manual_seed(1234)
I = gaussian_blur(torch.randn((B, 1, S, S, S), device=dev, generator=g), 2.0); I = I / I.std()
u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0); u = u * (4.0 / u.abs().max())
go = torch.randn((B, 1, S, S, S), device=dev, generator=g)
v3 = torch.randn((B, 3, S, S, S), device=dev, generator=g)
for _ in range(5):
    ext.interp_forward(I, u, 1.0)
    ext.interp_backward(go, I, u, 1.0, True, True)
    ext.interp_backward(go, I, u, 1.0, True, False)
    ext.jacobian_times_vectorfield_forward(v3, u, True, False)
    ext.compose(u, v3, -0.1, 1.0)
torch.cuda.synchronize()
