#!/usr/bin/env python3
"""The kernels of BASELINE configs[1] (interp + splat, batch 8 x 1 x 128^3) and of the backward path at batch 8 x 3 x 128^3,
a few launches each, for rocprofv3 --kernel-trace / --pmc passes: every kernel runs at ONE batch size here, so per-launch
averages are not mixtures."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur

ext = lm.lagomorph_ext
dev = torch.device("cuda")
S, B = 128, 8
g = torch.Generator(device=dev).manual_seed(1234)
I = gaussian_blur(torch.randn((B, 1, S, S, S), device=dev, generator=g), 2.0)
I = I / I.std()
u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0)
u = u * (4.0 / u.abs().max())
go = torch.randn((B, 1, S, S, S), device=dev, generator=g)
v3 = torch.randn((B, 3, S, S, S), device=dev, generator=g)
g3 = torch.randn((B, 3, S, S, S), device=dev, generator=g)
A = (torch.eye(3, device=dev)[None] + 0.05 * torch.randn((B, 3, 3), device=dev, generator=g)).contiguous()
T = torch.randn((B, 3), device=dev, generator=g)
for _ in range(5):
    ext.interp_forward(I, u, 1.0)                                  # interp_fwd3_unroll_kernel<float,false,2,true>, C = 1
    ext.interp_backward(go, I, u, 1.0, True, True)                 # splat_shear_kernel<1024,true,true,false,0>
    ext.interp_backward(g3, v3, u, 1.0, True, True)                # splat_shear_kernel<..., 4>: C = 3, d_u in registers
    ext.interp_backward(g3, v3, u, -0.2, True, True)               # splat_shear_mc_kernel<1024,false,false,2>: non-unit step
    ext.Ad_star(u, v3)                                             # ad_star3_tile_kernel
    ext.compose(u, v3, -0.1, 1.0)
    ext.jacobian_times_vectorfield_backward(g3, v3, u, True, False, True, True)
    ext.jacobian_times_vectorfield_forward(v3, u, True, False)
    ext.affine_interp_forward(I, A, T)
    ext.affine_interp_backward(go, I, A, T, True, True, True)
torch.cuda.synchronize()
