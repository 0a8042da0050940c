#!/usr/bin/env python3
"""The C = 1 splat with d_u at 8 x 128^3 (configs[1]) under lago_set_splat_shear_mc 4 / 3 / 2, five launches each, for
rocprofv3 --kernel-trace / --pmc passes.  env: S, B, C, DT, MODES"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur

ext = lm.lagomorph_ext
dev = torch.device("cuda")
S, B, C = int(os.environ.get("S", 128)), int(os.environ.get("B", 8)), int(os.environ.get("C", 1))
dt = float(os.environ.get("DT", 1.0))
g = torch.Generator(device=dev).manual_seed(1234)
I = gaussian_blur(torch.randn((B, C, S, S, S), device=dev, generator=g), 2.0)
I = I / I.std()
u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0)
u = u * (4.0 / u.abs().max())
go = torch.randn((B, C, S, S, S), device=dev, generator=g)
for mode in [int(x) for x in os.environ.get("MODES", "2").split(",")]:
    ext.set_splat_shear_mc(mode)
    for _ in range(5):
        ext.interp_backward(go, I, u, dt, True, True)
torch.cuda.synchronize()
