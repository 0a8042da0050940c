#!/usr/bin/env python3
"""A few launches of affine_interp_forward / regrid_forward at bench.py's `other_ops` shapes (profiling target)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm

ext = lm.lagomorph_ext
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(99)
I1 = torch.randn((8, 1, 128, 128, 128), device=dev, generator=g)
A = (torch.eye(3, device=dev)[None] + 0.05 * torch.randn((8, 3, 3), device=dev, generator=g)).contiguous()
T = torch.randn((8, 3), device=dev, generator=g)
small = torch.randn((8, 3, 64, 64, 64), device=dev, generator=g)
for _ in range(6):
    ext.affine_interp_forward(I1, A, T)
    ext.regrid_forward(small, [128] * 3, [31.5] * 3, [63 / 127] * 3)
torch.cuda.synchronize()
