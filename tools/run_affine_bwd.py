#!/usr/bin/env python3
"""A few launches of affine_interp_backward (d_I only) at 8 x 1 x 128^3, near-identity matrices (profiling target)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm

ext = lm.lagomorph_ext
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(5)
I = torch.randn((8, 1, 128, 128, 128), device=dev, generator=g)
go = torch.randn((8, 1, 128, 128, 128), device=dev, generator=g)
T = torch.randn((8, 3), device=dev, generator=g)
A = (torch.eye(3, device=dev)[None] + 0.05 * torch.randn((8, 3, 3), device=dev, generator=g)).contiguous()
for _ in range(4):
    ext.affine_interp_backward(go, I, A, T, True, False, False)
torch.cuda.synchronize()
