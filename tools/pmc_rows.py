#!/usr/bin/env python3
"""A short fixed sequence of row-splat launches (full and ablated) for rocprofv3 --pmc; dispatch order is printed so the
per-dispatch counter rows can be matched."""
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
u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0)
u = u * (4.0 / u.abs().max())
go = torch.randn((B, 1, S, S, S), device=dev, generator=g)
ext.set_splat_rows(1, tx=4, ty=8, nthreads=512, vpl=4)
seq = [0, 1 | 32 | 4 | 8 | 16, 1 | 32 | 4, 8 | 16, 32 | 4, 1]
for need_u in (False, True):
    for mask in seq:
        ext._lib.lago_debug_splat_rows_ablate(mask)
        for _ in range(3):
            ext.interp_backward(go, I, u, 1.0, True, need_u)
        torch.cuda.synchronize()
        print("SEQ", need_u, mask)
ext._lib.lago_debug_splat_rows_ablate(0)
ext.set_splat_rows(0)
for need_u in (False, True):
    for _ in range(3):
        ext.interp_backward(go, I, u, 1.0, True, need_u)
    torch.cuda.synchronize()
