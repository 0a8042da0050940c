#!/usr/bin/env python3
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lagomorph_amd as lm
from bench import gaussian_blur, time_op
ext = lm.lagomorph_ext; lib = ext._lib
dev = torch.device("cuda")
for (B, C) in ((8, 1), (32, 3)):
    S = 128
    g = torch.Generator(device=dev).manual_seed(1234)
    I = torch.randn((B, C, S, S, S), device=dev, generator=g)
    u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0); u = u * (4.0 / u.abs().max())
    V = B * S ** 3
    row = []
    for v, name in ((0, "full"), (1, "coalesced-instead-of-gather"), (2, "no-image")):
        lib.lago_debug_interp_variant(v)
        for vec in (1, 0):
            ext.set_vector_kernels(vec)
            med, _ = time_op(lambda: ext.interp_forward(I, u, 1.0), reps=10, warm=3)
            row.append(f"{name}{'' if vec else '(scalar)'}={med*1e3:.0f}us")
    lib.lago_debug_interp_variant(0); ext.set_vector_kernels(1)
    print(f"B={B} C={C}", " ".join(row))
    med, _ = time_op(lambda: torch.add(u, u), reps=10, warm=3); print("  torch add (3 streams of", u.numel()*4/1e6, "MB):", med*1e3, "us ->", 3*u.numel()*4/med/1e9, "TB/s")
