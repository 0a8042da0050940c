#!/usr/bin/env python3
"""Ablation timing of the tiled splat (results invalid when mask != 0)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lagomorph_amd as lm
from bench import gaussian_blur, time_op
ext = lm.lagomorph_ext
lib = ext._lib
lib.lago_debug_splat_ablate.argtypes = [ctypes.c_int]
dev = torch.device("cuda")
S, B = 128, 8
g = torch.Generator(device=dev).manual_seed(1234)
I = gaussian_blur(torch.randn((B, 1, S, S, S), device=dev, generator=g), 2.0); I = I / I.std()
u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0); u = u * (4.0 / u.abs().max())
go = torch.randn((B, 1, S, S, S), device=dev, generator=g)
import itertools
for mode in (1,):
  ext.set_splat_mode(mode)
  print("splat mode", mode)
  for cfg in ((8, 8, 32, 1, 1, 4, 512), (8, 8, 32, 1, 1, 4, 256), (8, 4, 32, 1, 1, 4, 256), (16, 8, 64, 1, 1, 4, 1024), (8, 8, 64, 1, 1, 4, 512), (4, 4, 64, 1, 1, 4, 256), (4, 4, 128, 1, 1, 4, 512)):
    ext.set_splat_tile(*cfg)
    row = []
    for mask, name in ((0, "full"), (1, "-lds"), (2, "-fallback"), (4, "-flush"), (7, "-all")):
        lib.lago_debug_splat_ablate(mask)
        for need_u in (False, True):
            med, _ = time_op(lambda: ext.interp_backward(go, I, u, 1.0, True, need_u), reps=8, warm=2)
            row.append(f"{name}{'+u' if need_u else ''}={med*1e3:.0f}")
    lib.lago_debug_splat_ablate(0)
    print("  ", cfg, " ".join(row))
# memset alone
med, _ = time_op(lambda: torch.zeros_like(I), reps=8, warm=2); print("zeros_like(I) us", med*1e3)
