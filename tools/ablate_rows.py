#!/usr/bin/env python3
"""Where the row-mapped splat spends its time: switch phases off one at a time (results are wrong by design;
only the durations matter) on the configs[1] workload, plus reference streaming rates of this box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur, time_op

ext = lm.lagomorph_ext
lib = ext._lib
dev = torch.device("cuda")
S, B = 128, 8
g = torch.Generator(device=dev).manual_seed(1234)
I = gaussian_blur(torch.randn((B, 1, S, S, S), device=dev, generator=g), 2.0)
u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0)
u = u * (4.0 / u.abs().max())
go = torch.randn((B, 1, S, S, S), device=dev, generator=g)
V = B * S ** 3
a, b = torch.empty_like(u), torch.empty_like(u)
t, _ = time_op(lambda: torch.add(u, 1.0, out=a), reps=20)
print(f"torch add 3ch (24 B/voxel): {t*1e3:.1f} us = {24*V/t/1e6:.0f} GB/s")
t, _ = time_op(lambda: a.zero_(), reps=20)
print(f"zero 3ch (12 B/voxel): {t*1e3:.1f} us = {12*V/t/1e6:.0f} GB/s")
z1 = torch.empty_like(go)
t, _ = time_op(lambda: z1.zero_(), reps=20)
print(f"zero 1ch (the d_I memset): {t*1e3:.1f} us")
names = {0: "full", 1: "no LDS adds", 2: "no flush atomics", 4: "no zeroing", 8: "no u/g loads", 16: "no probe",
         32: "no flush pass", 1 | 32 | 4: "no adds, no flush, no zero", 1 | 32 | 4 | 8 | 16: "VALU + stores only",
         2 | 8: "no atomics, no loads", 8 | 16: "no loads, no probe", 32 | 4: "no flush, no zero"}
for cfg in (dict(tx=4, ty=4, nthreads=512, vpl=1), dict(tx=4, ty=8, nthreads=512, vpl=4)):
    ext.set_splat_rows(1, **cfg)
    print("config", cfg)
    for need_u in (True, False):
        for mask, name in names.items():
            lib.lago_debug_splat_rows_ablate(mask)
            t, _ = time_op(lambda: ext.interp_backward(go, I, u, 1.0, True, need_u), reps=10, warm=2)
            print(f"  need_u={need_u!s:5} {name:28s} {t*1e3:7.1f} us")
lib.lago_debug_splat_rows_ablate(0)
