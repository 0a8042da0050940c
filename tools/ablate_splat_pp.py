#!/usr/bin/env python3
"""Profile by ablation: the persistent image-window splat with stages switched off (PROFILING library only; results are
WRONG with bits set).  1 LDS adds, 2 corner reads + gradient, 4 image-window loads, 8 flush atomics, 16 d_u stores,
32 voxel operand loads, 64 the whole flush.
    LAGO_HIP_LIBRARY=$PWD/lagomorph_amd/_lib/liblagomorph_hip_prof.so python tools/ablate_splat_pp.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur, time_op

ext = lm.lagomorph_ext
lib = ext._lib
assert hasattr(lib, "lago_debug_splat_skip"), "needs the profiling build (python -m lagomorph_amd.build --profiling)"
dev = torch.device("cuda")
S, B, C = int(os.environ.get("S", 128)), int(os.environ.get("B", 8)), int(os.environ.get("C", 1))
g = torch.Generator(device=dev).manual_seed(1234)
I = gaussian_blur(torch.randn((B, C, S, S, S), device=dev, generator=g), 2.0)
u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0)
u = u * (4.0 / u.abs().max())
go = torch.randn((B, C, S, S, S), device=dev, generator=g)
ext.set_splat_shear_mc(4)
masks = [0, 1, 2, 3, 4, 8, 16, 32, 64, 8 | 16, 4 | 32, 4 | 16 | 32, 1 | 2 | 64, 4 | 8 | 16 | 32, 1 | 2 | 4 | 8 | 16 | 32, 127]
names = {1: "adds", 2: "corners", 4: "imgDMA", 8: "atomics", 16: "du_st", 32: "vox_ld", 64: "flush"}
for r in range(2):
    for m in masks:
        lib.lago_debug_splat_skip(m)
        med, _ = time_op(lambda: ext.interp_backward(go, I, u, 1.0, True, True), reps=20, warm=10)
        off = "+".join(n for b, n in names.items() if m & b) or "nothing"
        print(f"round {r} skip {m:3d} ({off:40s}): {med * 1e3:7.1f} us")
lib.lago_debug_splat_skip(0)
