#!/usr/bin/env python3
"""interp_forward with the LDS window off / on (lago_set_gather_window) in one process: configs[1] forward (8 x 1 x 128^3),
C = 3, a broadcast image at 160^3, displacement amplitudes 1 / 4 / 10 voxels; bit compare."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur, time_op

ext = lm.lagomorph_ext
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1234)
for S, B, C, bc in ((128, 8, 1, False), (128, 8, 3, False), (160, 8, 1, True), (128, 32, 3, False)):
    I = gaussian_blur(torch.randn(((1 if bc else B), C, S, S, S), device=dev, generator=g), 2.0)
    u0 = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0)
    for amp in (1.0, 4.0, 10.0):
        u = u0 * (amp / u0.abs().max())
        outs = {}
        line = f"{B} x {C} x {S}^3 bc={int(bc)} amp {amp:4.1f}:"
        for mode in (0, 1, 0, 1):
            ext.set_gather_window(mode)
            outs[mode] = ext.interp_forward(I, u, 1.0)
            t, _ = time_op(lambda: ext.interp_forward(I, u, 1.0), reps=30, warm=20)
            line += f"  window {mode}: {t*1e3:6.1f} us"
        print(line, " bits", "same" if torch.equal(outs[0], outs[1]) else "DIFFER", flush=True)
ext.set_gather_window(1)
