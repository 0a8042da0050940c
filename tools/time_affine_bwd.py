#!/usr/bin/env python3
"""affine_interp_backward at 8 x 1 x 128^3 (bench.py's other_ops shape): image splat by target boxes (lago_tuning.affine_box = 1)
against the general tiled splat (0), for a near-identity matrix and a 20-degree rotation; d_I only and the whole call."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import lagomorph_amd as lm
from bench import time_op

ext = lm.lagomorph_ext
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(5)
B, S = 8, 128
I = torch.randn((B, 1, S, S, S), device=dev, generator=g)
go = torch.randn((B, 1, S, S, S), device=dev, generator=g)
T = torch.randn((B, 3), device=dev, generator=g)
c, s = np.cos(np.deg2rad(20)), np.sin(np.deg2rad(20))
mats = {"near identity (I + 0.05 randn)": (torch.eye(3, device=dev)[None] + 0.05 * torch.randn((B, 3, 3), device=dev, generator=g)).contiguous(),
        "rotation 20 deg about x": torch.tensor([[1, 0, 0], [0, c, -s], [0, s, c]], device=dev, dtype=torch.float32)[None].repeat(B, 1, 1).contiguous()}
for name, A in mats.items():
    for box in (1, 0, 1, 0):
        ext.tune(affine_box=box)
        t_img, _ = time_op(lambda: ext.affine_interp_backward(go, I, A, T, True, False, False), reps=20, warm=5)
        t_all, _ = time_op(lambda: ext.affine_interp_backward(go, I, A, T, True, True, True), reps=20, warm=5)
        print(f"{name:32s} {'target boxes' if box else 'tiled splat '}: d_I only {t_img * 1e3:7.1f} us   d_I + d_A + d_T {t_all * 1e3:7.1f} us")
ext.tune(affine_box=1)
