#!/usr/bin/env python3
"""sharp through the generic FFT passes, fused x pass (fluid_mode 3) against separate launches (4), alternated five times,
median of 30 each."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import time_op

ext = lm.lagomorph_ext
g = torch.Generator(device="cuda").manual_seed(3)
met = lm.FluidMetric([0.1, 0.0, 0.01])
for sp, B, dt in (((128, 128, 128), 8, torch.float64), ((160, 160, 160), 4, torch.float64), ((120, 120, 120), 8, torch.float32),
                  ((182, 218, 182), 2, torch.float32), ((100, 120, 60), 8, torch.float32), ((96, 96, 96), 8, torch.float64),
                  ((256, 128, 128), 4, torch.float64), ((1024, 1024), 16, torch.float32), ((512, 512), 32, torch.float64),
                  ((256, 256), 64, torch.float32)):
    x = torch.randn((B, len(sp)) + sp, device="cuda", generator=g, dtype=dt)
    t = {3: [], 4: []}
    with torch.no_grad():
        for rep in range(5):
            for mode in (3, 4):
                ext.set_fluid_mode(mode)
                t[mode].append(time_op(lambda: met.sharp(x), reps=30, warm=10)[0] * 1e3)
        ext.set_fluid_mode(3)
    f, s = sorted(t[3])[2], sorted(t[4])[2]
    print(f"{str(sp):16s} x{B} {str(dt)[6:]:8s}: fused {f:8.1f} us [{min(t[3]):.1f}..{max(t[3]):.1f}]  separate {s:8.1f} us [{min(t[4]):.1f}..{max(t[4]):.1f}]  {100*(f/s-1):+.1f} %", flush=True)
