#!/usr/bin/env python3
"""The generic FFT path's fused x pass (x forward + operator + x inverse in one launch, fluid_mode 3) against the three
separate launches (fluid_mode 4): bit comparison of sharp / flat and timings, float32 and float64, shapes that take the
generic passes (float64 everywhere; float32 off the tuned lengths)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import time_op

ext = lm.lagomorph_ext
g = torch.Generator(device="cuda").manual_seed(3)
cases = [((128, 128, 128), 8, torch.float64), ((160, 160, 160), 4, torch.float64), ((64, 48, 80), 4, torch.float64),
         ((33, 29, 31), 4, torch.float64), ((182, 218, 182), 1, torch.float32), ((100, 120, 60), 4, torch.float32),
         ((120, 120, 120), 8, torch.float32), ((176, 208, 176), 1, torch.float32), ((91, 77, 55), 2, torch.float32),
         ((7, 9, 6), 3, torch.float64), ((2, 3, 4), 2, torch.float32), ((59, 64, 64), 2, torch.float32), ((128, 128, 128), 8, torch.float32),
         ((256, 256), 8, torch.float32), ((256, 256), 8, torch.float64), ((100, 90), 4, torch.float32), ((33, 21), 3, torch.float64),
         ((512, 384), 4, torch.float32), ((26, 30), 2, torch.float32), ((59, 40), 2, torch.float64)]
bad = 0
for sp, B, dt in cases:
    for params in ([0.1, 0.0, 0.01], [0.1, 0.05, 0.01]):
        met = lm.FluidMetric(params)
        x = torch.randn((B, len(sp)) + sp, device="cuda", generator=g, dtype=dt)
        res, tim, paths = {}, {}, {}
        with torch.no_grad():
            for mode in (3, 4):
                ext.set_fluid_mode(mode)
                before = ext.path_launches()
                res[mode] = (met.sharp(x), met.flat(x))
                after = ext.path_launches()
                paths[mode] = [k for k in after if after[k] != before[k] and k.startswith("fluid")]
                if params[1] == 0.0:
                    tim[mode] = time_op(lambda: met.sharp(x), reps=10, warm=3)[0]
            ext.set_fluid_mode(3)
        same = torch.equal(res[3][0], res[4][0]) and torch.equal(res[3][1], res[4][1])
        bad += 0 if same else 1
        t = f"  sharp fused {tim[3]*1e3:8.1f} us  separate {tim[4]*1e3:8.1f} us  ({100*(tim[3]/tim[4]-1):+.1f} %)" if tim else ""
        print(f"{'ok ' if same else 'BAD'} {str(sp):16s} x{B} {str(dt)[6:]:8s} beta {params[1]}: bits {'same' if same else 'DIFFER'} {paths[3]}{t}", flush=True)
print("BITS", "ok" if not bad else f"{bad} DIFFER")
