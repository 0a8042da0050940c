#!/usr/bin/env python3
"""compose / interp_forward(C = 3) / the headline shoot of ONE library build (LAGO_HIP_LIBRARY picks it): for alternating
builds with different gather-window tiles (-DLAGO_GW_TY=8 against 16) on one box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur, time_op

tag = sys.argv[1] if len(sys.argv) > 1 else "lib"
ext = lm.lagomorph_ext
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(3)
out = []
for S, B in ((128, 32), (160, 8)):
    u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 6.0)
    v = torch.randn((B, 3, S, S, S), device=dev, generator=g)
    for amp in (0.5, 4.0):
        uu = u * (amp / u.abs().max())
        a, _ = time_op(lambda: ext.compose(uu, v, -0.1, 1.0), reps=30, warm=20)
        b, _ = time_op(lambda: ext.interp_forward(v, uu, 1.0), reps=30, warm=20)
        out.append(f"{B}x{S}^3 amp {amp}: compose {a*1e3:6.1f}  interp3 {b*1e3:6.1f}")
    del u, v, uu
    torch.cuda.empty_cache()
metric = lm.FluidMetric([0.1, 0.0, 0.01])
with torch.no_grad():
    m = gaussian_blur(torch.randn((32, 3, 128, 128, 128), device=dev), 4.0)
    m *= 5.0 / metric.sharp(m).abs().max()
    for _ in range(3):
        lm.expmap(metric, m, num_steps=10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        lm.expmap(metric, m, num_steps=10)
    torch.cuda.synchronize()
    out.append(f"shoot {(time.perf_counter() - t0) / 5 * 1e3:6.2f} ms")
print(f"{tag:>6s} | " + " | ".join(out), flush=True)
