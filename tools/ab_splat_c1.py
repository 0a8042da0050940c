#!/usr/bin/env python3
"""interp_backward C = 1 (configs[1]: d_I and d_u, memset included) for ONE library build, at 8 x 128^3 and 8 x 160^3 on
bench.py's smooth field; prints medians and a checksum of d_u (bit-comparable across builds) and of d_I.
usage: LAGO_HIP_LIBRARY=... python tools/ab_splat_c1.py <tag>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import lagomorph_amd as lm

tag = sys.argv[1] if len(sys.argv) > 1 else "lib"
ext = lm.lagomorph_ext
dev = torch.device("cuda")
out = []
for size in (128, 160):
    g = torch.Generator(device=dev).manual_seed(1234)
    I = bench.gaussian_blur(torch.randn((8, 1, size, size, size), device=dev, generator=g), 2.0)
    I = I / I.std()
    u = bench.gaussian_blur(torch.randn((8, 3, size, size, size), device=dev, generator=g), 8.0)
    u = u * (4.0 / u.abs().max())
    go = torch.randn((8, 1, size, size, size), device=dev, generator=g)
    for dt in (1.0, -0.3):
        med, _ = bench.time_op(lambda: ext.interp_backward(go, I, u, dt, True, True), reps=30, warm=30)
        dI, du = ext.interp_backward(go, I, u, dt, True, True)
        out.append(f"{size}^3 dt {dt:+.1f}: {med * 1e3:6.1f} us  d_u {du.double().sum().item():+.10e} d_I {dI.double().abs().sum().item():.8e}")
print(f"{tag:>5s}: " + " | ".join(out), flush=True)
