#!/usr/bin/env python3
"""Profiling aid: interp_backward at C = 3 (batch 8 x 128^3) with the channel-by-channel and the
multi-channel single-pass splat kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lagomorph_amd as lm
from bench import gaussian_blur, time_op
ext = lm.lagomorph_ext
dev = torch.device("cuda")
S, B = 128, 8
g = torch.Generator(device=dev).manual_seed(1234)
u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0); u = u * (4.0 / u.abs().max())
for C in (2, 3):
    I = gaussian_blur(torch.randn((B, C, S, S, S), device=dev, generator=g), 2.0); I = I / I.std()
    go = torch.randn((B, C, S, S, S), device=dev, generator=g)
    res = {}
    for mc in (0, 1):
        ext.tune(splat_mc=mc)
        t, _ = time_op(lambda: ext.interp_backward(go, I, u, 1.0, True, True), reps=10, warm=3)
        res[mc] = (t, ext.interp_backward(go, I, u, 1.0, True, True))
        print(f"C={C} multi-channel={mc}: {t*1e3:.1f} us", flush=True)
    print("   d_u identical:", torch.equal(res[0][1][1], res[1][1][1]),
          " d_I max diff / max:", float((res[0][1][0] - res[1][1][0]).abs().max() / res[0][1][0].abs().max()))
ext.tune(splat_mc=1)
