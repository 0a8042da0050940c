#!/usr/bin/env python3
"""Per-call event timing vs back-to-back timing of the configs[1] kernels (host launch latency shows up in the first
when a kernel is short); run under rocprofv3 --kernel-trace --stats for the device-side durations."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur, time_op

ext = lm.lagomorph_ext
dev = torch.device("cuda")
S, B = 128, 8
g = torch.Generator(device=dev).manual_seed(1234)
I = gaussian_blur(torch.randn((B, 1, S, S, S), device=dev, generator=g), 2.0)
u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0)
u = u * (4.0 / u.abs().max())
go = torch.randn((B, 1, S, S, S), device=dev, generator=g)


def b2b(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


cases = {
    "interp_forward": lambda: ext.interp_forward(I, u, 1.0),
    "interp_backward general (d_I + d_u)": (0, lambda: ext.interp_backward(go, I, u, 1.0, True, True)),
    "interp_backward sheared (d_I + d_u)": (1, lambda: ext.interp_backward(go, I, u, 1.0, True, True)),
    "interp_backward general (d_I)": (0, lambda: ext.interp_backward(go, I, u, 1.0, True, False)),
    "interp_backward sheared (d_I)": (1, lambda: ext.interp_backward(go, I, u, 1.0, True, False)),
}
for name, fn in cases.items():
    if isinstance(fn, tuple):
        ext.set_splat_shear(fn[0])
        fn = fn[1]
    per_call, _ = time_op(fn, reps=20, warm=3)
    print(f"{name:40s} per-call events {per_call*1e3:7.1f} us   back-to-back {b2b(fn)*1e3:7.1f} us")
