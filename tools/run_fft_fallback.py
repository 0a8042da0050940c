#!/usr/bin/env python3
"""What the rocFFT fallback of the fluid metric costs where the hand-written passes do not apply (2D fields, float64,
extents outside 2^a / 3*2^a / 5*2^a): time per sharp and achieved GB/s of the 72.8 B/voxel single-pass ideal (scaled
to the dtype and dimension), next to the hand-written float32 3D passes at the same voxel count."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import time_op

dev = torch.device("cuda")
cases = [
    ("3D f32 8x3x128^3 (hand-written passes)", (8, 3, 128, 128, 128), torch.float32),
    ("3D f64 8x3x128^3 (rocFFT 3D plan + operator kernel)", (8, 3, 128, 128, 128), torch.float64),
    ("3D f32 8x3x120^3 (extent 120 = 15*2^3: rocFFT)", (8, 3, 120, 120, 120), torch.float32),
    ("2D f32 64x2x512^2 (rocFFT 2D plan + operator kernel)", (64, 2, 512, 512), torch.float32),
    ("2D f32 16x2x1024^2", (16, 2, 1024, 1024), torch.float32),
    ("2D f64 16x2x1024^2", (16, 2, 1024, 1024), torch.float64),
]
if len(sys.argv) > 1:      # case indices to run (for a per-kernel profile of one shape)
    cases = [cases[int(i)] for i in sys.argv[1:]]
met = lm.FluidMetric([0.1, 0.0, 0.01])
ext = lm.lagomorph_ext
for name, shape, dt in cases:
    m = torch.randn(shape, device=dev, dtype=dt)
    nvox = m.numel()           # voxel-components
    esz = m.element_size()
    ideal = nvox * esz * 6.07  # read m, write+read+write+read the half spectrum, write out (per component)
    for mode, what in ((3, "tuned / generic hand-written passes"), (2, "tuned passes / rocFFT")):
        ext.set_fluid_mode(mode)
        try:
            with torch.no_grad():
                t, _ = time_op(lambda: met.sharp(m), reps=10, warm=5)
            print(f"{name:58s} fluid_mode {mode} ({what:36s}): {t*1e3:8.1f} us  {ideal/t/1e9:5.2f} TB/s of the 6-pass ideal", flush=True)
        except RuntimeError as e:
            print(f"{name:58s} fluid_mode {mode}: {str(e)[:120]}", flush=True)
    ext.set_fluid_mode(3)
    del m
    torch.cuda.empty_cache()
