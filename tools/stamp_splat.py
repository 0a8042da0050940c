#!/usr/bin/env python3
"""Per-wave shader-clock stamps of splat_shear_kernel (C = 1, 8 x 128^3; PROFILING library only):
0 start, 1 placement done, 2 window zeroed, 3 after barrier, 4 voxel loop done, 5 after barrier, 6 flush issued, 7 after barrier.
    LAGO_HIP_LIBRARY=$PWD/lagomorph_amd/_lib/liblagomorph_hip_prof.so python tools/stamp_splat.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import lagomorph_amd as lm
from bench import gaussian_blur

ext = lm.lagomorph_ext
lib = ext._lib
assert hasattr(lib, "lago_debug_splat_stamps"), "needs the profiling build"
lib.lago_debug_splat_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
S, B = int(os.environ.get("S", 128)), int(os.environ.get("B", 8))
g = torch.Generator(device=dev).manual_seed(1234)
I = gaussian_blur(torch.randn((B, 1, S, S, S), device=dev, generator=g), 2.0)
u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0)
u = u * (4.0 / u.abs().max())
go = torch.randn((B, 1, S, S, S), device=dev, generator=g)
ext.set_splat_shear_mc(2)
for _ in range(10):
    ext.interp_backward(go, I, u, 1.0, True, True)
nwg = 8192
st = torch.zeros((nwg, 16, 8), dtype=torch.int64, device=dev)
lib.lago_debug_splat_stamps(ctypes.c_void_p(st.data_ptr()))
ext.interp_backward(go, I, u, 1.0, True, True)
torch.cuda.synchronize()
lib.lago_debug_splat_stamps(ctypes.c_void_p(0))
t = st.cpu().numpy().astype(np.float64)
used = t[:, 0, 0] > 0
t = t[used]
print(f"{t.shape[0]} workgroups stamped; kernel span {(t[:, :, 7].max() - t[:, :, 0].min()) / 1e2:.1f} us at 100 MHz ticks"
      if False else f"{t.shape[0]} workgroups stamped")
d = np.diff(t, axis=2)   # per wave phase durations in shader clocks
names = ["decode+placement", "zero window", "barrier 1 wait", "voxel loop", "barrier 2 wait", "flush", "barrier 3 wait"]
tot = t[:, :, 7] - t[:, :, 0]
print(f"workgroup lifetime: mean {tot.mean():.0f} clk, median {np.median(tot):.0f}, p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f}")
for i, n in enumerate(names):
    print(f"  {n:18s}: mean {d[:, :, i].mean():8.0f} clk ({100 * d[:, :, i].mean() / tot.mean():5.1f} %)   wave-min {d[:, :, i].min(axis=1).mean():8.0f}  wave-max {d[:, :, i].max(axis=1).mean():8.0f}")
span = t[:, :, 7].max() - t[:, :, 0].min()
print(f"kernel span {span:.0f} clk; sum of workgroup lifetimes / (512 slots) = {tot.mean() * t.shape[0] / 512:.0f} clk")
