#!/usr/bin/env python3
"""Profiling aid: time FluidMetric.sharp (batch 32 x 3x128^3) with the x-pass kernel's debug variants
(1 = load/store only, 2 = FFTs but no operator, 3 = operator with a constant table row)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lagomorph_amd as lm
from lagomorph_amd import lagomorph_ext as ext

lib = ext._lib
assert hasattr(lib, "lago_debug_xpass_variant"), (
    "needs the profiling build: python -m lagomorph_amd.build --profiling; "
    "LAGO_HIP_LIBRARY=lagomorph_amd/_lib/liblagomorph_hip_prof.so python " + sys.argv[0])
m = torch.randn((32, 3, 128, 128, 128), device="cuda")
met = lm.FluidMetric([0.1, 0.0, 0.01])
for var in (0, 1, 2, 3, 0):
    lib.lago_debug_xpass_variant(var)
    with torch.no_grad():
        for _ in range(3):
            met.sharp(m)
        ts = []
        for _ in range(8):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); met.sharp(m); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
    ts.sort()
    print(f"variant {var}: sharp median {ts[len(ts)//2]:.3f} ms", flush=True)
lib.lago_debug_xpass_variant(0)
