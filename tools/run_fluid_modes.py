#!/usr/bin/env python3
"""Profiling aid: FluidMetric.sharp/flat (batch 32 x 3x128^3 fp32) in the three implementation modes,
and the native mode's passes one at a time (stage mask; results are garbage for masks != 7)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lagomorph_amd as lm
from lagomorph_amd import lagomorph_ext as ext

lib = ext._lib
assert hasattr(lib, "lago_debug_xpass_variant"), (
    "needs the profiling build: python -m lagomorph_amd.build --profiling; "
    "LAGO_HIP_LIBRARY=lagomorph_amd/_lib/liblagomorph_hip_prof.so python " + sys.argv[0])
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
m = torch.randn((batch, 3, size, size, size), device="cuda")
met = lm.FluidMetric([0.1, 0.0, 0.01])


def timeit(f, reps=10):
    for _ in range(3):
        f()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


with torch.no_grad():
    for mode in (0, 1, 2):
        ext.set_fluid_mode(mode)
        print(f"mode {mode}: sharp {timeit(lambda: met.sharp(m)):.3f} ms  flat {timeit(lambda: met.flat(m)):.3f} ms", flush=True)
    ext.set_fluid_mode(3)
    for mask in (1, 2, 4, 7):
        lib.lago_debug_fluid_stage_mask(mask)
        print(f"native stage mask {mask}: sharp {timeit(lambda: met.sharp(m)):.3f} ms", flush=True)
    lib.lago_debug_fluid_stage_mask(7)
    for ipw in (1, 2, 4, 8):
        ext.tune(fluid_xpass_ipw=ipw)
        lib.lago_debug_fluid_stage_mask(2)
        a = timeit(lambda: met.sharp(m))
        lib.lago_debug_fluid_stage_mask(7)
        print(f"x pass, {ipw} batch items per workgroup: x pass alone {a:.3f} ms, sharp {timeit(lambda: met.sharp(m)):.3f} ms", flush=True)
    ext.tune(fluid_xpass_ipw=2)
