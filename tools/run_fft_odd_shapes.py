"""Generic FFT passes (fluid_mode 3) against rocFFT (fluid_mode 2) on shapes with large prime factors (MNI brain volumes
182 x 218 x 182 and friends).  usage: python tools/run_fft_odd_shapes.py [case index ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lagomorph_amd as lm
from bench import time_op
ext = lm.lagomorph_ext
met = lm.FluidMetric([0.1, 0.0, 0.01])
dev = torch.device("cuda")
cases = (((182, 218, 182), 4), ((91, 109, 91), 8), ((176, 208, 176), 4), ((193, 229, 193), 2), ((256, 256), 32), ((181, 217), 32))
if len(sys.argv) > 1:
    cases = [cases[int(i)] for i in sys.argv[1:]]
for shape, B in cases:
    m = torch.randn((B, len(shape)) + shape, device=dev)
    line = f"{shape} B={B}:"
    for mode in (3, 2):
        ext.set_fluid_mode(mode)
        try:
            with torch.no_grad():
                t, _ = time_op(lambda: met.sharp(m), reps=10, warm=3)
            line += f"  mode {mode}: {t*1e3:8.1f} us"
        except RuntimeError as e:
            line += f"  mode {mode}: {str(e)[:60]}"
    ext.set_fluid_mode(3)
    print(line, flush=True)
    del m
