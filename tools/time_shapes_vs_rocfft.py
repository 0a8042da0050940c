#!/usr/bin/env python3
"""`sharp` on a list of float32 3D shapes: default path (tuned or generic passes, reported) against rocFFT (`fluid_mode 0`).
usage: time_shapes_vs_rocfft.py B n0,n1,n2 [n0,n1,n2 ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import time_op

ext = lm.lagomorph_ext
g = torch.Generator(device="cuda").manual_seed(3)
met = lm.FluidMetric([0.1, 0.0, 0.01])
B = int(sys.argv[1])
for a in sys.argv[2:]:
    sp = tuple(int(v) for v in a.split(","))
    x = torch.randn((B, 3) + sp, device="cuda", generator=g)
    with torch.no_grad():
        ext.set_fluid_mode(3)
        n0 = ext.path_launches("fluid_lds")
        ref = met.sharp(x)
        path = "tuned" if ext.path_launches("fluid_lds") == n0 + 1 else "generic"
        t3 = time_op(lambda: met.sharp(x), reps=20, warm=5)[0] * 1e3
        ext.set_fluid_mode(0)
        out = met.sharp(x)
        err = float((out.double() - ref.double()).abs().max() / ref.double().abs().max())
        t0 = time_op(lambda: met.sharp(x), reps=20, warm=5)[0] * 1e3
        ext.set_fluid_mode(3)
    vox = B * sp[0] * sp[1] * sp[2]
    print(f"{str(sp):16s} x{B}: {path:7s} {t3:8.1f} us ({vox * 72.8 / t3 / 1e6:.2f} TB/s of ideal)   rocFFT {t0:8.1f} us   ratio {t3 / t0:.2f}   diff {err:.1e}", flush=True)
