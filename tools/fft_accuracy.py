#!/usr/bin/env python3
"""float32 error of FluidMetric.sharp / flat against the same operator in float64 through HIP, on white-noise fields (every
frequency excited), per FFT path and shape, plus timings of the headline shapes -- for ONE library build
(LAGO_HIP_LIBRARY selects an A/B build).  usage: python tools/fft_accuracy.py <tag>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import time_op

tag = sys.argv[1] if len(sys.argv) > 1 else "lib"
ext = lm.lagomorph_ext
met = lm.FluidMetric([0.1, 0.0, 0.01])
g = torch.Generator(device="cuda").manual_seed(7)
shapes = [((128, 128, 128), 8), ((160, 160, 160), 4), ((64, 64, 64), 8), ((96, 80), 8), ((4, 11), 4), ((33, 29, 31), 4),
          ((182, 218, 182), 1), ((100, 120, 60), 2), ((256, 256), 4), ((6, 2053), 2)]
for sp, B in shapes:
    x = torch.randn((B, len(sp)) + sp, device="cuda", generator=g)
    with torch.no_grad():
        before = ext.path_launches()
        errs = []
        for op in (met.sharp, met.flat):
            t = op(x.double())
            errs.append(float((op(x).double() - t).abs().max() / t.abs().max()))
        after = ext.path_launches()
    path = [k for k in after if after[k] != before[k] and k.startswith("fluid")]
    print(f"{tag} {str(sp):18s} x{B}: sharp {errs[0]:.3e}  flat {errs[1]:.3e}   {path}", flush=True)
for sp, B in (((128, 128, 128), 32), ((160, 160, 160), 8)):
    x = torch.randn((B, 3) + sp, device="cuda", generator=g)
    with torch.no_grad():
        t, _ = time_op(lambda: met.sharp(x), reps=20, warm=10)
    print(f"{tag} sharp {sp} x{B}: {t*1e3:.1f} us", flush=True)
