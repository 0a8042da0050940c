#!/usr/bin/env python3
"""`sharp` through the generic FFT passes on one shape, 10 calls: for rocprofv3 --kernel-trace --stats.
usage: run_generic_shape.py B dtype n0 n1 [n2]   (e.g. 2 float32 182 218 182)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm

B, dt = int(sys.argv[1]), getattr(torch, sys.argv[2])
sp = tuple(int(a) for a in sys.argv[3:])
x = torch.randn((B, len(sp)) + sp, device="cuda", dtype=dt)
met = lm.FluidMetric([0.1, 0.0, 0.01])
with torch.no_grad():
    for _ in range(10):
        met.sharp(x)
torch.cuda.synchronize()
