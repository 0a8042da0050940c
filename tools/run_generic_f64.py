#!/usr/bin/env python3
"""sharp at 8 x 3 x 128^3 float64 (the generic FFT passes), 10 calls: for rocprofv3 --kernel-trace --stats."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm

S = int(os.environ.get("S", 128))
x = torch.randn((8, 3, S, S, S), device="cuda", dtype=torch.float64)
met = lm.FluidMetric([0.1, 0.0, 0.01])
with torch.no_grad():
    for _ in range(10):
        met.sharp(x)
torch.cuda.synchronize()
