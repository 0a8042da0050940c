#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lagomorph_amd as lm
m = torch.randn((32, 3, 128, 128, 128), device="cuda")
met = lm.FluidMetric([0.1, 0.0, 0.01])
with torch.no_grad():
    for _ in range(4):
        v = met.sharp(m)
torch.cuda.synchronize()
