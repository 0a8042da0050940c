#!/usr/bin/env python3
"""A/B of the fluid-metric pass settings in one process: x-pass workgroups of 256 threads or wide (the 256-point tile),
batch items per x-pass workgroup; `lago_set_fluid_zy_persist` switches the persistent zy kernels the same way.  env: S (160), B (8)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import time_op

ext = lm.lagomorph_ext
lib = ext._lib
S, B = int(os.environ.get("S", 160)), int(os.environ.get("B", 8))
shape = tuple(int(x) for x in os.environ["SHAPE"].split("x")) if "SHAPE" in os.environ else (S, S, S)
dev = torch.device("cuda")
m = torch.randn((B, 3) + shape, device=dev)
met = lm.FluidMetric([0.1, 0.0, 0.01])
ref = None
for r in range(2):
    for persist in (1,):
        for ipw in (0, 10, 11, 20):   # 10 / 11: automatic items per workgroup with 256-thread / wide x-pass workgroups; 20: one-shot x-pass workgroups
            ext.tune(fluid_zy_persist=persist)
            ext.tune(fluid_xpass_persist=0 if ipw == 20 else 1)
            ext.tune(fluid_xpass_wide=0 if ipw == 10 else 1)
            ext.tune(fluid_xpass_ipw=0 if ipw >= 10 else ipw)
            out = met.sharp(m)
            if ref is None:
                ref = out
            same = torch.equal(out, ref)
            t, _ = time_op(lambda: met.sharp(m), reps=30, warm=20)
            print(f"{shape} B={B} persist={persist} ipw={ipw}: {t*1e3:7.1f} us  bits {'same' if same else 'DIFFER'}")
ext.tune(fluid_zy_persist=1)
ext.tune(fluid_xpass_ipw=0)
ext.tune(fluid_xpass_wide=1)
ext.tune(fluid_xpass_persist=1)
