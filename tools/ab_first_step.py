#!/usr/bin/env python3
"""The first Euler step of the headline shoot, phi_1 = -dt sharp(m0): the factor inside the operator's last kernel
(`lago_fluid_metric_scaled`, the product since round 5) against a separate multiply pass (rounds 2-4).  Alternating
rounds in one process, one stream, bits compared.  env: S (128), B (32)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur
from lagomorph_amd import lddmm

S, B = int(os.environ.get("S", 128)), int(os.environ.get("B", 32))
dev = torch.device("cuda")
torch.manual_seed(1234)
met = lm.FluidMetric([0.1, 0.0, 0.01])
fused = lddmm._first_step


def separate(metric, m0, dt, v0=None, mommask=None):
    assert v0 is None and mommask is None
    return metric.sharp(m0) * (-dt)


streams = lddmm.EXPMAP_STREAMS
with torch.no_grad():
    m = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev), 4.0)
    m *= 5.0 / met.sharp(m).abs().max()
    ref = None
    for parts in (1, 2):
        lddmm.EXPMAP_STREAMS = parts
        for name, f in (("separate multiply", separate), ("inside the operator", fused)) * 3:
            lddmm._first_step = f
            for _ in range(3):
                lm.expmap(met, m, num_steps=10)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(6):
                out = lm.expmap(met, m, num_steps=10)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 6
            ref = out if ref is None else ref
            print(f"S={S} B={B} streams={parts} {name:20s}: {dt*1e3:7.3f} ms per shoot  bits {'same' if torch.equal(out, ref) else 'DIFFER'}",
                  flush=True)
lddmm._first_step, lddmm.EXPMAP_STREAMS = fused, streams
