#!/usr/bin/env python3
"""The headline shoot (expmap, 10 Euler steps, batch B x 3 x S^3) on ONE stream against the product's sub-batch split
over 2 / 3 / 4 HIP streams (`lddmm.EXPMAP_STREAMS`; default 2 since round 5): alternating rounds in one process, bits
compared.  env: S (128), B (32)."""
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
default = lddmm.EXPMAP_STREAMS
lddmm.EXPMAP_MIN_ITEMS = int(os.environ.get("MINPER", lddmm.EXPMAP_MIN_ITEMS))
with torch.no_grad():
    m = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev), 4.0)
    m *= 5.0 / met.sharp(m).abs().max()
    lddmm.EXPMAP_STREAMS = 1
    ref = lm.expmap(met, m, num_steps=10)
    for parts in (1, 2, 3, 4, 1, 2, 3, 4):
        if B < lddmm.EXPMAP_MIN_ITEMS * parts:
            continue
        lddmm.EXPMAP_STREAMS = parts
        for _ in range(4):
            lm.expmap(met, m, num_steps=10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = lm.expmap(met, m, num_steps=10)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"S={S} B={B} streams={parts}: {dt*1e3:7.2f} ms per shoot  ({B * S**3 * 10 / dt / 1e9:.2f} Gvoxel-steps/s)  "
              f"bits {'same' if torch.equal(out, ref) else 'DIFFER'}", flush=True)
lddmm.EXPMAP_STREAMS = default
