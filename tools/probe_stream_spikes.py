#!/usr/bin/env python3
"""Per-call times of the forward shoot while lddmm.EXPMAP_STREAMS changes (are there one-off stalls when the stream
count changes?).  env: S, B, SEQ (e.g. "1,2,1,2,1,2")"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur
from lagomorph_amd import lddmm

S, B = int(os.environ.get("S", 128)), int(os.environ.get("B", 8))
seq = [int(x) for x in os.environ.get("SEQ", "1,2,3,4,1,2,3,4").split(",")]
dev = torch.device("cuda")
torch.manual_seed(1234)
met = lm.FluidMetric([0.1, 0.0, 0.01])
with torch.no_grad():
    m = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev), 4.0)
    m *= 5.0 / met.sharp(m).abs().max()
    for parts in seq:
        lddmm.EXPMAP_STREAMS = parts
        ts = []
        for _ in range(9):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lm.expmap(met, m, num_steps=10)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        st = torch.cuda.memory_stats()
        print(f"S={S} B={B} streams={parts}: " + " ".join(f"{t:6.2f}" for t in ts) +
              f"   | reserved {st['reserved_bytes.all.current'] / 2**20:.0f} MiB, cudaMalloc calls {st['num_device_alloc']}, frees {st['num_device_free']}", flush=True)
