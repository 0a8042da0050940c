#!/usr/bin/env python3
"""Does capturing the shoot in a HIP graph (torch.cuda.CUDAGraph over the C-ABI launches) recover the launch tails at
small per-GPU batches?  expmap, 10 Euler steps, 128^3, eager vs graph replay, bit compare.  env: BATCHES (4,8,32)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur

dev = torch.device("cuda")
S = int(os.environ.get("S", 128))
metric = lm.FluidMetric([0.1, 0.0, 0.01])
for B in [int(x) for x in os.environ.get("BATCHES", "4,8,32").split(",")]:
    with torch.no_grad():
        m = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev), 4.0)
        m *= 5.0 / metric.sharp(m).abs().max()
        for _ in range(3):
            ref = lm.expmap(metric, m, num_steps=10)
        torch.cuda.synchronize()

        def timed(fn, reps=10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3

        te = timed(lambda: lm.expmap(metric, m, num_steps=10))
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                lm.expmap(metric, m, num_steps=10)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            out = lm.expmap(metric, m, num_steps=10)
        g.replay()
        torch.cuda.synchronize()
        same = torch.equal(out, ref)
        tg = timed(g.replay)
        V = B * S ** 3 * 10
        print(f"B={B}: eager {te:.3f} ms ({V/te/1e6:.2f} Gvox-step/s)  graph {tg:.3f} ms ({V/tg/1e6:.2f})  "
              f"gain {100*(te/tg-1):.1f} %  bits {'same' if same else 'DIFFER'}", flush=True)
        del g, out, ref, m
        torch.cuda.empty_cache()
