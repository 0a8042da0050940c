#!/usr/bin/env python3
"""Experiment: the headline shoot (expmap, 10 Euler steps, batch B x 3 x S^3) as ONE stream over the whole batch
against the batch split over 2 / 4 HIP streams -- the gather kernels (vector-memory bound) of one part can then run
beside the FFT passes (HBM bound) of another.  env: S (128), B (32)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur

S, B = int(os.environ.get("S", 128)), int(os.environ.get("B", 32))
dev = torch.device("cuda")
torch.manual_seed(1234)
met = lm.FluidMetric([0.1, 0.0, 0.01])
with torch.no_grad():
    m = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev), 4.0)
    m *= 5.0 / met.sharp(m).abs().max()
    ref = lm.expmap(met, m, num_steps=10)

    POOL = [torch.cuda.Stream() for _ in range(4)]   # made once: every stream has its own allocator pool to warm up

    def run(parts):
        if parts == 1:
            return [lm.expmap(met, m, num_steps=10)]
        streams = POOL[:parts]
        cur = torch.cuda.current_stream()
        outs = []
        n = B // parts
        for i, s in enumerate(streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs.append(lm.expmap(met, m[i * n:(i + 1) * n], num_steps=10))
        for s in streams:
            cur.wait_stream(s)
        return outs

    for parts in (1, 2, 4, 1, 2, 4):
        for _ in range(4):
            run(parts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            outs = run(parts)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        same = torch.equal(torch.cat(outs), ref)
        print(f"S={S} B={B} parts={parts}: {dt*1e3:7.2f} ms per shoot  ({B * S**3 * 10 / dt / 1e9:.2f} Gvoxel-steps/s)  bits {'same' if same else 'DIFFER'}")
