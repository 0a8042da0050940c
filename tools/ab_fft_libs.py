#!/usr/bin/env python3
"""Fluid-metric passes of ONE library build (LAGO_HIP_LIBRARY picks it; tools/ab_fft_libs.sh alternates builds on one
box): sharp / flat timings at the shapes of the two benchmark workloads and the deviation from the float64 path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import time_op

tag = sys.argv[1] if len(sys.argv) > 1 else "lib"
dev = torch.device("cuda")
met = lm.FluidMetric([0.1, 0.0, 0.01])
cases = (((128, 128, 128), 32), ((160, 160, 160), 8), ((128, 128, 128), 8), ((192, 160, 96), 4), ((96, 160, 192), 4), ((96, 128, 160), 4))
if "ONLY" in os.environ:   # e.g. ONLY=0,1 under the profiler (kernel names do not tell batch sizes apart)
    cases = [cases[int(i)] for i in os.environ["ONLY"].split(",")]
for shape, B in cases:
    torch.manual_seed(1)
    m = torch.randn((B, 3) + shape, device=dev)
    with torch.no_grad():
        out = met.sharp(m)
        ref = met.sharp(m[:1].double())
        err = float((out[:1].double() - ref).abs().max() / ref.abs().max())
        out = met.flat(m)
        ref = met.flat(m[:1].double())
        errf = float((out[:1].double() - ref).abs().max() / ref.abs().max())
        ts, _ = time_op(lambda: met.sharp(m), reps=30, warm=20)
        tf, _ = time_op(lambda: met.flat(m), reps=30, warm=20)
    print(f"{tag:>8s} {shape} B={B}: sharp {ts*1e3:7.1f} us  flat {tf*1e3:7.1f} us   max dev from float64: {err:.2e} / {errf:.2e}", flush=True)
    del m, out, ref
    torch.cuda.empty_cache()
