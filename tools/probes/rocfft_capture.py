"""Probe (round 4): can a rocFFT-backed lago_fluid_metric call be captured into a HIP graph (a) with a plan the spot check
has verified, (b) with a plan that is still unverified (the check must be skipped while the stream is capturing)?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import lagomorph_amd as lm

ext = lm.lagomorph_ext
ext.set_fluid_mode(0)
met = lm.FluidMetric([0.1, 0.05, 0.01])


def capture(m, what):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side, capture_error_mode="relaxed"):
                out = met.sharp(m)
        torch.cuda.current_stream().wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
        print(what, ": captured and replayed; plan state", ext.fft_plan_state())
        return out
    except Exception as e:
        print(what, ": FAILED:", str(e).splitlines()[0][:160])
        return None


m = torch.randn((7, 2, 44, 52), device="cuda")
ref = met.sharp(m)   # creates + verifies the plan eagerly
print("plan state", ext.fft_plan_state())
out = capture(m, "(a) verified plan")
print("    equal to the eager result:", out is not None and torch.equal(out, ref))
z = torch.zeros((5, 2, 44, 52), device="cuda")
met.sharp(z)         # a new plan (another batch size), left unverified by the all-zero field
print("plan state", ext.fft_plan_state())
m5 = torch.randn((5, 2, 44, 52), device="cuda")
out = capture(m5, "(b) unverified plan")
