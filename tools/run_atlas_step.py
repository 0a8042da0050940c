#!/usr/bin/env python3
"""Profiling aid: one atlas matching step (lddmm_step: expmap 5 steps -> interp -> loss -> backward -> update)
at batch B x S^3 (or B x S x S2 x S3), timed with HIP events; run under rocprofv3 --kernel-trace --stats for the kernel split.
usage: run_atlas_step.py B S [S2 S3]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lagomorph_amd as lm
from bench import gaussian_blur

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
sp = (S, int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (S, S, S)
dev = torch.device("cuda")
torch.manual_seed(0)
metric = lm.FluidMetric([0.1, 0.0, 0.01])
template = gaussian_blur(torch.randn((1, 1) + sp, device=dev), 3.0)
template = template / template.std()
I = template.clone().requires_grad_(True)
img = gaussian_blur(torch.randn((B, 1) + sp, device=dev), 3.0)
img = img / img.std()
# momenta that shoot to ~3 voxels of displacement; learning rate 0 keeps that state for every timed step
with torch.no_grad():
    m = gaussian_blur(torch.randn((B, 3) + sp, device=dev), 4.0)
    m *= 3.0 / metric.sharp(m).abs().max()
for it in range(3):
    m, loss, reg = lm.lddmm_step(I, m, img, metric, dataset_size=B, integration_steps=5, learning_rate_pose=0.0)
torch.cuda.synchronize()
ts = []
for it in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    m, loss, reg = lm.lddmm_step(I, m, img, metric, dataset_size=B, integration_steps=5, learning_rate_pose=0.0)
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ts.sort()
med = ts[len(ts) // 2]
vox = B * sp[0] * sp[1] * sp[2]
print(f"lddmm_step batch {B} x {sp}: median {med:.2f} ms  -> {vox / med / 1e6:.2f} Gvoxel/s per step; loss {float(loss):.4f}")
