#!/usr/bin/env python3
"""affine_interp_forward / regrid_forward at bench.py's `other_ops` shapes for ONE library build, one line.
usage: LAGO_HIP_LIBRARY=... python tools/ab_forward.py <tag>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import lagomorph_amd as lm

tag = sys.argv[1] if len(sys.argv) > 1 else "lib"
ext = lm.lagomorph_ext
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(99)
size, batch = 128, 8
I1 = torch.randn((batch, 1, size, size, size), device=dev, generator=g)
I3 = torch.randn((batch, 3, size, size, size), device=dev, generator=g)
A = (torch.eye(3, device=dev)[None] + 0.05 * torch.randn((batch, 3, 3), device=dev, generator=g)).contiguous()
T = torch.randn((batch, 3), device=dev, generator=g)
c, s = 0.9396926, 0.3420201
R = torch.tensor([[1, 0, 0], [0, c, -s], [0, s, c]], device=dev).repeat(batch, 1, 1).contiguous()
small = torch.randn((batch, 3, 64, 64, 64), device=dev, generator=g)
small80 = torch.randn((batch, 3, 80, 80, 80), device=dev, generator=g)
ops = {
    "affine C=1": lambda: ext.affine_interp_forward(I1, A, T),
    "affine C=3": lambda: ext.affine_interp_forward(I3, A, T),
    "affine rot20": lambda: ext.affine_interp_forward(I1, R, T),
    "regrid 64->128": lambda: ext.regrid_forward(small, [128] * 3, [31.5] * 3, [63 / 127] * 3),
    "regrid 80->160": lambda: ext.regrid_forward(small80, [160] * 3, [39.5] * 3, [79 / 159] * 3),
    "regrid 128->64": lambda: ext.regrid_forward(I3, [64] * 3, [63.5] * 3, [127 / 63] * 3),
}
out = []
for name, fn in ops.items():
    med, _ = bench.time_op(fn, reps=30, warm=20)
    out.append(f"{name} {med * 1e3:.1f}")
print(f"{tag:>6s}: " + "  ".join(out), flush=True)
