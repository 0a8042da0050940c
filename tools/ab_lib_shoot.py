#!/usr/bin/env python3
"""One library BUILD (LAGO_HIP_LIBRARY picks it; alternate builds on one box with a shell loop): the headline shoot and its
three operators at 32 x 3 x 128^3 on the shoot's own fields.  usage: LAGO_HIP_LIBRARY=... python tools/ab_lib_shoot.py <tag>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import time_op, gaussian_blur

tag = sys.argv[1] if len(sys.argv) > 1 else "lib"
B, S, E = 32, 128, 10
dev = torch.device("cuda")
met = lm.FluidMetric([0.1, 0.0, 0.01])
torch.manual_seed(1234)
with torch.no_grad():
    m = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev), 4.0)
    m *= 5.0 / met.sharp(m).abs().max()   # bench.py's headline momenta: max |expmap| ~ 5 voxels
    h = lm.expmap(met, m, num_steps=E)
    v = met.sharp(m)
    ext = lm.lagomorph_ext
    t_shoot, _ = time_op(lambda: lm.expmap(met, m, num_steps=E), reps=6, warm=3)
    t_comp, _ = time_op(lambda: ext.compose(h, v, 1.0, -0.1), reps=30, warm=20)
    t_ad, _ = time_op(lambda: lm.adjrep.Ad_star(h, m), reps=30, warm=20)
    t_sharp, _ = time_op(lambda: met.sharp(m), reps=30, warm=20)
print(f"{tag:>10s}: shoot {t_shoot:7.3f} ms   compose {t_comp*1e3:6.1f} us   Ad_star {t_ad*1e3:6.1f} us   sharp {t_sharp*1e3:6.1f} us", flush=True)
