#!/usr/bin/env python3
"""Write bench.py's smooth displacement field of configs[1] (8 x 3 x 128^3 float32) to a raw file for the probes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import gaussian_blur

dev = torch.device("cuda")
S, B = 128, int(sys.argv[2]) if len(sys.argv) > 2 else 8
g = torch.Generator(device=dev).manual_seed(1234)
I = gaussian_blur(torch.randn((B, 1, S, S, S), device=dev, generator=g), 2.0)
u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0)
u = u * (4.0 / u.abs().max())
u.cpu().numpy().tofile(sys.argv[1])
