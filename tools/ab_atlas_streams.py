#!/usr/bin/env python3
"""bench.py's atlas leg (LDDMMAtlasBuilder.iteration at 160^3, image update included) with the matching step on one
stream against its default two-stream split, alternating in one process.  usage: tools/ab_atlas_streams.py <batch> [size]"""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import lagomorph_amd as lm
from lagomorph_amd import lddmm

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 160
args = types.SimpleNamespace(atlas_size=S, atlas_batch=B, atlas_warmup=2, atlas_steps=6)
dev = torch.device("cuda")
default = lddmm.LDDMM_STEP_STREAMS
for parts in (1, 2, 1, 2, 1, 2):
    lddmm.LDDMM_STEP_STREAMS = parts
    r = bench.atlas_leg(lm, dev, 1, 0, args)
    print(f"atlas step {B} x {S}^3, step streams {parts}: {r['ms_per_step']:.2f} ms  ({r['value'] / 1e9:.3f} Gvoxel/s)", flush=True)
lddmm.LDDMM_STEP_STREAMS = default
