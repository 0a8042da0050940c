"""compose at 32 x 3 x 128^3 with a smooth displacement, three launches per gather-window mode (for rocprofv3 passes).
Usage: python tools/run_compose.py [amp] [modes, e.g. 01]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lagomorph_amd.lagomorph_ext as ext  # noqa: E402
from ab_gather_window import smooth  # noqa: E402

amp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.4
modes = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else "01")]
sp = (128, 128, 128)
u = smooth(32, sp, amp, 0.0, 1)
v = torch.randn((32, 3) + sp, device="cuda")
for mode in modes:
    ext.set_gather_window(mode)
    for _ in range(3):
        ext.compose(u, v, 1.0, -0.1)
torch.cuda.synchronize()
