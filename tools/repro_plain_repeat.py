"""Repeat the plain (no process group) atlas builder run of tests/test_gpu_rccl_world1.py many times and print, per
quantity, how far each repeat is from the first -- to tell the float-atomics rounding spread (1e-7) from anything larger.
    python tools/repro_plain_repeat.py [repeats] [lr_image_factor] [debug]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_rccl_world1 as T  # noqa: E402

import lagomorph_amd as lm  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
fac = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
if len(sys.argv) > 3 and sys.argv[3] == "debug":
    lm.set_debug_mode(True)
rel = lambda x, y: float((x.double() - y.double()).abs().max() / y.double().abs().max())
hist = lambda x, y: float(np.abs(np.asarray(x) - np.asarray(y)).max() / np.abs(np.asarray(y)).max())
for freq in (0, 2):
    data = T._dataset(12, 64, torch.float32, seed=9)
    ref = T._run(data, image_update_freq=freq, lr_image_factor=fac)
    for r in range(reps):
        b = T._run(data, image_update_freq=freq, lr_image_factor=fac)
        print(f"freq {freq} rep {r}: atlas {rel(b.I.detach(), ref.I.detach()):.2e} momenta "
              f"{max(rel(x, y) for x, y in zip(b.ms, ref.ms)):.2e} iter_loss {hist(b.iter_losses, ref.iter_losses):.2e} "
              f"iter_reg {hist(b.iter_reg_terms, ref.iter_reg_terms):.2e}  losses {[f'{x:.6f}' for x in b.iter_losses]}", flush=True)
