#!/usr/bin/env python3
"""Marching-window splat (lago_tuning.splat_shear_mc = 3, splat_march3_kernel) against the shipped geometry-once kernel
(mode 2) in one process.  NEEDS the library with `profiles/r06_march_window.patch` applied (`git apply` it, rebuild): the
kernel is not in the tree (measured, not faster: profiles/r06_march_window.md); on the shipped library mode 3 is mode 2.
What it does: parity on a sweep of shapes / steps / start modes / field roughness (d_u bits, d_I relative),
then timings at 8 x 128^3, 8 x 160^3 and 32 x 160^3.   env: QUICK=1 parity only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur, time_op

ext = lm.lagomorph_ext
dev = torch.device("cuda")


def fields(B, sp, amp, sigma, seed, bc=False):
    g = torch.Generator(device=dev).manual_seed(seed)
    sh = (B, 3) + sp
    if sigma > 0:
        u = gaussian_blur(torch.randn(sh, device=dev, generator=g), sigma)
        u = u * (amp / u.abs().max())
    else:
        u = amp * torch.randn(sh, device=dev, generator=g)
    I = gaussian_blur(torch.randn(((1 if bc else B), 3) + sp, device=dev, generator=g), 2.0)
    go = torch.randn(sh, device=dev, generator=g)
    return go, I, u


def run(mode, fn):
    ext.set_splat_shear_mc(mode)
    before = ext.path_launches() if hasattr(ext, "path_launches") else None
    out = fn()
    torch.cuda.synchronize()
    after = ext.path_launches() if hasattr(ext, "path_launches") else None
    ext.set_splat_shear_mc(2)
    took = None
    if before is not None:
        took = [k for k in after if after[k] != before[k]]
    return out, took


bad = 0
cases = []
for sp in ((128, 128, 128), (40, 36, 50), (33, 47, 16), (24, 20, 100), (17, 9, 160), (64, 64, 64), (9, 5, 32), (70, 3, 40)):
    for amp, sigma in ((4.0, 8.0), (1.0, 0.0), (0.3, 0.0), (30.0, 6.0)):
        for dt in (1.0, -1.0, -0.2):
            cases.append((sp, amp, sigma, dt))
for idx, (sp, amp, sigma, dt) in enumerate(cases):
    B = 2 if sp[0] * sp[1] * sp[2] > 500000 else 3
    go, I, u = fields(B, sp, amp, min(sigma, min(sp) / 4.0), 100 + idx)
    du0 = torch.randn_like(u)
    dI0 = torch.randn_like(I)
    forms = {
        "plain": lambda: ext.interp_backward(go, I, u, dt, True, True),
        "addgo": lambda: ext.interp_backward_fused(go, I, u, dt, True, addgo=dt),
        "acc": lambda: ext.interp_backward_fused(go, I, u, dt, True, d_u=du0.clone(), d_I=dI0.clone()),
    }
    for name, fn in forms.items():
        (rI, ru), _ = run(2, fn)
        (mI, mu), took = run(3, fn)
        scale = float(rI.abs().max())
        e = float((mI - rI).abs().max()) / scale
        same = torch.equal(mu, ru)
        ok = same and e <= 1e-5
        bad += 0 if ok else 1
        if not ok or idx % 12 == 0:
            nd = int((mu != ru).sum())
            print(f"{'ok ' if ok else 'BAD'} {sp} amp {amp} sigma {sigma} dt {dt} {name}: d_u {'same bits' if same else f'DIFFERS in {nd}'}, "
                  f"d_I rel {e:.1e}  path {took}", flush=True)
# broadcast image (BC)
go, I, u = fields(3, (40, 36, 50), 3.0, 6.0, 7, bc=True)
for dt in (1.0, -0.3):
    (rI, ru), _ = run(2, lambda: ext.interp_backward(go, I, u, dt, True, True))
    (mI, mu), took = run(3, lambda: ext.interp_backward(go, I, u, dt, True, True))
    e = float((mI - rI).abs().max() / rI.abs().max())
    ok = torch.equal(mu, ru) and e <= 1e-5
    bad += 0 if ok else 1
    print(f"{'ok ' if ok else 'BAD'} broadcast I dt {dt}: d_u same {torch.equal(mu, ru)} d_I rel {e:.1e} path {took} shapes {tuple(mI.shape)}", flush=True)
print("PARITY", "FAILED" if bad else "ok", f"({bad} bad of {3 * len(cases) + 2})", flush=True)
if os.environ.get("QUICK") == "1" or bad:
    sys.exit(1 if bad else 0)

for B, S in ((8, 128), (8, 160), (32, 160), (4, 128), (4, 160)):
    sp = (S, S, S)
    go, I, u = fields(B, sp, 4.0, 8.0, 1234)
    V = B * S ** 3
    du0, dI0 = torch.randn_like(u), torch.randn_like(I)
    forms = {
        "interp_backward dt=1": lambda: ext.interp_backward(go, I, u, 1.0, True, True),
        "interp_backward dt=-0.2": lambda: ext.interp_backward(go, I, u, -0.2, True, True),
        "fused addgo dt=-0.2": lambda: ext.interp_backward_fused(go, I, u, -0.2, True, addgo=-0.2),
        "fused acc dt=1": lambda: ext.interp_backward_fused(go, I, u, 1.0, True, d_u=du0, d_I=dI0),
    }
    for rep in range(2):
        for mode in (2, 3):
            ext.set_splat_shear_mc(mode)
            for name, fn in forms.items():
                t, _ = time_op(fn, reps=20, warm=10)
                print(f"{S}^3 B={B} mode={mode} {name:24s}: {t*1e3:8.1f} us  {60.0*V/t/1e9:5.2f} TB/s alg  frac {60.0*V/t/1e9/8.0:.3f}", flush=True)
    del go, I, u, du0, dI0
    torch.cuda.empty_cache()
ext.set_splat_shear_mc(2)
