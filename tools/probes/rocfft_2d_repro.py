#!/usr/bin/env python3
"""Pure-torch check of rocFFT's batched 2D real transforms in the order tests/test_gpu_parity.py::test_fused_2d_fluid_metric
visits the shapes: rfftn / irfftn on the GPU against the CPU (pocketfft), batch 10.  (Found while testing the fused 2D
fluid-metric kernel: after plans for other shapes exist, the (32, 128) transform came back wrong through BOTH hipFFT
plans of this library and torch.fft -- the hand-written passes were right.)"""
import torch

torch.manual_seed(0)
for sp in ((64, 64), (128, 128), (96, 64), (64, 128), (160, 96), (32, 128), (256, 64), (32, 128)):
    x = torch.randn((10,) + sp)
    Fc = torch.fft.rfftn(x, dim=(-2, -1), norm="ortho")
    Fg = torch.fft.rfftn(x.cuda(), dim=(-2, -1), norm="ortho").cpu()
    e1 = float((Fg - Fc).abs().max() / Fc.abs().max())
    yc = torch.fft.irfftn(Fc, s=sp, dim=(-2, -1), norm="ortho")
    yg = torch.fft.irfftn(Fc.cuda(), s=sp, dim=(-2, -1), norm="ortho").cpu()
    e2 = float((yg - yc).abs().max() / yc.abs().max())
    print(f"{sp}: rfftn rel err {e1:.2e}   irfftn rel err {e2:.2e}", flush=True)
