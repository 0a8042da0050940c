"""A/B of the LDS-window gathers (lago_set_gather_window 0 / 1) on compose and Ad_star at the headline sizes.
Usage: python tools/ab_gather_window.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import lagomorph_amd as lm
import lagomorph_amd.lagomorph_ext as ext


def smooth(nn, sp, amp, shift, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    grids = torch.meshgrid(*[torch.arange(n, device="cuda", dtype=torch.float32) for n in sp], indexing="ij")
    u = torch.empty((nn, 3) + tuple(sp), device="cuda")
    for n in range(nn):
        for c in range(3):
            k = (torch.rand(3, generator=g, device="cuda") * 0.1 + 0.02).tolist()
            ph = float(torch.rand(1, generator=g, device="cuda")) * 6
            u[n, c] = amp * torch.sin(k[0] * grids[0] + k[1] * grids[1] - k[2] * grids[2] + ph) + shift * (c - 1)
    return u


def timeit(f, reps):
    for _ in range(3):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for nn, sp in ((32, (128, 128, 128)), (8, (160, 160, 160))):
        v = torch.randn((nn, 3) + sp, device="cuda")
        for amp, shift in ((0.4, 0.0), (1.5, 0.0), (1.5, 6.3), (4.0, 0.0), (10.0, 0.0)):
            u = smooth(nn, sp, amp, shift, 1)
            row = []
            outs = []
            for mode in (0, 1):
                ext.set_gather_window(mode)
                row.append(timeit(lambda: ext.compose(u, v, 1.0, -0.1), reps))
                outs.append(ext.compose(u, v, 1.0, -0.1))
            same = all(torch.equal(outs[0], o) for o in outs[1:])
            row2 = []
            for mode in (0, 1):
                ext.set_gather_window(mode)
                row2.append(timeit(lambda: lm.Ad_star(u, v), reps))
            print(f"{nn}x3x{sp[0]}^3 amp {amp:4.1f} shift {shift:3.1f}: compose pair {row[0]:7.1f} us  window {row[1]:7.1f}  "
                  f"same bits {same} | Ad_star pair {row2[0]:7.1f}  window {row2[1]:7.1f}", flush=True)
    # the headline workload's own fields: displacement after k Euler steps of bench.py's shoot, velocity of the next step
    from bench import gaussian_blur

    metric = lm.FluidMetric([0.1, 0.0, 0.01])
    torch.manual_seed(1234)
    with torch.no_grad():
        m = gaussian_blur(torch.randn((32, 3, 128, 128, 128), device="cuda"), 4.0)
        m *= 5.0 / metric.sharp(m).abs().max()
        for ksteps in (2, 5, 9):
            h = lm.expmap(metric, m, num_steps=10)  # dt = 1/10 per step
            h = lm.expmap(metric, m * (ksteps / 10.0), num_steps=ksteps)  # the same flow stopped after k of 10 steps
            v = metric.sharp(lm.Ad_star(h, m))
            row = []
            for mode in (0, 1):
                ext.set_gather_window(mode)
                row.append(timeit(lambda: ext.compose(h, v, 1.0, -0.1), reps))
            gx = (h[:, :, 1:] - h[:, :, :-1]).abs().max().item()
            print(f"bench fields after {ksteps}/10 steps: |h|max {h.abs().max().item():.2f}, max |dh/dx| {gx:.2f}: compose pair "
                  f"{row[0]:7.1f} us  window {row[1]:7.1f}", flush=True)
    ext.set_gather_window(1)


if __name__ == "__main__":
    main()
