"""GPU: the C-ABI library called from several host threads at once, each on a HIP stream of its own.  autograd runs
backward passes on threads of its own, ctypes releases the GIL for the duration of a call, and the library keeps
process-wide state (the fluid metric's coefficient-table cache, the launch-direction counter, path telemetry, tuning
knobs), so entry points do overlap in practice.  Every thread's results must be the ones a single thread gets."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _work(lm, seed, sp, rounds, first_shape_only=False):
    ext = lm.lagomorph_ext
    g = torch.Generator(device="cuda").manual_seed(seed)
    u = 1.5 * torch.randn((2, 3) + sp, device="cuda", generator=g)
    u = torch.nn.functional.avg_pool3d(u, 3, stride=1, padding=1).contiguous()
    v = torch.randn((2, 3) + sp, device="cuda", generator=g)
    I = torch.randn((2, 1) + sp, device="cuda", generator=g)
    go = torch.randn((2, 1) + sp, device="cuda", generator=g)
    met = lm.FluidMetric([0.1, 0.0, 0.01 + 0.001 * seed])   # own parameters: own coefficient table in the shared cache
    out = None
    for _ in range(rounds):
        a = ext.interp_forward(I, u, 1.0)
        b = ext.compose(u, v, -0.1, 1.0)
        c = ext.Ad_star(u, v)
        d = ext.jacobian_times_vectorfield_forward(u, v, True, False)
        e = met.sharp(v)
        dI, du = ext.interp_backward(go, I, u, 1.0, True, True)
        f = lm.expmap(met, 0.05 * v, num_steps=3)
        out = (a, b, c, d, e, du, f, dI)
    torch.cuda.current_stream().synchronize()
    return out


def test_entry_points_from_four_threads_on_four_streams():
    import lagomorph_amd as lm

    shapes = [(24, 28, 64), (32, 32, 32), (20, 36, 40), (24, 28, 64)]   # (two threads share a shape: same cache keys)
    ref = [_work(lm, s, shapes[s], 1) for s in range(4)]
    torch.cuda.synchronize()
    got, errors = [None] * 4, []

    def run(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                got[i] = _work(lm, i, shapes[i], 12)
        except Exception as e:   # pragma: no cover - reported below
            errors.append((i, repr(e)))

    lm.lagomorph_ext.fluid_cache_clear() if hasattr(lm.lagomorph_ext, "fluid_cache_clear") else None
    ts = [threading.Thread(target=run, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    names = ("interp_forward", "compose", "Ad_star", "jtv_forward", "sharp", "d_u", "expmap")
    for i in range(4):
        for name, a, b in zip(names, got[i], ref[i]):
            assert torch.equal(a, b), f"thread {i}: {name} differs from the single-thread result"
        dI, rI = got[i][7], ref[i][7]   # scatter-add: arrival order of the atomics (north_star's bound)
        assert float((dI - rI).abs().max()) <= 1e-5 * float(rI.abs().max()), f"thread {i}: d_I"
