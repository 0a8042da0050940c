"""HIP-graph capture of the hot path.  The library promises capture-safety in four places -- no
synchronisation in `finish_launch` outside debug mode, Bluestein tables and the fluid metric's coefficient tables built
on FIRST use only (a use inside a capture must not build them silently wrong), the rocFFT guard skipping its
synchronising check (tests/test_gpu_parity.py::test_rocfft_guard_bookkeeping) -- and the host mirror adds one: the
sub-batch split of a forward-only shoot forks the capture onto its side streams and joins it again.

  * a whole default-path `expmap` (10 Euler steps, two sub-batches on two streams) captured after warm-up on a side
    stream replays bit-identically, repeatedly, and follows new contents of its input tensor;
  * a `lddmm_step` (forward, backward through every operator, in-place momentum update) captured the same way: two
    replays equal two eager steps to float32 rounding (its scatter-adds are float atomics: no two runs agree bitwise);
  * first use INSIDE a capture -- a shape whose coefficient table does not exist yet -- fails loudly (RuntimeError), and
    the library is usable afterwards.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _smooth(shape, sigma, seed, amp):
    import bench

    g = torch.Generator(device="cuda").manual_seed(seed)
    x = bench.gaussian_blur(torch.randn(shape, device="cuda", generator=g), sigma)
    return (x * (amp / x.abs().max())).contiguous()


def _capture(fn, warm=3):
    """torch's recipe: warm up on a side stream, capture there, hand back (graph, outputs)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = fn()
    return graph, out


@pytest.mark.parametrize("sp,B,steps", [((64, 64, 64), 6, 10), ((48, 40, 56), 5, 4), ((96, 80), 4, 6), ((33, 29, 31), 3, 5)])
def test_captured_expmap_replays_same_bits(sp, B, steps):
    import lagomorph_amd as lm
    from lagomorph_amd import lddmm

    d = len(sp)
    met = lm.FluidMetric([0.1, 0.0, 0.01])
    m = _smooth((B, d) + sp, 3.0, 1, 1.0)
    with torch.no_grad():
        m *= 2.5 / met.sharp(m).abs().max()
        assert lddmm.EXPMAP_STREAMS == 2   # the default path: two sub-batches on side streams, inside the capture too
        ref = lm.expmap(met, m, num_steps=steps)
        graph, out = _capture(lambda: lm.expmap(met, m, num_steps=steps))
        for _ in range(3):
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, ref)
        # the graph reads its input tensor at replay time: new momenta, new displacement
        m2 = _smooth((B, d) + sp, 2.0, 2, 1.0)
        m2 *= 1.5 / met.sharp(m2).abs().max()
        ref2 = lm.expmap(met, m2, num_steps=steps)
        m.copy_(m2)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref2) and not torch.equal(ref2, ref)
    del graph


@pytest.mark.parametrize("sp", [(48, 40, 56), (40, 36)])
def test_captured_lddmm_step_replays(sp):
    import lagomorph_amd as lm

    d, B = len(sp), 4
    met = lm.FluidMetric([0.1, 0.0, 0.01])
    I0 = _smooth((1, 1) + sp, 2.0, 3, 1.0)
    img = (I0 + 0.2 * _smooth((B, 1) + sp, 2.0, 4, 1.0)).contiguous()
    m0 = _smooth((B, d) + sp, 3.0, 5, 1.0)
    with torch.no_grad():
        m0 *= 1.5 / met.sharp(m0).abs().max()
    V = float(np.prod(sp))
    kw = dict(integration_steps=3, reg_weight=1e-1, learning_rate_pose=3e-7 * V)

    # eager: two steps
    Ie = I0.clone().requires_grad_(True)
    me = m0.clone()
    for _ in range(2):
        me, le, re_ = lm.lddmm_step(Ie, me, img, met, B, **kw)
    torch.cuda.synchronize()

    # captured: the same step as a graph over static tensors (momenta updated in place, I.grad accumulated in place)
    Ig = I0.clone().requires_grad_(True)
    mg = m0.clone()
    Ig.grad = torch.zeros_like(Ig)
    keep = {}

    def step():
        _, l, r = lm.lddmm_step(Ig, mg, img, met, B, **kw)
        keep["loss"], keep["reg"] = l, r
        return l

    graph, _ = _capture(step, warm=2)
    with torch.no_grad():   # the warm-up moved the state: reset it
        mg.copy_(m0)
        Ig.grad.zero_()
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()

    def rel(a, b):
        return float((a.double() - b.double()).abs().max() / b.double().abs().max())

    errs = {"momenta": rel(mg, me), "I.grad": rel(Ig.grad, Ie.grad), "loss": rel(keep["loss"], le), "reg": rel(keep["reg"], re_)}
    print("captured lddmm_step, two replays vs two eager steps:", errs)
    assert float((me - m0).abs().max()) > 0
    assert max(errs.values()) <= 1e-5, errs
    del graph


def test_first_use_inside_a_capture_fails_loudly():
    """A shape the library has never seen: its per-frequency coefficient table would have to be allocated, filled and
    waited for inside the capture.  The call must raise; afterwards the same call outside a capture works, and a capture
    of the now-warm shape replays correctly."""
    import lagomorph_amd as lm

    ext = lm.lagomorph_ext
    ext.fluid_cache_clear() if hasattr(ext, "fluid_cache_clear") else None
    sp = (40, 24, 48)   # a tuned-pass shape (powers of two times 3 / 5) no other test of this file uses
    met = lm.FluidMetric([0.13, 0.0, 0.017])   # parameters of its own: no table from another test matches
    m = _smooth((2, 3) + sp, 2.0, 9, 1.0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    raised = False
    try:
        with torch.cuda.graph(graph):
            met.sharp(m)
    except RuntimeError as e:
        raised = True
        print("first use inside a capture:", str(e).splitlines()[0][:160])
    assert raised, "a first use inside a capture must not pass silently"
    del graph
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = met.sharp(m)   # outside: builds the table
        assert torch.isfinite(ref).all() and float(ref.abs().max()) > 0
        graph2, out = _capture(lambda: met.sharp(m), warm=1)
        graph2.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
    del graph2
