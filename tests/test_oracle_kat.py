"""CPU: the oracle (and the host mirror's compositions on top of it) against the
known-answer table of SURVEY.md 8(c) -- outputs of the reference's own kernels."""
import torch

import kat


def test_oracle_matches_reference_kat_float64(oracle_ext):
    import lagomorph_amd as lm

    res = kat.evaluate(oracle_ext, lm, torch.float64, "cpu")
    # the table is printed to 13 significant digits; sums carry cancellation, hence the floor
    kat.check(res, rel=2e-12, abs_floor=2e-9)


def test_oracle_float32_tracks_float64(oracle_ext):
    """fp32 runs agree with the table to <=3e-6 relative in sums, <=1e-6 in elements."""
    import lagomorph_amd as lm

    res = kat.evaluate(oracle_ext, lm, torch.float32, "cpu")
    # sharp() divides by gamma^2 = 1e-4 at the zero frequency: its fp32 error is relative to 1e3-sized values
    kat.check(res, rel=2e-5, abs_floor=2e-4)


def test_threaded_oracle_is_bit_identical_to_the_scalar_one():
    """bench.py's cpu_baseline leg may run the forward kernels with OpenMP; voxels are independent
    there, so the thread count must not change a bit."""
    import numpy as np
    import oracle.lago_oracle as orc

    rng = np.random.default_rng(0)
    sp = (9, 10, 11)
    I = rng.standard_normal((2, 2) + sp).astype(np.float32)
    u = (2 * rng.standard_normal((2, 3) + sp)).astype(np.float32)
    m = rng.standard_normal((2, 3) + sp).astype(np.float32)

    def run():
        return [orc.interp_forward(I, u, 0.7), orc.jacobian_times_vectorfield_forward(u, m, True, False),
                orc.jacobian_times_vectorfield_forward(u, m, False, True),
                orc.fluid_metric_apply(m, [0.1, 0.05, 0.01], True), orc.fluid_metric_apply(m, [0.1, 0.05, 0.01], False)]

    one = run()
    orc.set_threads(4)
    try:
        four = run()
    finally:
        orc.set_threads(1)
    for a, b in zip(one, four):
        assert np.array_equal(a, b)
