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
