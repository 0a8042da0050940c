import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the checker (oracle) and the product library if they are stale/missing."""
    import subprocess

    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    # by path: `import lagomorph_amd` needs the library this builds
    import importlib.util

    spec = importlib.util.spec_from_file_location("lagomorph_amd_build", os.path.join(ROOT, "lagomorph_amd", "build.py"))
    lbuild = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lbuild)
    lbuild.build()
    # LAGO_TEST_DEBUG=1: the whole session in the library's debug mode (every launch followed by a stream synchronisation,
    # kernel faults returned by the call that caused them) -- a slow one-off sanity run, not the default
    if os.environ.get("LAGO_TEST_DEBUG") == "1" and _has_gpu():
        import lagomorph_amd

        lagomorph_amd.set_debug_mode(True)


@pytest.fixture
def oracle_ext(monkeypatch):
    """Stand the CPU oracle in for lagomorph_ext (tests only) so the host mirror's
    compositions can run on CPU tensors.  The product never does this."""
    import lagomorph_amd
    from oracle.lago_oracle import OracleExt

    ext = OracleExt()
    mod = lagomorph_amd.lagomorph_ext
    for name in (
        "interp_forward", "interp_backward", "interp_hessian_diagonal_image", "compose",
        "jacobian_times_vectorfield_forward", "jacobian_times_vectorfield_backward",
        "jacobian_times_vectorfield_adjoint_forward", "jacobian_times_vectorfield_adjoint_backward",
        "fluid_operator", "affine_interp_forward", "affine_interp_backward", "regrid_forward", "regrid_backward",
    ):
        monkeypatch.setattr(mod, name, getattr(ext, name))
    monkeypatch.delattr(mod, "interp_backward_fused")
    monkeypatch.delattr(mod, "fluid_metric")  # the host mirror then takes its rfft / fluid_operator / irfft form
    monkeypatch.delattr(mod, "Ad_star")
    monkeypatch.delattr(mod, "ad_star")       # ... and Ad_star its interp + jacobian_times_vectorfield form
    return ext
