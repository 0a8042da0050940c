"""affine_atlas / StandardizedDataset (host mirror of affine.py:288-438) on the oracle backend (CPU)."""
import numpy as np
import pytest
import torch


def _subjects(lm, n, sp, seed=3):
    """A smooth blob pushed through small random affine maps."""
    rng = np.random.default_rng(seed)
    grids = np.meshgrid(*[np.arange(s, dtype=np.float64) for s in sp], indexing="ij")
    c = [(s - 1) / 2 for s in sp]
    blob = np.exp(-sum((g - ci) ** 2 for g, ci in zip(grids, c)) / (2 * (min(sp) / 5) ** 2))
    base = torch.from_numpy(blob)[None, None]
    d = len(sp)
    A = torch.eye(d, dtype=torch.float64)[None] + 0.08 * torch.from_numpy(rng.standard_normal((n, d, d)))
    T = 0.8 * torch.from_numpy(rng.standard_normal((n, d)))
    return lm.affine_interp(base, A.contiguous(), T.contiguous()).detach()


@pytest.mark.parametrize("sp", [(14, 12), (8, 9, 7)])
def test_affine_atlas_reduces_the_loss_and_moves_the_parameters(oracle_ext, sp):
    import lagomorph_amd as lm

    n, d = 6, len(sp)
    images = _subjects(lm, n, sp)
    As = torch.zeros((n, d, d), dtype=torch.float64)
    Ts = torch.zeros((n, d), dtype=torch.float64)
    I, As2, Ts2, ep, it = lm.affine_atlas(images, As, Ts, num_epochs=6, batch_size=4, learning_rate_A=2e-3,
                                          learning_rate_T=5e-2, learning_rate_I=1.0)
    assert I.shape == (1, 1) + sp and len(ep) == 6 and len(it) == 6 * 2
    assert ep[-1] < ep[0]
    assert float(As2.abs().max()) > 0 and float(Ts2.abs().max()) > 0
    # epoch loss = sum of the per-iteration losses of that epoch
    assert ep[0] == pytest.approx(it[0] + it[1], rel=1e-12)


def test_affine_atlas_first_iteration_matches_the_formulas(oracle_ext):
    """lr = 0: the atlas stays the mean image and every loss is the plain mean squared error."""
    import lagomorph_amd as lm

    sp, n = (10, 11), 5
    images = _subjects(lm, n, sp, seed=9)
    As = torch.zeros((n, 2, 2), dtype=torch.float64)
    Ts = torch.zeros((n, 2), dtype=torch.float64)
    I, _, _, ep, it = lm.affine_atlas(images, As, Ts, num_epochs=1, batch_size=2, learning_rate_A=0.0, learning_rate_T=0.0,
                                      learning_rate_I=0.0, reg_weightA=0.3, reg_weightT=0.2)
    mean = images.mean(dim=0, keepdim=True)
    assert torch.allclose(I, mean, rtol=0, atol=1e-14)
    want = [(((mean - images[b:b + 2]) ** 2).sum() / (sp[0] * sp[1]) / images[b:b + 2].size(0)
             * (images[b:b + 2].size(0) / n)).item() for b in range(0, n, 2)]
    assert it == pytest.approx(want, rel=1e-10)
    assert ep[0] == pytest.approx(sum(want), rel=1e-10)


def test_affine_atlas_image_update_frequency(oracle_ext):
    """image_update_freq = 1 takes an image step after every minibatch (affine.py:389-396)."""
    import lagomorph_amd as lm

    sp, n = (9, 8), 4
    images = _subjects(lm, n, sp, seed=1)
    z = lambda *s: torch.zeros(s, dtype=torch.float64)
    I0, *_ = lm.affine_atlas(images, z(n, 2, 2), z(n, 2), num_epochs=1, batch_size=2, image_update_freq=0,
                             learning_rate_A=0.0, learning_rate_T=0.0, learning_rate_I=0.5)
    I1, *_ = lm.affine_atlas(images, z(n, 2, 2), z(n, 2), num_epochs=1, batch_size=2, image_update_freq=1,
                             learning_rate_A=0.0, learning_rate_T=0.0, learning_rate_I=0.5)
    mean = images.mean(dim=0, keepdim=True)
    # with A = T = 0 the deformed atlas is the atlas: d/dI of the per-minibatch loss is 2 (I - img) / (nvox B)
    g = [2 * (mean - images[b:b + 2]).sum(dim=0, keepdim=True) / (sp[0] * sp[1] * 2) for b in (0, 2)]
    assert torch.allclose(I0, mean - 0.5 * (g[0] + g[1]) / 2, atol=1e-13)
    step1 = mean - 0.5 * g[0]
    g1 = 2 * (step1 - images[2:4]).sum(dim=0, keepdim=True) / (sp[0] * sp[1] * 2)
    assert torch.allclose(I1, step1 - 0.5 * g1, atol=1e-13)


def test_standardized_dataset_inverts_the_fitted_map(oracle_ext):
    import lagomorph_amd as lm

    sp = (12, 12)
    base = _subjects(lm, 1, sp, seed=0)[0]  # (1, 12, 12), one channel
    A = torch.tensor([[[0.05, -0.03], [0.02, -0.04]]], dtype=torch.float64)
    T = torch.tensor([[0.4, -0.3]], dtype=torch.float64)
    eye = torch.eye(2, dtype=torch.float64)[None]
    moved = lm.affine_interp(base[None], (A + eye).contiguous(), T.contiguous())[0]
    ds = lm.StandardizedDataset([moved], A, T, device="cpu")
    back = ds[0]
    assert len(ds) == 1 and back.shape == base.shape
    inner = (slice(None), slice(3, 9), slice(3, 9))
    assert float((back[inner] - base[inner]).abs().max()) < 0.1 * float(base.abs().max())  # two bilinear resamplings


@pytest.mark.gpu
@pytest.mark.parametrize("sp", [(14, 12), (8, 9, 7)])
def test_affine_atlas_hip_matches_oracle_backend(sp, monkeypatch):
    """The same atlas run through the HIP kernels and through the oracle backend agree (float64; the
    scatter-add gradients differ only in summation order)."""
    import lagomorph_amd as lm
    from oracle.lago_oracle import OracleExt

    n, d = 6, len(sp)
    # subjects built on the oracle backend (CPU) so that both runs see identical inputs
    o = OracleExt()
    saved = {k: getattr(lm.lagomorph_ext, k) for k in ("affine_interp_forward", "affine_interp_backward")}
    for k in saved:
        monkeypatch.setattr(lm.lagomorph_ext, k, getattr(o, k))
    images = _subjects(lm, n, sp)
    kw = dict(num_epochs=3, batch_size=4, learning_rate_A=2e-3, learning_rate_T=5e-2, learning_rate_I=1.0,
              image_update_freq=1, affine_steps=2)
    z = lambda *s, dev="cpu": torch.zeros(s, dtype=torch.float64, device=dev)
    Ic, Ac, Tc, epc, _ = lm.affine_atlas(images, z(n, d, d), z(n, d), **kw)
    for k, f in saved.items():
        monkeypatch.setattr(lm.lagomorph_ext, k, f)
    Ig, Ag, Tg, epg, _ = lm.affine_atlas(images.cuda(), z(n, d, d, dev="cuda"), z(n, d, dev="cuda"), **kw)
    assert epg == pytest.approx(epc, rel=1e-9)
    assert torch.allclose(Ig.cpu(), Ic, rtol=0, atol=1e-10)
    assert torch.allclose(Ag.cpu(), Ac, rtol=0, atol=1e-10) and torch.allclose(Tg.cpu(), Tc, rtol=0, atol=1e-10)


def test_lddmm_atlas_builder_checkpoint_resume(oracle_ext, tmp_path):
    """Two epochs, save, one more epoch == load into a fresh builder, one more epoch (lddmm.py:238-285)."""
    import lagomorph_amd as lm

    g = torch.Generator().manual_seed(2)
    sp = (6, 6, 6)
    data = torch.randn((1, 1) + sp, generator=g, dtype=torch.float64) + 0.3 * torch.randn((4, 1) + sp, generator=g, dtype=torch.float64)
    kw = dict(batch_size=2, lddmm_integration_steps=2, reg_weight=1e-1, learning_rate_pose=1e-2, learning_rate_image=1e-1)
    a = lm.LDDMMAtlasBuilder(data, **kw)
    a.run(num_epochs=2)
    path = str(tmp_path / "ckpt.pt")
    a.save(path)
    Ia = a.run(num_epochs=1).clone()
    b = lm.LDDMMAtlasBuilder(data, **kw)
    b.load(path)
    Ib = b.run(num_epochs=1)
    assert torch.equal(Ia, Ib)
    assert all(torch.equal(x, y) for x, y in zip(a.ms, b.ms))
    assert [float(x) for x in a.epoch_losses] == [float(x) for x in b.epoch_losses]


def test_affine_atlas_result_file(tmp_path, monkeypatch):
    """save_affine_atlas writes the datasets of the reference's tool (affine.py:581-587: atlas, A, T, epoch_losses,
    iter_losses) through the h5py API when h5py is importable -- checked with an in-memory stand-in, h5py being absent
    from this image -- and load_affine_atlas reads both that and the torch.save form back."""
    import sys
    import types

    import numpy as np

    import lagomorph_amd as lm

    I = torch.randn(1, 1, 5, 6, 7)
    As, Ts = torch.randn(4, 3, 3, dtype=torch.float64), torch.randn(4, 3, dtype=torch.float64)
    el, il = [3.0, 2.0], [3.5, 3.0, 2.5, 2.0]
    p1 = lm.save_affine_atlas(str(tmp_path / "a.pt"), I, As, Ts, el, il)   # no h5py: torch.save
    back = lm.load_affine_atlas(p1)
    assert torch.equal(back[0], I) and torch.equal(back[1], As) and torch.equal(back[2], Ts) and back[3] == el and back[4] == il

    store = {}

    class File:
        def __init__(self, path, mode):
            self.path = path
            if mode == "w":
                store[path] = {}
                open(path, "wb").write(b"\x89HDF\r\n\x1a\n")

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def create_dataset(self, name, data=None):
            store[self.path][name] = np.array(data)

        def __getitem__(self, name):
            return store[self.path][name]

    fake = types.ModuleType("h5py")
    fake.File = File
    monkeypatch.setitem(sys.modules, "h5py", fake)
    p2 = lm.save_affine_atlas(str(tmp_path / "a.h5"), I, As, Ts, el, il)
    assert sorted(store[p2]) == ["A", "T", "atlas", "epoch_losses", "iter_losses"]
    assert store[p2]["atlas"].shape == (1, 1, 5, 6, 7) and store[p2]["A"].shape == (4, 3, 3) and store[p2]["T"].shape == (4, 3)
    back = lm.load_affine_atlas(p2)
    assert torch.equal(back[0], I) and torch.equal(back[1], As) and torch.equal(back[2], Ts) and back[3] == el and back[4] == il
