"""Shared evaluator for the SURVEY 8(c) known-answer table (tests/golden/kat_survey.json).

`ext` is any object with the lagomorph_ext surface over torch tensors (the HIP shim on a GPU,
or the oracle's OracleExt on CPU); `lm` is the lagomorph_amd package whose host mirror supplies
the compositions.  Returns {row name: tensor}."""
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    with open(os.path.join(HERE, "golden", "kat_survey.json")) as f:
        return json.load(f)


def field(spec, dtype, device):
    shape = spec["shape"]
    n = int(np.prod(shape))
    a = spec["amp"] * np.sin(spec["a"] * np.arange(n, dtype=np.float64) + spec["b"])
    return torch.from_numpy(a.reshape(shape)).to(dtype).to(device)


def evaluate(ext, lm, dtype=torch.float64, device="cpu"):
    k = load()["inputs"]
    I, u, m, g = (field(k[n], dtype, device) for n in ("I", "u", "m", "g"))
    I2, g2 = field(k["I2"], dtype, device), field(k["g2"], dtype, device)
    A = torch.tensor(k["A"], dtype=dtype, device=device)
    T = torch.tensor(k["T"], dtype=dtype, device=device)
    out = {}
    out["interp_forward(I,u,1.0)"] = ext.interp_forward(I, u, 1.0)
    out["interp_forward(I[:1],u,-0.5)"] = ext.interp_forward(I[:1].contiguous(), u, -0.5)
    dI, du = ext.interp_backward(g, I, u, 1.0, True, True)
    out["interp_backward(g,I,u,1,T,T).d_I"] = dI
    out["interp_backward(g,I,u,1,T,T).d_u"] = du
    dI, du = ext.interp_backward(g, I[:1].contiguous(), u, 1.0, True, True)
    out["interp_backward(g,I[:1],u,1,T,T).d_I"] = dI
    for disp, trans, tag in ((True, False, "T,F"), (True, True, "T,T"), (False, False, "F,F"), (False, True, "F,T")):
        out[f"jtv_forward(u,m,{tag})"] = ext.jacobian_times_vectorfield_forward(u, m, disp, trans)
    dv, dw = ext.jacobian_times_vectorfield_backward(m, u, m, True, False, True, True)
    out["jtv_backward(m,u,m,T,F).d_v"] = dv
    out["jtv_backward(m,u,m,T,F).d_w"] = dw
    out["jtv_adjoint_forward(m,u)"] = ext.jacobian_times_vectorfield_adjoint_forward(m, u)
    dv, dw = ext.jacobian_times_vectorfield_adjoint_backward(u, m, u, True, True)
    out["jtv_adjoint_backward(u,m,u).d_v"] = dv
    met = lm.FluidMetric([0.1, 0.05, 0.01])
    out["sharp(m;.1,.05,.01)"] = met.sharp(m)
    out["flat(m;.1,.05,.01)"] = met.flat(m)
    out["Ad_star(0.3u,m)"] = lm.Ad_star(0.3 * u, m)
    out["ad_star(u,m)"] = lm.ad_star(u, m)
    out["compose_disp_vel(u,m,-0.1)"] = lm.compose_disp_vel(u, m, dt=-0.1)
    out["expmap(.1,.05,.01;0.005m;3)"] = lm.expmap(lm.FluidMetric([0.1, 0.05, 0.01]), 0.005 * m, num_steps=3)
    out["expmap(.1,0,.01;0.005m;10)"] = lm.expmap(lm.FluidMetric([0.1, 0.0, 0.01]), 0.005 * m, num_steps=10)
    out["affine_interp_forward(I2,A,T)"] = ext.affine_interp_forward(I2, A, T)
    dI, dA, dT = ext.affine_interp_backward(g2, I2, A, T, True, True, True)
    out["affine_interp_backward(g2,I2,A,T).d_A"] = dA
    out["affine_interp_backward(g2,I2,A,T).d_T"] = dT
    out["regrid_forward(I,[5,7,9],[1.5,2,2.5],[.75,4/6,.625])"] = ext.regrid_forward(
        I, [5, 7, 9], [1.5, 2.0, 2.5], [3 / 4, 4 / 6, 5 / 8]
    )
    return out


def stats(t):
    t = t.detach().double().cpu()
    return [t.sum().item(), (t * t).sum().item(), t.flatten()[0].item(), t.flatten()[-1].item()]


def check(results, rel, abs_floor):
    """Compare against the table: |got - exp| <= rel*|exp| + abs_floor per statistic."""
    rows = load()["rows"]
    missing = set(rows) - set(results)
    assert not missing, f"rows not evaluated: {missing}"
    bad = []
    for name, exp in rows.items():
        got = stats(results[name])
        for gq, eq, what in zip(got, exp, ("sum", "sumsq", "first", "last")):
            if eq is None:
                continue
            if not abs(gq - eq) <= rel * abs(eq) + abs_floor:
                bad.append((name, what, gq, eq))
    assert not bad, "KAT mismatches: " + "; ".join(f"{n}[{w}] got {g:.12e} want {e:.12e}" for n, w, g, e in bad)
