"""The fluid metric's differential operator in the SPATIAL domain (test infrastructure; BASELINE configs[2] "FFT vs
finite-difference solver", SURVEY section 7: valid for `flat` only).

The reference applies L = l^2 in the Fourier domain with the symbol (cuda/metric.cu:236-254, LUTs lagomorph/metric.py:66-75)
    l_cc = gamma + alpha sum_d w_d - beta w_c,   l_cd = beta s_c s_d  (c != d),   w_d = 2 (1 - cos theta_d), s_d = sin theta_d.
w_d is the symbol of MINUS the periodic second difference  (d2_d f)(x) = f(x + e_d) - 2 f(x) + f(x - e_d), and i s_d the symbol
of the periodic central difference  (D_d f)(x) = (f(x + e_d) - f(x - e_d)) / 2.  Hence, on a periodic grid,
    (l v)_c = gamma v_c - alpha sum_d d2_d v_c + beta d2_c v_c - beta sum_{d != c} D_c D_d v_d
(the sign of the beta d2_c term is the reference's, as coded) and flat(v) = l(l(v)): ten-odd torch.roll stencils, no FFT, no
LUT.  An independent derivation: it shares no code and no table with the library or the oracle."""
import torch


def _d2(f, ax):
    return torch.roll(f, -1, ax) - 2.0 * f + torch.roll(f, 1, ax)


def _dc(f, ax):
    return 0.5 * (torch.roll(f, -1, ax) - torch.roll(f, 1, ax))


def apply_l(v, params):
    """One factor l of the operator on a field v (N, d, *spatial), periodic."""
    alpha, beta, gamma = (float(p) for p in params)
    d = v.dim() - 2
    axes = list(range(2, 2 + d))
    out = []
    for c in range(d):
        vc = v[:, c]
        acc = gamma * vc
        for ax in axes:
            acc = acc - alpha * _d2(vc, ax - 1)
        acc = acc + beta * _d2(vc, axes[c] - 1)
        for dd in range(d):
            if dd != c:
                acc = acc - beta * _dc(_dc(v[:, dd], axes[dd] - 1), axes[c] - 1)
        out.append(acc)
    return torch.stack(out, dim=1)


def flat_fd(v, params):
    """velocity -> momentum by finite differences: l applied twice."""
    return apply_l(apply_l(v, params), params)
