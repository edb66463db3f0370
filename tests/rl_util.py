"""Shared helpers of the RL parity tests: the tolerance metric and seeded asymmetric PSFs."""
import numpy as np


def rel_max(got, want):
    """max |got - want| / max |want|"""
    return float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max() / max(float(np.abs(want).max()), 1e-30))


def assert_close(got, want, rel=1e-4, rel_l2=1e-5, pt_rel=1e-4, pt_abs=1e-7, what=""):
    """BASELINE.json north_star: 1e-4 *relative*.  Three bounds, all required:
      * max |d| <= rel * max |want|               (the round-1 metric)
      * ||d||_2 <= rel_l2 * ||want||_2
      * point-wise |d| <= pt_rel * |want| + pt_abs * max(1, max |want|)   (background voxels next to bright beads are held to
        1e-4 of THEIR value; the absolute floor is the fp32 rounding of a transform whose largest output is max |want|)."""
    g = np.asarray(got, dtype=np.float64)
    w = np.asarray(want, dtype=np.float64)
    assert g.shape == w.shape, (g.shape, w.shape)
    d = np.abs(g - w)
    wmax = max(float(np.abs(w).max()), 1e-30)
    assert float(d.max()) <= rel * wmax, f"{what} max error {d.max() / wmax:.3e} of the maximum"
    l2 = float(np.sqrt((d * d).sum() / max(float((w * w).sum()), 1e-300)))
    assert l2 <= rel_l2, f"{what} relative L2 error {l2:.3e}"
    allow = pt_rel * np.abs(w) + pt_abs * max(1.0, wmax)
    worst = float((d / allow).max())
    assert worst <= 1.0, f"{what} point-wise error {worst:.3f} x the allowance at {np.unravel_index(np.argmax(d / allow), d.shape)}"


def asymmetric_psf(kshape, seed=0):
    """Seeded PSF without any mirror symmetry: random positive samples under a Gaussian envelope whose centre is off the
    array centre, with a linear skew along every axis; sum 1."""
    rng = np.random.default_rng(1000 + seed)
    p = rng.random(kshape) + 0.1
    for ax, k in enumerate(kshape):
        r = np.arange(k) - (k - 1) / 2.0 - 0.7
        env = np.exp(-0.5 * (r / max(k / 3.5, 0.8)) ** 2) * np.linspace(0.6, 1.4, k)
        p *= env.reshape([-1 if i == ax else 1 for i in range(3)])
    p /= p.sum()
    p = p.astype(np.float32)
    assert not np.allclose(p, p[::-1, ::-1, ::-1], rtol=1e-2)
    return p


def assert_close_device(got, want, rel=1e-4, rel_l2=1e-5, pt_rel=1e-4, pt_abs=1e-7, what=""):
    """`assert_close` for torch tensors that stay on the device (volumes of several GB): the same three bounds, evaluated plane
    by plane with float64 accumulators."""
    import torch
    assert got.shape == want.shape, (got.shape, want.shape)
    wmax = max(float(want.abs().max()), 1e-30)
    dmax, worst, d2, w2 = 0.0, 0.0, 0.0, 0.0
    floor = pt_abs * max(1.0, wmax)
    for z in range(got.shape[0]):
        g, w = got[z].double(), want[z].double()
        d = (g - w).abs()
        dmax = max(dmax, float(d.max()))
        worst = max(worst, float((d / (pt_rel * w.abs() + floor)).max()))
        d2 += float((d * d).sum())
        w2 += float((w * w).sum())
    assert dmax <= rel * wmax, f"{what} max error {dmax / wmax:.3e} of the maximum"
    l2 = (d2 / max(w2, 1e-300)) ** 0.5
    assert l2 <= rel_l2, f"{what} relative L2 error {l2:.3e}"
    assert worst <= 1.0, f"{what} point-wise error {worst:.3f} x the allowance"


def delta_on_ones_closed_form(psf, shifts, amp, d):
    """One RL iteration (decon.m:162-186, circular) on bl = 1 + amp * delta_p, evaluated at y = p + d in float64.  With P(u) =
    psf[u + shift] (sample j lands on offset j - shift, decon.m:131-133), sum(P) = 1:  conv = 1 + amp P(x - p), ratio = bl / conv,
    a(y) = sum_x ratio(x) P(x - y) = 1 - sum_u g(u) P(u - d) + [amp / (1 + amp P(0))] P(-d),  g = amp P / (1 + amp P);
    the iteration returns |bl(y) a(y)|."""
    P = psf.astype(np.float64)
    g = amp * P / (1.0 + amp * P)
    k = P.shape
    acc = 0.0
    lo = [max(0, dd) for dd in d]                        # j (index of g) with j - d inside the PSF
    hi = [min(kk, kk + dd) for kk, dd in zip(k, d)]
    if all(h > l for l, h in zip(lo, hi)):
        gs = g[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
        ps = P[lo[0] - d[0]:hi[0] - d[0], lo[1] - d[1]:hi[1] - d[1], lo[2] - d[2]:hi[2] - d[2]]
        acc = float((gs * ps).sum())
    jm = [s - dd for s, dd in zip(shifts, d)]            # P(-d) = psf[shift - d]
    pm = float(P[jm[0], jm[1], jm[2]]) if all(0 <= j < kk for j, kk in zip(jm, k)) else 0.0
    p0 = float(P[shifts[0], shifts[1], shifts[2]])
    a = 1.0 - acc + amp / (1.0 + amp * p0) * pm
    bl = 1.0 + (amp if tuple(d) == (0, 0, 0) else 0.0)
    return abs(bl * a)
