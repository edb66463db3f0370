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
