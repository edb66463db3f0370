"""GPU parity with ASYMMETRIC PSFs: with a centro-symmetric PSF forward == adjoint, so these are the tests that see the
adjoint (conj OTF / psf.inv), the complex-OTF form of the native pipeline, an explicit psf.inv != flip(psf), and the
even-extent placement rules.  Every case runs >= 4 RL iterations against oracle/rl_oracle.py (decon.m:61-79,162-186,
LsDeconv.m:163) through the C ABI."""
import numpy as np
import pytest
import torch

from oracle import rl_oracle as R
from tests.rl_util import assert_close, asymmetric_psf
from tests.slab_util import lockstep_iterate

pytestmark = pytest.mark.gpu

KSHAPES = [(7, 5, 9), (6, 4, 8), (5, 6, 7)]          # odd, even, mixed extents (z, y, x)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _vol(shape, seed):
    return R.bead_volume(shape, seed=seed, psf=R.gaussian_psf((5, 5, 5), (1.0, 1.0, 1.0)))


# ------------------------------------------------------------------ deconFFT flavour (decon.m:127-204)
@pytest.mark.parametrize("engine", [1, 2])
@pytest.mark.parametrize("kshape", KSHAPES)
@pytest.mark.parametrize("shape,F_xyz", [((16, 32, 64), None),          # native extents: the hand-written pipeline, complex OTF form
                                         ((20, 36, 44), None),          # rocFFT extents
                                         ((20, 36, 44), (64, 48, 32))])  # zero-padded to fft_shape (decon.m:144)
@pytest.mark.parametrize("niter,lam,interval", [(5, 0.0, 0), (6, 0.05, 2)])
def test_decon_fft_asymmetric_psf(dev, engine, kshape, shape, F_xyz, niter, lam, interval):
    from ipp_amd import decon
    psf = asymmetric_psf(kshape, seed=sum(kshape))
    vol = _vol(shape, 41)
    Fz = shape if F_xyz is None else (F_xyz[2], F_xyz[1], F_xyz[0])
    want = R.decon_fft(vol, psf, Fz, niter, lam, 0.0, interval)
    got = decon.decon(_t(vol, dev), psf, niter, lam, 0.0, interval, 1, True,
                      F_xyz if F_xyz is not None else (shape[2], shape[1], shape[0]), False, engine=engine).cpu().numpy()
    assert_close(got, want, what=f"deconFFT k={kshape} F={Fz} engine={engine}")


# ------------------------------------------------------------------ deconSpatial flavour (decon.m:26-124)
@pytest.mark.parametrize("engine", [1, 2])
@pytest.mark.parametrize("kshape", KSHAPES)
@pytest.mark.parametrize("inv", ["flip", "other"])
@pytest.mark.parametrize("niter,lam,interval", [(4, 0.0, 0), (6, 0.05, 2)])
def test_decon_spatial_asymmetric_psf_and_explicit_inv(dev, engine, kshape, inv, niter, lam, interval, monkeypatch):
    """psf.inv as LsDeconv.m:163 builds it (the flipped PSF) and a psf.inv that is NOT the flipped PSF: decon.m:64 convolves
    with whatever the struct holds."""
    from ipp_amd import decon
    monkeypatch.setenv("MI_FFT_NATIVE_INFLATE", "100")
    psf = asymmetric_psf(kshape, seed=3 + sum(kshape))
    psf_inv = R.flip3(psf) if inv == "flip" else asymmetric_psf(kshape, seed=77)
    vol = _vol((20, 36, 44), 43)
    want = R.decon_spatial(vol, psf, niter, lam, 0.0, interval, psf_inv=psf_inv)
    struct = {"psf": psf, "inv": psf_inv}
    got = decon.decon(_t(vol, dev), struct, niter, lam, 0.0, interval, 1, False, None, False, engine=engine).cpu().numpy()
    assert_close(got, want, what=f"deconSpatial k={kshape} inv={inv} engine={engine}")


# ------------------------------------------------------------------ fused mi_rl_iterate and the half-steps
@pytest.mark.parametrize("kshape", KSHAPES)
@pytest.mark.parametrize("shape", [(16, 32, 64), (8, 96, 32), (96, 16, 192),
                                   # 1024-point transforms (three LDS round trips with a 16-point stage) on each axis: z in the paired
                                   # and (8 rows) in the plain layout -- both with the complex OTF requested behind that stage --, y, x
                                   (1024, 16, 32), (1024, 8, 32), (16, 1024, 32), (8, 16, 2048)])
def test_fused_iterate_asymmetric_psf(dev, shape, kshape):
    from ipp_amd import capi, decon
    psf = asymmetric_psf(kshape, seed=5)
    vol = _vol(shape, 47)
    ctx = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
    assert ctx.fuses and not ctx.otf_is_real         # an asymmetric PSF needs the complex OTF form
    a, b = _t(vol, dev), _t(vol, dev)
    ratio = torch.empty_like(a)
    ctx.iterate(a, None, 5)
    for _ in range(5):
        ctx.forward_ratio(b, ratio)
        ctx.adjoint_update(ratio, b)
    want = R.decon_fft(vol, psf, shape, 5, skip_edgetaper=True)
    assert_close(a.cpu().numpy(), want, what="fused")
    assert_close(b.cpu().numpy(), want, what="half-steps")


@pytest.mark.parametrize("kshape", KSHAPES)
@pytest.mark.parametrize("boundary,engine", [(2, 2), (2, 1), (0, 2), (0, 1)])
def test_adjoint_is_exact_transpose_asymmetric_psf(dev, kshape, boundary, engine, monkeypatch):
    """<conv(a), b> == <a, conv_adj(b)> holds for the circular pair (OTF, conj OTF) and, with odd extents, for the zero-boundary
    pair (psf, flip(psf)); and each side equals the oracle's convolution."""
    from ipp_amd import decon
    monkeypatch.setenv("MI_FFT_NATIVE_INFLATE", "100")
    shape = (16, 32, 64)
    psf = asymmetric_psf(kshape, seed=9)
    inv = None if boundary == 2 else R.flip3(psf)
    ctx = decon.RLContext(shape, psf, inv, boundary=boundary, engine=engine, device=dev)
    rng = np.random.default_rng(3)
    a_np = (rng.random(shape) + 0.5).astype(np.float32)
    b_np = (rng.random(shape) + 0.5).astype(np.float32)
    a, b = _t(a_np, dev), _t(b_np, dev)
    ra = torch.empty_like(a)
    ctx.forward_ratio(a, ra)          # ra = a / conv(a)
    conv_a = (a / ra).cpu().numpy()
    ones = torch.ones_like(a)
    ctx.adjoint_update(b, ones)       # ones <- |1 * conv_adj(b)|
    adj_b = ones.cpu().numpy()
    if boundary == 2:
        otf = R.otf_from_psf(psf, shape)
        want_fwd = np.real(np.fft.ifftn(np.fft.fftn(a_np.astype(np.float64)) * otf))
        want_adj = np.real(np.fft.ifftn(np.fft.fftn(b_np.astype(np.float64)) * np.conj(otf)))
    else:
        want_fwd = R.convn_same(a_np, psf)
        want_adj = R.convn_same(b_np, inv)
    assert_close(conv_a, want_fwd, rel=2e-5, what="forward")
    assert_close(adj_b, want_adj, rel=2e-5, what="adjoint")
    if boundary == 2 or all(k % 2 for k in kshape):
        lhs = float((conv_a.astype(np.float64) * b_np).sum())
        rhs = float((a_np.astype(np.float64) * adj_b).sum())
        assert abs(lhs - rhs) / abs(lhs) < 1e-5


# ------------------------------------------------------------------ slabs
@pytest.mark.parametrize("flavour", ["fft", "spatial"])
@pytest.mark.parametrize("engine", [1, 2])
@pytest.mark.parametrize("kshape", [(5, 7, 5), (4, 6, 6)])
def test_four_slabs_lockstep_asymmetric_psf(dev, flavour, engine, kshape, monkeypatch):
    from ipp_amd import slab
    monkeypatch.setenv("MI_FFT_NATIVE_INFLATE", "100")
    psf = asymmetric_psf(kshape, seed=21)
    vol = _vol((16, 128, 32), 33)
    slabs = [slab.SlabRL(vol.shape, psf, rank=r, world_size=4, device=dev, flavour=flavour, engine=engine, volume=vol)
             for r in range(4)]
    got = lockstep_iterate(slabs, 4).cpu().numpy()
    want = (R.decon_fft(vol, psf, vol.shape, 4, skip_edgetaper=True) if flavour == "fft"
            else R.decon_spatial(vol, psf, 4, skip_edgetaper=True))
    assert_close(got, want, what=f"4 slabs {flavour} engine={engine}")
