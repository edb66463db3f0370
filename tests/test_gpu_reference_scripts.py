"""GPU parity at the sizes of the reference's own hand tests (the only pins the reference holds for the RL half: MATLAB is closed and
the scripts store no vectors), every case through the C ABI against the CPU oracle:

  gauss3d_gpu_test.m:12-16      the full 5 x 5 sigma x kernel product on [32 64 32] and [512 512 256], max error 5e-5
  edgetaper_3d_test.m:8-49      512^3, rand volume, random UN-symmetric 9 x 9 x 21 PSF, max 1e-5 / mean 1e-6
  edgetaper_3d_test.m:51-75     64 x 64 x 32, 15 x 15 x 7 PSF, central slice; :85-97 a 7 x 6 x 5 volume with a 3^3 PSF
  supplements/otf_gpu_test.m:8-11,82   120 x 120 x 150, 5 sigmas x 5 kernels, norm(diff) / norm(ref) < 2e-6
  mex_incubator/deconFFT_test.m:6-13   one fused RL step on the six shapes (incl. 512 x 512 x 128 with lambda), 2e-4

MATLAB sizes are [X Y Z]; arrays here are (Z, Y, X).  Where the direct form of the oracle would take minutes (a 1701-tap blur of
512^3 voxels, 51-tap filters of a 97-Mvoxel volume) the oracle runs in float64 FFT form, or on crops of the volume that keep
their distance from the crop's artificial edges -- each checked against the direct form on a small case in the same test."""

import numpy as np
import pytest
import torch

from oracle import rl_oracle as R

pytestmark = pytest.mark.gpu

SIGMAS = [2.5, [1.5, 1.5, 2.5], [0.5, 0.5, 2.5], 0.25, 8]              # gauss3d_gpu_test.m:13
KSIZES = ["auto", [9, 11, 15], 3, 51, [25, 25, 25]]                    # :14


def _ksize(sigma, ksz):
    s3 = [float(sigma)] * 3 if np.isscalar(sigma) else [float(v) for v in sigma]
    if ksz == "auto":
        return [max(3, 2 * int(np.ceil(3 * s)) + 1) for s in s3]       # odd_kernel_size, :164-170
    return [int(ksz)] * 3 if np.isscalar(ksz) else [int(v) for v in ksz]


@pytest.mark.parametrize("sigma", SIGMAS, ids=lambda s: f"s{s}")
@pytest.mark.parametrize("ksz", KSIZES, ids=lambda k: f"k{k}")
def test_gauss3d_script_small_volume(dev, sigma, ksz):
    """[32 64 32]: the replicate-padded volume is filtered and the unpadded part compared, like the script (:52-97)."""
    from ipp_amd import decon
    k = _ksize(sigma, ksz)                                             # [x y z]
    rng = np.random.default_rng(0)
    x = rng.random((32, 64, 32), dtype=np.float32)
    x /= x.max()
    pad = [(k[2] // 2,) * 2, (k[1] // 2,) * 2, (k[0] // 2,) * 2]
    xp = np.pad(x, pad, mode="edge")
    got = decon.gauss3d_gpu(torch.from_numpy(xp).to(dev), sigma, k).cpu().numpy()
    want = R.gauss3d(xp, sigma, k)
    un = tuple(slice(p[0], s - p[1]) for p, s in zip(pad, xp.shape))
    assert np.abs(got[un] - want[un]).max() < 5e-5                     # SINGLE_THRESH, :15


@pytest.mark.parametrize("sigma", SIGMAS, ids=lambda s: f"s{s}")
@pytest.mark.parametrize("ksz", KSIZES, ids=lambda k: f"k{k}")
def test_gauss3d_script_large_volume(dev, sigma, ksz):
    """[512 512 256] (replicate-padded: up to 562 x 562 x 306) on the device in one call; the oracle on three crops -- the corner at
    the origin, the far corner, the centre -- compared where the crop's own edges cannot reach (half a kernel inside)."""
    from ipp_amd import decon
    k = _ksize(sigma, ksz)
    kzyx = [k[2], k[1], k[0]]
    rng = np.random.default_rng(0)
    x = rng.random((256, 512, 512), dtype=np.float32)
    x /= x.max()
    pad = [(kk // 2,) * 2 for kk in kzyx]
    xp = np.pad(x, pad, mode="edge")
    got = decon.gauss3d_gpu(torch.from_numpy(xp).to(dev), sigma, k).cpu().numpy()
    ext = [min(n, 48 + 2 * (kk // 2)) for n, kk in zip(xp.shape, kzyx)]
    worst = 0.0
    for where in ("origin", "far", "centre"):
        lo = [0 if where == "origin" else (n - e if where == "far" else (n - e) // 2) for n, e in zip(xp.shape, ext)]
        crop = tuple(slice(a, a + e) for a, e in zip(lo, ext))
        want = R.gauss3d(np.ascontiguousarray(xp[crop]), sigma, k)
        # a crop edge that is not the volume's edge spoils half a kernel
        inner = tuple(slice(0 if a == 0 else kk // 2, e if a + e == n else e - kk // 2)
                      for a, e, n, kk in zip(lo, ext, xp.shape, kzyx))
        worst = max(worst, float(np.abs(got[crop][inner] - want[inner]).max()))
    assert worst < 5e-5, worst


def _edgetaper_fft_oracle(bl, psf):
    """edgetaper_3d (oracle/rl_oracle.py:edgetaper_3d) with the blur -- conv3d_gpu's replicate-clamped convolution with the
    sum-normalised PSF -- as a float64 FFT convolution of the replicate-padded volume."""
    from scipy import fft as sfft
    psf = psf.astype(np.float32)
    psfn = (psf / np.float32(psf.sum(dtype=np.float32))).astype(np.float32)
    k = psf.shape
    # conv3d_gpu.cu:77-98: out[n] = sum_m h[m] a[clamp(n + (k - 1 - k/2) - m)]: indices from n - k/2 to n + (k - 1 - k/2), so pad k/2
    # before and k - 1 - k/2 after; out[n] is then sample n + k - 1 of the full convolution of the padded array
    pre = [kk // 2 for kk in k]
    post = [kk - 1 - kk // 2 for kk in k]
    ap = np.pad(bl.astype(np.float64), list(zip(pre, post)), mode="edge")
    fs = [sfft.next_fast_len(n, real=True) for n in ap.shape]
    spec = sfft.rfftn(ap, s=fs, workers=-1)
    spec *= sfft.rfftn(psfn.astype(np.float64), s=fs, workers=-1)
    full = sfft.irfftn(spec, s=fs, workers=-1)
    del spec
    blur = full[tuple(slice(kk - 1, kk - 1 + n) for kk, n in zip(k, bl.shape))].astype(np.float32)
    tz, ty, tx = R.edgetaper_mask_vectors(bl.shape, psf.shape)
    mask = (tx[None, None, :] * ty[None, :, None]).astype(np.float32) * tz[:, None, None]
    return (mask * bl.astype(np.float32) + (np.float32(1.0) - mask) * blur).astype(np.float32)


def test_edgetaper_script(dev):
    from ipp_amd import decon
    # the FFT form of the oracle against its direct form (edgetaper_3d_test.m:51-75: 64 x 64 x 32, 15 x 15 x 7, central slice)
    rng = np.random.default_rng(7)
    A = rng.random((32, 64, 64), dtype=np.float32)
    g = np.exp(-((np.arange(15) - 7.0) ** 2) / (2 * 2.0 ** 2))
    g2 = np.outer(g, g)
    g2 /= g2.sum()
    P3 = np.stack([g2 * np.exp(-((z - 3.0) ** 2) / 6.0) for z in range(7)]).astype(np.float32)
    P3 /= P3.sum()
    want = R.edgetaper_3d(A, P3)
    assert np.abs(_edgetaper_fft_oracle(A, P3) - want).max() < 2e-7
    got = decon.edgetaper_3d(torch.from_numpy(A).to(dev), torch.from_numpy(P3).to(dev)).cpu().numpy()
    d = np.abs(got[16] - want[16])
    assert d.max() < 1e-5 and d.mean() < 1e-6                           # :66-72
    assert got.min() >= 0.0 and got.max() <= 1.0 and got.shape == A.shape  # :77-83, :99-105
    # :85-97: a 7 x 6 x 5 volume with a 3 x 3 x 3 PSF goes through (and equals the oracle)
    As = rng.random((5, 6, 7), dtype=np.float32)
    Ps = rng.random((3, 3, 3), dtype=np.float32)
    Ps /= Ps.sum()
    gs = decon.edgetaper_3d(torch.from_numpy(As).to(dev), torch.from_numpy(Ps).to(dev)).cpu().numpy()
    assert np.abs(gs - R.edgetaper_3d(As, Ps)).max() < 1e-5
    # :8-49: 512^3, rand volume, random 9 x 9 x 21 PSF (no symmetry of any kind)
    rng = np.random.default_rng(42)
    bl = rng.random((512, 512, 512), dtype=np.float32)
    psf = rng.random((21, 9, 9), dtype=np.float32)
    psf /= psf.sum()
    got = decon.edgetaper_3d(torch.from_numpy(bl).to(dev), torch.from_numpy(psf).to(dev)).cpu().numpy()
    want = _edgetaper_fft_oracle(bl, psf)
    d = np.abs(got - want)
    assert d.max() < 1e-5 and d.mean() < 1e-6, (float(d.max()), float(d.mean()))  # pass_thresh / mean_thresh, :4-5


@pytest.mark.parametrize("sigma", [2.5, [2.5, 2.5, 2.5], [0.5, 0.5, 2.5], 0.25, 8], ids=lambda s: f"s{s}")   # otf_gpu_test.m:10
@pytest.mark.parametrize("kernel", ["auto", 9, [9, 9, 21], 3, 41], ids=lambda k: f"k{k}")                     # :9
def test_otf_gpu_script(dev, sigma, kernel):
    from ipp_amd import decon
    sz = [120, 120, 150]                                               # [x y z], :8
    if kernel == "auto":
        ks = list(sz)
    elif np.isscalar(kernel):
        ks = [min(n, int(kernel)) for n in sz]
    else:
        ks = [min(n, int(v)) for n, v in zip(sz, kernel)]
    s3 = [float(sigma)] * 3 if np.isscalar(sigma) else [float(v) for v in sigma]
    ax = [np.exp(-0.5 * ((np.arange(1, n + 1) - (n + 1) / 2.0) / s) ** 2) for n, s in zip(ks, s3)]   # :24-31
    psf = (ax[2][:, None, None] * ax[1][None, :, None] * ax[0][None, None, :])
    psf = (psf / psf.sum()).astype(np.float32)
    otf = decon.otf_gpu(torch.from_numpy(psf).to(dev), sz).cpu().numpy()
    ref = R.otf_from_psf(psf, (sz[2], sz[1], sz[0]))[:, :, : sz[0] // 2 + 1]
    rel = np.linalg.norm((otf - ref).ravel()) / max(np.linalg.norm(ref.ravel()), 1.1920929e-07)
    assert rel < 2e-6, rel                                             # :82


DECON_FFT_CASES = [([64, 64, 32], 0.0), ([128, 128, 64], 0.0), ([512, 512, 128], 0.0), ([512, 512, 128], 0.05),
                   ([32, 48, 96], 0.0), ([32, 48, 96], 0.1)]            # deconFFT_test.m:6-13


@pytest.mark.parametrize("sz,lam", DECON_FFT_CASES, ids=lambda v: str(v))
def test_decon_fft_script_one_fused_step(dev, sz, lam):
    """One RL step with otf = fftn(psf, sz) (the PSF in the CORNER of the grid, deconFFT_test.m:22-24) and its conjugate: on the
    device through the fused iteration of the hand-written pipeline (lambda = 0) / the two half-steps with the Tikhonov term
    (decon.m:67-74; the script's own Laplacian belongs to the incubator MEX, not to decon.m), against the float64 oracle."""
    from ipp_amd import capi, decon
    shape = (sz[2], sz[1], sz[0])
    rng = np.random.default_rng(1)
    bl = rng.random(shape, dtype=np.float32)
    psf = R.gaussian_psf((9, 15, 15), (3.0, 3.0, 3.0))                 # fspecial3('gaussian', [15 15 9], 3), normalised
    # the library centres its PSF (ifftshift of the centre-padded kernel, decon.m:131-133); a kernel of twice the extent with the
    # PSF in its upper half puts psf(1,1,1) on the grid's origin for even grids: exactly fftn(psf, sz)
    big = np.zeros(tuple(2 * k for k in psf.shape), np.float32)
    big[psf.shape[0]:, psf.shape[1]:, psf.shape[2]:] = psf
    otf = np.fft.fftn(psf.astype(np.float64), s=shape, axes=(0, 1, 2))
    assert np.abs(R.otf_from_psf(big, shape) - otf).max() < 1e-12
    b64 = bl.astype(np.float64)
    buf = np.real(np.fft.ifftn(np.fft.fftn(b64) * otf)).astype(np.float32)
    buf = (bl / np.maximum(buf, R.EPS_SINGLE)).astype(np.float32)
    buf = np.real(np.fft.ifftn(np.fft.fftn(buf.astype(np.float64)) * np.conj(otf))).astype(np.float32)
    if lam > 0:
        want = (bl * buf * (np.float32(1) - np.float32(lam)) + R.convn_same(bl, R._reg_kernel()) * np.float32(lam)).astype(np.float32)
    else:
        want = (bl * buf).astype(np.float32)
    want = np.abs(want)
    ctx = decon.RLContext(shape, big, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
    t = torch.from_numpy(bl).to(dev)
    if lam > 0:
        ratio, reg = torch.empty_like(t), torch.empty_like(t)
        capi.check(capi.lib().mi_rl_reg_term(dev.index, capi.current_stream_ptr(dev), t.data_ptr(), reg.data_ptr(), *sz))
        ctx.forward_ratio(t, ratio)
        ctx.adjoint_update(ratio, t, lam, reg)
    else:
        ctx.iterate(t, None if ctx.fuses else torch.empty_like(t), 1)   # (the smallest grid runs unfused: scratch for the ratio)
    got = t.cpu().numpy()
    ctx.close()
    assert np.abs(got - want).max() < 2e-4, float(np.abs(got - want).max())   # tol, :15
    assert np.linalg.norm((got - want).ravel()) / np.linalg.norm(want.ravel()) < 1e-5
