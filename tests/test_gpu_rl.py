"""GPU parity: the HIP Richardson-Lucy path (through the C ABI) against the CPU oracle on the same seeded inputs.
Tolerance for floating point (BASELINE.json north_star: 1e-4 relative): tests/rl_util.py:assert_close -- 1e-4 of the
maximum AND relative L2 < 1e-5 AND point-wise |d| <= 1e-4 |want| + 1e-7; tighter where the reference's own scripts
use a tighter bound.  The unstable Wiener variant keeps the max-relative metric (see below)."""
import numpy as np
import pytest
import torch

from oracle import rl_oracle as R
from tests.rl_util import assert_close

pytestmark = pytest.mark.gpu

REL = 1e-4


def _rel(got, want):
    return float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max() / max(float(np.abs(want).max()), 1e-30))


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# ------------------------------------------------------------------ conv3d_gpu (R7)
@pytest.mark.parametrize("shape,kshape", [((22, 25, 24), (3, 5, 7)), ((5, 6, 7), (3, 3, 3)), ((3, 3, 3), (3, 3, 3)),
                                          ((16, 40, 150), (15, 9, 9)), ((9, 17, 130), (1, 1, 5)), ((12, 20, 33), (4, 2, 6)),
                                          ((6, 7, 5), (9, 11, 13))])
def test_conv3d_gpu_matches_oracle(dev, shape, kshape):
    from ipp_amd import decon
    rng = np.random.default_rng(42)
    img = rng.random(shape, dtype=np.float32)
    ker = rng.random(kshape, dtype=np.float32)
    got = decon.conv3d_gpu(_t(img, dev), _t(ker, dev)).cpu().numpy()
    want = R.conv3d_replicate_loops(img, ker) if any(k % 2 == 0 for k in kshape) else R.conv3d_replicate(img, ker)
    assert got.shape == img.shape
    assert np.abs(got - want).max() < 5e-4 * max(1.0, np.abs(want).max() / 50) and _rel(got, want) < 1e-5


def test_conv3d_all_ones_centre_27(dev):
    from ipp_amd import decon
    out = decon.conv3d_gpu(torch.ones((5, 5, 5), device=dev), torch.ones((3, 3, 3), device=dev)).cpu().numpy()
    assert abs(out[2, 2, 2] - 27.0) < 1e-4 and np.allclose(out, 27.0, atol=1e-4)  # edgetaper_3d_test.m:142-164


@pytest.mark.parametrize("boundary", [0, 1, 2])
@pytest.mark.parametrize("engine", [1, 2])
def test_convn_same_boundaries_and_engines(dev, boundary, engine):
    from ipp_amd import decon
    rng = np.random.default_rng(7)
    img = rng.random((14, 30, 41), dtype=np.float32)
    ker = rng.random((5, 7, 9), dtype=np.float32)
    got = decon.convn_same(_t(img, dev), _t(ker, dev), boundary=boundary, engine=engine).cpu().numpy()
    if boundary == 0:
        want = R.convn_same(img, ker)
    elif boundary == 1:
        want = R.conv3d_replicate(img, ker)
    else:
        from scipy import ndimage
        want = ndimage.convolve(img.astype(np.float64), ker.astype(np.float64), mode="wrap").astype(np.float32)
    assert _rel(got, want) < 2e-6 if engine == 1 else _rel(got, want) < 2e-5


def test_conv3d_errors(dev):
    from ipp_amd import capi, decon
    with pytest.raises(ValueError):
        decon.conv3d_gpu(torch.ones((5, 5), device=dev), torch.ones((3, 3, 3), device=dev))
    with pytest.raises(TypeError):
        decon.conv3d_gpu(torch.ones((5, 5, 5), device=dev, dtype=torch.float64), torch.ones((3, 3, 3), device=dev))
    a = torch.ones((4, 4, 4), device=dev)
    rc = capi.lib().mi_conv3d(0, None, a.data_ptr(), a.data_ptr(), a.data_ptr(), 4, 4, 4, 4, 4, 4, 0, 1)
    assert rc == -1 and "aliased" in capi.last_error()


# ------------------------------------------------------------------ gauss3d_gpu (R8)
@pytest.mark.parametrize("shape", [(32, 64, 32), (20, 33, 47), (37, 70, 130), (530, 12, 68),
                                   (9, 37, 520), (6, 70, 1156)])   # rows of >= 512 samples: 128-column tiles of the single pass, ragged in x and y
@pytest.mark.parametrize("sigma,ksize", [(2.5, None), ([1.5, 1.5, 2.5], [9, 11, 15]), ([0.5, 0.5, 2.5], None),
                                         (0.25, 3), (8, 51), ([0.5, 0.5, 2.5], [13, 13, 25])])
def test_gauss3d_gpu_matches_oracle(dev, shape, sigma, ksize):
    from ipp_amd import decon
    rng = np.random.default_rng(0)
    x = rng.random(shape, dtype=np.float32)
    t = _t(x, dev)
    out = decon.gauss3d_gpu(t, sigma, ksize)
    assert out.data_ptr() == t.data_ptr()  # destructive in place like the MEX (gauss3d_gpu.cu:289-293)
    assert np.abs(out.cpu().numpy() - R.gauss3d(x, sigma, ksize)).max() < 5e-5  # gauss3d_gpu_test.m:16


def test_gauss3d_errors(dev):
    from ipp_amd import capi, decon
    with pytest.raises(capi.MiError, match="MAX_KERNEL_SIZE"):
        decon.gauss3d_gpu(torch.ones((8, 8, 8), device=dev), 1.0, 53)
    with pytest.raises(ValueError):
        decon.gauss3d_gpu(torch.ones((8, 8, 8), device=dev), [1.0, 2.0])


# ------------------------------------------------------------------ edgetaper_3d (R6)
@pytest.mark.parametrize("shape,kshape", [((32, 64, 64), (7, 15, 15)), ((5, 6, 7), (3, 3, 3)), ((3, 3, 3), (3, 3, 3)),
                                          ((40, 50, 140), (21, 9, 9))])
def test_edgetaper_matches_oracle(dev, shape, kshape):
    from ipp_amd import decon
    rng = np.random.default_rng(42)
    bl = rng.random(shape, dtype=np.float32)
    psf = R.gaussian_psf(kshape, [k / 5.0 for k in kshape]) * 3.0  # un-normalised on purpose (edgetaper_3d.m:14)
    got = decon.edgetaper_3d(_t(bl, dev), _t(psf, dev)).cpu().numpy()
    want = R.edgetaper_3d(bl, psf)
    assert got.shape == bl.shape and np.abs(got - want).max() < 1e-5  # edgetaper_3d_test.m:4,40
    assert got.min() >= 0.0 and got.max() <= 1.0 + 1e-6               # :77-83


def test_edgetaper_rejects_negative_psf(dev):
    from ipp_amd import decon
    psf = -torch.ones((3, 3, 3), device=dev)
    with pytest.raises(AssertionError):
        decon.edgetaper_3d(torch.ones((8, 8, 8), device=dev), psf)


# ------------------------------------------------------------------ otf_gpu (K3) / ingest / pad
@pytest.mark.parametrize("F_xyz", [(25, 24, 20), (32, 32, 16), (21, 15, 9)])
def test_otf_gpu_matches_definition(dev, F_xyz):
    from ipp_amd import decon
    psf = R.gaussian_psf((9, 15, 15), (2, 3, 3))
    otf = decon.otf_gpu(_t(psf, dev), F_xyz).cpu().numpy()
    ref = R.otf_from_psf(psf, (F_xyz[2], F_xyz[1], F_xyz[0]))[:, :, : F_xyz[0] // 2 + 1]
    assert otf.shape == ref.shape
    assert np.abs(otf - ref).max() / np.abs(ref).max() < 2e-6  # otf_gpu_test.m:82


def test_u16_ingest_pad_crop_norm(dev):
    from ipp_amd import decon
    rng = np.random.default_rng(3)
    u = rng.integers(0, 65536, size=(7, 9, 13), dtype=np.uint16)
    got = decon.im2single(u).cpu().numpy()
    assert np.array_equal(got, R.u16_to_f32(u)) or np.abs(got - R.u16_to_f32(u)).max() < 1e-7
    a = _t(rng.random((6, 7, 9), dtype=np.float32), dev)
    p, pre, post = decon.pad_block_to_fft_shape(a, (12, 10, 9))
    want, wpre, wpost = R.pad_block_to_fft_shape(a.cpu().numpy(), (9, 10, 12))
    assert np.array_equal(p.cpu().numpy(), want) and pre == wpre[::-1] and post == wpost[::-1]
    assert torch.equal(decon.unpad_block(p, pre, post), a)
    assert decon.norm2(a) == pytest.approx(float(np.linalg.norm(a.cpu().numpy().astype(np.float64))), rel=1e-12)
    with pytest.raises(AssertionError):
        decon.pad_block_to_fft_shape(a, (4, 4, 4))
    assert [decon.next_fast_len(n) for n in (11, 2078, 512)] == [12, 2100, 512]


# ------------------------------------------------------------------ decon (R1-R5)
def _case(shape, kshape, sig, seed):
    psf = R.gaussian_psf(kshape, sig)
    return R.bead_volume(shape, seed=seed, psf=psf), psf


@pytest.mark.parametrize("engine", [1, 2])
@pytest.mark.parametrize("niter,lam,interval", [(5, 0.0, 0), (7, 0.0, 3), (7, 0.05, 2)])
def test_decon_spatial_matches_oracle(dev, engine, niter, lam, interval):
    from ipp_amd import decon
    vol, psf = _case((20, 36, 44), (7, 5, 5), (1.5, 1.0, 1.0), 11)
    want = R.decon_spatial(vol, psf, niter, lam, 0.0, interval)
    got = decon.decon(_t(vol, dev), decon.make_psf_struct(psf), niter, lam, 0.0, interval, 1, False, None, False,
                      engine=engine).cpu().numpy()
    assert_close(got, want)


@pytest.mark.parametrize("F_xyz", [None, (48, 40, 24)])
@pytest.mark.parametrize("niter,lam,interval", [(5, 0.0, 0), (7, 0.05, 2)])
def test_decon_fft_matches_oracle(dev, F_xyz, niter, lam, interval):
    from ipp_amd import decon
    vol, psf = _case((20, 36, 44), (7, 5, 5), (1.5, 1.0, 1.0), 12)
    Fz = vol.shape if F_xyz is None else (F_xyz[2], F_xyz[1], F_xyz[0])
    want = R.decon_fft(vol, psf, Fz, niter, lam, 0.0, interval)
    got = decon.decon(_t(vol, dev), psf, niter, lam, 0.0, interval, 1, True,
                      F_xyz if F_xyz is not None else (vol.shape[2], vol.shape[1], vol.shape[0]), False).cpu().numpy()
    assert got.shape == vol.shape
    assert_close(got, want)


# deconFFT_Wiener cuts the new PSF out of the far field of a spectral quotient (F{Y} conj F{X} / |F{X}|^2): the update is
# unstable -- the float64 oracle run with float32 transforms drifts from itself by ~100x per iteration (1e-5 after two
# iterations, 5 % after four) -- so whole runs are compared for niter = 2 and longer runs step by step: the oracle starts
# every iteration from the PSF the device estimated for it (device runs of niter = i return the PSF iteration i starts from),
# which holds the volume to the RL tolerance and each single PSF update to 2e-3 of the PSF's peak.
WIENER_F = [(64, 96, 32), (64, 64, 96)]


def _wiener_case():
    vol, psf = _case((20, 36, 44), (7, 5, 5), (1.5, 1.0, 1.0), 13)
    rng = np.random.default_rng(5)
    return rng.poisson(vol * 200 + 10).astype(np.float32), psf


@pytest.mark.parametrize("F_xyz", WIENER_F)
@pytest.mark.parametrize("lam,interval", [(0.0, 0), (0.05, 1), (0.0, 2)])
def test_decon_fft_wiener_two_iterations_match_oracle(dev, F_xyz, lam, interval):
    from ipp_amd import decon
    vol, psf = _wiener_case()
    Fz = (F_xyz[2], F_xyz[1], F_xyz[0])
    want, want_psf = R.decon_fft_wiener(vol, psf, Fz, 2, lam, 0.0, interval, return_psf=True)
    got, got_psf = decon.decon(_t(vol, dev), psf, 2, lam, 0.0, interval, 1, True, F_xyz, True, return_psf=True)
    got, got_psf = got.cpu().numpy(), got_psf.cpu().numpy()
    assert got.shape == vol.shape and got_psf.shape == psf.shape
    assert np.abs(got_psf - want_psf).max() <= 2e-3 * want_psf.max()
    assert abs(float(got_psf.sum()) - 1.0) < 1e-5 and got_psf.min() >= 0
    assert _rel(got, want) < 5 * REL


@pytest.mark.parametrize("F_xyz", WIENER_F)
@pytest.mark.parametrize("niter,interval", [(5, 0), (6, 3)])
def test_decon_fft_wiener_step_locked(dev, F_xyz, niter, interval):
    from ipp_amd import decon
    vol, psf = _wiener_case()
    Fz = (F_xyz[2], F_xyz[1], F_xyz[0])
    run = lambda n: decon.decon(_t(vol, dev), psf, n, 0.0, 0.0, interval, 1, True, F_xyz, True, return_psf=True)
    starts = {i: run(i)[1].cpu().numpy() for i in range(2, niter + 1)}    # PSF iteration i starts from
    got = run(niter)[0].cpu().numpy()
    trace = []
    want = R.decon_fft_wiener(vol, psf, Fz, niter, 0.0, 0.0, interval, forced_psfs=starts, trace=trace)
    # every oracle iteration started from the PSF the device used for it: the volume is held to the RL metric (1e-4 of the
    # maximum, relative L2 1e-5, point-wise 1e-4) like every other loop
    assert_close(got, want, what="step-locked Wiener volume:")
    assert len(trace) == niter - 1
    # The PSF update itself stays at 2e-3 of its peak: otf_new = F{Y} conj(F{X}) / max(|F{X}|^2, eps) (decon.m:283-290) divides
    # by the power spectrum of the current estimate, which is ~1e-10 of its DC value over most of the band -- there the quotient
    # amplifies the fp32 rounding of the two transforms by orders of magnitude -- and the new PSF is the real part of its inverse
    # transform cropped to the centre box, clamped and renormalised (decon.m:292-304): single-precision transforms on either side
    # (the float64 oracle run with float32 transforms drifts from itself just the same) differ there by 1e-3 of the peak.
    for i, est in enumerate(trace, start=2):                               # the oracle's own update from the same state
        assert np.abs(starts[i] - est).max() <= 2e-3 * est.max(), i


def test_decon_fft_wiener_stop_criterion(dev):
    # decon.m:310-317: no i > 1 guard, so a loose criterion stops after the first iteration already
    from ipp_amd import decon
    vol, psf = _wiener_case()
    _, n1 = decon.decon(_t(vol, dev), psf, 6, 0.0, 99.0, 0, 1, True, (64, 64, 32), True, return_iters=True)
    _, n2 = decon.decon(_t(vol, dev), psf, 3, 0.0, 1e-9, 0, 1, True, (64, 64, 32), True, return_iters=True)
    assert n1 == 1 and n2 == 3
    want = R.decon_fft_wiener(vol, psf, (32, 64, 64), 6, 0.0, 99.0, 0)
    got = decon.decon(_t(vol, dev), psf, 6, 0.0, 99.0, 0, 1, True, (64, 64, 32), True).cpu().numpy()
    assert _rel(got, want) < REL


def test_decon_fft_wiener_single_iteration_is_decon_fft(dev):
    # with one iteration there is no PSF update: deconFFT_Wiener == deconFFT, and the PSF comes back untouched
    from ipp_amd import decon
    vol, psf = _case((16, 32, 32), (5, 5, 5), (1.2, 1.0, 1.0), 14)
    a = decon.decon(_t(vol, dev), psf, 1, 0.0, 0.0, 0, 1, True, (32, 32, 16), False).cpu().numpy()
    b, p2 = decon.decon(_t(vol, dev), psf, 1, 0.0, 0.0, 0, 1, True, (32, 32, 16), True, return_psf=True)
    assert _rel(b.cpu().numpy(), a) < 1e-6
    assert np.array_equal(p2.cpu().numpy(), psf)


def test_decon_fft_semantics_on_direct_engine(dev):
    # the deconFFT placement quirk (even fft_shape -> one-voxel offset) reproduced by both engines
    from ipp_amd import decon
    vol, psf = _case((16, 24, 32), (5, 5, 7), (1.0, 1.0, 1.5), 13)
    F = (vol.shape[2], vol.shape[1], vol.shape[0])
    a = decon.decon(_t(vol, dev), psf, 4, 0.0, 0.0, 0, 1, True, F, False, engine=1).cpu().numpy()
    b = decon.decon(_t(vol, dev), psf, 4, 0.0, 0.0, 0, 1, True, F, False, engine=2).cpu().numpy()
    assert _rel(a, b) < REL
    assert_close(a, R.decon_fft(vol, psf, vol.shape, 4))
    assert_close(b, R.decon_fft(vol, psf, vol.shape, 4))


@pytest.mark.parametrize("engine,rocfft", [(0, False), (1, False), (2, False), (2, True)], ids=["auto", "direct", "native_fft", "rocfft"])
def test_decon_fft_psf_placed_by_a_named_grid(dev, engine, rocfft, monkeypatch):
    """mi_rl_options.psf_grid: a block on a larger grid than the reference's next_fast_len one, the PSF placed where the reference's
    grid puts it (odd 7-smooth extents: centre at index 0; the even extents of the hand-written transform alone: index -1).  Every
    engine against the oracle's restatement of the same rule, and -- the point of the option -- close to the reference grid's
    result where the default placement is far from it."""
    from ipp_amd import decon
    monkeypatch.setenv("MI_FFT_NATIVE_INFLATE", "100")
    if rocfft:
        monkeypatch.setenv("MI_FFT_ROCFFT", "1")
    vol, psf = _case((24, 26, 28), (7, 5, 5), (1.5, 1.0, 1.0), 3)
    ref_grid, big = (29, 27, 25), (32, 32, 32)                                       # [x y z]
    want = R.decon_fft(vol, psf, big[::-1], 4, regularize_interval=2, psf_grid_zyx=ref_grid[::-1])
    got = decon.decon(_t(vol, dev), psf, 4, 0.0, 0.0, 2, 1, True, big, False, engine=engine, psf_grid=ref_grid).cpu().numpy()
    assert_close(got, want)
    plain = decon.decon(_t(vol, dev), psf, 4, 0.0, 0.0, 2, 1, True, big, False, engine=engine).cpu().numpy()
    assert_close(plain, R.decon_fft(vol, psf, big[::-1], 4, regularize_interval=2))
    on_ref = R.decon_fft(vol, psf, ref_grid[::-1], 4, regularize_interval=2)
    core = (slice(7, -7), slice(5, -5), slice(5, -5))
    assert np.abs(got[core] - on_ref[core]).max() < 0.03 * on_ref[core].max() < np.abs(plain[core] - on_ref[core]).max()
    # naming the FFT shape itself is the default; the option belongs to deconFFT
    same = decon.decon(_t(vol, dev), psf, 4, 0.0, 0.0, 2, 1, True, big, False, engine=engine, psf_grid=big).cpu().numpy()
    assert np.array_equal(same, plain)
    with pytest.raises(ValueError, match="deconFFT only"):
        decon.decon(_t(vol, dev), psf, 1, 0.0, 0.0, 0, 1, False, None, False, psf_grid=big)


@pytest.mark.parametrize("kshape", [(6, 5, 4), (4, 4, 6), (7, 6, 5)], ids=["even_z_x", "all_even", "even_y"])
def test_decon_fft_named_grid_with_even_psf_extents(dev, kshape, monkeypatch):
    """psf_grid with PSFs of even extents (no centre sample: ifftshift(zero-pad-centre) lands sample k / 2 at index 0 whatever the
    parity of the grid, and the parity decides where the zero padding goes): device = oracle on the enlarged grid, for the
    hand-written pipeline, rocFFT and the direct engine."""
    from ipp_amd import decon
    monkeypatch.setenv("MI_FFT_NATIVE_INFLATE", "100")
    rng = np.random.default_rng(8)
    psf = rng.random(kshape, dtype=np.float32)
    psf /= psf.sum()
    vol = R.bead_volume((24, 26, 28), seed=9, psf=R.gaussian_psf((5, 5, 5), (1.0, 1.0, 1.0)))
    ref_grid, big = (29, 27, 25), (32, 32, 32)
    want = R.decon_fft(vol, psf, big[::-1], 3, psf_grid_zyx=ref_grid[::-1])
    for engine, rocfft in ((2, False), (2, True), (1, False)):
        if rocfft:
            monkeypatch.setenv("MI_FFT_ROCFFT", "1")
        else:
            monkeypatch.delenv("MI_FFT_ROCFFT", raising=False)
        got = decon.decon(_t(vol, dev), psf, 3, 0.0, 0.0, 0, 1, True, big, False, engine=engine, psf_grid=ref_grid).cpu().numpy()
        assert_close(got, want)


def test_decon_plan_is_keyed_by_the_psf_grid(dev, monkeypatch):
    from ipp_amd import decon
    monkeypatch.setenv("MI_FFT_NATIVE_INFLATE", "100")
    vol, psf = _case((24, 26, 28), (7, 5, 5), (1.5, 1.0, 1.0), 5)
    big = (32, 32, 32)
    plan = decon.DeconPlan(1)
    try:
        for grid in ((29, 27, 25), None, (29, 27, 25), (30, 27, 26)):
            got = decon.decon(_t(vol, dev), psf, 3, 0.0, 0.0, 0, 1, True, big, False, plan=plan, psf_grid=grid).cpu().numpy()
            assert_close(got, R.decon_fft(vol, psf, big[::-1], 3, psf_grid_zyx=None if grid is None else grid[::-1]))
    finally:
        plan.close()


def test_decon_stop_criterion_and_numpy_roundtrip(dev):
    from ipp_amd import decon
    vol, psf = _case((12, 16, 16), (5, 5, 5), (1, 1, 1), 4)
    want, it_want = R.decon_spatial(vol, psf, 50, stop_criterion=5.0, return_iters=True)
    got, it = decon.decon(vol, psf, 50, 0.0, 5.0, 0, 1, False, None, False, return_iters=True)
    assert isinstance(got, np.ndarray) and it == it_want
    assert_close(got, want)


def test_decon_config1_shape_parity(dev):
    # BASELINE config 1 (256x256x64, 9x9x15 Gaussian PSF, 10 iterations) on a 1/8 crop for the oracle's sake
    from ipp_amd import decon
    psf = R.gaussian_psf((15, 9, 9), (2.5, 1.5, 1.5))
    vol = R.bead_volume((32, 128, 128), seed=1234, psf=psf)
    want = R.decon_spatial(vol, psf, 10, 0.0, 0.0, 3)
    for engine in (1, 2):
        got = decon.decon(_t(vol, dev), psf, 10, 0.0, 0.0, 3, 1, False, None, False, engine=engine).cpu().numpy()
        assert_close(got, want)


def test_decon_errors(dev):
    from ipp_amd import capi, decon
    vol = torch.ones((8, 8, 8), device=dev)
    psf = np.ones((3, 3, 3), np.float32)
    with pytest.raises(ValueError):
        decon.decon(vol, psf, 1, 0, 0, 0, 1, False, None, True)  # adaptive needs use_fft (decwrap.py:216-217)
    with pytest.raises(capi.MiError, match="mi_fft_good_size"):
        decon.decon(vol, psf, 1, 0, 0, 0, 1, True, (10, 8, 8), True)  # deconFFT_Wiener: hand-written FFT extents only
    with pytest.raises(capi.MiError, match="cannot pad"):
        decon.decon(vol, psf, 1, 0, 0, 0, 1, True, (4, 8, 8), False)
    with pytest.raises(ValueError):
        decon.decon(vol, psf, 1, 0, 0, 0, 0, False, None, False)  # device 0 = the MATLAB CPU path


# ------------------------------------------------------------------ size-independent properties at scale
def test_properties_at_scale(dev):
    """256x256x64 (config 1 full size): linearity of the convolution, engines agree, RL keeps the volume
    non-negative and (for a normalised PSF with circular boundary) conserves the total flux."""
    from ipp_amd import decon
    g = torch.Generator(device="cpu").manual_seed(5)
    a = torch.rand((64, 256, 256), generator=g).to(dev)
    b = torch.rand((64, 256, 256), generator=g).to(dev)
    psf = _t(R.gaussian_psf((15, 9, 9), (2.5, 1.5, 1.5)), dev)
    ca, cb = decon.convn_same(a, psf, engine=1), decon.convn_same(b, psf, engine=1)
    cab = decon.convn_same(a + 2 * b, psf, engine=1)
    assert float((cab - (ca + 2 * cb)).abs().max()) < 2e-5
    assert float((decon.convn_same(a, psf, engine=2) - ca).abs().max()) < 2e-5
    x = a.clone()
    s0 = float(x.double().sum())
    out = decon.decon(x, psf, 5, 0.0, 0.0, 0, 1, True, (256, 256, 64), False, skip_edgetaper=True)
    assert float(out.min()) >= 0.0 and abs(float(out.double().sum()) - s0) / s0 < 1e-4


@pytest.mark.parametrize("engine", ["fft", "direct", "slabs"])
def test_edgetaper_fft_route_equals_direct_route(dev, engine, monkeypatch):
    """edgetaper's blur through the FFT engine on the replicate-padded volume, through six face slabs on circular FFT grids
    (the route of large blocks) and as shell-only direct convolution: same result within fp32 rounding."""
    from ipp_amd import decon
    monkeypatch.setenv("MI_EDGETAPER_ENGINE", engine)
    rng = np.random.default_rng(42)
    bl = rng.random((40, 50, 140), dtype=np.float32)
    psf = R.gaussian_psf((21, 9, 9), (4.0, 2.0, 2.0))
    got = decon.edgetaper_3d(_t(bl, dev), _t(psf, dev)).cpu().numpy()
    assert np.abs(got - R.edgetaper_3d(bl, psf)).max() < 1e-5
    # an asymmetric, un-normalised PSF and extents where the two slabs of an axis differ in thickness
    from tests.rl_util import asymmetric_psf
    bl2 = rng.random((45, 70, 100), dtype=np.float32)
    psf2 = asymmetric_psf((9, 5, 13), seed=2) * 2.5
    got2 = decon.edgetaper_3d(_t(bl2, dev), _t(psf2, dev)).cpu().numpy()
    assert np.abs(got2 - R.edgetaper_3d(bl2, psf2)).max() < 1e-5


def test_decon_plan_is_bit_identical_and_rebuilds(dev):
    """mi_decon_plan keeps the RL context and the taper engine between blocks: same bits as decon without a plan, for repeated
    blocks, a changed PSF, a changed shape and the spatial flavour."""
    from ipp_amd import decon
    vol, psf = _case((20, 36, 44), (7, 5, 5), (1.5, 1.0, 1.0), 21)
    vol2 = np.ascontiguousarray(vol[::-1])
    psf2 = np.ascontiguousarray(psf[::-1] * 0.5 + psf * 0.5)
    F = (64, 64, 32)
    with decon.DeconPlan(1) as plan:
        for v, p_, fshape, use_fft in ((vol, psf, F, True), (vol2, psf, F, True), (vol, psf2, F, True), (vol[:, :32, :40], psf, F, True),
                                       (vol, psf, None, False), (vol2, decon.make_psf_struct(psf), None, False), (vol, psf, F, True)):
            want = decon.decon(_t(v, dev), p_, 4, 0.0, 0.0, 2, 1, use_fft, fshape, False).cpu().numpy()
            got = decon.decon(_t(v, dev), p_, 4, 0.0, 0.0, 2, 1, use_fft, fshape, False, plan=plan).cpu().numpy()
            assert np.array_equal(got, want)
        # the adaptive variant passes through
        a = decon.decon(_t(vol, dev), psf, 2, 0.0, 0.0, 0, 1, True, F, True).cpu().numpy()
        b = decon.decon(_t(vol, dev), psf, 2, 0.0, 0.0, 0, 1, True, F, True, plan=plan).cpu().numpy()
        assert np.array_equal(a, b)


# ------------------------------------------------------------------ separable fast path of the direct engine
@pytest.mark.parametrize("boundary", [0, 1, 2])
@pytest.mark.parametrize("kshape", [(7, 5, 9), (4, 6, 5)])
def test_separable_psf_takes_three_1d_passes(dev, boundary, kshape, monkeypatch):
    """A rank-1 PSF (Gaussian, BASELINE config 1) runs as three 1-D convolutions on the direct engine; same result as the dense
    tap loop (MI_NO_SEPARABLE=1) and as the oracle; a PSF that is not an outer product keeps the dense loop."""
    from ipp_amd import decon
    from tests.rl_util import asymmetric_psf
    shape = (20, 36, 44)
    psf = R.gaussian_psf(kshape, (1.5, 1.0, 2.0))
    psf = (psf * np.linspace(0.7, 1.3, kshape[2])[None, None, :]).astype(np.float32)   # still rank 1, not symmetric
    vol = R.bead_volume(shape, seed=5, psf=R.gaussian_psf((5, 5, 5), (1, 1, 1)))
    inv = R.flip3(psf) if boundary != 2 else None

    single = []

    def run(p, pinv):
        ctx = decon.RLContext(shape, p, pinv, boundary=boundary, engine=1, device=dev)
        bl = _t(vol, dev)
        ratio = torch.empty_like(bl)
        ctx.iterate(bl, ratio, 3)
        single.append(ctx.separable_single_pass)
        return bl.cpu().numpy(), ctx.separable

    got, sep = run(psf, inv)
    assert sep and single[-1]                    # one pass over the volume (sep3d.hip)
    monkeypatch.setenv("MI_NO_SEP_SINGLE", "1")
    three, sep3 = run(psf, inv)                  # three launches of the dense kernel with 1-D tap tables
    monkeypatch.delenv("MI_NO_SEP_SINGLE")
    assert sep3 and not single[-1] and _rel(got, three) < 5e-6
    monkeypatch.setenv("MI_NO_SEPARABLE", "1")
    dense, sep_off = run(psf, inv)
    monkeypatch.delenv("MI_NO_SEPARABLE")
    assert not sep_off and _rel(got, dense) < 5e-6
    if boundary == 0:
        assert_close(got, R.decon_spatial(vol, psf, 3, skip_edgetaper=True))
    elif boundary == 2:
        assert_close(got, R.decon_fft(vol, psf, shape, 3, skip_edgetaper=True))
    _, sep_asym = run(asymmetric_psf(kshape, seed=1), R.flip3(asymmetric_psf(kshape, seed=1)) if boundary != 2 else None)
    assert not sep_asym


@pytest.mark.parametrize("boundary", [0, 2])
@pytest.mark.parametrize("kshape", [(19, 5, 7), (5, 33, 9), (35, 3, 5), (32, 4, 3)])
def test_separable_single_pass_variants(dev, boundary, kshape, monkeypatch):
    """Every build of the single-pass separable kernel (sep3d.hip): z windows of 17 - 32 taps in registers (19; 32 with an even
    window offset), the patch of a long y kernel in six register quads per thread (33 rows of taps), the LDS-ring kernel that
    keeps z windows beyond 32 taps (35) -- against the three-launch route and the dense loop of the same engine, and (first case)
    the oracle's loop."""
    from ipp_amd import decon
    shape = (44, 50, 72)
    psf = R.gaussian_psf(kshape, (kshape[0] / 5.0, kshape[1] / 5.0, kshape[2] / 5.0))
    psf = (psf * np.linspace(0.8, 1.2, kshape[0])[:, None, None]).astype(np.float32)       # rank 1, not symmetric along z
    vol = R.bead_volume(shape, seed=25, psf=R.gaussian_psf((3, 3, 3), (1, 1, 1)))
    inv = R.flip3(psf) if boundary != 2 else None

    def run(iters=2):
        ctx = decon.RLContext(shape, psf, inv, boundary=boundary, engine=1, device=dev)
        bl = _t(vol, dev)
        ctx.iterate(bl, torch.empty_like(bl), iters)
        return bl.cpu().numpy(), ctx.separable_single_pass

    got, single = run()
    assert single
    monkeypatch.setenv("MI_NO_SEP_SINGLE", "1")
    three, single3 = run()
    monkeypatch.delenv("MI_NO_SEP_SINGLE")
    assert not single3 and _rel(got, three) < 5e-6
    if kshape == (19, 5, 7):
        want = R.decon_fft(vol, psf, shape, 2, skip_edgetaper=True) if boundary == 2 else R.decon_spatial(vol, psf, 2, skip_edgetaper=True)
        assert_close(got, want)


@pytest.mark.parametrize("boundary", [0, 1, 2])
def test_separable_single_pass_edges_and_regularised_update(dev, boundary):
    """The single-pass separable kernel where its tiles are ragged: extents that are no multiples of the 64 x 16 tile, fewer planes
    than taps along z, a z chunk boundary inside the volume, the regularised update epilogue (lambda > 0) -- against the dense
    loop of the same engine and the oracle's whole deconSpatial / deconFFT run."""
    from ipp_amd import decon
    shape, kshape = ((9, 23, 68), (11, 5, 7)) if boundary != 2 else ((12, 23, 68), (11, 5, 7))   # (circular: the PSF fits the shape)
    psf = R.gaussian_psf(kshape, (2.0, 1.0, 1.5))
    vol = R.bead_volume(shape, seed=15, psf=R.gaussian_psf((3, 3, 3), (1, 1, 1)))
    ctx = decon.RLContext(shape, psf, None, boundary=boundary, engine=1, device=dev)
    assert ctx.separable_single_pass
    if boundary == 1:
        a = decon.conv3d_gpu(vol, psf)                                   # the dense kernel (conv3d_gpu.cu:68-99)
        bl = _t(vol, dev)
        ones = torch.ones_like(bl)
        ctx.forward_ratio(bl, ones)                                      # ones <- bl ./ max(conv(bl), eps)
        got = (bl / ones).cpu().numpy()
        assert_close(got, R.conv3d_replicate(vol, psf), what="single-pass conv, replicate rule:")
        assert _rel(got, a) < 5e-6
        return
    use_fft = boundary == 2
    F = (shape[2], shape[1], shape[0])
    want = (R.decon_fft(vol, psf, shape, 6, 0.05, 0.0, 2) if use_fft else R.decon_spatial(vol, psf, 6, 0.05, 0.0, 2))
    got = decon.decon(_t(vol, dev), decon.make_psf_struct(psf) if not use_fft else psf, 6, 0.05, 0.0, 2, 1, use_fft, F if use_fft else None,
                      False, engine=1).cpu().numpy()
    assert_close(got, want)
