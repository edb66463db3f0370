"""CPU: pins the numpy RL oracle with the known-answer / property checks of the reference's own MATLAB test
scripts (no stored vectors exist for this path: 'parity unpinned')."""
import numpy as np
import pytest
from scipy import ndimage, signal

from oracle import rl_oracle as R


def test_all_ones_centre_is_27():
    # edgetaper_3d_test.m:142-164: ones(5,5,5) conv ones(3,3,3) -> centre 27
    out = R.conv3d_replicate(np.ones((5, 5, 5), np.float32), np.ones((3, 3, 3), np.float32))
    assert abs(out[2, 2, 2] - 27.0) < 1e-4
    assert np.allclose(out, 27.0)  # replicate boundary keeps it 27 everywhere


def test_conv3d_equals_replicate_pad_plus_valid():
    # edgetaper_3d_test.m:107-140: conv3d_gpu == convn(padarray(I, floor(k/2), 'replicate'), K, 'valid')
    rng = np.random.default_rng(0)
    img = rng.random((22, 25, 24), dtype=np.float32)
    ker = rng.random((3, 5, 7), dtype=np.float32)
    pad = [(k // 2, k // 2) for k in ker.shape]
    ref = signal.convolve(np.pad(img, pad, mode="edge").astype(np.float64), ker.astype(np.float64), mode="valid")
    out = R.conv3d_replicate(img, ker)
    assert np.abs(out - ref).max() < 5e-4 and np.abs(out - ref).mean() < 1e-4
    assert np.abs(R.conv3d_replicate_loops(img, ker) - out).max() < 1e-5


def test_convn_same_is_zero_boundary_central_part():
    rng = np.random.default_rng(1)
    a = rng.random((9, 8, 11), dtype=np.float32)
    h = rng.random((3, 5, 7), dtype=np.float32)
    ref = signal.convolve(a.astype(np.float64), h.astype(np.float64), mode="same")
    assert np.abs(R.convn_same(a, h) - ref).max() < 1e-5


def test_make_taper_literal():
    # make_taper.m:19-35: ramp of w+1, plateau, mirrored ramp without its last sample, cut to n
    t = R.make_taper(20, 4)
    assert t.shape == (20,)
    assert np.allclose(t[:5], [0, 0.25, 0.5, 0.75, 1.0])
    assert np.all(t[5:17] == 1.0)
    assert np.allclose(t[17:], [0.75, 0.5, 0.25])  # final 0 is cut: not symmetric
    assert np.all(R.make_taper(5, 0) == 1) and R.make_taper(7, 100).shape == (7,)
    assert R.make_taper(6, 8).tolist() == pytest.approx([0, 1 / 3, 2 / 3, 1, 2 / 3, 1 / 3])
    assert R.taper_widths((61, 31, 31)) == [31, 16, 16] and R.taper_widths((15, 9, 9)) == [8, 8, 8]


def test_edgetaper_range_size_and_interior():
    # edgetaper_3d_test.m:77-105: output within [0,1], size preserved; mask == 1 leaves the block untouched
    rng = np.random.default_rng(42)
    bl = rng.random((32, 64, 64), dtype=np.float32)
    psf = R.gaussian_psf((7, 15, 15), (1.5, 3.0, 3.0))
    out = R.edgetaper_3d(bl, psf)
    assert out.shape == bl.shape and out.min() >= 0.0 and out.max() <= 1.0 + 1e-6
    assert np.array_equal(out[8:25, 8:57, 8:57], bl[8:25, 8:57, 8:57])
    for shape, k in [((5, 6, 7), (3, 3, 3)), ((3, 3, 3), (3, 3, 3))]:
        o = R.edgetaper_3d(rng.random(shape, dtype=np.float32), np.ones(k, np.float32))
        assert o.shape == shape and np.all(np.isfinite(o))


@pytest.mark.parametrize("sigma,ksize", [(2.5, None), ([1.5, 1.5, 2.5], [9, 11, 15]), ([0.5, 0.5, 2.5], None),
                                         (0.25, 3), (8, 51)])
def test_gauss3d_matches_spatial_replicate_gaussian(sigma, ksize):
    # gauss3d_gpu_test.m:12-16,51-113: vs imgaussfilt3(..., 'Padding','replicate','FilterDomain','spatial') < 5e-5
    rng = np.random.default_rng(0)
    x = rng.random((32, 64, 32), dtype=np.float32)
    out = R.gauss3d(x, sigma, ksize)
    sig = [sigma] * 3 if np.isscalar(sigma) else sigma
    ks = R.default_ksize(sig) if ksize is None else ([ksize] * 3 if np.isscalar(ksize) else ksize)
    ref = x.astype(np.float64)
    for ref_axis in range(3):
        r = ks[ref_axis] // 2
        i = np.arange(-r, r + 1)
        w = np.exp(-0.5 * i * i / (sig[ref_axis] ** 2))
        ref = ndimage.correlate1d(ref, w / w.sum(), axis=2 - ref_axis, mode="nearest")
    assert np.abs(out - ref).max() < 5e-5


def test_gaussian_taps_half_sigma_is_five_taps():
    assert R.default_ksize([0.5, 0.5, 0.5]) == [5, 5, 5]  # gauss3d_gpu.cu:259-260
    t = R.gaussian_taps(0.5, 5)
    assert t.dtype == np.float32 and abs(float(t.sum()) - 1) < 1e-6 and t[2] > 0.78


def test_otf_matches_reference_definition():
    # supplements/otf_gpu_test.m:8-11,44-48,82: fftn(ifftshift(pad(psf))) within 2e-6 relative
    psf = R.gaussian_psf((9, 15, 15), (2, 3, 3))
    F = (20, 24, 25)
    otf = R.otf_from_psf(psf, F)
    pad, _, _ = R.pad_block_to_fft_shape(psf, F)
    ref = np.fft.fftn(np.fft.ifftshift(pad))
    assert np.abs(otf - ref).max() / np.abs(ref).max() < 2e-6
    # odd shape: centred -> real, positive DC; DC == sum(psf)
    assert abs(otf[0, 0, 0] - psf.sum()) < 1e-6


def test_otf_placement_named_by_another_grid():
    # mi_rl_options.psf_grid: the grid whose parity places the PSF.  Naming the FFT shape itself is the reference's two steps
    # exactly; an even grid named on an odd one moves an odd PSF by one sample (decon.m:131-133 on an even extent), and only that
    psf = R.gaussian_psf((7, 5, 5), (1.5, 1.0, 1.0))
    for F in ((16, 12, 10), (15, 13, 11), (16, 13, 10)):
        assert np.array_equal(R.otf_from_psf(psf, F), R.otf_from_psf(psf, F, F))
    F, G = (15, 13, 11), (16, 14, 12)
    a = np.real(np.fft.ifftn(R.otf_from_psf(psf, F, G)))
    b = np.real(np.fft.ifftn(R.otf_from_psf(psf, F)))
    assert np.abs(a - np.roll(b, (-1, -1, -1), (0, 1, 2))).max() < 1e-12 and np.abs(a - b).max() > 1e-3
    # the same placement on two grids: the spectra agree wherever both sample the same frequency (here: the z axis doubled)
    F2 = (30, 13, 11)
    c = R.otf_from_psf(psf, F2, F)
    assert np.abs(c[::2] - R.otf_from_psf(psf, F)).max() < 1e-12
    # a block on a larger grid with the smaller grid's placement: the same deconvolution up to the wider zero margin
    vol = R.bead_volume((24, 26, 28), seed=3, psf=psf)
    ref = R.decon_fft(vol, psf, (25, 27, 29), 3)
    same = R.decon_fft(vol, psf, (32, 32, 32), 3, psf_grid_zyx=(25, 27, 29))
    off = R.decon_fft(vol, psf, (32, 32, 32), 3)
    core = (slice(7, -7), slice(5, -5), slice(5, -5))
    assert np.abs(same[core] - ref[core]).max() < 0.02 * ref[core].max() < np.abs(off[core] - ref[core]).max()


def test_fft_step_equals_composition_of_fft_convs():
    # mex_incubator/deconFFT_test.m:15,81-87: one fused RL step vs explicit FFT convolutions < 2e-4
    rng = np.random.default_rng(5)
    bl = (rng.random((12, 20, 18)) + 0.1).astype(np.float32)
    psf = R.gaussian_psf((9, 15, 15), (3, 3, 3))
    out = R.decon_fft(bl, psf, bl.shape, 1)
    x = R.edgetaper_3d(bl, psf)
    otf = R.otf_from_psf(psf, bl.shape)
    c = np.real(np.fft.ifftn(np.fft.fftn(x) * otf))
    r = x / np.maximum(c, R.EPS_SINGLE)
    a = np.real(np.fft.ifftn(np.fft.fftn(r) * np.conj(otf)))
    assert np.abs(out - np.abs(x * a)).max() / np.abs(out).max() < 2e-4


def test_spatial_and_fft_variants_agree_away_from_borders():
    psf = R.gaussian_psf((5, 5, 5), (1.0, 1.0, 1.0))
    vol = R.bead_volume((21, 31, 33), seed=2, psf=psf)  # odd shape: deconFFT placement is exactly centred
    a = R.decon_spatial(vol, psf, 3)
    b = R.decon_fft(vol, psf, vol.shape, 3)
    core = (slice(8, -8),) * 3
    assert np.abs(a[core] - b[core]).max() / np.abs(a[core]).max() < 1e-3


def test_wiener_variant_structure():
    # decon.m:206-321 restated: one iteration has no PSF update (== deconFFT); the refined PSF is the clamped, renormalised
    # centre box of real(ifftn(F{Y} conj F{X} / max(|F{X}|^2, eps))); forcing the run's own PSFs reproduces it exactly
    rng = np.random.default_rng(9)
    psf = R.gaussian_psf((5, 5, 7), (1.0, 1.0, 1.5))
    vol = rng.poisson(R.bead_volume((12, 20, 24), seed=3, psf=psf) * 100 + 5).astype(np.float32)
    F = (16, 24, 32)
    one, p1 = R.decon_fft_wiener(vol, psf, F, 1, return_psf=True)
    assert np.array_equal(one, R.decon_fft(vol, psf, F, 1)) and np.array_equal(p1, psf)
    trace = []
    out, p = R.decon_fft_wiener(vol, psf, F, 4, return_psf=True, trace=trace)
    assert len(trace) == 3 and np.array_equal(trace[-1], p)
    assert p.shape == psf.shape and p.min() >= 0 and abs(float(p.sum()) - 1) < 1e-6
    # first update by hand from the state after iteration 1
    x0, _, _ = R.pad_block_to_fft_shape(R.edgetaper_3d(vol, psf), F)
    x1, _, _ = R.pad_block_to_fft_shape(R.decon_fft(vol, psf, F, 1), F)
    fy, fx = np.fft.fftn(x0.astype(np.float64)), np.fft.fftn(x1.astype(np.float64))
    full = np.real(np.fft.ifftn(fy * np.conj(fx) / np.maximum(np.abs(fx) ** 2, R.EPS_SINGLE)))
    c = [(f - k) // 2 for f, k in zip(F, psf.shape)]
    box = np.maximum(full[c[0]:c[0] + 5, c[1]:c[1] + 5, c[2]:c[2] + 7], 0)
    assert np.abs(box / box.sum() - trace[0]).max() < 1e-4 * trace[0].max()
    again = R.decon_fft_wiener(vol, psf, F, 4, forced_psfs={i + 2: t for i, t in enumerate(trace)})
    assert np.array_equal(again, out)
    # quirks: no i > 1 guard on the stop test (:311-317); Tikhonov needs i < niter (:273)
    assert np.array_equal(R.decon_fft_wiener(vol, psf, F, 5, 0.0, 99.0), one)
    assert np.array_equal(R.decon_fft_wiener(vol, psf, F, 1, 0.2, 0.0, 1), R.decon_fft_wiener(vol, psf, F, 1, 0.0, 0.0, 1))


def test_regularisation_schedule_and_stop():
    # decon.m:54-55: i>1, i<niter, mod(i,interval)==0, 0<interval<niter
    assert [i for i in range(1, 10) if R.is_regularization_time(i, 9, 3)] == [3, 6]
    assert not any(R.is_regularization_time(i, 3, 3) for i in range(1, 4))
    assert not any(R.is_regularization_time(i, 6, 0) for i in range(1, 7))
    psf = R.gaussian_psf((5, 5, 5), (1, 1, 1))
    vol = R.bead_volume((12, 16, 16), seed=4, psf=psf)
    out, it = R.decon_spatial(vol, psf, 50, stop_criterion=5.0, return_iters=True)
    assert 2 <= it < 50 and np.all(np.isfinite(out)) and out.min() >= 0
    reg = R.decon_spatial(vol, psf, 6, lam=0.05, regularize_interval=2)
    assert reg.shape == vol.shape and np.all(np.isfinite(reg))


def test_split_stack_and_fast_len():
    p1, p2 = R.split_stack((10, 7, 5), (4, 4, 3), (3, 2, 2))
    assert p1.shape == (12, 3) and p1[0].tolist() == [1, 1, 1] and p2[0].tolist() == [4, 4, 3]
    assert p1[1].tolist() == [5, 1, 1] and p2[2].tolist() == [10, 4, 3] and p2[-1].tolist() == [10, 7, 5]
    assert [R.next_fast_len(n) for n in (1, 11, 13, 97, 2078, 512)] == [1, 12, 14, 98, 2100, 512]


def test_pad_unpad_roundtrip():
    a = np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4)
    p, pre, post = R.pad_block_to_fft_shape(a, (5, 6, 4))
    assert p.shape == (5, 6, 4) and pre == [1, 1, 0] and post == [2, 2, 0]
    assert np.array_equal(R.unpad_block(p, pre, post), a)
    assert R.u16_to_f32(np.array([0, 65535], np.uint16)).tolist() == [0.0, 1.0]


def test_prctile_is_matlab_hazen_rule():
    """prctile (LsDeconv.m:1301): sample i of n sorted values is the 100 (i - 0.5) / n percentile -- numpy's 'hazen' method;
    MATLAB's documented example: prctile([1 2 3 4 5], 50) = 3, the 10th percentile of 1..5 is 1 (clamped below 100 * 0.5 / 5)."""
    rng = np.random.default_rng(0)
    x = rng.normal(size=1001).astype(np.float32)
    for p in (0.0, 0.01, 10.0, 33.3, 50.0, 99.99, 100.0):
        assert R.prctile(x, [p])[0] == pytest.approx(np.percentile(x.astype(np.float64), p, method="hazen"), rel=1e-6, abs=1e-7)
    assert R.prctile([1, 2, 3, 4, 5], [50, 10, 30, 100]) == [3.0, 1.0, 2.0, 5.0]
    assert np.isnan(R.prctile([np.nan], [50])[0])
    assert R.prctile([1.0, np.nan, 3.0], [50])[0] == 2.0


def test_rescale_block_literal_values():
    """load_slab_lz4.cpp:134-157: linear branch when dmin <= 0, min-max branch otherwise; val -= ampl; round half away;
    clamp [0, scal]."""
    x = np.array([0.0, 0.5, 1.0, 2.0, -1.0], np.float32)
    # dmin = 0: val * (255 * 1 / 1) - 1 -> -1, 126.5 -> 127, 254, 509 -> 255, -256 -> 0
    assert R.rescale_block(x, 255, 1.0, 0.0, 1.0, np.uint8).tolist() == [0, 127, 254, 255, 0]
    # dmin = 0.5, dmax = 1.5: (val - 0.5) * 65535 - 1
    assert R.rescale_block(x, 65535, 1.0, 0.5, 1.5, np.uint16).tolist() == [0, 0, 32767, 65535, 0]


def test_float32_timing_leg_equals_decon_fft():
    # bench.py's cpu_baseline times decon_fft_f32 (scipy.fft, complex64, all cores): same loop as decon_fft
    psf = R.gaussian_psf((7, 5, 5), (1.5, 1.0, 1.0))
    psf = (psf * np.linspace(0.5, 1.5, psf.shape[2])[None, None, :]).astype(np.float32)   # asymmetric on purpose
    vol = R.bead_volume((20, 36, 44), seed=3, psf=psf)
    want = R.decon_fft(vol, psf, vol.shape, 4, skip_edgetaper=True)
    got = R.decon_fft_f32(vol, R.otf_half_f32(psf, vol.shape, workers=2), 4, workers=2)
    assert got.dtype == np.float32 and np.abs(got - want).max() <= 1e-4 * np.abs(want).max()


def test_even_kernel_centres():
    # convn(a, h, 'same') = full[ceil((k-1)/2):...] -> centre k/2; conv3d_gpu.cu:77-98 -> centre (k-1)/2 (literal loops)
    from scipy import signal
    rng = np.random.default_rng(0)
    a = rng.random((6, 7, 9)).astype(np.float32)
    h = rng.random((4, 2, 6)).astype(np.float32)
    full = signal.convolve(a.astype(np.float64), h.astype(np.float64), "full")
    c = [k // 2 for k in h.shape]
    same = full[c[0]:c[0] + 6, c[1]:c[1] + 7, c[2]:c[2] + 9]
    assert np.abs(R.convn_same(a, h) - same).max() < 1e-5
    assert np.array_equal(np.convolve([1., 2, 3, 4], [1., 1])[1:5], [3., 5, 7, 4])   # MATLAB: conv([1 2 3 4],[1 1],'same')
    assert np.abs(R.conv3d_replicate(a, h) - R.conv3d_replicate_loops(a, h)).max() < 1e-5
    h5 = rng.random((3, 5, 4)).astype(np.float32)
    assert np.abs(R.conv3d_replicate(a, h5) - R.conv3d_replicate_loops(a, h5)).max() < 1e-5
