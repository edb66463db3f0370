"""GPU: the hand-written FFT convolution pipeline (fft_native.hip: power-of-two shapes) against the oracle, against
scipy's circular convolution and against the rocFFT route of the same engine (MI_FFT_ROCFFT=1)."""
import os

import numpy as np
import pytest
import torch
from scipy import ndimage

from oracle import rl_oracle as R
from tests.rl_util import assert_close

pytestmark = pytest.mark.gpu

SHAPES = [(8, 8, 16), (16, 32, 64), (32, 64, 128), (64, 16, 256), (8, 128, 32), (128, 8, 16), (16, 16, 1024),
          (8, 96, 32), (16, 288, 64), (8, 192, 16), (8, 576, 16), (16, 1152, 32),  # y = 3 * 2^a, 9 * 2^a: radix-3/9 stage
          (8, 16, 192), (16, 8, 576), (8, 32, 384), (8, 8, 1152),                  # x/2 = 3 * 2^a, 9 * 2^a
          (8, 160, 16), (16, 320, 32), (8, 640, 16), (8, 1280, 16),                # y = 5 * 2^a (y only: the rows of a slab rank)
          (96, 16, 32), (288, 8, 16), (192, 16, 64), (576, 8, 16),                 # z = 3 * 2^a, 9 * 2^a
          (96, 96, 192), (288, 192, 576), (96, 288, 64)]                           # mixed


def _rel(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / np.abs(b).max())


@pytest.mark.parametrize("shape", SHAPES)
def test_circular_conv_native_vs_scipy_and_rocfft(dev, shape):
    from ipp_amd import decon
    rng = np.random.default_rng(sum(shape))
    img = rng.random(shape, dtype=np.float32)
    ker = rng.random((3, 5, 7), dtype=np.float32)
    want = ndimage.convolve(img.astype(np.float64), ker.astype(np.float64), mode="wrap")
    got = decon.convn_same(torch.from_numpy(img).to(dev), torch.from_numpy(ker).to(dev), boundary=2, engine=2).cpu().numpy()
    assert _rel(got, want) < 2e-5
    os.environ["MI_FFT_ROCFFT"] = "1"
    try:
        ref = decon.convn_same(torch.from_numpy(img).to(dev), torch.from_numpy(ker).to(dev), boundary=2, engine=2).cpu().numpy()
    finally:
        del os.environ["MI_FFT_ROCFFT"]
    assert _rel(got, ref) < 2e-5


@pytest.mark.parametrize("shape", [(16, 32, 64), (32, 16, 128), (8, 64, 32), (8, 288, 32), (16, 96, 16), (96, 32, 192), (8, 96, 576),
                                   (16, 320, 32), (64, 160, 32)])
@pytest.mark.parametrize("niter,lam,interval", [(4, 0.0, 0), (6, 0.05, 2), (7, 0.0, 3)])
def test_decon_fft_native_matches_oracle(dev, shape, niter, lam, interval):
    from ipp_amd import decon
    psf = R.gaussian_psf((5, 7, 5), (1.0, 1.5, 1.0))
    vol = R.bead_volume(shape, seed=17, psf=psf)
    want = R.decon_fft(vol, psf, vol.shape, niter, lam, 0.0, interval)
    got = decon.decon(torch.from_numpy(vol).to(dev), psf, niter, lam, 0.0, interval, 1, True,
                      (shape[2], shape[1], shape[0]), False).cpu().numpy()
    assert_close(got, want)


def test_rocfft_engine_never_silently_wrong_beside_another_plan(dev, monkeypatch):
    """Live rocFFT plans are not independent in ROCm 7.2: beside the plans of a 256 x 16 x 64 grid a new engine on 32 x 128 x 8 transforms
    wrongly (profiles/r05_rocfft_coexistence.txt).  Every engine on the rocFFT route checks its own transforms at creation (a few OTF
    bins against the PSF's direct sum, the inverse against the placed PSF): the context is either refused with MI_ERR_FFT or right."""
    import gc
    from ipp_amd import capi, decon
    monkeypatch.setenv("MI_FFT_ROCFFT", "1")
    rng = np.random.default_rng(1)
    ker = rng.random((3, 5, 7), dtype=np.float32)
    img = rng.random((8, 128, 32), dtype=np.float32) + 0.5

    def conv(ctx):
        a = torch.from_numpy(img).to(dev)
        r = torch.empty_like(a)
        ctx.forward_ratio(a, r)
        return (a / r).cpu().numpy()

    def make(shape):
        return decon.RLContext(shape, ker, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)

    alone = make(img.shape)
    want = conv(alone)
    alone.close()
    del alone
    gc.collect()
    other = make((64, 16, 256))
    try:
        beside = make(img.shape)
    except capi.MiError as e:
        assert e.code == capi.MI_ERR_FFT and "rocFFT" in str(e) and "r05_rocfft_coexistence" in str(e)
    else:
        assert _rel(conv(beside), want) < 2e-5
        beside.close()
    other.close()
    del other
    gc.collect()
    again = make(img.shape)                       # with the other plans gone the same shape is served, and right
    assert _rel(conv(again), want) < 2e-5
    again.close()


def test_adjoint_is_exact_transpose(dev):
    """<conv(a), b> == <a, conv_adj(b)> for the circular operator pair (forward = OTF, adjoint = conj OTF)."""
    from ipp_amd import capi, decon
    shape = (16, 32, 64)
    psf = R.gaussian_psf((5, 7, 9), (1.0, 1.5, 2.0))
    ctx = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
    g = torch.Generator().manual_seed(3)
    a = (torch.rand(shape, generator=g) + 0.5).to(dev)
    b = (torch.rand(shape, generator=g) + 0.5).to(dev)
    ra = torch.empty_like(a)
    ctx.forward_ratio(a, ra)          # ra = a / conv(a)  ->  conv(a) = a / ra
    conv_a = a / ra
    ones = torch.ones_like(a)
    ctx.adjoint_update(b, ones)       # ones <- |1 * conv_adj(b)| = conv_adj(b) for positive data
    lhs = float((conv_a.double() * b.double()).sum())
    rhs = float((a.double() * ones.double()).sum())
    assert abs(lhs - rhs) / abs(lhs) < 1e-5


@pytest.mark.parametrize("shape", [(16, 32, 64), (8, 16, 256), (96, 32, 192)])
def test_fused_iterations_equal_unfused_and_direct_engine(dev, shape):
    """mi_rl_iterate: the fused 8-pass iteration of the native pipeline vs the two half-steps vs the direct engine."""
    from ipp_amd import capi, decon
    psf = R.gaussian_psf((5, 5, 7), (1.0, 1.0, 1.5))
    vol = torch.from_numpy(R.bead_volume(shape, seed=9, psf=psf)).to(dev)
    fft = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
    direct = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_DIRECT, device=dev)
    a, b, c = vol.clone(), vol.clone(), vol.clone()
    ratio = torch.empty_like(vol)
    fft.iterate(a, None, 4)                      # fused, no ratio scratch needed
    for _ in range(4):
        fft.forward_ratio(b, ratio)
        fft.adjoint_update(ratio, b)
    direct.iterate(c, ratio, 4)
    want = R.decon_fft(vol.cpu().numpy(), psf, shape, 4, skip_edgetaper=True)
    assert _rel(a.cpu().numpy(), b.cpu().numpy()) < 1e-5
    assert _rel(a.cpu().numpy(), c.cpu().numpy()) < 1e-4
    assert_close(a.cpu().numpy(), want)
    assert_close(c.cpu().numpy(), want)
    with pytest.raises(capi.MiError, match="ratio scratch"):
        direct.iterate(c, None, 1)


PADDED = [((13, 37, 50), (3, 5, 7)), ((40, 61, 90), (9, 11, 13)), ((7, 130, 33), (4, 6, 8)), ((100, 100, 100), (5, 31, 31))]


@pytest.mark.parametrize("shape,kshape", PADDED)
@pytest.mark.parametrize("boundary", [0, 1])
def test_padded_conv_through_native_pipeline(dev, shape, kshape, boundary, monkeypatch):
    """Zero / replicate boundary 'same' convolution of arbitrary shapes on the FFT engine: the volume is staged into a
    2^a * {1,3,9} padded array and runs through the hand-written pipeline; same numbers as the rocFFT route and as the
    direct engine."""
    from ipp_amd import decon
    rng = np.random.default_rng(sum(shape) + boundary)
    img = rng.random(shape, dtype=np.float32)
    ker = rng.random(kshape, dtype=np.float32)
    ker /= ker.sum()
    t, k = torch.from_numpy(img).to(dev), torch.from_numpy(ker).to(dev)
    monkeypatch.setenv("MI_FFT_NATIVE_INFLATE", "100")
    got = decon.convn_same(t, k, boundary=boundary, engine=2).cpu().numpy()
    direct = decon.convn_same(t, k, boundary=boundary, engine=1).cpu().numpy()
    os.environ["MI_FFT_ROCFFT"] = "1"
    try:
        ref = decon.convn_same(t, k, boundary=boundary, engine=2).cpu().numpy()
    finally:
        del os.environ["MI_FFT_ROCFFT"]
    assert _rel(got, direct.astype(np.float64)) < 2e-5
    assert _rel(got, ref.astype(np.float64)) < 2e-5
    if boundary == 0:
        want = R.convn_same(img, ker)
        assert _rel(got, want.astype(np.float64)) < 2e-5


@pytest.mark.parametrize("kshape", [(5, 7, 9), (4, 6, 8)])
def test_spatial_rl_on_fft_engine_uses_explicit_adjoint(dev, kshape, monkeypatch):
    """decon.m spatial flavour (zero boundary, psf_inv given explicitly) on the FFT engine: for even PSF extents the
    adjoint is not conj(OTF), so the native pipeline carries a second OTF."""
    from ipp_amd import capi, decon
    shape = (20, 45, 70)
    psf = R.gaussian_psf(kshape, (1.0, 1.5, 2.0))
    psf_inv = np.ascontiguousarray(psf[::-1, ::-1, ::-1])
    monkeypatch.setenv("MI_FFT_NATIVE_INFLATE", "100")
    vol = torch.from_numpy(R.bead_volume(shape, seed=5, psf=R.gaussian_psf((5, 5, 5), (1.0, 1.0, 1.0)))).to(dev)
    fft = decon.RLContext(shape, psf, psf_inv, boundary=capi.BOUNDARY_ZERO, engine=capi.ENGINE_FFT, device=dev)
    direct = decon.RLContext(shape, psf, psf_inv, boundary=capi.BOUNDARY_ZERO, engine=capi.ENGINE_DIRECT, device=dev)
    a, b = vol.clone(), vol.clone()
    ratio = torch.empty_like(vol)
    fft.iterate(a, ratio, 3)
    direct.iterate(b, ratio, 3)
    assert _rel(a.cpu().numpy(), b.cpu().numpy().astype(np.float64)) < 1e-4


@pytest.mark.parametrize("shape,kshape", [((20, 45, 70), (5, 7, 9)), ((33, 64, 100), (4, 6, 8)), ((9, 200, 31), (3, 3, 5))])
def test_spatial_rl_fused_iterations_on_padded_grid(dev, shape, kshape, monkeypatch):
    """Zero-boundary RL (decon.m:25-120 flavour) on the FFT engine: the x passes pad and crop on the fly and consecutive
    convolutions share them; no ratio scratch is needed.  Same result as half-step calls and as the direct engine."""
    from ipp_amd import capi, decon
    psf = R.gaussian_psf(kshape, (1.0, 1.5, 2.0))
    psf_inv = np.ascontiguousarray(psf[::-1, ::-1, ::-1])
    vol = torch.from_numpy(R.bead_volume(shape, seed=11, psf=R.gaussian_psf((5, 5, 5), (1.0, 1.0, 1.0)))).to(dev)
    monkeypatch.setenv("MI_FFT_NATIVE_INFLATE", "100")   # small extents round up a lot: keep the hand-written pipeline anyway
    fft = decon.RLContext(shape, psf, psf_inv, boundary=capi.BOUNDARY_ZERO, engine=capi.ENGINE_FFT, device=dev)
    direct = decon.RLContext(shape, psf, psf_inv, boundary=capi.BOUNDARY_ZERO, engine=capi.ENGINE_DIRECT, device=dev)
    a, b, c = vol.clone(), vol.clone(), vol.clone()
    ratio = torch.empty_like(vol)
    fft.iterate(a, None, 5)
    for _ in range(5):
        fft.forward_ratio(b, ratio)
        fft.adjoint_update(ratio, b)
    direct.iterate(c, ratio, 5)
    assert _rel(a.cpu().numpy(), b.cpu().numpy().astype(np.float64)) < 1e-5
    assert _rel(a.cpu().numpy(), c.cpu().numpy().astype(np.float64)) < 1e-4


def test_replicate_padded_engine_does_not_fuse(dev, monkeypatch):
    from ipp_amd import capi, decon
    monkeypatch.setenv("MI_FFT_NATIVE_INFLATE", "100")
    psf = R.gaussian_psf((5, 5, 5), (1.0, 1.0, 1.0))
    ctx = decon.RLContext((16, 40, 50), psf, None, boundary=capi.BOUNDARY_REPLICATE, engine=capi.ENGINE_FFT, device=dev)
    bl = torch.rand((16, 40, 50), device=dev) + 0.5
    with pytest.raises(capi.MiError, match="ratio scratch"):
        ctx.iterate(bl, None, 1)
    want = bl.clone()
    ratio = torch.empty_like(bl)
    ctx.iterate(bl, ratio, 2)
    direct = decon.RLContext((16, 40, 50), psf, None, boundary=capi.BOUNDARY_REPLICATE, engine=capi.ENGINE_DIRECT, device=dev)
    direct.iterate(want, ratio, 2)
    assert _rel(bl.cpu().numpy(), want.cpu().numpy().astype(np.float64)) < 1e-4


@pytest.mark.parametrize("shape,flavour", [((64, 64, 128), "circular"), ((64, 64, 129), "circular_odd_x_is_rocfft"), ((40, 50, 100), "zero")])
def test_real_otf_form_for_symmetric_psfs(dev, shape, flavour, monkeypatch):
    """A PSF that is mirror-symmetric about its centre sample has a real OTF up to the phase ramp of its placement: the z pass
    then reads half the OTF bytes.  Same result as the complex form; asymmetric PSFs keep the complex form."""
    from ipp_amd import capi, decon
    monkeypatch.setenv("MI_FFT_NATIVE_INFLATE", "100")
    if shape[2] % 2:
        pytest.skip("odd x extents run through rocFFT")
    bnd = capi.BOUNDARY_CIRCULAR if flavour == "circular" else capi.BOUNDARY_ZERO
    psf = R.gaussian_psf((7, 5, 9), (1.5, 1.0, 2.0))
    vol = torch.from_numpy(R.bead_volume(shape, seed=3, psf=psf)).to(dev)

    def run(p, complex_only):
        if complex_only:
            monkeypatch.setenv("MI_FFT_COMPLEX_OTF", "1")
        else:
            monkeypatch.delenv("MI_FFT_COMPLEX_OTF", raising=False)
        ctx = decon.RLContext(shape, p, None, boundary=bnd, engine=capi.ENGINE_FFT, device=dev)
        bl = vol.clone()
        ctx.iterate(bl, None, 3)
        return bl.cpu().numpy(), ctx.device_bytes

    got_r, bytes_r = run(psf, False)
    got_c, bytes_c = run(psf, True)
    assert _rel(got_r, got_c.astype(np.float64)) < 2e-6
    assert bytes_r < bytes_c                      # the real form stores 8 instead of 16 bytes per OTF pair
    skew = psf.copy()
    skew[0, 0, 0] *= 3.0                           # not symmetric any more: complex form, same memory as the forced one
    skew /= skew.sum()
    _, bytes_s = run(skew, False)
    assert bytes_s == bytes_c
    want = (R.decon_fft(vol.cpu().numpy(), psf, shape, 3, skip_edgetaper=True) if flavour == "circular"
            else R.decon_spatial(vol.cpu().numpy(), psf, 3, skip_edgetaper=True))
    assert_close(got_r, want)


def _random_cases(n=int(__import__("os").environ.get("MI_TEST_SWEEP", "14")), seed=int(__import__("os").environ.get("MI_TEST_SWEEP_SEED", "2026"))):
    rng = np.random.default_rng(seed)
    sizes = [8, 16, 32, 64, 96, 128, 192, 288]          # native extents: 2^a and 3 * / 9 * 2^a
    out = []
    for i in range(n):
        circular = bool(rng.integers(0, 2))
        if circular:
            shape = tuple(int(rng.choice(sizes[:6])) for _ in range(3))
        else:
            shape = tuple(int(rng.integers(9, 90)) for _ in range(3))
        # (extent-1 axes only off the circular rule: deconFFT's even-shape placement puts a 1-sample PSF at index -1, which the
        # context rejects as a shift outside the PSF)
        k = tuple(int(rng.choice([3, 5, 7] if circular else [1, 3, 5, 7])) for _ in range(3))
        k = tuple(min(kk, s) for kk, s in zip(k, shape))
        out.append((shape, k, 2 if circular else int(rng.integers(0, 2)), bool(rng.integers(0, 2)), i))
    return out


@pytest.mark.parametrize("shape,kshape,boundary,symmetric,seed", _random_cases())
def test_random_shapes_fft_engine_equals_direct_engine(dev, shape, kshape, boundary, symmetric, seed, monkeypatch):
    """Seeded sweep over shapes, PSF extents, boundary rules and (a)symmetric PSFs: three RL iterations on the FFT engine (native
    grids incl. radix-3/9 axes, tiny transforms, padded + pruned grids, real and complex OTF forms) against the direct engine."""
    from ipp_amd import capi, decon
    monkeypatch.setenv("MI_FFT_NATIVE_INFLATE", "100")
    rng = np.random.default_rng(100 + seed)
    if symmetric:
        psf = R.gaussian_psf(kshape, (1.0, 1.3, 0.8))
    else:
        psf = rng.random(kshape, dtype=np.float32) + 0.05
        psf /= psf.sum()
    psf_inv = np.ascontiguousarray(psf[::-1, ::-1, ::-1]) if boundary != 2 else None
    vol = torch.from_numpy(rng.random(shape, dtype=np.float32) + 0.2).to(dev)
    fft = decon.RLContext(shape, psf, psf_inv, boundary=boundary, engine=capi.ENGINE_FFT, device=dev)
    direct = decon.RLContext(shape, psf, psf_inv, boundary=boundary, engine=capi.ENGINE_DIRECT, device=dev)
    a, b = vol.clone(), vol.clone()
    ratio = torch.empty_like(vol)
    fft.iterate(a, ratio, 3)
    direct.iterate(b, ratio, 3)
    assert _rel(a.cpu().numpy(), b.cpu().numpy().astype(np.float64)) < 1e-4


@pytest.mark.parametrize("shape", [(64, 64, 128), (128, 32, 64), (192, 96, 64)])
def test_tile_hand_out_is_bit_identical(dev, shape, monkeypatch):
    """How the persistent kernels get their tiles (device counter -- the default -- or the fixed stride, MI_X_DYN / MI_Z_DYN = 0)
    changes the ORDER of the work only: the results of fused iterations are identical bit for bit."""
    from ipp_amd import capi, decon
    psf = R.gaussian_psf((5, 7, 5), (1.0, 1.5, 1.0))
    vol = torch.from_numpy(R.bead_volume(shape, seed=23, psf=psf)).to(dev)

    def run():
        ctx = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
        assert ctx.fuses and ctx.pair_layout
        bl = vol.clone()
        ctx.iterate(bl, None, 3)
        return bl

    ref = run()
    for env in ({"MI_X_DYN": "0", "MI_Z_DYN": "0"},):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = run()
        for k in env:
            monkeypatch.delenv(k)
        assert torch.equal(got, ref), env


def test_overlap_settings_and_probe(dev):
    """mi_rl_set_overlap / mi_rl_overlap_probe: part 2 of a split step with compute units left free, with and without the device
    counter, beside a stand-in collective -- every setting leaves the arithmetic alone (the sharded step equals the whole step)."""
    from ipp_amd import capi, decon
    psf = R.gaussian_psf((5, 7, 5), (1.0, 1.5, 1.0))
    shape = (16, 128, 64)
    ctx = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
    assert ctx.fuses == 2
    vol = torch.from_numpy(R.bead_volume(shape, seed=29, psf=psf)).to(dev)
    edges = (16, 32, 96, 112)
    whole = vol.clone()
    ctx.sharded_begin(whole)
    ctx.sharded_ratio(whole)
    ctx.sharded_update(whole, True)
    for free, dyn in ((0, True), (0, False), (8, True), (200, False)):
        ctx.set_overlap(free, dyn)
        bl = vol.clone()
        ctx.sharded_begin(bl)
        ctx.sharded_ratio(bl, 1, edges)
        ctx.sharded_ratio(bl, 2, edges)
        ctx.sharded_update(bl, True, 1, edges)
        ctx.sharded_update(bl, True, 2, edges)
        assert torch.equal(bl, whole), (free, dyn)
    x_ms, all_ms = ctx.overlap_probe(vol.clone(), edges, busy_wgs=4, busy_us=200.0, reps=2)
    assert 0.0 < x_ms < 50.0 and all_ms >= 0.19
    with pytest.raises(capi.MiError, match="free_cus"):
        ctx.set_overlap(100000, True)


def test_placement_trials_keep_one_candidate_and_the_result(dev, monkeypatch, capfd):
    """Spectrum arrays placed by trial (fft_native.hip, NativeFft::init: several buffers allocated side by side, the ordered pair on
    which the passes run fastest becomes (S, T), the others go back): forced onto a small shape, the context computes what a plain
    allocation computes, bit for bit, and the log and mi_rl_fft_placement name the pairs."""
    from ipp_amd import capi, decon
    shape, kshape = (32, 64, 128), (7, 5, 9)
    rng = np.random.default_rng(3)
    vol = (rng.random(shape, dtype=np.float32) + 0.1)
    psf = R.gaussian_psf(kshape, (1.5, 1.0, 2.0))

    record = []

    def run(calls=(3,)):
        ctx = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
        record.append(ctx.fft_placement())            # mi_rl_fft_placement
        bl = torch.from_numpy(vol).to(dev)
        for n in calls:
            ctx.iterate(bl, None, n)
        return bl.cpu().numpy()

    monkeypatch.setenv("MI_FFT_PLACE_CANDIDATES", "1")
    plain = run()
    monkeypatch.setenv("MI_FFT_PLACE_CANDIDATES", "3")
    monkeypatch.setenv("MI_FFT_PLACE_MIN_MB", "0")
    monkeypatch.setenv("MI_FFT_PLACE_LOG", "1")
    capfd.readouterr()
    placed = run()
    err = capfd.readouterr().err
    assert "(T) of 3" in err and "placed on buffers" in err, err
    assert record[0] == ([], -1) and len(record[1][0]) == 6 and 0 <= record[1][1] < 6      # the ordered pairs of three buffers
    assert record[1][0][record[1][1]] == min(record[1][0])
    assert np.array_equal(plain, placed)
    assert_close(placed, R.decon_fft(vol, psf, shape, 3, skip_edgetaper=True))
    # a second buffer for S, settled on the loop's own update launches: inside one call of five iterations, and across calls of two
    monkeypatch.setenv("MI_FFT_PLACE_ALT_MIN_MB", "0")
    monkeypatch.setenv("MI_FFT_PLACE_CANDIDATES", "1")
    plain5 = run((5,))
    monkeypatch.setenv("MI_FFT_PLACE_CANDIDATES", "4")
    capfd.readouterr()
    one_call = run((5,))
    assert "S settled on the" in capfd.readouterr().err
    across = run((2, 2, 1))
    assert "S settled on the" in capfd.readouterr().err
    assert np.array_equal(plain5, one_call) and np.array_equal(plain5, across)
