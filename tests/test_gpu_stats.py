"""GPU: post-deconvolution statistics and output conversion through the C ABI against the oracle:
exact percentiles of ``deconvolved_stats`` (LsDeconv.m:1300-1307) and the rescale / round / clamp / convert of
``load_slab_lz4`` (load_slab_lz4.cpp:134-157)."""
import numpy as np
import pytest
import torch

from oracle import rl_oracle as R

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(5)
    yield "uniform", rng.random(100_003, dtype=np.float32)
    yield "beads", R.bead_volume((24, 80, 96), seed=3, psf=R.gaussian_psf((5, 5, 5), (1.0, 1.0, 1.0))).ravel()
    yield "mostly_zero", np.where(rng.random(70_000) < 0.97, 0.0, rng.random(70_000)).astype(np.float32)
    yield "signed", rng.normal(size=50_001).astype(np.float32)
    yield "constant", np.full(4097, 3.25, np.float32)
    yield "tiny", np.array([5.0, 1.0, 3.0], np.float32)
    yield "one", np.array([7.5], np.float32)
    x = rng.random(10_000, dtype=np.float32)
    x[::17] = np.nan
    yield "with_nan", x
    yield "wide_range", (10.0 ** rng.uniform(-30, 30, 60_000)).astype(np.float32) * rng.choice([-1, 1], 60_000).astype(np.float32)


@pytest.mark.parametrize("name,data", list(_cases()), ids=[c[0] for c in _cases()])
@pytest.mark.parametrize("pcts", [(0.01, 99.99), (0.0, 100.0), (50.0,), (25.0, 75.0)])
def test_prctile_matches_oracle(dev, name, data, pcts):
    from ipp_amd import decon
    t = torch.from_numpy(data).to(dev)
    got = decon.prctile(t, list(pcts))
    want = R.prctile(data, pcts)
    for g, w in zip(got, want):
        # the order statistics are exact; the interpolation between two neighbours is done in double on both sides
        assert g == pytest.approx(float(w), rel=2e-7, abs=0.0) or (np.isnan(g) and np.isnan(w))


def test_prctile_full_block_and_process_block_stats(dev):
    """deconvolved_stats on a block uses every voxel (no sub-sampling): equal to the oracle on 8.4 M voxels."""
    from ipp_amd import lsdeconv
    vol = R.bead_volume((32, 512, 512), seed=8, psf=R.gaussian_psf((5, 7, 7), (1.0, 1.5, 1.5)))
    lb, ub = lsdeconv.deconvolved_stats(torch.from_numpy(vol).to(dev), 99.99)
    wl, wu = R.prctile(vol, [0.01, 99.99])
    assert lb == pytest.approx(float(wl), rel=2e-7) and ub == pytest.approx(float(wu), rel=2e-7)


def test_prctile_errors(dev):
    from ipp_amd import capi, decon
    t = torch.rand(100, device=dev)
    with pytest.raises(capi.MiError, match="percentiles must be in"):
        decon.prctile(t, [101.0])
    with pytest.raises(ValueError):
        decon.prctile(t, [1.0, 2.0, 3.0])
    with pytest.raises(ValueError):
        decon.prctile(t.cpu(), [50.0])


@pytest.mark.parametrize("scal,dtype", [(255.0, np.uint8), (65535.0, np.uint16)])
@pytest.mark.parametrize("dmin,dmax,ampl", [(0.0, 5.3374, 1.0), (0.0123, 0.9871, 1.0), (0.02, 3.7, 2.5), (0.0, 1.0, 0.3)])
def test_rescale_block_bit_exact(dev, scal, dtype, dmin, dmax, ampl):
    from ipp_amd import decon
    rng = np.random.default_rng(int(scal) + int(ampl * 10))
    x = (rng.random(200_003, dtype=np.float32) * np.float32(dmax * 1.2)).astype(np.float32)
    x[:5] = [0.0, dmin, dmax, dmax * 2, -1.0]
    # values that land exactly on .5 after scaling exercise the round-half-away rule
    got = decon.rescale_block(torch.from_numpy(x).to(dev), scal, ampl, dmin, dmax).cpu().numpy()
    want = R.rescale_block(x, scal, ampl, dmin, dmax, dtype)
    assert got.dtype == dtype
    assert np.array_equal(got, want)


def test_output_scale_rule():
    from ipp_amd import lsdeconv
    assert lsdeconv.output_scale(200) == 255 and lsdeconv.output_scale(4095) == 65535
    assert lsdeconv.output_scale(70000.0) == 70000.0
    assert lsdeconv.output_scale(70000.0, convert_to_16bit=True) == 65535
    assert lsdeconv.output_scale(4095, convert_to_8bit=True) == 255
