"""CPU: the PSF generator (host-side input of the RL path) against arrays returned by the reference's
psf_generator.generate_psf (tests/golden/make_psf_golden.py)."""
import ast

import numpy as np
import pytest

from ipp_amd import psf as P


@pytest.mark.parametrize("idx", range(4))
def test_generate_psf_matches_reference(psf_golden, idx):
    g = psf_golden
    name = str(g["names"][idx])
    kw = dict(ast.literal_eval(str(g[f"{name}/kwargs"])))
    psf, dxy = P.generate_psf(**kw)
    ref = g[f"{name}/psf"]
    assert psf.shape == ref.shape and psf.dtype == np.float32
    assert dxy == pytest.approx(float(g[f"{name}/dxy_psf"]), rel=1e-12)
    assert np.abs(psf - ref).max() <= 1e-6 * ref.max()
    assert abs(float(psf.sum()) - 1.0) < 1e-5


def test_psf_is_odd_symmetric_and_matlab_flavour_differs_only_in_sampling():
    psf, _ = P.generate_psf(lambda_em=525.0, lambda_ex=488.0, dxy=422.0, dz=1000.0)
    assert all(s % 2 == 1 for s in psf.shape)
    assert np.allclose(psf, psf[::-1, ::-1, ::-1])
    m = P.LsMakePSF(422.0, 1000.0, 0.4, 1.42, 488.0, 525.0, 240.0, 12.0)
    assert m.shape == psf.shape and abs(float(m.sum()) - 1.0) < 1e-5


def test_resample_psf_shapes():
    psf, _ = P.generate_psf()
    for shape in [(31, 15, 15), (61, 31, 31), (5, 5, 5)]:
        r = P.resample_psf(psf, shape)
        assert r.shape == shape and abs(float(r.sum()) - 1.0) < 1e-5
