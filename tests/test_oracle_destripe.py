"""CPU: the destripe oracle (oracle/destripe_oracle.py, filter_subband_3d_z.m restated; parity unpinned -- MATLAB's Wavelet
Toolbox is closed) against its known answers: the published db9 table, orthonormality, perfect reconstruction, bookkeeping of
coefficient sizes, and what the filter is for -- stripes that run along z disappear."""
import numpy as np
import pytest

from oracle import destripe_oracle as D

# the 18 published coefficients of the db9 scaling filter (sum = sqrt(2)), e.g. PyWavelets' db9 rec_lo
DB9_REC_LO = [0.03807794736316728, 0.24383467463766728, 0.6048231236767786, 0.6572880780366389, 0.13319738582208895,
              -0.29327378327258685, -0.09684078322087904, 0.14854074933476008, 0.030725681478322865, -0.06763282905952399,
              0.00025094711499193845, 0.022361662123515244, -0.004723204757894831, -0.004281503681904723,
              0.0018476468829611268, 0.00023038576399541288, -0.0002519631889981789, 3.9347319995026124e-05]


def test_db9_filters():
    lo_d, hi_d, lo_r, hi_r = D.db_filters(9)
    assert np.abs(lo_r - np.array(DB9_REC_LO)).max() < 1e-10
    assert abs(lo_r.sum() - np.sqrt(2)) < 1e-12 and abs(hi_r.sum()) < 1e-9           # low-pass gain, zero DC of the wavelet
    for shift in range(0, 18, 2):                                                  # orthonormal even translates
        want = 1.0 if shift == 0 else 0.0
        assert abs(np.dot(lo_r[shift:], lo_r[:18 - shift]) - want) < 1e-12
        assert abs(np.dot(lo_r[shift:], hi_r[:18 - shift])) < 1e-12
    for k in range(9):                                                             # nine vanishing moments
        assert abs(np.dot(np.arange(18.0) ** k, hi_r)) < 1e-5 * 18.0 ** k
    assert np.array_equal(lo_d, lo_r[::-1]) and np.array_equal(hi_d, hi_r[::-1])
    assert np.allclose(hi_r[:3], [3.9347319995026124e-05, 0.0002519631889981789, 0.00023038576399541288], atol=1e-10)
    lo2 = D.db_filters(2)[2]                                                       # db2 in closed form
    s3 = np.sqrt(3.0)
    assert np.allclose(lo2, np.array([1 + s3, 3 + s3, 3 - s3, 1 - s3]) / (4 * np.sqrt(2)), atol=1e-12)


@pytest.mark.parametrize("shape", [(64, 40), (70, 52), (150, 36), (34, 34)])
def test_perfect_reconstruction_and_sizes(shape):
    rng = np.random.default_rng(1)
    x = rng.random(shape)
    f = D.db_filters(9)
    lev = max(D.wmaxlev(shape), 1)
    a, det, sizes = D.wavedec2(x, lev, f)
    n = list(shape)
    for H, V, Dd in det[::-1]:
        n = [(v + 17) // 2 for v in n]                                            # floor((n + lf - 1) / 2)
        assert H.shape == V.shape == Dd.shape == tuple(n)
    assert a.shape == tuple(n)
    assert np.abs(D.waverec2(a, det, sizes, f) - x).max() < 1e-12
    # Parseval does not hold exactly with the symmetric extension, but a constant image is all approximation
    a, det, sizes = D.wavedec2(np.ones(shape), 1, f)
    assert np.abs(a - 2.0).max() < 1e-9 and all(np.abs(c).max() < 1e-9 for c in det[0])


def test_wmaxlev_and_notch_filter():
    assert [D.wmaxlev((n, 4096)) for n in (16, 17, 33, 34, 67, 68, 512, 2048)] == [0, 0, 0, 1, 1, 2, 4, 6]
    g = D.gaussian_notch_filter_1d(8, 1.0)
    x = np.array([0, 1, 2, 3, -4, -3, -2, -1], np.float32)
    assert np.allclose(g, 1 - np.exp(-x * x / 2), atol=1e-7) and g[0] == 0
    # sigma / n << 1 (what filter_subband passes), even length: only the DC bin is removed -> the mean along the axis goes
    rng = np.random.default_rng(2)
    H = rng.normal(size=(6, 22)).astype(np.float32)
    out = D.filter_coefficient(H, 2.0 / 22, 2)
    assert np.abs(out - (H - H.mean(axis=1, keepdims=True))).max() < 1e-6
    wide = D.filter_coefficient(H, 3.0, 2)                                         # a wide notch touches several bins
    assert np.abs(np.fft.fft(wide, axis=1) - np.fft.fft(H, axis=1) * D.gaussian_notch_filter_1d(22, 3.0)).max() < 1e-4
    # odd length (:122 fftshift of an odd-length vector): the zero of g sits on the last bin, frequency -1, and the
    # real(.(1 + i)) of the reference returns Re - Im of the now complex signal
    H = rng.normal(size=(4, 21)).astype(np.float32)
    g = D.gaussian_notch_filter_1d(21, 2.0 / 21)
    assert g[20] == 0 and np.all(g[:20] == 1)
    F = np.fft.fft(H.astype(np.float64), axis=1)
    z = np.arange(21)
    r = H - (F[:, 20:21] * np.exp(2j * np.pi * 20 * z / 21)[None, :]) / 21
    assert np.abs(D.filter_coefficient(H, 2.0 / 21, 2) - (r.real - r.imag)).max() < 1e-6


def test_stripes_along_z_are_removed():
    rng = np.random.default_rng(3)
    Z, Y, X = 72, 2, 80
    clean = (rng.random((Z, Y, X)) * 0.05 + 1.0).astype(np.float32)
    gain = np.ones(X, np.float32)
    gain[::9] = 1.6                                                                # columns brighter at every z: stripes along z
    striped = clean * gain[None, None, :]
    out = D.filter_subband_3d_z(striped, 2.0)
    prof = lambda v: v.mean(axis=(0, 1))
    before, after = np.std(prof(striped)), np.std(prof(out))
    assert after < 0.5 * before
    assert out.shape == striped.shape and out.dtype == np.float32
    # a volume without x structure passes (nearly) unchanged: the H bands are empty
    flat = np.broadcast_to((rng.random((Z, 1, 1)) + 1).astype(np.float32), (Z, Y, X)).copy()
    assert np.abs(D.filter_subband_3d_z(flat, 2.0) - flat).max() < 1e-4
    # odd extents: padded to even, cropped back
    odd = D.filter_subband_3d_z(striped[:71, :, :79], 2.0)
    assert odd.shape == (71, 2, 79)
    with pytest.raises(ValueError):
        D.filter_subband_3d_z(striped, 1.0, 0, "db4")
