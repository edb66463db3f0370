"""TEST INFRASTRUCTURE -- CPU restatement of ``filter_subband_3d_z`` (LsDeconvolveMultiGPU/filter_subband_3d_z.m:1-123), the
optional wavelet + FFT-notch destripe of every XZ slice of a block.  Only tests/, smoke() and the
cpu_baseline leg of bench.py may import this file; the product path never does.

PARITY UNPINNED: the reference calls MATLAB's Wavelet Toolbox (``wavedec2`` / ``waverec2`` / ``wmaxlev`` with 'db9', default
extension mode 'sym'), which is closed and absent here, and stores no golden vectors for this filter.  What is restated is the
published algorithm of those builtins (the same one PyWavelets' 'symmetric' mode implements):

  * db9 filters: the extremal-phase spectral factor of the Daubechies half-band polynomial, ``Lo_R = sqrt(2) * dbwavf('db9')``,
    ``Lo_D = flip(Lo_R)``, ``Hi_R = qmf(Lo_R)`` (alternating signs of the reversed filter), ``Hi_D = flip(Hi_R)``;
    pinned by the published 18 coefficients of db9 (first / largest: 0.0380779473..., 0.6572880780...) and orthonormality;
  * ``dwt`` ('sym'): half-point symmetric extension by lf-1, full convolution, keep the even (1-based) samples:
    ``out[i] = sum_t F[t] x[sym(2 i + 1 - t)]``, ``floor((n + lf - 1) / 2)`` coefficients;
  * ``idwt``: zero-stuffing (``dyadup(.,0)``), full convolution with Lo_R / Hi_R, central ``s`` samples (offset lf - 2);
  * ``wavedec2`` / ``waverec2`` bookkeeping: [A_N | H_N V_N D_N | ... | H_1 V_1 D_1], H = high-pass along dim 1 (X) of the
    low-pass along dim 2 (Z); ``wmaxlev = fix(log2(min(size) / (lf - 1)))``.

Pinned only by known answers: perfect reconstruction, the db9 coefficient table, constant-along-z stripes being removed.
Arrays are (Z, Y, X) like the rest of the package; a MATLAB slice ``[X, Z]`` is ``bl[:, y, :].T``.
"""
from __future__ import annotations

import math

import numpy as np

EPS_SINGLE = np.float32(2.0 ** -23)


def db_filters(N: int = 9):
    """(Lo_D, Hi_D, Lo_R, Hi_R) of the Daubechies wavelet with N vanishing moments, float64."""
    # P(y) = sum_k C(N-1+k, k) y^k, y = (2 - z - 1/z) / 4; the roots inside the unit circle give the extremal-phase factor
    coeffs = [math.comb(N - 1 + k, k) for k in range(N)]
    y_roots = np.roots(coeffs[::-1])
    z_roots = []
    for y in y_roots:
        # z^2 - (2 - 4 y) z + 1 = 0
        b = 2.0 - 4.0 * y
        disc = np.sqrt(b * b - 4.0 + 0j)
        z1, z2 = (b + disc) / 2.0, (b - disc) / 2.0
        z_roots.append(z1 if abs(z1) < 1 else z2)
    poly = np.poly(np.concatenate([np.full(N, -1.0), np.array(z_roots)]))
    lo_r = np.real(poly)
    lo_r = lo_r / lo_r.sum() * math.sqrt(2.0)
    hi_r = lo_r[::-1].copy()
    hi_r[1::2] = -hi_r[1::2]              # qmf(x, 0): the even (1-based) entries of the reversed filter change sign
    return lo_r[::-1].copy(), hi_r[::-1].copy(), lo_r, hi_r


def _sym_index(j, n):
    """Half-point symmetric extension ('sym'), reflecting as often as needed."""
    j = np.asarray(j)
    period = 2 * n
    j = np.mod(j, period)
    return np.where(j < n, j, period - 1 - j)


def dwt_axis(x, F, axis):
    """One analysis filter along ``axis``: out[i] = sum_t F[t] x[sym(2 i + 1 - t)]."""
    x = np.moveaxis(x, axis, -1)
    n, lf = x.shape[-1], len(F)
    m = (n + lf - 1) // 2
    out = np.zeros(x.shape[:-1] + (m,), x.dtype)
    i = np.arange(m)
    for t in range(lf):
        out += x.dtype.type(F[t]) * x[..., _sym_index(2 * i + 1 - t, n)]
    return np.moveaxis(out, -1, axis)


def idwt_axis(a, d, Lo_R, Hi_R, s, axis):
    """Synthesis along ``axis`` to length s: out[j] = sum_k a[k] Lo_R[j + lf - 2 - 2 k] + d[k] Hi_R[...]."""
    a, d = np.moveaxis(a, axis, -1), np.moveaxis(d, axis, -1)
    m, lf = a.shape[-1], len(Lo_R)
    out = np.zeros(a.shape[:-1] + (s,), a.dtype)
    j = np.arange(s)
    for k in range(m):
        t = j + lf - 2 - 2 * k
        ok = (t >= 0) & (t < lf)
        if not ok.any():
            continue
        tt = np.clip(t, 0, lf - 1)
        out += np.where(ok, a.dtype.type(1), a.dtype.type(0)) * (a[..., k:k + 1] * Lo_R[tt].astype(a.dtype) + d[..., k:k + 1] * Hi_R[tt].astype(a.dtype))
    return np.moveaxis(out, -1, axis)


def wmaxlev(size, lf=18):
    lev = int(math.log2(min(size) / (lf - 1))) if min(size) >= lf - 1 else 0
    return max(lev, 0)


def wavedec2(img, levels, filters):
    """img [X, Z] (MATLAB orientation).  Returns (A_N, [(H, V, D) coarsest..finest], sizes finest-first for reconstruction)."""
    Lo_D, Hi_D = filters[0], filters[1]
    a = img
    details, sizes = [], []
    for _ in range(levels):
        sizes.append(a.shape)
        zl, zh = dwt_axis(a, Lo_D, 1), dwt_axis(a, Hi_D, 1)           # dim 2 (Z) first: dwt2.m
        A, H = dwt_axis(zl, Lo_D, 0), dwt_axis(zl, Hi_D, 0)           # then dim 1 (X)
        V, D = dwt_axis(zh, Lo_D, 0), dwt_axis(zh, Hi_D, 0)
        details.append((H, V, D))
        a = A
    return a, details[::-1], sizes[::-1]


def waverec2(a, details, sizes, filters):
    Lo_R, Hi_R = filters[2], filters[3]
    for (H, V, D), s in zip(details, sizes):
        zl = idwt_axis(a, H, Lo_R, Hi_R, s[0], 0)
        zh = idwt_axis(V, D, Lo_R, Hi_R, s[0], 0)
        a = idwt_axis(zl, zh, Lo_R, Hi_R, s[1], 1)
    return a


def gaussian_notch_filter_1d(n, sigma):
    """filter_subband_3d_z.m:117-123, single precision like the reference's ``x = single(x)``."""
    x = (np.arange(n) - n // 2).astype(np.float32)
    g = np.float32(1) - np.exp(-(x * x) / np.float32(2.0 * sigma * sigma)).astype(np.float32)
    return np.fft.fftshift(g)


def filter_coefficient(mat, sigma, axis):
    """:92-115: FFT along ``axis`` (1-based MATLAB axis), times ``complex(g, g)`` = g (1 + i), inverse FFT, real part.
    For an even length g is an even function of the frequency with its zero at DC and the (1 + i) factor drops out.  For an
    ODD length ``fftshift`` (:122) leaves the zero of the notch on the LAST bin (frequency -1) instead of DC, the filtered
    signal is complex and ``real(.(1 + i))`` returns Re - Im of it; kept as written."""
    sigma = max(float(sigma), float(EPS_SINGLE))
    ax = axis - 1
    n = mat.shape[ax]
    g = gaussian_notch_filter_1d(n, sigma).astype(np.float64)
    shape = [1, 1]
    shape[ax] = n
    gg = g.reshape(shape)
    spec = np.fft.fft(mat.astype(np.float64), axis=ax) * (gg + 1j * gg)
    return np.real(np.fft.ifft(spec, axis=ax)).astype(np.float32)


def filter_subband(img, sigma, levels, filters=None, axes=(2,)):
    """:45-90 on one slice [X, Z], float32."""
    filters = filters or db_filters(9)
    pad = [s % 2 for s in img.shape]
    img = np.pad(img, [(0, pad[0]), (0, pad[1])])
    if levels == 0:
        levels = wmaxlev(img.shape, len(filters[0]))
    a, details, sizes = wavedec2(img.astype(np.float32), levels, filters)
    out = []
    for H, V, D in details:
        if 2 in axes:
            H = filter_coefficient(H, sigma / H.shape[1], 2)
        if 1 in axes:
            V = filter_coefficient(V, sigma / V.shape[0], 1)
        out.append((H, V, D))
    img = waverec2(a, out, sizes, filters)
    return img[:img.shape[0] - pad[0], :img.shape[1] - pad[1]]


def filter_subband_3d_z(bl, sigma, levels=0, wavelet="db9"):
    """:1-43: log1p, every XZ slice through filter_subband along axis 2, expm1.  bl is (Z, Y, X) float32."""
    if wavelet != "db9":
        raise ValueError("only db9 is restated (LsDeconv.m:935 passes \"db9\")")
    filters = db_filters(9)
    bl = np.log1p(bl.astype(np.float32))
    out = np.empty_like(bl)
    for y in range(bl.shape[1]):
        out[:, y, :] = filter_subband(bl[:, y, :].T, sigma, levels, filters).T
    return np.expm1(out).astype(np.float32)
