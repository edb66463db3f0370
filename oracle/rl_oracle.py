"""CPU ORACLE (test infrastructure, NOT product code) for the Richardson-Lucy path.

numpy/scipy restatement of the reference's LsDeconvolveMultiGPU hot path.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the shipped path (``ipp_amd``) never does.

PARITY UNPINNED for this half of the oracle: the reference's arithmetic lives in
closed MATLAB builtins (convn / fftn / imfilter / imgaussfilt3) and in CUDA MEX
files that cannot be built here (no MATLAB, no nvcc), and the
reference holds no stored golden vectors.  The restatement therefore follows the
reference *text* line by line (citations below, relative to /root/reference) and is
pinned only by the known-answer / property checks taken from the reference's own
test scripts (tests/test_oracle_rl.py).

Array convention: numpy C-order ``(Z, Y, X)`` == MATLAB ``[X, Y, Z]`` column-major
(conv3d_gpu.cu:93,98; gauss3d_gpu.cu:123).  All PSF extents are odd
(LsMakePSF.m:32-38), so the centre is ``(k-1)/2 == k//2``.
"""
from __future__ import annotations

import math

import numpy as np
from scipy import ndimage, signal

EPS_SINGLE = np.float32(2.0 ** -23)  # eps('single'), decon.m:62,154


# --------------------------------------------------------------------------- convolutions
def convn_same(a: np.ndarray, h: np.ndarray) -> np.ndarray:
    """MATLAB ``convn(a, h, 'same')``: true convolution, zero outside, central part
    (decon.m:61,64,70): ``full[ceil((k-1)/2) : ...]``, i.e. centre ``k/2`` (integer division) also for
    even extents -- ndimage's convention (tests/test_oracle_rl.py checks it against the full
    convolution).  Computed in float64 and rounded once to float32."""
    out = ndimage.convolve(a.astype(np.float64), h.astype(np.float64), mode="constant", cval=0.0)
    return out.astype(np.float32)


def conv3d_replicate(a: np.ndarray, h: np.ndarray) -> np.ndarray:
    """``conv3d_gpu(a, h)``: same-size true convolution with the input index clamped
    to the array (conv3d_gpu.cu:77-98: offset ``d - k/2`` paired with the flipped kernel
    index ``k-1-d``, i.e. kernel sample m reads ``a[n + (k-1-k/2) - m]``: centre
    ``(k-1)/2`` -- for EVEN extents one sample before convn's 'same' centre ``k/2``,
    which is what ndimage uses; ``origin=-1`` on even axes moves it there, pinned by
    :func:`conv3d_replicate_loops`)."""
    origin = [-1 if k % 2 == 0 else 0 for k in h.shape]
    out = ndimage.convolve(a.astype(np.float64), h.astype(np.float64), mode="nearest", origin=origin)
    return out.astype(np.float32)


def conv3d_replicate_loops(a: np.ndarray, h: np.ndarray) -> np.ndarray:
    """Literal loop restatement of conv3d_gpu.cu:77-98 (small cases only); used by the
    tests to pin :func:`conv3d_replicate` independently of scipy's boundary code."""
    nz, ny, nx = a.shape
    kz, ky, kx = h.shape
    out = np.zeros(a.shape, np.float64)
    for dz in range(kz):
        iz = np.clip(np.arange(nz) + dz - kz // 2, 0, nz - 1)
        for dy in range(ky):
            iy = np.clip(np.arange(ny) + dy - ky // 2, 0, ny - 1)
            for dx in range(kx):
                ix = np.clip(np.arange(nx) + dx - kx // 2, 0, nx - 1)
                w = float(h[kz - 1 - dz, ky - 1 - dy, kx - 1 - dx])
                out += w * a[np.ix_(iz, iy, ix)].astype(np.float64)
    return out.astype(np.float32)


def flip3(psf: np.ndarray) -> np.ndarray:
    """``psf.inv = psf(end:-1:1,end:-1:1,end:-1:1)`` (LsDeconv.m:163)."""
    return np.ascontiguousarray(psf[::-1, ::-1, ::-1])


# --------------------------------------------------------------------------- Gaussian
def gaussian_taps(sigma: float, ksize: int) -> np.ndarray:
    """``make_gaussian_kernel`` (gauss3d_gpu.cu:81-90): sigma is a float, sigma*sigma is
    rounded in float, exp in double, stored as float, normalised by a double sum."""
    s = np.float32(sigma)
    s2 = float(np.float32(s * s))
    r = ksize // 2
    k = np.empty(ksize, np.float32)
    for i in range(-r, r + 1):
        k[i + r] = np.float32(math.exp(-0.5 * (i * i) / s2))
    total = float(np.sum(k.astype(np.float64)))
    return (k.astype(np.float64) / total).astype(np.float32)


def default_ksize(sigma) -> list[int]:
    """``ksize = 2*ceil(3*sigma)+1`` (gauss3d_gpu.cu:244-261)."""
    return [2 * int(math.ceil(3.0 * float(s))) + 1 for s in sigma]


def gauss3d(vol: np.ndarray, sigma, ksize=None) -> np.ndarray:
    """``gauss3d_gpu(x, sigma[, ksize])``: three 1-D passes X, Y, Z with replicate
    boundary (gauss3d_gpu.cu:93-138,163-192).  ``sigma``/``ksize`` are in reference
    order ``[x, y, z]``.  Each pass accumulates in float32 like the kernel does
    (we accumulate in float64 and round per pass: difference << 5e-5 test bound)."""
    sigma = [float(sigma)] * 3 if np.isscalar(sigma) else [float(s) for s in sigma]
    if ksize is None:
        ksize = default_ksize(sigma)
    elif np.isscalar(ksize):
        ksize = [int(ksize)] * 3
    out = vol.astype(np.float32)
    for ref_axis in range(3):  # 0 = x, 1 = y, 2 = z
        np_axis = 2 - ref_axis
        taps = gaussian_taps(sigma[ref_axis], int(ksize[ref_axis]))
        # symmetric taps: correlation == convolution
        out = ndimage.correlate1d(out.astype(np.float64), taps.astype(np.float64), axis=np_axis,
                                  mode="nearest").astype(np.float32)
    return out


# --------------------------------------------------------------------------- edge taper
def matlab_round(x: float) -> int:
    """MATLAB ``round``: half away from zero."""
    return int(math.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)


def make_taper(dimsz: int, taper_width: int) -> np.ndarray:
    """make_taper.m:13-35 (ramp of w+1 samples, plateau, mirrored ramp without its
    last sample, then cut/padded to ``dimsz``)."""
    w = min(int(taper_width), dimsz // 2)
    if w <= 0:
        return np.ones(dimsz, np.float32)
    ramp = np.linspace(0.0, 1.0, w + 1)
    down = ramp[:-1][::-1]
    if 2 * w < dimsz:
        t = np.concatenate([ramp, np.ones(dimsz - 2 * w), down])
    else:
        t = np.concatenate([ramp, down])
    t = t.astype(np.float32)
    if t.size > dimsz:
        t = t[:dimsz]
    elif t.size < dimsz:
        t = np.concatenate([t, np.ones(dimsz - t.size, np.float32)])
    return t


def taper_widths(psf_shape_zyx) -> list[int]:
    """``max(8, round(size(psf,d)/2))`` per axis (edgetaper_3d.m:32), numpy order."""
    return [max(8, matlab_round(k / 2.0)) for k in psf_shape_zyx]


def edgetaper_mask_vectors(shape_zyx, psf_shape_zyx):
    return [make_taper(n, w) for n, w in zip(shape_zyx, taper_widths(psf_shape_zyx))]


def edgetaper_3d(bl: np.ndarray, psf: np.ndarray) -> np.ndarray:
    """edgetaper_3d.m:13-44, GPU flavour (blur = conv3d_gpu with the sum-normalised PSF)."""
    psf = psf.astype(np.float32)
    assert np.all(np.isfinite(psf)) and np.all(psf >= 0)
    psfn = (psf / np.float32(psf.sum(dtype=np.float32))).astype(np.float32)
    blur = conv3d_replicate(bl, psfn)
    tz, ty, tx = edgetaper_mask_vectors(bl.shape, psf.shape)
    # mask = ((1 .* tx) .* ty) .* tz in single (edgetaper_3d.m:30-39, d = 1..3 -> x, y, z)
    mask = (tx[None, None, :] * ty[None, :, None]).astype(np.float32) * tz[:, None, None]
    one = np.float32(1.0)
    return (mask * bl.astype(np.float32) + (one - mask) * blur).astype(np.float32)


# --------------------------------------------------------------------------- RL loops
def is_regularization_time(i: int, niter: int, regularize_interval: int) -> bool:
    """decon.m:54-55 (i is 1-based)."""
    apply = (regularize_interval > 0) and (regularize_interval < niter)
    return bool(apply and (i > 1) and (i < niter) and (i % regularize_interval == 0))


def _reg_kernel() -> np.ndarray:
    r = np.full((3, 3, 3), np.float32(1.0 / 26.0), np.float32)  # decon.m:42
    r[1, 1, 1] = 0
    return r


def _norm2(a: np.ndarray) -> float:
    return float(np.sqrt(np.sum(a.astype(np.float64) ** 2)))


def decon_spatial(bl, psf, niter, lam=0.0, stop_criterion=0.0, regularize_interval=0,
                  psf_inv=None, gauss_flavour="gpu", return_iters=False, skip_edgetaper=False):
    """``deconSpatial`` (decon.m:26-124).  ``gauss_flavour``: "gpu" = gauss3d_gpu(bl,0.5)
    (5 taps), "cpu" = imgaussfilt3(bl,0.5) (3 taps, replicate) -- decon.m:58."""
    bl = bl.astype(np.float32)
    psf = psf.astype(np.float32)
    psf_inv = flip3(psf) if psf_inv is None else psf_inv.astype(np.float32)
    lam = np.float32(lam)
    R = _reg_kernel()
    delta_prev = _norm2(bl) if stop_criterion > 0 else 0.0
    if not skip_edgetaper:
        bl = edgetaper_3d(bl, psf)
    done = 0
    for i in range(1, niter + 1):
        reg = is_regularization_time(i, niter, regularize_interval)
        if reg:
            bl = gauss3d(bl, 0.5) if gauss_flavour == "gpu" else gauss3d(bl, 0.5, 3)
        buf = convn_same(bl, psf)
        buf = np.maximum(buf, EPS_SINGLE)
        buf = (bl / buf).astype(np.float32)
        buf = convn_same(buf, psf_inv)
        if reg and lam > 0:
            regv = convn_same(bl, R)
            buf = (bl * buf * (np.float32(1) - lam) + regv * lam).astype(np.float32)
        else:
            buf = (bl * buf).astype(np.float32)
        bl = np.abs(buf)
        done = i
        if stop_criterion > 0:
            cur = _norm2(bl)
            rel = abs(delta_prev - cur) / delta_prev * 100.0
            delta_prev = cur
            if i > 1 and rel <= stop_criterion:
                break
    return (bl, done) if return_iters else bl


def pad_block_to_fft_shape(bl, fft_shape_zyx):
    """decon.m:323-344 with mode 0 (zeros): pre = floor(missing/2), post = ceil."""
    missing = [f - s for f, s in zip(fft_shape_zyx, bl.shape)]
    assert all(m >= 0 for m in missing)
    pre = [m // 2 for m in missing]
    post = [m - p for m, p in zip(missing, pre)]
    return np.pad(bl, list(zip(pre, post))), pre, post


def unpad_block(bl, pre, post):
    """decon.m:346-374."""
    sl = tuple(slice(p, s - q) for p, q, s in zip(pre, post, bl.shape))
    return np.ascontiguousarray(bl[sl])


def otf_from_psf(psf, fft_shape_zyx, psf_grid_zyx=None):
    """``fftn(ifftshift(zero-pad-centre(psf)))`` (decon.m:131-133; otf_gpu.cu:36-67).

    ``psf_grid_zyx`` (no reference counterpart: ``mi_rl_options.psf_grid``): the PSF is placed on the ``fft_shape`` grid where
    ``ifftshift(zero-pad-centre(.))`` on a grid of THOSE extents puts it -- sample j of an axis at circular index
    j - (g // 2 - (g - k) // 2), which is what the two reference steps amount to (pre = floor((g - k) / 2) samples in front, then
    a rotation by floor(g / 2))."""
    if psf_grid_zyx is None:
        p, _, _ = pad_block_to_fft_shape(psf.astype(np.float32), fft_shape_zyx)
        return np.fft.fftn(np.fft.ifftshift(p).astype(np.float64))
    p = np.zeros(tuple(fft_shape_zyx), np.float32)
    idx = [(np.arange(k) - (g // 2 - (g - k) // 2)) % f for k, g, f in zip(psf.shape, psf_grid_zyx, fft_shape_zyx)]
    p[np.ix_(*idx)] = psf.astype(np.float32)
    return np.fft.fftn(p.astype(np.float64))


def decon_fft(bl, psf, fft_shape_zyx, niter, lam=0.0, stop_criterion=0.0, regularize_interval=0,
              gauss_flavour="gpu", return_iters=False, skip_edgetaper=False, psf_grid_zyx=None):
    """``deconFFT`` (decon.m:127-204): circular convolution on ``fft_shape``; float64
    transforms rounded to float32 at each ``real(ifftn(..))`` like the single-precision
    reference buffers."""
    bl = bl.astype(np.float32)
    psf = psf.astype(np.float32)
    lam = np.float32(lam)
    otf = otf_from_psf(psf, fft_shape_zyx, psf_grid_zyx)
    R = _reg_kernel()
    if not skip_edgetaper:
        bl = edgetaper_3d(bl, psf)
    bl, pre, post = pad_block_to_fft_shape(bl, fft_shape_zyx)
    delta_prev = _norm2(bl) if stop_criterion > 0 else 0.0
    done = 0
    for i in range(1, niter + 1):
        reg = is_regularization_time(i, niter, regularize_interval)
        if reg:
            bl = gauss3d(bl, 0.5) if gauss_flavour == "gpu" else gauss3d(bl, 0.5, 3)
        buf = np.real(np.fft.ifftn(np.fft.fftn(bl.astype(np.float64)) * otf)).astype(np.float32)
        buf = np.maximum(buf, EPS_SINGLE)
        buf = (bl / buf).astype(np.float32)
        buf = np.real(np.fft.ifftn(np.fft.fftn(buf.astype(np.float64)) * np.conj(otf))).astype(np.float32)
        if reg and lam > 0:
            regv = convn_same(bl, R)
            bl = (bl * buf * (np.float32(1) - lam) + regv * lam).astype(np.float32)
        else:
            bl = (bl * buf).astype(np.float32)
        bl = np.abs(bl)
        done = i
        if stop_criterion > 0:
            cur = _norm2(bl)
            rel = abs(delta_prev - cur) / delta_prev * 100.0
            delta_prev = cur
            if i > 1 and rel <= stop_criterion:
                break
    out = unpad_block(bl, pre, post)
    return (out, done) if return_iters else out


def otf_half_f32(psf, fft_shape_zyx, workers=1):
    """:func:`otf_from_psf` in single precision as the R2C half spectrum (scipy.fft, ``workers`` threads): the OTF of the
    timing leg :func:`decon_fft_f32`."""
    from scipy import fft as sfft
    p, _, _ = pad_block_to_fft_shape(psf.astype(np.float32), fft_shape_zyx)
    return sfft.rfftn(np.fft.ifftshift(p), workers=workers)


def decon_fft_f32(bl, otf_half, niter, workers=1):
    """The loop body of ``deconFFT`` (decon.m:162-186, lambda = 0, no regularisation, no stop test) with single-precision
    transforms on all host cores -- what the reference's MATLAB CPU path does with ``single`` arrays (multithreaded
    fftn).  ``bl`` already has the FFT shape.  bench.py's ``cpu_baseline`` times this; tests/test_oracle_rl.py holds it to
    :func:`decon_fft` (float64 transforms) within 1e-4."""
    from scipy import fft as sfft
    bl = bl.astype(np.float32)
    shape = bl.shape
    otf_c = np.conj(otf_half)
    for _ in range(niter):
        buf = sfft.irfftn(sfft.rfftn(bl, workers=workers) * otf_half, s=shape, workers=workers)
        np.maximum(buf, EPS_SINGLE, out=buf)
        np.divide(bl, buf, out=buf)
        buf = sfft.irfftn(sfft.rfftn(buf, workers=workers) * otf_c, s=shape, workers=workers)
        bl = np.abs(bl * buf)
    return bl


def decon_fft_wiener(bl, psf, fft_shape_zyx, niter, lam=0.0, stop_criterion=0.0, regularize_interval=0,
                     gauss_flavour="gpu", skip_edgetaper=False, return_psf=False, forced_psfs=None, trace=None):
    """``deconFFT_Wiener`` (decon.m:206-321): RL on ``fft_shape`` with a Wiener re-estimate of the PSF after every
    iteration but the last.  Quirks kept as written: the Gaussian pre-smoothing happens whenever
    ``regularize_interval > 0 and mod(i, interval) == 0`` (no ``1 < i < niter`` window, :252-256), the Tikhonov blend needs
    ``lambda > 0 and i < niter`` (:273), the new PSF is cut out of ``real(ifftn(otf_new))`` at the CENTRE of the array without
    an fftshift (:298-301), clamped at 0 and renormalised if its sum is positive, the stop test has no ``i > 1`` guard (:311-317).
    float64 transforms, float32 buffers.

    The PSF update is numerically unstable (the new PSF is the far field of a spectral quotient: a 1e-5 perturbation of the
    transforms grows ~100x per iteration), so implementations can only be compared step by step.  Test hooks for that:
    ``trace`` (a list) receives the PSF estimated after every iteration; ``forced_psfs`` ({iteration: psf}) replaces the
    PSF an iteration starts from, e.g. with the one another implementation estimated."""
    f32 = np.float32
    bl = bl.astype(f32)
    psf = psf.astype(f32)
    lam = f32(lam)
    R = _reg_kernel()
    if not skip_edgetaper:
        bl = edgetaper_3d(bl, psf)
    bl, pre, post = pad_block_to_fft_shape(bl, fft_shape_zyx)
    delta_prev = _norm2(bl) if stop_criterion > 0 else 0.0
    center = [(f - k) // 2 for f, k in zip(fft_shape_zyx, psf.shape)]     # floor((fft_shape - psf_sz)/2) + 1, 0-based here
    fy = None
    for i in range(1, niter + 1):
        if forced_psfs and i in forced_psfs:
            psf = forced_psfs[i].astype(f32)
        otf = otf_from_psf(psf, fft_shape_zyx)
        reg_i = regularize_interval > 0 and i % regularize_interval == 0
        if i == 1:
            fy = np.fft.fftn(bl.astype(np.float64))
        elif reg_i:
            bl = gauss3d(bl, 0.5) if gauss_flavour == "gpu" else gauss3d(bl, 0.5, 3)
            fy = np.fft.fftn(bl.astype(np.float64))
        buf = np.real(np.fft.ifftn(fy * otf)).astype(f32)
        buf = np.maximum(buf, EPS_SINGLE)
        buf = (bl / buf).astype(f32)
        buf = np.real(np.fft.ifftn(np.fft.fftn(buf.astype(np.float64)) * np.conj(otf))).astype(f32)
        if reg_i and lam > 0 and i < niter:
            regv = convn_same(bl, R)
            bl = (bl * buf * (f32(1) - lam) + regv * lam).astype(f32)
        else:
            bl = (bl * buf).astype(f32)
        bl = np.abs(bl)
        if i < niter:
            fx = np.fft.fftn(bl.astype(np.float64))
            den = np.maximum((fx * np.conj(fx)).real.astype(f32), EPS_SINGLE)
            otf_new = (fy * np.conj(fx)).astype(np.complex64) / den
            fy = fx
            full = np.real(np.fft.ifftn(otf_new.astype(np.complex128))).astype(f32)
            sl = tuple(slice(c, c + k) for c, k in zip(center, psf.shape))
            psf = np.maximum(full[sl], f32(0))
            tot = psf.sum(dtype=np.float64)
            if tot > 0:
                psf = (psf / f32(tot)).astype(f32)
            if trace is not None:
                trace.append(psf.copy())
        if stop_criterion > 0:
            cur = _norm2(bl)
            if abs(delta_prev - cur) / delta_prev * 100.0 <= stop_criterion:
                break
            delta_prev = cur
    out = unpad_block(bl, pre, post)
    return (out, psf) if return_psf else out


def decon(bl, psf, niter, lam, stop_criterion, regularize_interval, use_fft=False, fft_shape_zyx=None, **kw):
    """``decon`` dispatcher (decon.m:1-23), non-adaptive variants."""
    if use_fft:
        return decon_fft(bl, psf, fft_shape_zyx or bl.shape, niter, lam, stop_criterion, regularize_interval, **kw)
    return decon_spatial(bl, psf, niter, lam, stop_criterion, regularize_interval, **kw)


# --------------------------------------------------------------------------- block geometry
def split_stack(stack_xyz, block_xyz, nblocks_xyz):
    """split_stack.m:9-26: 1-based inclusive boxes, x fastest then y then z."""
    sx, sy, sz = stack_xyz
    bx, by, bz = block_xyz
    nx, ny, nz = nblocks_xyz
    p1, p2 = [], []
    for iz in range(nz):
        zs = iz * bz + 1
        for iy in range(ny):
            ys = iy * by + 1
            for ix in range(nx):
                xs = ix * bx + 1
                p1.append((xs, ys, zs))
                p2.append((min(xs + bx - 1, sx), min(ys + by - 1, sy), min(zs + bz - 1, sz)))
    return np.array(p1, np.int64), np.array(p2, np.int64)


def next_fast_len(n: int) -> int:
    """7-smooth size >= n (LsDeconv.m:405-419)."""
    while True:
        m = n
        for p in (2, 3, 5, 7):
            while m % p == 0:
                m //= p
        if m == 1:
            return n
        n += 1


def u16_to_f32(vol_u16: np.ndarray) -> np.ndarray:
    """``im2single`` of uint16 data (LsDeconv.m:860,873): x / 65535 in single."""
    return (vol_u16.astype(np.float32) / np.float32(65535.0)).astype(np.float32)


# --------------------------------------------------------------------------- synthetic inputs
def gaussian_psf(shape_zyx, sigma_zyx) -> np.ndarray:
    """Separable Gaussian PSF normalised to sum 1 (BASELINE config 1 input)."""
    axes = []
    for n, s in zip(shape_zyx, sigma_zyx):
        r = np.arange(n) - (n - 1) / 2.0
        axes.append(np.exp(-0.5 * (r / s) ** 2))
    p = axes[0][:, None, None] * axes[1][None, :, None] * axes[2][None, None, :]
    return (p / p.sum()).astype(np.float32)


def bead_volume(shape_zyx, seed=1234, psf=None) -> np.ndarray:
    """SURVEY.md section 8d synthetic volume: background U(0.01,0.02), N/4096 beads with
    amplitude U(0.2,1), blurred by the PSF, times (1 + N(0,0.01)) clipped >= 0."""
    rng = np.random.default_rng(seed)
    n = int(np.prod(shape_zyx))
    vol = rng.uniform(0.01, 0.02, size=shape_zyx).astype(np.float32)
    nb = max(1, n // 4096)
    idx = rng.integers(0, n, size=nb)
    vol.reshape(-1)[idx] += rng.uniform(0.2, 1.0, size=nb).astype(np.float32) * 20.0
    if psf is not None:
        vol = signal.fftconvolve(vol.astype(np.float64), psf.astype(np.float64), mode="same").astype(np.float32)
    vol *= (1.0 + rng.normal(0.0, 0.01, size=shape_zyx)).astype(np.float32)
    return np.clip(vol, 0.0, None).astype(np.float32)


def prctile(x, pct):
    """MATLAB ``prctile(x, pct, "all")`` (LsDeconv.m:1301): sorted sample i (1-based) is the 100 (i - 0.5) / n percentile,
    linear interpolation in between, clamped to min / max, NaN ignored -- numpy's "hazen" rule, spelled out."""
    v = np.sort(np.asarray(x, dtype=np.float32).ravel())
    v = v[~np.isnan(v)]
    n = v.size
    out = []
    for p in np.atleast_1d(pct):
        if n == 0:
            out.append(np.float32(np.nan))
            continue
        pos = min(max(float(p) / 100.0 * n - 0.5, 0.0), n - 1.0)
        lo = int(np.floor(pos))
        hi = min(lo + 1, n - 1)
        out.append(np.float32(np.float64(v[lo]) + (pos - lo) * (np.float64(v[hi]) - np.float64(v[lo]))))
    return out


def rescale_block(x, scal, ampl, dmin, dmax, dtype):
    """load_slab_lz4.cpp:134-157 in float32 arithmetic, in the reference's order of operations."""
    f = np.float32
    x = np.asarray(x, dtype=f)
    scal, ampl, dmin, dmax = f(scal), f(ampl), f(dmin), f(dmax)
    k_linear = f(f(scal * ampl) / dmax)
    if dmin > 0:
        k = f(f(scal * ampl) / f(dmax - dmin))
        val = (x - dmin).astype(f) * k
    else:
        val = x * k_linear
    val = (val.astype(f) - ampl).astype(f)
    val = np.where(val >= 0, np.floor(val + f(0.5)), np.ceil(val - f(0.5))).astype(f)
    val = np.clip(val, f(0), scal)
    return val.astype(dtype)
