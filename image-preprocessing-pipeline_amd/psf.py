"""Theoretical light-sheet PSF (host-side input generator of the RL path).

Mirrors ``generate_psf`` of the reference's LsDeconvolveMultiGPU/psf_generator.py:50-121 (same
name, arguments and return value) and, with ``flavour="matlab"``, ``LsMakePSF.m:2-9``.  Written
from the published formula -- PSF(x,y,z) = PSF_sheet(z,0,x; NA_ls, lambda_ex) * PSF_obj(x,y,z; NA,
lambda_em) with PSF = 4*|int_0^1 J0(2 pi NA r p/(lambda n)) exp(-i pi p^2 z NA^2/(lambda n^2)) p dp|^2
(LsMakePSF.m:100-115) -- using one vectorised Gauss-Legendre rule for the whole octant instead of one
adaptive quadrature per voxel.  Array order is (z, y, x).
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import j0

_GL_NODES, _GL_WEIGHTS = np.polynomial.legendre.leggauss(96)
_P = 0.5 * (_GL_NODES + 1.0)          # nodes on [0, 1]
_W = 0.5 * _GL_WEIGHTS


def psf_eq(x, y, z, numerical_aperture, refractive_index, lambda_val):
    """4*|integral|^2 for arrays x, y, z (broadcast) -- psf_generator.py:25-40 / LsMakePSF.m:105-115."""
    x, y, z = np.broadcast_arrays(np.asarray(x, np.float64), np.asarray(y, np.float64), np.asarray(z, np.float64))
    r = np.sqrt(x * x + y * y)[..., None]
    a = 2.0 * math.pi * numerical_aperture * r / (lambda_val * refractive_index)
    b = -math.pi * z[..., None] * numerical_aperture ** 2 / (lambda_val * refractive_index ** 2)
    integrand = j0(a * _P) * np.exp(1j * b * _P * _P) * _P
    integral = np.sum(integrand * _W, axis=-1)
    return 4.0 * np.abs(integral) ** 2


def ls_psf_eq(x, y, z, numerical_aperture_obj, refractive_index, lambda_ex, lambda_em, numerical_aperture_ls):
    """psf_generator.py:43-48 / LsMakePSF.m:100-102."""
    return (psf_eq(z, 0.0, x, numerical_aperture_ls, refractive_index, lambda_ex)
            * psf_eq(x, y, z, numerical_aperture_obj, refractive_index, lambda_em))


def _first_root(f, start):
    """Root of f nearest to ``start`` (the half-maximum crossing the reference finds with
    fsolve/fzero from a start inside the main lobe).  Bracket outward from start, then bisect."""
    lo = hi = float(start)
    flo = fhi = float(f(start))
    step = abs(start) * 0.05 + 1.0
    for _ in range(400):
        if flo * fhi <= 0.0 and lo != hi:
            break
        if flo > 0:  # above half maximum: the crossing is further out
            hi += step
            fhi = float(f(hi))
        else:        # below: the crossing is further in
            lo = max(0.0, lo - step)
            flo = float(f(lo))
    else:
        raise RuntimeError("PSF half-maximum crossing not bracketed")
    if lo > hi:
        lo, hi, flo, fhi = hi, lo, fhi, flo
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        fm = float(f(mid))
        if (fm > 0) == (flo > 0):
            lo, flo = mid, fm
        else:
            hi, fhi = mid, fm
        if hi - lo < 1e-9 * max(1.0, abs(hi)):
            break
    return 0.5 * (lo + hi)


def determine_psf_size(dxy_psf, dz, numerical_aperture, refractive_index, lambda_ex, lambda_em, f_cylinder_lens,
                       slit_width, resolution_xy, resolution_z):
    """psf_generator.py:124-155 / LsMakePSF.m:12-39: grid = 2 x FWHM, forced odd."""
    na_ls = math.sin(math.atan(0.5 * slit_width / f_cylinder_lens))
    half_max = 0.5 * float(ls_psf_eq(0.0, 0.0, 0.0, numerical_aperture, refractive_index, lambda_ex, lambda_em, na_ls))

    def fxy(x):
        return ls_psf_eq(x, 0.0, 0.0, numerical_aperture, refractive_index, lambda_ex, lambda_em, na_ls) - half_max

    def fz(x):
        return ls_psf_eq(0.0, 0.0, x, numerical_aperture, refractive_index, lambda_ex, lambda_em, na_ls) - half_max

    fwhm_xy = 2.0 * abs(_first_root(fxy, resolution_xy / 2.0))
    fwhm_z = 2.0 * abs(_first_root(fz, resolution_z / 2.0))
    nxy = math.ceil(2 * fwhm_xy / dxy_psf)
    nz = math.ceil(2 * fwhm_z / dz)
    nxy += 1 - nxy % 2
    nz += 1 - nz % 2
    return nxy, nz, fwhm_xy, fwhm_z


def mirror8(octant: np.ndarray) -> np.ndarray:
    """First octant -> full PSF of size 2n-1 per axis (psf_generator.py:196-212 / LsMakePSF.m:67-83)."""
    full = octant
    for ax in range(3):
        full = np.concatenate([np.flip(full, axis=ax), np.take(full, range(1, full.shape[ax]), axis=ax)], axis=ax)
    return np.ascontiguousarray(full.astype(np.float32))


def sample_psf(dxy, dz, nxy, nz, numerical_aperture_obj, rf, lambda_ex, lambda_em, numerical_aperture_ls,
               gaussian_sigma=0.0, doubling_effect=False):
    """psf_generator.py:158-193 / LsMakePSF.m:41-65."""
    if nxy % 2 == 0 or nz % 2 == 0:
        raise RuntimeError(f"sample_psf: nxy is {nxy} and nz is {nz}, but must be odd!")
    hz, hxy = (nz - 1) // 2 + 1, (nxy - 1) // 2 + 1
    z = (np.arange(hz) * dz)[:, None, None]
    y = (np.arange(hxy) * dxy)[None, :, None]
    x = (np.arange(hxy) * dxy)[None, None, :]
    octant = ls_psf_eq(x, y, z, numerical_aperture_obj, rf, lambda_ex, lambda_em, numerical_aperture_ls)
    psf = mirror8(octant.astype(np.float32))
    if gaussian_sigma > 0:
        from scipy.ndimage import gaussian_filter
        sigma = (gaussian_sigma, gaussian_sigma, round(gaussian_sigma, 0) + (2.0 if doubling_effect else 1.5))
        psf = gaussian_filter(psf, sigma=sigma).astype(np.float32)
    if doubling_effect:
        psf = np.concatenate([psf, psf], axis=0)
    psf /= psf.sum(dtype=np.float32)
    return psf


def generate_psf(lambda_em: float = 642.0, lambda_ex: float = 680.0, numerical_aperture: float = 0.4,
                 dxy: float = 422.0, dz: float = 1000.0, refractive_index: float = 1.42,
                 f_cylinder_lens: float = 240.0, slit_width: float = 12.0, gaussian_sgima: float = 0,
                 doubled_psf: bool = False, flavour: str = "python"):
    """Returns ``(psf[z,y,x] float32 with sum 1, dxy_psf)`` like psf_generator.generate_psf (:50-121).

    ``flavour="matlab"`` follows LsMakePSF.m instead: the octant is sampled at the *camera* pixel
    pitch ``dxy`` while the grid size is computed from the corrected pitch (LsMakePSF.m:3-7,26-29).
    """
    resolution_xy = 0.61 * lambda_em / numerical_aperture
    resolution_z = 2.0 * lambda_ex * refractive_index / numerical_aperture ** 2
    dxy_psf = min(dxy, resolution_xy / 3)
    nxy, nz, _, _ = determine_psf_size(dxy_psf, dz, numerical_aperture, refractive_index, lambda_ex, lambda_em,
                                       f_cylinder_lens, slit_width, resolution_xy, resolution_z)
    na_ls = math.sin(math.atan(slit_width / (2.0 * f_cylinder_lens)))
    sample_dxy = dxy if flavour == "matlab" else dxy_psf
    psf = sample_psf(sample_dxy, dz, nxy, nz, numerical_aperture, refractive_index, lambda_ex, lambda_em, na_ls,
                     gaussian_sigma=gaussian_sgima, doubling_effect=doubled_psf)
    return psf, dxy_psf


def LsMakePSF(dxy, dz, NA, nf, lambda_ex, lambda_em, fcyl, slitwidth):
    """MATLAB entry point (LsMakePSF.m:2): returns psf in (z, y, x) order."""
    return generate_psf(lambda_em=lambda_em, lambda_ex=lambda_ex, numerical_aperture=NA, dxy=dxy, dz=dz,
                        refractive_index=nf, f_cylinder_lens=fcyl, slit_width=slitwidth, flavour="matlab")[0]


def resample_psf(psf: np.ndarray, shape_zyx) -> np.ndarray:
    """Centre crop / zero-pad a PSF to an odd target extent and renormalise (used to build the
    BASELINE.json PSF extents 15x15x31, 31x31x61, 63x63x127 from one physical model)."""
    out = np.zeros(shape_zyx, np.float32)
    src, dst = [], []
    for n_src, n_dst in zip(psf.shape, shape_zyx):
        c = min(n_src, n_dst)
        s0, d0 = (n_src - c) // 2, (n_dst - c) // 2
        src.append(slice(s0, s0 + c))
        dst.append(slice(d0, d0 + c))
    out[tuple(dst)] = psf[tuple(src)]
    return out / out.sum(dtype=np.float32)
