"""Host-side mirror of the reference's RL deconvolution operators (LsDeconvolveMultiGPU).

Same names, argument meaning and error behaviour as the reference's MATLAB / MEX entry points:

    decon(bl, psf, niter, lambda_, stop_criterion, regularize_interval, device_id, use_fft, fft_shape,
          adaptive_psf)                                   decon.m:1
    conv3d_gpu(img, kernel)                               conv3d_gpu.cu:101-148
    gauss3d_gpu(x, sigma[, ksize])  (destructive)         gauss3d_gpu.cu:209-311
    edgetaper_3d(bl, psf)                                 edgetaper_3d.m:1
    otf_gpu(psf, fft_shape)                               supplements/otf_gpu.cu:69-150

Volumes are torch CUDA tensors (float32, C-contiguous, shape (Z, Y, X) == MATLAB [X,Y,Z]) -- the
analogue of a ``gpuArray`` -- or numpy arrays, which are uploaded, processed on the GPU and gathered
back (``gpuArray(bl)`` ... ``gather(bl)``, LsDeconv.m:914,941).  Vector arguments that the reference
takes in ``[x y z]`` order (sigma, ksize, fft_shape) keep that order here.  Everything runs through the
C ABI in ``include/mi_lsdeconv.h``; there is no CPU implementation in this package.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import numpy as np
import torch

from . import capi
from .capi import (BOUNDARY_CIRCULAR, BOUNDARY_REPLICATE, BOUNDARY_ZERO, ENGINE_AUTO, ENGINE_DIRECT, ENGINE_FFT,
                   RlOptions, check, lib)

__all__ = ["decon", "conv3d_gpu", "convn_same", "gauss3d_gpu", "edgetaper_3d", "otf_gpu", "im2single", "RLContext",
           "make_psf_struct", "norm2", "prctile", "rescale_block", "pad_block_to_fft_shape", "unpad_block", "next_fast_len", "engine_select"]


def _device(device_id=None) -> torch.device:
    """MATLAB device ids are 1-based (gpuDevice(gpu), LsDeconv.m:913); 0 means CPU in the reference, which
    this GPU-only build rejects."""
    capi.require_gpu()
    if device_id is None:
        return torch.device("cuda", torch.cuda.current_device())
    if isinstance(device_id, torch.device):
        return device_id
    if int(device_id) <= 0:
        raise ValueError("device_id 0 selects the reference's MATLAB CPU path, which this build does not contain")
    return torch.device("cuda", int(device_id) - 1)


def _to_dev(x, device, dtype=torch.float32, name="array"):
    """Returns (tensor_on_device, was_numpy)."""
    if isinstance(x, np.ndarray):
        a = np.ascontiguousarray(x)
        if any(st < 0 for st in a.strides):  # flipped axes of extent 1 keep their negative stride through ascontiguousarray
            a = a.copy(order="C")
        t = torch.from_numpy(a).to(device=device, dtype=dtype)
        return t, True
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"{name} must be a torch tensor or numpy array")
    if not x.is_cuda:
        return x.to(device=device, dtype=dtype).contiguous(), False
    if x.dtype != dtype:
        raise TypeError(f"{name} must be {dtype} (single), got {x.dtype}")
    if not x.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return x, False


def _check3d(t, name):
    if t.dim() != 3:
        raise ValueError(f"{name} must be 3D")  # conv3d_gpu:Input / gauss3d_gpu: "Input must be 3D."
    if t.numel() == 0:
        raise ValueError(f"{name} is empty")


def _stream(t):
    return capi.current_stream_ptr(t.device)


def _xyz(shape_zyx):
    return int(shape_zyx[2]), int(shape_zyx[1]), int(shape_zyx[0])


def make_psf_struct(psf):
    """``psf.psf`` / ``psf.inv = psf(end:-1:1,end:-1:1,end:-1:1)`` (LsDeconv.m:160-163)."""
    p = psf if isinstance(psf, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(psf, dtype=np.float32))
    return SimpleNamespace(psf=p, inv=torch.flip(p, dims=(0, 1, 2)).contiguous())


def im2single(u16, device_id=None):
    """``im2single`` of a uint16 block on the device (LsDeconv.m:860,873): x / 65535."""
    dev = _device(device_id)
    if isinstance(u16, np.ndarray):
        if u16.dtype != np.uint16:
            raise TypeError("im2single: expected uint16")
        # torch has limited uint16 support: move the bytes as int16 and reinterpret on the device
        src = torch.from_numpy(np.ascontiguousarray(u16).view(np.int16)).to(dev)
    else:
        src = u16.to(dev).contiguous()
        if src.dtype not in (torch.int16, torch.uint16):
            raise TypeError("im2single: expected a 16-bit integer tensor")
    out = torch.empty(src.shape, dtype=torch.float32, device=dev)
    check(lib().mi_u16_to_f32(dev.index, _stream(out), src.data_ptr(), out.data_ptr(), src.numel(), 1.0 / 65535.0))
    return out


def conv3d_gpu(img, kernel):
    """``out = conv3d_gpu(img, kernel)``: same-size convolution, replicate boundary (conv3d_gpu.cu:68-148)."""
    return convn_same(img, kernel, boundary=BOUNDARY_REPLICATE, engine=ENGINE_DIRECT)


def convn_same(img, kernel, boundary=BOUNDARY_ZERO, engine=ENGINE_AUTO):
    """``convn(img, kernel, 'same')`` (decon.m:61) with a selectable boundary rule / engine."""
    dev = _device(img.device if isinstance(img, torch.Tensor) and img.is_cuda else None)
    a, was_np = _to_dev(img, dev, name="img")
    k, _ = _to_dev(kernel, dev, name="kernel")
    _check3d(a, "Image")
    _check3d(k, "Kernel")
    out = torch.empty_like(a)
    nx, ny, nz = _xyz(a.shape)
    kx, ky, kz = _xyz(k.shape)
    check(lib().mi_conv3d(dev.index, _stream(a), a.data_ptr(), k.data_ptr(), out.data_ptr(), nx, ny, nz, kx, ky, kz,
                          boundary, engine))
    return out.cpu().numpy() if was_np else out


def _vec3(v, name, cast):
    if np.isscalar(v):
        return [cast(v)] * 3
    v = list(np.asarray(v).reshape(-1))
    if len(v) != 3:
        raise ValueError(f"{name} must be scalar or 3-vector")  # gauss3d_gpu.cu:239,257
    return [cast(x) for x in v]


def gauss3d_gpu(x, sigma, ksize=None):
    """``x = gauss3d_gpu(x, sigma[, kernel_size])``: overwrites a CUDA tensor in place and returns it;
    sigma / ksize in reference order [x y z] (gauss3d_gpu.cu:230-261)."""
    dev = _device(x.device if isinstance(x, torch.Tensor) and x.is_cuda else None)
    t, was_np = _to_dev(x, dev, name="x")
    _check3d(t, "Input")
    sig = (C.c_float * 3)(*_vec3(sigma, "sigma", float))
    ks = None
    if ksize is not None and not (isinstance(ksize, (list, tuple, np.ndarray)) and len(ksize) == 0):
        ks = (C.c_int * 3)(*_vec3(ksize, "kernel_size", int))
    work = torch.empty_like(t)
    nx, ny, nz = _xyz(t.shape)
    check(lib().mi_gauss3d_inplace(dev.index, _stream(t), t.data_ptr(), work.data_ptr(), nx, ny, nz, sig, ks))
    return t.cpu().numpy() if was_np else t


def filter_subband_3d_z(bl, sigma, levels=0, wavelet="db9"):
    """``bl = filter_subband_3d_z(bl, sigma, levels, wavelet)`` (filter_subband_3d_z.m:1-43): wavelet + Gaussian-notch destripe
    of every XZ slice; a CUDA tensor is modified in place.  Only "db9" exists here (what LsDeconv.m:935 passes)."""
    if str(wavelet) != "db9":
        raise ValueError(f"filter_subband_3d_z: wavelet {wavelet!r} is not built (LsDeconv.m:935 uses \"db9\")")
    dev = _device(bl.device if isinstance(bl, torch.Tensor) and bl.is_cuda else None)
    t, was_np = _to_dev(bl, dev, name="bl")
    _check3d(t, "bl")
    nx, ny, nz = _xyz(t.shape)
    check(lib().mi_destripe_z(dev.index, _stream(t), t.data_ptr(), nx, ny, nz, float(sigma), int(levels)))
    return t.cpu().numpy() if was_np else t


def edgetaper_3d(bl, psf):
    """``bl = edgetaper_3d(bl, psf)`` (edgetaper_3d.m:1-45); a CUDA tensor is modified in place."""
    dev = _device(bl.device if isinstance(bl, torch.Tensor) and bl.is_cuda else None)
    t, was_np = _to_dev(bl, dev, name="bl")
    p, _ = _to_dev(psf, dev, name="psf")
    _check3d(t, "bl")
    _check3d(p, "psf")
    if not bool(torch.isfinite(p).all()) or bool((p < 0).any()):
        raise AssertionError("PSF must be non-negative and finite")  # edgetaper_3d.m:13
    work = torch.empty_like(t)
    nx, ny, nz = _xyz(t.shape)
    kx, ky, kz = _xyz(p.shape)
    check(lib().mi_edgetaper3d(dev.index, _stream(t), t.data_ptr(), work.data_ptr(), p.data_ptr(), nx, ny, nz, kx, ky, kz))
    return t.cpu().numpy() if was_np else t


def otf_gpu(psf, fft_shape, scale=1.0):
    """``otf = otf_gpu(psf, [nx ny nz])`` (supplements/otf_gpu.cu:69-150) as the R2C half spectrum: a
    complex64 tensor of shape (fz, fy, fx//2+1)."""
    dev = _device(psf.device if isinstance(psf, torch.Tensor) and psf.is_cuda else None)
    p, _ = _to_dev(psf, dev, name="psf")
    _check3d(p, "psf")
    fx, fy, fz = (int(v) for v in fft_shape)
    kx, ky, kz = _xyz(p.shape)
    if fx < kx or fy < ky or fz < kz:
        raise ValueError("fft_shape must be >= psf size in every dimension")  # otf_gpu.cu:114-118
    out = torch.empty((fz, fy, fx // 2 + 1), dtype=torch.complex64, device=dev)
    check(lib().mi_otf(dev.index, _stream(p), p.data_ptr(), kx, ky, kz, out.data_ptr(), fx, fy, fz, float(scale)))
    return out


def norm2(x) -> float:
    """``norm(bl(:))`` with fp64 accumulation (decon.m:47,109)."""
    out = C.c_double()
    check(lib().mi_norm2(x.device.index, _stream(x), x.data_ptr(), x.numel(), C.byref(out)))
    return out.value


def prctile(x, pct):
    """``prctile(x, pct, "all")`` of a device volume (LsDeconv.m:1301): exact, MATLAB's interpolation rule; ``pct`` = one or
    two percentiles in [0, 100]; returns a list of floats."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
        raise ValueError("prctile: x must be a contiguous float32 device tensor")
    p = [float(v) for v in (pct if hasattr(pct, "__len__") else [pct])]
    if not 1 <= len(p) <= 2:
        raise ValueError("prctile: one or two percentiles per call")
    pa = (C.c_double * len(p))(*p)
    out = (C.c_float * len(p))()
    check(lib().mi_prctile(x.device.index, _stream(x), x.data_ptr(), x.numel(), pa, len(p), out))
    return [float(v) for v in out]


def rescale_block(x, scal, ampl, dmin, dmax, out=None):
    """The float -> uint8/uint16 conversion of ``load_slab_lz4`` (load_slab_lz4.cpp:134-157): rescale by the global
    [dmin, dmax] of the deconvolved stack, amplification, round half away from zero, clamp to [0, scal].  ``scal`` <= 255
    gives uint8, else uint16 (LsDeconv.m:1017-1024)."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
        raise ValueError("rescale_block: x must be a contiguous float32 device tensor")
    bits = 8 if scal <= 255 else 16
    dt = torch.uint8 if bits == 8 else torch.uint16
    if out is None:
        out = torch.empty(x.shape, dtype=dt, device=x.device)
    elif out.dtype != dt or out.shape != x.shape or not out.is_contiguous() or out.device != x.device:
        raise ValueError("rescale_block: out must be a contiguous device tensor of the target integer type and x's shape")
    check(lib().mi_rescale_block(x.device.index, _stream(x), x.data_ptr(), out.data_ptr(), x.numel(), bits, float(scal),
                                 float(ampl), float(dmin), float(dmax)))
    return out


def next_fast_len(n: int) -> int:
    """LsDeconv.m:405-419."""
    return int(lib().mi_next_fast_len(int(n)))


def engine_select(shape_zyx, psf_shape_zyx, boundary=BOUNDARY_ZERO) -> int:
    nx, ny, nz = _xyz(shape_zyx)
    kx, ky, kz = _xyz(psf_shape_zyx)
    return int(lib().mi_engine_select(nx, ny, nz, kx, ky, kz, boundary))


def pad_block_to_fft_shape(bl, fft_shape):
    """decon.m:323-344 with mode 0; ``fft_shape`` in [x y z] order.  Returns (padded, pad_pre, pad_post) with the
    pads in [x y z] order."""
    nx, ny, nz = _xyz(bl.shape)
    fx, fy, fz = (int(v) for v in fft_shape)
    if fx < nx or fy < ny or fz < nz:
        raise AssertionError(f"pad_block_to_fft_shape: bl [{nx} {ny} {nz}] is larger than FFT shape "
                             f"[{fx} {fy} {fz}], cannot pad")
    out = torch.empty((fz, fy, fx), dtype=torch.float32, device=bl.device)
    check(lib().mi_pad_center(bl.device.index, _stream(bl), bl.data_ptr(), nx, ny, nz, out.data_ptr(), fx, fy, fz))
    pre = [(fx - nx) // 2, (fy - ny) // 2, (fz - nz) // 2]
    post = [fx - nx - pre[0], fy - ny - pre[1], fz - nz - pre[2]]
    return out, pre, post


def unpad_block(bl, pad_pre, pad_post):
    """decon.m:346-374."""
    fx, fy, fz = _xyz(bl.shape)
    nx, ny, nz = (f - a - b for f, a, b in zip((fx, fy, fz), pad_pre, pad_post))
    if min(nx, ny, nz) < 1:
        raise ValueError("unpad_block: Output block size is empty in at least one dimension!")
    out = torch.empty((nz, ny, nx), dtype=torch.float32, device=bl.device)
    check(lib().mi_crop_center(bl.device.index, _stream(bl), bl.data_ptr(), fx, fy, fz, out.data_ptr(), nx, ny, nz))
    return out


class RLContext:
    """The two fused half-steps of one RL iteration for arrays of a fixed shape (``mi_rl_create``): what the
    multi-GPU slab driver calls between halo exchanges."""

    def __init__(self, shape_zyx, psf, psf_inv=None, boundary=BOUNDARY_ZERO, engine=ENGINE_AUTO, device=None,
                 shift_xyz=None):
        """``boundary``: one mi_boundary or a triple (x, y, z); ``shift_xyz``: optional PSF placement per axis
        (mi_rl_create_ex)."""
        self.device = _device(device)
        self.shape = tuple(int(s) for s in shape_zyx)
        p, _ = _to_dev(psf, self.device, name="psf")
        pi = None
        if psf_inv is not None:
            pi, _ = _to_dev(psf_inv, self.device, name="psf_inv")
        nx, ny, nz = _xyz(self.shape)
        kx, ky, kz = _xyz(p.shape)
        self._h = C.c_void_p()
        if np.isscalar(boundary) and shift_xyz is None:
            check(lib().mi_rl_create(self.device.index, capi.current_stream_ptr(self.device), nx, ny, nz, p.data_ptr(),
                                     pi.data_ptr() if pi is not None else None, kx, ky, kz, int(boundary), engine,
                                     C.byref(self._h)))
        else:
            b = [int(boundary)] * 3 if np.isscalar(boundary) else [int(v) for v in boundary]
            sh = [-1, -1, -1] if shift_xyz is None else [int(v) for v in shift_xyz]
            check(lib().mi_rl_create_ex(self.device.index, capi.current_stream_ptr(self.device), nx, ny, nz,
                                        p.data_ptr(), pi.data_ptr() if pi is not None else None, kx, ky, kz,
                                        (C.c_int * 3)(*b), (C.c_int * 3)(*sh), engine, C.byref(self._h)))
        self.engine = int(lib().mi_rl_engine(self._h))
        self.device_bytes = int(lib().mi_rl_device_bytes(self._h))

    def _chk(self, t):
        if tuple(t.shape) != self.shape or t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
            raise ValueError(f"expected a contiguous float32 tensor of shape {self.shape} on {self.device}")

    def forward_ratio(self, bl, ratio):
        """ratio = bl ./ max(conv(bl, psf), eps)   (decon.m:61-63)"""
        self._chk(bl)
        self._chk(ratio)
        check(lib().mi_rl_forward_ratio(self._h, _stream(bl), bl.data_ptr(), ratio.data_ptr()))

    def adjoint_update(self, ratio, bl, lambda_=0.0, reg=None):
        """bl = abs(bl .* conv(ratio, psf_inv)) [Tikhonov blend with reg]   (decon.m:64-79)"""
        self._chk(bl)
        self._chk(ratio)
        check(lib().mi_rl_adjoint_update(self._h, _stream(bl), ratio.data_ptr(), bl.data_ptr(), float(lambda_),
                                         reg.data_ptr() if reg is not None else None))

    def iterate(self, bl, ratio=None, n_iters=1):
        """``n_iters`` plain RL iterations on ``bl`` in place (mi_rl_iterate); ``ratio`` is scratch for engines
        that cannot fuse the two convolutions."""
        self._chk(bl)
        if ratio is not None:
            self._chk(ratio)
        check(lib().mi_rl_iterate(self._h, _stream(bl), bl.data_ptr(), ratio.data_ptr() if ratio is not None else None,
                                  int(n_iters)))

    # ---- sharded fused iteration (slab driver): see include/mi_lsdeconv.h "Sharded fused iteration"
    @property
    def fuses(self) -> int:
        """0: no fused iteration; 1: fused; 2: fused, and the x pass can be split around the halo sends (``part``)."""
        return int(lib().mi_rl_fuses(self._h))

    @property
    def separable(self) -> bool:
        """The direct engine found the PSF to be an outer product and applies it as three 1-D convolutions."""
        return bool(lib().mi_rl_separable(self._h))

    @property
    def separable_single_pass(self) -> bool:
        """... and does so in one pass over the volume (``mi_rl_separable`` == 2)."""
        return int(lib().mi_rl_separable(self._h)) == 2

    @property
    def pair_layout(self) -> bool:
        """The FFT engine keeps the spectra around its z pass pair-interleaved (``mi_rl_pair_layout``)."""
        return bool(lib().mi_rl_pair_layout(self._h))

    @property
    def otf_is_real(self) -> bool:
        return bool(lib().mi_rl_otf_is_real(self._h))

    @property
    def spectrum_row_floats(self) -> int:
        """float32 words of one row (all z, all x frequencies) of the x-transformed input buffer."""
        return int(lib().mi_rl_spectrum_row_floats(self._h))

    def sharded_begin(self, bl):
        self._chk(bl)
        check(lib().mi_rl_sharded_begin(self._h, _stream(bl), bl.data_ptr()))

    @staticmethod
    def _edges(edge_rows):
        return (C.c_int * 4)(*[int(v) for v in edge_rows]) if edge_rows is not None else None

    def sharded_ratio(self, bl, part=0, edge_rows=None):
        """S <- x-forward(bl ./ max(conv(S), eps)).  part 1: y/z passes + the x tiles holding ``edge_rows`` =
        (a0, a1, b0, b1); part 2: the remaining x tiles."""
        self._chk(bl)
        check(lib().mi_rl_sharded_ratio(self._h, _stream(bl), bl.data_ptr(), int(part), self._edges(edge_rows)))

    def sharded_update(self, bl, more=True, part=0, edge_rows=None):
        self._chk(bl)
        check(lib().mi_rl_sharded_update(self._h, _stream(bl), bl.data_ptr(), int(bool(more)), int(part), self._edges(edge_rows)))

    def set_overlap(self, free_cus=0, dynamic_tiles=False):
        """Launch geometry of the part-2 x launches that run beside a halo exchange (``mi_rl_set_overlap``)."""
        check(lib().mi_rl_set_overlap(self._h, int(free_cus), int(bool(dynamic_tiles))))

    def overlap_probe(self, bl, edge_rows, busy_wgs=8, busy_us=1000.0, reps=5):
        """(ms of the part-2 x launch, ms until it and a stand-in for a collective's kernels have both finished)"""
        self._chk(bl)
        out = (C.c_float * 2)()
        check(lib().mi_rl_overlap_probe(self._h, _stream(bl), bl.data_ptr(), self._edges(edge_rows), int(busy_wgs), float(busy_us),
                                        int(reps), out))
        return float(out[0]), float(out[1])

    @property
    def z_granule(self) -> int:
        """z chunks of the staged sharded step must be cut at multiples of this (0: the context cannot run it)."""
        return int(lib().mi_rl_z_granule(self._h))

    def sharded_stage(self, bl, update, stage, z0=0, z1=0, edge_rows=None):
        """One stage of a half-step of the z-chunked exchange (``mi_rl_sharded_stage``): 0 y-forward of the planes [z0, z1),
        1 z pass + y-inverse, 2 edge tiles of the planes [z0, z1), 3 all other tiles."""
        self._chk(bl)
        check(lib().mi_rl_sharded_stage(self._h, _stream(bl), bl.data_ptr(), int(bool(update)), int(stage), int(z0), int(z1),
                                        self._edges(edge_rows)))

    def spectrum_rows_z(self, buf, y0, rows, z0, z1, direction):
        """Pack (0) / unpack (1) / zero (2) the rows [y0, y0 + rows) of the planes [z0, z1); ``buf`` is the WHOLE packed buffer (a
        float32 device tensor or a raw device pointer), of which the chunk keeps its place."""
        ptr = None if buf is None else (buf.data_ptr() if isinstance(buf, torch.Tensor) else int(buf))
        check(lib().mi_rl_spectrum_rows_z(self._h, capi.current_stream_ptr(self.device), int(y0), int(rows), int(z0), int(z1), ptr,
                                          int(direction)))

    def spectrum_pack(self, y0, rows, out=None):
        """Rows [y0, y0+rows) of the x-transformed input buffer as a contiguous float32 device tensor (``out``: written there)."""
        n = int(rows) * int(lib().mi_rl_spectrum_row_floats(self._h))
        buf = torch.empty(n, dtype=torch.float32, device=self.device) if out is None else out
        if not (buf.is_cuda and buf.dtype == torch.float32 and buf.is_contiguous() and buf.numel() == n):
            raise ValueError("spectrum_pack: `out` does not hold `rows` spectrum rows")
        check(lib().mi_rl_spectrum_rows(self._h, _stream(buf), int(y0), int(rows), buf.data_ptr(), 0))
        return buf

    def spectrum_unpack_ptr(self, ptr, y0, rows):
        """``spectrum_unpack`` from a raw device pointer (a receive buffer of the copy-engine transport)."""
        check(lib().mi_rl_spectrum_rows(self._h, capi.current_stream_ptr(self.device), int(y0), int(rows), int(ptr), 1))

    def spectrum_unpack(self, buf, y0, rows):
        if buf is None:
            check(lib().mi_rl_spectrum_rows(self._h, capi.current_stream_ptr(self.device), int(y0), int(rows), None, 2))
            return
        if not (buf.is_cuda and buf.dtype == torch.float32 and buf.is_contiguous()
                and buf.numel() == int(rows) * int(lib().mi_rl_spectrum_row_floats(self._h))):
            raise ValueError("spectrum_unpack: buffer does not hold `rows` spectrum rows")
        check(lib().mi_rl_spectrum_rows(self._h, _stream(buf), int(y0), int(rows), buf.data_ptr(), 1))

    PASSES = {"x_forward": 0, "y_forward": 1, "z_conv": 2, "y_inverse": 3, "x_fused": 4, "x_fused_update": 5}

    def time_pass(self, which, bl, reps=5) -> float:
        """Average ms of one launch of a single pass of the native FFT pipeline (HIP events, mi_rl_time_pass).
        ``x_fused_update`` overwrites ``bl`` with meaningless values."""
        self._chk(bl)
        ms = C.c_float()
        check(lib().mi_rl_time_pass(self._h, _stream(bl), self.PASSES[which], bl.data_ptr(), int(reps), C.byref(ms)))
        return float(ms.value)

    def fft_placement(self):
        """(costs in ms of the candidate placements of the spectrum arrays -- the ordered pairs (S, T) of the buffers tried --, index of
        the kept one): mi_rl_fft_placement.
        ([], -1) for a plain allocation."""
        cost = (C.c_float * 64)()           # (ordered pairs of up to eight buffers)
        n, kept = C.c_int(), C.c_int()
        check(lib().mi_rl_fft_placement(self._h, cost, 64, C.byref(n), C.byref(kept)))
        return [round(float(cost[i]), 3) for i in range(min(n.value, 64))], int(kept.value)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().mi_rl_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeconPlan:
    """What ``decon`` would rebuild for every block of a volume -- the RL context (OTF, twiddles, scratch) and the FFT engine of
    the edge taper's blur -- kept between calls (``mi_decon_plan``).  Pass it as ``decon(..., plan=plan)``; shape, options and
    PSF values are compared on every call and the kept objects rebuilt when they differ, results are bit-identical to a call
    without a plan.  One plan per worker thread."""

    def __init__(self, device=None):
        self.device = _device(device)
        h = C.c_void_p()
        check(lib().mi_decon_plan_create(self.device.index, C.byref(h)))
        self.handle = h

    def close(self):
        if self.handle is not None:
            check(lib().mi_decon_plan_destroy(self.handle))
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decon(bl, psf, niter, lambda_=0.0, stop_criterion=0.0, regularize_interval=0, device_id=None, use_fft=False,
          fft_shape=None, adaptive_psf=False, *, engine=ENGINE_AUTO, skip_edgetaper=False, gauss_taps=0,
          return_iters=False, return_psf=False, plan=None, psf_grid=None):
    """``bl = decon(bl, psf, niter, lambda, stop_criterion, regularize_interval, device_id, use_fft, fft_shape,
    adaptive_psf)`` (decon.m:1-23).  ``adaptive_psf`` with ``use_fft`` runs ``deconFFT_Wiener`` (decon.m:206-321), whose
    ``fft_shape`` must be made of extents the hand-written FFT takes (``fft_good_size``); ``return_psf`` then also hands
    back the refined PSF, which the reference keeps local.

    ``psf`` is the reference's struct with fields ``psf`` and ``inv`` (anything with those attributes / keys), or
    a bare array (``inv`` is then the flipped PSF).  ``fft_shape`` is [x y z].  A CUDA tensor ``bl`` is updated
    in place and returned; a numpy ``bl`` is uploaded and gathered.  Keyword-only extras select the convolution
    engine and let a caller that already tapered the block skip the taper (used by the benchmark).  ``psf_grid`` [x y z]
    (deconFFT): the grid whose parity decides where the PSF's centre sample lands (``mi_rl_options.psf_grid``; default: ``fft_shape``,
    the reference's rule) -- for a caller that enlarges the reference's ``next_fast_len`` grid and wants the reference's placement."""
    if adaptive_psf and not use_fft:
        raise ValueError("--adaptive-psf requires --use-fft")  # decwrap.py:216-217
    dev = _device(device_id if device_id is not None else
                  (bl.device if isinstance(bl, torch.Tensor) and bl.is_cuda else None))
    if isinstance(psf, dict):
        p_fwd, p_inv = psf["psf"], psf.get("inv")
    elif hasattr(psf, "psf"):
        p_fwd, p_inv = psf.psf, getattr(psf, "inv", None)
    else:
        p_fwd, p_inv = psf, None
    t, was_np = _to_dev(bl, dev, name="bl")
    _check3d(t, "bl")
    p, _ = _to_dev(p_fwd, dev, name="psf")
    _check3d(p, "psf")
    pi_ptr = None
    if p_inv is not None and not use_fft:
        pi, _ = _to_dev(p_inv, dev, name="psf.inv")
        # the struct built by LsDeconv.m:163 always holds the flipped PSF: let the library use its implied adjoint
        if not torch.equal(pi, torch.flip(p, dims=(0, 1, 2))):
            pi_ptr = pi.data_ptr()
    nx, ny, nz = _xyz(t.shape)
    kx, ky, kz = _xyz(p.shape)
    opt = RlOptions(int(niter), float(lambda_), float(stop_criterion), int(regularize_interval), int(engine),
                    1 if skip_edgetaper else 0, int(gauss_taps))
    if psf_grid is not None:
        if not use_fft or adaptive_psf:
            raise ValueError("decon: psf_grid applies to deconFFT only")
        opt.psf_grid[:] = [int(v) for v in psf_grid]
    done = C.c_int(0)
    fs = None
    if use_fft:
        fs = (C.c_int * 3)(*(int(v) for v in (fft_shape if fft_shape is not None else (nx, ny, nz))))
    refined = None
    if adaptive_psf:
        refined = p.clone()
        check(lib().mi_rl_fft_wiener(dev.index, _stream(t), t.data_ptr(), refined.data_ptr(), nx, ny, nz, kx, ky, kz,
                                     fs[0], fs[1], fs[2], C.byref(opt), C.byref(done)))
    elif plan is not None:
        if plan.device != dev:
            raise ValueError(f"decon: the plan lives on {plan.device}, the block on {dev}")
        check(lib().mi_decon_plan_run(plan.handle, _stream(t), t.data_ptr(), p.data_ptr(), pi_ptr, nx, ny, nz, kx, ky, kz,
                                      C.byref(opt), 1 if use_fft else 0, fs, 0, C.byref(done)))
    else:
        check(lib().mi_decon(dev.index, _stream(t), t.data_ptr(), p.data_ptr(), pi_ptr, nx, ny, nz, kx, ky, kz,
                             C.byref(opt), 1 if use_fft else 0, fs, 0, C.byref(done)))
    out = t.cpu().numpy() if was_np else t
    res = (out,) + ((done.value,) if return_iters else ()) + ((refined.cpu().numpy() if was_np else refined,) if return_psf else ())
    return res if len(res) > 1 else out
