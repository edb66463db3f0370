"""Block geometry and per-block processing around ``decon`` (host logic of LsDeconv.m that defines the inputs
of the hot path: SURVEY.md R9 and the ``process_block`` row of section 8f).

    split_stack(stack_info, block)           split_stack.m:1-27
    decon_pad_size / gaussian_pad_size       LsDeconv.m:387-403
    next_fast_len                            LsDeconv.m:405-419
    autosplit(...)                           LsDeconv.m:308-385 (re-targeted to HBM capacity, Appendix C of SURVEY.md)
    load_block(volume, p1, p2, pad)          LsDeconv.m:817-904 (in-memory volume; symmetric fill at the volume edge)
    process_block(bl, ...)                   LsDeconv.m:906-948

Coordinates follow the reference: 1-based inclusive boxes in [x y z] order; arrays are (Z, Y, X).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch

from . import capi, decon as D


@dataclass
class Block:
    """``block`` struct of LsDeconv.m:204-208."""
    x: int
    y: int
    z: int
    nx: int
    ny: int
    nz: int
    x_pad: int = 0
    y_pad: int = 0
    z_pad: int = 0
    fft_shape: tuple | None = None  # [x y z]
    psf_grid: tuple | None = None   # [x y z]: the reference's grid when fft_shape was enlarged past it (decon(..., psf_grid=))
    p1: np.ndarray = field(default=None, repr=False)
    p2: np.ndarray = field(default=None, repr=False)


@dataclass
class Filter:
    """``filter`` struct (decwrap.py:294-305 defaults)."""
    gaussian_sigma: tuple = (0.5, 0.5, 2.5)
    gaussian_size: tuple = (13, 13, 25)
    dark: float = 0.0
    destripe_sigma: float = 0.0
    regularize_interval: int = 3
    use_fft: bool = False
    adaptive_psf: bool = False


def split_stack(stack_xyz, block: Block):
    """split_stack.m:9-26: boxes p1, p2 (n x 3, 1-based inclusive), x fastest, then y, then z."""
    sx, sy, sz = stack_xyz
    p1, p2 = [], []
    for iz in range(block.nz):
        zs = iz * block.z + 1
        for iy in range(block.ny):
            ys = iy * block.y + 1
            for ix in range(block.nx):
                xs = ix * block.x + 1
                p1.append((xs, ys, zs))
                p2.append((min(xs + block.x - 1, sx), min(ys + block.y - 1, sy), min(zs + block.z - 1, sz)))
    return np.array(p1, np.int64), np.array(p2, np.int64)


def decon_pad_size(psf_size_xyz):
    """LsDeconv.m:401-403: the full PSF extent per side."""
    return [int(math.ceil(k)) for k in psf_size_xyz]


def gaussian_pad_size(sigma_xyz, kernel_xyz):
    """LsDeconv.m:387-399: max(floor((2*ceil(3*sigma)+1)/2), kernel) per axis."""
    return [int(math.ceil(max((2 * math.ceil(3 * s) + 1) // 2, k))) for s, k in zip(sigma_xyz, kernel_xyz)]


def next_fast_len(n_vec):
    return [D.next_fast_len(int(n)) for n in n_vec]


def native_fft_shape(shape_xyz):
    """FFT shape >= ``shape_xyz`` that the hand-written FFT pipeline takes without the rocFFT fallback (2^a * {1,3,9}
    per axis, x even: ``mi_fft_good_size``).  Like ``next_fast_len`` it only enlarges the zero padding of deconFFT
    (decon.m:144), which the block's own pads absorb.  An axis with no such extent (z > 2304) keeps its 7-smooth one."""
    out = []
    for axis, n in enumerate(shape_xyz):
        g = int(capi.lib().mi_fft_good_size(int(n), axis))
        out.append(g if g > 0 else D.next_fast_len(int(n)))
    return out


def block_fft_shape(bl_xyz, main_fft_shape=None):
    """FFT shape of ONE padded block (decwrap's run_block): the grid the hand-written pipeline takes whenever that costs less than the
    7-smooth grid through rocFFT -- 1 against 3.5 per grid point (``cost_per_core_voxel``), and a rocFFT plan of a new shape takes
    0.7 s to create where the pipeline's set-up takes 10 ms (profiles/r05_decwrap_spans.txt) -- i.e. up to 3.4x the 7-smooth grid's
    volume when the grid stays inside the main block's (``main_fft_shape``: the one ``autosplit`` sized against --block-size-max, so
    memory is not in question; remainder blocks are smaller on every axis, so is their native grid), up to 1.3x otherwise.  It also keeps
    blocks of different shapes away from rocFFT plans that are alive at the same time (profiles/r05_rocfft_coexistence.txt).  The native
    extents are all even, the 7-smooth ones need not be: the caller hands the 7-smooth grid to ``decon`` as ``psf_grid`` so that the
    PSF lands where the reference's grid puts it (``mi_rl_options.psf_grid``)."""
    smooth, native = next_fast_len(bl_xyz), native_fft_shape(bl_xyz)
    ratio = float(np.prod(native)) / float(np.prod(smooth))
    if ratio <= 1.3:
        return native
    inside = main_fft_shape is not None and all(int(a) <= int(b) for a, b in zip(native, main_fft_shape))
    return native if (ratio <= 3.4 and inside) else smooth


def autosplit(stack_xyz, psf_size_xyz, filt: Filter, block_size_max: int, numit: int, ram_available: int | None = None,
              output_bytes: int = 2, cores_in_flight: int = 4) -> Block:
    """The block (core + 2*pad, an FFT-friendly shape for the FFT path) with fewer than ``block_size_max`` elements whose CORE is
    largest (the reference's score, LsDeconv.m:365), square in xy.

    Same ingredients as LsDeconv.m:308-385 -- pad = max(PSF extent, Gaussian pad) per side, FFT shapes rounded to 7-smooth, square
    xy blocks, score = core volume, and the host-memory terms: post-processing holds one z slab of bricks at the output type
    (``output_bytes`` x stack_x x stack_y x block_z, at most half of ``ram_available``: :317-322) while ``cores_in_flight`` float32
    cores wait for their writers (:359) -- without MATLAB's 2^31-element / 1290-per-side gpuArray limits.
    ``ram_available`` None: no host-memory term (tests)."""
    pad = [1, 1, 1] if filt.destripe_sigma > 0 else [0, 0, 0]                                   # LsDeconv.m:339-340
    if numit > 0:
        pad = [max(a, b) for a, b in zip(pad, decon_pad_size(psf_size_xyz))]
    if any(s > 0 for s in filt.gaussian_sigma):
        pad = [max(a, b) for a, b in zip(pad, gaussian_pad_size(filt.gaussian_sigma, filt.gaussian_size))]
    sx, sy, sz = stack_xyz
    z_max = sz
    if ram_available is not None:
        z_max = max(1, min(sz, int(0.5 * ram_available // (output_bytes * sx * sy))))

    def shape_of(core):
        shape = [c + 2 * p for c, p in zip(core, pad)]
        if filt.use_fft:
            smooth, native = next_fast_len(shape), native_fft_shape(shape)
            # prefer the native-pipeline shape unless it inflates the block by more than 30 % over the 7-smooth one
            shape = native if np.prod(native) <= 1.3 * np.prod(smooth) else smooth
        return shape

    def cost_per_core_voxel(core, shape):
        """deconFFT blocks: transform cost of the block per voxel it contributes -- grid points x 1 (hand-written pipeline) or x 3.5
        (rocFFT) over core voxels.  The reference's score is the core volume alone (LsDeconv.m:365); with two
        transform routes of different speed the same intent -- least work per output voxel -- needs the route in it."""
        if not filt.use_fft:
            return 1.0
        native = list(shape) == native_fft_shape(shape)
        return float(np.prod(shape)) * (1.0 if native else 3.5) / float(np.prod(core))

    def fits(xy, z):
        core = [min(xy, sx), min(xy, sy), z]
        shape = shape_of(core)
        if shape[0] * shape[1] * shape[2] >= block_size_max:
            return None
        if ram_available is not None and output_bytes * sx * sy * z + core[0] * core[1] * z * cores_in_flight * 4 > ram_available:
            return None
        return core, shape

    # candidate depths: every depth near the ends, a geometric ladder in between (the score is smooth in z)
    zs = sorted({z_max} | {max(1, int(round(z_max * 0.85 ** i))) for i in range(60)} | set(range(1, min(z_max, 8) + 1)), reverse=True)
    best, best_score = None, -1
    for z in zs:
        lo, hi = 1, max(sx, sy)                  # largest square xy that fits at this depth (monotone: bisection)
        if fits(lo, z) is None:
            continue
        while lo < hi:
            mid = (lo + hi + 1) // 2
            if fits(mid, z) is not None:
                lo = mid
            else:
                hi = mid - 1
        if filt.use_fft:
            # the padded volume is not monotone in xy -- shape_of moves between the native and the 7-smooth grid by its 1.3x rule,
            # and both grids are staircases -- so the bisection can stop below a larger xy that fits.  The two grids of one core differ
            # by at most 1.3x in volume, i.e. 1.14x per xy side: nothing beyond that fits
            top = min(max(sx, sy), int(1.15 * (lo + 2 * max(pad))) + 8)
            for xy in range(top, lo, -1):
                if fits(xy, z) is not None:
                    lo = xy
                    break
        # the largest xy of this depth, and -- FFT path -- the largest xy below it whose grid the hand-written pipeline takes
        cands = [fits(lo, z)]
        if filt.use_fft:
            xy = lo
            for _ in range(64):
                c = fits(xy, z)
                if c is not None and list(c[1]) == native_fft_shape(c[1]):
                    # (the walk down takes steps of xy / 64: the largest side of this native grid lies within the last step)
                    for up in range(xy + 1, min(lo, xy + max(1, xy // 64) + 1) + 1):
                        c2 = fits(up, z)
                        if c2 is None or list(c2[1]) != native_fft_shape(c2[1]):
                            break
                        c = c2
                    cands.append(c)
                    break
                xy -= max(1, xy // 64)
                if xy < 1:
                    break
        for core, shape in cands:
            score = core[0] * core[1] * core[2] / cost_per_core_voxel(core, shape)
            if score > best_score:
                best, best_score = (core, shape), score
    if best is None:
        raise RuntimeError("autosplit: No block shape fits in memory. Try increasing block_size_max or reducing min_block.")
    core, shape = best
    # (the reference's grid of the full-size block travels with it: process_block places the PSF by ITS parity, block_fft_shape)
    blk = Block(core[0], core[1], core[2], math.ceil(sx / core[0]), math.ceil(sy / core[1]), math.ceil(sz / core[2]),
                pad[0], pad[1], pad[2], tuple(shape) if filt.use_fft else None,
                tuple(next_fast_len([c + 2 * p for c, p in zip(core, pad)])) if filt.use_fft else None)
    blk.p1, blk.p2 = split_stack(stack_xyz, blk)
    return blk


def estimate_block_size_max(device=None, n_real=2, n_complex=0) -> int:
    """decwrap.py:133-169 re-targeted: elements = usable HBM / 4 B / (#real + 2*#complex work volumes)."""
    capi.require_gpu()
    free, _total = torch.cuda.mem_get_info(device)
    usable = max(0, free - (3 << 30))
    return int(usable // 4 // max(1, n_real + 2 * n_complex))


_DTYPE_CODE = {np.dtype(np.uint8): 1, np.dtype(np.uint16): 2, np.dtype(np.float32): 4}
_COPY_POOL = []


def _copy_pool():
    """threads that copy boxes into pinned memory (shared by the block workers of the process)"""
    if not _COPY_POOL:
        from concurrent.futures import ThreadPoolExecutor
        _COPY_POOL.append(ThreadPoolExecutor(max_workers=8, thread_name_prefix="box-copy"))
    return _COPY_POOL[0]


def load_block_device(volume: np.ndarray, p1, p2, pad_xyz, device, staging=None):
    """``load_block`` (LsDeconv.m:817-904) with the conversion and the symmetric padding on the device: only the raw samples of
    the box read from the volume cross PCIe (2 B per voxel for uint16 instead of a float32 block built on the host).
    ``staging``: a dict the caller keeps per worker for the pinned host buffer.  Returns a float32 CUDA tensor (Z, Y, X),
    bit-identical to ``load_block``.  uint8 / uint16 / float32 volumes; others go through ``load_block``."""
    if volume.dtype not in _DTYPE_CODE:
        return torch.from_numpy(load_block(volume, p1, p2, pad_xyz)).to(device)
    vol_xyz = volume.shape[::-1]
    req_s = [int(a) - int(p) for a, p in zip(p1, pad_xyz)]
    req_e = [int(b) + int(p) for b, p in zip(p2, pad_xyz)]
    rd_s = [max(1, s) for s in req_s]
    rd_e = [min(n, e) for n, e in zip(vol_xyz, req_e)]
    sub = volume[rd_s[2] - 1:rd_e[2], rd_s[1] - 1:rd_e[1], rd_s[0] - 1:rd_e[0]]
    nbytes = sub.size * sub.dtype.itemsize
    staging = {} if staging is None else staging
    if staging.get("load_bytes", 0) < nbytes:
        staging["load"] = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        staging["load_bytes"] = nbytes
    if staging.get("load_event") is not None:
        staging["load_event"].synchronize()                                            # the previous block's upload has left the buffer
    host = staging["load"][:nbytes]
    dst_np = host.numpy().view(sub.dtype).reshape(sub.shape)
    # the box straight from the (memory-mapped) volume into pinned memory, in z slices on a few threads: one thread copies a
    # 0.5-GB box in 70 ms -- as long as the block's kernels take -- and numpy releases the GIL inside the copy
    nz_ = sub.shape[0]
    if nbytes >= (64 << 20) and nz_ >= 8:
        cuts = [nz_ * i // 4 for i in range(5)]
        list(_copy_pool().map(lambda ab: dst_np.__setitem__(slice(ab[0], ab[1]), sub[ab[0]:ab[1]]), zip(cuts[:-1], cuts[1:])))
    else:
        dst_np[...] = sub
    raw = torch.empty(nbytes, dtype=torch.uint8, device=device)
    raw.copy_(host, non_blocking=True)
    staging["load_event"] = torch.cuda.Event()
    staging["load_event"].record(torch.cuda.current_stream(device))
    shape = tuple(e - s + 1 for s, e in zip(req_s[::-1], req_e[::-1]))               # (Z, Y, X) of the padded block
    dst = torch.empty(shape, dtype=torch.float32, device=device)
    before = [a - b for a, b in zip(rd_s, req_s)]
    capi.check(capi.lib().mi_load_block(device.index, capi.current_stream_ptr(device), raw.data_ptr(), _DTYPE_CODE[volume.dtype],
                                        sub.shape[2], sub.shape[1], sub.shape[0], dst.data_ptr(), shape[2], shape[1], shape[0],
                                        before[0], before[1], before[2]))
    raw.record_stream(torch.cuda.current_stream(device))
    return dst


def load_block(volume: np.ndarray, p1, p2, pad_xyz) -> np.ndarray:
    """LsDeconv.m:817-904 on an in-memory (Z, Y, X) volume: read the padded box where it exists, fill the rest with
    ``padarray(..., 'symmetric')`` (edge-inclusive mirror), return float32 in [0,1] for integer inputs (im2single)."""
    vol_xyz = volume.shape[::-1]
    req_s = [int(a) - int(p) for a, p in zip(p1, pad_xyz)]
    req_e = [int(b) + int(p) for b, p in zip(p2, pad_xyz)]
    rd_s = [max(1, s) for s in req_s]
    rd_e = [min(n, e) for n, e in zip(vol_xyz, req_e)]
    sub = volume[rd_s[2] - 1:rd_e[2], rd_s[1] - 1:rd_e[1], rd_s[0] - 1:rd_e[0]]
    if np.issubdtype(sub.dtype, np.integer):
        sub = sub.astype(np.float32) / np.float32(np.iinfo(sub.dtype).max)
    else:
        sub = sub.astype(np.float32)
    before = [a - b for a, b in zip(rd_s, req_s)]
    after = [a - b for a, b in zip(req_e, rd_e)]
    if any(before) or any(after):
        sub = np.pad(sub, [(before[2], after[2]), (before[1], after[1]), (before[0], after[0])], mode="symmetric")
    return np.ascontiguousarray(sub)


def deconvolved_stats(bl: torch.Tensor, clipval: float):
    """LsDeconv.m:1300-1307: ``prctile(bl, [100-clipval, clipval], "all")`` -- exact, on the device (``mi_prctile``)."""
    lb, ub = D.prctile(bl, [100.0 - clipval, clipval])
    return lb, ub


def output_scale(rawmax: float, convert_to_8bit=False, convert_to_16bit=False) -> float:
    """Target data-type maximum of postprocess_save (LsDeconv.m:1009-1024)."""
    if convert_to_8bit:
        rawmax = 255
    elif convert_to_16bit:
        rawmax = 65535
    if convert_to_8bit or rawmax <= 255:
        return 255.0
    if convert_to_16bit or rawmax <= 65535:
        return 65535.0
    return float(rawmax)


def process_block(bl, block: Block, psf, niter, lambda_, stop_criterion, filt: Filter, clipval=99.99, gpu=1, plan=None):
    """``[bl, lb, ub] = process_block(bl, block, psf, niter, lambda, stop_criterion, filter, clipval, gpu, ...)``
    (LsDeconv.m:906-948) on device ``gpu`` (1-based like ``gpuDevice(gpu)``)."""
    dev = torch.device("cuda", int(gpu) - 1)
    t = torch.from_numpy(np.ascontiguousarray(bl, dtype=np.float32)).to(dev) if isinstance(bl, np.ndarray) else bl
    size0 = tuple(t.shape)
    if any(s > 0 for s in filt.gaussian_sigma):
        D.gauss3d_gpu(t, list(filt.gaussian_sigma), list(filt.gaussian_size))                  # LsDeconv.m:917-919
        if filt.dark > 0:
            capi.check(capi.lib().mi_subtract_dark(dev.index, capi.current_stream_ptr(dev), t.data_ptr(), t.data_ptr(),
                                                   t.numel(), float(filt.dark)))                # :924-927
    if niter > 0 and float(t.max()) > 2.0 ** -23:                                               # :929
        grid = block.psf_grid if (filt.use_fft and not filt.adaptive_psf) else None
        D.decon(t, psf, niter, lambda_, stop_criterion, filt.regularize_interval, gpu, filt.use_fft,
                block.fft_shape if filt.use_fft else None, filt.adaptive_psf, plan=plan, psf_grid=grid)
    if filt.destripe_sigma > 0:
        D.filter_subband_3d_z(t, filt.destripe_sigma, 0, "db9")                                 # :934-936
    lb, ub = deconvolved_stats(t, clipval)
    assert tuple(t.shape) == size0, "[process_block]: block size mismatch!"
    return t, lb, ub
