#!/usr/bin/env python3
"""Deconvolution launcher: the command line of the reference's LsDeconvolveMultiGPU/decwrap.py (:230-327) on top
of the MI355X library -- no MATLAB script is written, no MATLAB is spawned (decwrap.py:408-494); the blocks are
processed in this process through ``ipp_amd.lsdeconv.process_block``.

Input: a folder of 2-D slices (``*.tif`` / ``*.tiff`` like the reference, LsDeconv.m:585-588, or ``*.npy``) or one ``*.npy``
volume (Z, Y, X).  Output: ``<input>/deconvolved/deconvolved.npy`` (float32; + ``deconvolution_config.json`` like
decwrap.py:474-478, ``min_max.json``) and the rescaled integer stack ``deconvolved_{8,16}bit.npy`` that the reference's
postprocess_save writes as a TIFF series (LsDeconv.m:950-1100: percentile clip range of all blocks, amplification, round,
clamp; ``mi_rescale_block``); for a TIFF input also the ``img_%06d.tif`` series itself (LsDeconv.m:1120-1145).
Nothing of volume size lives in host memory: the input is read box by box (memory-mapped ``*.npy`` or lazily decoded TIFF
slices), every finished block goes to the cache folder (``--cache-drive``, default ``<out>/cache``) as ``bl_<n>.lz4`` in the
reference's brick format (save_lz4_mex.c), and the output is assembled one z slab of bricks at a time (postprocess_save).
A block is claimed by creating its brick file, so several workers -- or several machines sharing the cache folder, started
with different ``--start-block`` (decwrap.py:317-321; only ``--start-block 1`` assembles the output) -- split the work
(LsDeconv.m:696-706); ``block.json`` / ``min_max.json`` in the cache play the part of block.mat / min_max.mat, an interrupted
run resumes with the blocks that are missing (``--no-resume`` starts over), bricks are validated by header (type + core
shape) and the cache is removed after a complete run like LsDeconv.m:286-296.
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import sys
from concurrent.futures import CancelledError
from pathlib import Path

logging.basicConfig(level=logging.INFO, format="%(message)s")
log = logging.getLogger("decwrap")

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

SUPPORTED_PAIRS = [(350, 460), (405, 450), (430, 470), (458, 480), (488, 525), (514, 530), (532, 555), (561, 600),
                   (594, 620), (633, 660), (642, 690), (680, 710)]  # decwrap.py:181-194


def get_all_gpu_indices():
    """1-based device indices like the reference (decwrap.py:74-91 uses nvidia-smi; here HIP's device count)."""
    try:
        from ipp_amd import capi
        return list(range(1, max(0, capi.lib().mi_device_count()) + 1))
    except Exception:
        return []


def build_parser(default_gpus):
    p = argparse.ArgumentParser(description="Python wrapper for MI355X deconvolution.",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--version", action="version", version="DeconvWrapper v1.5 (MI355X)")
    p.add_argument("-i", "--input", type=Path, required=True, help="Path to the input image folder")
    p.add_argument("-dxy", "--dxy", type=float, required=True, help="Lateral resolution in micrometers")
    p.add_argument("-dz", "--dz", type=float, help="Axial resolution in micrometers")
    p.add_argument("-ex", "--lambda-ex", type=int, required=True, help="Excitation wavelength (488, 561, 642)")
    p.add_argument("-em", "--lambda-em", type=int, required=True, help="Emission wavelength (525, 600, 690)")
    p.add_argument("--use-fft", action="store_true", default=False, help="use FFT-based convolution (deconFFT semantics)")
    p.add_argument("--adaptive-psf", action="store_true", default=False, help="Wiener PSF update (deconFFT_Wiener)")
    p.add_argument("--cache-drive", type=str, default=None)
    p.add_argument("-it", "--numit", type=int, default=6, help="Number of deconvolution iterations [1-50]")
    p.add_argument("--na", type=float, default=0.40)
    p.add_argument("--rf", type=float, default=1.42)
    p.add_argument("--fcyl", type=int, default=240)
    p.add_argument("--slitwidth", type=float, default=12.0)
    p.add_argument("--lambda-damping", type=float, default=0.0)
    p.add_argument("--clipval", type=float, default=99.99)
    p.add_argument("--stop-criterion", type=float, default=0)
    p.add_argument("--block-size-max", type=int, default=0, help="Max elements per GPU block (0: from free HBM)")
    p.add_argument("--gpu-indices", type=int, nargs="+", default=default_gpus, help="1-based GPU indices")
    p.add_argument("--gpu-workers-per-gpu", type=int, default=5,
                   help="workers (own stream + pinned upload buffer) per GPU: they overlap one block's box read / PCIe / host staging "
                        "with the others' kernels (measured on a 17-GB volume, round 4: 3 -> 11.1 s, 4 -> 9.5 s, 5 -> 8.9 s, 6 -> 10.0 s)")
    p.add_argument("--cpu-workers", type=int, default=0)
    p.add_argument("--signal-amp", type=float, default=1.0)
    p.add_argument("--gaussian-sigma", type=float, nargs=3, default=[0.5, 0.5, 2.5])
    p.add_argument("--gaussian-filter-size", type=int, nargs=3, default=[13, 13, 25])
    p.add_argument("--denoise-strength", type=int, default=1)
    p.add_argument("--destripe-sigma", type=float, default=0.0)
    p.add_argument("--regularize-interval", type=int, default=3)
    p.add_argument("--no-resume", dest="resume", action="store_false")
    p.set_defaults(resume=True)
    p.add_argument("--flip", action="store_true")
    p.add_argument("--convert-to-8bit", action="store_true")
    p.add_argument("--convert-to-16bit", action="store_true")
    p.add_argument("--start-block", type=int, default=1)
    p.add_argument("--dry-run", action="store_true", help="Print the plan and exit without executing it")
    p.add_argument("--use-jemalloc", action="store_true", default=False)
    p.add_argument("--use-tcmalloc", action="store_true", default=False)
    return p


def validate_args(args):
    """decwrap.py:176-217."""
    if not Path(args.input).exists():
        raise ValueError(f"Path does not exist: {args.input}")
    args.input = Path(args.input).resolve()
    if (args.lambda_ex, args.lambda_em) not in SUPPORTED_PAIRS:
        pairs = ", ".join(f"{e}/{m}" for e, m in SUPPORTED_PAIRS)
        raise RuntimeError(f"Unsupported excitation/emission pair: {args.lambda_ex}/{args.lambda_em}. Valid pairs are: {pairs}")
    if len(args.gaussian_sigma) != 3:
        raise ValueError("Gaussian sigma must be a triplet, e.g., --gaussian-sigma 0.5 0.5 1.5")
    if len(args.gaussian_filter_size) != 3:
        raise ValueError("Gaussian filter size must be a triplet, e.g., --gaussian-filter-size 5 5 15")
    if args.adaptive_psf and not args.use_fft:
        raise RuntimeError("--adaptive-psf and --use-fft should be used simultaneously.")
    if args.cpu_workers:
        raise RuntimeError("--cpu-workers: this build has no CPU deconvolution path")


class LazyTiffVolume:
    """A folder of 2-D TIFF slices as a (Z, Y, X) array that is never whole in memory: ``vol[z0:z1, y0:y1, x0:x1]`` reads the
    slices it needs (the reference's load_block reads its box from the files too, LsDeconv.m:817-904, load_bl_tif.cpp) and keeps
    the most recently used ones -- the blocks of one z slab share their slices -- within ``cache_bytes``."""

    def __init__(self, folder: Path, cache_bytes: int):
        import threading
        from collections import OrderedDict
        import numpy as np
        from ipp_amd import brickio
        self.files = brickio.list_tiff_series(folder)
        if not self.files:
            raise RuntimeError(f"no *.tif slices in {folder}")
        first = brickio.load_tiff_series(folder, 0, 1)
        self.dtype, self.ndim = first.dtype, 3
        self.shape = (len(self.files),) + first.shape[1:]
        self._folder, self._np, self._brickio = folder, np, brickio
        self._cache, self._lock = OrderedDict(), threading.Lock()
        self._budget = max(1, int(cache_bytes) // max(1, first[0].nbytes))
        # slices the library's reader decodes (strips of raw / deflate samples: include/mi_tiffio.h) arrive sixteen at a time on all
        # cores; one bulk read at a time -- the workers of a z slab ask for the same slices at the same moment
        info = brickio.tiff_info(self.files[0])
        self._fast = bool(info is not None and info[2] and info[0] == first.shape[1:] and info[1] == first.dtype)
        self._bulk = threading.Lock()
        # ... and ahead of the workers: a thread walks through the folder in z order for as long as the cache has room (it never pushes a
        # slice out), so that inflating the input -- 4 GB/s on 16 CPUs, as long as the deconvolution itself -- runs beside the blocks
        # instead of in front of every z slab of them (17-GB stack: blocks phase 5.3 -> 5.1 s: inflating IS the blocks phase of such a
        # run, 3.4 GB/s); MI_DECWRAP_TIFF_PREFETCH=0: on demand only
        if self._fast and os.environ.get("MI_DECWRAP_TIFF_PREFETCH", "1") != "0":
            threading.Thread(target=self._prefetch, daemon=True).start()

    def close(self):
        """ends the read-ahead and drops the cached slices"""
        self._closed = True
        with self._lock:
            self._cache.clear()

    def _prefetch(self):
        z = 0
        while z < len(self.files) and self._fast and not getattr(self, "_closed", False):
            with self._lock:
                room = len(self._cache) + 16 <= self._budget
            if not room:
                return
            part = list(range(z, min(z + 16, len(self.files))))
            if self._chunk(part) is None:
                return
            z += len(part)

    def _slice(self, z):
        with self._lock:
            a = self._cache.get(z)
            if a is not None:
                self._cache.move_to_end(z)
                return a
        a = self._brickio.load_tiff_series(self._folder, z, z + 1)[0]
        if a.shape != self.shape[1:] or a.dtype != self.dtype:
            raise ValueError(f"{self.files[z]}: slice shape / type differs from the first slice")
        with self._lock:
            self._cache[z] = a
            while len(self._cache) > self._budget:
                self._cache.popitem(last=False)
        return a

    def _chunk(self, part):
        """up to sixteen slices as arrays: from the cache, the others decoded together by the library's reader (and cached); None when
        the reader does not take one of the files (slice by slice through Pillow from then on)"""
        with self._bulk:
            with self._lock:
                got = {z: self._cache.get(z) for z in part}
            missing = [z for z in part if got[z] is None]
            if missing:
                try:
                    arr = self._brickio.read_tiff_box([self.files[z] for z in missing], self.shape[1:], self.dtype, 0, self.shape[1], 0, self.shape[2])
                except Exception:
                    self._fast = False
                    return None
                with self._lock:
                    for j, z in enumerate(missing):
                        got[z] = self._cache[z] = arr[j]
                    while len(self._cache) > self._budget:
                        self._cache.popitem(last=False)
            else:
                with self._lock:
                    for z in part:
                        self._cache.move_to_end(z)
        return [got[z] for z in part]

    def __getitem__(self, key):
        np = self._np
        kz, ky, kx = key
        zs = list(range(*kz.indices(self.shape[0])))
        if self._fast and len(zs) > 1:
            out, i = None, 0
            while i < len(zs) and self._fast:
                got = self._chunk(zs[i:i + 16])
                if got is None:
                    break
                for j, a in enumerate(got):
                    b = a[ky, kx]
                    if out is None:
                        out = np.empty((len(zs),) + b.shape, self.dtype)
                    out[i + j] = b
                i += len(got)
            if out is not None and i == len(zs):
                return out
        out = None
        for i, z in enumerate(zs):
            a = self._slice(z)[ky, kx]
            if out is None:
                out = np.empty((len(zs),) + a.shape, self.dtype)
            out[i] = a
        return out if out is not None else np.empty((0,) + self._slice(0)[ky, kx].shape, self.dtype)


def open_volume(path: Path):
    """The input as something that answers ``.shape``, ``.dtype`` and box reads: a memory-mapped ``*.npy`` volume, a stack of
    ``*.npy`` slices, or a lazily read TIFF series (nothing of reference scale -- hundreds of Gvoxels, README.md:55-63 -- can be
    loaded whole)."""
    import numpy as np
    if path.is_file():
        return np.load(path, mmap_mode="r")
    files = sorted(path.glob("*.npy"))
    if files:
        return np.stack([np.load(f) for f in files])
    from ipp_amd import brickio
    if brickio.list_tiff_series(path):
        try:
            import psutil
            budget = int(psutil.virtual_memory().available * 0.25)
        except Exception:
            budget = 8 << 30
        return LazyTiffVolume(path, int(os.environ.get("MI_DECWRAP_SLICE_CACHE_BYTES", budget)))
    raise RuntimeError(f"no *.npy / *.tif slices in {path}")


def host_cores():
    """Host cores this process may keep busy: the affinity mask capped by the cgroup CPU quota (a container of 16 CPUs on a 256-core
    host shows 256 in its mask; threads beyond the quota only make the kernel stop EVERY thread of the group -- the GPU workers too --
    until the next period)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 4
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
        try:
            with open(path) as f:
                quota, period = parse(f.read())
            if quota not in ("max", "-1") and int(period) > 0:
                n = max(1, min(n, -(-int(quota) // int(period))))
            break
        except (OSError, ValueError):
            continue
    return n


def evict_cores(g, resident, resident_dev, brick_future, lock, evict_lock, brick_complete, write_brick):
    """The cores of device ``g`` leave ``resident`` (block -> core); a core whose brick is not complete is written first
    (``write_brick(n, core)``).  Safe with several workers per device that run out of memory together: evictions of one device take
    turns (``evict_lock``) and the second finds nothing left; a core is popped under ``lock`` (a block is evicted once); a block whose
    brick is still being written by the writer pool (``brick_future[n]``) is waited for, never written a second time -- two writers
    of one ``bl_n.lz4.tmp`` would interleave.  Returns the number of cores evicted."""
    done = 0
    with evict_lock:
        with lock:
            mine = [n for n, gg in resident_dev.items() if gg == g and n in resident]
        for n in mine:
            with lock:
                core = resident.pop(n, None)
                fut = brick_future.get(n)
            if core is None:
                continue
            if fut is not None:
                try:
                    fut.result()
                except CancelledError:     # (a brick that was to be written behind the workers and never begun)
                    pass
            if not brick_complete(n):
                write_brick(n, core)
            del core
            done += 1
    return done


def main(argv=None):
    import time as _time
    t_main0 = _time.perf_counter()
    gpus = get_all_gpu_indices()
    args = build_parser(gpus).parse_args(argv)
    validate_args(args)
    out_dir = (args.input.parent if args.input.is_file() else args.input) / (
        "deconvolved_flipped_upside_down" if args.flip else "deconvolved")                  # LsDeconv.m:91-94
    cfg = {k: (str(v) if isinstance(v, Path) else v) for k, v in vars(args).items()}
    if args.dry_run:
        print(json.dumps({"would_write": str(out_dir), "config": cfg}, indent=2))
        return 0
    if not args.gpu_indices:
        raise RuntimeError("no GPU available: this build has no CPU deconvolution path")

    import numpy as np
    import torch
    from ipp_amd import capi, decon as D, lsdeconv as L, psf as P
    capi.require_gpu()
    # torch's lazy device initialisation has to happen on this thread: the workers below only create streams and tensors
    torch.cuda.init()
    for g in sorted(set(args.gpu_indices)):
        if not 1 <= g <= torch.cuda.device_count():
            raise RuntimeError(f"--gpu-indices {g}: this host has {torch.cuda.device_count()} GPU(s) (indices are 1-based)")

    out_dir.mkdir(exist_ok=True)
    with open(out_dir / "deconvolution_config.json", "w") as f:
        json.dump(cfg, f, indent=2)
    vol = open_volume(args.input)
    sz, sy, sx = vol.shape
    dz = args.dz if args.dz else args.dxy
    # the library's code objects load at the first launch of each kernel family (tenths of a second each): touch the ones a block
    # needs on a thread while the host computes the PSF
    def _warm():
        try:
            with torch.cuda.device(args.gpu_indices[0] - 1):
                t = torch.rand((8, 16, 64), device="cuda")
                D.gauss3d_gpu(t, 0.5)
                D.norm2(t)
                D.prctile(t, [0.5, 99.5])
                with D.DeconPlan(args.gpu_indices[0]) as pl_:
                    D.decon(t, np.ones((3, 3, 3), np.float32) / 27.0, 2, 0.0, 0.0, 0, use_fft=args.use_fft,
                            fft_shape=[64, 16, 8] if args.use_fft else None, plan=pl_)
                torch.cuda.synchronize()
        except Exception as e:                                                             # (a warm-up must never end the run)
            log.debug(f"warm-up skipped: {e!r}")

    import threading as _threading
    # (the free memory that sizes the block grid is read BEFORE the warm-up allocates anything on this device: the grid, and with it
    # the shapes a resumed cache folder must match, does not depend on how far the thread has got)
    free0 = torch.cuda.mem_get_info(args.gpu_indices[0] - 1)[0]
    _warm_thread = _threading.Thread(target=_warm, daemon=True)
    _warm_thread.start()
    t_psf0 = _time.perf_counter()
    psf = P.LsMakePSF(args.dxy * 1000.0, dz * 1000.0, args.na, args.rf, float(args.lambda_ex), float(args.lambda_em),
                      float(args.fcyl), args.slitwidth)                                    # LsDeconv.m:160 (nm units)
    log.info(f"imports + device {t_psf0 - t_main0:.1f} s, LsMakePSF {_time.perf_counter() - t_psf0:.1f} s")
    psf_struct = D.make_psf_struct(psf)
    log.info(f"PSF size (x y z): {psf.shape[::-1]}")
    filt = L.Filter(tuple(args.gaussian_sigma), tuple(args.gaussian_filter_size), 0.0, args.destripe_sigma,
                    args.regularize_interval, args.use_fft, args.adaptive_psf)
    gpu = args.gpu_indices[0]
    # workers that share a device share its memory (decwrap.py:133-169 divides by the workers per GPU as well)
    per_dev = max(1, args.gpu_workers_per_gpu) * max(1, args.gpu_indices.count(gpu))
    # Finished cores stay in device memory until the assembly needs them (288 GB of HBM hold a 30-Gvoxel result in float32): the
    # assembly then rescales them where they are instead of reading its bricks back, decompressing them and sending them up again.
    # MI_DECWRAP_RESIDENT=0 switches that off.  The bricks of RESIDENT cores are only there for a resumed run, and writing 64 GB of
    # float32 into the cache folder took as long as computing it (workers done after 5.7 s with, 2.8 s without: r05_decwrap_scale.txt):
    # MI_DECWRAP_BRICKS=trail (default) writes them BEHIND the workers -- a core stays on its device, a writer thread
    # (MI_DECWRAP_TRAIL_THREADS, default one) fetches it when it has a free buffer, and whatever is not begun when the last block is
    # done is not written at all: the output is assembled from the resident cores and the cache folder goes away with a complete run;
    # an interrupted run resumes from the bricks it has and recomputes the rest.  This only concerns results that fit into device
    # memory, i.e. runs of seconds to a few minutes; cores that find no room on the device get their brick before their worker goes on,
    # as do all blocks of a helper process (--start-block > 1).  MI_DECWRAP_BRICKS=1: every brick is complete before its worker goes on (the reference's
    # order, LsDeconv.m:799-806); =0: no brick for a core that stays resident (no D2H of the float32 core, no LZ4, no file: the run
    # cannot be resumed).  Memory: with an explicit --block-size-max whatever the workers' blocks leave free; otherwise the result's
    # share of this device is set aside first, if that is at most half of it.
    n_work_vols = 3 + 2 * (2 if args.use_fft else 0)
    keep_resident = os.environ.get("MI_DECWRAP_RESIDENT", "1") != "0" and int(args.start_block) == 1
    bricks_mode = os.environ.get("MI_DECWRAP_BRICKS", "trail")
    if bricks_mode not in ("0", "1", "trail"):
        raise ValueError("MI_DECWRAP_BRICKS must be 0, 1 or trail")
    keep_bricks = bricks_mode != "0" or not keep_resident
    trail_bricks = bricks_mode == "trail" and keep_resident
    res_share = sz * sy * sx * 4 // len(set(args.gpu_indices)) + (1 << 30)
    if args.block_size_max:
        bmax = args.block_size_max
        # (at most half of the free memory: the workers' real allocations -- transform grids 1.3 x the block, taper engines, plan
        # buffers -- are larger than the estimate below, and resident cores must never be what makes them fail)
        res_budget = min(free0 // 2, max(0, free0 - (3 << 30) - per_dev * bmax * 4 * n_work_vols)) if keep_resident else 0
    else:
        res_budget = res_share if (keep_resident and res_share <= free0 // 2) else 0
        bmax = max(1, (free0 - (3 << 30) - res_budget) // 4 // n_work_vols) // per_dev
        # ... which is what fits, not what is fast: the workers form a pipeline (box read, upload, kernels, download) that fills and
        # drains once per run, and the last blocks leave workers idle.  At least a dozen blocks per worker, blocks of 3e8 elements or
        # more (smaller ones pay the pads twice over): the 17-GB probe takes 5.7 s with 3e8, 5.9 s with 1.5e8, 6.8 s with 6e8 and
        # 17.6 s with the 1.2e9 the memory rule alone would allow (profiles/r05_decwrap_scale.txt)
        n_workers_all = max(1, len(args.gpu_indices) * max(1, args.gpu_workers_per_gpu))
        bmax = min(bmax, max(300_000_000, int(1.25 * sz * sy * sx) // (12 * n_workers_all)))

    import shutil
    import threading
    import time
    from concurrent.futures import ThreadPoolExecutor
    from ipp_amd import brickio
    # LZ4 runs as chunk jobs on one pool for the whole run (bricks are written -- and later read -- as 32-MiB chunks, which the
    # brick format allows: every chunk's sizes are in the header, save_lz4_mex.c:56-67): a single core compresses side by side
    # on every host core instead of on one, and the few writer threads only put the chunks into their file in order
    n_cores = host_cores()
    n_writers = max(2, min(4, n_cores // 4))
    # (the GPU workers, which launch the kernels, and the brick writers need cores of their own: with every core compressing, the
    # device sat idle half of the time)
    n_busy = len(args.gpu_indices) * max(1, args.gpu_workers_per_gpu) + n_writers // 2
    codec = ThreadPoolExecutor(max_workers=int(os.environ.get("MI_DECWRAP_CODEC_THREADS", max(2, n_cores - n_busy))))
    brick_chunk = int(os.environ.get("MI_DECWRAP_BRICK_CHUNK", 32 << 20))
    cache = Path(args.cache_drive) if args.cache_drive else out_dir / "cache"
    if not args.resume and args.start_block == 1:
        if cache.exists():
            shutil.rmtree(cache)                                                           # LsDeconv.m:136-139
        for f in out_dir.glob("img_*.tif"):
            f.unlink()                                                                     # LsDeconv.m:124
    cache.mkdir(parents=True, exist_ok=True)

    # ---- block geometry: kept in the cache folder (block.mat of LsDeconv.m:176-193) and checked against this run on resume
    stack_info = {"x": int(sx), "y": int(sy), "z": int(sz), "dtype": str(np.dtype(vol.dtype))}
    block_path = cache / "block.json"
    block = None
    if args.resume and block_path.exists():
        with open(block_path) as f:
            saved = json.load(f)
        for k, v in stack_info.items():
            if saved["stack_info"].get(k) != v:
                raise RuntimeError(f"Loaded block.json stack_info.{k} ({saved['stack_info'].get(k)}) does not match current stack_info ({v})")
        b = saved["block"]
        block = L.Block(b["x"], b["y"], b["z"], b["nx"], b["ny"], b["nz"], b["x_pad"], b["y_pad"], b["z_pad"],
                        tuple(b["fft_shape"]) if b["fft_shape"] else None)
        block.p1, block.p2 = L.split_stack((sx, sy, sz), block)
        if len(block.p1) != block.nx * block.ny * block.nz:
            raise RuntimeError("block.p1 shape mismatch with block.nx, block.ny, block.nz")
        log.info("Resuming by loading block info ...")
    if block is None:
        try:
            import psutil
            ram = int(os.environ.get("MI_DECWRAP_RAM_BYTES", psutil.virtual_memory().available))
        except Exception:
            ram = int(os.environ.get("MI_DECWRAP_RAM_BYTES", 64 << 30))
        out_bytes = 1 if (args.convert_to_8bit or (np.dtype(vol.dtype).itemsize == 1 and not args.convert_to_16bit)) else 2
        block = L.autosplit((sx, sy, sz), psf.shape[::-1], filt, bmax, args.numit, ram_available=ram, output_bytes=out_bytes,
                            cores_in_flight=n_writers + len(args.gpu_indices) * max(1, args.gpu_workers_per_gpu))
        tmp = block_path.with_suffix(".json.tmp")
        with open(tmp, "w") as f:
            json.dump({"stack_info": stack_info,
                       "block": {k: (list(getattr(block, k)) if k == "fft_shape" and block.fft_shape else getattr(block, k))
                                 for k in ("x", "y", "z", "nx", "ny", "nz", "x_pad", "y_pad", "z_pad", "fft_shape")}}, f)
        os.replace(tmp, block_path)
    num_blocks = len(block.p1)
    log.info(f"block grid {block.nx} x {block.ny} x {block.nz}, core ({block.x} {block.y} {block.z}), "
             f"pad ({block.x_pad} {block.y_pad} {block.z_pad}), fft_shape {block.fft_shape}")
    pad = (block.x_pad, block.y_pad, block.z_pad)

    def core_shape(n):
        p1, p2 = block.p1[n - 1], block.p2[n - 1]
        return (int(p2[2] - p1[2] + 1), int(p2[1] - p1[1] + 1), int(p2[0] - p1[0] + 1))

    def brick_path(n):
        return cache / f"bl_{n}.lz4"

    resident = {}                                   # block number -> its float32 core on the device that computed it
    resident_dev = {}                               # block number -> that device
    res_used = {g: 0 for g in set(args.gpu_indices)}

    def brick_complete(n):
        """a finished brick of this block: non-empty, readable header, float32, the block's core shape (or its core is resident)"""
        if n in resident:
            return True
        p = brick_path(n)
        try:
            if p.stat().st_size == 0:
                return False
            with open(p, "rb") as f:
                h = brickio.read_header(f)
            dims = tuple(int(v) for v in h["dims"][:int(h["ndims"])])[::-1]
            return int(h["dtype"]) == brickio.DT_SINGLE and dims == core_shape(n)
        except (OSError, ValueError):
            return False

    # ---- running statistics shared through the cache folder (min_max.mat of LsDeconv.m:760-790): clip range over the blocks
    # finished so far and the largest raw sample (only needed for inputs that are not 8 / 16 bit)
    lock = threading.Lock()
    mm_path = cache / "min_max.json"
    int_input = np.issubdtype(np.dtype(vol.dtype), np.integer)
    state = {"deconvmin": float("inf"), "deconvmax": 0.0, "rawmax": float(np.iinfo(vol.dtype).max) if int_input else float("-inf")}

    mm_written = [0.0]

    def merge_min_max(lb=None, ub=None, rawmax=None):
        with lock:
            if lb is not None:
                state["deconvmin"], state["deconvmax"] = min(state["deconvmin"], lb), max(state["deconvmax"], ub)
            if rawmax is not None:
                state["rawmax"] = max(state["rawmax"], rawmax)
            # the file is what other processes on a shared cache folder see: refreshed at most once a second while blocks finish
            # (half a dozen file operations per block were a fifth of a block's time on an overlay file system), always at the end
            now = time.monotonic()
            if lb is not None and now - mm_written[0] < 1.0:
                return
            mm_written[0] = now
            if mm_path.exists():                                                           # another process may have updated it
                try:
                    with open(mm_path) as f:
                        disk = json.load(f)
                    state["deconvmin"] = min(state["deconvmin"], disk["deconvmin"])
                    state["deconvmax"] = max(state["deconvmax"], disk["deconvmax"])
                    state["rawmax"] = max(state["rawmax"], disk["rawmax"])
                except (OSError, ValueError, KeyError):
                    pass
            if lb is not None:
                state["deconvmin"], state["deconvmax"] = min(state["deconvmin"], lb), max(state["deconvmax"], ub)
            if rawmax is not None:
                state["rawmax"] = max(state["rawmax"], rawmax)
            tmp = mm_path.with_suffix(".json.tmp")
            with open(tmp, "w") as f:
                json.dump(state, f)
            os.replace(tmp, mm_path)

    # One worker per entry of the device list (--gpu-indices x --gpu-workers-per-gpu, like the reference's pool of parfeval
    # workers, LsDeconv.m:620-668).  Blocks are independent; a block is CLAIMED by creating its (empty) brick file, so several
    # workers -- and several machines started with different --start-block on one shared cache folder (decwrap.py:317-321) --
    # never take the same block (LsDeconv.m:696-706); the result reaches the cache as bl_<n>.lz4.tmp + rename (:805-806).
    workers = [g for g in args.gpu_indices for _ in range(max(1, args.gpu_workers_per_gpu))]

    import socket
    me = f"{socket.gethostname()}:{os.getpid()}"

    claimed_here = set()

    def claim(n):
        """creates the (empty) brick file; who holds the claim is noted beside it, for the reaper of another round or process.
        (Without bricks nobody else can share this run: the workers of this process settle the blocks among themselves.)"""
        if not keep_bricks:
            with lock:
                if n in claimed_here:
                    return False
                claimed_here.add(n)
            return True
        try:
            os.close(os.open(brick_path(n), os.O_CREAT | os.O_EXCL | os.O_WRONLY))
        except FileExistsError:
            return False
        try:
            with open(brick_path(n).with_suffix(".claim"), "w") as f:
                f.write(me)
        except OSError:
            pass
        return True

    stale_s = float(os.environ.get("MI_DECWRAP_STALE_S", 1800))

    def claim_is_dead(n):
        """an incomplete brick may be removed when nobody can be working on it: it is this process' own leftover, its owner is a
        process of this host that no longer exists, or it has not been touched for MI_DECWRAP_STALE_S seconds (default 30 min: far
        beyond a block's run time) -- a live claim of another machine on a shared cache folder is left alone"""
        try:
            owner = brick_path(n).with_suffix(".claim").read_text().strip()
        except OSError:
            owner = ""
        host, _, pid = owner.rpartition(":")
        if owner == me:
            return True
        if host == socket.gethostname() and pid.isdigit():
            try:
                os.kill(int(pid), 0)
            except ProcessLookupError:
                return True
            except PermissionError:
                pass
        newest = 0.0
        for q in (brick_path(n), brick_path(n).with_suffix(".lz4.tmp"), brick_path(n).with_suffix(".claim")):
            try:
                newest = max(newest, q.stat().st_mtime)
            except OSError:
                pass
        # (no owner note: the claimant died between its two steps, or the file was placed by hand -- a minute is enough)
        return time.time() - newest > (60.0 if owner == "" else stale_s)

    # Pinned host buffers of one core each, shared by the workers (D2H target), the brick writers (LZ4 source) and, later, the brick
    # readers of the assembly (LZ4 target, H2D source): a core goes device -> pinned buffer -> compressed chunks -> file without
    # another copy on the host.  The pool is the back-pressure too: a worker waits for a buffer when the writers are behind.
    import queue
    core_max = int(block.x) * int(block.y) * int(block.z)
    n_stage = len(workers) + n_writers
    stage_free, stage_made = queue.Queue(), [0]

    def stage_get():
        with lock:
            make = stage_made[0] < n_stage
            if make:
                stage_made[0] += 1
        if make and stage_free.empty():
            return torch.empty(core_max, dtype=torch.float32, pin_memory=True)
        if make:
            with lock:
                stage_made[0] -= 1
        return stage_free.get()

    def stage_warm():
        """pins the pool's buffers while the first blocks are being read and deconvolved (pinning ~1 GB takes a few tenths of a second)"""
        try:
            while True:
                with lock:
                    if stage_made[0] >= n_stage:
                        return
                    stage_made[0] += 1
                stage_free.put(torch.empty(core_max, dtype=torch.float32, pin_memory=True))
        except Exception:                                                                  # (the lazy path reports what is wrong)
            with lock:
                stage_made[0] -= 1

    threading.Thread(target=stage_warm, daemon=True).start()

    # The assembly's host buffers -- one integer z slab of the whole stack (fresh pages: a second per 4 GB) and two pinned integer
    # cores -- are made while the blocks are being deconvolved, when the output type is already known (integer input: the scale
    # follows from the input type, LsDeconv.m:1009-1024); the device used to sit idle for 1.5 s between the two phases.
    prep = {}

    def prepare_assembly():
        try:
            scal0 = L.output_scale(float(np.iinfo(vol.dtype).max), args.convert_to_8bit, args.convert_to_16bit)
            dt = np.uint8 if scal0 <= 255 else np.uint16
            q = [torch.empty(core_max, dtype=torch.uint8 if dt == np.uint8 else torch.uint16, pin_memory=True) for _ in range(2)]
            slab0 = np.empty((int(block.z), sy, sx), dt)
            slab0.fill(0)                                                                  # (touches every page)
            prep.update(dtype=dt, q_host=q, slab=slab0)
        except Exception as e:                                                             # (the assembly then makes its own)
            log.debug(f"assembly buffers not prepared ahead: {e}")

    prep_thread = None
    # (not for a TIFF folder whose slices are going to be deflated on the device: that slab lives there)
    tiff_in = not args.input.is_file() and bool(brickio.list_tiff_series(args.input))
    slab_on_device = (tiff_in and os.environ.get("MI_DECWRAP_TIFF_DEVICE", "1") != "0" and
                      os.environ.get("MI_DECWRAP_NPY", "1" if sz * sy * sx <= (1 << 27) else "0") != "1")
    if int_input and int(args.start_block) == 1 and not slab_on_device:
        prep_thread = threading.Thread(target=prepare_assembly, daemon=True)
        prep_thread.start()

    evict_locks = {g: threading.Lock() for g in set(workers)}   # one eviction at a time per device (several workers share one)
    brick_future = {}                                            # block -> the writer's future of its brick (save_brick / trail_brick)
    stop_trailing = threading.Event()                            # the last block is done: no further brick of a resident core is begun
    trailing = []                                                # futures of trail_brick
    trail_local = threading.local()

    def trail_brick(n, lb, ub):
        """MI_DECWRAP_BRICKS=trail: the brick of a core that stays on its device, fetched by a writer thread when it has a pinned buffer
        to spare.  True: written.  (evict_cores waits for this future before it decides whether the core still needs a brick.)"""
        if stop_trailing.is_set():
            return False
        with lock:
            core = resident.get(n)
        if core is None:                                                                   # (evicted meanwhile: evict_cores wrote it)
            return False
        host = stage_get()
        try:
            if stop_trailing.is_set():
                stage_free.put(host)
                return False
            with torch.cuda.device(core.device):
                st = getattr(trail_local, "stream", None)
                if st is None or st.device != core.device:
                    st = trail_local.stream = torch.cuda.Stream(device=core.device)
                view = host[:core.numel()].view(core.shape)
                with torch.cuda.stream(st):
                    view.copy_(core, non_blocking=True)
                st.synchronize()
        except BaseException:
            stage_free.put(host)
            raise
        del core
        save_brick(n, view.numpy(), host, lb, ub)                                          # (hands the buffer back)
        with lock:
            timing["bricks_trailed"] += 1
        return True

    def write_evicted(n, core):
        arr = core.cpu().numpy()
        brickio.save_lz4(brick_path(n).with_suffix(".lz4.tmp"), arr, chunk_size=brick_chunk, pool=codec)
        try:
            os.replace(brick_path(n).with_suffix(".lz4.tmp"), brick_path(n))
        except FileNotFoundError:       # (the claim was reaped by another process meanwhile: save_brick has the same race)
            log.warning(f"block {n}: the claim was removed by another process while its brick was being written; skipped")

    def evict_resident(g):
        """Allocation failure on GPU g: its resident cores are given up (a core without a brick is written first), the budget of
        every device drops to zero, cached blocks go back to the driver."""
        nonlocal res_budget
        with lock:
            res_budget = 0
        k = evict_cores(g, resident, resident_dev, brick_future, lock, evict_locks[g], brick_complete, write_evicted)
        log.warning(f"GPU {g}: out of device memory; {k} resident core(s) went back to their bricks")
        with lock:
            res_used[g] = 0
        torch.cuda.empty_cache()
        capi.release_cached_memory()

    def run(worker_id, first_block):
        g = workers[worker_id]
        stream = torch.cuda.Stream(device=g - 1)
        staging = {}  # the worker's pinned upload buffer (load_block_device)
        # blocks of equal shape share the RL context and the taper's FFT engine (MI_NO_DECON_PLAN: rebuild them per block)
        plan = None if os.environ.get("MI_NO_DECON_PLAN") else D.DeconPlan(g)
        try:
            for n in range(first_block, num_blocks + 1):
                if n in resident or not claim(n):
                    continue                                                               # finished or being worked on elsewhere
                run_block(n, g, stream, staging, plan)
        finally:
            with lock:
                timing["worker_end_s"].append(time.perf_counter() - t_blocks0)
            if plan is not None:
                with torch.cuda.device(g - 1):
                    plan.close()

    def run_block(n, g, stream, staging, plan):
        p1, p2 = block.p1[n - 1], block.p2[n - 1]
        bl_xyz = tuple(int(b) - int(a) + 1 + 2 * int(q) for a, b, q in zip(p1, p2, pad))   # padded block, [x y z]
        fshape = ref_grid = None
        if args.use_fft:
            fshape = L.block_fft_shape(bl_xyz, block.fft_shape)
            ref_grid = tuple(L.next_fast_len(bl_xyz))                                      # LsDeconv.m:405-419: the reference's grid
        blk = L.Block(block.x, block.y, block.z, block.nx, block.ny, block.nz, *pad, fft_shape=fshape, psf_grid=ref_grid)
        t_a = time.perf_counter()
        with torch.cuda.device(g - 1), torch.cuda.stream(stream):
            # load_block on the device: the raw samples cross PCIe, conversion and symmetric padding happen there
            ev0 = torch.cuda.Event(enable_timing=True)
            ev1 = torch.cuda.Event(enable_timing=True)
            bl = L.load_block_device(vol, p1, p2, pad, torch.device("cuda", g - 1), staging)
            t_b = time.perf_counter()
            ev0.record()
            rawmax = None if int_input else float(bl.max())                               # LsDeconv.m:717-721
            out_of_memory = False
            try:
                t, lb, ub = L.process_block(bl, blk, psf_struct, args.numit, args.lambda_damping, args.stop_criterion, filt,
                                            args.clipval, g, plan=plan)
            except (torch.cuda.OutOfMemoryError, capi.MiError) as e:
                if isinstance(e, capi.MiError) and e.code != capi.MI_ERR_NOMEM:
                    raise
                out_of_memory = True               # (recovery happens OUTSIDE the handler: the exception's traceback keeps the failed
                #                                    attempt's frames -- and their tensors -- alive for as long as it is being handled)
            if out_of_memory:
                # the device is full: the resident cores go back to being bricks (they are copies of what the cache folder holds,
                # or are written now), nothing more is kept on the device, and the block is tried once more
                bl = t = None
                evict_resident(g)
                bl = L.load_block_device(vol, p1, p2, pad, torch.device("cuda", g - 1), staging)
                t, lb, ub = L.process_block(bl, blk, psf_struct, args.numit, args.lambda_damping, args.stop_criterion, filt,
                                            args.clipval, g, plan=plan)
            core = t[pad[2]:t.shape[0] - pad[2] or None, pad[1]:t.shape[1] - pad[1] or None, pad[0]:t.shape[2] - pad[0] or None]
            core = core.contiguous()                                                       # strip pads, LsDeconv.m:750-752
            assert tuple(core.shape) == core_shape(n), "[remove padding]: Output block size mismatch!"
            ev1.record()
            t_c = time.perf_counter()
            nbytes = core.numel() * 4
            with lock:
                stays = res_used[g] + nbytes <= res_budget
                if stays:
                    res_used[g] += nbytes
            host = None
            behind = trail_bricks and stays                                                # its brick is written behind the workers
            if (keep_bricks and not behind) or not stays:
                host = stage_get()                                                         # (waits while the writers are behind)
            t_d = time.perf_counter()
            if host is not None:
                view = host[:core.numel()].view(core.shape)
                view.copy_(core, non_blocking=True)
            if stays and core.untyped_storage().nbytes() > nbytes:
                core = core.clone()            # a block padded along z only: `core` is a VIEW that would keep the whole padded block alive
            stream.synchronize()
            if stays:
                resident[n] = core                                                         # (complete: the stream has been waited for)
                resident_dev[n] = g
        t_e = time.perf_counter()
        with lock:                                                                         # where a block's time goes (summary at the end)
            timing["blocks"] += 1
            timing["host_in_device_s"] = timing.get("host_in_device_s", 0.0) + t_c - t_b
            timing["box_read_s"] += t_b - t_a
            timing["device_ms"] += ev0.elapsed_time(ev1)
            timing["wait_buffer_s"] += t_d - t_c
            timing["d2h_s"] += t_e - t_d
        merge_min_max(lb, ub, rawmax)
        # the brick is compressed and written behind the worker's back, straight from the pinned buffer, its chunks side by side
        # on the codec pool; the buffer returns to the pool when the file is complete
        if host is not None:
            if not keep_bricks:                                                            # (a core that found no room on the device)
                try:
                    os.close(os.open(brick_path(n), os.O_CREAT | os.O_WRONLY))
                except OSError:
                    pass
            with lock:
                fut = writers.submit(save_brick, n, view.numpy(), host, lb, ub)
                pending.append(fut)
                brick_future[n] = fut
        elif behind:
            with lock:
                fut = trailers.submit(trail_brick, n, lb, ub)
                trailing.append(fut)
                brick_future[n] = fut
        log.info(f"block {n}/{num_blocks} done on GPU {g}: stats [{lb:.4g}, {ub:.4g}]" + ("" if host is not None else " (kept on the device only)"))

    def save_brick(n, arr, host, lb, ub):
        try:
            brick = brick_path(n)
            with open(brick.with_suffix(".json"), "w") as f:                               # the block's own clip range
                json.dump({"lb": lb, "ub": ub}, f)
            brickio.save_lz4(brick.with_suffix(".lz4.tmp"), arr, chunk_size=brick_chunk, pool=codec)   # LsDeconv.m:805-806
            try:
                os.replace(brick.with_suffix(".lz4.tmp"), brick)
            except FileNotFoundError:
                # another process that shares the cache folder reaped this claim (it looked stale to it): the block is redone by
                # whoever claims it next -- the same race exists in LsDeconv.m:621-640 -- and nothing of this run is lost but time
                log.warning(f"block {n}: the claim was removed by another process while its brick was being written; skipped")
        finally:
            stage_free.put(host)

    # ---- the deconvolution rounds (LsDeconv.m:618-660): stale claims and half-written bricks of an earlier run are removed,
    # the workers take what is missing from --start-block on; whatever is still missing afterwards (blocks below the start
    # block, claims of a process that died) is taken by another round from block 1
    start = max(1, min(int(args.start_block), num_blocks))
    pending = []
    timing = {"blocks": 0, "box_read_s": 0.0, "device_ms": 0.0, "wait_buffer_s": 0.0, "d2h_s": 0.0, "worker_end_s": [], "flush_s": 0.0,
              "bricks_trailed": 0, "bricks_not_written": 0}
    # (trail_brick; ended before the assembly.  ONE writer by default: four of them wrote 59 of 75 bricks of the 17-GB probe and the
    # workers took 5.5 s, as long as with every brick complete; one writes 12 and the workers take 3.1 s -- 2.7 s without any brick:
    # what slows the workers is the host's memory traffic of the writers, not waiting for them)
    trailers = ThreadPoolExecutor(max_workers=max(1, int(os.environ.get("MI_DECWRAP_TRAIL_THREADS", max(1, n_writers // 4)))))
    t_blocks0 = time.perf_counter()
    log.info(f"set-up (imports, device, PSF, block grid, cache folder): {t_blocks0 - t_main0:.1f} s")
    with ThreadPoolExecutor(max_workers=n_writers) as writers:                             # one brick file each; liblz4 runs outside the GIL
        while True:
            missing = 0
            for n in range(1, num_blocks + 1):
                if brick_complete(n):
                    if n >= start:
                        log.info(f"block {n}/{num_blocks} taken from the cache")
                    continue
                missing += 1
                if n >= start and claim_is_dead(n):
                    for q in (brick_path(n), brick_path(n).with_suffix(".lz4.tmp"), brick_path(n).with_suffix(".claim")):
                        try:
                            q.unlink()
                        except FileNotFoundError:
                            pass
            if missing == 0:
                break
            _warm_thread.join(timeout=120.0)       # (its plan is closed before the workers make their first allocations)
            done_before = timing["blocks"]
            with ThreadPoolExecutor(max_workers=len(workers)) as pool:
                for f in [pool.submit(run, w, min(num_blocks, start + w)) for w in range(len(workers))]:
                    f.result()                                                             # re-raises a worker's exception
            t_f = time.perf_counter()
            for f in pending:
                f.result()
            timing["flush_s"] += time.perf_counter() - t_f
            if timing["blocks"] == done_before and start == 1:
                time.sleep(2.0)                                                            # the missing blocks are live claims of another process
            pending.clear()
            if start > 1 and all(brick_complete(n) for n in range(start, num_blocks + 1)):
                break                                                                      # a helper machine is done with its share
            start = 1
    # bricks of resident cores that no writer has begun are not written (trail_brick); the ones in flight finish
    stop_trailing.set()
    trailers.shutdown(wait=True, cancel_futures=True)
    for f in trailing:
        if not f.cancelled():
            f.result()                                                                     # (re-raises a writer's exception)
    timing["bricks_not_written"] = len(trailing) - timing["bricks_trailed"]
    trailing.clear()
    t_blocks = time.perf_counter() - t_blocks0
    if timing["blocks"]:
        log.info(f"{timing['blocks']} blocks in {t_blocks:.1f} s: device work {timing['device_ms'] / 1e3:.1f} s "
                 f"({timing['device_ms'] / 1e3 / max(t_blocks, 1e-9) * 100:.0f} % of that time), box reads {timing['box_read_s']:.1f} s, "
                 f"D2H of the cores {timing['d2h_s']:.1f} s, waiting for a free core buffer {timing['wait_buffer_s']:.1f} s "
                 f"(summed over {len(workers)} workers); the workers ended after "
                 f"{', '.join(f'{v:.1f}' for v in sorted(timing['worker_end_s']))} s, the last bricks were written {timing['flush_s']:.1f} s later"
                 + (f"; bricks behind the workers: {timing['bricks_trailed']} written, {timing['bricks_not_written']} not begun when the last "
                    f"block was done" if trail_bricks else ""))
    main.last_timing = dict(timing, blocks_wall_s=t_blocks)
    merge_min_max()                                                                        # (helpers too: rawmax of their last blocks)
    if int(args.start_block) != 1:
        log.info("--start-block > 1: the blocks are in the cache; the process started with --start-block 1 assembles the output")
        codec.shutdown(wait=True)
        return 0                                                                           # LsDeconv.m:579-583, 667-670

    # ---- postprocess_save (LsDeconv.m:950-1111): clip range over all blocks, target scale, then one z slab of bricks at a time:
    # rescale on the device (mi_rescale_block = load_slab_lz4.cpp:134-157), optional flip, slices out.  Only the integer slab is
    # held in memory.
    merge_min_max()
    lo, hi = state["deconvmin"], state["deconvmax"]
    for n in range(1, num_blocks + 1):                                                     # bricks placed by hand carry their own range
        try:
            with open(brick_path(n).with_suffix(".json")) as f:
                st = json.load(f)
            lo, hi = min(lo, st["lb"]), max(hi, st["ub"])
        except (OSError, ValueError, KeyError):
            pass
    rawmax = state["rawmax"]
    scal = L.output_scale(rawmax, args.convert_to_8bit, args.convert_to_16bit)
    with open(out_dir / "min_max.json", "w") as f:
        json.dump({"deconvmin": float(lo), "deconvmax": float(hi), "rawmax": rawmax, "scal": scal}, f)
    bits = 8 if scal <= 255 else 16
    tiff_out = not args.input.is_file() and bool(brickio.list_tiff_series(args.input))
    # whole-volume *.npy copies (float32 result, integer stack): always for *.npy inputs, for TIFF inputs only while they are small
    # (a TIFF folder of more than 512^3 voxels gets its slices and nothing else, like the reference: the float32 copy of a 4-GB stack
    #  is 8.6 GB through the page cache)
    want_npy = os.environ.get("MI_DECWRAP_NPY", "1" if (not tiff_out or sz * sy * sx <= (1 << 27)) else "0") == "1"
    npy_f = np.lib.format.open_memmap(out_dir / "deconvolved.npy", mode="w+", dtype=np.float32, shape=(sz, sy, sx)) if want_npy else None
    npy_i = (np.lib.format.open_memmap(out_dir / f"deconvolved_{bits}bit.npy", mode="w+", dtype=np.uint8 if bits == 8 else np.uint16,
                                       shape=(sz, sy, sx)) if want_npy else None)
    dev = torch.device("cuda", args.gpu_indices[0] - 1)
    per_slab = block.nx * block.ny
    n_tif = 0
    out_dtype = np.uint8 if bits == 8 else np.uint16
    slab_buf = None                      # one integer slab, reused by every z slab of bricks (fresh pages cost seconds per slab)
    if prep_thread is not None:
        prep_thread.join()
    if prep.get("dtype") == out_dtype:
        slab_buf, q_host = prep["slab"], prep["q_host"]
    else:
        q_host = [torch.empty(core_max, dtype=torch.uint8 if bits == 8 else torch.uint16, pin_memory=True) for _ in range(2)]
    prep.clear()
    n_resident = [0]

    def read_brick(n):
        """brick -> a pinned core buffer (chunks decompressed side by side on the codec pool)"""
        host = stage_get()
        try:
            core = brickio.load_lz4(brick_path(n), pool=codec, out=host.numpy())
        except BaseException:
            stage_free.put(host)
            raise
        if core.shape != core_shape(n):
            stage_free.put(host)
            raise RuntimeError(f"brick {brick_path(n)} has shape {core.shape}, block {n} needs {core_shape(n)}")
        return host, core

    for iz in range(block.nz):
        ids = list(range(iz * per_slab + 1, (iz + 1) * per_slab + 1))
        z1, z2 = int(block.p1[ids[0] - 1][2]), int(block.p2[ids[0] - 1][2])
        if tiff_out and all((out_dir / f"img_{z:06d}.tif").exists() for z in range(z1, z2 + 1)) and not want_npy:
            continue                                                                       # resume: this slab's slices exist (:1037-1054)
        # TIFF slices and nothing else to write: the integer slab is put together ON THE DEVICE and its slices are deflated there
        # (brickio.save_tiff_series_device: 14-18 GB/s from device memory to files where the host's cores deflate 2.2 GB/s,
        # profiles/r05_tiff_device.txt); MI_DECWRAP_TIFF_DEVICE=0, or no room for the slab: through the host as before
        d_slab = None
        if tiff_out and npy_i is None and npy_f is None and os.environ.get("MI_DECWRAP_TIFF_DEVICE", "1") != "0":
            need = (z2 - z1 + 1) * sy * sx * (1 if bits == 8 else 2)
            if torch.cuda.mem_get_info(dev)[0] > need + (4 << 30):
                try:
                    d_slab = torch.empty((z2 - z1 + 1, sy, sx), dtype=torch.uint8 if bits == 8 else torch.uint16, device=dev)
                except torch.cuda.OutOfMemoryError:
                    d_slab = None
        if d_slab is None and slab_buf is None:
            slab_buf = np.empty((int(block.z), sy, sx), out_dtype)
        slab = slab_buf[:z2 - z1 + 1] if d_slab is None else None
        with ThreadPoolExecutor(max_workers=2) as readers, ThreadPoolExecutor(max_workers=2) as copiers:
            # bricks are read ahead of the device; the integer cores are copied into the slab behind it
            placed = [None, None]
            ahead = max(1, min(2, n_stage - 1))
            from_disk = [k for k, n in enumerate(ids) if n not in resident]               # (the others wait on their device)
            futs = {k: readers.submit(read_brick, ids[k]) for k in from_disk[:ahead]}
            nxt = ahead
            for k, n in enumerate(ids):
                host = None
                if n in resident:
                    d_core = resident.pop(n)
                    shape, size = tuple(d_core.shape), d_core.numel()
                    n_resident[0] += 1
                else:
                    host, core = futs.pop(k).result()
                    if nxt < len(from_disk):
                        futs[from_disk[nxt]] = readers.submit(read_brick, ids[from_disk[nxt]])
                        nxt += 1
                    shape, size = core.shape, core.size
                p1, p2 = block.p1[n - 1], block.p2[n - 1]
                box = (slice(int(p1[2]) - z1, int(p2[2]) - z1 + 1), slice(int(p1[1]) - 1, int(p2[1])), slice(int(p1[0]) - 1, int(p2[0])))
                if npy_f is not None:
                    npy_f[int(p1[2]) - 1:int(p2[2]), box[1], box[2]] = core if host is not None else d_core.cpu().numpy()
                # rescale on the device (load_slab_lz4.cpp:134-157): a resident core where it is, a brick's via a pinned buffer;
                # pinned integers down
                if placed[k & 1] is not None:
                    placed[k & 1].result()                                                 # this integer buffer's last core is in the slab
                qh = q_host[k & 1][:size].view(shape)
                with torch.cuda.device(dev if host is not None else d_core.device):
                    if host is not None:
                        d_core = host[:size].view(shape).to(dev, non_blocking=True)          # (the pinned tensor itself: a true async copy)
                    q = D.rescale_block(d_core, scal, args.signal_amp, lo, hi)
                    if d_slab is None:
                        qh.copy_(q, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                ev.synchronize()
                del d_core
                if host is not None:
                    stage_free.put(host)
                if d_slab is not None:
                    with torch.cuda.device(dev):
                        d_slab[box] = q.to(dev)                                                # (the integer core into its box of the device slab)
                    del q
                    continue
                placed[k & 1] = copiers.submit(slab.__setitem__, box, qh.numpy())
            for fu in placed:
                if fu is not None:
                    fu.result()
        if d_slab is not None:
            with torch.cuda.device(dev):
                if args.flip:
                    d_slab = d_slab.flip(1).contiguous()                                   # R = flip(R, 2): the y axis (LsDeconv.m:1097-1099)
                try:
                    n_tif += brickio.save_tiff_series_device(out_dir, d_slab, first_index=z1)   # img_%06d.tif, existing slices kept
                except Exception as e:             # (no pinned memory left, say: the slab is complete -- its slices through the host)
                    log.warning(f"slices {z1}..{z2} could not be deflated on the device ({e}); writing them from the host")
                    n_tif += brickio.save_tiff_series(out_dir, d_slab.cpu().numpy(), first_index=z1)
            del d_slab
            continue
        if args.flip:
            slab = np.ascontiguousarray(slab[:, ::-1])                                     # R = flip(R, 2): the y axis (LsDeconv.m:1097-1099)
        if npy_i is not None:
            npy_i[z1 - 1:z2] = slab
        if tiff_out:
            n_tif += brickio.save_tiff_series(out_dir, slab, first_index=z1)               # img_%06d.tif, existing slices kept
    for m in (npy_f, npy_i):
        if m is not None:
            m.flush()
    del npy_f, npy_i
    if tiff_out:
        log.info(f"wrote {n_tif} TIFF slices to {out_dir}")
    codec.shutdown(wait=True)
    main.last_timing["assembly_wall_s"] = time.perf_counter() - t_blocks0 - t_blocks
    main.last_timing["cores_from_device"] = n_resident[0]
    log.info(f"assembly: {n_resident[0]} of {num_blocks} cores came straight from device memory, the others from their bricks")
    # LsDeconv.m:286-296 removes the block cache once the output is complete.  Tens of gigabytes of bricks take seconds to unlink
    # (2.8 s of a 13-s run on the 17-GB probe): the folder is renamed -- from then on a rerun starts afresh, like after the
    # reference's rmdir -- and removed by a detached child; MI_DECWRAP_SYNC_CLEANUP=1 waits for it instead.
    t_clean0 = time.perf_counter()
    doomed = cache.with_name(cache.name + f".removing.{os.getpid()}")
    try:
        os.replace(cache, doomed)
        if os.environ.get("MI_DECWRAP_SYNC_CLEANUP") == "1":
            shutil.rmtree(doomed, ignore_errors=True)
        else:
            import subprocess
            subprocess.Popen(["rm", "-rf", str(doomed)], stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                             start_new_session=True)
    except OSError:
        shutil.rmtree(cache, ignore_errors=True)
    if hasattr(vol, "close"):
        vol.close()
    main.last_timing["cleanup_s"] = time.perf_counter() - t_clean0
    main.last_timing["main_s"] = time.perf_counter() - t_main0
    log.info(f"wrote {out_dir} (scale {scal:g}, clip [{lo:.4g}, {hi:.4g}])")
    return 0


if __name__ == "__main__":
    sys.exit(main())
