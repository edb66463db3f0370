#!/usr/bin/env python3
"""Deconvolution launcher: the command line of the reference's LsDeconvolveMultiGPU/decwrap.py (:230-327) on top
of the MI355X library -- no MATLAB script is written, no MATLAB is spawned (decwrap.py:408-494); the blocks are
processed in this process through ``ipp_amd.lsdeconv.process_block``.

Input: a folder of 2-D slices (``*.tif`` / ``*.tiff`` like the reference, LsDeconv.m:585-588, or ``*.npy``) or one ``*.npy``
volume (Z, Y, X).  Output: ``<input>/deconvolved/deconvolved.npy`` (float32; + ``deconvolution_config.json`` like
decwrap.py:474-478, ``min_max.json``) and the rescaled integer stack ``deconvolved_{8,16}bit.npy`` that the reference's
postprocess_save writes as a TIFF series (LsDeconv.m:950-1100: percentile clip range of all blocks, amplification, round,
clamp; ``mi_rescale_block``); for a TIFF input also the ``img_%06d.tif`` series itself (LsDeconv.m:1120-1145).
Every finished block is kept as ``bl_<n>.lz4`` in the cache folder (``--cache-drive``, default ``<out>/cache``) in the
reference's brick format (save_lz4_mex.c), so an interrupted run resumes with the blocks that are missing (LsDeconv.m:695-705;
``--no-resume`` starts over); the cache is removed after a complete run like LsDeconv.m:286-296.
TIFF series / LZ4 brick cache / resume of the reference are I/O rows outside this hot path.
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import sys
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
    p.add_argument("--gpu-workers-per-gpu", type=int, default=2,
                   help="workers (own stream + pinned staging) per GPU: two overlap one block's PCIe / host staging with another's kernels")
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


def load_volume(path: Path):
    import numpy as np
    if path.is_file():
        return np.load(path, mmap_mode="r")
    files = sorted(path.glob("*.npy"))
    if files:
        return np.stack([np.load(f) for f in files])
    from ipp_amd import brickio
    if brickio.list_tiff_series(path):
        return brickio.load_tiff_series(path)
    raise RuntimeError(f"no *.npy / *.tif slices in {path}")


def main(argv=None):
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
    vol = load_volume(args.input)
    sz, sy, sx = vol.shape
    dz = args.dz if args.dz else args.dxy
    psf = P.LsMakePSF(args.dxy * 1000.0, dz * 1000.0, args.na, args.rf, float(args.lambda_ex), float(args.lambda_em),
                      float(args.fcyl), args.slitwidth)                                    # LsDeconv.m:160 (nm units)
    psf_struct = D.make_psf_struct(psf)
    log.info(f"PSF size (x y z): {psf.shape[::-1]}")
    filt = L.Filter(tuple(args.gaussian_sigma), tuple(args.gaussian_filter_size), 0.0, args.destripe_sigma,
                    args.regularize_interval, args.use_fft, args.adaptive_psf)
    gpu = args.gpu_indices[0]
    # workers that share a device share its memory (decwrap.py:133-169 divides by the workers per GPU as well)
    per_dev = max(1, args.gpu_workers_per_gpu) * max(1, args.gpu_indices.count(gpu))
    bmax = args.block_size_max or L.estimate_block_size_max(gpu - 1, n_real=3, n_complex=2 if args.use_fft else 0) // per_dev
    block = L.autosplit((sx, sy, sz), psf.shape[::-1], filt, bmax, args.numit)
    log.info(f"block grid {block.nx} x {block.ny} x {block.nz}, core ({block.x} {block.y} {block.z}), "
             f"pad ({block.x_pad} {block.y_pad} {block.z_pad}), fft_shape {block.fft_shape}")
    out = np.zeros(vol.shape, np.float32)
    pad = (block.x_pad, block.y_pad, block.z_pad)
    # One worker per entry of the device list (--gpu-indices x --gpu-workers-per-gpu, like the reference's pool of parfeval
    # workers, LsDeconv.m:620-668): blocks are independent, a worker takes the next unprocessed block, runs it on its device
    # and on its own stream, and writes the core of the result into the output volume.
    import shutil
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from ipp_amd import brickio
    cache = Path(args.cache_drive) if args.cache_drive else out_dir / "cache"
    if not args.resume and cache.exists():
        shutil.rmtree(cache)                                                               # LsDeconv.m:136-139
    cache.mkdir(parents=True, exist_ok=True)
    workers = [g for g in args.gpu_indices for _ in range(max(1, args.gpu_workers_per_gpu))]
    # --start-block only matters to the reference's multi-process start-up; here every block is either taken from the cache
    # or processed
    todo = [(n, p1, p2) for n, (p1, p2) in enumerate(zip(block.p1, block.p2), start=1)]
    lock = threading.Lock()
    stats = []

    def run(worker_id):
        g = workers[worker_id]
        stream = torch.cuda.Stream(device=g - 1)
        staging = {}  # pinned host buffers by core shape: D2H at PCIe rate instead of page-faulting a fresh pageable array
        # blocks of equal shape share the RL context and the taper's FFT engine (MI_NO_DECON_PLAN: rebuild them per block)
        plan = None if os.environ.get("MI_NO_DECON_PLAN") else D.DeconPlan(g)
        try:
            run_blocks(g, stream, staging, plan)
        finally:
            if plan is not None:
                with torch.cuda.device(g - 1):
                    plan.close()

    def run_blocks(g, stream, staging, plan):
        while True:
            with lock:
                if not todo:
                    return
                n, p1, p2 = todo.pop(0)
            brick = cache / f"bl_{n}.lz4"
            if brick.exists() and brick.stat().st_size > 0:                                # resume: LsDeconv.m:695-705, 799-801
                core = brickio.load_lz4(brick)
                if core.shape == (p2[2] - p1[2] + 1, p2[1] - p1[1] + 1, p2[0] - p1[0] + 1):
                    out[p1[2] - 1:p2[2], p1[1] - 1:p2[1], p1[0] - 1:p2[0]] = core
                    with open(brick.with_suffix(".json")) as f:
                        st = json.load(f)
                    with lock:
                        stats.append((st["lb"], st["ub"]))
                    log.info(f"block {n}/{len(block.p1)} taken from the cache")
                    continue
            bl_xyz = tuple(int(b) - int(a) + 1 + 2 * int(q) for a, b, q in zip(p1, p2, pad))   # padded block, [x y z]
            fshape = None
            if args.use_fft:
                smooth, native = L.next_fast_len(bl_xyz), L.native_fft_shape(bl_xyz)
                fshape = native if np.prod(native) <= 1.3 * np.prod(smooth) else smooth
            blk = L.Block(block.x, block.y, block.z, block.nx, block.ny, block.nz, *pad, fft_shape=fshape)
            with torch.cuda.device(g - 1), torch.cuda.stream(stream):
                # load_block on the device: the raw samples cross PCIe, conversion and symmetric padding happen there
                bl = L.load_block_device(vol, p1, p2, pad, torch.device("cuda", g - 1), staging)
                t, lb, ub = L.process_block(bl, blk, psf_struct, args.numit, args.lambda_damping, args.stop_criterion, filt,
                                            args.clipval, g, plan=plan)
                core = t[pad[2]:t.shape[0] - pad[2] or None, pad[1]:t.shape[1] - pad[1] or None, pad[0]:t.shape[2] - pad[0] or None]
                core = core.contiguous()                                                   # strip pads, LsDeconv.m:750-752
                if tuple(core.shape) not in staging:
                    staging[tuple(core.shape)] = torch.empty(core.shape, dtype=torch.float32, pin_memory=True)
                host = staging[tuple(core.shape)]
                host.copy_(core, non_blocking=True)
                stream.synchronize()
                core = host.numpy()
            box = (slice(p1[2] - 1, p2[2]), slice(p1[1] - 1, p2[1]), slice(p1[0] - 1, p2[0]))
            out[box] = core                                                                # disjoint boxes: no lock needed
            # the brick cache (resume) is written behind the worker's back: LZ4 of a float32 core takes longer than its kernels
            with lock:
                stats.append((lb, ub))
                pending.append(writers.submit(save_brick, brick, box, lb, ub))
            log.info(f"block {n}/{len(block.p1)} done on GPU {g}: stats [{lb:.4g}, {ub:.4g}]")

    def save_brick(brick, box, lb, ub):
        with open(brick.with_suffix(".json"), "w") as f:                                   # the block's clip range (min_max.mat entry)
            json.dump({"lb": lb, "ub": ub}, f)
        brickio.save_lz4(brick.with_suffix(".lz4.tmp"), np.ascontiguousarray(out[box]))    # LsDeconv.m:805-806
        os.replace(brick.with_suffix(".lz4.tmp"), brick)

    pending = []
    with ThreadPoolExecutor(max_workers=max(2, min(8, (os.cpu_count() or 4) // 2))) as writers:   # liblz4 runs outside the GIL
        with ThreadPoolExecutor(max_workers=len(workers)) as pool:
            for f in [pool.submit(run, w) for w in range(len(workers))]:
                f.result()                                                                 # re-raises a worker's exception
        for f in pending:
            f.result()
    lo = min((s[0] for s in stats), default=np.inf)
    hi = max((s[1] for s in stats), default=-np.inf)
    np.save(out_dir / "deconvolved.npy", out)
    # postprocess_save (LsDeconv.m:996-1024, 1091-1093): rawmax from the input class, target scale, rescale every slab with the
    # clip range [deconvmin, deconvmax] collected over the blocks
    if np.issubdtype(vol.dtype, np.integer):
        rawmax = float(np.iinfo(vol.dtype).max)
    else:
        rawmax = float(np.max(vol))
    scal = L.output_scale(rawmax, args.convert_to_8bit, args.convert_to_16bit)
    with open(out_dir / "min_max.json", "w") as f:
        json.dump({"deconvmin": float(lo), "deconvmax": float(hi), "rawmax": rawmax, "scal": scal}, f)
    bits = 8 if scal <= 255 else 16
    out_int = np.empty(vol.shape, np.uint8 if bits == 8 else np.uint16)
    dev = torch.device("cuda", args.gpu_indices[0] - 1)
    slab = max(1, (1 << 28) // max(1, sy * sx))                                           # <= 1 GiB of float32 per slab
    for z0 in range(0, sz, slab):
        t = torch.from_numpy(out[z0:z0 + slab]).to(dev)
        q = D.rescale_block(t, scal, args.signal_amp, lo, hi)
        out_int[z0:z0 + slab] = q.cpu().numpy()
    if args.flip:
        out_int = np.ascontiguousarray(out_int[:, ::-1])                                   # R = flip(R, 2): the y axis (LsDeconv.m:1097-1099)
    np.save(out_dir / f"deconvolved_{bits}bit.npy", out_int)
    if not args.input.is_file() and brickio.list_tiff_series(args.input):
        n_tif = brickio.save_tiff_series(out_dir, out_int)                                 # img_%06d.tif, existing slices kept
        log.info(f"wrote {n_tif} TIFF slices to {out_dir}")
    shutil.rmtree(cache, ignore_errors=True)                                               # LsDeconv.m:286-296
    log.info(f"wrote {out_dir / 'deconvolved.npy'} and deconvolved_{bits}bit.npy (scale {scal:g}, clip [{lo:.4g}, {hi:.4g}])")
    return 0


if __name__ == "__main__":
    sys.exit(main())
