#!/usr/bin/env python3
"""Headline benchmark: RL-deconvolution throughput (Gvoxel*iterations/s) of the MI355X hot path.

    python bench.py --gpus N --steps K --warmup W

A step is one Richardson-Lucy iteration (forward blur + ratio, adjoint blur + update; decon.m:162-186) over
the whole resident volume.  Workload at N = 1: BASELINE.json config C3 -- 2048 x 2048 x 512 fp32 volume,
31 x 31 x 61 light-sheet PSF, deconFFT semantics (circular on fft_shape = volume shape), lambda = 0,
regularize_interval = 0, stop_criterion = 0; synthetic seeded data already resident in HBM when the timed
region starts.  At N > 1 the same volume is cut into N slabs along Y (one rank per GPU, RCCL halo exchange
over xGMI twice per iteration) => strong scaling.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {  # name: (volume (z,y,x), psf (z,y,x))
    "c1": ((64, 256, 256), (15, 9, 9)),
    "c2": ((256, 1024, 1024), (31, 15, 15)),
    "c3": ((512, 2048, 2048), (61, 31, 31)),
    "c4": ((1024, 4096, 4096), (127, 63, 63)),   # whole on ONE device: 258 GB of the 288 (--workload c4, or the c4_single row at N = 1)
}
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
ALGO_BYTES_PER_VOXEL_ITER = 48  # RL FFT path, SURVEY.md section 8d / DESIGN.md


def make_psf(kshape, gaussian=False):
    if gaussian:  # BASELINE config 1: a Gaussian PSF (an outer product of three lines: the direct engine's single-pass separable kernel)
        import numpy as np
        ax = [np.exp(-0.5 * ((np.arange(k) - (k - 1) / 2.0) / max((k - 1) / 6.0, 0.5)) ** 2) for k in kshape]
        p = ax[0][:, None, None] * ax[1][None, :, None] * ax[2][None, None, :]
        return (p / p.sum()).astype(np.float32)
    from ipp_amd import psf as P
    base, _ = P.generate_psf(lambda_em=525.0, lambda_ex=488.0, numerical_aperture=0.4, dxy=100.0, dz=250.0,
                             refractive_index=1.42, f_cylinder_lens=240.0, slit_width=12.0)
    return P.resample_psf(base, kshape)


def make_volume(shape, device, seed=1234):
    """Seeded synthetic volume generated on the device: background U(0.01,0.02) + sparse bright beads
    (SURVEY.md 8d recipe without the host-side PSF blur, which would take minutes at this size)."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    vol = torch.empty(shape, dtype=torch.float32, device=device).uniform_(0.01, 0.02, generator=g)
    n = vol.numel()
    nb = max(1, n // 4096)
    idx = torch.randint(0, n, (nb,), generator=g, device=device)
    amp = torch.empty(nb, dtype=torch.float32, device=device).uniform_(0.2, 1.0, generator=g)
    vol.view(-1).index_put_((idx,), amp, accumulate=True)
    return vol


def pmc_traffic(pass_name, workload):
    """(HBM bytes per launch of the dominant kernel, the file they come from): the committed PMC passes of this same command
    (profiles/r05_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in two separate runs, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes).  Counters cannot be collected from inside the timed process, so this is a CONSTANT of that
    profile, quoted beside the live launch time -- `roofline.traffic_source` says so in the JSON line.  (None, None) when the
    profile does not hold the kernel."""
    if workload != "c3":
        return None, None
    kern, name = None, None
    for name in ("r05_pmc_traffic.json", "r04_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                kern = json.load(f)["kernels"]
            break
        except (OSError, KeyError, ValueError):
            continue
    if kern is None:
        return None, None
    # (the pair-interleaved layout runs k_y_pair / k_z_pair_pipe, the plain one k_y_pass / k_z_conv_pipe)
    # (k_x_fused_pipe<.., 0> is the fused pass; modes 1 / 2 are the forward-only / inverse-only launches around a chain)
    want = {"z_conv": (("k_z_pair_pipe<", ">"), ("k_z_conv_pipe<", ">")), "x_fused": (("k_x_fused_pipe<", ", 0>"), ("k_x_fused_pipe<", ">")),
            "y_forward": (("k_y_pair<", "false>"), ("k_y_pass<", "false>")),
            "y_inverse": (("k_y_pair<", "true>"), ("k_y_pass<", "true>"))}[pass_name]
    for pre, post in want:
        for kname, v in kern.items():
            if kname.startswith(pre) and kname.endswith(post):
                return round(v["hbm_bytes_per_launch"]), "profiles/" + name
    return None, None


def host_cores():
    """Host cores this process may use: the affinity mask, capped by the cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(vshape, kshape, seconds_budget=15.0, gaussian=False):
    """The oracle's deconFFT loop (oracle/rl_oracle.py:decon_fft_f32 -- scipy.fft, float32 / complex64, every host core)
    on a bounded sub-volume of the same workload: 1/8 of the volume (every extent halved) when one iteration fits the
    budget, else 1/64 (every extent quartered); as many iterations as fit the budget, at most 8."""
    import numpy as np
    from oracle import rl_oracle
    cores = host_cores()
    psf = make_psf(kshape, gaussian)
    rng = np.random.default_rng(1234)

    def run(shape, iters):
        vol = rng.uniform(0.01, 0.02, size=shape).astype(np.float32)
        otf = rl_oracle.otf_half_f32(psf, shape, workers=cores)
        t0 = time.perf_counter()
        rl_oracle.decon_fft_f32(vol, otf, iters, workers=cores)
        return time.perf_counter() - t0

    target = tuple(max(k + 3, v // 2) for v, k in zip(vshape, kshape))      # 1/8 of the volume
    probe = tuple(max(k + 3, v // 4) for v, k in zip(vshape, kshape))       # 1/64: calibrates the host
    run(probe, 1)                                                             # page in scipy / warm the thread pool
    t_probe = run(probe, 1)
    scale = float(np.prod(target)) / float(np.prod(probe))
    shape = target if t_probe * scale * 1.15 <= seconds_budget else probe
    t_iter = t_probe * (scale * 1.15 if shape == target else 1.0)           # expected seconds per iteration of the sample
    iters = max(1, min(8, int(seconds_budget / max(t_iter, 1e-3))))         # about 10-20 s of CPU work, at most 8 iterations
    dt = run(shape, iters)
    return {"value": float(np.prod(shape)) * iters / dt / 1e9, "unit": "Gvoxel*iter/s", "cores": cores, "kind": "port",
            "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS"),
            "sample": f"{iters} deconFFT iteration(s) (oracle/rl_oracle.py:decon_fft_f32, scipy.fft rfftn/irfftn, float32/complex64, "
                      f"workers={cores}) on a {shape[2]}x{shape[1]}x{shape[0]} sub-volume "
                      f"({float(np.prod(shape)) / float(np.prod(vshape)):.4f} of the workload) with the "
                      f"{kshape[2]}x{kshape[1]}x{kshape[0]} PSF, {dt:.1f} s"}


def block_stages(vshape, psf_np, dev):
    """what surrounds the iterations of one block, timed separately from the loop -- edgetaper_3d (decon.m:50,143)
    and the host <-> device hand-over of the block (gpuArray / gather, LsDeconv.m:906-948) through pinned memory."""
    import torch
    from ipp_amd import decon
    out = {}
    n = vshape[0] * vshape[1] * vshape[2]
    vol = make_volume(vshape, dev, seed=99)
    psf_t = torch.from_numpy(psf_np).to(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    try:
        decon.edgetaper_3d(vol, psf_t)                 # first call builds the slab engines (a deconvolution plan keeps them)
        torch.cuda.synchronize(dev)
        ev0.record()
        decon.edgetaper_3d(vol, psf_t)
        ev1.record()
        torch.cuda.synchronize(dev)
        out["edgetaper_ms"] = round(ev0.elapsed_time(ev1), 3)
    except Exception as e:
        out["edgetaper_ms"] = None
        sys.stderr.write(f"edgetaper timing unavailable: {e!r}\n")
    try:
        host = torch.empty(vshape, dtype=torch.float32, pin_memory=True)
        host.fill_(0.5)
        for key, fn in (("h2d_ms", lambda: vol.copy_(host, non_blocking=True)), ("d2h_ms", lambda: host.copy_(vol, non_blocking=True))):
            fn()
            torch.cuda.synchronize(dev)
            ev0.record()
            fn()
            ev1.record()
            torch.cuda.synchronize(dev)
            out[key] = round(ev0.elapsed_time(ev1), 3)
        out["pcie_GBps"] = round(4.0 * n / 1e9 / (0.5e-3 * (out["h2d_ms"] + out["d2h_ms"])), 1)
        del host
    except Exception as e:
        out.setdefault("h2d_ms", None)
        out.setdefault("d2h_ms", None)
        sys.stderr.write(f"PCIe timing unavailable: {e!r}\n")
    del vol
    torch.cuda.empty_cache()
    return out


def secondary_rows(vshape, kshape, psf_np, dev, gaussian, steps):
    """SURVEY.md 8(d)'s second row and the reference's own configuration of BASELINE config 2, both through `decon` (the entry
    decwrap.py calls per block, with a deconvolution plan as its workers keep one; edge taper skipped: the loop alone, like the
    headline).
      defaults    the headline workload with decwrap.py's defaults (decwrap.py:294-321): 6 iterations, regularize_interval = 3
                  (the 5-tap Gaussian of decon.m:157-161 before iteration 3), lambda = 0, deconFFT semantics;
      c2_spatial  BASELINE config 2 (1024 x 1024 x 256, 15 x 15 x 31 PSF) in decon.m's spatial flavour (zero boundary, psf.inv =
                  flipped PSF, decon.m:41-124), 20 iterations -- whatever engine mi_engine_select picks for it."""
    import torch
    from ipp_amd import decon
    out = {}
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(shape, psf, niter, reps, reg, **kw):
        bl = make_volume(shape, dev, seed=4321)
        with decon.DeconPlan(dev) as plan:
            decon.decon(bl, psf, niter, 0.0, 0.0, reg, plan=plan, skip_edgetaper=True, **kw)   # builds the plan
            torch.cuda.synchronize(dev)
            ev0.record()
            for _ in range(reps):
                decon.decon(bl, psf, niter, 0.0, 0.0, reg, plan=plan, skip_edgetaper=True, **kw)
            ev1.record()
            torch.cuda.synchronize(dev)
        del bl
        return ev0.elapsed_time(ev1) / (niter * reps)

    try:
        reps = max(1, steps // 6)
        ms = timed(vshape, psf_np, 6, reps, 3, use_fft=True, fft_shape=[vshape[2], vshape[1], vshape[0]])
        n = vshape[0] * vshape[1] * vshape[2]
        out["defaults"] = {"ms_per_step": round(ms, 4), "value": round(n / (ms * 1e-3) / 1e9, 4), "unit": "Gvoxel*iter/s",
                           "config": "decwrap.py defaults on the headline workload: 6 iterations per call, regularize_interval=3 "
                                     "(Gaussian before iteration 3), lambda=0, deconFFT semantics; device time of "
                                     f"{reps} call(s) of decon() on a kept plan, edge taper skipped"}
    except Exception as e:
        out["defaults"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    try:
        c2v, c2k = WORKLOADS["c2"]
        psf2 = make_psf(c2k)
        ms = timed(c2v, psf2, 20, 2, 0, use_fft=False)
        n = c2v[0] * c2v[1] * c2v[2]
        out["c2_spatial"] = {"ms_per_step": round(ms, 4), "value": round(n / (ms * 1e-3) / 1e9, 4), "unit": "Gvoxel*iter/s",
                             "engine": {1: "direct", 2: "fft"}.get(decon.engine_select(c2v, c2k), "?"),
                             "config": f"c2: {c2v[2]}x{c2v[1]}x{c2v[0]} volume, {c2k[2]}x{c2k[1]}x{c2k[0]} PSF, decon.m spatial flavour "
                                       "(zero boundary, psf.inv = flipped PSF), 20 iterations per call, 2 calls on a kept plan"}
    except Exception as e:
        out["c2_spatial"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    return out


def c4_single(dev, steps=3):
    """BASELINE config 4 WHOLE on one device (4096 x 4096 x 1024 voxels, 63 x 63 x 127 PSF, deconFFT semantics): 17.2 G voxels, 258 GB of
    the 288 -- the volume, two spectrum arrays, the real OTF.  The reference cannot hold such a block at all (2^31 - 1 elements,
    LsDeconv.m:308-385).  Device time of `steps` fused iterations after one warm-up iteration."""
    import torch
    from ipp_amd import capi, decon
    free_b, total_b = torch.cuda.mem_get_info(dev)
    if free_b < 265e9:
        return {"skipped": f"needs about 258 GB on the device: {free_b / 1e9:.1f} GB free of {total_b / 1e9:.1f} GB"}
    vshape, kshape = WORKLOADS["c4"]
    ctx = decon.RLContext(vshape, make_psf(kshape), None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
    bl = make_volume(vshape, dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx.iterate(bl, None, 1)
    torch.cuda.synchronize(dev)
    ev0.record()
    ctx.iterate(bl, None, steps)
    ev1.record()
    torch.cuda.synchronize(dev)
    ms = ev0.elapsed_time(ev1) / steps
    n = vshape[0] * vshape[1] * vshape[2]
    out = {"ms_per_step": round(ms, 3), "value": round(n / (ms * 1e-3) / 1e9, 4), "unit": "Gvoxel*iter/s", "steps": steps,
           "device_bytes_context": int(ctx.device_bytes), "device_bytes_volume": 4 * n,
           "iteration": {"algorithmic_bytes_per_voxel_iter": ALGO_BYTES_PER_VOXEL_ITER,
                         "achieved_GBps": round(ALGO_BYTES_PER_VOXEL_ITER * n / (ms * 1e-3) / 1e9, 1),
                         "frac": round(ALGO_BYTES_PER_VOXEL_ITER * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
           "config": f"c4 whole on one device: {vshape[2]}x{vshape[1]}x{vshape[0]} fp32 volume, {kshape[2]}x{kshape[1]}x{kshape[0]} PSF, "
                     "deconFFT semantics, lambda=0, fused iterations (mi_rl_iterate)"}
    try:
        out["pass_ms"] = {k: round(ctx.time_pass(k, bl, reps=2), 3) for k in ("y_forward", "z_conv", "y_inverse", "x_fused")}
    except Exception as e:
        out["pass_ms"] = {"error": repr(e)}
    del ctx, bl
    torch.cuda.empty_cache()
    capi.release_cached_memory()
    return out


def decwrap_e2e(shape=(1024, 2048, 2048), tiff=False):
    """The application around the hot path, end to end: decwrap.py (the reference's command line, decwrap.py:294-305 defaults: 6
    iterations, Gaussian pre-filter 13 x 13 x 25, regularisation every 3rd iteration, deconFFT flavour) on a synthetic uint16 volume
    in a memory-mapped *.npy under /tmp -- block grid, box reads, uploads, per-block deconvolution, percentiles, bricks behind the
    workers, assembly and integer rescale.  Wall time of decwrap.main; the volume's generation is not counted."""
    import shutil
    import tempfile
    import numpy as np
    from ipp_amd import decwrap
    root = tempfile.mkdtemp(prefix="bench_decwrap_", dir="/tmp")
    try:
        rng = np.random.default_rng(1)

        def scene(n):                                  # sparse beads on a noisy background
            sl = rng.integers(600, 700, size=(n,) + tuple(shape[1:]), dtype=np.uint16)
            idx = rng.integers(0, sl.size, size=sl.size // 2000)
            sl.reshape(-1)[idx] = rng.integers(5000, 60000, size=idx.size, dtype=np.uint16)
            return sl
        if tiff:                                       # the reference's own input form: a folder of 2-D TIFF slices (deflate)
            from ipp_amd import brickio
            src = os.path.join(root, "stack")
            for z0 in range(0, shape[0], 64):
                brickio.save_tiff_series(src, scene(min(64, shape[0] - z0)), first_index=z0 + 1)
        else:
            src = os.path.join(root, "vol.npy")
            vol = np.lib.format.open_memmap(src, mode="w+", dtype=np.uint16, shape=shape)
            for z in range(shape[0]):
                vol[z] = scene(1)[0]
            vol.flush()
            del vol
        keep = os.environ.get("MI_DECWRAP_NPY")
        os.environ["MI_DECWRAP_NPY"] = "0"             # (no whole-volume *.npy copies of the result: the streaming pipeline is what is timed)
        try:
            t0 = time.perf_counter()
            rc = decwrap.main(["-i", src, "-dxy", "0.422", "-dz", "1.0", "-ex", "488", "-em", "525", "--use-fft",
                               "-it", "6", "--block-size-max", "300000000", "--gpu-indices", "1", "--gpu-workers-per-gpu", "5"])
            dt = time.perf_counter() - t0
        finally:
            if keep is None:
                os.environ.pop("MI_DECWRAP_NPY", None)
            else:
                os.environ["MI_DECWRAP_NPY"] = keep
        tm = dict(getattr(decwrap.main, "last_timing", {}))
        n = float(shape[0]) * shape[1] * shape[2]
        return {"rc": rc, "wall_s": round(dt, 3), "value": round(n / dt / 1e6, 1), "unit": "Mvoxel/s end to end (6 iterations each)",
                "blocks": tm.get("blocks"), "blocks_phase_s": round(tm.get("blocks_wall_s", 0.0), 3),
                "assembly_phase_s": round(tm.get("assembly_wall_s", 0.0), 3), "bricks_written_behind_the_workers": tm.get("bricks_trailed"),
                "bricks_not_begun": tm.get("bricks_not_written"),
                "config": f"decwrap.py on a {shape[2]}x{shape[1]}x{shape[0]} uint16 volume ({n * 2 / 1e9:.1f} GB, " +
                          ("a folder of deflate TIFF slices in, deflate TIFF slices out (deflated on the device)" if tiff else "memory-mapped .npy") +
                          "), --use-fft, 6 iterations, default filters, --block-size-max 3e8, 5 workers on one GPU, MI_DECWRAP_BRICKS=trail (default)"}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def launch_ranks(n, argv):
    """``python bench.py --gpus N`` from a plain shell: start the N ranks as a CHILD process tree (torch.distributed.run, one rank
    per GPU, rendezvous on 127.0.0.1), relay rank 0's JSON line and return the child's exit code.  Nothing in this process has
    touched the GPU (no torch import, no HIP call) -- the reference fans its workers out itself too (LsDeconv.m:643-654)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout.splitlines():
        if line.startswith('{"metric"'):
            print(line, flush=True)
        elif line.strip():
            sys.stderr.write(line + "\n")
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--engine", default="auto", choices=["auto", "direct", "fft"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ncc", action="store_true")
    ap.add_argument("--no-stages", action="store_true", help="skip the edge-taper / PCIe timings around the loop (N = 1)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the product path); gloo only to rehearse N > 1 on a one-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: all ranks on cuda:0 (with --backend gloo)")
    ap.add_argument("--zchunks", type=int, default=1,
                    help="halo rows travel in this many z chunks (slab.SlabRL zchunks: sent per chunk of planes as their x tiles finish, "
                         "the next y-forward pass starts per chunk as its rows land)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "peer"],
                    help="halo exchange at N > 1: grouped send/recv of the process group (RCCL), or hipMemcpyPeerAsync into the "
                         "neighbour's buffers (copy engines; slab.PeerLink)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: fan the ranks out as a child process tree before anything here touches the GPU
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    ncc_workers = None
    if args.gpus == 1 and not args.no_ncc and not args.no_cpu_baseline:
        import bench_ncc
        ncc_workers = bench_ncc.start_cpu_workers(host_cores())   # idle until the NCC CPU leg; started before the GPU is touched

    import torch
    from ipp_amd import capi, decon
    capi.require_gpu()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    dev = torch.device("cuda", 0 if args.share_gpu else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:   # a rank that never arrives must end the run with a message, not hang it
            tmo = datetime.timedelta(seconds=int(os.environ.get("MI_BENCH_RENDEZVOUS_S", "180")))
            if args.backend == "nccl":
                dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
            else:
                dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=tmo)
        except Exception as e:
            raise SystemExit(f"bench.py: rank {rank} of {world}: rendezvous over {args.backend} failed: {e!r}")

    vshape, kshape = WORKLOADS[args.workload]
    psf_np = make_psf(kshape, gaussian=args.workload == "c1")
    engine = {"auto": capi.ENGINE_AUTO, "direct": capi.ENGINE_DIRECT, "fft": capi.ENGINE_FFT}[args.engine]
    n_vox_global = vshape[0] * vshape[1] * vshape[2]

    if world == 1:
        if os.environ.get("MI_BENCH_VOLUME_FIRST"):
            bl = make_volume(vshape, dev)
            ctx = decon.RLContext(vshape, psf_np, None, boundary=capi.BOUNDARY_CIRCULAR, engine=engine, device=dev)
        else:
            # the context first: its spectrum arrays (21.6 GB in one allocation) are mapped while device memory is still one piece
            ctx = decon.RLContext(vshape, psf_np, None, boundary=capi.BOUNDARY_CIRCULAR, engine=engine, device=dev)
            bl = make_volume(vshape, dev)
        ratio = None if ctx.fuses else torch.empty_like(bl)   # scratch only for engines that cannot fuse an iteration

        def step():
            ctx.iterate(bl, ratio, 1)
        run_steps = lambda k: ctx.iterate(bl, ratio, k)  # noqa: E731  (consecutive iterations fuse inside the library)
        parallelism = "single"
        engine_used = ctx.engine
    else:
        from ipp_amd import slab
        try:
            drv = slab.SlabRL(vshape, psf_np, rank=rank, world_size=world, device=dev, flavour="fft", engine=engine,
                              seed=1234, transport=args.transport, zchunks=args.zchunks)
        except (TimeoutError, RuntimeError) as e:   # e.g. the peer transport's handle exchange timing out
            raise SystemExit(f"bench.py: rank {rank} of {world}: setting up the {args.transport} halo exchange failed: {e!r}")
        step = drv.iterate
        run_steps = None
        pg_world = dist.get_world_size()
        if pg_world != world:
            raise SystemExit(f"process group has {pg_world} ranks, expected {world}")
        parallelism = (f"y-slabs x{pg_world}, process group backend {dist.get_backend()}, halo exchange: " +
                       ((("grouped send/recv (RCCL over xGMI)" if args.backend == "nccl" else
                          "grouped send/recv through gloo (host staging: a REHEARSAL of the N > 1 path, not an RCCL result)")
                         if args.transport == "rccl" else "hipMemcpyPeerAsync into the neighbour's buffers")) +
                       (f", {len(drv.zb)} z chunks" if drv.zb is not None else ""))
        engine_used = drv.ctx.engine
        ctx, bl = drv.ctx, drv.bl                # rank-local context / slab (interior + halo rows) for the per-pass timing

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    # HIP events on the stream the kernels are launched on (torch's current stream of this device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    if run_steps is not None:
        run_steps(args.steps)
    else:
        for _ in range(args.steps):
            step()
    ev1.record()
    sync()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = n_vox_global * args.steps / elapsed / 1e9
        local_vox = n_vox_global / world
        # voxels one launch of a pass processes: the rank-local grid (interior + halo rows, rounded to a native FFT extent)
        launch_vox = float(bl.numel())
        iteration_gbs = ALGO_BYTES_PER_VOXEL_ITER * local_vox / (dev_ms / args.steps * 1e-3) / 1e9
        roofline = None
        if engine_used == 2:
            try:
                # per-pass launch durations, HIP events on the launch stream (mi_rl_time_pass); the dominant kernel
                # is the one with the largest share of an iteration
                # the fused x pass runs once as the ratio step (12 B/voxel) and once as the update step (16 B/voxel, it also
                # writes bl) per iteration: one kernel, quoted as the mean of the two launches (14 B/voxel)
                per_iter = {"x_fused": 2, "y_forward": 2, "z_conv": 2, "y_inverse": 2}
                times = {k: ctx.time_pass(k, bl, reps=5) for k in ("y_forward", "z_conv", "y_inverse")}
                t_ratio = ctx.time_pass("x_fused", bl, reps=5)
                t_update = ctx.time_pass("x_fused_update", bl, reps=5)  # overwrites bl: the timed region is over
                times["x_fused"] = 0.5 * (t_ratio + t_update)
                dom = max(times, key=lambda k: times[k] * per_iter[k])
                algo_b = {"z_conv": 10 if ctx.otf_is_real else 12, "y_forward": 8, "y_inverse": 8, "x_fused": 14}[dom]  # B/voxel/launch
                ach = algo_b * launch_vox / (times[dom] * 1e-3) / 1e9
                traffic, traffic_src = pmc_traffic(dom, args.workload) if world == 1 else (None, None)
                roofline = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                            "traffic_source": (traffic_src + " (PMC passes of this command kept in the repository; not collected in this run)")
                            if traffic_src else None,
                            "kernel": {"z_conv": ("k_z_pair_pipe" if ctx.pair_layout else "k_z_conv_pipe") +
                                                 " (z-forward FFT + untangle*OTF + z-inverse FFT, one pass)",
                                       "x_fused": "k_x_fused_pipe (x-inverse FFT + RL epilogue + x-forward FFT; mean of the "
                                                  "ratio and the update launch)",
                                       "y_forward": "k_y_pair<fwd>" if ctx.pair_layout else "k_y_pass<fwd>",
                                       "y_inverse": "k_y_pair<inv>" if ctx.pair_layout else "k_y_pass<inv>"}[dom],
                            "algorithmic_bytes_per_voxel_per_launch": algo_b, "voxels_per_launch": int(launch_vox),
                            "launch_ms": round(times[dom], 4), "launches_per_iteration": per_iter[dom],
                            "pass_ms": dict({k: round(v, 4) for k, v in times.items()},
                                            x_fused_ratio=round(t_ratio, 4), x_fused_update=round(t_update, 4))}
                if args.workload == "c3" and world == 1:
                    # the passes run in one of two speeds per allocation (physical page placement): which one this run got
                    roofline["pass_mode"] = {"mode": "fast" if t_update < 5.95 else "slow", "x_fused_update_ms": round(t_update, 4),
                                             "note": "update launch of the x pass: ~5.6 ms (fast) or ~6.3 ms (slow) per allocation"}
                    try:   # how the library placed the spectrum arrays (candidates' cost 4 y + update in ms, the kept one)
                        cand, kept = ctx.fft_placement()
                        roofline["placement"] = {"candidates_ms": cand, "kept": kept,
                                                 "note": "plan-time trial: buffers allocated side by side, cost 4 y + 3 update of every ordered pair (S, T), the cheapest kept (DESIGN.md 5)"}
                    except Exception:
                        pass
            except Exception as e:  # e.g. rocFFT fallback: no per-pass hook
                roofline = None
                sys.stderr.write(f"per-pass timing unavailable: {e!r}\n")
        if roofline is None:
            roofline = {"bound": "hbm", "achieved": round(iteration_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(iteration_gbs / HBM_PEAK_GBS, 5), "traffic": None,
                        "kernel": "rl_iteration (whole pipeline; no per-kernel hook on this engine)"}
        roofline["iteration"] = {"algorithmic_bytes_per_voxel_iter": ALGO_BYTES_PER_VOXEL_ITER,
                                 "achieved_GBps": round(iteration_gbs, 2), "frac": round(iteration_gbs / HBM_PEAK_GBS, 5),
                                 "device_ms_per_iteration": round(dev_ms / args.steps, 4)}
        out = {
            "metric": "RL-deconv Gvoxels/sec", "value": round(value, 4), "unit": "Gvoxel*iter/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "iterations_per_s": round(args.steps / elapsed, 4),
            "config": {"workload": f"{args.workload}: {vshape[2]}x{vshape[1]}x{vshape[0]} fp32 volume, "
                                   f"{kshape[2]}x{kshape[1]}x{kshape[0]} PSF, deconFFT semantics, lambda=0, reg_interval=0",
                       "engine": {1: "direct", 2: "fft"}.get(engine_used, str(engine_used)), "parallelism": parallelism},
            "roofline": roofline,
        }
        if world == 1 and not args.no_stages:
            ctx = bl = ratio = None                  # (the closures above see the rebinding: the context and its volume are freed)
            torch.cuda.empty_cache()
            capi.release_cached_memory()
            out["block_stages"] = block_stages(vshape, psf_np, dev)
            capi.release_cached_memory()
            out.update(secondary_rows(vshape, kshape, psf_np, dev, args.workload == "c1", args.steps))
            capi.release_cached_memory()
            if args.workload == "c3":
                try:
                    out["c4_single"] = c4_single(dev)
                except Exception as e:
                    out["c4_single"] = {"error": repr(e)}
                    torch.cuda.empty_cache()
                    capi.release_cached_memory()
                try:
                    out["decwrap"] = decwrap_e2e()
                except Exception as e:
                    out["decwrap"] = {"error": repr(e)}
                try:
                    out["decwrap_tiff"] = decwrap_e2e(shape=(512, 2048, 2048), tiff=True)
                except Exception as e:
                    out["decwrap_tiff"] = {"error": repr(e)}
                torch.cuda.empty_cache()
                capi.release_cached_memory()
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(vshape, kshape, gaussian=args.workload == "c1")
    ncc = None
    if world > 1:
        drv.close()                                  # (the copy-engine link: mapped peer memory, interprocess events)
    if not args.no_ncc:
        # second metric (BASELINE config 5).  N > 1: tile-row blocks, one block per rank (bench_ncc.py); every rank takes part
        if world > 1:
            step = drv = ctx = bl = None             # the slab context and its volume make room for the tile rows
            torch.cuda.empty_cache()
            capi.release_cached_memory()
        try:
            import bench_ncc
            ncc = bench_ncc.run(dev, cpu_workers=ncc_workers, rank=rank, world=world, dist=dist,
                                dist_device=(dev if args.backend == "nccl" else "cpu"))
        except Exception as e:  # the NCC leg must not hide the headline number
            ncc = {"error": repr(e)}
            if world > 1:
                raise
        finally:
            if ncc_workers:
                bench_ncc.stop_cpu_workers(ncc_workers)
    if rank == 0:
        if ncc is not None:
            out["ncc"] = ncc
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
