"""Direct engine on a rank-1 (Gaussian) PSF: single-pass separable kernel (sep3d.hip) vs three 1-D launches vs the dense tap loop vs the FFT engine.
    python profiles/separable_time.py [kz ky kx]   (default: BASELINE config 1 PSF 9 x 9 x 15; on a C2-sized volume, zero and circular boundary)
MI_IPP_PROBES=1 adds the LDS-ring version of the single pass (MI_SEP_RING=1) to the line."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ipp_amd import capi, decon  # noqa: E402

dev = torch.device("cuda", 0)
shape = (256, 1024, 1024)
taps = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (15, 9, 9)
ax = [np.exp(-0.5 * ((np.arange(n) - (n - 1) / 2) / (n / 6.0)) ** 2) for n in taps]
psf = (ax[0][:, None, None] * ax[1][None, :, None] * ax[2][None, None, :])
psf = (psf / psf.sum()).astype(np.float32)
bl0 = torch.rand(shape, device=dev) + 0.1


def run(engine, boundary, env=None, iters=4):
    for k, v in (env or {}).items():
        os.environ[k] = v
    try:   # (some switches are read when the context is built, others at every launch)
        ctx = decon.RLContext(shape, psf, None, boundary=boundary, engine=engine, device=dev)
        bl = bl0.clone()
        ratio = torch.empty_like(bl)
        ctx.iterate(bl, ratio, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.iterate(bl, ratio, iters)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / iters * 1e3
    finally:
        for k in (env or {}):
            del os.environ[k]
    return ms, ctx.separable, bl


for bname, b in (("zero boundary (deconSpatial)", capi.BOUNDARY_ZERO), ("circular (deconFFT)", capi.BOUNDARY_CIRCULAR)):
    ms_sep, sep, a = run(capi.ENGINE_DIRECT, b)
    ms_three, _, _ = run(capi.ENGINE_DIRECT, b, {"MI_NO_SEP_SINGLE": "1"})
    ring = ""
    if os.environ.get("MI_IPP_PROBES") == "1":
        ms_ring, _, r_ = run(capi.ENGINE_DIRECT, b, {"MI_SEP_RING": "1"})
        ring = f" (LDS ring {ms_ring:.2f}, max |difference| {float((a - r_).abs().max()):.1e})"
    ms_dense, sep2, c = run(capi.ENGINE_DIRECT, b, {"MI_NO_SEPARABLE": "1"}, iters=1)
    ms_fft, _, d = run(capi.ENGINE_FFT, b)
    a1 = run(capi.ENGINE_DIRECT, b, iters=1)[2]
    err = float((a1 - c).abs().max() / c.abs().max())   # both after 2 iterations (1 warm-up + 1)
    nvox = float(np.prod(shape))
    print(f"{bname}, taps {taps}: direct separable, single pass {ms_sep:.2f} ms/iteration{ring} = {2 * 12 * nvox / ms_sep / 1e9:.2f} TB/s on 12 B/voxel "
          f"(separable={sep}), three launches {ms_three:.2f}, "
          f"direct dense {ms_dense:.2f} (separable={sep2}), "
          f"FFT engine {ms_fft:.2f}; separable vs dense max rel diff {err:.2e}", flush=True)
