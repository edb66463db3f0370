"""Cache-blocked middle of the FFT convolution: y forward -> z * OTF -> y inverse on a chunk of plane pairs
small enough for source + destination to stay in the 256-MiB Infinity Cache, chunk after chunk (MI_FFT_CHUNK=<plane pairs>[,<chunks
in flight>]), against the three full-volume passes.  C3 by default (one plane pair = 16.8 MB + padding).

    python profiles/chunk_probe.py [workload]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ipp_amd import capi, decon  # noqa: E402

dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
vshape, kshape = bench.WORKLOADS[wl]
psf = bench.make_psf(kshape)
vol = bench.make_volume(vshape, dev)
ref = None
print(f"{wl}: {vshape[2]} x {vshape[1]} x {vshape[0]}; ms per iteration (10 fused iterations), middle = y-fwd + z + y-inv of one convolution")
settings = [None] + [(c, s) for c in (2, 4, 8, 16, 32) for s in (1, 2, 3)]
if len(sys.argv) > 2:
    settings = [None] + [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]]
for st in settings:
    if st is None:
        os.environ.pop("MI_FFT_CHUNK", None)
    else:
        os.environ["MI_FFT_CHUNK"] = f"{st[0]},{st[1]}"
    ctx = decon.RLContext(vshape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
    bl = vol.clone()
    ctx.iterate(bl, None, 2)
    torch.cuda.synchronize()
    if ref is None:
        ref = bl.clone()
        same = "reference"
    else:
        same = "bit-identical" if torch.equal(bl, ref) else f"max |diff| {float((bl - ref).abs().max()):.3e}"
    t0 = time.perf_counter()
    ctx.iterate(bl, None, 10)
    torch.cuda.synchronize()
    it_ms = (time.perf_counter() - t0) / 10 * 1e3
    if st is None:
        mid = sum(ctx.time_pass(k, bl, reps=5) for k in ("y_forward", "z_conv", "y_inverse"))
    else:
        mid = ctx.time_pass("y_forward", bl, reps=5)
    xr, xu = ctx.time_pass("x_fused", bl, reps=5), ctx.time_pass("x_fused_update", bl, reps=5)
    name = "3 full passes" if st is None else f"chunk {st[0]:>2} plane pairs x {st[1]} in flight"
    print(f"{name:>34}: iteration {it_ms:7.2f} ms   middle {mid:6.2f} ms   x ratio/update {xr:5.2f} / {xu:5.2f} ms   result {same}", flush=True)
    del ctx, bl
    torch.cuda.empty_cache()
    capi.release_cached_memory()
