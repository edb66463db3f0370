"""Stress of the pair-interleaved layout: random native shapes and PSFs, forward and adjoint convolution on the paired layout against
the plain layout of the same library (MI_FFT_NO_PAIR=1).  usage: python profiles/pair_stress.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ipp_amd import capi, decon

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)
ZS = [64, 128, 256, 512, 1024, 192, 384, 768, 576, 1152]
YS = [16, 32, 48 * 2, 64, 128, 160, 192, 256, 288, 320, 384, 512, 576, 640, 768, 1024, 1152, 1280, 2048]
XS = [16, 32, 64, 96 * 2, 128, 36 * 2 * 4]
worst = 0.0
done = 0
while done < n_cases:
    shape = (int(rng.choice(ZS)), int(rng.choice(YS)), int(rng.choice(XS)))
    if np.prod(shape) > 3e8:
        continue
    k = tuple(int(v) for v in (rng.integers(1, 8) * 2 + 1, rng.integers(1, 5) * 2 + 1, rng.integers(1, 5) * 2 + 1))
    if any(kk > s for kk, s in zip(k, shape)):
        continue
    sym = bool(rng.integers(0, 2))
    psf = rng.random(k).astype(np.float32) + 0.05
    if sym:
        psf = (psf + psf[::-1, ::-1, ::-1]) / 2
    psf /= psf.sum()
    os.environ.pop("MI_FFT_NO_PAIR", None)
    a = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
    os.environ["MI_FFT_NO_PAIR"] = "1"
    b = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
    os.environ.pop("MI_FFT_NO_PAIR", None)
    if not a.pair_layout:
        continue
    x = torch.rand(shape, device=dev) + 0.5
    errs = []
    for ctx_pair in (0,):
        ra, rb = torch.empty_like(x), torch.empty_like(x)
        a.forward_ratio(x, ra)
        b.forward_ratio(x, rb)
        errs.append(float((ra - rb).abs().max() / rb.abs().max()))
        ua, ub = torch.ones_like(x), torch.ones_like(x)
        a.adjoint_update(x, ua)
        b.adjoint_update(x, ub)
        errs.append(float((ua - ub).abs().max() / ub.abs().max()))
        ia, ib = x.clone(), x.clone()
        a.iterate(ia, None, 2)
        b.iterate(ib, None, 2)
        errs.append(float((ia - ib).abs().max() / ib.abs().max()))
    worst = max(worst, max(errs))
    done += 1
    flag = "" if max(errs) < 2e-5 else "   <-- LARGE"
    print(f"{done:3d} shape {shape} psf {k} {'sym' if sym else 'asym'} real_otf={a.otf_is_real}: {max(errs):.2e}{flag}", flush=True)
    del a, b
print(f"worst relative difference over {done} cases: {worst:.2e}")
