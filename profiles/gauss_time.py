"""Times gauss3d_gpu on a C3-sized volume (the regularisation step sigma=0.5 and decwrap's default pre-filter).
MI_IPP_PROBES=1: also through the separable single-pass kernel (MI_GAUSS_VIA_SEP=1, csrc/sep3d.hip) with the results compared."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipp_amd import decon
dev = torch.device("cuda", 0)
x0 = torch.rand((512, 2048, 2048), device=dev)


import ctypes as C
from ipp_amd import capi
x, work, ref, alt = (torch.empty_like(x0) for _ in range(4))   # (every buffer up front: a volume is 8.6 GB)


def once(sig3, ks3):
    ks = None if ks3 is None else (C.c_int * 3)(*ks3)
    capi.check(capi.lib().mi_gauss3d_inplace(0, capi.current_stream_ptr(dev), x.data_ptr(), work.data_ptr(), 2048, 2048, 512, (C.c_float * 3)(*sig3), ks))


def run(sig, ks, keep):
    sig3 = [sig] * 3 if not isinstance(sig, list) else sig
    x.copy_(x0)
    once(sig3, ks)
    keep.copy_(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        once(sig3, ks)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 3 * 1e3


for name, sig, ks in [("reg step sigma=0.5 (5 taps)", 0.5, None), ("pre-filter sigma=(0.5,0.5,2.5) size=(13,13,25)", [0.5, 0.5, 2.5], [13, 13, 25])]:
    ms = run(sig, ks, ref)
    line = f"{name}: {ms:.2f} ms  ({x0.numel() * 8 / ms / 1e6:.0f} GB/s on the 8 B/voxel ideal)"
    if os.environ.get("MI_IPP_PROBES") == "1":
        for var, val, what in (("MI_GAUSS_WX", "1", "tiles of 64 columns"), ("MI_GAUSS_WX", "2", "tiles of 128 columns"),
                               ("MI_GAUSS_VIA_SEP", "1", "as a separable convolution (k_sep3d_acc)")):
            os.environ[var] = val
            ms2 = run(sig, ks, alt)
            del os.environ[var]
            line += f"; {what} {ms2:.2f} ms (max |difference| {float((ref - alt).abs().max()):.1e})"
    print(line, flush=True)
