"""Do the SOURCE's and the DESTINATION's regions of device memory decide the speed of a strided pass independently?  C3: K buffers of
the size of one spectrum array (10.8 GB) allocated side by side; the forward y pass timed from buffer i to buffer j for every
ordered pair, and the update launch of the x pass likewise (mi_rl_time_between).
    python profiles/spectrum_halves_probe.py [K]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ipp_amd import capi, decon

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
os.environ["MI_FFT_PLACE_CANDIDATES"] = "1"
shape, kshape = (512, 2048, 2048), (61, 31, 31)
dev = torch.device("cuda", 0)
z, y, x = np.meshgrid(*[np.arange(k) - (k - 1) / 2 for k in kshape], indexing="ij")
psf = np.exp(-(z / 8.0) ** 2 - (y / 3.0) ** 2 - (x / 3.0) ** 2).astype(np.float32)
psf /= psf.sum()
ctx = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
L = capi.lib()
nbytes = int(L.mi_rl_fft_spectrum_bytes(ctx._h))
bufs = [torch.zeros(nbytes + 8192, dtype=torch.uint8, device=dev) for _ in range(K)]
bl = torch.rand(shape, device=dev) + 0.1
ms = C.c_float()


def table(which, what):
    t = np.zeros((K, K))
    for i in range(K):
        for j in range(K):
            if i == j:
                continue
            capi.check(L.mi_rl_time_between(ctx._h, capi.current_stream_ptr(dev), which, C.c_void_p(bufs[i].data_ptr()), C.c_void_p(bufs[j].data_ptr()),
                                            C.c_void_p(bl.data_ptr()), 3, C.byref(ms)))
            t[i, j] = ms.value
    print(f"{what}, C3, reading buffer i (rows), writing buffer j (columns), ms; buffers of {nbytes / 1e9:.1f} GB allocated one after the other:")
    for i in range(K):
        print("  " + "  ".join("  -  " if i == j else f"{t[i, j]:5.3f}" for j in range(K)))
    rows = np.array([t[i, [j for j in range(K) if j != i]].mean() for i in range(K)])
    cols = np.array([t[[i for i in range(K) if i != j], j].mean() for j in range(K)])
    print("mean by the buffer read:    " + "  ".join(f"{v:5.3f}" for v in rows))
    print("mean by the buffer written: " + "  ".join(f"{v:5.3f}" for v in cols), flush=True)


table(0, "forward y pass")
table(1, "update launch of the x pass")
table(3, "z pass")
table(4, "z pass on the context's own arrays, buffer i in the place of the OTF (j unused)")
