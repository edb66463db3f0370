"""Where the time of one block goes: process_block (LsDeconv.m:906-948) on a C3-sized block with decwrap.py's defaults
(6 iterations, regularisation every 3rd, pre-filter sigma (0.5,0.5,2.5) size (13,13,25)), spatial and FFT flavours.
    python profiles/process_block_probe.py [c2|c3]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from ipp_amd import decon as D, lsdeconv as L

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
shape, kshape = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
psf = D.make_psf_struct(bench.make_psf(kshape))
filt = L.Filter()


def stamp(label, t0):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"   {label:34s} {(t1 - t0) * 1e3:8.1f} ms", flush=True)
    return t1


for use_fft in (False, True):
    for rep in range(2):
        filt.use_fft = use_fft
        blk = L.Block(shape[2], shape[1], shape[0], 1, 1, 1, 0, 0, 0, fft_shape=shape[::-1] if use_fft else None)
        host = np.random.default_rng(0).random(shape, dtype=np.float32)
        print(f"{wl} use_fft={use_fft} run {rep}", flush=True)
        t0 = time.perf_counter()
        t = torch.from_numpy(host).to(dev)
        t0 = stamp("H2D (pageable host memory)", t0)
        D.gauss3d_gpu(t, list(filt.gaussian_sigma), list(filt.gaussian_size))
        t0 = stamp("pre-filter gauss3d", t0)
        m = float(t.max())
        t0 = stamp("max(bl) guard", t0)
        D.decon(t, psf, 6, 0.0, 0.0, filt.regularize_interval, 1, use_fft, blk.fft_shape, False)
        t0 = stamp("decon (edge taper + 6 iterations)", t0)
        lb, ub = L.deconvolved_stats(t, 99.99)
        t0 = stamp("deconvolved_stats (prctile)", t0)
        q = D.rescale_block(t, 65535.0, 1.0, lb, ub)
        t0 = stamp("rescale to uint16", t0)
        out = q.cpu()
        t0 = stamp("D2H of the uint16 block", t0)
        del t, q
