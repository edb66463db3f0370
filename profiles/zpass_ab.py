"""A / B of one pass inside ONE process (two contexts on the same shape cannot share a placement, so the switch is an environment
variable of the probes build read at launch time): time_pass of `which` with the variable unset / set, alternating.
    MI_IPP_PROBES=1 python profiles/zpass_ab.py <VARIABLE> z_conv 1024 576 4096 [kz ky kx]
Round 4 used it for two variants of the 1024-point z pass that were not kept (next tile requested a whole tile ahead: 6.28 against 5.90 ms;
512 threads with a line pair per wave: 6.00 against 5.88 ms on 1024 x 576 x 4096; both bit-identical to the kept kernel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ipp_amd import capi, decon

var, which = sys.argv[1], sys.argv[2]
shape = tuple(int(v) for v in sys.argv[3:6])
kshape = tuple(int(v) for v in sys.argv[6:9]) if len(sys.argv) >= 9 else (31, 15, 15)
dev = torch.device("cuda", 0)
z, y, x = np.meshgrid(*[np.arange(k) - (k - 1) / 2 for k in kshape], indexing="ij")
psf = np.exp(-(z / 8.0) ** 2 - (y / 3.0) ** 2 - (x / 3.0) ** 2).astype(np.float32)
psf /= psf.sum()
ctx = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
bl = torch.rand(shape, device=dev) + 0.1
ctx.iterate(bl, None, 2)
torch.cuda.synchronize()
for rep in range(3):
    os.environ.pop(var, None)
    a = ctx.time_pass(which, bl, reps=5)
    os.environ[var] = "1"
    b = ctx.time_pass(which, bl, reps=5)
    print(f"{which} on {shape}: default {a:.3f} ms, {var}=1 {b:.3f} ms", flush=True)
# the two variants must agree: two iterations each from the same start
os.environ.pop(var, None)
b0 = bl.clone()
ctx.iterate(b0, None, 2)
os.environ[var] = "1"
b1 = bl.clone()
ctx.iterate(b1, None, 2)
torch.cuda.synchronize()
print(f"max |difference| after two iterations: {float((b0 - b1).abs().max()):.3e} (values up to {float(b0.max()):.3f})", flush=True)
