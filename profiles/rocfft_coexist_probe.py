"""Are live rocFFT plans independent of each other?  Two RL contexts on the rocFFT route (MI_FFT_ROCFFT=1), A created first and kept
alive, B created and used beside it; B's circular convolution against the same context created alone, and A's (used after B exists)
likewise.  ROCm 7.2 on gfx950: one of the pairs below is not (profiles/r05_rocfft_coexistence.txt); MI_FFT_NO_VERIFY=1 switches off
the check every rocFFT engine makes of itself at creation and shows the wrong values instead of the refusal.
    python profiles/rocfft_coexist_probe.py"""
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MI_FFT_ROCFFT"] = "1"
import numpy as np, torch
from ipp_amd import capi, decon
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
ker = rng.random((3, 5, 7), dtype=np.float32)
def conv_with(ctx, img):
    a = torch.from_numpy(img).to(dev); r = torch.empty_like(a)
    ctx.forward_ratio(a, r)
    return (a / r).cpu().numpy()
def make(shape):
    return decon.RLContext(shape, ker, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
def alone(shape, img):
    c = make(shape); out = conv_with(c, img); del c; gc.collect(); return out
pairs = [((8, 8, 16), (8, 128, 32)), ((16, 32, 64), (8, 128, 32)), ((32, 64, 128), (8, 128, 32)), ((64, 16, 256), (8, 128, 32)),
         ((8, 128, 32), (32, 64, 128)), ((32, 64, 128), (64, 32, 128)), ((280, 135, 135), (280, 512, 135)), ((960, 512, 135), (960, 135, 512)),
         ((48, 40, 36), (40, 36, 48)), ((20, 24, 25), (25, 20, 24)), ((30, 35, 27), (27, 30, 35))]
for A, B in pairs:
    img = rng.random(B, dtype=np.float32) + 0.5
    ref = alone(B, img)
    a = make(A)
    try:
        b = make(B)
    except capi.MiError as e:
        print(f"A {A} alive, B {B}: B refused at creation: {str(e)[:150]} ...", flush=True)
        del a; gc.collect()
        continue
    got = conv_with(b, img)
    # and A used after B exists
    imgA = rng.random(A, dtype=np.float32) + 0.5
    gotA = conv_with(a, imgA)
    del a, b; gc.collect()
    refA = alone(A, imgA)
    print(f"A {A} alive, B {B}: B max rel diff {float(np.abs(got - ref).max() / np.abs(ref).max()):.3g}; A (used after B was made) {float(np.abs(gotA - refA).max() / np.abs(refA).max()):.3g}", flush=True)
