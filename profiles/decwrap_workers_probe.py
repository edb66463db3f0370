"""Wall time of decwrap.py on a multi-block volume with one and with two workers per GPU (the second worker overlaps one
block's PCIe / host staging with another block's kernels)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipp_amd import decwrap

rng = np.random.default_rng(0)
vol = (rng.random((256, 1024, 1024), dtype=np.float32) * 3000 + 200).astype(np.uint16)
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "vol.npy")
    np.save(path, vol)
    base = ["-i", path, "-dxy", "0.422", "-dz", "1.0", "-ex", "488", "-em", "525", "-it", "6", "--use-fft", "--gpu-indices", "1",
            "--block-size-max", str(80_000_000), "--no-resume"]
    for workers in (1, 2, 1, 2):
        t0 = time.perf_counter()
        rc = decwrap.main(base + ["--gpu-workers-per-gpu", str(workers)])
        print(f"workers per GPU {workers}: rc {rc}, {time.perf_counter() - t0:.2f} s for {vol.size / 1e6:.0f} Mvoxel, 6 iterations", flush=True)
