"""cProfile of one decwrap.py run (268 Mvoxel uint16 volume in 16 blocks): where the host time of a run goes."""
import cProfile, os, pstats, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipp_amd import decwrap

rng = np.random.default_rng(0)
vol = (rng.random((256, 1024, 1024), dtype=np.float32) * 3000 + 200).astype(np.uint16)
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "vol.npy")
    np.save(path, vol)
    base = ["-i", path, "-dxy", "0.422", "-dz", "1.0", "-ex", "488", "-em", "525", "-it", "6", "--use-fft", "--gpu-indices", "1",
            "--block-size-max", str(80_000_000), "--no-resume", "--gpu-workers-per-gpu", "1"]
    decwrap.main(base)   # warm-up
    pr = cProfile.Profile()
    pr.enable()
    decwrap.main(base)
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(22)
