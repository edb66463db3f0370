"""decwrap on a volume that is larger than what its host-side buffers may hold: wall time, blocks, peak resident set.
    python profiles/decwrap_scale_probe.py [nz ny nx] [block_size_max | auto]     (auto: decwrap's own choice)
Default: a 512 x 2048 x 2048 uint16 volume (4.3 GB) written as a memory-mapped *.npy under /tmp, deconFFT flavour, 6 iterations,
blocks of at most 300 M elements (incl. pads), no whole-volume output copies (MI_DECWRAP_NPY=0): what is measured is the streaming
pipeline -- box reads, device work, LZ4 bricks, slab-wise assembly + rescale."""
import os
import resource
import shutil
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (512, 2048, 2048)
bmax = sys.argv[4] if len(sys.argv) >= 5 else "300000000"
nworkers = os.environ.get("PROBE_WORKERS", "2")
root = "/tmp/decwrap_scale"
shutil.rmtree(root, ignore_errors=True)
os.makedirs(root)
t0 = time.perf_counter()
vol = np.lib.format.open_memmap(os.path.join(root, "vol.npy"), mode="w+", dtype=np.uint16, shape=shape)
rng = np.random.default_rng(1)
for z in range(shape[0]):                      # sparse beads on a noisy background, slice by slice
    sl = rng.integers(600, 700, size=shape[1:], dtype=np.uint16)
    idx = rng.integers(0, sl.size, size=sl.size // 2000)
    sl.reshape(-1)[idx] = rng.integers(5000, 60000, size=idx.size, dtype=np.uint16)
    vol[z] = sl
vol.flush()
del vol
t_gen = time.perf_counter() - t0
rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
# the memory-mapped input counts towards the resident set page by page (reclaimable page cache): what the pipeline itself holds is
# the ANONYMOUS resident memory (staging buffers, cores waiting for a writer, the integer slab), sampled during the run
import threading
peak_anon = [0.0]
stop = threading.Event()


def sample():
    while not stop.is_set():
        with open("/proc/self/status") as f:
            for line in f:
                if line.startswith("RssAnon:"):
                    peak_anon[0] = max(peak_anon[0], int(line.split()[1]) / 1e6)
        stop.wait(0.1)


threading.Thread(target=sample, daemon=True).start()

# PROBE_SAMPLE=1: where the host threads are -- every 5 ms the innermost frames of all threads (a poor man's sampling profiler)
stacks = {}


def sample_stacks():
    import collections
    me = threading.get_ident()
    while not stop.is_set():
        for tid, fr in sys._current_frames().items():
            if tid == me:
                continue
            chain = []
            f = fr
            while f is not None:
                chain.append(f"{os.path.basename(f.f_code.co_filename)}:{f.f_lineno}:{f.f_code.co_name}")
                f = f.f_back
            if not any(":run_block" in c for c in chain):
                continue                                   # only the block workers
            own = [c for c in chain if c.startswith(("decwrap.py", "lsdeconv.py", "decon.py", "capi.py", "brickio.py"))]
            key = " < ".join(own[:3])
            stacks[key] = stacks.get(key, 0) + 1
        stop.wait(0.005)


if os.environ.get("PROBE_SAMPLE"):
    threading.Thread(target=sample_stacks, daemon=True).start()
os.environ["MI_DECWRAP_NPY"] = "0"
from ipp_amd import decwrap  # noqa: E402

# PROBE_SPANS=1: host wall time per stage of a block, summed over the workers (Python calls here; with MI_IPP_PROBES=1 the library's
# own spans -- MI_SPAN_BEGIN/END, probe build only -- are printed after the run)
py_spans = {}
if os.environ.get("PROBE_SPANS"):
    from ipp_amd import decon as _D, lsdeconv as _L
    span_lock = threading.Lock()

    def timed(mod, name, label=None):
        fn = getattr(mod, name)

        def wrapper(*a, **k):
            t = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                dt_ = time.perf_counter() - t
                with span_lock:
                    e = py_spans.setdefault(label or name, [0, 0.0])
                    e[0] += 1
                    e[1] += dt_
        setattr(mod, name, wrapper)

    timed(_L, "load_block_device")
    timed(_L, "process_block")
    timed(_L, "deconvolved_stats")
    timed(_D, "gauss3d_gpu")
    timed(_D, "decon")
    timed(_D, "rescale_block")
t0 = time.perf_counter()
rc = decwrap.main(["-i", os.path.join(root, "vol.npy"), "-dxy", "0.422", "-dz", "1.0", "-ex", "488", "-em", "525", "--use-fft", "-it", "6"]
                  + ([] if bmax == "auto" else ["--block-size-max", bmax]) + ["--gpu-indices", "1", "--gpu-workers-per-gpu", nworkers])
dt = time.perf_counter() - t0
stop.set()
rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
tm = getattr(decwrap.main, "last_timing", {})
if tm:
    print(f"  blocks phase {tm['blocks_wall_s']:.1f} s (device work {tm['device_ms'] / 1e3:.1f} s = {tm['device_ms'] / 10 / max(tm['blocks_wall_s'], 1e-9):.0f} % "
          f"device-busy; box reads {tm['box_read_s']:.1f} s, D2H {tm['d2h_s']:.1f} s, waiting for a core buffer {tm['wait_buffer_s']:.1f} s, summed over "
          f"the workers), assembly phase {tm.get('assembly_wall_s', 0.0):.1f} s; host time between the device events {tm.get('host_in_device_s', 0.0):.1f} s; "
          f"cache folder handed to its remover in {tm.get('cleanup_s', 0.0):.2f} s, main() {tm.get('main_s', 0.0):.2f} s of the {dt:.2f} s the call took "
          f"(the rest: its buffers being freed on return)", flush=True)
nvox = float(np.prod(shape))
print(f"{nworkers} workers per GPU; volume {shape[2]} x {shape[1]} x {shape[0]} uint16 = {nvox * 2 / 1e9:.1f} GB (generated in {t_gen:.0f} s), block-size-max {bmax}: rc {rc}, "
      f"{dt:.1f} s wall = {nvox / dt / 1e6:.0f} Mvoxel/s end to end (6 RL iterations, default filters), peak resident set {rss:.1f} GB "
      f"incl. the mapped input file (before the run {rss0:.1f} GB), peak ANONYMOUS resident memory {peak_anon[0]:.1f} GB "
      f"(a float32 copy of the volume would be {nvox * 4 / 1e9:.1f} GB)", flush=True)
if py_spans:
    print("  host wall time per stage, summed over the workers (calls, seconds):")
    for k, (c, v) in sorted(py_spans.items(), key=lambda kv: -kv[1][1]):
        print(f"    {k:40s} {c:7d} {v:10.3f}")
    import ctypes
    from ipp_amd import capi
    lib = capi.lib()
    if hasattr(lib, "mi_probe_host_spans"):
        buf = ctypes.create_string_buffer(1 << 16)
        n = lib.mi_probe_host_spans(buf, len(buf))
        print("  spans inside the library (label, calls, seconds):")
        print("    " + buf.raw[:n].decode().replace("\n", "\n    "))
if stacks:
    tot = sum(stacks.values())
    for k, v in sorted(stacks.items(), key=lambda kv: -kv[1])[:40]:
        print(f"  {100.0 * v / tot:5.1f} %  {k}")
shutil.rmtree(root, ignore_errors=True)
