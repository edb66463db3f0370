"""NCC batch on the C5 grid: wall time of the raw C-ABI batch call (mi_ncc_mips_batch) vs the Python pair enumeration around it.
    python profiles/ncc_batch_probe.py [repeats]          (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench_ncc  # noqa: E402
from ipp_amd import capi, crossmips  # noqa: E402
from ipp_amd.capi import NccDescr, NccParams, check, lib  # noqa: E402

dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
tiles, jit, step = bench_ncc.make_grid(dev)
U16 = bool(os.environ.get("PROBE_U16"))   # the grid stored as 16-bit samples (mi_ncc_mips_batch_u16)
if U16:
    tiles = [[(t * 65535.0).round_().clamp_(0, 65535).to(torch.uint16) for t in row] for row in tiles]
R, Cc = len(tiles), len(tiles[0])
flat = [tiles[r][c] for r in range(R) for c in range(Cc)]
pairs = list(crossmips.enumerate_pairs(R, Cc))
n = len(pairs)
dim_D, dim_V, dim_H = (int(s) for s in flat[0].shape)
ptrs = (C.c_void_p * len(flat))(*[t.data_ptr() for t in flat])
a_idx = (C.c_int * n)(*[r * Cc + c for r, c, _, _, _ in pairs])
b_idx = (C.c_int * n)(*[rb * Cc + cb for _, _, rb, cb, _ in pairs])
ov = bench_ncc.OVERLAP
ni = (C.c_int * n)(*[dim_V - ov if d == 0 else 0 for *_, d in pairs])
nj = (C.c_int * n)(*[dim_H - ov if d == 1 else 0 for *_, d in pairs])
side = (C.c_int * n)(*[d for *_, d in pairs])


def raw():
    params = (NccParams * n)()
    for q in range(n):
        lib().mi_ncc_default_params(*bench_ncc.DISPL, C.byref(params[q]))
    out = (NccDescr * n)()
    if U16:
        check(lib().mi_ncc_mips_batch_u16(dev.index, capi.current_stream_ptr(dev), n, ptrs, 65535.0, a_idx, b_idx, dim_D, dim_V, dim_H, ni, nj,
                                          bench_ncc.DISPL[2], bench_ncc.DISPL[0], bench_ncc.DISPL[1], side, params, out))
        return out
    check(lib().mi_ncc_mips_batch(dev.index, capi.current_stream_ptr(dev), n, ptrs, a_idx, b_idx, dim_D, dim_V, dim_H, ni, nj,
                                  bench_ncc.DISPL[2], bench_ncc.DISPL[0], bench_ncc.DISPL[1], side, params, out))
    return out


raw()
torch.cuda.synchronize(dev)
t0 = time.perf_counter()
for _ in range(reps):
    raw()
torch.cuda.synchronize(dev)
t_raw = (time.perf_counter() - t0) / reps
t0 = time.perf_counter()
for _ in range(reps):
    crossmips.compute_displacements(tiles, ov, ov, *bench_ncc.DISPL)
torch.cuda.synchronize(dev)
t_py = (time.perf_counter() - t0) / reps
print(f"pairs {n}: raw batch call {t_raw * 1e3:.2f} ms ({n / t_raw:.0f} pairs/s), compute_displacements {t_py * 1e3:.2f} ms "
      f"({n / t_py:.0f} pairs/s)", flush=True)
if os.environ.get("MI_PROBE_FAST_EXIT"):
    os._exit(0)
