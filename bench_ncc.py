"""Second metric of BASELINE.json: MIP-NCC tile-pairs/s on config-5-shaped tiles (2048 x 2048 x 32 fp32,
15 % overlap = 307 px, search (25, 25, 10)), tiles resident in HBM.  Called by bench.py (extra "ncc" object of
the JSON line) and runnable on its own:  python bench_ncc.py"""
from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TILE = (32, 2048, 2048)
OVERLAP = 307
DISPL = (25, 25, 10)


GRID = int(os.environ.get("MI_NCC_GRID", "8"))   # BASELINE config 5: an 8 x 8 grid (112 adjacent pairs)


def make_grid(dev, rows=GRID, cols=GRID, seed=1234):
    """rows x cols tiles cut, with per-tile integer jitter, from one seeded, 3x box-blurred bead field that is periodic in y and x
    with period 2 * step: every pair of adjacent tiles truly overlaps while the field stays small (the 8 x 8 mosaic itself would
    be 32 GB before blurring)."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator(device=dev).manual_seed(seed)
    step = TILE[1] - OVERLAP
    period = 2 * step
    shape = (TILE[0] + 8, period, period)
    field = torch.empty(shape, dtype=torch.float32, device=dev).uniform_(0.01, 0.02, generator=g)
    n = field.numel()
    nb = n // 512
    idx = torch.randint(0, n, (nb,), generator=g, device=dev)
    field.view(-1)[idx] = torch.empty(nb, dtype=torch.float32, device=dev).uniform_(0.2, 1.0, generator=g) * 27.0
    f = field[None, None]
    for _ in range(3):
        f = F.pad(F.pad(f, (1, 1, 1, 1, 0, 0), mode="circular"), (0, 0, 0, 0, 1, 1), mode="replicate")
        f = F.avg_pool3d(f, 3, stride=1)
    field = (f[0, 0] / f.max()).clamp_(0, 1).contiguous()
    del f
    cpu_g = torch.Generator().manual_seed(seed)
    jit = torch.randint(-5, 6, (rows, cols, 3), generator=cpu_g)
    jit[..., 2] = torch.randint(-2, 3, (rows, cols), generator=cpu_g)
    ar_v, ar_h = torch.arange(TILE[1], device=dev), torch.arange(TILE[2], device=dev)
    tiles = [[None] * cols for _ in range(rows)]
    for r in range(rows):
        for c in range(cols):
            v, h, d = (int(x) for x in jit[r, c])
            ys, xs = (r * step + v + ar_v) % period, (c * step + h + ar_h) % period
            tiles[r][c] = field[4 + d:4 + d + TILE[0]].index_select(1, ys).index_select(2, xs).contiguous()
    return tiles, jit, step


def mips_roofline(dev, tiles):
    """Dominant kernel of the pair pipeline: k_mips, the one streaming pass over both overlap views (compute_3_MIPs,
    compute_funcs.cu:502-521).  ALGORITHMIC bytes per launch = pairs x 2 tiles x dimk x dimi_v x dimj_v x 4 B (161 MB
    per C5 pair); duration = HIP events around 5 launches on the launch stream (mi_ncc_time_mips), per side of the grid."""
    import ctypes as C
    from ipp_amd import capi, crossmips
    R, Cc = len(tiles), len(tiles[0])
    flat = [tiles[r][c] for r in range(R) for c in range(Cc)]
    dk, di, dj = (int(v) for v in flat[0].shape)
    ptrs = (C.c_void_p * len(flat))(*[t.data_ptr() for t in flat])
    out = {}
    tot_bytes, tot_ms = 0.0, 0.0
    for side, name in ((1, "west_east"), (0, "north_south")):
        pairs = [p for p in crossmips.enumerate_pairs(R, Cc) if p[4] == side]
        n = len(pairs)
        a_idx = (C.c_int * n)(*[r * Cc + c for r, c, _, _, _ in pairs])
        b_idx = (C.c_int * n)(*[rb * Cc + cb for _, _, rb, cb, _ in pairs])
        ni, nj = (di - OVERLAP if side == 0 else 0), (dj - OVERLAP if side == 1 else 0)
        ms = C.c_float()
        capi.check(capi.lib().mi_ncc_time_mips(dev.index, capi.current_stream_ptr(dev), n, ptrs, a_idx, b_idx, dk, di, dj, ni, nj, side, 5,
                                               C.byref(ms)))
        nbytes = n * 2.0 * dk * (di - ni) * (dj - nj) * 4
        out[name] = {"pairs_per_launch": n, "launch_ms": round(ms.value, 4), "GBps": round(nbytes / (ms.value * 1e-3) / 1e9, 1)}
        tot_bytes += nbytes
        tot_ms += ms.value
    ach = tot_bytes / (tot_ms * 1e-3) / 1e9
    traffic = None   # HBM bytes per launch from the committed PMC passes (profiles/collect.sh: FETCH_SIZE x 2 + WRITE_SIZE, 56 pairs)
    try:
        with open(os.path.join(ROOT, "profiles", "r02_ncc_pmc_traffic.json")) as f:
            traffic = round(json.load(f)["kernels"]["k_mips"]["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        pass
    return {"bound": "hbm", "kernel": "k_mips (six MIPs of every pair of a group in one streaming pass)", "achieved": round(ach, 1),
            "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": traffic,
            "algorithmic_bytes_per_launch": int(tot_bytes / 2), "algorithmic_bytes_per_pair": int(2 * dk * (di - 0) * OVERLAP * 4), "launches": out}


def run(dev, repeats=3, cpu=True):
    import torch
    from ipp_amd import crossmips
    tiles, jit, step = make_grid(dev)
    res = crossmips.compute_displacements(tiles, OVERLAP, OVERLAP, *DISPL)  # warm-up + correctness
    ok = 0
    for (r, c, rb, cb, direction), d in res.items():
        dj = jit[rb, cb] - jit[r, c]
        nominal = [step if direction == 0 else 0, step if direction == 1 else 0]
        ok += all(d.VHD_coords[ax] == nominal[ax] + int(dj[ax]) for ax in range(2))
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(repeats):
        crossmips.compute_displacements(tiles, OVERLAP, OVERLAP, *DISPL)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    n_pairs = len(res)
    stats = crossmips.ncc_stats()
    per_pair_s = dt / (n_pairs * repeats)
    roof = mips_roofline(dev, tiles)
    # whole pipeline against the same byte count, and its fp64 work (lag transforms + per-frequency correlation)
    roof["pipeline"] = {"algorithmic_GBps": round(roof["algorithmic_bytes_per_pair"] / per_pair_s / 1e9, 1),
                        "frac": round(roof["algorithmic_bytes_per_pair"] / per_pair_s / 8e12, 4)}
    out = {"metric": "NCC tile-pairs/sec", "value": round(n_pairs * repeats / dt, 3), "unit": "pairs/s",
           "ms_per_pair": round(dt * 1e3 / (n_pairs * repeats), 3), "roofline": roof, "path_counters": stats,
           "workload": f"{len(tiles)}x{len(tiles[0])} grid of {TILE[2]}x{TILE[1]}x{TILE[0]} tiles, overlap {OVERLAP}, search {DISPL}",
           "pairs_with_exact_VH_offsets": f"{ok}/{n_pairs}"}
    if cpu:
        from oracle import ncc_oracle
        kind = "ref" if ncc_oracle.have_ref() else "oracle"
        A, B = tiles[0][0].cpu().numpy(), tiles[0][1].cpu().numpy()
        t0 = time.perf_counter()
        r = ncc_oracle.pdalgo_execute(A, B, *DISPL, 1, OVERLAP, kind=kind)
        ct = time.perf_counter() - t0
        d = res[(0, 0, 0, 1, 1)]
        out["cpu_baseline"] = {"value": round(1.0 / ct, 4), "unit": "pairs/s", "cores": 1,
                               "kind": "reference" if kind == "ref" else "port",
                               "sample": "one W-E pair of the same grid, single thread",
                               "offsets_equal_gpu": r["coord"] == d.VHD_coords and r["NCC_widths"] == d.NCC_widths}
    return out


if __name__ == "__main__":
    import torch
    print(json.dumps(run(torch.device("cuda", 0))))
