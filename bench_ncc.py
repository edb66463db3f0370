"""Second metric of BASELINE.json: MIP-NCC tile-pairs/s on config-5-shaped tiles (2048 x 2048 x 32 fp32,
15 % overlap = 307 px, search (25, 25, 10)), tiles resident in HBM.  Called by bench.py (extra "ncc" object of
the JSON line) and runnable on its own:  python bench_ncc.py

N > 1 (bench.py --gpus N): the tile grid is shared by ROW BLOCKS (crossmips.tile_row_blocks): a rank keeps its rows plus the
first row behind its cut resident and computes the pairs that start in its rows -- every tile lives on one GPU, the cut row on
two, no collective on the data path (the reference farms (pair, layer) jobs over MPI ranks, Parastitcher.py:1367,1440-1560)."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TILE = (32, 2048, 2048)
OVERLAP = 307
DISPL = (25, 25, 10)


GRID = int(os.environ.get("MI_NCC_GRID", "8"))   # BASELINE config 5: an 8 x 8 grid (112 adjacent pairs)


def make_grid(dev, rows=GRID, cols=GRID, seed=1234, keep_rows=None):
    """rows x cols tiles cut, with per-tile integer jitter, from one seeded, 3x box-blurred bead field that is periodic in y and x
    with period 2 * step: every pair of adjacent tiles truly overlaps while the field stays small (the 8 x 8 mosaic itself would
    be 32 GB before blurring).  ``keep_rows``: only these rows are cut (the others stay None): the row block of one rank."""
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
        if keep_rows is not None and r not in keep_rows:
            continue
        for c in range(cols):
            v, h, d = (int(x) for x in jit[r, c])
            ys, xs = (r * step + v + ar_v) % period, (c * step + h + ar_h) % period
            tiles[r][c] = field[4 + d:4 + d + TILE[0]].index_select(1, ys).index_select(2, xs).contiguous()
    return tiles, jit, step


def mips_roofline(dev, tiles, pairs):
    """Dominant kernel of the pair pipeline: k_mips, the one streaming pass over both overlap views (compute_3_MIPs,
    compute_funcs.cu:502-521).  ALGORITHMIC bytes per launch = pairs x 2 tiles x dimk x dimi_v x dimj_v x 4 B (161 MB
    per C5 pair); duration = HIP events around 5 launches on the launch stream (mi_ncc_time_mips), per side of the grid."""
    import ctypes as C
    from ipp_amd import capi
    R, Cc = len(tiles), len(tiles[0])
    flat = [tiles[r][c] for r in range(R) for c in range(Cc)]
    some = next(t for t in flat if t is not None)
    dk, di, dj = (int(v) for v in some.shape)
    ptrs = (C.c_void_p * len(flat))(*[(t.data_ptr() if t is not None else None) for t in flat])
    out = {}
    tot_bytes, tot_ms, launches = 0.0, 0.0, 0
    for side, name in ((1, "west_east"), (0, "north_south")):
        mine = [p for p in pairs if p[4] == side]
        n = len(mine)
        if n == 0:
            continue
        a_idx = (C.c_int * n)(*[r * Cc + c for r, c, _, _, _ in mine])
        b_idx = (C.c_int * n)(*[rb * Cc + cb for _, _, rb, cb, _ in mine])
        ni, nj = (di - OVERLAP if side == 0 else 0), (dj - OVERLAP if side == 1 else 0)
        ms = C.c_float()
        capi.check(capi.lib().mi_ncc_time_mips(dev.index, capi.current_stream_ptr(dev), n, ptrs, a_idx, b_idx, dk, di, dj, ni, nj, side, 5,
                                               C.byref(ms)))
        nbytes = n * 2.0 * dk * (di - ni) * (dj - nj) * 4
        out[name] = {"pairs_per_launch": n, "launch_ms": round(ms.value, 4), "GBps": round(nbytes / (ms.value * 1e-3) / 1e9, 1)}
        tot_bytes += nbytes
        tot_ms += ms.value
        launches += 1
    ach = tot_bytes / (tot_ms * 1e-3) / 1e9
    # HBM bytes per launch from the committed PMC passes (profiles/collect.sh: FETCH_SIZE x 2 + WRITE_SIZE, 56 pairs per launch);
    # a constant of that profile, not a measurement of this run -- only quoted for the launch geometry it was taken on
    traffic, source = None, None
    for name in ("r05_ncc_pmc_traffic.json", "r04_ncc_pmc_traffic.json", "r03_ncc_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                ks = json.load(f)["kernels"]
                traffic = round(ks[next(k for k in ("k_mips5<4>", "k_mips<true>", "k_mips") if k in ks)]["hbm_bytes_per_launch"])
            source = "profiles/" + name
            break
        except (OSError, KeyError, ValueError, StopIteration):
            continue
    if not (len(pairs) == 112 and launches == 2):
        traffic, source = None, None
    return {"bound": "hbm", "kernel": "k_mips5 (six MIPs of every pair of a group in one streaming pass)", "achieved": round(ach, 1),
            "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": traffic, "traffic_source": source,
            "algorithmic_bytes_per_launch": int(tot_bytes / max(launches, 1)), "algorithmic_bytes_per_pair": int(2 * dk * (di - 0) * OVERLAP * 4),
            "launches": out}


def _same_record(ref, d):
    import numpy as np
    return (list(ref["coord"]) == list(d.VHD_coords) and list(ref["NCC_widths"]) == list(d.NCC_widths)
            and list(ref["wRangeThr"]) == list(d.wRangeThrs)
            and bool(np.allclose(np.array(ref["NCC_maxs"], np.float32), np.array(d.NCC_maxs, np.float32), atol=2e-6, equal_nan=True)))


def start_cpu_workers(n):
    """One idle worker process per host core for the CPU baseline (oracle/ncc_ref_worker.py).  MUST be called before anything in
    this process touches the GPU: a process that has initialised HIP must not start programs (fork + exec) on this pool."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "ncc_ref_worker.py")]
    return [subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, env=env) for _ in range(max(1, n))]


def stop_cpu_workers(workers):
    for w in workers or []:
        try:
            w.stdin.close()
            w.wait(timeout=10)
        except Exception:
            w.kill()


def cpu_baseline(tiles, res, workers):
    """The compiled reference (oracle/_ref, kind "reference"; the C restatement when it is absent) on the host: one pair on ONE core
    and one pair per worker on all cores at once (one process per pair: the reference keeps static state, and its own parallel
    driver is one MPI rank per job, Parastitcher.py:1440-1560).  Every record the CPU produces -- V, H, D, the three peaks, the
    three widths and the mutated wRangeThr -- is compared with the GPU's record of that pair."""
    import numpy as np
    from oracle import ncc_oracle
    from ipp_amd import crossmips
    kind = "ref" if ncc_oracle.have_ref() else "oracle"
    cores = len(workers)
    R, Cc = len(tiles), len(tiles[0])
    # a corner of the grid that holds max(cores, 8) pairs of both directions on few tiles
    rows, cols = min(R, 3), min(Cc, 4)
    cand = list(crossmips.enumerate_pairs(rows, cols))
    ns, we = [p for p in cand if p[4] == 0], [p for p in cand if p[4] == 1]
    picked = []
    while (ns or we) and len(picked) < max(cores, 8):
        if we:
            picked.append(we.pop(0))
        if ns and len(picked) < max(cores, 8):
            picked.append(ns.pop(0))
    tmp = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    tag = f"mi_ncc_{os.getpid()}"
    paths = {}
    try:
        for p in picked:
            for rc in ((p[0], p[1]), (p[2], p[3])):
                if rc not in paths:
                    paths[rc] = os.path.join(tmp, f"{tag}_{rc[0]}_{rc[1]}.npy")
                    np.save(paths[rc], tiles[rc[0]][rc[1]].cpu().numpy())

        def run(batch):                  # one pair per worker, started together
            live = workers[:len(batch)]
            for w, p in zip(live, batch):
                w.stdin.write(json.dumps({"a": paths[(p[0], p[1])], "b": paths[(p[2], p[3])], "displ": list(DISPL), "direction": p[4],
                                          "overlap": OVERLAP, "kind": kind}) + "\n")
                w.stdin.flush()
            for w in live:
                if w.stdout.readline().strip() != "ready":
                    raise RuntimeError("NCC CPU worker died")
            t0 = time.perf_counter()
            for w in live:
                w.stdin.write("go\n")
                w.stdin.flush()
            outs = [json.loads(w.stdout.readline()) for w in live]
            return time.perf_counter() - t0, outs

        t1, o1 = run(picked[:1])
        tn, on = run(picked[:cores])
        recs = {picked[0]: o1[0]}
        recs.update(dict(zip(picked[:cores], on)))
        rest = [p for p in picked[:8] if p not in recs]   # hosts with few cores: still eight pairs of both directions
        while rest:
            _, more = run(rest[:cores])
            recs.update(dict(zip(rest[:cores], more)))
            rest = rest[cores:]
    finally:
        for pth in paths.values():
            try:
                os.unlink(pth)
            except OSError:
                pass
    equal = sum(_same_record(o, res[p]) for p, o in recs.items())
    return {"value": round(1.0 / t1, 4), "unit": "pairs/s", "cores": 1, "kind": "reference" if kind == "ref" else "port",
            "sample": "one W-E pair of the same grid, one process on one core",
            "all_cores": {"value": round(len(on) / tn, 4), "unit": "pairs/s", "cores": cores,
                          "sample": f"{len(on)} different pairs (both directions) at once, one single-threaded process per pair, {tn:.1f} s"},
            "records_equal_gpu": f"{equal}/{len(recs)}",
            "records_compared": "coord V/H/D, NCC_maxs (2e-6), NCC_widths, wRangeThr of every pair the CPU ran",
            "pairs_compared": {"north_south": sum(1 for p in recs if p[4] == 0), "west_east": sum(1 for p in recs if p[4] == 1)}}


def run(dev, repeats=10, cpu_workers=None, rank=0, world=1, dist=None, dist_device=None):
    import torch
    from ipp_amd import crossmips
    blocks = crossmips.tile_row_blocks(GRID, world, GRID)
    r0, r1 = blocks[rank]
    keep = None if world == 1 else set(range(r0, min(r1 + 1, GRID)))
    tiles, jit, step = make_grid(dev, keep_rows=keep)
    block = None if world == 1 else (r0, r1)

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    res = crossmips.compute_displacements(tiles, OVERLAP, OVERLAP, *DISPL, row_block=block)  # warm-up + correctness
    ok = 0
    for (r, c, rb, cb, direction), d in res.items():
        dj = jit[rb, cb] - jit[r, c]
        nominal = [step if direction == 0 else 0, step if direction == 1 else 0]
        ok += all(d.VHD_coords[ax] == nominal[ax] + int(dj[ax]) for ax in range(2))
    sync()
    t0 = time.perf_counter()
    for _ in range(repeats):
        crossmips.compute_displacements(tiles, OVERLAP, OVERLAP, *DISPL, row_block=block)
    sync()
    dt = time.perf_counter() - t0
    n_mine = len(res)
    n_pairs = n_mine
    if dist is not None:
        t = torch.tensor([dt, float(ok), float(n_mine)], dtype=torch.float64, device=dist_device)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, ok, n_pairs = float(tmax[0]), int(t[1]), int(t[2])
    if rank != 0:
        return None
    stats = crossmips.ncc_stats()
    per_pair_s = dt / (n_pairs * repeats)
    roof = mips_roofline(dev, tiles, list(res.keys()))
    # whole pipeline against the same byte count, and its fp64 work (lag transforms + per-frequency correlation)
    roof["pipeline"] = {"algorithmic_GBps": round(roof["algorithmic_bytes_per_pair"] / per_pair_s / 1e9, 1),
                        "frac": round(roof["algorithmic_bytes_per_pair"] / per_pair_s / (8e12 * world), 4)}
    out = {"metric": "NCC tile-pairs/sec", "value": round(n_pairs * repeats / dt, 3), "unit": "pairs/s", "n_gpus": world,
           "ms_per_pair": round(dt * 1e3 / (n_pairs * repeats), 4), "roofline": roof, "path_counters": stats,
           "workload": f"{GRID}x{GRID} grid of {TILE[2]}x{TILE[1]}x{TILE[0]} tiles, overlap {OVERLAP}, search {DISPL}",
           "pairs_with_exact_VH_offsets": f"{ok}/{n_pairs}"}
    if world > 1:
        out["partition"] = {"kind": "tile-row blocks, balanced by pair count; the first row behind a cut is resident on two ranks",
                            "row_blocks": [list(b) for b in blocks], "pairs_on_rank0": n_mine,
                            "tiles_resident_on_rank0": sum(t is not None for row in tiles for t in row)}
        roof["note"] = "k_mips launches of rank 0's pairs"
    if world == 1:
        out["u16_tiles"] = int_leg(dev, tiles, jit, step, repeats, 16)
        out["u8_tiles"] = int_leg(dev, tiles, jit, step, repeats, 8)
    if cpu_workers and world == 1:
        out["cpu_baseline"] = cpu_baseline(tiles, res, cpu_workers)
    return out


def int_leg(dev, tiles, jit, step, repeats, bits):
    """Secondary lines, NOT the metric (BASELINE config 5 is defined on float32 tiles): the same grid with the tiles kept as 16-bit
    (8-bit) samples, as the reference's TIFF tiles are stored (it divides them by 65535 (255) when it loads them,
    tiff2D.cpp:606-610) -- mi_ncc_mips_batch_u16 (_u8) reads those in its MIP pass and returns the records of the converted tiles
    bit for bit."""
    import ctypes as C
    import torch
    from ipp_amd import capi, crossmips
    top = float((1 << bits) - 1)
    t16 = [[(t * top).round_().clamp_(0, top).to(torch.uint16 if bits == 16 else torch.uint8) for t in row] for row in tiles]
    res = crossmips.compute_displacements(t16, OVERLAP, OVERLAP, *DISPL)
    ok = 0
    for (r, c, rb, cb, direction), d in res.items():
        dj = jit[rb, cb] - jit[r, c]
        nominal = [step if direction == 0 else 0, step if direction == 1 else 0]
        ok += all(d.VHD_coords[ax] == nominal[ax] + int(dj[ax]) for ax in range(2))
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(repeats):
        crossmips.compute_displacements(t16, OVERLAP, OVERLAP, *DISPL)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    n = len(res)
    flat = [t for row in t16 for t in row]
    Cc = len(t16[0])
    dk, di, dj_ = (int(v) for v in flat[0].shape)
    ptrs = (C.c_void_p * len(flat))(*[t.data_ptr() for t in flat])
    launches = {}
    for side, name in ((1, "west_east"), (0, "north_south")):
        mine = [p for p in res.keys() if p[4] == side]
        a_idx = (C.c_int * len(mine))(*[r * Cc + c for r, c, _, _, _ in mine])
        b_idx = (C.c_int * len(mine))(*[rb * Cc + cb for _, _, rb, cb, _ in mine])
        ni, nj = (di - OVERLAP if side == 0 else 0), (dj_ - OVERLAP if side == 1 else 0)
        ms = C.c_float()
        timer = capi.lib().mi_ncc_time_mips_u16 if bits == 16 else capi.lib().mi_ncc_time_mips_u8
        capi.check(timer(dev.index, capi.current_stream_ptr(dev), len(mine), ptrs, top, a_idx, b_idx, dk, di, dj_, ni, nj, side, 5, C.byref(ms)))
        nbytes = len(mine) * 2.0 * dk * (di - ni) * (dj_ - nj) * (bits // 8)
        launches[name] = {"pairs_per_launch": len(mine), "launch_ms": round(ms.value, 4), "GBps": round(nbytes / (ms.value * 1e-3) / 1e9, 1)}
    return {"value": round(n * repeats / dt, 3), "unit": "pairs/s", "pairs_with_exact_VH_offsets": f"{ok}/{n}",
            "kernel": f"k_mips_int<{bits // 8}> ({32 // bits} columns per lane, packed 16-bit maxima)", "bytes_per_sample": bits // 8, "launches": launches,
            "note": f"tiles stored as uint{bits} samples = round(float tile * {int(top)}); not the headline metric"}


if __name__ == "__main__":
    import bench
    pool = start_cpu_workers(bench.host_cores())      # before the GPU is touched
    import torch
    try:
        print(json.dumps(run(torch.device("cuda", 0), cpu_workers=pool)))
    finally:
        stop_cpu_workers(pool)
