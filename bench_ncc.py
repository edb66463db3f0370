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


def make_grid(dev, rows=2, cols=2, seed=1234):
    """rows x cols tiles cut from one seeded, 3x box-blurred bead field with per-tile integer jitter."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator(device=dev).manual_seed(seed)
    step = TILE[1] - OVERLAP
    shape = (TILE[0] + 8, rows * step + OVERLAP + 16, cols * step + OVERLAP + 16)
    field = torch.empty(shape, dtype=torch.float32, device=dev).uniform_(0.01, 0.02, generator=g)
    n = field.numel()
    nb = n // 512
    idx = torch.randint(0, n, (nb,), generator=g, device=dev)
    field.view(-1)[idx] = torch.empty(nb, dtype=torch.float32, device=dev).uniform_(0.2, 1.0, generator=g) * 27.0
    f = field[None, None]
    for _ in range(3):
        f = F.avg_pool3d(F.pad(f, (1, 1, 1, 1, 1, 1), mode="replicate"), 3, stride=1)
    field = (f[0, 0] / f.max()).clamp_(0, 1).contiguous()
    cpu_g = torch.Generator().manual_seed(seed)
    jit = torch.randint(-5, 6, (rows, cols, 3), generator=cpu_g)
    jit[..., 2] = torch.randint(-2, 3, (rows, cols), generator=cpu_g)
    tiles = [[None] * cols for _ in range(rows)]
    for r in range(rows):
        for c in range(cols):
            v, h, d = (int(x) for x in jit[r, c])
            z0, y0, x0 = 4 + d, 8 + r * step + v, 8 + c * step + h
            tiles[r][c] = field[z0:z0 + TILE[0], y0:y0 + TILE[1], x0:x0 + TILE[2]].contiguous()
    return tiles, jit, step


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
    out = {"metric": "NCC tile-pairs/sec", "value": round(n_pairs * repeats / dt, 3), "unit": "pairs/s",
           "ms_per_pair": round(dt * 1e3 / (n_pairs * repeats), 3),
           "workload": f"2x2 grid of {TILE[2]}x{TILE[1]}x{TILE[0]} tiles, overlap {OVERLAP}, search {DISPL}",
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
