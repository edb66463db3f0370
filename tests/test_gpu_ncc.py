"""GPU parity: MIP-NCC registration (through the C ABI) against the golden vectors of the compiled reference
and against the C oracle.  Integer outputs (offsets, widths, mutated wRangeThr) must be bit-exact."""
import numpy as np
import pytest
import torch

from oracle import ncc_oracle as N
from tests.golden_util import case_inputs

pytestmark = pytest.mark.gpu


def _names(g):
    return [str(n) for n in g["names"]]


@pytest.mark.parametrize("idx", range(13))
def test_golden_cases(dev, ncc_golden, idx):
    from ipp_amd import crossmips
    g = ncc_golden
    name = _names(g)[idx]
    A, B, overlap, side, dmax = case_inputs(g, name)
    a, b = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    d = crossmips.PDAlgoMIPNCC.execute(a, b, dmax[0], dmax[1], dmax[2], side, overlap)
    assert d.VHD_coords == list(g[f"{name}/coord"])
    assert d.NCC_widths == list(g[f"{name}/NCC_widths"])
    assert d.wRangeThrs == list(g[f"{name}/wRangeThr"])
    assert d.invWidths == [int(g[f"{name}/INF_W"])] * 3
    assert d.delays == list(dmax)
    want = g[f"{name}/NCC_maxs"]
    assert np.allclose(np.array(d.NCC_maxs, np.float32), want, rtol=0, atol=2e-6, equal_nan=True)
    # building blocks: MIPs exact, NCC maps to float rounding of an fp64 sum in a different order
    dimk, dimi, dimj = A.shape
    ni = dimi - overlap if side == 0 else 0
    nj = dimj - overlap if side == 1 else 0
    mips = crossmips.compute_mips(a, b, ni, nj, side)
    for m, nm in enumerate(["xy1", "xz1", "yz1", "xy2", "xz2", "yz2"]):
        assert np.array_equal(mips[m].cpu().numpy(), g[f"{name}/mip_{nm}"]), nm
    di, dj, dk = (int(v) for v in g[f"{name}/delays_ijk"])
    for m, (nm, du, dv) in enumerate([("xy", di, dj), ("xz", di, dk), ("yz", dj, dk)]):
        want_map = g[f"{name}/map_{nm}"]
        for lag in (False, True):      # shift-by-shift cross terms / the lag transform of the batched pipeline
            got = crossmips.compute_NCC_map(mips[m], mips[m + 3], du, dv, lag=lag).cpu().numpy()
            assert np.array_equal(np.isnan(got), np.isnan(want_map))
            assert _ulps(got, want_map) <= MAX_ULPS, (nm, lag, _ulps(got, want_map))


def _ulps(got, want):
    """largest distance between two float32 arrays in units in the last place (NaN pattern must already agree)"""
    ok = ~np.isnan(want)
    if not ok.any():
        return 0
    a = got[ok].astype(np.float32).view(np.int32).astype(np.int64)
    b = want[ok].astype(np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7fffffff), a)
    b = np.where(b < 0, -(b & 0x7fffffff), b)
    return int(np.abs(a - b).max())


# device maps vs the reference's maps: both are fp64 sums rounded once to float, in different orders -> at most one unit in
# the last place on values of magnitude ~1 (profiles/r02_ncc_map_ulps.txt holds the measured distribution); entries near 0 can
# sit more float steps apart for the same absolute difference, hence the absolute bound next to it
MAX_ULPS = 4


def test_golden_maps_ulp_statistics(dev, ncc_golden):
    """Records, per golden case and route, how far the device maps are from the reference's maps in ulps and in absolute terms
    (written to gpurun_out/ when that directory exists: the committed copy is profiles/r02_ncc_map_ulps.txt)."""
    import os
    from ipp_amd import crossmips
    g = ncc_golden
    lines = ["case plane route max_ulps max_abs_diff entries entries_differing"]
    worst = 0
    for name in _names(g):
        di, dj, dk = (int(v) for v in g[f"{name}/delays_ijk"])
        mips = [torch.from_numpy(g[f"{name}/mip_{nm}"]).to(dev) for nm in ("xy1", "xz1", "yz1", "xy2", "xz2", "yz2")]
        for m, (nm, du, dv) in enumerate([("xy", di, dj), ("xz", di, dk), ("yz", dj, dk)]):
            want = g[f"{name}/map_{nm}"]
            for lag in (False, True):
                got = crossmips.compute_NCC_map(mips[m], mips[m + 3], du, dv, lag=lag).cpu().numpy()
                ok = ~np.isnan(want)
                u = _ulps(got, want)
                big = np.abs(want) > 1e-3 if ok.any() else ok
                worst = max(worst, _ulps(np.where(big, got, 0), np.where(big, want, 0)) if ok.any() else 0)
                diff = float(np.abs(got[ok] - want[ok]).max()) if ok.any() else 0.0
                lines.append(f"{name} {nm} {'lag' if lag else 'direct'} {u} {diff:.3e} {int(ok.sum())} {int((got[ok] != want[ok]).sum())}")
                assert diff <= 2.5e-7, (name, nm, lag, diff)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "r02_ncc_map_ulps.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    assert worst <= 2, worst


@pytest.mark.parametrize("idx", range(13))
def test_golden_cases_direct_path(dev, ncc_golden, idx, monkeypatch):
    """MI_NCC_DIRECT=1: the per-pair path (shift-by-shift cross terms, host-driven refinement) -- the fallback and careful path
    of the batched pipeline -- reproduces the same golden integers."""
    from ipp_amd import crossmips
    monkeypatch.setenv("MI_NCC_DIRECT", "1")
    g = ncc_golden
    name = _names(g)[idx]
    A, B, overlap, side, dmax = case_inputs(g, name)
    d = crossmips.PDAlgoMIPNCC.execute(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), dmax[0], dmax[1], dmax[2], side, overlap)
    assert d.VHD_coords == list(g[f"{name}/coord"]) and d.NCC_widths == list(g[f"{name}/NCC_widths"])
    assert d.wRangeThrs == list(g[f"{name}/wRangeThr"])
    assert np.allclose(np.array(d.NCC_maxs, np.float32), g[f"{name}/NCC_maxs"], rtol=0, atol=2e-6, equal_nan=True)


def test_random_pairs_vs_oracle(dev):
    """Random tile shapes, overlaps, shifts and search ranges against the oracle (MI_TEST_SWEEP: number of trials, default 8)."""
    import os
    from ipp_amd import crossmips
    rng = np.random.default_rng(int(os.environ.get("MI_TEST_SWEEP_SEED", "2024")))
    for trial in range(int(os.environ.get("MI_TEST_SWEEP", "8"))):
        side = trial % 2
        shape = (int(rng.integers(26, 40)), int(rng.integers(90, 200)), int(rng.integers(90, 200)))
        ov = int(rng.integers(30, 60))
        sv, sh, sd = int(rng.integers(3, 14)), int(rng.integers(3, 14)), int(rng.integers(0, 5))
        shift = tuple(int(v) for v in rng.integers(-5, 6, size=3))
        A, B = N.tile_pair(shape, ov, side, shift, seed=300 + trial)
        want = N.pdalgo_execute(A, B, sv, sh, sd, side, ov, kind="oracle")
        got = crossmips.PDAlgoMIPNCC.execute(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), sv, sh, sd, side, ov)
        case = (trial, shape, ov, (sv, sh, sd), shift)
        assert got.VHD_coords == want["coord"] and got.NCC_widths == want["NCC_widths"], case
        assert got.wRangeThrs == want["wRangeThr"], case
        assert np.allclose(np.array(got.NCC_maxs, np.float32), want["NCC_maxs"], atol=2e-6, equal_nan=True), case


def test_wide_search_and_wide_mips_vs_oracle(dev):
    """Launch geometries the golden cases do not reach: a search range of 40 (81 shifts = 11 blocks of 8 -> two groups of
    blocks per u-row), MIPs wider than one 512-column segment, and the refinement's block lists on such maps."""
    from ipp_amd import crossmips
    for side, shape, ov, shift in ((0, (12, 150, 1100), 120, (3, -4, 1)), (1, (10, 1100, 160), 130, (-6, 2, 0))):
        A, B = N.tile_pair(shape, ov, side, shift, seed=77 + side)
        want = N.pdalgo_execute(A, B, 40, 40, 3, side, ov, kind="oracle", debug=True)
        got = crossmips.PDAlgoMIPNCC.execute(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), 40, 40, 3, side, ov)
        assert got.VHD_coords == want["coord"] and got.NCC_widths == want["NCC_widths"]
        assert np.allclose(np.array(got.NCC_maxs, np.float32), want["NCC_maxs"], atol=2e-6, equal_nan=True)
        di, dj, dk = want["delays"]
        for m, (du, dv) in enumerate([(di, dj), (di, dk), (dj, dk)]):
            mp = crossmips.compute_NCC_map(torch.from_numpy(want["mips"][m]).to(dev), torch.from_numpy(want["mips"][m + 3]).to(dev), du, dv)
            assert np.allclose(mp.cpu().numpy(), want["maps"][m], rtol=0, atol=2e-6, equal_nan=True), (side, m)


def test_overlap_wider_than_the_prefetch_registers_vs_oracle(dev):
    """An xy plane whose SHORT axis is 1650 samples: a tile of the correlation kernel (ncc_lag.hip:k_lag_mac) no longer fits the registers
    that carry the next tile's spectra, so the kernel variant that fetches the next tile behind the current one runs (and one
    frequency per work-group instead of four)."""
    from ipp_amd import crossmips
    for side, shape, ov, shift in ((0, (8, 1700, 1800), 1650, (2, -3, 0)), (1, (8, 1800, 1700), 1650, (-4, 1, 1))):
        A, B = N.tile_pair(shape, ov, side, shift, seed=900 + side)
        want = N.pdalgo_execute(A, B, 8, 8, 2, side, ov, kind="oracle")
        got = crossmips.PDAlgoMIPNCC.execute(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), 8, 8, 2, side, ov)
        assert got.VHD_coords == want["coord"] and got.NCC_widths == want["NCC_widths"], (side, got.VHD_coords, want["coord"])
        assert got.wRangeThrs == want["wRangeThr"]
        assert np.allclose(np.array(got.NCC_maxs, np.float32), want["NCC_maxs"], atol=2e-6, equal_nan=True)


@pytest.mark.parametrize("tile,ov,scale", [((30, 130, 150), 41, 65535.0), ((32, 96, 256), 64, 65535.0), ((9, 70, 258), 35, 255.0),
                                           ((26, 200, 132), 57, 65535.0), ((31, 90, 268), 75, 255.0), ((12, 300, 96), 33, 255.0)])
def test_u16_tiles_give_the_records_of_the_converted_tiles(dev, tile, ov, scale, monkeypatch):
    """Tiles kept as the 16-bit samples they were loaded from (mi_ncc_mips_batch_u16: packed 16-bit MIP kernel, two columns per
    lane) against the same tiles converted like the reference converts them (sample / 65535 or / 255, tiff2D.cpp:606-610): every
    field of every record identical -- odd and even view origins, odd view widths, partial last bands, stacks that are no
    multiple of four slices -- through the batched pipeline and through the per-pair path."""
    from ipp_amd import crossmips
    rng = np.random.default_rng(5)
    R_, C_ = 2, 3
    step_v, step_h = tile[1] - ov, tile[2] - ov
    field = N.bead_field((tile[0] + 6, R_ * step_v + ov + 12, C_ * step_h + ov + 12), seed=21, density=1 / 250)
    top = 255 if scale == 255.0 else 65535
    q = np.clip(np.rint(field / max(float(field.max()), 1e-6) * top), 0, top).astype(np.uint16)
    tiles16 = [[None] * C_ for _ in range(R_)]
    for r in range(R_):
        for c in range(C_):
            v, h, d = (int(x) for x in rng.integers(-2, 3, size=3))
            z0, y0, x0 = 3 + max(-1, min(1, d)), 6 + r * step_v + v, 6 + c * step_h + h
            tiles16[r][c] = torch.from_numpy(np.ascontiguousarray(q[z0:z0 + tile[0], y0:y0 + tile[1], x0:x0 + tile[2]])).to(dev)
    # (numpy's float32 division is the reference's `(real_t) sample / 65535.0f`; torch divides by multiplying with the reciprocal)
    tilesf = [[torch.from_numpy(t.cpu().numpy().astype(np.float32) / np.float32(scale)).to(dev) for t in row] for row in tiles16]
    for direct in ("0", "1"):
        monkeypatch.setenv("MI_NCC_DIRECT", direct)
        want = crossmips.compute_displacements(tilesf, ov, ov, 7, 7, 2)
        got = crossmips.compute_displacements(tiles16, ov, ov, 7, 7, 2, sample_scale=scale)
        assert got.keys() == want.keys() and len(got) == 2 * R_ * C_ - R_ - C_
        for k in want:
            a, b = got[k], want[k]
            assert a.VHD_coords == b.VHD_coords and a.NCC_widths == b.NCC_widths and a.wRangeThrs == b.wRangeThrs, (direct, k)
            assert np.array_equal(np.array(a.NCC_maxs, np.float32).view(np.uint32), np.array(b.NCC_maxs, np.float32).view(np.uint32)), (direct, k)
        if scale == 255.0:   # ... and the same samples as 8-bit tensors (four columns per lane)
            tiles8 = [[t.to(torch.uint8) for t in row] for row in tiles16]
            got8 = crossmips.compute_displacements(tiles8, ov, ov, 7, 7, 2)
            for k in want:
                a, b = got8[k], want[k]
                assert a.VHD_coords == b.VHD_coords and a.NCC_widths == b.NCC_widths and a.wRangeThrs == b.wRangeThrs, (direct, k)
                assert np.array_equal(np.array(a.NCC_maxs, np.float32).view(np.uint32), np.array(b.NCC_maxs, np.float32).view(np.uint32)), (direct, k)
    monkeypatch.delenv("MI_NCC_DIRECT")
    # what the 16-bit kernel does not take (an odd row length) is converted on the device like the reference converts it: same records
    odd16 = [[t[:, :, :-1].contiguous() for t in row] for row in tiles16]
    oddf = [[t[:, :, :-1].contiguous() for t in row] for row in tilesf]
    want = crossmips.compute_displacements(oddf, ov, ov - 1, 7, 7, 2)
    got = crossmips.compute_displacements(odd16, ov, ov - 1, 7, 7, 2, sample_scale=scale)
    for k in want:
        assert got[k].VHD_coords == want[k].VHD_coords and got[k].NCC_widths == want[k].NCC_widths
        assert np.array_equal(np.array(got[k].NCC_maxs, np.float32).view(np.uint32), np.array(want[k].NCC_maxs, np.float32).view(np.uint32))
    # the C entry itself refuses such tiles
    import ctypes as C
    from ipp_amd import capi
    z = (C.c_int * 1)(0)
    one = (C.c_void_p * 1)(odd16[0][0].data_ptr())
    p = (capi.NccParams * 1)()
    o = (capi.NccDescr * 1)()
    rc = capi.lib().mi_ncc_mips_batch_u16(dev.index, None, 1, one, 65535.0, z, z, *[int(v) for v in odd16[0][0].shape], z, z, 2, 7, 7, z, p, o)
    assert rc != 0


def test_u16_tiles_random_geometries(dev):
    """Random stack depths (1 .. 32 slices, also fewer than the four waves of a work-group), tile sizes, overlaps and search ranges:
    the 16-bit route equals the float route on every field (MI_TEST_SWEEP trials, default 10)."""
    import os
    from ipp_amd import crossmips
    rng = np.random.default_rng(int(os.environ.get("MI_TEST_SWEEP_SEED", "77")))
    for trial in range(int(os.environ.get("MI_TEST_SWEEP", "10"))):
        D = int(rng.choice([1, 2, 3, 5, 8, 13, 26, 31, 32]))
        V, H = int(rng.integers(40, 220)), 2 * int(rng.integers(20, 160))
        ov_v, ov_h = int(rng.integers(26, min(V, 90))), int(rng.integers(26, min(H, 90)))
        sv, sh, sd = int(rng.integers(2, 9)), int(rng.integers(2, 9)), int(rng.integers(0, 3))
        grid = [[None, None], [None, None]]
        for r in range(2):
            for c in range(2):
                a = N.bead_field((D, V, H), seed=500 + 17 * trial + 2 * r + c, density=1 / 200)
                grid[r][c] = np.clip(np.rint(a / max(float(a.max()), 1e-6) * 65535), 0, 65535).astype(np.uint16)
        top = 65535.0
        if trial % 2:   # 8-bit samples (rows of whole words when H is a multiple of 4, converted on the device otherwise)
            grid = [[(t >> 8).astype(np.uint8) for t in row] for row in grid]
            top = 255.0
        t16 = [[torch.from_numpy(t).to(dev) for t in row] for row in grid]
        tf = [[torch.from_numpy(t.astype(np.float32) / np.float32(top)).to(dev) for t in row] for row in grid]
        want = crossmips.compute_displacements(tf, ov_v, ov_h, sv, sh, sd)
        got = crossmips.compute_displacements(t16, ov_v, ov_h, sv, sh, sd)
        case = (trial, (D, V, H), (ov_v, ov_h), (sv, sh, sd))
        for k in want:
            a, b = got[k], want[k]
            assert a.VHD_coords == b.VHD_coords and a.NCC_widths == b.NCC_widths and a.wRangeThrs == b.wRangeThrs, (case, k)
            assert np.array_equal(np.array(a.NCC_maxs, np.float32).view(np.uint32), np.array(b.NCC_maxs, np.float32).view(np.uint32)), (case, k)


def test_host_pointer_entry_and_errors(dev):
    import ctypes as C
    from ipp_amd import capi, crossmips
    A, B = N.tile_pair((26, 64, 64), 30, 1, (1, 0, 0), seed=1)
    p = crossmips.NCC_parms_t(6, 6, 1)
    out = capi.NccDescr()
    rc = capi.lib().mi_ncc_mips_host(0, None, A.ctypes.data, B.ctypes.data, 26, 64, 64, 0, 0, 34, 1, 6, 6, 1, C.byref(p),
                                     C.byref(out))
    assert rc == 0
    want = N.pdalgo_execute(A, B, 6, 6, 1, 1, 30, kind="oracle")
    assert list(out.coord) == want["coord"] and list(out.NCC_widths) == want["NCC_widths"]
    # wRangeThr larger than the search range: the reference throws (libcrossmips.cpp:212-219)
    p = crossmips.NCC_parms_t(10, 10, 10)
    with pytest.raises(capi.MiError, match="too large"):
        crossmips.norm_cross_corr_mips(A, B, ni=0, nj=34, delayk=2, delayi=10, delayj=10, side=1, NCC_params=p)
    with pytest.raises(ValueError, match="missing configuration"):
        crossmips.norm_cross_corr_mips(A, B, nj=34, side=1)
    with pytest.raises(capi.MiError, match="unexpected alignment"):
        crossmips.norm_cross_corr_mips(A, B, nj=34, delayk=1, delayi=6, delayj=6, side=3, NCC_params=crossmips.NCC_parms_t(6, 6, 1))
    with pytest.raises(ValueError, match="same dimensions"):
        crossmips.PDAlgoMIPNCC.execute(torch.zeros((4, 8, 8)), torch.zeros((4, 8, 9)), 1, 1, 1, 0, 4)


def test_grid_batch_recovers_jitter(dev):
    """3x3 grid cut from one bead field with per-tile integer jitter (config 5 in miniature): every reliable
    pair must return nominal + jitter difference, and the batch entry must equal pair-by-pair calls."""
    from ipp_amd import crossmips
    rng = np.random.default_rng(77)
    tile, ov, R_, C_ = (30, 128, 128), 40, 3, 3
    step = tile[1] - ov
    field = N.bead_field((tile[0] + 8, R_ * step + ov + 16, C_ * step + ov + 16), seed=9, density=1 / 300)
    jit = rng.integers(-3, 4, size=(R_, C_, 3))
    jit[..., 2] = rng.integers(-1, 2, size=(R_, C_))
    tiles = [[None] * C_ for _ in range(R_)]
    for r in range(R_):
        for c in range(C_):
            v, h, d = jit[r, c]
            z0, y0, x0 = 4 + d, 8 + r * step + v, 8 + c * step + h
            tiles[r][c] = torch.from_numpy(np.ascontiguousarray(field[z0:z0 + tile[0], y0:y0 + tile[1], x0:x0 + tile[2]])).to(dev)
    res = crossmips.compute_displacements(tiles, ov, ov, 8, 8, 3)
    assert len(res) == 2 * R_ * C_ - R_ - C_
    reliable = 0
    for (r, c, rb, cb, direction), d in res.items():
        dj = jit[rb, cb] - jit[r, c]
        nominal = [step if direction == 0 else 0, step if direction == 1 else 0, 0]
        want = N.pdalgo_execute(tiles[r][c].cpu().numpy(), tiles[rb][cb].cpu().numpy(), 8, 8, 3, direction, ov, kind="oracle")
        assert d.VHD_coords == want["coord"] and d.NCC_widths == want["NCC_widths"] and d.wRangeThrs == want["wRangeThr"]
        for ax in range(2):
            if d.NCC_widths[ax] < d.invWidths[ax]:  # reliable estimate: must be the ground truth
                reliable += 1
                assert d.VHD_coords[ax] == nominal[ax] + int(dj[ax]), (r, c, direction, ax)
        single = crossmips.PDAlgoMIPNCC.execute(tiles[r][c], tiles[rb][cb], 8, 8, 3, direction, ov)
        assert single.VHD_coords == d.VHD_coords and single.NCC_widths == d.NCC_widths
    assert reliable >= 0.75 * 2 * len(res)
    # the same batch cut into chunks of a few pairs (device memory budget of 1 MB per chunk): identical records
    import os
    os.environ["MI_NCC_CHUNK_MB"] = "1"
    try:
        chunked = crossmips.compute_displacements(tiles, ov, ov, 8, 8, 3)
    finally:
        del os.environ["MI_NCC_CHUNK_MB"]
    assert {k: (d.VHD_coords, d.NCC_widths, d.NCC_maxs) for k, d in chunked.items()} == \
           {k: (d.VHD_coords, d.NCC_widths, d.NCC_maxs) for k, d in res.items()}
    # pair sharding across ranks: the union of the shards is the whole set, no overlap
    shards = [crossmips.compute_displacements(tiles, ov, ov, 8, 8, 3, rank=k, world_size=2) for k in range(2)]
    assert set(shards[0]) | set(shards[1]) == set(res) and not (set(shards[0]) & set(shards[1]))
    assert 0.0 <= next(iter(res.values())).evalReliability(0) <= 1.0
    # tile-row blocks (the N > 1 partition of bench.py): a rank holds only its rows and the first row behind its cut
    n_rows, n_cols = len(tiles), len(tiles[0])
    for world in (2, n_rows):
        got = {}
        for r0, r1 in crossmips.tile_row_blocks(n_rows, world, n_cols):
            keep = set(range(r0, min(r1 + 1, n_rows)))
            part = [[tiles[r][c] if r in keep else None for c in range(n_cols)] for r in range(n_rows)]
            mine = crossmips.compute_displacements(part, ov, ov, 8, 8, 3, row_block=(r0, r1))
            assert not (set(mine) & set(got))
            got.update(mine)
        assert {k: (d.VHD_coords, d.NCC_widths, d.NCC_maxs, d.wRangeThrs) for k, d in got.items()} == \
               {k: (d.VHD_coords, d.NCC_widths, d.NCC_maxs, d.wRangeThrs) for k, d in res.items()}
    with pytest.raises(ValueError, match="not resident"):
        crossmips.compute_displacements([[None] * n_cols for _ in range(n_rows)], ov, ov, 8, 8, 3, row_block=(0, 1))


# ------------------------------------------------------------------ cases that sit on a tie (tests/golden/make_ncc_tie_golden.py)
def _tie_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ncc_golden_ties.npz"))


@pytest.mark.parametrize("direct", [False, True], ids=["batched_pipeline", "per_pair_path"])
@pytest.mark.parametrize("name", ["dup_0", "dup_1", "dup_2", "thr_0", "thr_1", "thr_2", "amax_0", "amax_1", "amax_2"])
def test_tie_cases_are_decided_like_the_reference(dev, name, direct, monkeypatch):
    """Duplicate maxima (several map entries equal to 1.0f: compute_MAX_ind's first strict maximum decides), an entry within
    1e-6 of widthThr * peak (the `<= thr` scans of compute_NCC_width decide on the last bits), two largest entries < 2e-6 apart.
    Such decisions lie inside the resolution of the fast map values: the pair is re-decided on entries recomputed in the
    reference's two-pass form (k_ncc_exact) -- the counters show that path was taken -- and every integer equals the
    reference's."""
    from ipp_amd import crossmips
    if direct:
        monkeypatch.setenv("MI_NCC_DIRECT", "1")
    g = _tie_golden()
    side, overlap, *dmax = (int(v) for v in g[f"{name}/recipe"])
    A = g[f"{name}/A_u8"].astype(np.float32) / np.float32(255.0)
    B = g[f"{name}/B_u8"].astype(np.float32) / np.float32(255.0)
    crossmips.ncc_stats(reset=True)
    d = crossmips.PDAlgoMIPNCC.execute(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), dmax[0], dmax[1], dmax[2], side, overlap)
    st = crossmips.ncc_stats()
    assert d.VHD_coords == list(g[f"{name}/coord"]) and d.NCC_widths == list(g[f"{name}/NCC_widths"])
    assert d.wRangeThrs == list(g[f"{name}/wRangeThr"])
    assert np.allclose(np.array(d.NCC_maxs, np.float32), g[f"{name}/NCC_maxs"], rtol=0, atol=2e-6, equal_nan=True)
    assert st["pairs_per_pair_path"] + st["pairs_batched"] == 1
    # the near-threshold entry of a thr_* case only matters when the scans of compute_NCC_width reach it (the second scan starts at
    # the width the first one found, :186-200): thr_2 is decided on it, thr_0 / thr_1 are not
    if not name.startswith("thr_") or name == "thr_2":
        assert st["pairs_per_pair_path"] == 1 and st["pairs_batched"] == 0    # handed to the careful path
        assert st["entries_recomputed_exactly"] > 0
    # the maps themselves, both routes, against the reference's
    dimk, dimi, dimj = A.shape
    ni, nj = (dimi - overlap if side == 0 else 0), (dimj - overlap if side == 1 else 0)
    mips = crossmips.compute_mips(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), ni, nj, side)
    dl = N.clamp_delays((dimk, dimi, dimj), (0, ni, nj), (dmax[2], dmax[0], dmax[1]))
    dk, di, dj = dl
    for m, (nm, du, dv) in enumerate([("xy", di, dj), ("xz", di, dk), ("yz", dj, dk)]):
        want = g[f"{name}/map_{nm}"]
        for lag in (False, True):
            got = crossmips.compute_NCC_map(mips[m], mips[m + 3], du, dv, lag=lag).cpu().numpy()
            assert np.array_equal(np.isnan(got), np.isnan(want)) and _ulps(got, want) <= MAX_ULPS, (nm, lag)


def test_layers_one_batch_ahead_equal_layer_by_layer(dev):
    """``compute_displacements_layers`` (mi_ncc_mips_batch_begin / _end: layer l + 1 enqueued before layer l is finished, the next
    layer's first MIP pass beside this layer's last lag chain) returns, layer by layer, exactly the records of one synchronous
    call per layer (StackStitcher.cpp:223-374 walks the layers in turn) -- float and 16-bit tiles, a layer without pairs and a
    pending batch that is dropped unfinished included."""
    from ipp_amd import crossmips
    rng = np.random.default_rng(5)
    tile, ov, R_, C_, L = (24, 128, 192), 40, 2, 3, 4
    field = N.bead_field((L * tile[0] + 8, R_ * (tile[1] - ov) + ov + 16, C_ * (tile[2] - ov) + ov + 16), seed=21, density=1 / 300)
    layers = []
    for layer in range(L):
        jit = rng.integers(-3, 4, size=(R_, C_, 2))
        grid = [[None] * C_ for _ in range(R_)]
        for r in range(R_):
            for c in range(C_):
                z0, y0, x0 = 4 + layer * tile[0], 8 + r * (tile[1] - ov) + jit[r, c, 0], 8 + c * (tile[2] - ov) + jit[r, c, 1]
                grid[r][c] = torch.from_numpy(np.ascontiguousarray(field[z0:z0 + tile[0], y0:y0 + tile[1], x0:x0 + tile[2]])).to(dev)
        layers.append(grid)
    for as_u16 in (False, True):
        ls = [[[(t * 65535.0).round().clamp(0, 65535).to(torch.uint16) for t in row] for row in g] for g in layers] if as_u16 else layers
        want = [crossmips.compute_displacements(g, ov, ov, 8, 8, 3) for g in ls]
        got = list(crossmips.compute_displacements_layers(ls, ov, ov, 8, 8, 3))
        assert len(got) == L
        for w, g in zip(want, got):
            assert w.keys() == g.keys() and len(w) == 2 * R_ * C_ - R_ - C_
            for k in w:
                assert w[k].VHD_coords == g[k].VHD_coords and w[k].NCC_widths == g[k].NCC_widths and w[k].wRangeThrs == g[k].wRangeThrs
                assert np.array_equal(np.array(w[k].NCC_maxs, np.float32), np.array(g[k].NCC_maxs, np.float32), equal_nan=True)
    assert list(crossmips.compute_displacements_layers([[[layers[0][0][0]]]], ov, ov, 8, 8, 3)) == [{}]   # one tile: no pair
    pending = crossmips.compute_displacements_begin(layers[0], ov, ov, 8, 8, 3)                          # never asked for its result
    del pending
    assert crossmips.compute_displacements(layers[1], ov, ov, 8, 8, 3).keys() == want[1].keys()


def test_mips_shape_sweep_is_exact(dev):
    """compute_3_MIPs (compute_funcs.cu:502-521) on views whose extents sit on and around every boundary of the MIP pass: stacks of
    1 .. 33 slices (k_mips5 takes up to 32 -- a wave owns slices w, w + 4, ...; the deep-stack pass beyond), views of 1 .. 130 rows
    (bands of 16, groups of 4 bands) and columns (blocks of 64 aligned to the TILE rows, so the first block of a west-east view is
    partial), both sides.  The six MIPs must equal numpy's maxima bit for bit (a maximum has no rounding)."""
    from ipp_amd import crossmips
    rng = np.random.default_rng(123)
    cases = []
    for dimk in (1, 2, 3, 4, 5, 8, 31, 32, 33):
        cases.append((dimk, 70, 200, 1, 77))            # west-east: 123 columns from column 77 (first block partial), 70 rows
        cases.append((dimk, 70, 200, 0, 40))            # north-south: 30 rows, 200 columns
    for rows, cols in ((1, 1), (15, 63), (16, 64), (17, 65), (64, 128), (65, 129), (130, 3)):
        cases.append((6, rows + 9, cols + 13, 1, 13))   # views of rows + 9 x cols rows / columns
        cases.append((6, rows + 9, cols + 13, 0, 9))
    for dimk, dimi, dimj, side, off in cases:
        A = rng.random((dimk, dimi, dimj), dtype=np.float32)
        B = rng.random((dimk, dimi, dimj), dtype=np.float32)
        ni, nj = (off, 0) if side == 0 else (0, off)
        got = [m.cpu().numpy() for m in crossmips.compute_mips(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), ni, nj, side)]
        va = A[:, ni:, nj:]                              # libcrossmips.cpp:296-317: the first stack's view starts at (ni, nj) ...
        vb = B[:, :dimi - ni, :dimj - nj]               # ... the second's at the origin, both of the overlap's extent
        for k, v in ((0, va), (3, vb)):
            assert np.array_equal(got[k], v.max(axis=0)), (dimk, dimi, dimj, side, "xy")
            assert np.array_equal(got[k + 1], v.max(axis=2).T), (dimk, dimi, dimj, side, "xz")
            assert np.array_equal(got[k + 2], v.max(axis=1).T), (dimk, dimi, dimj, side, "yz")
