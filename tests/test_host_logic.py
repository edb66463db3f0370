"""CPU: host-side geometry and CLI logic (no kernels are launched)."""
import json

import numpy as np
import pytest

from oracle import rl_oracle as R


def test_split_stack_matches_oracle_restatement():
    from ipp_amd import lsdeconv as L
    blk = L.Block(4, 4, 3, 3, 2, 2)
    p1, p2 = L.split_stack((10, 7, 5), blk)
    q1, q2 = R.split_stack((10, 7, 5), (4, 4, 3), (3, 2, 2))
    assert np.array_equal(p1, q1) and np.array_equal(p2, q2)


def test_load_block_symmetric_padding_and_im2single():
    from ipp_amd import lsdeconv as L
    vol = (np.arange(6 * 7 * 8) % 65536).astype(np.uint16).reshape(6, 7, 8)
    bl = L.load_block(vol, (1, 1, 1), (4, 4, 3), (2, 3, 1))       # box at the volume corner: pre-pads are mirrored
    assert bl.shape == (3 + 2, 4 + 6, 4 + 4) and bl.dtype == np.float32
    ref = np.pad(vol[0:4, 0:7, 0:6].astype(np.float32) / 65535.0, [(1, 0), (3, 0), (2, 0)], mode="symmetric")
    assert np.array_equal(bl, ref.astype(np.float32))
    inner = L.load_block(vol, (3, 3, 2), (5, 5, 4), (1, 1, 1))    # interior box: real neighbours, no mirroring
    assert np.array_equal(inner, vol[0:5, 1:6, 1:6].astype(np.float32) / np.float32(65535))


def test_pad_rules_and_autosplit():
    from ipp_amd import lsdeconv as L
    assert L.decon_pad_size((9, 9, 19)) == [9, 9, 19]
    assert L.gaussian_pad_size((0.5, 0.5, 2.5), (13, 13, 25)) == [13, 13, 25]
    f = L.Filter(use_fft=True)
    blk = L.autosplit((300, 300, 100), (9, 9, 19), f, block_size_max=200 ** 3, numit=6)
    shape = [c + 2 * p for c, p in zip((blk.x, blk.y, blk.z), (blk.x_pad, blk.y_pad, blk.z_pad))]
    smooth = tuple(R.next_fast_len(s) for s in shape)
    assert all(f >= s for f, s in zip(blk.fft_shape, shape)) and np.prod(blk.fft_shape) <= 1.3 * np.prod(smooth)
    assert np.prod(blk.fft_shape) < 200 ** 3 and (blk.x_pad, blk.y_pad, blk.z_pad) == (13, 13, 25)
    assert blk.nx * blk.x >= 300 and blk.nz * blk.z >= 100 and len(blk.p1) == blk.nx * blk.ny * blk.nz
    with pytest.raises(RuntimeError, match="No block shape fits"):
        L.autosplit((300, 300, 100), (9, 9, 19), f, block_size_max=10, numit=6)
    # the reference's score (largest core, LsDeconv.m:365) and its host-memory terms (:317-322, 359): the z slab of bricks that
    # post-processing assembles at the output type may take half of the available memory
    g = L.Filter(use_fft=False)
    free = L.autosplit((2048, 2048, 2048), (9, 9, 19), g, 300_000_000, 6)
    tight = L.autosplit((2048, 2048, 2048), (9, 9, 19), g, 300_000_000, 6, ram_available=8 << 30, output_bytes=2)
    assert tight.z <= (8 << 30) // 2 // (2 * 2048 * 2048) < free.z
    assert tight.x * tight.y * tight.z <= free.x * free.y * free.z
    for b in (free, tight):
        assert (b.x + 2 * b.x_pad) * (b.y + 2 * b.y_pad) * (b.z + 2 * b.z_pad) < 300_000_000
        # no other square-xy block of the candidate depths has a larger core
        assert b.x == b.y and (b.x + 1 + 2 * b.x_pad) ** 2 * (b.z + 2 * b.z_pad) >= 300_000_000 or b.x == 2048


def test_autosplit_fft_blocks_are_the_largest_that_fit():
    """use_fft: the padded volume is not monotone in the xy side (native vs 7-smooth grid, both staircases), so the block must be
    found by more than a bisection: for the depth autosplit chose, no larger square core fits (brute force over every side), and
    the chosen block's score (core voxels per unit of transform cost) is not beaten by any candidate of the other depths it
    looked at [LsDeconv.m:308-385: largest core under block_size_max]."""
    from ipp_amd import lsdeconv as L
    f = L.Filter(use_fft=True)
    for stack, bmax in (((700, 700, 300), 40_000_000), ((1500, 900, 120), 90_000_000), ((400, 400, 400), 9_000_000)):
        blk = L.autosplit(stack, (9, 9, 19), f, bmax, 6)
        pad = (blk.x_pad, blk.y_pad, blk.z_pad)

        def grid(core):
            shape = [c + 2 * p for c, p in zip(core, pad)]
            smooth, native = L.next_fast_len(shape), L.native_fft_shape(shape)
            return native if np.prod(native) <= 1.3 * np.prod(smooth) else smooth

        assert list(blk.fft_shape) == grid((blk.x, blk.y, blk.z)) and np.prod(blk.fft_shape) < bmax
        side = max(blk.x, blk.y)
        larger = [xy for xy in range(side + 1, max(stack[0], stack[1]) + 1)
                  if np.prod(grid((min(xy, stack[0]), min(xy, stack[1]), blk.z))) < bmax]
        # a larger side may fit only on a grid the hand-written pipeline does not take (autosplit prefers the native one by cost)
        for xy in larger:
            g = grid((min(xy, stack[0]), min(xy, stack[1]), blk.z))
            assert g != L.native_fft_shape(g) or list(blk.fft_shape) != L.native_fft_shape(list(blk.fft_shape)) or \
                min(xy, stack[0]) * min(xy, stack[1]) <= blk.x * blk.y, (stack, xy)


def test_block_fft_shape_of_remainder_blocks():
    """decwrap's per-block grid: the hand-written pipeline's extents up to 3.4x the 7-smooth grid inside the main block's grid (a
    rocFFT plan per new shape costs 0.7 s, its transforms 3.5x per point), up to 1.3x without one; the reference's grid travels as
    psf_grid."""
    from ipp_amd import lsdeconv as L
    main = (512, 512, 1024)
    assert L.block_fft_shape((512, 512, 959), main) == [512, 512, 1024]
    assert L.block_fft_shape((130, 512, 959), main) == [192, 512, 1024]        # 1.52x the 7-smooth [135, 512, 960]
    assert L.block_fft_shape((130, 130, 280), main) == [192, 160, 288]         # 1.73x
    assert L.block_fft_shape((130, 130, 280), (160, 160, 288)) == [135, 135, 280]   # would leave the main block's grid
    s, n = L.next_fast_len((66, 66, 66)), L.native_fft_shape((66, 66, 66))
    assert 3.4 < np.prod(n) / np.prod(s) < 3.5 and L.block_fft_shape((66, 66, 66), main) == s      # beyond the break-even
    assert L.block_fft_shape((130, 66, 66), main) == [192, 96, 96]                                 # 2.67x, inside the main grid
    assert L.block_fft_shape((130, 66, 66)) == [135, 70, 70]                                       # no main grid: the 1.3x rule
    blk = L.Block(10, 10, 10, 1, 1, 1, fft_shape=(192, 160, 288), psf_grid=(135, 135, 280))
    assert blk.psf_grid == (135, 135, 280)


def test_decwrap_cli_validation_and_dry_run(tmp_path, capsys):
    from ipp_amd import decwrap
    np.save(tmp_path / "vol.npy", np.zeros((4, 4, 4), np.uint16))
    base = ["-i", str(tmp_path / "vol.npy"), "-dxy", "0.4", "-dz", "1.0"]
    assert decwrap.main(base + ["-ex", "488", "-em", "525", "--dry-run"]) == 0
    cfg = json.loads(capsys.readouterr().out)["config"]
    assert cfg["numit"] == 6 and cfg["regularize_interval"] == 3 and cfg["gaussian_sigma"] == [0.5, 0.5, 2.5]
    assert cfg["gaussian_filter_size"] == [13, 13, 25] and cfg["clipval"] == 99.99 and cfg["lambda_damping"] == 0.0
    with pytest.raises(RuntimeError, match="Unsupported excitation/emission pair"):
        decwrap.main(base + ["-ex", "500", "-em", "525", "--dry-run"])
    with pytest.raises(RuntimeError, match="adaptive-psf"):
        decwrap.main(base + ["-ex", "488", "-em", "525", "--adaptive-psf", "--dry-run"])
    with pytest.raises(ValueError, match="Path does not exist"):
        decwrap.main(["-i", str(tmp_path / "nope"), "-dxy", "0.4", "-ex", "488", "-em", "525", "--dry-run"])


def test_pair_enumeration_and_layers():
    from ipp_amd import crossmips as X
    pairs = list(X.enumerate_pairs(8, 8))
    assert len(pairs) == 2 * 64 - 8 - 8  # StackStitcher.cpp:217
    assert pairs[0] == (0, 0, 0, 1, X.dir_horizontal) and pairs[1] == (0, 0, 1, 0, X.dir_vertical)
    assert X.subvolume_layers(32, 200) == [(0, 32)]
    assert X.subvolume_layers(450, 200) == [(0, 150), (150, 300), (300, 450)]
    assert X.subvolume_layers(401, 200) == [(0, 134), (134, 268), (268, 401)]  # first Z mod n layers one longer
    p = X.NCC_parms_t(25, 25, 10)
    assert (p.wRangeThr_i, p.wRangeThr_j, p.wRangeThr_k, p.INF_W) == (25, 25, 10, 26)
    assert X.NCC_parms_t(40, 40, 40).wRangeThr_i == 29 and abs(p.widthThr - 0.8) < 1e-7 and p.maxIter == 2


def test_tile_row_blocks_partition_pairs_once_and_balance():
    """crossmips.tile_row_blocks: contiguous row blocks, every pair of the grid in exactly one block, the first row behind a cut is
    the only row a second rank needs, largest block minimal for contiguous cuts (Parastitcher.py:1440-1560 farms jobs instead)."""
    from ipp_amd import crossmips as X
    for rows, cols, world in [(8, 8, 1), (8, 8, 2), (8, 8, 3), (8, 8, 8), (8, 8, 16), (5, 3, 2), (2, 7, 4), (1, 4, 3)]:
        blocks = X.tile_row_blocks(rows, world, cols)
        assert len(blocks) == world and blocks[0][0] == 0 and max(b[1] for b in blocks) == rows
        assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]) if b[0] < rows)
        pairs = list(X.enumerate_pairs(rows, cols))
        owned = [[p for p in pairs if r0 <= p[0] < r1] for r0, r1 in blocks]
        assert sorted(sum(owned, [])) == sorted(pairs)
        for (r0, r1), mine in zip(blocks, owned):
            need = {p[0] for p in mine} | {p[2] for p in mine}
            assert need <= set(range(r0, min(r1 + 1, rows)))
        # no contiguous partition into the same number of blocks has a smaller largest block
        import itertools
        cost = [(cols - 1) + (cols if r + 1 < rows else 0) for r in range(rows)]
        k = min(world, rows)
        best = min(max(sum(cost[a:b]) for a, b in zip((0,) + cut, cut + (rows,)))
                   for cut in itertools.combinations(range(1, rows), k - 1)) if rows else 0
        assert max(len(m) for m in owned) == best
    assert X.tile_row_blocks(8, 8, 8) == [(r, r + 1) for r in range(8)]


def test_fft_good_size_per_axis():
    """mi_fft_good_size: 2^a * {1,3,9} extents (x: twice such a number; z bounded by the LDS tile; y also 5 * 2^a, a in 5..8)."""
    from ipp_amd import capi
    g = capi.lib().mi_fft_good_size
    assert [g(n, 1) for n in (1, 8, 9, 33, 97, 130, 257, 289, 600, 1100, 1153)] == [8, 8, 16, 64, 128, 160, 288, 320, 640, 1152, 1280]
    assert [g(n, 0) for n in (10, 17, 70, 193, 200, 600, 2100)] == [16, 32, 128, 256, 256, 768, 2304]
    assert [g(n, 2) for n in (61, 100, 530, 2049, 2305)] == [64, 128, 576, 2304, 0]
    for axis in range(3):
        for n in range(1, 700, 7):
            m = g(n, axis)
            h = m // 2 if axis == 0 else m
            assert m >= n and any(h % r == 0 and (h // r) & (h // r - 1) == 0 for r in ((1, 3, 5, 9) if axis == 1 else (1, 3, 9)))


def _displ(coords, peaks, widths, inv=26, default=(0, 717, 0)):
    from ipp_amd import crossmips
    d = crossmips.DisplacementMIPNCC(list(coords), list(peaks), list(widths), [25, 25, 10], [25, 25, 10], [inv] * 3)
    d.VHD_def_coords = list(default)
    return d


def test_displacement_reliability_formula():
    """DisplacementMIPNCC::evalReliability (:130-147): sqrt(0.5*((100 - w*100/invW)/100)^2 + 0.5*peak^2) as float."""
    d = _displ((3, 720, -1), (0.9, 0.5, 0.0), (2, 13, 26))
    want = [np.float32(np.sqrt(0.5 * (1 - 2 / 26) ** 2 + 0.5 * 0.81)), np.float32(np.sqrt(0.5 * 0.25 + 0.5 * 0.25)), 0.0]
    got = [d.evalReliability(k) for k in range(3)]
    assert got == pytest.approx([float(w) for w in want], rel=2e-7, abs=1e-7)
    assert got[2] == 0.0 and d.rel_factors == got
    fresh = _displ((0, 0, 0), (0, 0, 0), (1, 1, 1))
    with pytest.raises(RuntimeError, match="not yet computed"):
        fresh.getReliability(0)
    nominal = __import__("ipp_amd.crossmips", fromlist=["x"]).DisplacementMIPNCC.nominal(0, 717, 0)
    assert [nominal.evalReliability(k) for k in range(3)] == [0.0, 0.0, 0.0]  # width == invWidth == 30, peak 0


def test_combine_projection_and_threshold():
    """combine (:312-346) keeps, per direction, the more reliable record in BOTH objects (ties keep the first);
    projectDisplacements (Displacement.cpp:84-106) folds the layers front to back; threshold (:217-235) resets unreliable
    directions to the default displacement with peak 0 / width invW (reliability 0)."""
    from ipp_amd import crossmips
    a = _displ((1, 715, 0), (0.95, 0.30, 0.2), (1, 20, 26))   # strong V, weak H, D unreliable
    b = _displ((4, 718, 2), (0.40, 0.90, 0.2), (15, 2, 26))   # weak V, strong H, same D as a (tie)
    c = _displ((9, 700, 5), (0.10, 0.10, 0.8), (25, 25, 3))   # only D is good
    last = crossmips.project_displacements([a, b, c])
    assert last is c and last.VHD_coords == [1, 718, 5]
    assert last.NCC_maxs == pytest.approx([0.95, 0.90, 0.8]) and last.NCC_widths == [1, 2, 3]
    assert a.VHD_coords[0] == b.VHD_coords[0] == 1           # combine made the first two equal in V
    # tie: equal reliabilities keep self (a's D record was copied into b, not the other way round)
    x, y = _displ((1, 1, 7), (0.5, 0.5, 0.5), (5, 5, 5)), _displ((2, 2, 8), (0.5, 0.5, 0.5), (5, 5, 5))
    x.combine(y)
    assert x.VHD_coords == [1, 1, 7] and y.VHD_coords == [1, 1, 7]
    with pytest.raises(ValueError, match="EMPTY"):
        crossmips.project_displacements([])
    t = _displ((3, 720, -4), (0.9, 0.2, 0.7), (2, 25, 20), default=(0, 717, 0))
    t.threshold(0.65)
    assert t.VHD_coords == [3, 717, 0]                       # H and D fell back to the stage displacement
    assert t.NCC_maxs[1:] == [0.0, 0.0] and t.NCC_widths[1:] == [26, 26] and t.rel_factors[1:] == [0.0, 0.0]
    assert t.rel_factors[0] > 0.65
    m = t.getMirrored()
    assert m.VHD_coords == [-3, -717, 0] and m.VHD_def_coords == [0, -717, 0] and m.NCC_widths == t.NCC_widths
    assert t.getMirrored(0).VHD_coords == [-3, 717, 0]
    better = _displ((0, 0, 0), (0.99, 0.99, 0.99), (1, 1, 1))
    assert t.isBetter(better) and not better.isBetter(t)


def test_displacement_xml_round_trip_and_threshold_grid(tmp_path):
    import xml.etree.ElementTree as ET
    from ipp_amd import crossmips, process_images
    recs = []
    grid = {(0, 0, 0, 1): _displ((0, 716, 1), (0.9, 0.9, 0.9), (2, 2, 2)), (0, 0, 1, 0): _displ((715, 2, 0), (0.1, 0.1, 0.1), (25, 25, 25), default=(717, 0, 0)),
            (0, 1, 1, 1): _displ((718, 0, 0), (0.8, 0.2, 0.2), (3, 20, 20), default=(717, 0, 0)), (1, 0, 1, 1): _displ((0, 717, 0), (0.3, 0.2, 0.1), (20, 22, 24))}
    for (ra, ca, rb, cb), d in grid.items():
        for k in range(3):
            d.evalReliability(k)
        recs.append(({"rowA": ra, "colA": ca, "rowB": rb, "colB": cb, "direction": "x"}, d))
    process_images.write_pairs(tmp_path / "p.xml", {"step": "3"}, recs)
    attrib, back = process_images.read_pairs(tmp_path / "p.xml")
    assert attrib["step"] == "3" and len(back) == 4
    for (_, d0), (_, d1) in zip(recs, back):
        assert d0.VHD_coords == d1.VHD_coords and d0.NCC_widths == d1.NCC_widths and d0.VHD_def_coords == d1.VHD_def_coords
        assert d1.rel_factors == d0.rel_factors and d1.NCC_maxs == pytest.approx(d0.NCC_maxs)
    old = ET.fromstring('<Displacement TYPE="MIP_NCC"><V displ="1" default_displ="0" reliability="0.5" nccPeak="0.5" nccWidth="3"/>'
                        '<H displ="2" default_displ="9" reliability="0.5" nccPeak="0.5" nccWidth="3"/>'
                        '<D displ="3" default_displ="0" reliability="0.5" nccPeak="0.5" nccWidth="3"/></Displacement>')
    d = crossmips.DisplacementMIPNCC.loadXML(old)          # records written before 2013 carry no search parameters
    assert d.wRangeThrs == [29] * 3 and d.invWidths == [30] * 3 and d.delays == [-1] * 3
    stacks = crossmips.threshold_displacements(grid, 2, 2, 0.65)
    # (1,0)'s two pairs are both unreliable in every direction; (1,1) keeps the V displacement of its northern pair
    assert stacks == {(0, 0): True, (0, 1): True, (1, 0): False, (1, 1): True}
    assert grid[(0, 0, 1, 0)].VHD_coords == [717, 0, 0] and grid[(0, 1, 1, 1)].VHD_coords == [718, 0, 0]
    del grid[(1, 0, 1, 1)]
    with pytest.raises(ValueError, match="one and only displacement"):
        crossmips.threshold_displacements(grid, 2, 2, 0.65)


def test_delta_on_ones_closed_form_matches_the_oracle():
    """The closed form the full-size config-4 test checks the device against (tests/rl_util.py) IS one deconFFT iteration of the
    oracle (decon.m:162-186) on 1 + amp * delta -- at an impulse next to the wrap-around of every axis, with an unsymmetric PSF."""
    from tests.rl_util import delta_on_ones_closed_form
    rng = np.random.default_rng(0)
    psf = rng.random((9, 7, 5)).astype(np.float32)
    psf /= psf.sum()
    shape = (32, 48, 40)
    shifts = [n // 2 - (n - k) // 2 for n, k in zip(shape, psf.shape)]
    amp = 10.0 / float(psf[tuple(shifts)])
    vol = np.ones(shape, np.float32)
    p = (30, 46, 39)
    vol[p] = 1 + amp
    out = R.decon(vol, psf, 1, 0.0, 0.0, 0, use_fft=True, fft_shape_zyx=shape, skip_edgetaper=True)
    for d in [(0, 0, 0), (1, 0, 0), (0, -1, 0), (3, -2, 1), (-4, 3, -2), (8, 6, 4), (-8, -6, -4), (5, 0, 0), (9, 0, 0), (0, 20, 0)]:
        y = tuple((a + b) % n for a, b, n in zip(p, d, shape))
        assert out[y] == pytest.approx(delta_on_ones_closed_form(psf, shifts, amp, d), rel=2e-6), d


def test_evict_cores_from_two_workers_of_one_device():
    """decwrap's out-of-memory recovery with several workers per device: two workers evict at once while the writer pool
    still holds one block's brick -- every core leaves once, no brick is written twice or by two writers, nobody raises."""
    import threading
    import time
    from concurrent.futures import ThreadPoolExecutor
    from ipp_amd import decwrap
    lock, evict_lock = threading.Lock(), threading.Lock()
    resident = {n: object() for n in range(1, 9)}
    resident_dev = {n: (1 if n <= 6 else 2) for n in resident}
    complete, writing, written = set(), set(), []

    def slow_writer(n):                      # the writer pool's save_brick of block 3
        with lock:
            assert n not in writing
            writing.add(n)
        time.sleep(0.3)
        with lock:
            writing.discard(n)
            complete.add(n)
            written.append(n)

    def write_brick(n, core):
        with lock:
            assert n not in writing and n not in complete
            writing.add(n)
        time.sleep(0.02)
        with lock:
            writing.discard(n)
            complete.add(n)
            written.append(n)

    with ThreadPoolExecutor(2) as pool:
        futures = {3: pool.submit(slow_writer, 3)}
        complete.add(5)                       # a brick that was finished long ago
        errors, counts = [], []

        def worker():
            try:
                counts.append(decwrap.evict_cores(1, resident, resident_dev, futures, lock, evict_lock, lambda n: n in complete, write_brick))
            except Exception as e:            # noqa: BLE001
                errors.append(e)
        ts = [threading.Thread(target=worker) for _ in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    assert not errors, errors
    assert sorted(counts) == [0, 6]
    assert sorted(resident) == [7, 8]                       # the other device's cores stay
    assert sorted(written) == [1, 2, 3, 4, 6]               # 3 by the writer pool (waited for), 5 was complete, none twice


def test_evict_cores_with_bricks_that_trail_behind_the_workers():
    """MI_DECWRAP_BRICKS=trail: the brick of a resident core is a job of a one-thread pool that may not have begun, may be running, or
    may have been cancelled when the core is evicted -- a cancelled or never-written brick is written by the eviction, a finished one
    is left alone."""
    import threading
    import time
    from concurrent.futures import ThreadPoolExecutor
    from ipp_amd import decwrap
    lock, evict_lock = threading.Lock(), threading.Lock()
    resident = {n: object() for n in range(1, 5)}
    resident_dev = {n: 1 for n in resident}
    complete, written = set(), []

    def trail(n, wait):                      # trail_brick: gives up when its core has left `resident`
        time.sleep(wait)
        with lock:
            if n not in resident:
                return False
            complete.add(n)
            written.append(("trail", n))
        return True

    def write_brick(n, core):
        with lock:
            assert n not in complete
            complete.add(n)
            written.append(("evict", n))

    pool = ThreadPoolExecutor(1)
    futures = {1: pool.submit(trail, 1, 0.0)}
    futures[1].result()                                      # block 1: its brick was written behind the workers
    futures[2] = pool.submit(trail, 2, 0.3)                  # block 2: being looked at by the writer while the eviction pops it
    futures[3] = pool.submit(trail, 3, 0.0)                  # block 3: queued behind it
    futures[4] = pool.submit(trail, 4, 0.0)
    assert futures[4].cancel()                               # block 4: never begun (the blocks phase ended)
    assert decwrap.evict_cores(1, resident, resident_dev, futures, lock, evict_lock, lambda n: n in complete, write_brick) == 4
    pool.shutdown()
    assert not resident and sorted(complete) == [1, 2, 3, 4]
    assert ("trail", 1) in written and ("evict", 4) in written and len(written) == 4   # every brick exactly once
