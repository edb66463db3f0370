"""GPU: the preserved entry points (decwrap.py / process_images.py step 2) end to end on tiny inputs, checked
against the oracle composition of the same steps."""
import xml.etree.ElementTree as ET

import numpy as np
import pytest
import torch

from oracle import ncc_oracle as N
from oracle import rl_oracle as R

pytestmark = pytest.mark.gpu


def test_decwrap_end_to_end(dev, tmp_path):
    from ipp_amd import decwrap, psf as P
    rng = np.random.default_rng(8)
    vol16 = (rng.random((20, 28, 30)) * 4000 + 500).astype(np.uint16)
    np.save(tmp_path / "vol.npy", vol16)
    rc = decwrap.main(["-i", str(tmp_path / "vol.npy"), "-dxy", "0.422", "-dz", "1.0", "-ex", "488", "-em", "525", "-it", "4",
                       "--gaussian-sigma", "0.5", "0.5", "1.0", "--gaussian-filter-size", "3", "3", "5",
                       "--regularize-interval", "2", "--gpu-indices", "1"])
    assert rc == 0
    got = np.load(tmp_path / "deconvolved" / "deconvolved.npy")
    assert (tmp_path / "deconvolved" / "deconvolution_config.json").exists()
    # oracle composition: symmetric-padded block -> Gaussian pre-filter -> deconSpatial -> strip pads
    psf = P.LsMakePSF(422.0, 1000.0, 0.40, 1.42, 488.0, 525.0, 240.0, 12.0)
    pad = [max(k, g) for k, g in zip(psf.shape, (5, 3, 3))]
    bl = np.pad(vol16.astype(np.float32) / np.float32(65535), [(p, p) for p in pad], mode="symmetric")
    bl = R.gauss3d(bl, [0.5, 0.5, 1.0], [3, 3, 5])
    want = R.decon_spatial(bl, psf, 4, 0.0, 0.0, 2)
    want = want[pad[0]:-pad[0], pad[1]:-pad[1], pad[2]:-pad[2]]
    assert got.shape == vol16.shape
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-4
    # postprocess: clip range = percentiles of the whole padded block (one block here), uint16 input -> 16-bit output
    import json
    mm = json.load(open(tmp_path / "deconvolved" / "min_max.json"))
    padded = R.decon_spatial(bl, psf, 4, 0.0, 0.0, 2)
    wl, wu = R.prctile(padded, [0.01, 99.99])
    assert mm["deconvmin"] == pytest.approx(float(wl), rel=1e-3) and mm["deconvmax"] == pytest.approx(float(wu), rel=1e-3)
    assert mm["rawmax"] == 65535 and mm["scal"] == 65535
    q = np.load(tmp_path / "deconvolved" / "deconvolved_16bit.npy")
    assert q.dtype == np.uint16 and q.shape == vol16.shape
    assert np.array_equal(q, R.rescale_block(got, 65535, 1.0, mm["deconvmin"], mm["deconvmax"], np.uint16))


def test_process_images_step2(dev, tmp_path):
    from ipp_amd import process_images
    tile, ov = (26, 96, 96), 32
    step = tile[1] - ov
    field = N.bead_field((tile[0], 2 * step + ov, 2 * step + ov), seed=4, density=1 / 300)
    tiles = {}
    for r in range(2):
        for c in range(2):
            t = field[:, r * step:r * step + tile[1], c * step:c * step + tile[2]]
            q = np.clip(np.rint(t * 255), 0, 255).astype(np.uint8)
            np.save(tmp_path / f"tile_{r}_{c}.npy", q)
            tiles[(r, c)] = q.astype(np.float32) / np.float32(255)
    assert process_images.main(["-2", "--input", str(tmp_path), "--oV", str(ov), "--oH", str(ov), "--sV", "6", "--sH", "6",
                                "--sD", "1"]) == 0
    root = ET.parse(tmp_path / "xml_displcomp.xml").getroot()
    pairs = root.findall("Pair")
    assert len(pairs) == 4
    for p in pairs:
        a = tiles[(int(p.get("rowA")), int(p.get("colA")))]
        b = tiles[(int(p.get("rowB")), int(p.get("colB")))]
        side = 0 if p.get("direction") == "NORTH_SOUTH" else 1
        want = N.pdalgo_execute(a, b, 6, 6, 1, side, ov, kind="oracle")
        d = p.find("Displacement")
        assert d.get("TYPE") == "MIP_NCC"
        assert int(d.find("V").get("default_displ")) == (step if side == 0 else 0)
        assert int(d.find("H").get("default_displ")) == (step if side == 1 else 0)
        for i, name in enumerate("VHD"):
            e = d.find(name)
            assert int(e.get("displ")) == want["coord"][i] and int(e.get("nccWidth")) == want["NCC_widths"][i]
            assert int(e.get("nccWRangeThr")) == want["wRangeThr"][i] and int(e.get("nccInvWidth")) == want["INF_W"]


def test_process_images_steps_3_and_4(dev, tmp_path):
    """Step 2 on two z-layers, then projection (one record per pair) and thresholding of the XML (host logic of
    StackStitcher::projectDisplacements / thresholdDisplacements)."""
    from ipp_amd import process_images
    tile, ov = (52, 96, 96), 32
    step = tile[1] - ov
    field = N.bead_field((tile[0], 2 * step + ov, 2 * step + ov), seed=6, density=1 / 300)
    for r in range(2):
        for c in range(2):
            t = field[:, r * step:r * step + tile[1], c * step:c * step + tile[2]]
            np.save(tmp_path / f"tile_{r}_{c}.npy", np.clip(np.rint(t * 255), 0, 255).astype(np.uint8))
    common = ["--input", str(tmp_path)]
    assert process_images.main(["-2", *common, "--oV", str(ov), "--oH", str(ov), "--sV", "6", "--sH", "6", "--sD", "1",
                                "--subvoldim", "26"]) == 0
    comp = ET.parse(tmp_path / "xml_displcomp.xml").getroot().findall("Pair")
    assert len(comp) == 8 and {p.get("layer") for p in comp} == {"0", "1"}
    assert process_images.main(["-3", *common]) == 0
    proj = ET.parse(tmp_path / "xml_displproj.xml").getroot().findall("Pair")
    assert len(proj) == 4 and all(p.get("layer") is None for p in proj)
    for p in proj:  # per direction the projected reliability is the maximum over the pair's layers
        key = [p.get(k) for k in ("rowA", "colA", "rowB", "colB")]
        layers = [q for q in comp if [q.get(k) for k in ("rowA", "colA", "rowB", "colB")] == key]
        for ax in "VHD":
            best = max(float(q.find("Displacement").find(ax).get("reliability")) for q in layers)
            assert float(p.find("Displacement").find(ax).get("reliability")) == best
    assert process_images.main(["-4", *common, "--threshold", "0.65"]) == 0
    root = ET.parse(tmp_path / "xml_displthres.xml").getroot()
    assert root.get("step") == "4" and len(root.findall("Stack")) == 4
    for p in root.findall("Pair"):
        for ax in "VHD":
            e = p.find("Displacement").find(ax)
            rel = float(e.get("reliability"))
            assert rel >= 0.65 or (rel == 0.0 and e.get("displ") == e.get("default_displ") and float(e.get("nccPeak")) == 0.0)
    # exact copies of one bead field: the in-plane offsets are reliable, every stack is stitchable
    assert all(s.get("stitchable") == "yes" for s in root.findall("Stack"))


def test_process_images_on_a_terastitcher_project(dev, tmp_path):
    """Steps 2 -> 3 -> 4 as drop-ins for ``terastitcher -2/-3/-4``: a project file in (xml_import), TIFF tiles from its
    stacks_dir, project files out.  Every per-layer record equals the oracle's PDAlgoMIPNCC::execute on the same slices."""
    from PIL import Image
    from ipp_amd import process_images, tsproject
    tile, ov, slices = (96, 96), 32, 40
    step = tile[0] - ov
    field = N.bead_field((slices, 2 * step + ov, 2 * step + ov), seed=9, density=1 / 300)
    proj = tsproject.Project(tmp_path / "tiles", 2, 2, slices, VXL=(0.8, 0.8, 2.0), MEC=(step * 0.8, step * 0.8))
    tiles = {}
    for r in range(2):
        for c in range(2):
            name = f"{r:03d}/{r:03d}_{c:03d}"
            (tmp_path / "tiles" / name).mkdir(parents=True)
            # a real shift on the east tiles so that the displacement differs from the stage offset
            dv, dh = (2, -3) if c == 1 else (0, 0)
            t = np.roll(field, (dv, dh), axis=(1, 2))[:, r * step:r * step + tile[0], c * step:c * step + tile[1]]
            q = np.clip(np.rint(t * 65535), 0, 65535).astype(np.uint16)
            for z in range(slices):
                Image.fromarray(q[z]).save(tmp_path / "tiles" / name / f"{z:06d}.tif")
            tiles[(r, c)] = q.astype(np.float32) / np.float32(65535)
            proj.STACKS[r][c] = tsproject.Stack(r, c, name, ABS_V=r * step, ABS_H=c * step, N_BYTESxCHAN=2, z_ranges=[(0, slices)])
    x1, x2, x3, x4 = (tmp_path / f"xml_import_step_{k}.xml" for k in (1, 2, 3, 4))
    proj.save(x1)
    on_dev = proj.loadImageStackDevice(proj.STACKS[1][0], 3, 9, dev)            # same bits as the host conversion
    assert np.array_equal(on_dev.cpu().numpy(), proj.loadImageStack(proj.STACKS[1][0], 3, 9))
    assert process_images.main(["-2", "--sV", "6", "--sH", "6", "--sD", "1", "--subvoldim", "20", "--threshold", "0.65",
                                f"--projin={x1}", f"--projout={x2}"]) == 0
    comp = tsproject.Project.load(x2)
    assert comp.getOVERLAP_V() == ov and comp.getDEFAULT_DISPLACEMENT_H() == step
    layers = [(0, 20), (20, 40)]
    for (r, c, rb, cb, side_name, side) in [(0, 0, 0, 1, "EAST", 1), (1, 0, 1, 1, "EAST", 1), (0, 0, 1, 0, "SOUTH", 0), (0, 1, 1, 1, "SOUTH", 0)]:
        recs = getattr(comp.STACKS[r][c], side_name)
        assert len(recs) == 2
        for (z0, z1), d in zip(layers, recs):
            want = N.pdalgo_execute(tiles[(r, c)][z0:z1], tiles[(rb, cb)][z0:z1], 6, 6, 1, side, ov, kind="oracle")
            assert d.VHD_coords == list(want["coord"]) and d.NCC_widths == list(want["NCC_widths"])
            assert d.NCC_maxs == pytest.approx([float(v) for v in want["NCC_maxs"]], rel=2e-5)
            assert d.VHD_def_coords == ([step, 0, 0] if side == 0 else [0, step, 0])
        mirrored = getattr(comp.STACKS[rb][cb], "WEST" if side else "NORTH")
        assert [m.VHD_coords for m in mirrored] == [[-v for v in d.VHD_coords] for d in recs]
    found = comp.STACKS[0][0].EAST[0].VHD_coords[:2]
    assert found != [0, step] and abs(found[0]) == 2 and abs(found[1] - step) == 3   # the shift of the east tiles was found
    assert process_images.main(["-3", f"--projin={x2}", f"--projout={x3}"]) == 0
    assert process_images.main(["-4", "--threshold", "0.65", f"--projin={x3}", f"--projout={x4}"]) == 0
    done = tsproject.Project.load(x4)
    for row in done.STACKS:
        for s in row:
            assert s.stitchable
            for lst, n in ((s.NORTH, s.ROW_INDEX == 1), (s.SOUTH, s.ROW_INDEX == 0), (s.WEST, s.COL_INDEX == 1), (s.EAST, s.COL_INDEX == 0)):
                assert len(lst) == int(n)
    e = done.STACKS[0][0].EAST[0]
    assert e.VHD_coords[:2] == found and min(e.rel_factors[:2]) >= 0.65


def test_decwrap_block_parallel_workers_equal_sequential(dev, tmp_path):
    """Several blocks, two workers on one device (--gpu-workers-per-gpu 2: own streams, shared output volume) against one
    worker: identical stacks -- the blocks are independent (LsDeconv.m:620-668)."""
    from ipp_amd import decwrap
    rng = np.random.default_rng(11)
    vol16 = (rng.random((24, 40, 44)) * 3000 + 200).astype(np.uint16)
    outs = []
    for k, workers in enumerate((1, 2)):
        d = tmp_path / f"w{workers}"
        d.mkdir()
        np.save(d / "vol.npy", vol16)
        rc = decwrap.main(["-i", str(d / "vol.npy"), "-dxy", "0.422", "-dz", "1.0", "-ex", "488", "-em", "525", "-it", "3",
                           "--gaussian-sigma", "0", "0", "0", "--block-size-max", "60000", "--gpu-indices", "1",
                           "--gpu-workers-per-gpu", str(workers)])
        assert rc == 0
        outs.append((np.load(d / "deconvolved" / "deconvolved.npy"), np.load(d / "deconvolved" / "deconvolved_16bit.npy")))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert float(np.abs(outs[0][0]).max()) > 0


def test_decwrap_resident_cores_equal_the_brick_route(dev, tmp_path, monkeypatch):
    """The assembly takes finished cores straight from device memory (default; their bricks are written behind the workers as far as
    the writers get: MI_DECWRAP_BRICKS=trail), from their LZ4 bricks (MI_DECWRAP_RESIDENT=0, the reference's route:
    LsDeconv.m:799-806, load_slab_lz4.cpp), from device memory with every brick complete before its worker goes on
    (MI_DECWRAP_BRICKS=1), or -- MI_DECWRAP_BRICKS=0 -- with no brick written at all: four identical stacks."""
    from ipp_amd import decwrap
    rng = np.random.default_rng(12)
    vol16 = (rng.random((24, 40, 44)) * 3000 + 200).astype(np.uint16)
    outs = []
    for k, env in enumerate(({}, {"MI_DECWRAP_RESIDENT": "0"}, {"MI_DECWRAP_BRICKS": "0"}, {"MI_DECWRAP_BRICKS": "1"})):
        for name in ("MI_DECWRAP_RESIDENT", "MI_DECWRAP_BRICKS"):
            monkeypatch.delenv(name, raising=False)
        for name, v in env.items():
            monkeypatch.setenv(name, v)
        d = tmp_path / f"m{k}"
        d.mkdir()
        np.save(d / "vol.npy", vol16)
        rc = decwrap.main(["-i", str(d / "vol.npy"), "-dxy", "0.422", "-dz", "1.0", "-ex", "488", "-em", "525", "-it", "3",
                           "--gaussian-sigma", "0", "0", "0", "--block-size-max", "60000", "--gpu-indices", "1",
                           "--gpu-workers-per-gpu", "2"])
        assert rc == 0
        outs.append((np.load(d / "deconvolved" / "deconvolved.npy"), np.load(d / "deconvolved" / "deconvolved_16bit.npy")))
        tm = decwrap.main.last_timing
        n_dev = tm["cores_from_device"]
        assert (n_dev == 0) if env.get("MI_DECWRAP_RESIDENT") == "0" else (n_dev > 1), (env, n_dev)
        if not env:      # every resident core's brick was either written behind the workers or never begun
            assert tm["bricks_trailed"] + tm["bricks_not_written"] == n_dev and tm["bricks_trailed"] >= 0
        else:
            assert tm["bricks_trailed"] == 0 and tm["bricks_not_written"] == 0
    for o in outs[1:]:
        assert np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1])


def test_device_memory_pool_reuses_and_releases(dev):
    """Scratch memory a call releases stays in the library's per-device pool and is handed out again; it goes back to the driver
    on request (mi_release_cached_memory)."""
    import os
    if os.environ.get("MI_NO_MEMORY_POOL"):
        pytest.skip("the pool is switched off")
    from ipp_amd import capi, decon
    capi.release_cached_memory()
    assert capi.lib().mi_cached_memory_bytes() == 0
    psf = R.gaussian_psf((5, 5, 5), (1.0, 1.0, 1.0))
    shape = (64, 128, 128)
    ctx = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
    held = ctx.device_bytes
    assert held > 2 * 4 * 64 * 128 * 128          # the two spectrum buffers at least
    ctx.close()
    cached = capi.lib().mi_cached_memory_bytes()
    assert cached >= held // 2                        # the big buffers are kept (blocks under 1 MiB are not)
    ctx2 = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
    assert capi.lib().mi_cached_memory_bytes() < cached  # ... and reused by the next context of that shape
    bl = torch.rand(shape, device=dev) + 0.5
    ctx2.iterate(bl, None, 2)
    assert bool(torch.isfinite(bl).all())
    ctx2.close()
    assert capi.release_cached_memory() >= cached and capi.lib().mi_cached_memory_bytes() == 0


def test_decwrap_tiff_folder_cache_and_resume(dev, tmp_path):
    """A folder of TIFF slices in, img_%06d.tif out; a brick found in the cache folder is taken instead of recomputing its
    block (resume, LsDeconv.m:695-705), --no-resume starts over; the cache is removed after a complete run."""
    import json
    from ipp_amd import brickio, decwrap
    rng = np.random.default_rng(4)
    vol16 = (rng.random((12, 40, 44)) * 3000 + 200).astype(np.uint16)
    src = tmp_path / "stack"
    brickio.save_tiff_series(src, vol16)
    base = ["-i", str(src), "-dxy", "0.422", "-dz", "1.0", "-ex", "488", "-em", "525", "-it", "2", "--gaussian-sigma", "0", "0", "0",
            "--block-size-max", "60000", "--gpu-indices", "1"]
    assert decwrap.main(base) == 0
    out = src / "deconvolved"
    ref = np.load(out / "deconvolved.npy")
    tif = brickio.load_tiff_series(out)
    assert tif.dtype == np.uint16 and tif.shape == vol16.shape and np.array_equal(tif, np.load(out / "deconvolved_16bit.npy"))
    assert not (out / "cache").exists()
    # an earlier, interrupted run left block 1 in the cache (here: a brick of a recognisable constant)
    for f in out.glob("img_*.tif"):
        f.unlink()
    (out / "cache").mkdir()
    first = np.load(out / "deconvolved.npy")
    nz = np.nonzero(first)  # shape of block 1's core: read it off a dry computation of the split instead of guessing
    from ipp_amd import lsdeconv as L, psf as P
    psf = P.LsMakePSF(422.0, 1000.0, 0.40, 1.42, 488.0, 525.0, 240.0, 12.0)
    blk = L.autosplit((44, 40, 12), psf.shape[::-1], L.Filter((0, 0, 0), (13, 13, 25), 0.0, 0.0, 3, False, False), 60000, 2)
    assert len(blk.p1) > 1
    p1, p2 = blk.p1[0], blk.p2[0]
    core_shape = (p2[2] - p1[2] + 1, p2[1] - p1[1] + 1, p2[0] - p1[0] + 1)
    brickio.save_lz4(out / "cache" / "bl_1.lz4", np.full(core_shape, 0.125, np.float32))
    with open(out / "cache" / "bl_1.json", "w") as f:
        json.dump({"lb": 0.1, "ub": 0.2}, f)
    assert decwrap.main(base) == 0
    got = np.load(out / "deconvolved.npy")
    box = (slice(p1[2] - 1, p2[2]), slice(p1[1] - 1, p2[1]), slice(p1[0] - 1, p2[0]))
    assert np.all(got[box] == 0.125)                                  # taken from the cache ...
    mask = np.ones(got.shape, bool)
    mask[box] = False
    assert np.array_equal(got[mask], ref[mask])                       # ... the other blocks recomputed, identical
    assert decwrap.main(base + ["--no-resume"]) == 0
    assert np.array_equal(np.load(out / "deconvolved.npy"), ref)


def test_decwrap_tiff_slices_from_the_device_slab(dev, tmp_path, monkeypatch):
    """A TIFF folder that gets its slices and nothing else (MI_DECWRAP_NPY=0: what every stack of more than 512^3 voxels gets): the
    integer slab is assembled on the device and its slices are deflated there (mi_tiff_write_series_device) -- the same stack as
    through the host (MI_DECWRAP_TIFF_DEVICE=0), with and without --flip, read back by libtiff."""
    from PIL import Image
    from ipp_amd import brickio, decwrap
    rng = np.random.default_rng(21)
    vol16 = (rng.random((20, 70, 90)) * 3000 + 200).astype(np.uint16)
    monkeypatch.setenv("MI_DECWRAP_NPY", "0")
    stacks = {}
    for name, env, extra in (("device", {}, []), ("host", {"MI_DECWRAP_TIFF_DEVICE": "0"}, []), ("device_flip", {}, ["--flip"]),
                             ("host_flip", {"MI_DECWRAP_TIFF_DEVICE": "0"}, ["--flip"])):
        monkeypatch.delenv("MI_DECWRAP_TIFF_DEVICE", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        src = tmp_path / name
        brickio.save_tiff_series(src, vol16)
        assert decwrap.main(["-i", str(src), "-dxy", "0.422", "-dz", "1.0", "-ex", "488", "-em", "525", "-it", "2", "--gaussian-sigma", "0", "0", "0",
                             "--block-size-max", "60000", "--gpu-indices", "1", "--use-fft"] + extra) == 0
        out = src / ("deconvolved_flipped_upside_down" if extra else "deconvolved")
        files = brickio.list_tiff_series(out)
        assert len(files) == 20 and not (out / "deconvolved.npy").exists() and not (out / "cache").exists()
        stacks[name] = np.stack([np.asarray(Image.open(f)) for f in files])
        assert stacks[name].dtype == np.uint16 and np.array_equal(brickio.load_tiff_series(out), stacks[name])
    assert np.array_equal(stacks["device"], stacks["host"]) and np.array_equal(stacks["device_flip"], stacks["host_flip"])
    assert np.array_equal(stacks["device_flip"], stacks["device"][:, ::-1]) and int(stacks["device"].max()) > 0


def test_decwrap_flip_and_destripe_options(dev, tmp_path):
    """--flip writes into deconvolved_flipped_upside_down with every slice mirrored along y (LsDeconv.m:91-94, 1097-1099);
    --destripe-sigma runs filter_subband_3d_z after the deconvolution of a block (LsDeconv.m:934-936)."""
    from ipp_amd import brickio, decwrap
    rng = np.random.default_rng(14)
    vol16 = (rng.random((40, 36, 48)) * 3000 + 200).astype(np.uint16)
    src = tmp_path / "stack"
    brickio.save_tiff_series(src, vol16)
    base = ["-i", str(src), "-dxy", "0.422", "-dz", "1.0", "-ex", "488", "-em", "525", "-it", "2", "--gaussian-sigma", "0", "0", "0",
            "--gpu-indices", "1"]
    assert decwrap.main(base) == 0
    assert decwrap.main(base + ["--flip"]) == 0
    plain = brickio.load_tiff_series(src / "deconvolved")
    flipped = brickio.load_tiff_series(src / "deconvolved_flipped_upside_down")
    assert np.array_equal(flipped, plain[:, ::-1, :])
    assert np.array_equal(np.load(src / "deconvolved_flipped_upside_down" / "deconvolved.npy"), np.load(src / "deconvolved" / "deconvolved.npy"))
    # destripe changes the float result, and what it changes it by is the oracle's filter on the padded block
    (src / "deconvolved").rename(src / "plain")
    assert decwrap.main(base + ["--destripe-sigma", "2.0"]) == 0
    a, b = np.load(src / "plain" / "deconvolved.npy"), np.load(src / "deconvolved" / "deconvolved.npy")
    assert a.shape == b.shape and not np.allclose(a, b, rtol=1e-3)


def test_entry_points_in_a_fresh_process(dev, tmp_path):
    """``python decwrap.py ...`` / ``python process_images.py ...`` as their own processes: nothing has initialised torch or the
    HIP runtime before the entry point does (the library must come up after torch's bundled runtime, see capi.lib)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "image-preprocessing-pipeline_amd")
    rng = np.random.default_rng(3)
    np.save(tmp_path / "vol.npy", (rng.random((16, 24, 28)) * 3000 + 100).astype(np.uint16))
    r = subprocess.run([sys.executable, os.path.join(pkg, "decwrap.py"), "-i", str(tmp_path / "vol.npy"), "-dxy", "0.422", "-dz", "1.0",
                        "-ex", "488", "-em", "525", "-it", "2", "--gaussian-sigma", "0", "0", "0", "--gpu-indices", "1",
                        "--block-size-max", "50000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert (tmp_path / "deconvolved" / "deconvolved.npy").exists()
    tile, ov = (26, 64, 64), 24
    field = N.bead_field((tile[0], 2 * (tile[1] - ov) + ov, tile[2]), seed=8, density=1 / 300)
    for rr in range(2):
        t = field[:, rr * (tile[1] - ov):rr * (tile[1] - ov) + tile[1], :]
        np.save(tmp_path / f"tile_{rr}_0.npy", np.clip(np.rint(t * 255), 0, 255).astype(np.uint8))
    r = subprocess.run([sys.executable, os.path.join(pkg, "process_images.py"), "-2", "--input", str(tmp_path), "--oV", str(ov), "--oH", str(ov),
                        "--sV", "5", "--sH", "5", "--sD", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert (tmp_path / "xml_displcomp.xml").exists()


@pytest.mark.parametrize("extra", [[], ["--transport", "peer", "--zchunks", "2"]], ids=["send_recv", "peer_copy_z_chunked"])
def test_bench_two_ranks_on_one_gpu_prints_the_contract_line(dev, extra):
    """``python bench.py --gpus 2`` from a plain shell: it fans its two ranks out itself (gloo rendezvous on 127.0.0.1, both ranks on
    the one GPU of the test box -- a rehearsal of the driver's N > 1 launch), runs the slab-sharded RL loop and the NCC leg on tile-row
    blocks, and rank 0 prints ONE JSON line with the contract's keys, ``roofline`` and ``ncc`` at N = 2."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MI_NCC_GRID="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--workload", "c2",
                        "--steps", "3", "--warmup", "1"] + extra, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0 and d["scaling"] == "strong" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "y-slabs x2" in d["config"]["parallelism"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1 and rf["traffic"] is None
    ncc = d["ncc"]
    assert ncc["n_gpus"] == 2 and ncc["value"] > 0 and ncc["pairs_with_exact_VH_offsets"] == "4/4"
    assert ncc["partition"]["row_blocks"] == [[0, 1], [1, 2]] and "roofline" in ncc and "cpu_baseline" not in d


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_load_block_on_the_device_is_bit_identical(dev, dtype):
    """mi_load_block (conversion like im2single + padarray 'symmetric' on the device) against lsdeconv.load_block on the host:
    interior blocks, blocks at every face of the volume, pads larger than what the volume offers."""
    from ipp_amd import lsdeconv as L
    rng = np.random.default_rng(17)
    vol = (rng.random((20, 26, 30)) * (250 if dtype == np.uint8 else 60000)).astype(dtype)
    staging = {}
    for p1, p2, pad in (((1, 1, 1), (30, 26, 20), (5, 4, 3)), ((9, 7, 5), (20, 18, 14), (4, 4, 4)), ((1, 9, 1), (12, 26, 8), (6, 2, 9)),
                        ((19, 1, 13), (30, 10, 20), (7, 7, 7)), ((11, 11, 11), (12, 12, 12), (25, 3, 22))):
        want = L.load_block(vol, p1, p2, pad)
        got = L.load_block_device(vol, p1, p2, pad, dev, staging)
        assert got.dtype == torch.float32 and tuple(got.shape) == want.shape
        assert np.array_equal(got.cpu().numpy(), want), (p1, p2, pad)


def test_decwrap_start_block_claims_and_brick_validation(dev, tmp_path):
    """Block streaming at the reference's semantics: a helper started with --start-block 3 works from block 3 on and leaves
    the output alone (decwrap.py:317-321, LsDeconv.m:579-583); the run with --start-block 1 takes what is missing, assembles the
    slabs and equals a single run; a brick whose header does not fit its block (wrong shape) is recomputed, a block.json of
    another stack is refused (LsDeconv.m:176-193)."""
    import json
    from ipp_amd import brickio, decwrap
    rng = np.random.default_rng(21)
    vol16 = (rng.random((24, 40, 44)) * 3000 + 200).astype(np.uint16)
    common = ["-dxy", "0.422", "-dz", "1.0", "-ex", "488", "-em", "525", "-it", "2", "--gaussian-sigma", "0", "0", "0",
              "--block-size-max", "60000", "--gpu-indices", "1"]
    ref_dir = tmp_path / "ref"
    ref_dir.mkdir()
    np.save(ref_dir / "vol.npy", vol16)
    assert decwrap.main(["-i", str(ref_dir / "vol.npy")] + common) == 0
    want = np.load(ref_dir / "deconvolved" / "deconvolved.npy")
    want16 = np.load(ref_dir / "deconvolved" / "deconvolved_16bit.npy")

    d = tmp_path / "shared"
    d.mkdir()
    np.save(d / "vol.npy", vol16)
    base = ["-i", str(d / "vol.npy")] + common
    assert decwrap.main(base + ["--start-block", "3"]) == 0            # helper: bricks only
    cache = d / "deconvolved" / "cache"
    geom = json.load(open(cache / "block.json"))
    n_blocks = geom["block"]["nx"] * geom["block"]["ny"] * geom["block"]["nz"]
    assert n_blocks >= 4
    done = sorted(int(p.stem.split("_")[1]) for p in cache.glob("bl_*.lz4") if p.stat().st_size > 0)
    assert done == list(range(3, n_blocks + 1))
    assert not (d / "deconvolved" / "deconvolved.npy").exists()
    # a brick of the wrong shape and a stale claim (empty file) of a process that died
    brickio.save_lz4(cache / "bl_3.lz4", np.zeros((2, 2, 2), np.float32))
    (cache / "bl_1.lz4").touch()
    assert decwrap.main(base) == 0                                      # master: blocks 1, 2 and 3, then the output
    assert np.array_equal(np.load(d / "deconvolved" / "deconvolved.npy"), want)
    assert np.array_equal(np.load(d / "deconvolved" / "deconvolved_16bit.npy"), want16)
    assert not cache.exists()
    # block.json of another stack: refused
    cache.mkdir()
    geom["stack_info"]["x"] += 1
    json.dump(geom, open(cache / "block.json", "w"))
    with pytest.raises(RuntimeError, match="does not match current stack_info"):
        decwrap.main(base)
