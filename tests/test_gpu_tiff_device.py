"""GPU: TIFF slices deflated on the device (mi_tiff_write_series_device: one dynamic-Huffman block per strip, no string matching) --
the files are read back by libtiff (Pillow) and by the library's own reader and must hold exactly the samples of the device volume."""
import zlib

import numpy as np
import pytest
import torch

from ipp_amd import brickio

pytestmark = pytest.mark.gpu


def _volumes():
    rng = np.random.default_rng(11)
    smooth = (np.cumsum(rng.standard_normal((3, 300, 1100)), axis=2) * 30 + 20000).clip(0, 65535).astype(np.uint16)
    yield "smooth_u16_several_strips", smooth
    yield "noise_u16_odd_extents", (rng.random((2, 257, 513)) * 65535).astype(np.uint16)
    yield "uniform_bytes_u8", rng.integers(0, 256, size=(2, 1200, 1001), dtype=np.uint8)
    yield "constant_u16", np.full((2, 64, 70), 1234, np.uint16)
    yield "zeros_u8_one_row", np.zeros((1, 1, 5), np.uint8)
    yield "two_values_u8", (rng.random((1, 90, 33)) > 0.999).astype(np.uint8) * 255
    yield "float32", rng.standard_normal((2, 200, 333)).astype(np.float32)
    skew = np.zeros((1, 700, 800), np.uint8)              # frequencies spread over 20 octaves: codes longer than 15 bits before limiting
    flat = skew.reshape(-1)
    at = 0
    for v in range(40):
        n = max(1, flat.size >> (v + 1))
        flat[at:at + n] = v
        at += n
    yield "skewed_u8_length_limited", skew
    yield "large_u16_2048", (rng.random((2, 2048, 2048)) * 3000 + 200).astype(np.uint16)


@pytest.mark.parametrize("name,vol", list(_volumes()), ids=[n for n, _ in _volumes()])
def test_slices_deflated_on_the_device_read_back_exactly(dev, tmp_path, name, vol):
    from PIL import Image
    t = torch.from_numpy(vol).to(dev)
    assert brickio.save_tiff_series_device(tmp_path / "d", t, first_index=5) == vol.shape[0]
    files = brickio.list_tiff_series(tmp_path / "d")
    assert files[0].name == "img_000005.tif" and not list((tmp_path / "d").glob("*.tmp"))
    back = np.stack([np.asarray(Image.open(f)) for f in files])                      # libtiff's inflater
    assert back.dtype == vol.dtype and np.array_equal(back, vol)
    assert np.array_equal(brickio.load_tiff_series(tmp_path / "d"), vol)            # the library's reader (libdeflate or zlib)
    shape, dt, fast = brickio.tiff_info(files[0])
    assert shape == vol.shape[1:] and dt == vol.dtype and fast
    # the same strips through Python's zlib, and their size against the host writer's (no string matching: a little larger at most on
    # samples like these, never larger than stored + the block header)
    im = Image.open(files[0])
    offs, cnts = im.tag_v2[273], im.tag_v2[279]
    raw = files[0].read_bytes()
    plain = b"".join(zlib.decompress(raw[o:o + c]) for o, c in zip(offs, cnts))
    if im.tag_v2.get(317, 1) == 1:
        assert plain == vol[0].tobytes()
    else:                                          # stored horizontally differenced: undo it per row
        d = np.frombuffer(plain, vol.dtype).reshape(vol.shape[1:])
        assert np.array_equal(np.cumsum(d, axis=1, dtype=vol.dtype), vol[0])
    assert sum(cnts) <= vol[0].nbytes * 1.002 + 400 * len(cnts)
    assert brickio.save_tiff_series_device(tmp_path / "d", torch.zeros_like(t), first_index=5) == 0      # existing slices are kept


def test_device_writer_against_the_host_writer_sizes(dev, tmp_path):
    rng = np.random.default_rng(12)
    vol = (np.cumsum(rng.standard_normal((4, 1024, 1024)), axis=2) * 6 + 3000 + rng.random((4, 1024, 1024)) * 40).clip(0, 65535).astype(np.uint16)
    brickio.save_tiff_series(tmp_path / "h", vol)
    brickio.save_tiff_series_device(tmp_path / "g", torch.from_numpy(vol).to(dev))
    size = lambda d: sum(f.stat().st_size for f in brickio.list_tiff_series(d))   # noqa: E731
    h, g = size(tmp_path / "h"), size(tmp_path / "g")
    assert np.array_equal(brickio.load_tiff_series(tmp_path / "g"), vol)
    from PIL import Image
    assert np.array_equal(np.stack([np.asarray(Image.open(f)) for f in brickio.list_tiff_series(tmp_path / "g")]), vol)
    # smooth content is stored horizontally differenced (TIFF predictor 2, chosen per slice from the two histograms): entropy coding
    # of the differences beats deflate level 1 on the samples
    assert Image.open(brickio.list_tiff_series(tmp_path / "g")[0]).tag_v2.get(317) == 2
    assert g < 1.02 * h and g < 0.8 * vol.nbytes, (h, g, vol.nbytes)
    # noise keeps its samples as they are (differences of noise are noisier), and so does float32
    noise = (rng.random((2, 512, 640)) * 65535).astype(np.uint16)
    brickio.save_tiff_series_device(tmp_path / "n", torch.from_numpy(noise).to(dev))
    assert Image.open(brickio.list_tiff_series(tmp_path / "n")[0]).tag_v2.get(317, 1) == 1
    assert np.array_equal(brickio.load_tiff_series(tmp_path / "n"), noise)
    # rows whose length is no multiple of the lanes' 32 bytes, differenced: the row starts fall inside the lanes' chunks
    ramp = (np.arange(3 * 97 * 211).reshape(3, 97, 211) % 5000 + 100).astype(np.uint16)
    brickio.save_tiff_series_device(tmp_path / "r", torch.from_numpy(ramp).to(dev))
    assert Image.open(brickio.list_tiff_series(tmp_path / "r")[0]).tag_v2.get(317) == 2
    assert np.array_equal(brickio.load_tiff_series(tmp_path / "r"), ramp)
    assert np.array_equal(np.stack([np.asarray(Image.open(f)) for f in brickio.list_tiff_series(tmp_path / "r")]), ramp)
    ramp8 = (np.arange(2 * 50 * 77).reshape(2, 50, 77) % 200).astype(np.uint8)
    brickio.save_tiff_series_device(tmp_path / "r8", torch.from_numpy(ramp8).to(dev))
    assert np.array_equal(np.stack([np.asarray(Image.open(f)) for f in brickio.list_tiff_series(tmp_path / "r8")]), ramp8)
    assert np.array_equal(brickio.load_tiff_series(tmp_path / "r8"), ramp8)


def test_device_writer_on_many_small_shapes(dev, tmp_path):
    """Forty random volumes -- extents from one sample to a few hundred, 8 / 16-bit and float samples, contents from constant to noise,
    ramps that pick the predictor -- through the device writer and back through libtiff and the library's reader."""
    from PIL import Image
    rng = np.random.default_rng(77)
    for case in range(40):
        nz, ny, nx = int(rng.integers(1, 4)), int(rng.integers(1, 200)), int(rng.integers(1, 300))
        dt = (np.uint8, np.uint16, np.float32)[case % 3]
        kind = case % 5
        if kind == 0:
            v = np.full((nz, ny, nx), 7, dt)
        elif kind == 1:
            v = (rng.random((nz, ny, nx)) * 250).astype(dt)
        elif kind == 2:
            v = (np.arange(nz * ny * nx).reshape(nz, ny, nx) % 251).astype(dt)
        elif kind == 3:
            v = (np.cumsum(rng.standard_normal((nz, ny, nx)), axis=2) * 3 + 120).clip(0, 255).astype(dt)
        else:
            v = (rng.random((nz, ny, nx)) > 0.97).astype(dt) * (200 if dt != np.float32 else 0.5)
        v = np.ascontiguousarray(v.astype(dt))
        d = tmp_path / f"c{case}"
        assert brickio.save_tiff_series_device(d, torch.from_numpy(v).to(dev)) == nz
        files = brickio.list_tiff_series(d)
        back = np.stack([np.asarray(Image.open(f)).reshape(ny, nx) for f in files])
        assert back.dtype == v.dtype and np.array_equal(back, v), (case, v.shape, dt)
        assert np.array_equal(brickio.load_tiff_series(d), v), (case, v.shape, dt)
