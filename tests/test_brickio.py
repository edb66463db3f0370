"""CPU: block / slab file formats: LZ4 brick files in the layout of save_lz4_mex.c / load_slab_lz4.cpp and
the TIFF series reader / writer."""
import subprocess

import numpy as np
import pytest

from ipp_amd import brickio


def test_header_layout_matches_a_naturally_aligned_c_struct(tmp_path):
    """Both C files of the reference declare the header as a plain struct and read / write 33280 bytes of it; the numpy dtype
    must therefore have the compiler's natural alignment.  Checked against gcc's offsetof on a struct of the same field types."""
    src = tmp_path / "hdr.c"
    src.write_text("""
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
typedef struct { uint32_t a; uint8_t b; uint8_t c; uint64_t d[16]; uint64_t e; uint64_t f; uint32_t g; uint64_t h[2048]; uint64_t i[2048]; } hdr_t;
int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu", offsetof(hdr_t, a), offsetof(hdr_t, b), offsetof(hdr_t, c), offsetof(hdr_t, d),
                        offsetof(hdr_t, e), offsetof(hdr_t, f), offsetof(hdr_t, g), offsetof(hdr_t, h), offsetof(hdr_t, i)); return 0; }
""")
    exe = tmp_path / "hdr"
    subprocess.run(["gcc", "-O0", "-o", str(exe), str(src)], check=True)
    offs = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    names = ["magic", "dtype", "ndims", "dims", "total_uncompressed", "chunk_size", "num_chunks", "chunk_uncomp", "chunk_comp"]
    assert [brickio.HEADER.fields[n][1] for n in names] == offs
    assert brickio.HEADER.itemsize == 33280 and brickio.MAGIC == 0x4C5A4331


@pytest.mark.parametrize("dtype", [np.float32, np.uint16, np.float64])
@pytest.mark.parametrize("chunk", [brickio.CHUNK_SIZE, 4096, 1000])
def test_lz4_brick_round_trip(tmp_path, dtype, chunk):
    rng = np.random.default_rng(3)
    a = (rng.random((7, 33, 21)) * 1000).astype(dtype)
    a[2:4] = 5                                        # compressible planes
    p = tmp_path / "bl_1.lz4"
    brickio.save_lz4(p, a, chunk_size=chunk)
    b = brickio.load_lz4(p)
    assert b.dtype == a.dtype and b.shape == a.shape and np.array_equal(a, b)
    with open(p, "rb") as f:
        h = brickio.read_header(f)
    assert [int(v) for v in h["dims"][:3]] == [21, 33, 7]                 # MATLAB [X Y Z]
    assert int(h["total_uncompressed"]) == a.nbytes and int(h["num_chunks"]) == -(-a.nbytes // chunk)
    assert int(h["chunk_uncomp"][:int(h["num_chunks"])].sum()) == a.nbytes
    assert p.stat().st_size == brickio.HEADER_SIZE + int(h["chunk_comp"][:int(h["num_chunks"])].sum())


def test_lz4_brick_errors(tmp_path):
    with pytest.raises(TypeError, match="Only double, single, and uint16"):
        brickio.save_lz4(tmp_path / "x.lz4", np.zeros((2, 2), np.int32))
    p = tmp_path / "bad.lz4"
    p.write_bytes(b"\0" * brickio.HEADER_SIZE)
    with pytest.raises(ValueError, match="bad magic"):
        brickio.load_lz4(p)
    a = np.arange(5000, dtype=np.float32)
    brickio.save_lz4(p, a)
    raw = bytearray(p.read_bytes())
    raw[brickio.HEADER_SIZE + 10] ^= 0xFF              # corrupt the stream
    raw = raw[:-7]
    p.write_bytes(bytes(raw))
    with pytest.raises(ValueError, match="LZ4 error|I/O error"):
        brickio.load_lz4(p)


@pytest.mark.parametrize("dtype,scale", [(np.uint8, 255), (np.uint16, 65535), (np.float32, 1.0)])
def test_tiff_series_round_trip_and_resume(tmp_path, dtype, scale):
    rng = np.random.default_rng(1)
    vol = (rng.random((5, 12, 17)) * scale).astype(dtype)
    assert brickio.save_tiff_series(tmp_path, vol) == 5
    names = sorted(p.name for p in tmp_path.glob("*.tif"))
    assert names == [f"img_{k:06d}.tif" for k in range(1, 6)]
    back = brickio.load_tiff_series(tmp_path)
    assert back.dtype == vol.dtype and np.array_equal(back, vol)
    assert np.array_equal(brickio.load_tiff_series(tmp_path, 1, 3), vol[1:3])
    (tmp_path / "img_000003.tif").unlink()
    assert brickio.save_tiff_series(tmp_path, vol) == 1            # only the missing slice is written again
    with pytest.raises(RuntimeError, match="no \\*.tif"):
        brickio.load_tiff_series(tmp_path / "nothing_here")


def test_lazy_tiff_volume_reads_boxes(tmp_path):
    """decwrap's input for TIFF folders: box reads equal the same box of the fully loaded series; the slice cache stays
    within its budget."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "image-preprocessing-pipeline_amd"))
    from ipp_amd import brickio, decwrap
    rng = np.random.default_rng(5)
    vol = (rng.random((9, 20, 23)) * 60000).astype(np.uint16)
    brickio.save_tiff_series(tmp_path / "s", vol)
    lazy = decwrap.LazyTiffVolume(tmp_path / "s", cache_bytes=3 * vol[0].nbytes)
    assert lazy.shape == vol.shape and lazy.dtype == vol.dtype
    for box in [(slice(0, 9), slice(0, 20), slice(0, 23)), (slice(2, 5), slice(3, 17), slice(1, 8)), (slice(8, 9), slice(19, 20), slice(0, 23))]:
        assert np.array_equal(lazy[box], vol[box])
    assert len(lazy._cache) <= 3
    # more slices than one bulk read of the library's reader takes (sixteen), a cache that holds a few of them, several threads at once
    from concurrent.futures import ThreadPoolExecutor
    big = (rng.random((45, 33, 40)) * 60000).astype(np.uint16)
    brickio.save_tiff_series(tmp_path / "b", big)
    for budget in (5, 100):
        lazy = decwrap.LazyTiffVolume(tmp_path / "b", cache_bytes=budget * big[0].nbytes)
        boxes = [(slice(0, 45), slice(0, 33), slice(0, 40)), (slice(3, 40), slice(5, 30), slice(7, 33)), (slice(17, 18), slice(0, 33), slice(0, 40)),
                 (slice(10, 45), slice(32, 33), slice(39, 40))]
        with ThreadPoolExecutor(4) as ex:
            for box, got in zip(boxes, ex.map(lambda b: lazy[b], boxes)):
                assert np.array_equal(got, big[box])
        assert len(lazy._cache) <= budget and lazy._fast
    # a folder the library's reader does not take (LZW): the same answers through Pillow
    from PIL import Image
    (tmp_path / "l").mkdir()
    for k in range(20):
        Image.fromarray(big[k]).save(tmp_path / "l" / f"s{k:03d}.tif", format="TIFF", compression="tiff_lzw")
    lazy = decwrap.LazyTiffVolume(tmp_path / "l", cache_bytes=4 * big[0].nbytes)
    assert not lazy._fast and np.array_equal(lazy[2:19, 1:30, 2:39], big[2:19, 1:30, 2:39])


def test_parallel_chunks_round_trip_and_out_buffer(tmp_path):
    """Bricks written as small chunks on a thread pool (decwrap's writers) are byte-identical to the same chunks written one after
    the other, read back by either route, and land in a caller's buffer when one is given (save_lz4_mex.c:131-175,
    load_lz4_mex.c:135-165: the chunk table of the header drives the loaders)."""
    from concurrent.futures import ThreadPoolExecutor
    from ipp_amd import brickio
    rng = np.random.default_rng(3)
    a = (rng.random((7, 33, 65)) * np.linspace(0, 1, 65)).astype(np.float32)
    a[2:4] = 0.0                                                         # a compressible stretch
    with ThreadPoolExecutor(4) as pool:
        brickio.save_lz4(tmp_path / "par.lz4", a, chunk_size=4096, pool=pool)
        brickio.save_lz4(tmp_path / "ser.lz4", a, chunk_size=4096)
        assert (tmp_path / "par.lz4").read_bytes() == (tmp_path / "ser.lz4").read_bytes()
        with open(tmp_path / "par.lz4", "rb") as f:
            h = brickio.read_header(f)
        assert int(h["num_chunks"]) == -(-a.nbytes // 4096) and int(h["chunk_size"]) == 4096
        b = brickio.load_lz4(tmp_path / "par.lz4", pool=pool)
        buf = np.full(a.nbytes + 64, 0xAB, np.uint8)
        c = brickio.load_lz4(tmp_path / "par.lz4", pool=pool, out=buf)
    assert np.array_equal(a, b) and np.array_equal(a, c) and b.dtype == np.float32 and c.shape == a.shape
    assert np.shares_memory(c, buf) and np.all(buf[a.nbytes:] == 0xAB)
    import pytest
    with pytest.raises(ValueError, match="too small"):
        brickio.load_lz4(tmp_path / "par.lz4", out=np.empty(16, np.uint8))


def test_unshrinkable_chunks_are_stored_as_literal_runs(tmp_path):
    """float32 cores are mantissa noise to LZ4: a chunk whose sample does not shrink is written as ONE literal run -- a valid LZ4 block
    (token 0xF0, length bytes, the samples) that LZ4_decompress_safe, the call of both reference loaders (load_lz4_mex.c:150,
    load_slab_lz4.cpp:118), reads back -- straight from the caller's buffer, the chunks of a brick side by side with pwrite."""
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(9)
    a = rng.random((3, 512, 600)).astype(np.float32)              # 3.7 MB of noise ...
    a[1, :300] = 0.0                                               # ... with 0.6 MB of zeros in the second chunk
    chunk = 1 << 20
    with ThreadPoolExecutor(4) as pool:
        brickio.save_lz4(tmp_path / "par.lz4", a, chunk_size=chunk, pool=pool)
        brickio.save_lz4(tmp_path / "ser.lz4", a, chunk_size=chunk)
        assert (tmp_path / "par.lz4").read_bytes() == (tmp_path / "ser.lz4").read_bytes()
        with open(tmp_path / "par.lz4", "rb") as f:
            h = brickio.read_header(f)
        n = int(h["num_chunks"])
        usz, csz = [int(v) for v in h["chunk_uncomp"][:n]], [int(v) for v in h["chunk_comp"][:n]]
        assert n == 4 and sum(usz) == a.nbytes
        head = 1 + (chunk - 15) // 255 + 1
        assert csz[0] == chunk + head and csz[2] == chunk + head          # literal runs
        assert csz[1] < 0.6 * chunk                                        # the chunk with the zeros went through LZ4
        raw = (tmp_path / "par.lz4").read_bytes()
        at = brickio.HEADER_SIZE
        assert raw[at] == 0xF0 and raw[at + 1] == 0xFF and raw[at + head:at + head + 64] == a.tobytes()[:64]
        assert np.array_equal(brickio.load_lz4(tmp_path / "par.lz4", pool=pool), a) and np.array_equal(brickio.load_lz4(tmp_path / "ser.lz4"), a)


# ------------------------------------------------------------------------------------------------ the library's TIFF reader / writer
# (include/mi_tiffio.h, csrc/tiffio.hip: host code -- strips of raw or deflate samples, one slice per task on all cores; the checker
#  is Pillow's libtiff, which reads what the library writes and writes what the library reads)
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
@pytest.mark.parametrize("compression", ["tiff_adobe_deflate", None])
def test_native_tiff_writer_is_read_by_libtiff(tmp_path, dtype, compression):
    from PIL import Image
    rng = np.random.default_rng(3)
    vol = (rng.random((4, 301, 517)) * 200).astype(dtype)           # odd extents; 301 rows of 517 samples: one strip (uint8) or several
    big = (rng.random((2, 1500, 1111)) * 60000).astype(dtype)       # several strips of ~1 MiB
    for k, v in enumerate((vol, big)):
        d = tmp_path / f"s{k}"
        assert brickio.save_tiff_series(d, v, first_index=7, compression=compression) == v.shape[0]
        files = brickio.list_tiff_series(d)
        assert [f.name for f in files][:2] == ["img_000007.tif", "img_000008.tif"] and not list(d.glob("*.tmp"))
        back = np.stack([np.asarray(Image.open(f)) for f in files])
        assert back.dtype == np.dtype(dtype) and np.array_equal(back, v)
        im = Image.open(files[0])
        assert im.tag_v2[259] == (8 if compression else 1) and im.tag_v2[277] == 1 and im.tag_v2[262] == 1
        assert brickio.save_tiff_series(d, v * 0, first_index=7, compression=compression) == 0     # existing slices are kept (LsDeconv.m:1120-1132)
        assert np.array_equal(brickio.load_tiff_series(d), v)


@pytest.mark.parametrize("compression,kwargs,fast", [("tiff_adobe_deflate", {}, True), ("tiff_deflate", {}, True), (None, {}, True),
                                                     ("tiff_adobe_deflate", {"tiffinfo": {317: 2}}, True), ("tiff_lzw", {}, False)],
                         ids=["adobe_deflate", "deflate_32946", "raw", "horizontal_predictor", "lzw_goes_to_pillow"])
def test_native_tiff_reader_on_files_libtiff_wrote(tmp_path, compression, kwargs, fast):
    from PIL import Image
    rng = np.random.default_rng(4)
    vol = (np.cumsum(rng.standard_normal((3, 700, 640)), axis=2) * 40 + 30000).clip(0, 65535).astype(np.uint16)
    for k in range(3):
        Image.fromarray(vol[k]).save(tmp_path / f"s{k:03d}.tif", format="TIFF", compression=compression, **kwargs)
    files = brickio.list_tiff_series(tmp_path)
    shape, dt, is_fast = brickio.tiff_info(files[0])
    assert shape == (700, 640) and is_fast == fast and (dt == np.uint16 or not fast)
    assert np.array_equal(brickio.load_tiff_series(tmp_path), vol)                 # (the LZW folder through Pillow)
    assert np.array_equal(brickio.load_tiff_series(tmp_path, 1, 3), vol[1:3])
    if fast:
        for box in ((0, 700, 0, 640), (13, 14, 5, 6), (100, 650, 33, 600), (699, 700, 639, 640)):
            got = brickio.read_tiff_box(files, (700, 640), np.uint16, *box, threads=3)
            assert np.array_equal(got, vol[:, box[0]:box[1], box[2]:box[3]])
        from ipp_amd import capi
        with pytest.raises(capi.MiError, match="outside"):
            brickio.read_tiff_box(files, (700, 640), np.uint16, 0, 701, 0, 640)
        with pytest.raises(capi.MiError, match="differs from the first slice"):
            brickio.read_tiff_box(files, (700, 641), np.uint16, 0, 700, 0, 640)


def test_native_tiff_errors_and_the_zlib_route(tmp_path):
    import os
    import sys
    (tmp_path / "junk.tif").write_bytes(b"not a tiff at all")
    assert brickio.tiff_info(tmp_path / "junk.tif") is None and brickio.tiff_info(tmp_path / "missing.tif") is None
    # a truncated file: the reader says which, the folder loader raises
    d = tmp_path / "t"
    vol = (np.arange(2 * 64 * 64).reshape(2, 64, 64) % 251).astype(np.uint8)
    brickio.save_tiff_series(d, vol)
    f = brickio.list_tiff_series(d)[1]
    f.write_bytes(f.read_bytes()[:100])
    with pytest.raises(Exception):
        brickio.load_tiff_series(d)
    # libdeflate is loaded at run time when the host has it; without it (MI_TIFF_ZLIB=1: a fresh process) zlib writes the same format
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from ipp_amd import brickio, capi; "
            "v = (np.arange(3 * 200 * 300).reshape(3, 200, 300) %% 1000).astype(np.uint16); "
            "assert capi.lib().mi_tiff_codec() == b'zlib'; assert brickio.save_tiff_series(%r, v) == 3; "
            "assert np.array_equal(brickio.load_tiff_series(%r), v)" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "z"),
                                                                        str(tmp_path / "z")))
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, MI_TIFF_ZLIB="1"))
    from PIL import Image
    v = (np.arange(3 * 200 * 300).reshape(3, 200, 300) % 1000).astype(np.uint16)
    assert np.array_equal(np.stack([np.asarray(Image.open(p)) for p in brickio.list_tiff_series(tmp_path / "z")]), v)
    assert np.array_equal(brickio.load_tiff_series(tmp_path / "z"), v)              # ... and this process' codec reads it


def test_pillow_route_still_works(tmp_path, monkeypatch):
    monkeypatch.setenv("MI_TIFF_PILLOW", "1")
    vol = (np.arange(2 * 40 * 50).reshape(2, 40, 50) % 777).astype(np.uint16)
    assert brickio.tiff_info(tmp_path / "x.tif") is None
    assert brickio.save_tiff_series(tmp_path / "p", vol) == 2
    assert np.array_equal(brickio.load_tiff_series(tmp_path / "p"), vol)


def test_native_tiff_reader_survives_corrupted_files(tmp_path):
    """Five hundred corruptions of a valid deflate TIFF -- random bytes anywhere, half of them in the directory, some files cut short --
    are each either refused or decoded (the byte hit padding); run in a process of its own, which must end normally."""
    import os
    import sys
    code = r'''
import sys, os
import numpy as np
sys.path.insert(0, %r)
from ipp_amd import brickio, capi
rng = np.random.default_rng(5)
d = %r
vol = (rng.random((1, 120, 160)) * 60000).astype(np.uint16)
brickio.save_tiff_series(d, vol)
path = os.path.join(d, "img_000001.tif")
good = open(path, "rb").read()
decoded = refused = 0
for it in range(500):
    b = bytearray(good)
    for _ in range(int(rng.integers(1, 6))):
        pos = int(rng.integers(0, len(b))) if rng.random() < 0.5 else int(rng.integers(max(0, len(b) - 300), len(b)))
        b[pos] = int(rng.integers(0, 256))
    if rng.random() < 0.1:
        b = b[:int(rng.integers(0, len(b)))]
    open(path, "wb").write(bytes(b))
    try:
        info = brickio.tiff_info(path)
        if info is not None and info[2] and info[0] == (120, 160) and info[1] == np.uint16:
            brickio.read_tiff_box([path], (120, 160), np.uint16, 0, 120, 0, 160)
            decoded += 1
        else:
            refused += 1
    except capi.MiError:
        refused += 1
assert decoded + refused == 500 and refused > 300, (decoded, refused)
print("ok", decoded, refused)
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "f"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.returncode, r.stdout[-300:], r.stderr[-600:])
