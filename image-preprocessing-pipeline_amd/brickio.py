"""Block / slab file formats around the deconvolution path (host-side I/O, no GPU):

  save_lz4 / load_lz4     the LZ4 brick cache of LsDeconv.m (``bl_<n>.lz4``): save_lz4_mex.c:50-175, load_lz4_mex.c,
                          load_slab_lz4.cpp:60-90.  33280-byte header (magic 'LZC1', dtype, dims in MATLAB order,
                          chunk table) followed by LZ4 *block*-compressed chunks of at most 1 GiB, back to back.
  load_tiff_series        the input reader (LsDeconv.m:585-588, load_bl_tif.cpp): a folder of 2-D ``*.tif`` slices,
                          sorted by name, 8 / 16-bit integers or 32-bit float -> (Z, Y, X) array.
  save_tiff_series        the output writer (LsDeconv.m:1120-1145, save_bl_tif.cpp): ``img_%06d.tif`` per z slice,
                          existing slices are skipped (resume).

Arrays are C-order (Z, Y, X) == MATLAB [X, Y, Z] column-major, so the raw bytes of a brick are identical and ``dims`` is stored
as (X, Y, Z).  liblz4 (the system's ``liblz4.so.1``) is bound with ctypes; TIFF goes through Pillow.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import os
from pathlib import Path

import numpy as np

MAGIC = 0x4C5A4331          # 'LZC1' (save_lz4_mex.c:52)
HEADER_SIZE = 33280         # save_lz4_mex.c:49
MAX_DIMS, MAX_CHUNKS = 16, 2048
CHUNK_SIZE = 1 << 30        # save_lz4_mex.c:50
DT_DOUBLE, DT_SINGLE, DT_UINT16 = 1, 2, 3
_DTYPES = {DT_DOUBLE: np.float64, DT_SINGLE: np.float32, DT_UINT16: np.uint16}
_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}

# file_header_t with the natural alignment both C files compile it with (save_lz4_mex.c:56-67, load_slab_lz4.cpp:66-74)
HEADER = np.dtype({
    "names": ["magic", "dtype", "ndims", "dims", "total_uncompressed", "chunk_size", "num_chunks", "chunk_uncomp", "chunk_comp"],
    "formats": ["<u4", "u1", "u1", ("<u8", MAX_DIMS), "<u8", "<u8", "<u4", ("<u8", MAX_CHUNKS), ("<u8", MAX_CHUNKS)],
    "offsets": [0, 4, 5, 8, 136, 144, 152, 160, 160 + 8 * MAX_CHUNKS],
    "itemsize": HEADER_SIZE,
})

_lz4 = None


def _lib():
    global _lz4
    if _lz4 is None:
        name = ctypes.util.find_library("lz4") or "liblz4.so.1"
        try:
            lib = C.CDLL(name)
        except OSError as e:
            raise RuntimeError("the LZ4 brick format needs liblz4 (liblz4.so.1)") from e
        lib.LZ4_compressBound.restype = C.c_int
        lib.LZ4_compressBound.argtypes = [C.c_int]
        lib.LZ4_compress_default.restype = C.c_int
        lib.LZ4_compress_default.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        lib.LZ4_decompress_safe.restype = C.c_int
        lib.LZ4_decompress_safe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        _lz4 = lib
    return _lz4


def save_lz4(filename, array, chunk_size=CHUNK_SIZE, pool=None):
    """``save_lz4_mex(filename, array)``: the header is written twice -- a placeholder first, the final one with the chunk sizes
    at the end (save_lz4_mex.c:131-175); the file appears under its name only when complete (callers write ``*.tmp`` and rename,
    LsDeconv.m:805-806).  ``chunk_size``: the reference compresses chunks of 1 GiB one after the other; the format records every
    chunk's sizes in its header and both of its loaders walk that table (load_lz4_mex.c:135-165, load_slab_lz4.cpp:100-130), so
    smaller chunks read back the same -- and ``pool`` (a ``concurrent.futures`` executor) compresses them side by side: liblz4
    runs outside the GIL.  The array's memory is read in place (no copy for contiguous input)."""
    a = np.ascontiguousarray(array)
    if a.dtype not in _CODES:
        raise TypeError("save_lz4_mex:BadType: Only double, single, and uint16 arrays are supported.")
    if not 1 <= chunk_size <= 0x7E000000:
        raise ValueError("chunk size outside LZ4's input limit")
    raw = a.reshape(-1).view(np.uint8)
    total = raw.size
    n_chunks = (total + chunk_size - 1) // chunk_size
    if n_chunks > MAX_CHUNKS:
        raise ValueError("save_lz4_mex:TooManyChunks: Too many chunks. Increase MAX_CHUNKS or chunk size.")
    h = np.zeros((), dtype=HEADER)
    h["magic"], h["dtype"], h["ndims"] = MAGIC, _CODES[a.dtype], a.ndim
    h["dims"][:a.ndim] = a.shape[::-1]                                   # MATLAB order: fastest axis first
    h["total_uncompressed"], h["chunk_size"], h["num_chunks"] = total, chunk_size, n_chunks
    lib = _lib()

    # Every chunk is either deflated by liblz4, or -- when a 256-KB sample from its middle does not shrink by 3 % (float32 cores: LZ4
    # finds nothing in mantissa noise), or the whole does not shrink -- left as it is: one literal run is a valid LZ4 block (token
    # 0xF0, the length in bytes of 255, the bytes; LZ4_decompress_safe of both loaders reads it), 0.4 % longer than the samples.
    probe_n = 256 << 10

    def literal_header(n):
        if n < 15:
            return bytes([n << 4])
        q, r = divmod(n - 15, 255)
        return b"\xf0" + b"\xff" * q + bytes([r])

    def prepare(i):
        src = raw[i * chunk_size:(i + 1) * chunk_size]
        if src.size >= 4 * probe_n:
            mid = (src.size // 2) & ~4095
            tmp = np.empty(lib.LZ4_compressBound(probe_n), np.uint8)
            k = lib.LZ4_compress_default(src[mid:].ctypes.data, tmp.ctypes.data, probe_n, int(tmp.size))
            if k > 0.97 * probe_n:
                return i, literal_header(int(src.size)), None
        dst = np.empty(lib.LZ4_compressBound(int(src.size)), np.uint8)
        n = lib.LZ4_compress_default(src.ctypes.data, dst.ctypes.data, int(src.size), int(dst.size))
        if n <= 0:
            raise RuntimeError(f"save_lz4_mex:CompressionFailed: chunk {i}")
        if n >= src.size:                                     # (shrank in the sample, not as a whole)
            return i, literal_header(int(src.size)), None
        return i, None, dst[:n]

    with open(filename, "wb") as f:
        f.write(h.tobytes())
        if pool is None or n_chunks <= 1:
            for i in range(n_chunks):
                _, head, comp = prepare(i)
                src = raw[i * chunk_size:(i + 1) * chunk_size]
                if comp is None:
                    f.write(head)
                    f.write(memoryview(src))
                else:
                    f.write(memoryview(comp))
                h["chunk_uncomp"][i], h["chunk_comp"][i] = int(src.size), (len(head) + int(src.size)) if comp is None else int(comp.size)
        else:
            # Chunks side by side, in two steps: (1) prepared on the pool; (2) the sizes are known, so every chunk has its place in the
            # file: the pool writes them with pwrite, the unshrinkable ones straight from the caller's buffer.  (Before: chunks written
            # one after the other by the calling thread -- 0.3 s per 860-MB core, which was what decwrap's writers did all day.)
            f.flush()
            fd = f.fileno()

            def pwrite_all(buf, at):
                mv = memoryview(buf).cast("B")
                while len(mv):
                    k = os.pwrite(fd, mv, at)
                    mv, at = mv[k:], at + k

            parts = [fu.result() for fu in [pool.submit(prepare, i) for i in range(n_chunks)]]
            offs, at = [], HEADER_SIZE
            for i, head, comp in parts:
                usize = int(min(chunk_size, total - i * chunk_size))
                csize = len(head) + usize if comp is None else int(comp.size)
                h["chunk_uncomp"][i], h["chunk_comp"][i] = usize, csize
                offs.append(at)
                at += csize

            def put(i, head, comp):
                if comp is not None:
                    pwrite_all(comp, offs[i])
                else:
                    pwrite_all(head, offs[i])
                    pwrite_all(raw[i * chunk_size:(i + 1) * chunk_size], offs[i] + len(head))

            for fu in [pool.submit(put, *part) for part in parts]:
                fu.result()
        f.seek(0)
        f.write(h.tobytes())


def read_header(f):
    h = np.frombuffer(f.read(HEADER_SIZE), dtype=HEADER, count=1)[0]
    if h["magic"] != MAGIC:
        raise ValueError("bad magic")
    if int(h["dtype"]) not in _DTYPES:
        raise ValueError("unknown dtype code")
    if not 0 < int(h["num_chunks"]) <= MAX_CHUNKS and int(h["total_uncompressed"]) > 0:
        raise ValueError("bad chunk count")
    return h


def load_lz4(filename, pool=None, out=None):
    """``load_lz4_mex(filename)`` -> array in (Z, Y, X) order (the reverse of the stored MATLAB dims).  ``pool``: the chunks are
    decompressed side by side; ``out``: a writable uint8 buffer of at least the uncompressed size (e.g. pinned memory) that
    receives the bytes instead of a fresh array."""
    lib = _lib()
    with open(filename, "rb") as f:
        try:
            h = read_header(f)
        except ValueError as e:
            raise ValueError(f"{filename}: {e}") from None
        dt = np.dtype(_DTYPES[int(h["dtype"])])
        total = int(h["total_uncompressed"])
        nch = int(h["num_chunks"])
        clens = [int(h["chunk_comp"][i]) for i in range(nch)]
        ulens = [int(h["chunk_uncomp"][i]) for i in range(nch)]
        if sum(ulens) != total:
            raise ValueError(f"{filename}: size mismatch")
        if out is None:
            out = np.empty(total, np.uint8)
        else:
            out = np.asarray(out).reshape(-1).view(np.uint8)
            if out.size < total or not out.flags.writeable:
                raise ValueError("load_lz4: `out` is too small or read-only")
            out = out[:total]
        comp = np.empty(sum(clens), np.uint8)
        if f.readinto(memoryview(comp)) != comp.size:
            raise ValueError(f"{filename}: chunk: I/O error")

    def expand(i, coff, uoff):
        clen, ulen = clens[i], ulens[i]
        n = lib.LZ4_decompress_safe(comp[coff:].ctypes.data, out[uoff:].ctypes.data, clen, ulen) if ulen else 0
        if n != ulen:
            raise ValueError(f"{filename}: LZ4 error")

    jobs, coff, uoff = [], 0, 0
    for i in range(nch):
        jobs.append((i, coff, uoff))
        coff += clens[i]
        uoff += ulens[i]
    if pool is None or nch <= 1:
        for jb in jobs:
            expand(*jb)
    else:
        for fu in [pool.submit(expand, *jb) for jb in jobs]:
            fu.result()
    dims = [int(v) for v in h["dims"][:int(h["ndims"])]][::-1]
    return out.view(dt).reshape(dims)


# ------------------------------------------------------------------------------------------------ TIFF series
def _pil():
    try:
        from PIL import Image
    except ImportError as e:
        raise RuntimeError("reading / writing TIFF slices needs Pillow") from e
    return Image


def list_tiff_series(folder):
    folder = Path(folder)
    files = sorted(folder.glob("*.tif")) or sorted(folder.glob("*.tiff"))   # LsDeconv.m:585-588
    return files


_TIFF_DTYPES = {1: np.uint8, 2: np.uint16, 4: np.float32}   # include/mi_tiffio.h
_TIFF_CODES = {np.dtype(v): k for k, v in _TIFF_DTYPES.items()}


def _native():
    """The library's TIFF reader / writer (include/mi_tiffio.h: strips of raw or deflate samples on all cores, no interpreter lock),
    or None -- MI_TIFF_PILLOW=1, or the library is not built: Pillow then does everything, as it does for the files the native
    reader does not take (tiles, LZW, BigTIFF, big-endian)."""
    if os.environ.get("MI_TIFF_PILLOW"):
        return None
    try:
        from . import capi
        return capi.lib()
    except (ImportError, OSError, AttributeError):
        return None


def tiff_info(path):
    """(ny, nx), dtype (or None) and whether the native reader decodes the file; None when there is no native reader."""
    lib = _native()
    if lib is None:
        return None
    nx, ny, dt, fast = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    if lib.mi_tiff_info(os.fsencode(str(path)), C.byref(nx), C.byref(ny), C.byref(dt), C.byref(fast)) != 0:
        return None
    return (ny.value, nx.value), _TIFF_DTYPES.get(dt.value), bool(fast.value)


def read_tiff_box(files, shape_yx, dtype, y0, y1, x0, x1, out=None, threads=0):
    """``out[k] = slice files[k][y0:y1, x0:x1]`` through the native reader (every file must be one it decodes: ``tiff_info``)."""
    from . import capi
    lib = capi.lib()
    n = len(files)
    if out is None:
        out = np.empty((n, y1 - y0, x1 - x0), dtype)
    assert out.flags.c_contiguous and out.dtype == np.dtype(dtype) and out.shape == (n, y1 - y0, x1 - x0)
    paths = (C.c_char_p * n)(*[os.fsencode(str(f)) for f in files])
    capi.check(lib.mi_tiff_read_box(paths, n, int(shape_yx[1]), int(shape_yx[0]), _TIFF_CODES[np.dtype(dtype)], int(y0), int(y1), int(x0),
                                    int(x1), out.ctypes.data, int(threads)))
    return out


def load_tiff_series(folder, z0=0, z1=None):
    """Slices ``[z0, z1)`` of a folder of 2-D grayscale TIFFs as one (Z, Y, X) array (uint8 / uint16 / float32)."""
    files = list_tiff_series(folder)[z0:z1]
    if not files:
        raise RuntimeError(f"no *.tif slices in {folder}")
    info = tiff_info(files[0])
    if info is not None and info[2]:
        (ny, nx), dt, _ = info
        try:
            return read_tiff_box(files, (ny, nx), dt, 0, ny, 0, nx)
        except Exception as e:          # a later file of another kind: the general reader decides what is wrong with the folder
            if "differs from the first slice" in str(e):
                raise ValueError(str(e)) from e
    Image = _pil()
    first = np.asarray(Image.open(files[0]))
    if first.ndim != 2 or first.dtype not in (np.uint8, np.uint16, np.float32):
        raise TypeError(f"{files[0]}: 16-bit or 32bit float grayscale images supported (LsDeconv.m:1231), got {first.dtype} {first.shape}")
    vol = np.empty((len(files),) + first.shape, first.dtype)
    vol[0] = first
    for k, f in enumerate(files[1:], start=1):
        a = np.asarray(Image.open(f))
        if a.shape != first.shape or a.dtype != first.dtype:
            raise ValueError(f"{f}: slice shape / type differs from the first slice")
        vol[k] = a
    return vol


def save_tiff_series_device(folder, vol, first_index=1):
    """``save_tiff_series`` for a volume that lies in device memory (a contiguous uint8 / uint16 / float32 CUDA tensor (Z, Y, X)): the
    slices are deflated on the device (``mi_tiff_write_series_device``), the host frames and writes them.  Existing slices are kept."""
    import torch
    from . import capi
    if not (isinstance(vol, torch.Tensor) and vol.is_cuda and vol.dim() == 3 and vol.is_contiguous()):
        raise ValueError("save_tiff_series_device: a contiguous 3-D device tensor is expected")
    code = {torch.uint8: 1, torch.uint16: 2, torch.float32: 4}.get(vol.dtype)
    if code is None:
        raise TypeError("save_tiff_series_device: uint8 / uint16 / float32 samples")
    folder = Path(folder)
    folder.mkdir(parents=True, exist_ok=True)
    n = int(vol.shape[0])
    paths = (C.c_char_p * n)(*[os.fsencode(str(folder / f"img_{first_index + k:06d}.tif")) for k in range(n)])
    made = C.c_int(0)
    capi.check(capi.lib().mi_tiff_write_series_device(vol.device.index, capi.current_stream_ptr(vol.device), paths, n, vol.data_ptr(), code,
                                                      int(vol.shape[2]), int(vol.shape[1]), 0, C.byref(made)))
    return int(made.value)


def save_tiff_series(folder, vol, first_index=1, compression="tiff_adobe_deflate"):
    """``img_%06d.tif`` per z slice, deflate-compressed; slices that already exist are left alone (LsDeconv.m:1120-1145).
    Returns the number of slices written."""
    folder = Path(folder)
    folder.mkdir(parents=True, exist_ok=True)
    vol = np.asarray(vol)
    if vol.ndim != 3 or vol.dtype not in (np.uint8, np.uint16, np.float32):
        raise TypeError("save_tiff_series: a 3-D uint8 / uint16 / float32 volume is expected")
    lib = _native()
    if lib is not None and compression in ("tiff_adobe_deflate", None):
        # one slice per task on all cores, Adobe deflate at level 1 like save_bl_tif.cpp:343-346 (Pillow: 29 MB/s under the lock)
        from . import capi
        v = np.ascontiguousarray(vol)
        n = v.shape[0]
        paths = (C.c_char_p * n)(*[os.fsencode(str(folder / f"img_{first_index + k:06d}.tif")) for k in range(n)])
        made = C.c_int(0)
        capi.check(lib.mi_tiff_write_series(paths, n, v.ctypes.data, _TIFF_CODES[v.dtype], v.shape[2], v.shape[1],
                                            0 if compression is None else 1, 1, 0, C.byref(made)))
        return int(made.value)
    Image = _pil()
    written = 0
    for k in range(vol.shape[0]):
        path = folder / f"img_{first_index + k:06d}.tif"
        if path.exists():
            continue
        tmp = path.with_suffix(".tif.tmp")
        Image.fromarray(np.ascontiguousarray(vol[k])).save(tmp, format="TIFF", compression=compression)
        os.replace(tmp, path)
        written += 1
    return written
