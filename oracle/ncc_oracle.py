"""CPU ORACLE wrapper (test infrastructure, NOT product code) for MIP-NCC registration.

ctypes front-end to ``oracle/libncc_oracle.so`` (our C restatement, ``ncc_oracle.c``) and, when
it has been built, to ``oracle/_ref/libcrossmips_ref.so`` (the unmodified reference crossmips
sources compiled by ``oracle/Makefile``).  Only tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NORTH_SOUTH, WEST_EAST = 0, 1  # CrossMIPs.h:51-52


class Params(C.Structure):
    """orc_params / ref_params (subset of NCC_parms_t, CrossMIPs.h:65-86; enhance is always false,
    PDAlgoMIPNCC.cpp:81)."""
    _fields_ = [("maxIter", C.c_int), ("maxThr", C.c_float), ("widthThr", C.c_float),
                ("wRangeThr_i", C.c_int), ("wRangeThr_j", C.c_int), ("wRangeThr_k", C.c_int),
                ("minPoints", C.c_int), ("minDim_NCCsrc", C.c_int), ("minDim_NCCmap", C.c_int),
                ("UNR_NCC", C.c_float), ("INF_W", C.c_int), ("INV_COORD", C.c_int)]


class Descr(C.Structure):
    _fields_ = [("coord", C.c_int * 3), ("NCC_maxs", C.c_float * 3), ("NCC_widths", C.c_int * 3)]


def pdalgo_params(displ_max_V: int, displ_max_H: int, displ_max_D: int) -> Params:
    """Fixed parameters of PDAlgoMIPNCC::execute (PDAlgoMIPNCC.cpp:80-94)."""
    p = Params()
    p.maxIter, p.maxThr, p.UNR_NCC, p.minPoints = 2, 0.10, 0.0, 3
    p.wRangeThr_i, p.wRangeThr_j, p.wRangeThr_k = (min(displ_max_V, 29), min(displ_max_H, 29), min(displ_max_D, 29))
    p.minDim_NCCsrc, p.minDim_NCCmap = 25, 3
    p.INF_W = max(p.wRangeThr_i, p.wRangeThr_j, p.wRangeThr_k) + 1
    p.widthThr, p.INV_COORD = 0.80, 0
    return p


def build(force: bool = False) -> None:
    if force or not os.path.exists(os.path.join(_HERE, "libncc_oracle.so")):
        subprocess.run(["make", "-C", _HERE, "libncc_oracle.so"], check=True, capture_output=True)


_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_libs: dict = {}


def _lib(kind: str):
    if kind in _libs:
        return _libs[kind]
    if kind == "oracle":
        build()
        lib = C.CDLL(os.path.join(_HERE, "libncc_oracle.so"))
        lib.orc_norm_cross_corr_mips.restype = C.c_int
        lib.orc_norm_cross_corr_mips.argtypes = [_fp, _fp] + [C.c_int] * 10 + [C.POINTER(Params), C.POINTER(Descr),
                                                                                C.POINTER(_fp), _ip]
        lib.orc_ncc_map.restype = None
        lib.orc_ncc_map.argtypes = [_fp, _fp, _fp] + [C.c_int] * 4
        lib.orc_argmax.restype = C.c_int
        lib.orc_argmax.argtypes = [_fp, C.c_int]
    elif kind == "ref":
        path = os.path.join(_HERE, "_ref", "libcrossmips_ref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        lib = C.CDLL(path)
        lib.ref_norm_cross_corr_mips.restype = C.c_int
        lib.ref_norm_cross_corr_mips.argtypes = [_fp, _fp] + [C.c_int] * 10 + [C.POINTER(Params), _ip, _fp, _ip]
        lib.ref_compute_NCC_map.restype = None
        lib.ref_compute_NCC_map.argtypes = [_fp, _fp, _fp] + [C.c_int] * 4
        lib.ref_compute_3_MIPs.restype = None
        lib.ref_compute_3_MIPs.argtypes = [_fp] * 8 + [C.c_int] * 5
    else:
        raise ValueError(kind)
    _libs[kind] = lib
    return lib


def have_ref() -> bool:
    return os.path.exists(os.path.join(_HERE, "_ref", "libcrossmips_ref.so"))


def _f(a):
    return a.ctypes.data_as(_fp)


def clamp_delays(dims_kij, n_kij, delays_kij, minDim_NCCsrc=25):
    """libcrossmips.cpp:260-262."""
    return [min(d, max(0, dim - n - minDim_NCCsrc)) for dim, n, d in zip(dims_kij, n_kij, delays_kij)]


def view_dims(dimk, dimi, dimj, ni, nj, side):
    return (dimk, dimi - ni, dimj) if side == NORTH_SOUTH else (dimk, dimi, dimj - nj)


def norm_cross_corr_mips(A, B, ni, nj, delayk, delayi, delayj, side, params: Params, kind="oracle", debug=False):
    """Returns dict(rc, coord, NCC_maxs, NCC_widths, wRangeThr[, mips, maps, delays]).  ``params`` is
    mutated like the reference does (libcrossmips.cpp:275-277)."""
    A = np.ascontiguousarray(A, np.float32)
    B = np.ascontiguousarray(B, np.float32)
    dimk, dimi, dimj = A.shape
    assert B.shape == A.shape
    out = {}
    if kind == "oracle":
        lib = _lib("oracle")
        d = Descr()
        dbg_arrays, dbg_ptrs, delays = None, None, (C.c_int * 3)()
        if debug:
            dk, di, dj = clamp_delays((dimk, dimi, dimj), (0, ni, nj), (delayk, delayi, delayj), params.minDim_NCCsrc)
            kv, iv, jv = view_dims(dimk, dimi, dimj, ni, nj, side)
            shapes = [(iv, jv), (iv, kv), (jv, kv)] * 2 + [(2 * di + 1, 2 * dj + 1), (2 * di + 1, 2 * dk + 1),
                                                          (2 * dj + 1, 2 * dk + 1)]
            dbg_arrays = [np.zeros(s, np.float32) for s in shapes]
            dbg_ptrs = (_fp * 9)(*[_f(a) for a in dbg_arrays])
        rc = lib.orc_norm_cross_corr_mips(_f(A), _f(B), dimk, dimi, dimj, 0, ni, nj, delayk, delayi, delayj, side,
                                          C.byref(params), C.byref(d), dbg_ptrs, delays)
        out.update(rc=rc, coord=list(d.coord), NCC_maxs=np.array(list(d.NCC_maxs), np.float32),
                   NCC_widths=list(d.NCC_widths), delays=list(delays))
        if debug:
            out["mips"] = dbg_arrays[:6]
            out["maps"] = dbg_arrays[6:]
    else:
        lib = _lib("ref")
        coord, maxs, widths = (C.c_int * 3)(), (C.c_float * 3)(), (C.c_int * 3)()
        rc = lib.ref_norm_cross_corr_mips(_f(A), _f(B), dimk, dimi, dimj, 0, ni, nj, delayk, delayi, delayj, side,
                                          C.byref(params), coord, maxs, widths)
        out.update(rc=rc, coord=list(coord), NCC_maxs=np.array(list(maxs), np.float32), NCC_widths=list(widths))
        if debug and rc == 0:
            dk, di, dj = clamp_delays((dimk, dimi, dimj), (0, ni, nj), (delayk, delayi, delayj), params.minDim_NCCsrc)
            kv, iv, jv = view_dims(dimk, dimi, dimj, ni, nj, side)
            mips = [np.zeros(s, np.float32) for s in [(iv, jv), (iv, kv), (jv, kv)] * 2]
            if side == NORTH_SOUTH:
                stridei, stridek, off = 0, ni * dimj, ni * dimj
            else:
                stridei, stridek, off = nj, 0, nj
            a1 = C.cast(C.addressof(_f(A).contents) + 4 * off, _fp)
            lib.ref_compute_3_MIPs(a1, _f(B), *[_f(m) for m in mips], iv, jv, kv, stridei, stridek)
            maps = []
            for m, (du, dv) in enumerate([(di, dj), (di, dk), (dj, dk)]):
                mp = np.zeros((2 * du + 1, 2 * dv + 1), np.float32)
                lib.ref_compute_NCC_map(_f(mp), _f(mips[m]), _f(mips[m + 3]), mips[m].shape[0], mips[m].shape[1], du, dv)
                maps.append(mp)
            out["mips"], out["maps"], out["delays"] = mips, maps, [di, dj, dk]
    out["wRangeThr"] = [params.wRangeThr_i, params.wRangeThr_j, params.wRangeThr_k]
    return out


def pdalgo_execute(A, B, displ_max_V, displ_max_H, displ_max_D, direction, overlap, kind="oracle", debug=False):
    """PDAlgoMIPNCC::execute (PDAlgoMIPNCC.cpp:55-114): direction 0 = vertical (N-S), 1 = horizontal (W-E)."""
    dimk, dimi, dimj = A.shape
    p = pdalgo_params(displ_max_V, displ_max_H, displ_max_D)
    ni = dimi - overlap if direction == NORTH_SOUTH else 0
    nj = dimj - overlap if direction == WEST_EAST else 0
    r = norm_cross_corr_mips(A, B, ni, nj, displ_max_D, displ_max_V, displ_max_H, direction, p, kind=kind, debug=debug)
    r["INF_W"] = p.INF_W
    return r


# --------------------------------------------------------------------------- synthetic tiles
def box_blur3(vol: np.ndarray, passes: int = 3) -> np.ndarray:
    from scipy import ndimage
    out = vol.astype(np.float32)
    for _ in range(passes):
        out = ndimage.uniform_filter(out, size=3, mode="nearest")
    return out.astype(np.float32)


def bead_field(shape_kij, seed=1234, density=1.0 / 512, passes=3) -> np.ndarray:
    """Seeded sparse-bead field in [0,1], box-blurred so features are >= 3 px wide."""
    rng = np.random.default_rng(seed)
    n = int(np.prod(shape_kij))
    vol = rng.uniform(0.01, 0.02, size=shape_kij).astype(np.float32)
    nb = max(4, int(n * density))
    idx = rng.integers(0, n, size=nb)
    vol.reshape(-1)[idx] = rng.uniform(0.2, 1.0, size=nb).astype(np.float32) * 27.0
    vol = box_blur3(vol, passes)
    return np.clip(vol / max(float(vol.max()), 1e-6), 0.0, 1.0).astype(np.float32)


def tile_pair(tile_kij, overlap, side, shift_vhd=(0, 0, 0), seed=1234, margin=12):
    """Cut two tiles A, B of shape ``tile_kij`` from one bead field so that B's true offset relative to
    A is nominal + ``shift_vhd`` (V, H, D)."""
    dk, di, dj = tile_kij
    sv, sh, sd = shift_vhd
    if side == NORTH_SOUTH:
        off = (sd, di - overlap + sv, sh)
    else:
        off = (sd, sv, dj - overlap + sh)
    m = margin
    field = bead_field((dk + abs(off[0]) + 2 * m, di + abs(off[1]) + 2 * m, dj + abs(off[2]) + 2 * m), seed)
    a0 = (m + max(0, -off[0]), m + max(0, -off[1]), m + max(0, -off[2]))
    b0 = tuple(a + o for a, o in zip(a0, off))
    A = field[a0[0]:a0[0] + dk, a0[1]:a0[1] + di, a0[2]:a0[2] + dj]
    B = field[b0[0]:b0[0] + dk, b0[1]:b0[1] + di, b0[2]:b0[2] + dj]
    return np.ascontiguousarray(A), np.ascontiguousarray(B)
