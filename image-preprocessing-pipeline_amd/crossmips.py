"""Host-side mirror of TeraStitcher's MIP-NCC pairwise registration (crossmips + its caller).

    norm_cross_corr_mips(A, B, dimk, dimi, dimj, nk, ni, nj, delayk, delayi, delayj, side, NCC_params)
                                                             CrossMIPs.h:89-91 / libcrossmips.cpp:101
    PDAlgoMIPNCC.execute(stk_A, ..., stk_B, ..., displ_max_V, displ_max_H, displ_max_D, direction, overlap)
                                                             stitcher/PDAlgoMIPNCC.cpp:55-114
    compute_displacements(tiles, ...)                        stitcher/StackStitcher.cpp:201-374 (pair loop)

Stacks are float32 (dimk, dimi, dimj) arrays in [0,1] (loadImageStack convention, tiff2D.cpp:606-610): torch
CUDA tensors stay on the device, numpy arrays are uploaded.  All arithmetic runs in libmi_ipp.so
(include/mi_crossmips.h); errors the reference throws as iom::exception surface as ``capi.MiError``.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np
import torch

from . import capi
from .capi import NORTH_SOUTH, WEST_EAST, NccDescr, NccParams, check, lib

dir_vertical, dir_horizontal = NORTH_SOUTH, WEST_EAST  # Displacement.h:44 / CrossMIPs.h:51-52
S_NCC_WIDTH_MAX = 30  # S_config.h:86
S_DISPL_SEARCH_RADIUS_DEF = 25  # S_config.h:59 (--sV/--sH/--sD default)
S_SUBVOL_DIM_D_DEFAULT = 200  # S_config.h:60


def NCC_parms_t(displ_max_V=25, displ_max_H=25, displ_max_D=25) -> NccParams:
    """The parameter block PDAlgoMIPNCC::execute builds (PDAlgoMIPNCC.cpp:80-94)."""
    p = NccParams()
    lib().mi_ncc_default_params(int(displ_max_V), int(displ_max_H), int(displ_max_D), C.byref(p))
    return p


def _dev_tensor(x, device):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(device)
    if x.dtype != torch.float32:
        raise TypeError("stacks must be float32 (iom::real_t)")
    return x.to(device).contiguous()


def norm_cross_corr_mips(A, B, dimk=None, dimi=None, dimj=None, nk=0, ni=0, nj=0, delayk=0, delayi=0, delayj=0,
                         side=NORTH_SOUTH, NCC_params: NccParams | None = None, device=None) -> NccDescr:
    """Alignment of stack B relative to stack A (CrossMIPs.h:92-115).  ``NCC_params`` is mandatory like in the
    reference (libcrossmips.cpp:147) and its ``wRangeThr_*`` fields are updated in place (:275-277)."""
    if NCC_params is None:
        raise ValueError("CrossMIPs: missing configuration parameters")
    capi.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    a, b = _dev_tensor(A, dev), _dev_tensor(B, dev)
    if a.dim() != 3 or a.shape != b.shape:
        raise ValueError("stacks A and B don't have the same dimensions")  # PDAlgoMIPNCC.cpp:67-68
    dk, di, dj = (int(s) for s in a.shape)
    if (dimk, dimi, dimj) != (None, None, None) and (dimk, dimi, dimj) != (dk, di, dj):
        raise ValueError("dimk/dimi/dimj do not match the stack shape")
    out = NccDescr()
    check(lib().mi_ncc_mips(dev.index, capi.current_stream_ptr(dev), a.data_ptr(), b.data_ptr(), dk, di, dj, int(nk),
                            int(ni), int(nj), int(delayk), int(delayi), int(delayj), int(side), C.byref(NCC_params),
                            C.byref(out)))
    return out


@dataclass
class DisplacementMIPNCC:
    """What PDAlgoMIPNCC::execute stores on the displacement (PDAlgoMIPNCC.cpp:100-109,
    DisplacementMIPNCC.h): offsets (V,H,D), peak values, half widths + the search parameters."""
    VHD_coords: list
    NCC_maxs: list
    NCC_widths: list
    delays: list
    wRangeThrs: list
    invWidths: list
    extra: dict = field(default_factory=dict)

    def evalReliability(self, axis: int) -> float:
        """DisplacementMIPNCC.cpp:130-147: sqrt(0.5*(1 - w/invW)^2 + 0.5*peak^2)."""
        w, inv, peak = self.NCC_widths[axis], self.invWidths[axis], self.NCC_maxs[axis]
        return math.sqrt(0.5 * (1.0 - w / inv) ** 2 + 0.5 * peak * peak)


class PDAlgoMIPNCC:
    """Pairwise displacement algorithm "MIP-NCC" (stitcher/PDAlgoMIPNCC.cpp)."""

    @staticmethod
    def execute(stk_A, stk_B, displ_max_V, displ_max_H, displ_max_D, overlap_direction, overlap, device=None):
        a_shape = tuple(stk_A.shape)
        if a_shape != tuple(stk_B.shape):
            raise ValueError("in PDAlgoMIPNCC::execute(...): stacks A and B don't have the same dimensions")
        if overlap_direction not in (dir_horizontal, dir_vertical):
            raise ValueError("in PDAlgoMIPNCC::execute(...): unsupported overlapping direction")
        dim_D, dim_V, dim_H = a_shape
        params = NCC_parms_t(displ_max_V, displ_max_H, displ_max_D)
        d = norm_cross_corr_mips(stk_A, stk_B, nk=0,
                                 ni=dim_V - overlap if overlap_direction == dir_vertical else 0,
                                 nj=dim_H - overlap if overlap_direction == dir_horizontal else 0,
                                 delayk=displ_max_D, delayi=displ_max_V, delayj=displ_max_H, side=overlap_direction,
                                 NCC_params=params, device=device)
        return DisplacementMIPNCC(list(d.coord), [float(v) for v in d.NCC_maxs], list(d.NCC_widths),
                                  [displ_max_V, displ_max_H, displ_max_D],
                                  [params.wRangeThr_i, params.wRangeThr_j, params.wRangeThr_k], [params.INF_W] * 3)


def enumerate_pairs(n_rows: int, n_cols: int):
    """East and south neighbour pairs of an n_rows x n_cols tile grid: 2RC - R - C pairs
    (StackStitcher.cpp:217,223-374).  Yields (row, col, row_b, col_b, direction)."""
    for r in range(n_rows):
        for c in range(n_cols):
            if c + 1 < n_cols:
                yield r, c, r, c + 1, dir_horizontal
            if r + 1 < n_rows:
                yield r, c, r + 1, c, dir_vertical


def subvolume_layers(z_size: int, subvol_dim_D: int = S_SUBVOL_DIM_D_DEFAULT):
    """z split of computeDisplacements (StackStitcher.cpp:201-202,225): n = ceil(Z/subvoldim) layers of
    floor(Z/n) slices, the first Z mod n one slice longer.  Returns [(z0, z1)) per layer."""
    n = int(math.ceil(z_size / float(subvol_dim_D)))
    base = z_size // n
    out, z = [], 0
    for k in range(1, n + 1):
        d = base + 1 if k <= z_size % n else base
        out.append((z, z + d))
        z += d
    return out


def compute_displacements(tiles, overlap_V: int, overlap_H: int, displ_max_V=S_DISPL_SEARCH_RADIUS_DEF,
                          displ_max_H=S_DISPL_SEARCH_RADIUS_DEF, displ_max_D=S_DISPL_SEARCH_RADIUS_DEF,
                          rank: int = 0, world_size: int = 1):
    """Pairwise displacement computation over one z-layer of a tile grid (step 2 of the stitcher).

    ``tiles[r][c]`` are device-resident float32 (D, V, H) tensors of identical shape.  Pairs are independent
    (StackStitcher.cpp:223-374; the reference farms them out over MPI ranks, Parastitcher.py:1440-1560): with
    ``world_size`` > 1 this rank takes pairs ``rank::world_size`` -- no collective is involved.
    Returns {(r, c, r_b, c_b, direction): DisplacementMIPNCC}."""
    capi.require_gpu()
    n_rows, n_cols = len(tiles), len(tiles[0])
    flat = [tiles[r][c] for r in range(n_rows) for c in range(n_cols)]
    dev = flat[0].device
    dim_D, dim_V, dim_H = (int(s) for s in flat[0].shape)
    for t in flat:
        if tuple(t.shape) != (dim_D, dim_V, dim_H) or t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
            raise ValueError("all tiles must be contiguous float32 tensors of one shape on one device")
    pairs = list(enumerate_pairs(n_rows, n_cols))[rank::world_size]
    n = len(pairs)
    if n == 0:
        return {}
    ptrs = (C.c_void_p * len(flat))(*[t.data_ptr() for t in flat])
    a_idx = (C.c_int * n)(*[r * n_cols + c for r, c, _, _, _ in pairs])
    b_idx = (C.c_int * n)(*[rb * n_cols + cb for _, _, rb, cb, _ in pairs])
    ni = (C.c_int * n)(*[dim_V - overlap_V if d == dir_vertical else 0 for *_, d in pairs])
    nj = (C.c_int * n)(*[dim_H - overlap_H if d == dir_horizontal else 0 for *_, d in pairs])
    side = (C.c_int * n)(*[d for *_, d in pairs])
    params = (NccParams * n)()
    for q in range(n):
        lib().mi_ncc_default_params(displ_max_V, displ_max_H, displ_max_D, C.byref(params[q]))
    inf_w = params[0].INF_W
    out = (NccDescr * n)()
    check(lib().mi_ncc_mips_batch(dev.index, capi.current_stream_ptr(dev), n, ptrs, a_idx, b_idx, dim_D, dim_V, dim_H,
                                  ni, nj, displ_max_D, displ_max_V, displ_max_H, side, params, out))
    res = {}
    for q, key in enumerate(pairs):
        d, p = out[q], params[q]
        res[key] = DisplacementMIPNCC(list(d.coord), [float(v) for v in d.NCC_maxs], list(d.NCC_widths),
                                      [displ_max_V, displ_max_H, displ_max_D],
                                      [p.wRangeThr_i, p.wRangeThr_j, p.wRangeThr_k], [inf_w] * 3)
    return res


def compute_mips(A, B, ni, nj, side):
    """compute_3_MIPs on the overlap views (compute_funcs.cu:502-521): returns the six MIPs as CUDA tensors
    [xy1, xz1, yz1, xy2, xz2, yz2] (xz/yz stored [i][k] / [j][k])."""
    capi.require_gpu()
    dev = A.device if isinstance(A, torch.Tensor) and A.is_cuda else torch.device("cuda", torch.cuda.current_device())
    a, b = _dev_tensor(A, dev), _dev_tensor(B, dev)
    dk, di, dj = (int(s) for s in a.shape)
    iv = di - ni if side == NORTH_SOUTH else di
    jv = dj - nj if side == WEST_EAST else dj
    shapes = [(iv, jv), (iv, dk), (jv, dk)] * 2
    outs = [torch.empty(s, dtype=torch.float32, device=dev) for s in shapes]
    check(lib().mi_ncc_compute_mips(dev.index, capi.current_stream_ptr(dev), a.data_ptr(), b.data_ptr(), dk, di, dj,
                                    int(ni), int(nj), int(side), *[o.data_ptr() for o in outs]))
    return outs


def compute_NCC_map(MIP_1, MIP_2, delayu, delayv):
    """compute_NCC_map (compute_funcs.cu:939): (2*delayu+1, 2*delayv+1) float32 CUDA tensor."""
    capi.require_gpu()
    dev = MIP_1.device if isinstance(MIP_1, torch.Tensor) and MIP_1.is_cuda else torch.device("cuda", torch.cuda.current_device())
    m1, m2 = _dev_tensor(MIP_1, dev), _dev_tensor(MIP_2, dev)
    dimu, dimv = (int(s) for s in m1.shape)
    out = torch.empty((2 * delayu + 1, 2 * delayv + 1), dtype=torch.float32, device=dev)
    check(lib().mi_ncc_compute_map(dev.index, capi.current_stream_ptr(dev), m1.data_ptr(), m2.data_ptr(), dimu, dimv,
                                   int(delayu), int(delayv), out.data_ptr()))
    return out
