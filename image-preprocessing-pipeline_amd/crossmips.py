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
import functools
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


S_NCC_WIDTH_MAX = 30        # S_config.h:86
S_NCC_PEAK_WEIGHT = 0.5     # S_config.h:87
S_NCC_WIDTH_WEIGHT = 0.5    # S_config.h:88
_INT_MAX = 2 ** 31 - 1


@dataclass
class DisplacementMIPNCC:
    """The displacement record of a pair of adjacent stacks (stitcher/DisplacementMIPNCC.{h,cpp}): what
    PDAlgoMIPNCC::execute stores (offsets (V,H,D), peak values, half widths + the search parameters,
    PDAlgoMIPNCC.cpp:100-109) and the steps that follow the pairwise computation: reliability, combination of the
    per-layer records, thresholding, XML."""
    VHD_coords: list
    NCC_maxs: list
    NCC_widths: list
    delays: list
    wRangeThrs: list
    invWidths: list
    VHD_def_coords: list = field(default_factory=lambda: [_INT_MAX] * 3)   # Displacement.cpp:40
    rel_factors: list = field(default_factory=lambda: [-1.0] * 3)           # DisplacementMIPNCC.cpp:72
    extra: dict = field(default_factory=dict)

    @classmethod
    def nominal(cls, V, H, D):
        """DisplacementMIPNCC(int Vnominal, int Hnominal, int Dnominal) (DisplacementMIPNCC.cpp:81-98)."""
        return cls([V, H, D], [0.0] * 3, [S_NCC_WIDTH_MAX] * 3, [-1] * 3, [S_NCC_WIDTH_MAX - 1] * 3, [S_NCC_WIDTH_MAX] * 3,
                   [V, H, D], [0.0] * 3)

    def evalReliability(self, axis: int) -> float:
        """DisplacementMIPNCC.cpp:130-147: ``sqrt(0.5*((100 - w*100/invW)/100)^2 + 0.5*peak^2)`` -- the width term in
        float, the weighted sum and the root in double (the weights are double literals), the result cast to float."""
        if axis not in (0, 1, 2):
            raise ValueError("in DisplacementMIPNCC::evalReliability(...): wrong direction value")
        f = np.float32
        wn = (f(100.0) - (f(self.NCC_widths[axis]) * f(100.0) / f(self.invWidths[axis]))) / f(100.0)
        peak = f(self.NCC_maxs[axis])
        r = math.sqrt(S_NCC_WIDTH_WEIGHT * float(wn) * float(wn) + S_NCC_PEAK_WEIGHT * float(peak) * float(peak))
        self.rel_factors[axis] = float(f(r))
        return self.rel_factors[axis]

    def getReliability(self, axis: int) -> float:
        if axis not in (0, 1, 2):
            raise ValueError("in DisplacementMIPNCC::evalReliability(...): wrong direction value")
        if self.rel_factors[axis] == -1.0:
            raise RuntimeError("in DisplacementMIPNCC::evalReliability(direction _direction): reliability factor not yet computed")
        return self.rel_factors[axis]

    _PER_AXIS = ("rel_factors", "NCC_maxs", "NCC_widths", "VHD_coords", "VHD_def_coords", "delays", "wRangeThrs", "invWidths")

    def combine(self, other: "DisplacementMIPNCC"):
        """DisplacementMIPNCC::combine (:312-346): per axis the more reliable of the two records wins and is copied into
        the other, so both are equal afterwards (ties keep ``self``)."""
        for k in range(3):
            self.evalReliability(k)
            other.evalReliability(k)
            src, dst = (other, self) if self.rel_factors[k] < other.rel_factors[k] else (self, other)
            for name in self._PER_AXIS:
                getattr(dst, name)[k] = getattr(src, name)[k]

    def isBetter(self, other: "DisplacementMIPNCC") -> bool:
        """:349-364: true when ``other``'s reliabilities sum to at least this record's."""
        cur = sum(np.float32(self.evalReliability(k)) for k in range(3))
        oth = sum(np.float32(other.evalReliability(k)) for k in range(3))
        return bool(oth >= cur)

    def threshold(self, rel_threshold: float):
        """:217-235: axes whose reliability is below the threshold fall back to the default (stage) displacement."""
        for k in range(3):
            self.evalReliability(k)
        for k in range(3):
            if self.rel_factors[k] < np.float32(rel_threshold):
                self.VHD_coords[k] = self.VHD_def_coords[k]
                self.NCC_maxs[k] = 0.0
                self.NCC_widths[k] = self.invWidths[k]
                self.evalReliability(k)

    def getMirrored(self, direction: int = -1):
        """:237-306; ``direction`` -1 = dir_all (what VirtualVolume::insertDisplacement stores on the other stack)."""
        m = DisplacementMIPNCC(list(self.VHD_coords), list(self.NCC_maxs), list(self.NCC_widths), list(self.delays),
                               list(self.wRangeThrs), list(self.invWidths), list(self.VHD_def_coords), list(self.rel_factors))
        if direction not in (-1, 0, 1, 2):
            raise ValueError("in DisplacementMIPNCC::getMirrored(...): unsupported or wrong given mirroring direction")
        for k in range(3):
            if direction == -1 or direction == k:
                m.VHD_coords[k] = -self.VHD_coords[k]
                m.VHD_def_coords[k] = -self.VHD_def_coords[k]
        return m

    def getXML(self):
        """<Displacement TYPE="MIP_NCC"> with V/H/D children (DisplacementMIPNCC::getXML, :367-400)."""
        import xml.etree.ElementTree as ET
        e = ET.Element("Displacement", TYPE="MIP_NCC")
        for i, name in enumerate("VHD"):
            ET.SubElement(e, name, displ=str(self.VHD_coords[i]), default_displ=str(self.VHD_def_coords[i]),
                          reliability=repr(float(self.rel_factors[i])), nccPeak=repr(float(self.NCC_maxs[i])),
                          nccWidth=str(self.NCC_widths[i]), nccWRangeThr=str(self.wRangeThrs[i]),
                          nccInvWidth=str(self.invWidths[i]), delay=str(self.delays[i]))
        return e

    @classmethod
    def loadXML(cls, node):
        """DisplacementMIPNCC::loadXML (:401-434) incl. the defaults of records written before 2013."""
        d = cls([_INT_MAX] * 3, [0.0] * 3, [0] * 3, [-1] * 3, [-1] * 3, [-1] * 3)
        for i, name in enumerate("VHD"):
            e = node.find(name)
            if e is None:
                raise ValueError(f"Displacement record without <{name}>")
            d.VHD_coords[i] = int(e.get("displ"))
            d.VHD_def_coords[i] = int(e.get("default_displ"))
            d.rel_factors[i] = float(e.get("reliability"))
            d.NCC_maxs[i] = float(e.get("nccPeak"))
            d.NCC_widths[i] = int(e.get("nccWidth"))
            d.wRangeThrs[i] = int(e.get("nccWRangeThr")) if e.get("nccWRangeThr") is not None else S_NCC_WIDTH_MAX - 1
            d.invWidths[i] = int(e.get("nccInvWidth")) if e.get("nccInvWidth") is not None else S_NCC_WIDTH_MAX
            d.delays[i] = int(e.get("delay")) if e.get("delay") is not None else -1
        return d


def project_displacements(displacements):
    """Displacement::projectDisplacements (Displacement.cpp:84-106): the per-layer records of one pair are combined pairwise,
    front to back; the last one carries the result."""
    if not displacements:
        raise ValueError("in Displacement::projectDisplacements(...): the given vector of displacements is EMPTY. Nothing to project.")
    for i in range(len(displacements) - 1):
        displacements[i].combine(displacements[i + 1])
    return displacements[-1]


def threshold_displacements(grid_pairs, n_rows, n_cols, rel_threshold):
    """StackStitcher::thresholdDisplacements (StackStitcher.cpp:1626-1720) on the projected records: ``grid_pairs`` maps
    (rowA, colA, rowB, colB) -> the ONE record of an adjacent pair (east and south neighbours).  Thresholds every record and
    returns the stitchable flag of every stack: at least one single-direction displacement of one of its pairs is reliable."""
    for r in range(n_rows):
        for c in range(n_cols):
            if c + 1 < n_cols and (r, c, r, c + 1) not in grid_pairs or r + 1 < n_rows and (r, c, r + 1, c) not in grid_pairs:
                raise ValueError("in StackStitcher::thresholdDisplacements(...): one and only displacement must exist for each "
                                 "pair of adjacent stacks.")
    for d in grid_pairs.values():
        d.threshold(rel_threshold)
    stitchable = {}
    thr = np.float32(rel_threshold)
    for r in range(n_rows):
        for c in range(n_cols):
            mine = [grid_pairs[k] for k in ((r - 1, c, r, c), (r, c, r, c + 1), (r, c, r + 1, c), (r, c - 1, r, c)) if k in grid_pairs]
            stitchable[(r, c)] = any(np.float32(d.getReliability(k)) >= thr for d in mine for k in range(3))
    return stitchable


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
        out = DisplacementMIPNCC(list(d.coord), [float(v) for v in d.NCC_maxs], list(d.NCC_widths),
                                 [displ_max_V, displ_max_H, displ_max_D],
                                 [params.wRangeThr_i, params.wRangeThr_j, params.wRangeThr_k], [params.INF_W] * 3)
        # VirtualVolume::insertDisplacement (vmVirtualVolume.cpp:279-306): reliabilities evaluated, default = the stage offset
        # along the overlap direction (the nominal offset the search was centred on), 0 elsewhere
        for k in range(3):
            out.evalReliability(k)
        out.VHD_def_coords = [dim_V - overlap if overlap_direction == dir_vertical else 0,
                              dim_H - overlap if overlap_direction == dir_horizontal else 0, 0]
        return out


def ncc_stats(reset=False) -> dict:
    """Counters of the library (mi_ncc_stats): pairs finished by the batched pipeline / by the per-pair path, and map entries
    recomputed in the reference's two-pass form to take decisions that were inside the resolution of the map values."""
    out = (C.c_longlong * 3)()
    lib().mi_ncc_stats(out, 1 if reset else 0)
    return {"pairs_batched": int(out[0]), "pairs_per_pair_path": int(out[1]), "entries_recomputed_exactly": int(out[2])}


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


def tile_row_blocks(n_rows: int, world_size: int, n_cols: int | None = None):
    """Partition of a tile grid over ranks by ROW BLOCKS: rank q owns the rows [r0, r1) -- all their west-east pairs and the
    north-south pairs (r, r + 1) for r in [r0, r1) -- and therefore keeps the rows [r0, min(r1 + 1, n_rows)) resident: every tile
    is uploaded once, the first row behind a cut twice.  (The reference farms (pair, layer) jobs out over MPI ranks with
    CUDA_VISIBLE_DEVICES = rank % num_gpus, each job re-reading its two stacks: Parastitcher.py:1367,1440-1560,
    StackStitcher.cpp:223-374.)  Blocks are contiguous and balanced by PAIR count (a row costs n_cols - 1 west-east + n_cols
    north-south pairs, the last row only the former): the cut that minimises the largest block.  Returns [(r0, r1)] per rank;
    ranks beyond the rows get empty blocks."""
    world_size = max(1, int(world_size))
    n_cols = n_rows if n_cols is None else int(n_cols)
    cost = [(n_cols - 1) + (n_cols if r + 1 < n_rows else 0) for r in range(n_rows)]
    parts = min(world_size, n_rows)

    def cuts(limit):                      # greedy: fewest blocks whose cost stays <= limit
        out, r0, acc = [], 0, 0
        for r, c in enumerate(cost):
            if acc + c > limit and r > r0:
                out.append((r0, r))
                r0, acc = r, 0
            acc += c
        out.append((r0, n_rows))
        return out

    lo, hi = max(cost) if cost else 0, sum(cost)
    while lo < hi:                        # smallest limit that needs at most `parts` blocks
        mid = (lo + hi) // 2
        if len(cuts(mid)) <= parts:
            hi = mid
        else:
            lo = mid + 1
    blocks = cuts(lo) if n_rows else []
    return blocks + [(n_rows, n_rows)] * (world_size - len(blocks))


@functools.lru_cache(maxsize=64)
def _pair_list(n_rows, n_cols, row_block, rank, world_size):
    """The pairs one rank computes and the tiles they touch (see compute_displacements)."""
    pairs = list(enumerate_pairs(n_rows, n_cols))
    if row_block is not None:
        pairs = [p for p in pairs if row_block[0] <= p[0] < row_block[1]]
    else:
        pairs = pairs[rank::world_size]
    used = sorted({r * n_cols + c for r, c, _, _, _ in pairs} | {rb * n_cols + cb for _, _, rb, cb, _ in pairs})
    return tuple(pairs), tuple(used)


@functools.lru_cache(maxsize=64)
def _pair_arrays(n_rows, n_cols, row_block, rank, world_size, ni_v, nj_h):
    """ctypes arrays of that pair list: tile indices, the extents a pair drops (ni for vertical, nj for horizontal pairs), sides."""
    pairs, _ = _pair_list(n_rows, n_cols, row_block, rank, world_size)
    n = len(pairs)
    ni_l = [ni_v if d == dir_vertical else 0 for *_, d in pairs]
    nj_l = [nj_h if d == dir_horizontal else 0 for *_, d in pairs]
    return ((C.c_int * n)(*[r * n_cols + c for r, c, _, _, _ in pairs]), (C.c_int * n)(*[rb * n_cols + cb for _, _, rb, cb, _ in pairs]),
            (C.c_int * n)(*ni_l), (C.c_int * n)(*nj_l), (C.c_int * n)(*[d for *_, d in pairs]),
            tuple((a, b, 0) for a, b in zip(ni_l, nj_l)))


def compute_displacements_begin(tiles, overlap_V: int, overlap_H: int, displ_max_V=S_DISPL_SEARCH_RADIUS_DEF,
                          displ_max_H=S_DISPL_SEARCH_RADIUS_DEF, displ_max_D=S_DISPL_SEARCH_RADIUS_DEF,
                          rank: int = 0, world_size: int = 1, row_block=None, sample_scale=None):
    """``compute_displacements`` in two halves (``mi_ncc_mips_batch_begin`` / ``_end``): enqueues the device stage of the layer's pairs
    and returns a ``PendingDisplacements``; its ``result()`` waits, runs the host rules and returns the dictionary.  A caller that
    walks the z layers of a grid begins layer l + 1 before it takes the result of layer l (``compute_displacements_layers``): the
    next layer's first MIP pass then runs beside this layer's last lag chain.  The tiles must stay unchanged until ``result()``.

    Pairwise displacement computation over one z-layer of a tile grid (step 2 of the stitcher).

    ``tiles[r][c]`` are device-resident float32 (D, V, H) tensors of identical shape -- or uint16 / uint8 tensors holding the samples
    the reference would have divided by ``sample_scale`` when it loaded the tiles (default 65535 / 255: tiff2D.cpp:606-610): the
    records are identical, the MIP pass reads half / a quarter of the bytes (``mi_ncc_mips_batch_u16`` / ``_u8``; tiles whose rows are
    no whole 32-bit words or that have more than 32 slices are converted on the device first).  Pairs are independent
    (StackStitcher.cpp:223-374; the reference farms them out over MPI ranks, Parastitcher.py:1440-1560) -- no collective is
    involved.  Two ways to share a grid among ranks: ``row_block = (r0, r1)`` (see ``tile_row_blocks``): this rank computes the
    pairs that start in rows [r0, r1), and only the rows r0 .. min(r1, n_rows - 1) of ``tiles`` need to hold tensors (the others
    may be None); or, with every tile resident on every rank, ``rank`` / ``world_size``: pairs ``rank::world_size``.
    Returns {(r, c, r_b, c_b, direction): DisplacementMIPNCC}."""
    capi.require_gpu()
    n_rows, n_cols = len(tiles), len(tiles[0])
    flat = [tiles[r][c] for r in range(n_rows) for c in range(n_cols)]
    pairs, used = _pair_list(n_rows, n_cols, None if row_block is None else (int(row_block[0]), int(row_block[1])), int(rank), int(world_size))
    n = len(pairs)
    if n == 0:
        return PendingDisplacements(None)
    if any(flat[i] is None for i in used):
        raise ValueError("a tile of a pair this rank computes is not resident (None)")
    first = flat[used[0]]
    dev = first.device
    dim_D, dim_V, dim_H = (int(s) for s in first.shape)
    for i in used:
        t = flat[i]
        if tuple(t.shape) != (dim_D, dim_V, dim_H) or t.dtype != first.dtype or not t.is_contiguous() or t.device != dev:
            raise ValueError("all tiles must be contiguous float32 (or uint16) tensors of one shape on one device")
    if first.dtype not in (torch.float32, torch.uint16, torch.uint8):
        raise TypeError("stacks must be float32 (iom::real_t) or uint16 / uint8 samples")
    as_int = {torch.uint16: 2, torch.uint8: 1}.get(first.dtype, 0)
    if as_int and sample_scale is None:
        sample_scale = 65535.0 if as_int == 2 else 255.0
    if as_int and (dim_H % (4 // as_int) or dim_D > 32):
        # what the integer MIP kernels do not take is converted the way the reference converts it when it loads a stack: a true
        # float32 division, element by element (dividing by a Python scalar, torch multiplies by the reciprocal: 1 ulp off in places)
        div = torch.full((), float(sample_scale), dtype=torch.float32, device=dev)
        conv = {i: torch.div(flat[i].to(torch.float32), div) for i in used}
        flat = [conv.get(i) for i in range(len(flat))]
        as_int = 0
    ptrs = (C.c_void_p * len(flat))(*[(t.data_ptr() if t is not None else None) for t in flat])
    # (index and extent arrays of the pair list: built once per grid shape -- the callee only reads them)
    a_idx, b_idx, ni, nj, side, ninj = _pair_arrays(n_rows, n_cols, None if row_block is None else (int(row_block[0]), int(row_block[1])),
                                                    int(rank), int(world_size), dim_V - overlap_V, dim_H - overlap_H)
    p0 = NccParams()
    lib().mi_ncc_default_params(displ_max_V, displ_max_H, displ_max_D, C.byref(p0))
    inf_w = p0.INF_W
    params = (NccParams * n).from_buffer_copy(bytes(p0) * n)   # one parameter block per pair (the callee clamps wRangeThr_* in place)
    job = C.c_void_p()
    check(lib().mi_ncc_mips_batch_begin(dev.index, capi.current_stream_ptr(dev), n, ptrs, {0: 4, 2: 2, 1: 1}[as_int], float(sample_scale or 1.0), a_idx, b_idx,
                                        dim_D, dim_V, dim_H, ni, nj, displ_max_D, displ_max_V, displ_max_H, side, params, C.byref(job)))
    return PendingDisplacements(job, n=n, pairs=pairs, params=params, keep=(flat, ptrs, a_idx, b_idx, ni, nj, side), inf_w=inf_w, ninj=ninj,
                                delays=[displ_max_V, displ_max_H, displ_max_D])


class PendingDisplacements:
    """A layer's batch under way on the device (``compute_displacements_begin``)."""

    def __init__(self, job, **kw):
        self.job = job
        self.__dict__.update(kw)
        self._res = {} if job is None else None

    def __del__(self):                      # a batch nobody asked the result of is still ended: its device stage holds buffers
        try:
            if self._res is None and self.job is not None and self.job.value:
                lib().mi_ncc_mips_batch_end(self.job, None, (NccDescr * self.n)())
        except Exception:
            pass

    def result(self):
        if self._res is None:
            self._res = self._finish()
        return self._res

    def _finish(self):
        n, pairs, params, inf_w, ninj, delays = self.n, self.pairs, self.params, self.inf_w, self.ninj, self.delays
        out = (NccDescr * n)()
        job, self.job = self.job, None
        check(lib().mi_ncc_mips_batch_end(job, params, out))
        self.keep = None
        # the records of all pairs at once: NccDescr = 3 ints, 3 floats, 3 ints; evalReliability (DisplacementMIPNCC.cpp:130-147)
        # vectorised with the same float / double steps as the per-record method
        raw = np.frombuffer(out, dtype=np.int32).reshape(n, 9)
        coords, widths = raw[:, 0:3], raw[:, 6:9]
        peaks = raw[:, 3:6].copy().view(np.float32)
        f = np.float32
        wn = (f(100.0) - (widths.astype(f) * f(100.0) / f(inf_w))) / f(100.0)
        with np.errstate(invalid="ignore"):
            rel = np.sqrt(S_NCC_WIDTH_WEIGHT * wn.astype(np.float64) ** 2 + S_NCC_PEAK_WEIGHT * peaks.astype(np.float64) ** 2).astype(f)
        praw = np.frombuffer(params, dtype=np.int32).reshape(n, C.sizeof(NccParams) // 4)
        o_i = NccParams.wRangeThr_i.offset // 4
        thr = praw[:, o_i:o_i + 3]
        res = {}
        import gc
        gc_on = gc.isenabled()
        gc.disable()    # a few thousand small containers: the cyclic collector's passes over a torch-sized heap cost 0.1 ms per record
        try:
            cl, pl, wl, tl, rl = coords.tolist(), peaks.astype(np.float64).tolist(), widths.tolist(), thr.tolist(), rel.astype(np.float64).tolist()
            new = object.__new__   # (the dataclass constructor with its default factories costs 4 us per record; 112 records a call)
            for q, key in enumerate(pairs):
                d = new(DisplacementMIPNCC)
                d.__dict__ = {"VHD_coords": cl[q], "NCC_maxs": pl[q], "NCC_widths": wl[q], "delays": list(delays), "wRangeThrs": tl[q],
                              "invWidths": [inf_w] * 3, "VHD_def_coords": list(ninj[q]), "rel_factors": rl[q], "extra": {}}   # vmVirtualVolume.cpp:279-306
                res[key] = d
        finally:
            if gc_on:
                gc.enable()
        return res


def compute_displacements(tiles, overlap_V: int, overlap_H: int, displ_max_V=S_DISPL_SEARCH_RADIUS_DEF,
                          displ_max_H=S_DISPL_SEARCH_RADIUS_DEF, displ_max_D=S_DISPL_SEARCH_RADIUS_DEF,
                          rank: int = 0, world_size: int = 1, row_block=None, sample_scale=None):
    """Pairwise displacement computation over one z-layer of a tile grid (step 2 of the stitcher): ``compute_displacements_begin``
    and its result at once.  Returns {(r, c, r_b, c_b, direction): DisplacementMIPNCC}."""
    return compute_displacements_begin(tiles, overlap_V, overlap_H, displ_max_V, displ_max_H, displ_max_D, rank, world_size, row_block,
                                       sample_scale).result()


def compute_displacements_layers(layers, overlap_V: int, overlap_H: int, *args, **kw):
    """The layers of a grid one after the other (``layers``: an iterable of tile grids, one per z layer: StackStitcher.cpp:223-374),
    one batch ahead: the device stage of layer l + 1 is enqueued before the host takes the result of layer l.  Yields the
    dictionaries of ``compute_displacements`` in order."""
    prev = None
    for tiles in layers:
        cur = compute_displacements_begin(tiles, overlap_V, overlap_H, *args, **kw)
        if prev is not None:
            yield prev.result()
        prev = cur
    if prev is not None:
        yield prev.result()


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


def compute_NCC_map(MIP_1, MIP_2, delayu, delayv, lag=False):
    """compute_NCC_map (compute_funcs.cu:939): (2*delayu+1, 2*delayv+1) float32 CUDA tensor.  ``lag``: cross terms through the
    lag transform of the batched pair pipeline instead of shift by shift."""
    capi.require_gpu()
    dev = MIP_1.device if isinstance(MIP_1, torch.Tensor) and MIP_1.is_cuda else torch.device("cuda", torch.cuda.current_device())
    m1, m2 = _dev_tensor(MIP_1, dev), _dev_tensor(MIP_2, dev)
    dimu, dimv = (int(s) for s in m1.shape)
    out = torch.empty((2 * delayu + 1, 2 * delayv + 1), dtype=torch.float32, device=dev)
    fn = lib().mi_ncc_compute_map_lag if lag else lib().mi_ncc_compute_map
    check(fn(dev.index, capi.current_stream_ptr(dev), m1.data_ptr(), m2.data_ptr(), dimu, dimv, int(delayu), int(delayv),
             out.data_ptr()))
    return out
