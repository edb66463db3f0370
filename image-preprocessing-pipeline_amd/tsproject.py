"""TeraStitcher project files (``xml_import`` -> ``xml_displcomp`` -> ``xml_displproj`` -> ``xml_displthres``) for stitching
steps 2-4: the volume descriptor the reference's ``terastitcher -1`` writes and ``terastitcher -5``
reads, with the per-stack NORTH / EAST / SOUTH / WEST displacement lists in between.  Host-side bookkeeping; the pairwise
computation itself is ``crossmips.compute_displacements`` (GPU).

  Project.load / save            StackedVolume::initFromXML / saveXML (vmStackedVolume.cpp:661-746, 748-870), Stack::getXML /
                                 loadXML (vmStack.cpp:356-399, 401-548); format id "TiledXY|2Dseries" (vmStackedVolume.cpp:96)
  insertDisplacement             VirtualVolume::insertDisplacement (vmVirtualVolume.cpp:280-314)
  adjustDisplacements            VirtualVolume::adjustDisplacements (vmVirtualVolume.cpp:317-345), applied after loading
  computeDisplacements           StackStitcher::computeDisplacements (StackStitcher.cpp:128-400): z layers, pair loop
  projectDisplacements           StackStitcher::projectDisplacements (StackStitcher.cpp:1563-1624)
  thresholdDisplacements         StackStitcher::thresholdDisplacements (StackStitcher.cpp:1626-1720)

Numbers are written the way TinyXML writes them: integers with %d, doubles with %g (tinyxml.cpp:1216-1225), so reliabilities
and NCC peaks carry six significant digits from one step's file to the next, exactly as between the reference's own steps.
Slices are 2-D TIFFs in ``stacks_dir/DIR_NAME`` (filtered by IMG_REGEX, sorted by name), scaled to [0, 1] like
loadImageStack (tiff2D.cpp:606-610).
"""
from __future__ import annotations

import math
import os
import re
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

from .crossmips import (DisplacementMIPNCC, S_DISPL_SEARCH_RADIUS_DEF, S_SUBVOL_DIM_D_DEFAULT, dir_horizontal, dir_vertical,
                        project_displacements, subvolume_layers)

FORMAT_ID = "TiledXY|2Dseries"      # vmStackedVolume.cpp:96
S_OVERLAP_MIN = 1                   # S_config.h
_SIDES = ("NORTH", "EAST", "SOUTH", "WEST")


def _g(v) -> str:
    return "%g" % float(v)


def _displacement_xml(d: DisplacementMIPNCC):
    """DisplacementMIPNCC::getXML (DisplacementMIPNCC.cpp:367-400) with TinyXML's number formatting."""
    e = ET.Element("Displacement", TYPE="MIP_NCC")
    for i, name in enumerate("VHD"):
        ET.SubElement(e, name, displ="%d" % d.VHD_coords[i], default_displ="%d" % d.VHD_def_coords[i],
                      reliability=_g(d.rel_factors[i]), nccPeak=_g(d.NCC_maxs[i]), nccWidth="%d" % d.NCC_widths[i],
                      nccWRangeThr="%d" % d.wRangeThrs[i], nccInvWidth="%d" % d.invWidths[i], delay="%d" % d.delays[i])
    return e


@dataclass
class Stack:
    """One tile (vmStack.h): grid position, stage position in voxels, its slice files and the displacement lists."""
    ROW_INDEX: int
    COL_INDEX: int
    DIR_NAME: str
    ABS_V: int = 0
    ABS_H: int = 0
    ABS_D: int = 0
    N_CHANS: int = 1
    N_BYTESxCHAN: int = 1
    stitchable: bool = False
    z_ranges: list = field(default_factory=list)     # [(start, end)) intervals with data (sparse tiles)
    img_regex: str = ""
    NORTH: list = field(default_factory=list)
    EAST: list = field(default_factory=list)
    SOUTH: list = field(default_factory=list)
    WEST: list = field(default_factory=list)

    def isComplete(self, z0: int, z1: int) -> bool:
        """vmVirtualStack: every slice of [z0, z1] (inclusive) is present."""
        return any(a <= z0 and z1 < b for a, b in self.z_ranges)

    def getXML(self):
        e = ET.Element("Stack", N_CHANS="%d" % self.N_CHANS, N_BYTESxCHAN="%d" % self.N_BYTESxCHAN, ROW="%d" % self.ROW_INDEX,
                       COL="%d" % self.COL_INDEX, ABS_V="%d" % self.ABS_V, ABS_H="%d" % self.ABS_H, ABS_D="%d" % self.ABS_D,
                       STITCHABLE="yes" if self.stitchable else "no", DIR_NAME=self.DIR_NAME,
                       Z_RANGES=";".join("[%d,%d)" % r for r in self.z_ranges), IMG_REGEX=self.img_regex)
        for side in _SIDES:
            lst = ET.SubElement(e, f"{side}_displacements")
            for d in getattr(self, side):
                lst.append(_displacement_xml(d))
        return e

    @classmethod
    def loadXML(cls, node, row, col, z_end):
        s = cls(row, col, node.get("DIR_NAME"))
        s.N_CHANS, s.N_BYTESxCHAN = int(node.get("N_CHANS", -1)), int(node.get("N_BYTESxCHAN", -1))
        if s.N_CHANS == -1 or s.N_BYTESxCHAN == -1:      # old xml file (vmStack.cpp:413-418)
            s.N_CHANS = s.N_BYTESxCHAN = 1
        s.ABS_V, s.ABS_H, s.ABS_D = (int(node.get(k)) for k in ("ABS_V", "ABS_H", "ABS_D"))
        s.stitchable = node.get("STITCHABLE") == "yes"
        s.img_regex = node.get("IMG_REGEX") or ""
        zr = node.get("Z_RANGES")
        if zr is None:
            s.z_ranges = [(0, z_end)]
        else:
            for tok in re.sub(r"\s", "", zr).split(";"):
                if not tok:
                    continue
                m = re.fullmatch(r"\[(-?\d+),(-?\d+)\)", tok)
                if not m:
                    raise ValueError(f"in Stack({s.DIR_NAME})::loadXML(): cannot parse 'Z_RANGES' subentry \"{tok}\"")
                a, b = int(m.group(1)), int(m.group(2))
                if a < 0 or a >= b or (z_end > 0 and b > z_end):
                    raise ValueError(f"in Stack({s.DIR_NAME})::loadXML(): 'Z_RANGES' subentry \"{tok}\" is out of range [0,{z_end}) ")
                s.z_ranges.append((a, b))
            for (a0, b0), (a1, b1) in zip(s.z_ranges, s.z_ranges[1:]):
                if a1 <= b0:
                    raise ValueError(f"in Stack({s.DIR_NAME})::loadXML(): wrong sequence in 'Z_RANGES' attribute.")
        for side in _SIDES:
            lst = node.find(f"{side}_displacements")
            if lst is not None:
                getattr(s, side).extend(DisplacementMIPNCC.loadXML(d) for d in lst.findall("Displacement"))
        return s


class Project:
    """The stitching project: volume descriptor + STACKS[row][col] (vmStackedVolume.h, vmVirtualVolume.h)."""

    def __init__(self, stacks_dir, n_rows, n_cols, n_slices, VXL=(1.0, 1.0, 1.0), ORG=(0.0, 0.0, 0.0), MEC=(0.0, 0.0),
                 ref_sys=(1, 2, 3), input_plugin="tiff2D", mdata_bin=None):
        self.stacks_dir = str(stacks_dir)
        self.N_ROWS, self.N_COLS, self.N_SLICES = int(n_rows), int(n_cols), int(n_slices)
        self.VXL_V, self.VXL_H, self.VXL_D = (float(np.float32(v)) for v in VXL)     # QueryFloatAttribute: float members
        self.ORG_V, self.ORG_H, self.ORG_D = (float(np.float32(v)) for v in ORG)
        self.MEC_V, self.MEC_H = (float(np.float32(v)) for v in MEC)
        self.ref_sys = tuple(int(v) for v in ref_sys)
        self.input_plugin = input_plugin
        self.mdata_bin = mdata_bin
        self.STACKS = [[None] * self.N_COLS for _ in range(self.N_ROWS)]
        self._dims = None

    # ---- geometry (vmVirtualVolume.cpp:75-79)
    def _stack_dims(self):
        if self._dims is None:
            from PIL import Image
            with Image.open(self.slice_files(self.STACKS[0][0])[0]) as im:
                self._dims = (im.height, im.width)
        return self._dims

    def getStacksHeight(self): return self._stack_dims()[0]
    def getStacksWidth(self): return self._stack_dims()[1]
    # MEC / VXL are float members and the quotient is a single-precision division (57.6f / 0.8f == 72.0f, while the same
    # quotient in double is 71.99999999999999 and truncates to 71) -- pinned by tests/golden/terastitcher (the reference binary)
    @staticmethod
    def _fdiv(a, b): return np.float32(a) / np.float32(b)
    def getOVERLAP_V(self): return int(np.float32(self.getStacksHeight()) - self._fdiv(self.MEC_V, self.VXL_V))
    def getOVERLAP_H(self): return int(np.float32(self.getStacksWidth()) - self._fdiv(self.MEC_H, self.VXL_H))
    def getDEFAULT_DISPLACEMENT_V(self): return int(abs(self._fdiv(self.MEC_V, self.VXL_V)))
    def getDEFAULT_DISPLACEMENT_H(self): return int(abs(self._fdiv(self.MEC_H, self.VXL_H)))
    def getDEFAULT_DISPLACEMENT_D(self): return 0

    # ---- files
    @classmethod
    def load(cls, xml_filepath):
        try:
            root = ET.parse(xml_filepath).getroot()
        except (OSError, ET.ParseError) as e:
            raise ValueError(f"in StackedVolume::initFromXML(xml_filepath = \"{xml_filepath}\") : unable to load xml") from e
        if root.tag != "TeraStitcher":
            raise ValueError(f"{xml_filepath}: not a TeraStitcher project (root <{root.tag}>)")
        fmt = root.get("volume_format")
        if fmt and fmt != FORMAT_ID:
            raise ValueError(f"in StackedVolume::initFromXML(): unsupported volume_format = \"{fmt}\" (current format is \"{FORMAT_ID}\")")

        def need(tag):
            e = root.find(tag)
            if e is None:
                raise ValueError(f"{xml_filepath}: <{tag}> is missing")
            return e

        dims = need("dimensions")
        ref = root.find("ref_sys")
        vx, org, mec = need("voxel_dims"), need("origin"), need("mechanical_displacements")
        mdata = root.find("mdata_bin")
        p = cls(need("stacks_dir").get("value"), int(dims.get("stack_rows")), int(dims.get("stack_columns")),
                int(dims.get("stack_slices", 0)), [vx.get(k) for k in "VHD"], [org.get(k) for k in "VHD"],
                [mec.get(k) for k in "VH"], [int(ref.get(k)) for k in ("ref1", "ref2", "ref3")] if ref is not None else (1, 2, 3),
                root.get("input_plugin") or "tiff2D", mdata.get("value") if mdata is not None else None)
        nodes = need("STACKS").findall("Stack")
        if len(nodes) != p.N_ROWS * p.N_COLS:
            raise ValueError(f"{xml_filepath}: {len(nodes)} <Stack> nodes for a {p.N_ROWS} x {p.N_COLS} grid")
        it = iter(nodes)
        for i in range(p.N_ROWS):               # row-major, like the nested loop of initFromXML (:728-741)
            for j in range(p.N_COLS):
                p.STACKS[i][j] = Stack.loadXML(next(it), i, j, p.N_SLICES)
        if p.N_SLICES <= 0:                     # externally generated import file (:719-726): the stacks know
            p.N_SLICES = max(b for row in p.STACKS for s in row for _, b in s.z_ranges)
        p.adjustDisplacements()
        return p

    def save(self, xml_filepath):
        root = ET.Element("TeraStitcher", volume_format=FORMAT_ID, input_plugin=self.input_plugin)
        ET.SubElement(root, "stacks_dir", value=self.stacks_dir)
        if self.mdata_bin is not None:
            ET.SubElement(root, "mdata_bin", value=self.mdata_bin)
        ET.SubElement(root, "ref_sys", ref1="%d" % self.ref_sys[0], ref2="%d" % self.ref_sys[1], ref3="%d" % self.ref_sys[2])
        ET.SubElement(root, "voxel_dims", V=_g(self.VXL_V), H=_g(self.VXL_H), D=_g(self.VXL_D))
        ET.SubElement(root, "origin", V=_g(self.ORG_V), H=_g(self.ORG_H), D=_g(self.ORG_D))
        ET.SubElement(root, "mechanical_displacements", V=_g(self.MEC_V), H=_g(self.MEC_H))
        ET.SubElement(root, "dimensions", stack_rows="%d" % self.N_ROWS, stack_columns="%d" % self.N_COLS,
                      stack_slices="%d" % self.N_SLICES)
        stacks = ET.SubElement(root, "STACKS")
        for row in self.STACKS:
            for s in row:
                stacks.append(s.getXML())
        ET.indent(root, space="    ")
        body = ET.tostring(root, encoding="unicode")
        tmp = f"{xml_filepath}.tmp"
        with open(tmp, "w", encoding="utf-8") as f:        # header of saveXML (:789-790)
            f.write('<?xml version="1.0" encoding="UTF-8" ?>\n<!DOCTYPE TeraStitcher SYSTEM "TeraStitcher.DTD">\n')
            f.write(body + "\n")
        os.replace(tmp, xml_filepath)

    def slice_files(self, stk: Stack):
        folder = Path(self.stacks_dir) / stk.DIR_NAME
        rx = re.compile(stk.img_regex) if stk.img_regex else None
        files = sorted(f for f in folder.iterdir()
                       if f.suffix.lower() in (".tif", ".tiff") and (rx is None or rx.search(f.name)))
        if not files:
            raise RuntimeError(f"in Stack[{stk.ROW_INDEX},{stk.COL_INDEX}]: no TIFF slices in {folder}")
        return files

    def _read_slices(self, stk: Stack, z0: int, z1: int):
        """Raw slices [z0, z1] (inclusive) of a stack as one (D, V, H) uint8 / uint16 array."""
        from PIL import Image
        if not stk.isComplete(z0, z1):
            raise ValueError(f"in Stack[{stk.ROW_INDEX},{stk.COL_INDEX}]::loadImageStack: slices [{z0},{z1}] are not all present")
        files = self.slice_files(stk)
        # sparse tiles: file i holds the i-th z of the concatenated ranges (vmStack.cpp:476-497)
        zs = [z for a, b in stk.z_ranges for z in range(a, b)]
        if len(files) < len(zs):
            raise ValueError(f"in Stack({stk.DIR_NAME})::loadXML(): no more slices available to cover the z-ranges")
        index = {z: i for i, z in enumerate(zs)}
        # the library's reader (include/mi_tiffio.h: strips of raw / deflate samples, a slice per task on all cores) where it takes
        # the files; Pillow -- 65 MB/s, one slice after the other -- for the others and for every error message
        from . import brickio
        want = [files[index[z]] for z in range(z0, z1 + 1)]
        info = brickio.tiff_info(want[0])
        if info is not None and info[2] and info[1] in (np.uint8, np.uint16):
            try:
                return brickio.read_tiff_box(want, info[0], info[1], 0, info[0][0], 0, info[0][1])
            except Exception:
                pass
        out = None
        for k, z in enumerate(range(z0, z1 + 1)):
            a = np.asarray(Image.open(files[index[z]]))
            if a.ndim != 2 or a.dtype not in (np.uint8, np.uint16):
                raise TypeError(f"{files[index[z]]}: 8 or 16 bits per channel, single-channel slices are supported (tiff2D.cpp:600-612)")
            if out is None:
                out = np.empty((z1 - z0 + 1,) + a.shape, a.dtype)
            elif a.shape != out.shape[1:] or a.dtype != out.dtype:
                raise ValueError(f"{files[index[z]]}: slice shape / type differs from the first slice of the stack")
            out[k] = a
        return out

    def loadImageStack(self, stk: Stack, z0: int, z1: int):
        """Stack::loadImageStack(first, last) (vmStack.cpp:562-640): slices [z0, z1] inclusive as float32 (D, V, H) in [0, 1]
        (value / 255 or / 65535, tiff2D.cpp:606-610)."""
        raw = self._read_slices(stk, z0, z1)
        return raw.astype(np.float32) / np.float32(255.0 if raw.dtype == np.uint8 else 65535.0)

    def loadImageStackDevice(self, stk: Stack, z0: int, z1: int, device):
        """The same stack as a float32 CUDA tensor: the raw samples cross PCIe, the conversion (the same float32 division) runs
        on the device (``mi_load_block`` without padding)."""
        import torch
        from . import capi
        raw = self._read_slices(stk, z0, z1)
        d_raw = torch.from_numpy(raw.view(np.uint8).reshape(-1)).to(device)
        dst = torch.empty(raw.shape, dtype=torch.float32, device=device)
        dk, dv, dh = raw.shape
        capi.check(capi.lib().mi_load_block(device.index, capi.current_stream_ptr(device), d_raw.data_ptr(), raw.dtype.itemsize, dh, dv, dk,
                                            dst.data_ptr(), dh, dv, dk, 0, 0, 0))
        d_raw.record_stream(torch.cuda.current_stream(device))
        return dst

    def loadImageStackSamplesDevice(self, stk: Stack, z0: int, z1: int, device):
        """The same slices as the integer SAMPLES on the device and the divisor that turns them into the reference's floats
        (tiff2D.cpp:606-610): what ``crossmips.compute_displacements`` takes instead of float32 tiles -- a half / a quarter of the
        bytes across PCIe, in device memory and in the MIP pass, identical records.  8-bit tiles whose rows are no whole 32-bit
        words are widened to 16 bits (the 16-bit kernel needs an even width only)."""
        import torch
        raw = self._read_slices(stk, z0, z1)
        scale = 255.0 if raw.dtype == np.uint8 else 65535.0
        if raw.dtype == np.uint8 and raw.shape[2] % 4:
            raw = raw.astype(np.uint16)
        return torch.from_numpy(np.ascontiguousarray(raw)).to(device), scale

    # ---- displacement bookkeeping
    def insertDisplacement(self, stk_A: Stack, stk_B: Stack, d: DisplacementMIPNCC):
        for k in range(3):
            d.evalReliability(k)
        if stk_B.ROW_INDEX == stk_A.ROW_INDEX and stk_B.COL_INDEX == stk_A.COL_INDEX + 1:
            d.VHD_def_coords = [0, self.getDEFAULT_DISPLACEMENT_H(), self.getDEFAULT_DISPLACEMENT_D()]
            stk_A.EAST.append(d)
            stk_B.WEST.append(d.getMirrored(-1))
        elif stk_B.ROW_INDEX == stk_A.ROW_INDEX + 1 and stk_B.COL_INDEX == stk_A.COL_INDEX:
            d.VHD_def_coords = [self.getDEFAULT_DISPLACEMENT_V(), 0, self.getDEFAULT_DISPLACEMENT_D()]
            stk_A.SOUTH.append(d)
            stk_B.NORTH.append(d.getMirrored(-1))
        else:
            raise ValueError(f"in VirtualVolume::insertDisplacement(stk_A[{stk_A.ROW_INDEX},{stk_A.COL_INDEX}], "
                             f"stk_B[{stk_B.ROW_INDEX},{stk_B.COL_INDEX}], displacement): stacks are not adjacent")

    def adjustDisplacements(self):
        """WEST / NORTH lists rebuilt as the mirrors of the neighbours' EAST / SOUTH lists."""
        for i in range(self.N_ROWS):
            for j in range(self.N_COLS):
                a = self.STACKS[i][j]
                if j < self.N_COLS - 1:
                    self.STACKS[i][j + 1].WEST = [d.getMirrored(-1) for d in a.EAST]
                if i < self.N_ROWS - 1:
                    self.STACKS[i + 1][j].NORTH = [d.getMirrored(-1) for d in a.SOUTH]

    def projectDisplacements(self):
        V, H, D = self.getDEFAULT_DISPLACEMENT_V(), self.getDEFAULT_DISPLACEMENT_H(), self.getDEFAULT_DISPLACEMENT_D()
        for i in range(self.N_ROWS):
            for j in range(self.N_COLS):
                s = self.STACKS[i][j]
                for side, present, nominal in (("NORTH", i != 0, (-V, 0, D)), ("EAST", j != self.N_COLS - 1, (0, H, D)),
                                               ("SOUTH", i != self.N_ROWS - 1, (V, 0, D)), ("WEST", j != 0, (0, -H, D))):
                    if not present:
                        continue
                    lst = getattr(s, side)
                    setattr(s, side, [project_displacements(lst)] if lst else [DisplacementMIPNCC.nominal(*nominal)])

    def thresholdDisplacements(self, reliability_threshold: float):
        def sides(i, j):
            return (("NORTH", i != 0), ("EAST", j != self.N_COLS - 1), ("SOUTH", i != self.N_ROWS - 1), ("WEST", j != 0))
        for i in range(self.N_ROWS):
            for j in range(self.N_COLS):
                for side, present in sides(i, j):
                    if present and len(getattr(self.STACKS[i][j], side)) != 1:
                        raise ValueError("in StackStitcher::thresholdDisplacements(...): one and only displacement must exist for "
                                         "each pair of adjacent stacks.")
        for i in range(self.N_ROWS):
            for j in range(self.N_COLS):
                for side, present in sides(i, j):
                    if present:
                        getattr(self.STACKS[i][j], side)[0].threshold(reliability_threshold)
        thr = np.float32(reliability_threshold)
        for i in range(self.N_ROWS):
            for j in range(self.N_COLS):
                s = self.STACKS[i][j]
                s.stitchable = any(np.float32(getattr(s, side)[0].getReliability(k)) >= thr
                                   for side, present in sides(i, j) if present for k in range(3))

    def computeDisplacements(self, overlap_V=-1, overlap_H=-1, displ_max_V=S_DISPL_SEARCH_RADIUS_DEF,
                             displ_max_H=S_DISPL_SEARCH_RADIUS_DEF, displ_max_D=S_DISPL_SEARCH_RADIUS_DEF,
                             subvol_DIM_D=S_SUBVOL_DIM_D_DEFAULT, z0=-1, z1=-1, device=None, rank=0, world_size=1):
        """Step 2 over the whole grid: per z layer the tiles go to the device once and every east / south pair is computed
        there (``crossmips.compute_displacements``; pairs are independent, ``rank::world_size`` of them per process).
        Layers with an incomplete (sparse) tile skip that tile's pairs, like the isComplete() checks of the reference."""
        import torch
        from . import crossmips
        overlap_V = self.getOVERLAP_V() if overlap_V == -1 else overlap_V
        overlap_H = self.getOVERLAP_H() if overlap_H == -1 else overlap_H
        if (overlap_V < S_OVERLAP_MIN or overlap_V > self.getStacksHeight()) and self.N_ROWS > 1:
            raise ValueError(f"in StackStitcher::computeDisplacements(...): overlap_V(={overlap_V}) must be in [{S_OVERLAP_MIN},{self.getStacksHeight()}]")
        if (overlap_H < S_OVERLAP_MIN or overlap_H > self.getStacksWidth()) and self.N_COLS > 1:
            raise ValueError(f"in StackStitcher::computeDisplacements(...): overlap_H(={overlap_H}) must be in [{S_OVERLAP_MIN},{self.getStacksWidth()}]")
        z0 = 0 if z0 == -1 else z0
        z1 = self.N_SLICES - 1 if z1 == -1 else z1
        if z0 < 0 or z0 > z1 or z1 >= self.N_SLICES:
            raise ValueError(f"in StackStitcher::computeDisplacements(): incorrect subdata selection [{z0},{z1}] along Z")
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        n = 0
        for a, b in subvolume_layers(z1 - z0 + 1, subvol_DIM_D):
            za, zb = z0 + a, z0 + b - 1
            complete = [[s.isComplete(za, zb) for s in row] for row in self.STACKS]
            if all(all(r) for r in complete):
                # integer tiles stay integers on the device when the 16-bit MIP kernel takes the geometry (MI_NCC_FLOAT_TILES=1:
                # always convert to float32 first, the reference's in-memory form)
                as_samples = zb - za + 1 <= 32 and self.getStacksWidth() % 2 == 0 and not os.environ.get("MI_NCC_FLOAT_TILES")
                loaded = [[self.loadImageStackSamplesDevice(s, za, zb, dev) for s in row] for row in self.STACKS] if as_samples else None
                if loaded is not None and len({sc for row in loaded for _, sc in row}) == 1:
                    tiles, scale = [[t for t, _ in row] for row in loaded], loaded[0][0][1]
                    res = crossmips.compute_displacements(tiles, overlap_V, overlap_H, displ_max_V, displ_max_H, displ_max_D,
                                                          rank=rank, world_size=world_size, sample_scale=scale)
                else:
                    tiles = [[self.loadImageStackDevice(s, za, zb, dev) for s in row] for row in self.STACKS]
                    res = crossmips.compute_displacements(tiles, overlap_V, overlap_H, displ_max_V, displ_max_H, displ_max_D,
                                                          rank=rank, world_size=world_size)
            else:   # sparse layer: pair by pair over the tiles that exist
                res, q = {}, 0
                for (r, c, rb, cb, direction) in crossmips.enumerate_pairs(self.N_ROWS, self.N_COLS):
                    if not (complete[r][c] and complete[rb][cb]):
                        continue
                    q += 1
                    if (q - 1) % world_size != rank:
                        continue
                    A = self.loadImageStackDevice(self.STACKS[r][c], za, zb, dev)
                    B = self.loadImageStackDevice(self.STACKS[rb][cb], za, zb, dev)
                    res[(r, c, rb, cb, direction)] = crossmips.PDAlgoMIPNCC.execute(
                        A, B, displ_max_V, displ_max_H, displ_max_D, direction,
                        overlap_V if direction == dir_vertical else overlap_H, device=dev)
            for (r, c, rb, cb, _), d in sorted(res.items()):
                self.insertDisplacement(self.STACKS[r][c], self.STACKS[rb][cb], d)
                n += 1
        return n

    def mergeDisplacements(self, other: "Project"):
        """mergedisplacements (Parastitcher.py:474-508): the displacement lists of another partial result are appended."""
        if (other.N_ROWS, other.N_COLS) != (self.N_ROWS, self.N_COLS):
            raise ValueError("mergedisplacements: projects of different grids")
        for row_a, row_b in zip(self.STACKS, other.STACKS):
            for a, b in zip(row_a, row_b):
                for side in _SIDES:
                    getattr(a, side).extend(getattr(b, side))
