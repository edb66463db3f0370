#!/usr/bin/env python3
"""Stitching steps 2-4 ("pairwise displacement computation", projection, thresholding) entry point.

The reference's process_images.py builds ``mpiexec ... Parastitcher.py -2 --sD=.. --subvoldim=.. --threshold=0.65
--projin=.. --projout=..`` and toggles the GPU kernels with ``USECUDA_X_NCC`` (process_images.py:542-571,1552); the
work behind that command is StackStitcher::computeDisplacements -> PDAlgoMIPNCC::execute -> norm_cross_corr_mips.
This entry point keeps the step-2 flags and runs that work in-process on the MI355X library:

    python process_images.py -2 --input TILES_DIR --oV 307 --oH 307 [--sV 25 --sH 25 --sD 10 --subvoldim 200]
                                [--projout xml_displcomp.xml]

``TILES_DIR`` holds ``tile_<row>_<col>.npy`` stacks (D, V, H), uint8/uint16 (scaled to [0,1] like loadImageStack,
tiff2D.cpp:606-610) or float32.  One ``<Displacement TYPE="MIP_NCC">`` record per pair and z-layer is written in
the shape of DisplacementMIPNCC::getXML (DisplacementMIPNCC.cpp:367-400).

    python process_images.py -3 --input TILES_DIR [--projin xml_displcomp.xml --projout xml_displproj.xml]
    python process_images.py -4 --input TILES_DIR [--threshold 0.65 --projin xml_displproj.xml --projout xml_displthres.xml]

With a TeraStitcher project file the three steps are drop-ins for ``terastitcher -2 / -3 / -4`` between the reference's
import (``-1``) and placement (``-5``) steps (process_images.py:461-576 builds exactly these command lines):

    python process_images.py -2 --projin xml_import_step_1.xml --projout xml_import_step_2.xml [--sD 10 --subvoldim 100]
    python process_images.py -3 --projin xml_import_step_2.xml --projout xml_import_step_3.xml
    python process_images.py -4 --projin xml_import_step_3.xml --projout xml_import_step_4.xml --threshold 0.65

The tiles are then the 2-D TIFF series the project names (``stacks_dir/DIR_NAME``), the overlaps default to the stage
geometry of the project (``ipp_amd.tsproject``), and the output is the same project with the displacement lists filled in.

Step 3 combines the per-layer records of every pair into the most reliable one per direction
(StackStitcher::projectDisplacements, Displacement::projectDisplacements, DisplacementMIPNCC::combine); step 4 resets the
directions whose reliability is below the threshold to the stage displacement and flags the stitchable stacks
(StackStitcher::thresholdDisplacements).  Both are host-side bookkeeping on the XML.  Steps 1 and 5
(import, global placement + merge) and the interactive pipeline around them stay with the reference's tools.
With ``torchrun`` the pairs of a layer are dealt round-robin to the ranks (no collective; the per-rank XML fragments
are merged by rank 0, like Parastitcher's mergedisplacements).
"""
from __future__ import annotations

import argparse
import os
import re
import sys
import xml.etree.ElementTree as ET
from pathlib import Path

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)


def build_parser():
    p = argparse.ArgumentParser(description="MIP-NCC pairwise displacement computation (stitching step 2)")
    p.add_argument("-2", "--displcompute", dest="step2", action="store_true", help="step 2: pairwise displacements (GPU)")
    p.add_argument("-3", "--displproj", dest="step3", action="store_true", help="step 3: projection of the per-layer records")
    p.add_argument("-4", "--displthres", dest="step4", action="store_true", help="step 4: reliability thresholding")
    p.add_argument("--input", type=Path, default=None, help="folder with tile_<row>_<col>.npy stacks (or --projin PROJECT.xml)")
    p.add_argument("--oV", type=int, default=None, help="overlap (pixels) between adjacent tiles along V (step 2)")
    p.add_argument("--oH", type=int, default=None, help="overlap (pixels) between adjacent tiles along H (step 2)")
    p.add_argument("--projin", type=Path, default=None, help="input XML of steps 3 / 4")
    p.add_argument("--sV", type=int, default=25, help="displacement search radius along V (S_config.h:59)")
    p.add_argument("--sH", type=int, default=25, help="displacement search radius along H")
    p.add_argument("--sD", type=int, default=10, help="displacement search radius along D (process_images.py:562)")
    p.add_argument("--subvoldim", type=int, default=200, help="slices per z-layer (S_config.h:60)")
    p.add_argument("--threshold", type=float, default=0.65, help="reliability threshold recorded for step 4")
    p.add_argument("--projout", type=Path, default=None, help="output XML (default <input>/xml_displcomp.xml)")
    return p


def load_tiles(folder: Path):
    import numpy as np
    grid = {}
    for f in folder.glob("tile_*_*.npy"):
        m = re.fullmatch(r"tile_(\d+)_(\d+)\.npy", f.name)
        if m:
            grid[(int(m.group(1)), int(m.group(2)))] = f
    if not grid:
        raise RuntimeError(f"no tile_<row>_<col>.npy stacks in {folder}")
    rows, cols = 1 + max(r for r, _ in grid), 1 + max(c for _, c in grid)
    if len(grid) != rows * cols:
        raise RuntimeError("sparse tile grids are not supported")
    tiles = []
    for r in range(rows):
        row = []
        for c in range(cols):
            a = np.load(grid[(r, c)])
            if a.dtype in (np.uint8, np.uint16):
                row.append(np.ascontiguousarray(a))          # (kept as samples: see sample_scale / to_device below)
                continue
            if np.issubdtype(a.dtype, np.integer):
                a = a.astype(np.float32) / np.float32(np.iinfo(a.dtype).max)
            row.append(np.ascontiguousarray(a, dtype=np.float32))
        tiles.append(row)
    return tiles


def layer_to_device(host_tiles, z0, z1, dev):
    """One z layer of the grid on the device (loadImageStack(z0, z1)): float32 in [0, 1] like the reference's stacks (sample / 255 or
    / 65535, tiff2D.cpp:606-610) -- or, when all tiles are 8 / 16-bit samples and the 16-bit MIP kernel takes the geometry, the
    samples themselves as uint16 and the divisor (identical records, half the bytes).  Returns (tiles, sample_scale or None)."""
    import numpy as np
    import torch
    kinds = {t.dtype for row in host_tiles for t in row}
    width = host_tiles[0][0].shape[2]
    if len(kinds) == 1 and next(iter(kinds)) in (np.dtype(np.uint8), np.dtype(np.uint16)) and z1 - z0 <= 32 and width % 2 == 0 \
            and not os.environ.get("MI_NCC_FLOAT_TILES"):
        kind = next(iter(kinds))
        scale = 255.0 if kind == np.dtype(np.uint8) else 65535.0
        if kind == np.dtype(np.uint8) and width % 4:
            kind = np.dtype(np.uint16)                  # (rows that are no whole 32-bit words: widened, the 16-bit kernel takes them)
        return [[torch.from_numpy(np.ascontiguousarray(t[z0:z1], dtype=kind)).to(dev) for t in row] for row in host_tiles], scale

    def as_float(t):
        if t.dtype in (np.uint8, np.uint16):
            return t[z0:z1].astype(np.float32) / np.float32(np.iinfo(t.dtype).max)
        return t[z0:z1]
    return [[torch.from_numpy(np.ascontiguousarray(as_float(t))).to(dev) for t in row] for row in host_tiles], None


def read_pairs(path):
    """(root attributes, [(pair attributes, DisplacementMIPNCC)]) of an XML written by one of the steps."""
    from ipp_amd import crossmips
    root = ET.parse(path).getroot()
    recs = [(dict(p.attrib), crossmips.DisplacementMIPNCC.loadXML(p.find("Displacement"))) for p in root.findall("Pair")]
    return dict(root.attrib), recs


def write_pairs(path, root_attrib, recs, stacks=None):
    root = ET.Element("TeraStitcher", **{k: str(v) for k, v in root_attrib.items()})
    for attrib, d in recs:
        ET.SubElement(root, "Pair", **{k: str(v) for k, v in attrib.items()}).append(d.getXML())
    for (r, c), flag in sorted((stacks or {}).items()):
        ET.SubElement(root, "Stack", row=str(r), col=str(c), stitchable="yes" if flag else "no")
    ET.indent(root)
    ET.ElementTree(root).write(path, xml_declaration=True, encoding="utf-8")


def _pair_key(attrib):
    return tuple(int(attrib[k]) for k in ("rowA", "colA", "rowB", "colB"))


def step3_project(args):
    """projectDisplacements: one record per pair, per direction the most reliable of its layers."""
    from ipp_amd import crossmips
    src = args.projin or (args.input / "xml_displcomp.xml")
    root_attrib, recs = read_pairs(src)
    groups = {}
    for attrib, d in sorted(recs, key=lambda r: int(r[0].get("layer", 0))):
        groups.setdefault(_pair_key(attrib), []).append((attrib, d))
    out = []
    for key in sorted(groups):
        attrib = {k: v for k, v in groups[key][0][0].items() if k not in ("layer", "z0", "z1")}
        out.append((attrib, crossmips.project_displacements([d for _, d in groups[key]])))
    root_attrib["step"] = "3"
    dst = args.projout or (args.input / "xml_displproj.xml")
    write_pairs(dst, root_attrib, out)
    print(f"wrote {dst} ({len(out)} projected displacement records from {len(recs)})")
    return 0


def step4_threshold(args):
    """thresholdDisplacements: unreliable directions fall back to the stage displacement; stitchable stacks are flagged."""
    from ipp_amd import crossmips
    src = args.projin or (args.input / "xml_displproj.xml")
    root_attrib, recs = read_pairs(src)
    grid = {}
    for attrib, d in recs:
        if _pair_key(attrib) in grid:
            raise RuntimeError("in StackStitcher::thresholdDisplacements(...): one and only displacement must exist for each pair "
                               "of adjacent stacks.")
        grid[_pair_key(attrib)] = d
    n_rows = 1 + max(k[2] for k in grid)
    n_cols = 1 + max(k[3] for k in grid)
    stacks = crossmips.threshold_displacements(grid, n_rows, n_cols, args.threshold)
    root_attrib.update(step="4", threshold=str(args.threshold))
    dst = args.projout or (args.input / "xml_displthres.xml")
    write_pairs(dst, root_attrib, recs, stacks)
    print(f"wrote {dst} ({sum(stacks.values())} of {len(stacks)} stacks stitchable at threshold {args.threshold})")
    return 0


def _is_project(path):
    """True for a TeraStitcher project file (<TeraStitcher> with <STACKS>), false for the flat pair list of the .npy mode."""
    if path is None or not Path(path).exists():
        return False
    root = ET.parse(path).getroot()
    return root.tag == "TeraStitcher" and root.find("STACKS") is not None


def project_steps(args):
    """terastitcher -2 / -3 / -4 on a project file: --projin -> --projout."""
    from ipp_amd import tsproject
    if args.projout is None:
        raise SystemExit("--projout is required with a TeraStitcher project")
    proj = tsproject.Project.load(args.projin)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if args.step2:
        import torch
        from ipp_amd import capi
        capi.require_gpu()
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(dev)
        n = proj.computeDisplacements(-1 if args.oV is None else args.oV, -1 if args.oH is None else args.oH, args.sV, args.sH,
                                      args.sD, args.subvoldim, device=dev, rank=rank, world_size=world)
        if world > 1:   # per-rank partial projects, merged by rank 0 (mergedisplacements)
            import torch.distributed as dist
            proj.save(f"{args.projout}.rank{rank}")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
            if rank == 0:
                for k in range(1, world):
                    part = Path(f"{args.projout}.rank{k}")
                    # adjustDisplacements on load rebuilt the mirrors; only the owning side's lists are merged
                    other = tsproject.Project.load(part)
                    for row_a, row_b in zip(proj.STACKS, other.STACKS):
                        for a, b in zip(row_a, row_b):
                            a.EAST.extend(b.EAST)
                            a.SOUTH.extend(b.SOUTH)
                    part.unlink()
                Path(f"{args.projout}.rank0").unlink()
                proj.adjustDisplacements()
                proj.save(args.projout)
            dist.barrier()
            dist.destroy_process_group()
        else:
            proj.save(args.projout)
        if rank == 0:
            print(f"wrote {args.projout} ({n} displacement records computed by this rank)")
        return 0
    if args.step3:
        proj.projectDisplacements()
    else:
        proj.thresholdDisplacements(args.threshold)
    proj.save(args.projout)
    print(f"wrote {args.projout}")
    return 0


def main(argv=None):
    args = build_parser().parse_args(argv)
    if (args.step2 or args.step3 or args.step4) and _is_project(args.projin):
        return project_steps(args)
    if args.input is None:
        raise SystemExit("--input TILES_DIR (tile_<row>_<col>.npy stacks) or --projin PROJECT.xml is required")
    if args.step3:
        return step3_project(args)
    if args.step4:
        return step4_threshold(args)
    if not args.step2:
        raise SystemExit("steps 2 (-2, pairwise displacement computation), 3 (-3) and 4 (-4) are built; see SURVEY.md section 8f")
    if args.oV is None or args.oH is None:
        raise SystemExit("step 2 needs --oV and --oH")
    import torch
    from ipp_amd import capi, crossmips
    capi.require_gpu()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    host_tiles = load_tiles(args.input)
    dim_D, dim_V, dim_H = host_tiles[0][0].shape
    root = ET.Element("TeraStitcher", step="2", threshold=str(args.threshold), sV=str(args.sV), sH=str(args.sH), sD=str(args.sD),
                      subvoldim=str(args.subvoldim))
    for layer, (z0, z1) in enumerate(crossmips.subvolume_layers(dim_D, args.subvoldim)):
        tiles, scale = layer_to_device(host_tiles, z0, z1, dev)                              # loadImageStack(z0, z1)
        res = crossmips.compute_displacements(tiles, args.oV, args.oH, args.sV, args.sH, args.sD, rank=rank, world_size=world,
                                              **({} if scale is None else {"sample_scale": scale}))
        for (r, c, rb, cb, direction), d in sorted(res.items()):
            pair = ET.SubElement(root, "Pair", layer=str(layer), z0=str(z0), z1=str(z1), rowA=str(r), colA=str(c), rowB=str(rb),
                                 colB=str(cb), direction="NORTH_SOUTH" if direction == crossmips.dir_vertical else "WEST_EAST")
            pair.append(d.getXML())
    out = args.projout or (args.input / "xml_displcomp.xml")
    if world > 1:
        part = Path(f"{out}.rank{rank}")
        ET.ElementTree(root).write(part)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
        if rank == 0:  # mergedisplacements
            for k in range(1, world):
                other = Path(f"{out}.rank{k}")
                root.extend(ET.parse(other).getroot())
                other.unlink()
            part.unlink()
            ET.indent(root)
            ET.ElementTree(root).write(out, xml_declaration=True, encoding="utf-8")
        dist.barrier()
        dist.destroy_process_group()
    else:
        ET.indent(root)
        ET.ElementTree(root).write(out, xml_declaration=True, encoding="utf-8")
    if rank == 0:
        print(f"wrote {out} ({len(root)} displacement records)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
