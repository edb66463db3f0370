"""TeraStitcher project files for stitching steps 2-4 (ipp_amd.tsproject): format, geometry and the
displacement bookkeeping of StackStitcher / VirtualVolume, on the CPU."""
import re
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from ipp_amd import crossmips, tsproject

PIL = pytest.importorskip("PIL.Image")


def _displ(coords, peaks, widths):
    d = crossmips.DisplacementMIPNCC(list(coords), list(peaks), list(widths), [25, 25, 10], [29, 29, 29], [30, 30, 30])
    for k in range(3):
        d.evalReliability(k)
    return d


def make_project(tmp_path, rows=2, cols=2, slices=6, tile=(24, 32), step=(18, 26), dtype=np.uint16, sparse=None):
    rng = np.random.default_rng(0)
    p = tsproject.Project(tmp_path / "tiles", rows, cols, slices, VXL=(0.5, 0.5, 2.0), ORG=(1.25, 2.5, 0.0),
                          MEC=(step[0] * 0.5, step[1] * 0.5))
    for r in range(rows):
        for c in range(cols):
            name = f"{r * step[0] * 5:06d}/{r * step[0] * 5:06d}_{c * step[1] * 5:06d}"
            folder = tmp_path / "tiles" / name
            folder.mkdir(parents=True)
            zr = (sparse or {}).get((r, c), [(0, slices)])
            for k, z in enumerate(z for a, b in zr for z in range(a, b)):
                a = rng.integers(0, np.iinfo(dtype).max, tile, dtype=dtype)
                PIL.fromarray(a).save(folder / f"{z * 20:06d}.tif")
            p.STACKS[r][c] = tsproject.Stack(r, c, name, ABS_V=r * step[0], ABS_H=c * step[1], N_BYTESxCHAN=np.dtype(dtype).itemsize,
                                             z_ranges=list(zr))
    return p


def test_geometry_and_round_trip(tmp_path):
    p = make_project(tmp_path)
    assert (p.getStacksHeight(), p.getStacksWidth()) == (24, 32)
    assert (p.getOVERLAP_V(), p.getOVERLAP_H()) == (6, 6) and (p.getDEFAULT_DISPLACEMENT_V(), p.getDEFAULT_DISPLACEMENT_H()) == (18, 26)
    a, b = _displ((0, 27, 1), (0.93, 0.88, 0.5), (2, 3, 9)), _displ((17, -1, 0), (0.7, 0.6, 0.4), (4, 5, 12))
    p.insertDisplacement(p.STACKS[0][0], p.STACKS[0][1], a)
    p.insertDisplacement(p.STACKS[0][0], p.STACKS[1][0], b)
    assert a.VHD_def_coords == [0, 26, 0] and b.VHD_def_coords == [18, 0, 0]           # vmVirtualVolume.cpp:291-302
    w = p.STACKS[0][1].WEST[0]
    assert w.VHD_coords == [0, -27, -1] and w.VHD_def_coords == [0, -26, 0] and p.STACKS[1][0].NORTH[0].VHD_coords == [-17, 1, 0]
    with pytest.raises(ValueError, match="not adjacent"):
        p.insertDisplacement(p.STACKS[0][0], p.STACKS[1][1], _displ((0, 0, 0), (0, 0, 0), (30, 30, 30)))
    out = tmp_path / "xml_displcomp.xml"
    p.save(out)
    text = out.read_text()
    assert text.startswith('<?xml version="1.0" encoding="UTF-8" ?>\n<!DOCTYPE TeraStitcher SYSTEM "TeraStitcher.DTD">\n')
    root = ET.parse(out).getroot()
    assert root.get("volume_format") == "TiledXY|2Dseries" and root.find("voxel_dims").attrib == {"V": "0.5", "H": "0.5", "D": "2"}
    assert root.find("dimensions").attrib == {"stack_rows": "2", "stack_columns": "2", "stack_slices": "6"}
    st = root.find("STACKS").findall("Stack")
    assert [(s.get("ROW"), s.get("COL")) for s in st] == [("0", "0"), ("0", "1"), ("1", "0"), ("1", "1")]
    assert st[0].get("Z_RANGES") == "[0,6)" and st[0].get("STITCHABLE") == "no" and st[1].get("ABS_H") == "26"
    assert [c.tag for c in st[0]] == [f"{s}_displacements" for s in ("NORTH", "EAST", "SOUTH", "WEST")]
    v = st[0].find("EAST_displacements/Displacement/H")
    assert v.get("displ") == "27" and v.get("default_displ") == "26" and v.get("nccWidth") == "3" and v.get("delay") == "25"
    assert re.fullmatch(r"0\.\d{1,6}", v.get("reliability")) and v.get("nccPeak") == "0.88"  # %g: six significant digits
    q = tsproject.Project.load(out)
    assert (q.N_ROWS, q.N_COLS, q.N_SLICES, q.stacks_dir) == (2, 2, 6, str(tmp_path / "tiles"))
    assert (q.VXL_V, q.VXL_D, q.ORG_V, q.MEC_H) == (0.5, 2.0, 1.25, 13.0) and q.ref_sys == (1, 2, 3)
    e = q.STACKS[0][0].EAST[0]
    assert e.VHD_coords == a.VHD_coords and e.NCC_widths == a.NCC_widths and e.delays == [25, 25, 10]
    assert e.NCC_maxs == pytest.approx(a.NCC_maxs, rel=1e-5) and e.rel_factors == pytest.approx(a.rel_factors, rel=1e-5)
    assert q.STACKS[0][1].WEST[0].VHD_coords == [0, -27, -1]                             # adjustDisplacements on load
    q.save(tmp_path / "again.xml")
    assert (tmp_path / "again.xml").read_text() == text
    with pytest.raises(ValueError, match="unsupported volume_format"):
        bad = tmp_path / "bad.xml"
        bad.write_text(text.replace("TiledXY|2Dseries", "TiledXY|3Dseries"))
        tsproject.Project.load(bad)


def test_load_image_stack_and_sparse_tiles(tmp_path):
    p = make_project(tmp_path, rows=1, cols=2, slices=6, sparse={(0, 1): [(0, 2), (4, 6)]})
    s0, s1 = p.STACKS[0]
    full = p.loadImageStack(s0, 1, 4)
    assert full.shape == (4, 24, 32) and full.dtype == np.float32 and 0 <= full.min() and full.max() <= 1
    raw = np.asarray(PIL.open(p.slice_files(s0)[1]))
    assert np.array_equal(full[0], raw.astype(np.float32) / np.float32(65535))          # tiff2D.cpp:606-610
    assert s1.isComplete(0, 1) and s1.isComplete(4, 5) and not s1.isComplete(1, 4) and not s1.isComplete(2, 3)
    assert len(p.slice_files(s1)) == 4
    tail = p.loadImageStack(s1, 4, 5)                                                   # third and fourth file of the folder
    assert np.array_equal(tail[0], np.asarray(PIL.open(p.slice_files(s1)[2])).astype(np.float32) / np.float32(65535))
    with pytest.raises(ValueError, match="not all present"):
        p.loadImageStack(s1, 1, 4)
    p.save(tmp_path / "p.xml")
    q = tsproject.Project.load(tmp_path / "p.xml")
    assert q.STACKS[0][1].z_ranges == [(0, 2), (4, 6)]
    bad = (tmp_path / "p.xml").read_text().replace("[0,2);[4,6)", "[0,4);[4,6)")
    (tmp_path / "bad.xml").write_text(bad)
    with pytest.raises(ValueError, match="wrong sequence"):
        tsproject.Project.load(tmp_path / "bad.xml")
    p8 = make_project(tmp_path / "u8", rows=1, cols=1, dtype=np.uint8)
    assert p8.loadImageStack(p8.STACKS[0][0], 0, 0).max() <= 1.0


def test_project_and_threshold_follow_the_stitcher(tmp_path):
    p = make_project(tmp_path)
    S = p.STACKS
    # two layers for the (0,0)-(0,1) pair: layer 0 is better in V, layer 1 in H and D
    p.insertDisplacement(S[0][0], S[0][1], _displ((1, 27, 0), (0.95, 0.30, 0.2), (1, 20, 26)))
    p.insertDisplacement(S[0][0], S[0][1], _displ((4, 25, 2), (0.40, 0.90, 0.9), (15, 2, 3)))
    p.insertDisplacement(S[0][0], S[1][0], _displ((17, 0, 0), (0.1, 0.1, 0.1), (25, 25, 25)))
    p.insertDisplacement(S[0][1], S[1][1], _displ((19, 1, 0), (0.8, 0.2, 0.2), (3, 20, 20)))
    # (1,0)-(1,1) has no record: projection inserts the nominal stage displacement (StackStitcher.cpp:1579-1613)
    with pytest.raises(ValueError, match="one and only displacement"):
        p.thresholdDisplacements(0.65)
    p.projectDisplacements()
    assert all(len(getattr(S[i][j], side)) == (1 if ok else 0) for i in range(2) for j in range(2)
               for side, ok in (("NORTH", i == 1), ("SOUTH", i == 0), ("WEST", j == 1), ("EAST", j == 0)))
    assert S[0][0].EAST[0].VHD_coords == [1, 25, 2] and S[0][1].WEST[0].VHD_coords == [-1, -25, -2]
    nominal = S[1][0].EAST[0]
    assert nominal.VHD_coords == [0, 26, 0] and nominal.NCC_widths == [30] * 3 and nominal.rel_factors == [0.0] * 3
    assert S[1][1].WEST[0].VHD_coords == [0, -26, 0]
    p.save(tmp_path / "proj.xml")
    q = tsproject.Project.load(tmp_path / "proj.xml")
    q.thresholdDisplacements(0.65)
    T = q.STACKS
    assert T[0][0].SOUTH[0].VHD_coords == [18, 0, 0] and T[1][0].NORTH[0].VHD_coords == [-18, 0, 0]   # fell back to the stage offsets
    assert T[0][1].SOUTH[0].VHD_coords == [19, 0, 0] and T[0][1].SOUTH[0].NCC_maxs[1:] == [0.0, 0.0]
    assert [[s.stitchable for s in row] for row in T] == [[True, True], [False, True]]
    # the same decisions as the flat-grid implementation used by the .npy mode
    grid = {(0, 0, 0, 1): p.STACKS[0][0].EAST[0], (0, 0, 1, 0): p.STACKS[0][0].SOUTH[0], (0, 1, 1, 1): p.STACKS[0][1].SOUTH[0],
            (1, 0, 1, 1): p.STACKS[1][0].EAST[0]}
    flags = crossmips.threshold_displacements(grid, 2, 2, 0.65)
    assert flags == {(i, j): T[i][j].stitchable for i in range(2) for j in range(2)}
    q.save(tmp_path / "thres.xml")
    st = ET.parse(tmp_path / "thres.xml").getroot().find("STACKS").findall("Stack")
    assert [s.get("STITCHABLE") for s in st] == ["yes", "yes", "no", "yes"]


def test_merge_of_partial_projects(tmp_path):
    a, b = make_project(tmp_path / "a"), make_project(tmp_path / "b")
    a.insertDisplacement(a.STACKS[0][0], a.STACKS[0][1], _displ((0, 26, 0), (0.9, 0.9, 0.9), (2, 2, 2)))
    b.insertDisplacement(b.STACKS[0][0], b.STACKS[0][1], _displ((1, 27, 0), (0.8, 0.8, 0.8), (3, 3, 3)))
    b.insertDisplacement(b.STACKS[1][0], b.STACKS[1][1], _displ((0, 25, 1), (0.7, 0.7, 0.7), (4, 4, 4)))
    a.mergeDisplacements(b)
    assert [d.VHD_coords for d in a.STACKS[0][0].EAST] == [[0, 26, 0], [1, 27, 0]] and len(a.STACKS[0][1].WEST) == 2
    assert a.STACKS[1][1].WEST[0].VHD_coords == [0, -25, -1]


def test_default_displacement_is_a_single_precision_quotient():
    # vmVirtualVolume.cpp:75-79 on float members: 57.6f / 0.8f = 72 (the double quotient truncates to 71); the reference
    # binary wrote default_displ="72" for this geometry (tests/golden/terastitcher/xml_displcomp.xml)
    import os
    from ipp_amd import tsproject
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "terastitcher", "xml_import.xml")
    p = tsproject.Project.load(gold)
    assert (p.MEC_V, p.MEC_H) == pytest.approx((51.2, 57.6), rel=1e-6)
    assert p.getDEFAULT_DISPLACEMENT_V() == 64 and p.getDEFAULT_DISPLACEMENT_H() == 72
