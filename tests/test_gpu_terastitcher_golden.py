"""GPU: stitch steps 2-4 against the reference's OWN binary.  tests/golden/terastitcher/ holds the project files that
/root/reference/TeraStitcher/Linux/AVX2/terastitcher wrote for a small synthetic tiled TIFF volume (generator:
tests/golden/make_terastitcher_golden.py, build container only) -- ``process_images.py -2/-3/-4`` on the same TIFFs must
reproduce every displacement record: integers exact, peaks / reliabilities to float tolerance
(StackStitcher.cpp:119-397,1563-1720, DisplacementMIPNCC.cpp:367-434)."""
import os
import xml.etree.ElementTree as ET

import numpy as np
import pytest

GOLD_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# "terastitcher": 2 x 3 grid of 16-bit tiles, 3 thin z layers (D unreliable by construction), one all-zero layer (NaN maps);
# "terastitcher_8bit": 3 x 2 grid of 8-bit tiles, one 36-slice layer with a live D search
DATASETS = {"terastitcher": (42, 14), "terastitcher_8bit": (14, 14)}     # records after step 2 / after steps 3 and 4
INT_FIELDS = ("displ", "default_displ", "nccWidth", "nccWRangeThr", "nccInvWidth", "delay")
FLOAT_FIELDS = ("reliability", "nccPeak")


def write_tiff_tree(root, npz):
    """The TIFF tree the golden run read: <V>/<V>_<H>/<V>_<H>_<D>.tif, positions in 0.1 um."""
    from PIL import Image
    rows, cols = int(npz["rows"]), int(npz["cols"])
    vxl = npz["vxl"]
    ov_v, ov_h = (int(v) for v in npz["overlap"])
    for r in range(rows):
        for c in range(cols):
            vol = npz[f"tile_{r}_{c}"]
            step_v, step_h = vol.shape[1] - ov_v, vol.shape[2] - ov_h
            v = int(round(r * step_v * vxl[0] * 10))
            h = int(round(c * step_h * vxl[1] * 10))
            d = os.path.join(root, f"{v:06d}", f"{v:06d}_{h:06d}")
            os.makedirs(d, exist_ok=True)
            for z in range(vol.shape[0]):
                Image.fromarray(vol[z]).save(os.path.join(d, f"{v:06d}_{h:06d}_{int(round(z * vxl[2] * 10)):06d}.tif"))


def records(path):
    """{(row, col, side): [ {axis: {field: text}} per layer ]} of a TeraStitcher project file."""
    out = {}
    for st in ET.parse(path).getroot().find("STACKS"):
        for side in ("NORTH", "EAST", "SOUTH", "WEST"):
            lst = []
            for d in st.find(f"{side}_displacements"):
                assert d.get("TYPE") == "MIP_NCC"
                lst.append({ax: dict(d.find(ax).attrib) for ax in "VHD"})
            out[(int(st.get("ROW")), int(st.get("COL")), side)] = lst
    return out


def stack_flags(path):
    return {(int(st.get("ROW")), int(st.get("COL"))): st.get("STITCHABLE") for st in ET.parse(path).getroot().find("STACKS")}


def compare(got_path, want_path, what):
    got, want = records(got_path), records(want_path)
    assert got.keys() == want.keys()
    n = 0
    for key in want:
        assert len(got[key]) == len(want[key]), (what, key, len(got[key]), len(want[key]))
        for layer, (g, w) in enumerate(zip(got[key], want[key])):
            for ax in "VHD":
                for f in INT_FIELDS:
                    assert int(g[ax][f]) == int(w[ax][f]), (what, key, layer, ax, f, g[ax], w[ax])
                for f in FLOAT_FIELDS:
                    assert float(g[ax][f]) == pytest.approx(float(w[ax][f]), rel=2e-5, abs=2e-6), (what, key, layer, ax, f)
            n += 1
    return n


@pytest.mark.gpu
@pytest.mark.parametrize("dataset", sorted(DATASETS))
def test_steps_2_3_4_reproduce_the_reference_binary(dev, tmp_path, dataset):
    from ipp_amd import process_images
    GOLD = os.path.join(GOLD_ROOT, dataset)
    n_step2, n_step34 = DATASETS[dataset]
    npz = np.load(os.path.join(GOLD, "tiles.npz"))
    tiles_dir = tmp_path / "tiles"
    write_tiff_tree(str(tiles_dir), npz)
    x1 = tmp_path / "xml_import.xml"
    x1.write_text(open(os.path.join(GOLD, "xml_import.xml")).read().replace("TILES_DIR", str(tiles_dir)))
    x2, x3, x4 = (tmp_path / f"xml_{k}.xml" for k in ("displcomp", "displproj", "displthres"))
    sv, sh, sd = (int(v) for v in npz["search"])
    thr = float(npz["threshold"])
    assert process_images.main(["-2", f"--sV={sv}", f"--sH={sh}", f"--sD={sd}", f"--subvoldim={int(npz['subvoldim'])}",
                                f"--threshold={thr}", f"--projin={x1}", f"--projout={x2}"]) == 0
    n2 = compare(x2, os.path.join(GOLD, "xml_displcomp.xml"), "step 2")
    assert n2 == n_step2                  # adjacent pairs x layers, stored on both tiles of a pair
    assert process_images.main(["-3", f"--projin={x2}", f"--projout={x3}"]) == 0
    assert compare(x3, os.path.join(GOLD, "xml_displproj.xml"), "step 3") == n_step34
    assert process_images.main(["-4", f"--threshold={thr}", f"--projin={x3}", f"--projout={x4}"]) == 0
    assert compare(x4, os.path.join(GOLD, "xml_displthres.xml"), "step 4") == n_step34
    assert stack_flags(x4) == stack_flags(os.path.join(GOLD, "xml_displthres.xml"))
    # steps 3 and 4 are host bookkeeping: from the REFERENCE's step-2 file they must give the reference's files too
    g2 = tmp_path / "gold2.xml"
    g2.write_text(open(os.path.join(GOLD, "xml_displcomp.xml")).read().replace("TILES_DIR", str(tiles_dir)))
    y3, y4 = tmp_path / "y3.xml", tmp_path / "y4.xml"
    assert process_images.main(["-3", f"--projin={g2}", f"--projout={y3}"]) == 0
    assert process_images.main(["-4", f"--threshold={thr}", f"--projin={y3}", f"--projout={y4}"]) == 0
    compare(y3, os.path.join(GOLD, "xml_displproj.xml"), "step 3 from the reference's step 2")
    compare(y4, os.path.join(GOLD, "xml_displthres.xml"), "step 4 from the reference's step 2")


@pytest.mark.parametrize("dataset", sorted(DATASETS))
def test_steps_3_4_from_the_reference_step2_file(tmp_path, dataset):
    """CPU: projection (StackStitcher.cpp:1563-1624) and thresholding (:1626-1720) are host bookkeeping on the XML --
    fed with the reference binary's own step-2 file they must reproduce its step-3 and step-4 files."""
    from ipp_amd import tsproject
    GOLD = os.path.join(GOLD_ROOT, dataset)
    thr = float(np.load(os.path.join(GOLD, "tiles.npz"))["threshold"])
    proj = tsproject.Project.load(os.path.join(GOLD, "xml_displcomp.xml"))
    y3, y4 = tmp_path / "y3.xml", tmp_path / "y4.xml"
    proj.projectDisplacements()
    proj.save(y3)
    compare(y3, os.path.join(GOLD, "xml_displproj.xml"), "step 3")
    proj.thresholdDisplacements(thr)
    proj.save(y4)
    compare(y4, os.path.join(GOLD, "xml_displthres.xml"), "step 4")
    assert stack_flags(y4) == stack_flags(os.path.join(GOLD, "xml_displthres.xml"))
