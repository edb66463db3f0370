#!/usr/bin/env python3
"""Golden fixtures from the reference's OWN binary (build container only; /root/reference never travels).

Writes a small synthetic tiled volume in TeraStitcher's two-level layout (16-bit TIFF series), runs the reference's prebuilt
``/root/reference/TeraStitcher/Linux/AVX2/terastitcher`` on it exactly as the reference's process_images.py does
(process_images.py:461-476 import, :544-571 steps 2-4: ``-1 --sparse_data`` -> ``-2`` -> ``-3`` -> ``-4 --threshold=0.65``)
and stores, under tests/golden/terastitcher/:

    tiles.npz                uint16 tile stacks (the TIFF pixel data; the test re-creates the TIFF tree from them)
    xml_import.xml           project after step 1 (the input of our step 2)
    xml_displcomp.xml        after step 2 (every <Displacement TYPE="MIP_NCC">)
    xml_displproj.xml        after step 3
    xml_displthres.xml       after step 4

tests/test_gpu_terastitcher_golden.py then requires ``process_images.py -2/-3/-4`` on the same TIFFs to reproduce every
displ / nccWidth / nccWRangeThr / reliability of those files.  Only data travels: pixel arrays and the XML the binary wrote.
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ncc_oracle as N  # noqa: E402  (bead_field: the synthetic data generator of the NCC goldens)

TS = "/root/reference/TeraStitcher/Linux/AVX2/terastitcher"

# two datasets: "terastitcher" = 16-bit tiles, 3 thin z layers (D unreliable by construction, one empty layer);
# "terastitcher_8bit" = 8-bit tiles, one layer of 36 slices so that the D search is live (delay_k = min(sD, 36 - 25))
DATASETS = {
    "terastitcher": dict(ROWS=2, COLS=3, TILE=(96, 112), OV=(32, 40), SLICES=44, VXL=(0.8, 0.8, 2.0), SEARCH=(6, 7, 2), SUBVOL=20,
                         THRESHOLD=0.65, BITS=16, SEED=2027, JIT=(3, 1), EMPTY=((1, 2), 30)),
    "terastitcher_8bit": dict(ROWS=3, COLS=2, TILE=(80, 88), OV=(30, 34), SLICES=36, VXL=(1.25, 1.25, 5.0), SEARCH=(5, 5, 3), SUBVOL=100,
                              THRESHOLD=0.65, BITS=8, SEED=31, JIT=(2, 2), EMPTY=None),
}


def make_tiles(cfg):
    ROWS, COLS, TILE, SLICES, seed = cfg["ROWS"], cfg["COLS"], cfg["TILE"], cfg["SLICES"], cfg["SEED"]
    OV_V, OV_H = cfg["OV"]
    jv, jd = cfg["JIT"]
    full = 60000.0 if cfg["BITS"] == 16 else 250.0
    step_v, step_h = TILE[0] - OV_V, TILE[1] - OV_H
    field = N.bead_field((SLICES + 8, (ROWS - 1) * step_v + TILE[0] + 16, (COLS - 1) * step_h + TILE[1] + 16), seed=seed,
                         density=1 / 260)
    rng = np.random.default_rng(seed)
    tiles = {}
    for r in range(ROWS):
        for c in range(COLS):
            dv, dh = (0, 0) if (r, c) == (0, 0) else (int(rng.integers(-jv, jv + 1)), int(rng.integers(-jv, jv + 1)))
            dd = 0 if (r, c) == (0, 0) else int(rng.integers(-jd, jd + 1))
            v0, h0 = 8 + r * step_v + dv, 8 + c * step_h + dh
            t = field[4 + dd:4 + dd + SLICES, v0:v0 + TILE[0], h0:h0 + TILE[1]]
            noise = rng.normal(0.0, 0.004, size=t.shape)
            q = np.clip(np.rint((t + noise) * full), 0, 65535 if cfg["BITS"] == 16 else 255)
            tiles[(r, c)] = q.astype(np.uint16 if cfg["BITS"] == 16 else np.uint8)
    if cfg["EMPTY"]:
        # one tile with an empty (all-zero) layer: the pairs it takes part in give NaN maps -> unreliable records in that layer
        (r, c), z0 = cfg["EMPTY"]
        tiles[(r, c)][z0:] = 0
    return tiles


def write_tree(root, tiles, cfg):
    """<root>/<V in 0.1 um, 6 digits>/<V>_<H>/<V>_<H>_<D>.tif (the layout the reference's converters write)."""
    TILE, VXL = cfg["TILE"], cfg["VXL"]
    step_v, step_h = TILE[0] - cfg["OV"][0], TILE[1] - cfg["OV"][1]
    for (r, c), vol in tiles.items():
        v = int(round(r * step_v * VXL[0] * 10))
        h = int(round(c * step_h * VXL[1] * 10))
        d = os.path.join(root, f"{v:06d}", f"{v:06d}_{h:06d}")
        os.makedirs(d, exist_ok=True)
        for z in range(vol.shape[0]):
            Image.fromarray(vol[z]).save(os.path.join(d, f"{v:06d}_{h:06d}_{int(round(z * VXL[2] * 10)):06d}.tif"))


def run(cmd):
    print(" ".join(cmd), flush=True)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0:
        print(p.stdout[-3000:])
        raise SystemExit(f"terastitcher failed with {p.returncode}")
    return p.stdout


def generate(name, cfg):
    OUT = os.path.join(ROOT, "tests", "golden", name)
    VXL, SEARCH, SUBVOL, THRESHOLD = cfg["VXL"], cfg["SEARCH"], cfg["SUBVOL"], cfg["THRESHOLD"]
    os.makedirs(OUT, exist_ok=True)
    tiles = make_tiles(cfg)
    work = tempfile.mkdtemp(prefix="ts_golden_")
    try:
        vol = os.path.join(work, "tiles")
        write_tree(vol, tiles, cfg)
        x = {k: os.path.join(work, f"xml_{k}.xml") for k in ("import", "displcomp", "displproj", "displthres")}
        run([TS, "-1", "--ref1=V", "--ref2=H", "--ref3=D", f"--vxl1={VXL[0]}", f"--vxl2={VXL[1]}", f"--vxl3={VXL[2]}", "--sparse_data",
             f"--volin={vol}", f"--projout={x['import']}", "--noprogressbar"])
        run([TS, "-2", f"--sV={SEARCH[0]}", f"--sH={SEARCH[1]}", f"--sD={SEARCH[2]}", f"--subvoldim={SUBVOL}", f"--threshold={THRESHOLD}",
             f"--projin={x['import']}", f"--projout={x['displcomp']}", "--noprogressbar"])
        run([TS, "-3", f"--projin={x['displcomp']}", f"--projout={x['displproj']}", "--noprogressbar"])
        run([TS, "-4", f"--threshold={THRESHOLD}", f"--projin={x['displproj']}", f"--projout={x['displthres']}", "--noprogressbar"])
        for k, path in x.items():
            text = open(path).read().replace(vol, "TILES_DIR")       # the absolute stacks_dir of this run -> a placeholder
            with open(os.path.join(OUT, f"xml_{k}.xml"), "w") as f:
                f.write(text)
        np.savez_compressed(os.path.join(OUT, "tiles.npz"), rows=cfg["ROWS"], cols=cfg["COLS"], vxl=np.array(VXL), search=np.array(SEARCH),
                            subvoldim=SUBVOL, threshold=THRESHOLD, overlap=np.array(cfg["OV"]),
                            **{f"tile_{r}_{c}": v for (r, c), v in tiles.items()})
    finally:
        shutil.rmtree(work, ignore_errors=True)
    print("wrote", name, sorted(os.listdir(OUT)))


if __name__ == "__main__":
    for name, cfg in DATASETS.items():
        generate(name, cfg)
