"""Stitching step 2 (pairwise displacements) on a TeraStitcher-style project of TIFF tiles: a rows x cols grid of tiles of
2048 x 2048 x 32 16-bit slices with BASELINE config 5's overlap (bench_ncc.make_grid's synthetic scene, written out as one folder of
2-D TIFFs per tile), tsproject.Project.computeDisplacements on one GPU -- where the time goes between reading the slices and the NCC.
    python profiles/stitch_tiff_probe.py [rows cols]          (MI_TIFF_PILLOW=1: the slices through Pillow)"""
import os
import shutil
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench_ncc  # noqa: E402
from ipp_amd import brickio, tsproject  # noqa: E402

rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) >= 3 else (4, 4)
dev = torch.device("cuda", 0)
root = "/tmp/stitch_tiff"
shutil.rmtree(root, ignore_errors=True)
tiles, jit, step = bench_ncc.make_grid(dev)                      # 8 x 8 float tiles in [0, 1] on the device
D, V, H = (int(s) for s in tiles[0][0].shape)
ov = bench_ncc.OVERLAP
p = tsproject.Project(os.path.join(root, "tiles"), rows, cols, D, VXL=(1.0, 1.0, 1.0), ORG=(0.0, 0.0, 0.0), MEC=(float(V - ov), float(H - ov)))
t0 = time.perf_counter()
nbytes = 0
for r in range(rows):
    for c in range(cols):
        name = f"{r * (V - ov) * 10:06d}/{r * (V - ov) * 10:06d}_{c * (H - ov) * 10:06d}"
        folder = os.path.join(root, "tiles", name)
        a = (tiles[r][c] * 65535.0).round().clamp(0, 65535).to(torch.uint16).cpu().numpy()
        os.makedirs(folder)
        # (slice files named by their z like the reference's acquisitions; brickio's writer names them img_<n>)
        brickio.save_tiff_series(folder, a, first_index=0)
        for k in range(D):
            os.replace(os.path.join(folder, f"img_{k:06d}.tif"), os.path.join(folder, f"{k * 10:06d}.tif"))
        nbytes += a.nbytes
        p.STACKS[r][c] = tsproject.Stack(r, c, name, ABS_V=r * (V - ov), ABS_H=c * (H - ov), N_BYTESxCHAN=2, z_ranges=[(0, D)])
t_gen = time.perf_counter() - t0
del tiles
torch.cuda.empty_cache()
for rep in range(2):
    q = tsproject.Project(os.path.join(root, "tiles"), rows, cols, D, VXL=(1.0, 1.0, 1.0), ORG=(0.0, 0.0, 0.0), MEC=(float(V - ov), float(H - ov)))
    q.STACKS = p.STACKS
    t0 = time.perf_counter()
    n = q.computeDisplacements(displ_max_V=25, displ_max_H=25, displ_max_D=10, subvol_DIM_D=200, device=dev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    raw = [q._read_slices(s, 0, D - 1) for row in q.STACKS for s in row]
    t_read = time.perf_counter() - t1
    del raw
    print(f"run {rep}: {rows} x {cols} tiles of {H} x {V} x {D} uint16 = {nbytes / 1e9:.1f} GB of slices (written in {t_gen:.1f} s), "
          f"{'Pillow' if os.environ.get('MI_TIFF_PILLOW') else 'the library reader'}: computeDisplacements {dt:.2f} s for {n} pairs = {n / dt:.1f} pairs/s; "
          f"reading the slices alone {t_read:.2f} s = {nbytes / t_read / 1e9:.2f} GB/s", flush=True)
shutil.rmtree(root, ignore_errors=True)
