"""k_mips5 (round 5) beside k_mips<true> (round 4) on the C5 grid, same process, same tiles: launch time per 56 pairs and GB/s of
the algorithmic bytes, with parts of the work switched off (MI_NCC_MIPS_KNOCK: 1 no xz, 2 no yz, 4 no xy store, 8 no atomic merge).
Needs the probes build (make -C image-preprocessing-pipeline_amd/csrc probes).    MI_IPP_PROBES=1 python profiles/r5_mips_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MI_IPP_PROBES"] = "1"
import torch  # noqa: E402

import bench_ncc  # noqa: E402
from ipp_amd import crossmips  # noqa: E402

dev = torch.device("cuda", 0)
tiles, jit, step = bench_ncc.make_grid(dev)
pairs = list(crossmips.enumerate_pairs(len(tiles), len(tiles[0])))
for rep in range(2):
    for old in ("1", "0"):
        for knock in ("0", "4", "8", "15"):
            os.environ["MI_NCC_MIPS_OLD"] = old
            os.environ["MI_NCC_MIPS_KNOCK"] = knock
            r = bench_ncc.mips_roofline(dev, tiles, pairs)
            print(f"rep {rep} {'k_mips<true>' if old == '1' else 'k_mips5     '} knock {knock:>2}: "
                  + "  ".join(f"{k} {v['launch_ms']:.4f} ms {v['GBps']:.0f} GB/s" for k, v in r["launches"].items()), flush=True)
