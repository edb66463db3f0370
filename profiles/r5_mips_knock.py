"""k_mips5 on the C5 grid with every combination of its knock bits (MI_NCC_MIPS_KNOCK: 1 no xz maxima, 2 no yz maxima, 4 no xy store,
8 no atomic merge; probes build), one process, same tiles.    python profiles/r5_mips_knock.py [knock ...]"""
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
knocks = [int(a) for a in sys.argv[1:]] or list(range(16))
for knock in knocks:
    os.environ["MI_NCC_MIPS_KNOCK"] = str(knock)
    r = bench_ncc.mips_roofline(dev, tiles, pairs)
    print(f"knock {knock:>3}: " + "  ".join(f"{k} {v['launch_ms']:.4f} ms {v['GBps']:.0f} GB/s" for k, v in r["launches"].items()), flush=True)
