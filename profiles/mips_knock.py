"""k_mips on the C5 grid with parts of its work switched off (MI_NCC_MIPS_KNOCK: 1 no xz maxima, 2 no yz maxima, 4 no xy store,
): which part of the pass costs what.    python profiles/mips_knock.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import torch
    import bench_ncc
    dev = torch.device("cuda", 0)
    tiles, jit, step = bench_ncc.make_grid(dev)
    from ipp_amd import crossmips
    pairs = list(crossmips.enumerate_pairs(len(tiles), len(tiles[0])))
    r = bench_ncc.mips_roofline(dev, tiles, pairs)
    print({k: (round(v["launch_ms"], 4), round(v["GBps"])) for k, v in r["launches"].items()}, flush=True)
    sys.exit(0)
for knock in (0, 1, 2, 4, 7):
    env = dict(os.environ, MI_NCC_MIPS_KNOCK=str(knock))
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, capture_output=True, text=True, timeout=300)
    print(f"knock {knock:2d}: {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]}", flush=True)
