import cProfile, pstats, sys, os
sys.path.insert(0, os.getcwd())
import torch, bench_ncc
from ipp_amd import crossmips
dev = torch.device("cuda", 0)
tiles, jit, step = bench_ncc.make_grid(dev)
crossmips.compute_displacements(tiles, 307, 307, 25, 25, 10)
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    crossmips.compute_displacements(tiles, 307, 307, 25, 25, 10)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
os._exit(0)
