import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ipp_amd import crossmips
dev = torch.device("cuda", 0)
A = torch.rand((32, 2048, 2048), device=dev); B = torch.rand((32, 2048, 2048), device=dev)
for side, ni, nj in ((0, 1741, 0), (1, 0, 1741)):
    crossmips.compute_mips(A, B, ni, nj, side); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): crossmips.compute_mips(A, B, ni, nj, side)
    torch.cuda.synchronize()
    print("side", side, "dbg", os.environ.get("MI_MIPS_DBG", "0"), f"{(time.perf_counter()-t0)/20*1e3:.3f} ms per MIP set (incl. output allocs)", flush=True)
