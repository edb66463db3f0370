import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ipp_amd import crossmips
dev = torch.device("cuda", 0)
for shape in [(2048, 307), (307, 2048)]:
    a = torch.rand(shape, device=dev); b = torch.rand(shape, device=dev)
    crossmips.compute_NCC_map(a, b, 25, 25); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): crossmips.compute_NCC_map(a, b, 25, 25)
    torch.cuda.synchronize()
    print(shape, "dbg", os.environ.get("MI_NCC_DBG", "0"), f"{(time.perf_counter()-t0)/20*1e3:.3f} ms per map (incl. tables)", flush=True)
