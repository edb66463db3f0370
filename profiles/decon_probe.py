import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from ipp_amd import decon
dev = torch.device("cuda", 0)
shape, kshape = bench.WORKLOADS["c3"]
psf = bench.make_psf(kshape)
psf_t = torch.from_numpy(psf).to(dev)
bl = bench.make_volume(shape, dev)
def t(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter()-t0)*1e3:.0f} ms   (torch reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB, free {torch.cuda.mem_get_info()[0]/2**30:.0f} GiB)", flush=True)
fs = (shape[2], shape[1], shape[0])
for i in range(3):
    t(f"edgetaper #{i}", lambda: decon.edgetaper_3d(bl, psf_t))
t("decon 2 it skip_edgetaper", lambda: decon.decon(bl, psf, 2, 0.0, 0.0, 0, 1, True, fs, False, skip_edgetaper=True))
for i in range(2):
    t(f"edgetaper after decon #{i}", lambda: decon.edgetaper_3d(bl, psf_t))
t("decon 2 it", lambda: decon.decon(bl, psf, 2, 0.0, 0.0, 0, 1, True, fs, False))
t("decon 2 it", lambda: decon.decon(bl, psf, 2, 0.0, 0.0, 0, 1, True, fs, False))
t("decon 0 it skip_edgetaper (context build only)", lambda: decon.decon(bl, psf, 0, 0.0, 0.0, 0, 1, True, fs, False, skip_edgetaper=True))
t("decon 6 it skip_edgetaper", lambda: decon.decon(bl, psf, 6, 0.0, 0.0, 0, 1, True, fs, False, skip_edgetaper=True))
with decon.DeconPlan(1) as plan:
    for i in range(3):
        t(f"decon 6 it with a plan #{i}", lambda: decon.decon(bl, psf, 6, 0.0, 0.0, 0, 1, True, fs, False, plan=plan))
t("decon 6 it without a plan", lambda: decon.decon(bl, psf, 6, 0.0, 0.0, 0, 1, True, fs, False))
