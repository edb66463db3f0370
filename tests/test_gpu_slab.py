"""GPU: the slab driver with the HIP context (mi_rl_create_ex: per-axis boundary + explicit PSF placement) and
the pack/unpack kernels; several slabs are run lock-step on the one GPU of the test box and must reproduce the
un-sharded result.  (The RCCL transport itself is exercised by the driver's multi-GPU bench; its message pattern
is covered on CPU ranks by tests/test_slab.py.)"""
import numpy as np
import pytest
import torch

from oracle import rl_oracle as R
from tests.slab_util import lockstep_iterate

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / np.abs(b).max())


@pytest.mark.parametrize("flavour", ["fft", "spatial"])
@pytest.mark.parametrize("engine", [1, 2])
@pytest.mark.parametrize("world", [1, 4])
def test_lockstep_slabs_on_gpu(dev, flavour, engine, world):
    from ipp_amd import decon, slab
    psf = R.gaussian_psf((5, 7, 5), (1.0, 1.5, 1.0))
    vol = R.bead_volume((12, 64, 24), seed=31, psf=psf)
    slabs = [slab.SlabRL(vol.shape, psf, rank=r, world_size=world, device=dev, flavour=flavour, engine=engine, volume=vol)
             for r in range(world)]
    got = lockstep_iterate(slabs, 3).cpu().numpy()
    if flavour == "fft":
        want = R.decon_fft(vol, psf, vol.shape, 3, skip_edgetaper=True)
        single = decon.decon(torch.from_numpy(vol).to(dev), psf, 3, 0, 0, 0, 1, True, (24, 64, 12), False,
                             skip_edgetaper=True, engine=engine).cpu().numpy()
    else:
        want = R.decon_spatial(vol, psf, 3, skip_edgetaper=True)
        single = decon.decon(torch.from_numpy(vol).to(dev), psf, 3, 0, 0, 0, 1, False, None, False, skip_edgetaper=True,
                             engine=engine).cpu().numpy()
    assert _rel(got, want) < 1e-4 and _rel(got, single) < 2e-5


def test_pack_unpack_rows(dev):
    from ipp_amd import slab
    ops = slab.HipOps(dev)
    v = torch.arange(5 * 9 * 7, dtype=torch.float32, device=dev).reshape(5, 9, 7)
    p = ops.pack(v, 2, 3)
    assert torch.equal(p, v[:, 2:5, :])
    w = torch.zeros_like(v)
    ops.unpack(p, w, 6)
    assert torch.equal(w[:, 6:9, :], v[:, 2:5, :]) and float(w[:, :6].abs().sum()) == 0
