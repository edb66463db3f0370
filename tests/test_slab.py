"""CPU: the multi-GPU slab driver (sharding, halo exchange, ring / zero edges) with world_size-2 gloo ranks and a
numpy convolution context injected in place of the HIP one; results must equal the un-sharded oracle."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import rl_oracle as R
from tests.slab_util import NumpyOps, lockstep_iterate


def _case(seed=21, shape=(10, 40, 18), kshape=(5, 7, 3)):
    psf = R.gaussian_psf(kshape, (1.2, 1.6, 0.8))
    return R.bead_volume(shape, seed=seed, psf=psf), psf


def _rel(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / np.abs(b).max())


def _worker(rank, world, port, flavour, vol, psf, niter, out, fuses=0, transport="rccl", zchunks=1):
    import torch.distributed as dist
    from ipp_amd import slab
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        drv = slab.SlabRL(vol.shape, psf, rank=rank, world_size=world, flavour=flavour, volume=vol, ops=NumpyOps(fuses),
                          transport=transport, zchunks=zchunks)
        assert drv.sharded == bool(fuses) and drv.overlap == (fuses == 2) and (drv.zb is not None) == (zchunks > 1 and fuses == 2)
        n0 = drv.norm2()
        drv.run(niter)
        if transport == "peer":
            # exchanges so far (fused protocol: 1 + 2 per iteration; real-space protocol: 2 per iteration), both buffer sets used
            assert drv.link is not None and drv.link.n == (2 * niter + 1 if fuses else 2 * niter) and drv.link.n > drv.link.SETS
            assert drv.link.C == (len(drv.zb) if drv.zb is not None else 1)
        full = drv.gather()
        drv.close()
        if rank == 0:
            out.put((full.numpy(), n0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("flavour", ["fft", "spatial"])
@pytest.mark.parametrize("transport", ["rccl", "peer"], ids=["send_recv", "peer_copy"])
@pytest.mark.parametrize("fuses", [0, 1, 2], ids=["real_halos", "spectrum_halos", "spectrum_halos_overlapped"])
def test_two_gloo_ranks_equal_unsharded_oracle(flavour, fuses, transport):
    """Both halo protocols: real-space rows around forward_ratio / adjoint_update, and x-transformed rows around the fused
    steps (the protocol of the native FFT pipeline); both transports: grouped send / receive of the process group, and the
    copy-engine link (buffers of the neighbour written directly, sequence numbers in shared memory; host double of the device
    side: tests/slab_util.py:ShmPeer)."""
    vol, psf = _case()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29600 + (os.getpid() % 200) + (0 if flavour == "fft" else 1) + 2 * fuses + (6 if transport == "peer" else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, flavour, vol, psf, 3, out, fuses, transport)) for r in range(2)]
    for p in procs:
        p.start()
    got, n0 = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if flavour == "fft":
        want = R.decon_fft(vol, psf, vol.shape, 3, skip_edgetaper=True)
    else:
        want = R.decon_spatial(vol, psf, 3, skip_edgetaper=True)
    assert got.shape == vol.shape and _rel(got, want) < 2e-5
    assert n0 == pytest.approx(float(np.linalg.norm(vol.astype(np.float64))), rel=1e-6)


@pytest.mark.parametrize("flavour", ["fft", "spatial"])
@pytest.mark.parametrize("transport", ["rccl", "peer"], ids=["send_recv", "peer_copy"])
@pytest.mark.parametrize("zchunks", [2, 4])
def test_two_gloo_ranks_z_chunked_exchange(flavour, transport, zchunks):
    """The halo rows travel in z chunks: a chunk leaves as soon as the x tiles of its planes have run, the next half-step's
    y-forward pass starts on a chunk as soon as its rows have landed (SlabRL._iterate_zchunked), through both transports; the
    double of the context snapshots a chunk's planes when they are transformed, so a chunk that is used before its halo rows
    have been delivered gives a wrong result.  The exchange left in flight after the last iteration is drained."""
    vol, psf = _case()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29650 + (os.getpid() % 150) + (0 if flavour == "fft" else 1) + 2 * zchunks + (11 if transport == "peer" else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, flavour, vol, psf, 3, out, 2, transport, zchunks)) for r in range(2)]
    for p in procs:
        p.start()
    got, _ = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = (R.decon_fft(vol, psf, vol.shape, 3, skip_edgetaper=True) if flavour == "fft"
            else R.decon_spatial(vol, psf, 3, skip_edgetaper=True))
    assert _rel(got, want) < 2e-5


@pytest.mark.parametrize("flavour", ["fft", "spatial"])
@pytest.mark.parametrize("world", [1, 3])
@pytest.mark.parametrize("zchunks", [2, 3, 16])
def test_lockstep_slabs_z_chunked(flavour, world, zchunks):
    """Several slabs in one process through the z-chunked stages, and a single self-ring slab through the driver's own iterate()."""
    from ipp_amd import slab
    from tests.slab_util import lockstep_iterate_zchunked
    vol, psf = _case(seed=6, shape=(9, 41, 16))
    want = (R.decon_fft(vol, psf, vol.shape, 2, skip_edgetaper=True) if flavour == "fft"
            else R.decon_spatial(vol, psf, 2, skip_edgetaper=True))
    slabs = [slab.SlabRL(vol.shape, psf, rank=r, world_size=world, flavour=flavour, volume=vol, ops=NumpyOps(2), zchunks=zchunks)
             for r in range(world)]
    assert all(s.zb is not None and s.zb[0][0] == 0 and s.zb[-1][1] == s.nzp and len(s.zb) <= zchunks for s in slabs)
    assert _rel(lockstep_iterate_zchunked(slabs, 2).numpy(), want) < 2e-5
    if world == 1:
        one = slab.SlabRL(vol.shape, psf, rank=0, world_size=1, flavour=flavour, volume=vol, ops=NumpyOps(2), zchunks=zchunks)
        one.run(1)                                       # (run drains; a second run goes on from the drained state)
        one.run(1)
        assert _rel(one.interior().numpy(), want) < 2e-5


@pytest.mark.parametrize("flavour", ["fft", "spatial"])
@pytest.mark.parametrize("world", [1, 3, 4])
@pytest.mark.parametrize("fuses", [0, 1, 2], ids=["real_halos", "spectrum_halos", "spectrum_halos_overlapped"])
def test_lockstep_slabs_uneven_rows(flavour, world, fuses):
    from ipp_amd import slab
    vol, psf = _case(seed=5, shape=(9, 41, 16))  # 41 rows: uneven split; odd extent: centred deconFFT placement
    slabs = [slab.SlabRL(vol.shape, psf, rank=r, world_size=world, flavour=flavour, volume=vol, ops=NumpyOps(fuses))
             for r in range(world)]
    assert sum(s.n_loc for s in slabs) == 41
    got = lockstep_iterate(slabs, 2).numpy()
    want = (R.decon_fft(vol, psf, vol.shape, 2, skip_edgetaper=True) if flavour == "fft"
            else R.decon_spatial(vol, psf, 2, skip_edgetaper=True))
    assert _rel(got, want) < 2e-5


def test_slab_rows_and_halo_rules():
    from ipp_amd import slab
    assert slab.slab_rows(2048, 8) == [(256 * r, 256 * (r + 1)) for r in range(8)]
    assert slab.slab_rows(10, 4) == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert slab.psf_shift(2048, 31, "fft") == 16 and slab.psf_shift(2047, 31, "fft") == 15  # decon.m:131-133 quirk
    assert slab.psf_shift(2048, 31, "spatial") == 15
    vol, psf = _case(shape=(6, 16, 8), kshape=(3, 9, 3))
    with pytest.raises(ValueError, match="thinner than the halo"):
        slab.SlabRL(vol.shape, psf, rank=0, world_size=4, flavour="fft", volume=vol, ops=NumpyOps())


@pytest.mark.parametrize("flavour", ["fft", "spatial"])
@pytest.mark.parametrize("kshape", [(5, 7, 3), (4, 6, 2)])
@pytest.mark.parametrize("fuses", [0, 2], ids=["real_halos", "spectrum_halos_overlapped"])
def test_lockstep_slabs_asymmetric_psf(flavour, kshape, fuses):
    """Asymmetric PSFs (odd and even extents): the halo widths of the forward and the adjoint convolution differ and, for the
    spatial flavour with even extents, the adjoint kernel is the flipped PSF at convn's 'same' centre, not the transpose."""
    from ipp_amd import slab
    from tests.rl_util import asymmetric_psf
    psf = asymmetric_psf(kshape, seed=4)
    vol = R.bead_volume((9, 44, 16), seed=8, psf=R.gaussian_psf((3, 3, 3), (1, 1, 1)))
    slabs = [slab.SlabRL(vol.shape, psf, rank=r, world_size=3, flavour=flavour, volume=vol, ops=NumpyOps(fuses)) for r in range(3)]
    got = lockstep_iterate(slabs, 3).numpy()
    want = (R.decon_fft(vol, psf, vol.shape, 3, skip_edgetaper=True) if flavour == "fft"
            else R.decon_spatial(vol, psf, 3, skip_edgetaper=True))
    assert _rel(got, want) < 2e-5
