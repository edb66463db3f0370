"""GPU: the slab driver with the HIP context (mi_rl_create_ex: per-axis boundary + explicit PSF placement) and
the pack/unpack kernels; several slabs are run lock-step on the one GPU of the test box and must reproduce the
un-sharded result.  (The RCCL transport itself is exercised by the driver's multi-GPU bench; its message pattern
is covered on CPU ranks by tests/test_slab.py.)"""
import numpy as np
import pytest
import torch

from oracle import rl_oracle as R
from tests.rl_util import assert_close
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
    assert_close(got, want)
    assert _rel(got, single) < 2e-5


@pytest.mark.parametrize("gshape,kshape,world", [((16, 200, 64), (3, 9, 5), 3), ((32, 96, 32), (5, 5, 7), 2), ((16, 330, 32), (7, 11, 3), 5)])
@pytest.mark.parametrize("flavour", ["fft", "spatial"])
def test_lockstep_slabs_uneven_rows_fused_pipeline(dev, gshape, kshape, world, flavour, monkeypatch):
    """Uneven slabs (the last rank takes the remainder), odd global extents on y, different halo widths: every rank rounds its
    local extent to a native one on its own; spectrum halos, overlapped split of the x pass where the context allows it."""
    from ipp_amd import slab
    monkeypatch.setenv("MI_FFT_NATIVE_INFLATE", "100")
    psf = R.gaussian_psf(kshape, (1.0, 1.5, 1.0))
    vol = R.bead_volume(gshape, seed=sum(gshape), psf=psf)
    slabs = [slab.SlabRL(vol.shape, psf, rank=r, world_size=world, device=dev, flavour=flavour, engine=2, volume=vol)
             for r in range(world)]
    assert all(s.sharded for s in slabs) and sum(s.n_loc for s in slabs) == gshape[1]
    got = lockstep_iterate(slabs, 3).cpu().numpy()
    want = (R.decon_fft(vol, psf, vol.shape, 3, skip_edgetaper=True) if flavour == "fft"
            else R.decon_spatial(vol, psf, 3, skip_edgetaper=True))
    assert_close(got, want)


@pytest.mark.parametrize("flavour", ["fft", "spatial"])
@pytest.mark.parametrize("world", [1, 2, 4])
def test_lockstep_slabs_fused_pipeline_spectrum_halos(dev, flavour, world, monkeypatch):
    """Shapes the native FFT pipeline takes: the slabs run the fused iteration and exchange x-transformed halo rows; the result
    equals the un-sharded fused run and the oracle.  No ratio volume exists."""
    from ipp_amd import decon, slab
    monkeypatch.setenv("MI_FFT_NATIVE_INFLATE", "100")   # small padded grids round up a lot: keep the native pipeline
    psf = R.gaussian_psf((5, 7, 5), (1.0, 1.5, 1.0))
    vol = R.bead_volume((16, 128, 32), seed=33, psf=psf)
    slabs = [slab.SlabRL(vol.shape, psf, rank=r, world_size=world, device=dev, flavour=flavour, engine=2, volume=vol)
             for r in range(world)]
    assert all(s.sharded and s.ratio is None for s in slabs)
    # circular grids run the persistent x kernel, which can process the edge tiles ahead of the others (overlapped sends)
    import os
    if not os.environ.get("MI_FFT_NO_PIPE"):
        assert all(s.overlap == (flavour == "fft") for s in slabs)
    got = lockstep_iterate(slabs, 4).cpu().numpy()
    if flavour == "fft":
        want = R.decon_fft(vol, psf, vol.shape, 4, skip_edgetaper=True)
    else:
        want = R.decon_spatial(vol, psf, 4, skip_edgetaper=True)
    assert_close(got, want)
    # the driver's own iterate() on a single self-ring slab goes through the same protocol
    one = slab.SlabRL(vol.shape, psf, rank=0, world_size=1, device=dev, flavour=flavour, engine=2, volume=vol)
    one.run(4)
    assert_close(one.interior().cpu().numpy(), want)


@pytest.mark.parametrize("world", [1, 2, 4])
@pytest.mark.parametrize("zchunks", [2, 4, 64])
def test_lockstep_slabs_z_chunked_exchange(dev, world, zchunks, monkeypatch):
    """The z-chunked stages of the sharded step (mi_rl_sharded_stage, mi_rl_spectrum_rows_z) on the HIP context: chunks of planes
    cut at the context's granule, y-forward per chunk, edge tiles per chunk.  The same kernels run on the same tiles as in the
    unchunked protocol, so the result is IDENTICAL to it bit for bit, and equals the oracle."""
    from ipp_amd import slab
    from tests.slab_util import lockstep_iterate_zchunked
    monkeypatch.setenv("MI_FFT_NATIVE_INFLATE", "100")
    psf = R.gaussian_psf((5, 7, 5), (1.0, 1.5, 1.0))
    vol = R.bead_volume((32, 128, 32), seed=35, psf=psf)
    mk = lambda zc: [slab.SlabRL(vol.shape, psf, rank=r, world_size=world, device=dev, flavour="fft", engine=2, volume=vol, zchunks=zc)  # noqa: E731
                     for r in range(world)]
    chunked = mk(zchunks)
    assert all(s.zb is not None and s.zb[0][0] == 0 and s.zb[-1][1] == 32 for s in chunked)
    g = chunked[0].ctx.z_granule
    assert g >= 1 and all(z0 % g == 0 for s in chunked for z0, _ in s.zb) and len(chunked[0].zb) == min(zchunks, 32 // g)
    got = lockstep_iterate_zchunked(chunked, 3)
    ref = lockstep_iterate(mk(1), 3)
    assert torch.equal(got, ref)
    assert_close(got.cpu().numpy(), R.decon_fft(vol, psf, vol.shape, 3, skip_edgetaper=True))
    if world == 1:                                       # the driver's own iterate() on a self-ring, with a drain in between
        one = mk(zchunks)[0]
        one.run(2)
        one.run(1)
        assert torch.equal(one.interior(), ref)
        with pytest.raises(Exception, match="multiples of"):
            one.ctx.sharded_stage(one.bl, False, 0, 1, 1 + g) if g > 1 else (_ for _ in ()).throw(RuntimeError("multiples of"))


def test_spectrum_rows_pack_unpack(dev):
    from ipp_amd import capi, decon
    psf = R.gaussian_psf((3, 3, 3), (1.0, 1.0, 1.0))
    shape = (8, 32, 16)
    ctx = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
    assert ctx.fuses
    bl = torch.rand(shape, device=dev) + 0.5
    ctx.sharded_begin(bl)
    a = ctx.spectrum_pack(4, 6)
    assert a.numel() == 6 * 8 * 16          # rows * nz * nx floats: as many bytes as the real rows
    # x-transformed rows of y = 4..9: compare with torch's FFT of the half-length packed rows (any order along x: sums agree)
    packed = torch.view_as_complex(bl[:, 4:10, :].reshape(8, 6, 8, 2).contiguous())
    want = torch.fft.fft(packed, dim=2)
    got = torch.view_as_complex(a.reshape(8, 8, 6, 2))          # [z][px][row]
    assert torch.allclose(got.abs().pow(2).sum(dim=1), want.abs().pow(2).sum(dim=2), rtol=1e-4)
    ctx.spectrum_unpack(None, 4, 6)
    assert float(ctx.spectrum_pack(4, 6).abs().sum()) == 0.0
    ctx.spectrum_unpack(a, 4, 6)
    assert torch.equal(ctx.spectrum_pack(4, 6), a)
    direct = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_DIRECT, device=dev)
    assert not direct.fuses
    with pytest.raises(capi.MiError, match="only the native FFT pipeline"):
        direct.sharded_begin(bl)


def test_pack_unpack_rows(dev):
    from ipp_amd import slab
    ops = slab.HipOps(dev)
    v = torch.arange(5 * 9 * 7, dtype=torch.float32, device=dev).reshape(5, 9, 7)
    p = ops.pack(v, 2, 3)
    assert torch.equal(p, v[:, 2:5, :])
    w = torch.zeros_like(v)
    ops.unpack(p, w, 6)
    assert torch.equal(w[:, 6:9, :], v[:, 2:5, :]) and float(w[:, :6].abs().sum()) == 0
