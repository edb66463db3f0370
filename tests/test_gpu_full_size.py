"""GPU: BASELINE.json's full sizes, checked through size-independent properties (the oracle cannot run these in
seconds): impulse response == PSF at the reference's placement, flux conservation and non-negativity of RL with a
normalised PSF under circular boundary, fused == unfused iterations, direct == FFT engine; and one full-size C5 tile
pair against the C oracle."""
import numpy as np
import pytest
import torch

from oracle import ncc_oracle as N
from oracle import rl_oracle as R

pytestmark = pytest.mark.gpu


def _impulse_response_check(ctx, shape, psf, dev, at):
    """conv(delta at `at`) must be the PSF around `at` with deconFFT's placement: sample j at offset j - shift,
    shift = F/2 - floor((F - k)/2) per axis (decon.m:131-133, 323-344)."""
    bl = torch.zeros(shape, device=dev)
    bl[at] = 1.0
    ratio = torch.empty_like(bl)
    ctx.forward_ratio(bl, ratio)                       # ratio = bl / max(conv, eps): at `at` it is 1 / conv[at]
    conv_at = 1.0 / float(ratio[at])
    shifts = [n // 2 - (n - k) // 2 for n, k in zip(shape, psf.shape)]
    want = float(psf[shifts[0], shifts[1], shifts[2]])  # the sample that lands on offset 0
    assert conv_at == pytest.approx(want, rel=2e-4)
    # adjoint of a delta picks the mirrored sample: conv_adj[p] = psf[j] with p = at - (j - shift)
    ones = torch.ones_like(bl)
    ctx.adjoint_update(bl, ones)                       # ones <- |1 * conv_adj(delta)|
    for j in [(0, 0, 0), (psf.shape[0] - 1, 3, 5), (shifts[0], shifts[1], shifts[2])]:
        p = tuple((a - (jj - s)) % n for a, jj, s, n in zip(at, j, shifts, shape))
        assert float(ones[p]) == pytest.approx(float(psf[j]), rel=2e-4, abs=1e-9)


@pytest.mark.parametrize("workload", ["c2", "c3"])
def test_rl_full_size_properties(dev, workload):
    import bench
    from ipp_amd import capi, decon
    shape, kshape = bench.WORKLOADS[workload]
    psf = bench.make_psf(kshape)
    ctx = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
    _impulse_response_check(ctx, shape, psf, dev, (shape[0] // 3, 5, shape[2] - 2))
    bl = bench.make_volume(shape, dev)
    s0 = float(bl.double().sum())
    ref = bl.clone()
    ratio = torch.empty_like(bl)
    ctx.iterate(bl, None, 3)                            # fused 8-pass iterations
    for _ in range(3):                                  # the same through the two half-steps
        ctx.forward_ratio(ref, ratio)
        ctx.adjoint_update(ratio, ref)
    assert float(bl.min()) >= 0.0
    assert abs(float(bl.double().sum()) - s0) / s0 < 1e-4          # sum(psf) = 1, circular: flux is conserved
    assert float((bl - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    del ref, ratio, ctx
    torch.cuda.empty_cache()
    if workload == "c2":                                # the direct engine on the same volume (one iteration)
        a = bench.make_volume(shape, dev)
        b = a.clone()
        r = torch.empty_like(a)
        decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev).iterate(a, r, 1)
        decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_DIRECT, device=dev).iterate(b, r, 1)
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max())


def test_c4_shaped_slab_rank(dev):
    """One rank of BASELINE config 4 (4096 x 4096 x 1024 over 8 GPUs, 63 x 63 x 127 PSF): local extent 512 + 2 x 32 halo
    rows -> 576 = 9 * 64 (radix-9 stage).  Flux conservation is a property of the whole ring, so only the interior
    response to an impulse is checked here."""
    from ipp_amd import capi, decon, slab
    import bench
    gshape, kshape = (1024, 4096, 4096), (127, 63, 63)
    psf = bench.make_psf(kshape)
    sy = slab.psf_shift(gshape[1], kshape[1], "fft")
    h = max(sy, kshape[1] - 1 - sy)
    rows = capi.lib().mi_fft_good_size(gshape[1] // 8 + 2 * h, 1)
    assert rows == 576
    shape = (gshape[0], rows, gshape[2])
    shifts = (slab.psf_shift(gshape[2], kshape[2], "fft"), sy, slab.psf_shift(gshape[0], kshape[0], "fft"))
    ctx = decon.RLContext(shape, psf, None, boundary=(2, 2, 2), engine=capi.ENGINE_FFT, device=dev, shift_xyz=shifts)
    assert ctx.engine == capi.ENGINE_FFT
    at = (100, h + 50, 77)
    bl = torch.zeros(shape, device=dev)
    bl[at] = 1.0
    ratio = torch.empty_like(bl)
    ctx.forward_ratio(bl, ratio)
    want = float(psf[shifts[2], shifts[1], shifts[0]])
    assert 1.0 / float(ratio[at]) == pytest.approx(want, rel=2e-4)


def test_ncc_full_size_pair_vs_oracle(dev):
    import bench_ncc
    from ipp_amd import crossmips
    tiles, jit, step = bench_ncc.make_grid(dev, rows=1, cols=2, seed=4321)
    d = crossmips.PDAlgoMIPNCC.execute(tiles[0][0], tiles[0][1], *bench_ncc.DISPL, 1, bench_ncc.OVERLAP)
    want = N.pdalgo_execute(tiles[0][0].cpu().numpy(), tiles[0][1].cpu().numpy(), *bench_ncc.DISPL, 1, bench_ncc.OVERLAP,
                            kind="oracle")
    assert d.VHD_coords == want["coord"] and d.NCC_widths == want["NCC_widths"] and d.wRangeThrs == want["wRangeThr"]
    assert np.allclose(np.array(d.NCC_maxs, np.float32), want["NCC_maxs"], atol=2e-6, equal_nan=True)
    dj = jit[0, 1] - jit[0, 0]
    assert d.VHD_coords[0] == int(dj[0]) and d.VHD_coords[1] == step + int(dj[1])
