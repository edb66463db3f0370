"""GPU: BASELINE.json's full sizes, checked through size-independent properties (the oracle cannot run these in
seconds): impulse response == PSF at the reference's placement, flux conservation and non-negativity of RL with a
normalised PSF under circular boundary, fused == unfused iterations, direct == FFT engine; and one full-size C5 tile
pair against the C oracle."""
import numpy as np
import pytest
import torch

from oracle import ncc_oracle as N
from oracle import rl_oracle as R
from tests.rl_util import delta_on_ones_closed_form as _delta_on_ones_closed_form

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


def _sum64(t, planes=32):
    """float64 sum of a float32 volume in slabs of `planes` z planes (torch casts the whole operand first: 137 GB for config 4)"""
    return float(sum(torch.sum(t[z:z + planes], dtype=torch.float64) for z in range(0, t.shape[0], planes)))


def test_rl_config4_whole_on_one_gpu(dev):
    """BASELINE config 4 WHOLE on one device: 4096 x 4096 x 1024 voxels (17.2 G: linear voxel indices past 2^32), 63 x 63 x 127 PSF,
    deconFFT semantics.  The reference caps a block at 2^31 - 1 elements (LsDeconv.m:308-385); nothing here does.  Device memory:
    volume 68.7 GB + two spectrum arrays 2 x 77.6 GB (rows of 32 KB + 4 KB + 128 B of padding against channel camping) + real OTF
    34.4 GB = 258 GB of the 288 -- so only the fused iteration runs (the two half-steps need a fourth array), and it is checked (i) against the closed form of one iteration on 1 + amp * delta for two
    impulses, one of them at a linear index above 2^32 and next to the wrap-around of x and y: forward placement (decon.m:131-133),
    mirrored adjoint, both through 64-bit addressing; (ii) by flux conservation and non-negativity over two more iterations on the
    bead volume of the bench."""
    import bench
    from ipp_amd import capi, decon
    free_b, total_b = torch.cuda.mem_get_info(dev)
    if free_b < 265e9:
        pytest.skip(f"config 4 whole needs about 258 GB on the device: {free_b / 1e9:.1f} GB free of {total_b / 1e9:.1f} GB")
    shape, kshape = bench.WORKLOADS["c4"]
    psf = bench.make_psf(kshape)
    ctx = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
    assert ctx.engine == capi.ENGINE_FFT and ctx.fuses
    assert ctx.device_bytes < 195e9, ctx.device_bytes   # 2 spectrum arrays + the real OTF; no trial candidates, no spare array
    shifts = [n // 2 - (n - k) // 2 for n, k in zip(shape, psf.shape)]
    amp = 10.0 / float(psf[shifts[0], shifts[1], shifts[2]])
    bl = torch.ones(shape, device=dev)
    hi = (700, 4070, 4090)                                # linear index 1.18e10 > 2^32; x and y neighbourhoods wrap around
    lo = (100, 150, 200)
    assert (hi[0] * shape[1] + hi[1]) * shape[2] + hi[2] > 2 ** 32
    for p in (hi, lo):
        bl[p] = 1.0 + amp
    ctx.iterate(bl, None, 1)
    offs = [(0, 0, 0), (1, 0, 0), (0, -1, 0), (0, 0, 1), (5, -7, 11), (-20, 3, -2), (40, 25, -30), (-62, -30, 31), (63, 31, -31),
            (-100, 40, 40), (3, 60, -5)]
    for p in (hi, lo):
        for d in offs:
            y = tuple((a + b) % n for a, b, n in zip(p, d, shape))
            want = _delta_on_ones_closed_form(psf, shifts, amp, d)
            assert float(bl[y]) == pytest.approx(want, rel=2e-4), (p, d)
    assert float(bl[512, 2048, 2048]) == pytest.approx(1.0, rel=1e-5)      # far from both impulses nothing moves
    n = float(bl.numel())
    assert abs(_sum64(bl) - (n + 2 * amp)) / n < 1e-5   # flux of the iteration (sum(psf) = 1)
    # the bench's bead volume, generated into the same array
    g = torch.Generator(device=dev).manual_seed(1234)
    bl.uniform_(0.01, 0.02, generator=g)
    nb = bl.numel() // 4096
    idx = torch.randint(0, bl.numel(), (nb,), generator=g, device=dev)
    bl.view(-1).index_put_((idx,), torch.empty(nb, device=dev).uniform_(0.2, 1.0, generator=g), accumulate=True)
    del idx
    s0 = _sum64(bl)
    ctx.iterate(bl, None, 2)
    assert float(bl.min()) >= 0.0 and np.isfinite(float(bl.max()))
    assert abs(_sum64(bl) - s0) / s0 < 1e-4
    del bl, ctx
    torch.cuda.empty_cache()
    capi.release_cached_memory()


C4_G, C4_K = (1024, 4096, 4096), (127, 63, 63)


def _c4_rank_geometry():
    from ipp_amd import capi, slab
    sy = slab.psf_shift(C4_G[1], C4_K[1], "fft")
    h = max(sy, C4_K[1] - 1 - sy)
    rows = capi.lib().mi_fft_good_size(C4_G[1] // 8 + 2 * h, 1)
    assert rows == 576 and h == 32                      # 512 interior + 2 x 32 halo rows -> 576 = 9 * 64 (radix-9 stage)
    shifts = (slab.psf_shift(C4_G[2], C4_K[2], "fft"), sy, slab.psf_shift(C4_G[0], C4_K[0], "fft"))
    return (C4_G[0], rows, C4_G[2]), shifts, h


def test_c4_shaped_slab_rank_impulse_forward_and_adjoint(dev):
    """One rank of BASELINE config 4 (4096 x 4096 x 1024 over 8 GPUs, 63 x 63 x 127 PSF): forward impulse response == PSF at
    deconFFT's placement, adjoint impulse response == mirrored PSF (decon.m:131-133,162-172)."""
    import bench
    from ipp_amd import capi, decon
    shape, shifts, h = _c4_rank_geometry()
    psf = bench.make_psf(C4_K)
    ctx = decon.RLContext(shape, psf, None, boundary=(2, 2, 2), engine=capi.ENGINE_FFT, device=dev, shift_xyz=shifts)
    assert ctx.engine == capi.ENGINE_FFT
    # _impulse_response_check derives the shifts from the array shape; the rank-local y extent (576) gives the same y shift (32)
    assert [n // 2 - (n - k) // 2 for n, k in zip(shape, psf.shape)] == [shifts[2], shifts[1], shifts[0]]
    _impulse_response_check(ctx, shape, psf, dev, (100, h + 50, 77))


def test_c4_shaped_slab_rank_fused_iterations(dev):
    """Two fused iterations on a ring-closed single slab of a C4 rank's local shape (1024 x 576 x 4096: 512 interior rows + 2 x 32
    halo rows, x-transformed halo rows wrap around onto the slab itself): flux conservation and non-negativity -- properties of
    the whole ring, which one slab closes on itself."""
    import bench
    from ipp_amd import slab
    psf = bench.make_psf(C4_K)
    drv = slab.SlabRL((C4_G[0], C4_G[1] // 8, C4_G[2]), psf, rank=0, world_size=1, device=dev, flavour="fft", engine=2, seed=7)
    assert drv.lshape == (1024, 576, 4096) and drv.h == 32 and drv.sharded
    s0 = float(drv.interior().double().sum())
    m0 = float(drv.interior().max())
    drv.run(2)
    out = drv.interior()
    assert float(out.min()) >= 0.0 and bool(torch.isfinite(out).all())
    assert abs(float(out.double().sum()) - s0) / s0 < 1e-4
    assert float(out.max()) > m0                         # beads sharpen
    # the sharded steps never run the fused loop that settles the spare S array of large plans (9.7 GB here): it is released by
    # their first call -- two spectrum arrays and the real OTF (half an array) are what stays
    from ipp_amd import capi
    spec = int(capi.lib().mi_rl_fft_spectrum_bytes(drv.ctx._h))
    assert spec > 9e9 and int(capi.lib().mi_rl_device_bytes(drv.ctx._h)) < 3.2 * spec


def test_c4_shaped_slab_rank_edgetaper(dev):
    """edgetaper_3d (edgetaper_3d.m:13-44) with the 63 x 63 x 127 PSF on an array of a C4 rank's shape.  The volume is
    a(z) + b(y) + c(x), so the replicate-boundary blur is the sum of three 1-D blurs with the PSF's marginals and every
    voxel of the result is known in closed form (float64) -- checked on whole lines through the shell and the interior."""
    import bench
    from ipp_amd import decon
    shape, _, _ = _c4_rank_geometry()
    psf = bench.make_psf(C4_K)
    nz, ny, nx = shape
    rng = np.random.default_rng(11)
    comp = [0.2 + 0.1 * np.sin(np.arange(n) * w) + 0.05 * rng.random(n) for n, w in zip(shape, (0.05, 0.11, 0.013))]
    a, b, c = (torch.from_numpy(v.astype(np.float32)).to(dev) for v in comp)
    vol = (a[:, None, None] + b[None, :, None]) + c[None, None, :]
    vol_in = {}
    lines = [(0, 5, 9), (0, 300, 2000), (1, 3, 4090), (1, 512, 1000), (2, 2, 2), (2, 1020, 570), (2, 500, 288)]  # (axis, i, j)
    def take(t, ax, i, j):
        return (t[:, i, j] if ax == 0 else t[i, :, j] if ax == 1 else t[i, j, :]).cpu().numpy().astype(np.float64)
    for ln in lines:
        vol_in[ln] = take(vol, *ln)
    out = decon.edgetaper_3d(vol, torch.from_numpy(psf).to(dev))
    pn = psf.astype(np.float64) / float(psf.astype(np.float32).sum(dtype=np.float32))
    marg = [pn.sum(axis=(1, 2)), pn.sum(axis=(0, 2)), pn.sum(axis=(0, 1))]
    comp32 = [v.astype(np.float32).astype(np.float64) for v in comp]
    blur1 = []
    for v, m in zip(comp32, marg):                       # conv3d_gpu.cu:77-98 along one axis: centre k/2, clamped index
        k = m.size
        idx = np.clip(np.arange(v.size)[:, None] + (k // 2) - np.arange(k)[None, :], 0, v.size - 1)
        blur1.append((v[idx] * m[None, :]).sum(axis=1))
    tz, ty, tx = R.edgetaper_mask_vectors(shape, psf.shape)
    for ax, i, j in lines:
        if ax == 0:
            blur = blur1[0] + blur1[1][i] + blur1[2][j]; mask = tz.astype(np.float64) * float(ty[i]) * float(tx[j])
        elif ax == 1:
            blur = blur1[0][i] + blur1[1] + blur1[2][j]; mask = float(tz[i]) * ty.astype(np.float64) * float(tx[j])
        else:
            blur = blur1[0][i] + blur1[1][j] + blur1[2]; mask = float(tz[i]) * float(ty[j]) * tx.astype(np.float64)
        want = mask * vol_in[(ax, i, j)] + (1.0 - mask) * blur
        got = take(out, ax, i, j)
        assert np.abs(got - want).max() < 1e-5, (ax, i, j, float(np.abs(got - want).max()))   # edgetaper_3d_test.m:4,40
    # where the mask is exactly one the block is untouched
    zi, yi, xi = 500, 288, 2000
    assert tz[zi] == 1 and ty[yi] == 1 and tx[xi] == 1
    core = out[zi, yi, 1000:3000].cpu().numpy().astype(np.float64)
    assert np.array_equal(core, vol_in[(2, 500, 288)][1000:3000])


@pytest.mark.parametrize("direction", [1, 0], ids=["west_east", "north_south"])
def test_ncc_full_size_pair_vs_oracle(dev, direction):
    """One full-size C5 pair per direction (2048 x 2048 x 32 tiles, 307-px overlap, search (25, 25, 10)) against the C oracle: all
    nine scalars of the record -- V, H and D offsets, the three peaks, the three widths -- and the mutated wRangeThr
    (libcrossmips.cpp:275-277, 339-481)."""
    import bench_ncc
    from ipp_amd import crossmips
    tiles, jit, step = bench_ncc.make_grid(dev, rows=2, cols=2, seed=4321)
    a, b = tiles[0][0], (tiles[0][1] if direction == 1 else tiles[1][0])
    d = crossmips.PDAlgoMIPNCC.execute(a, b, *bench_ncc.DISPL, direction, bench_ncc.OVERLAP)
    want = N.pdalgo_execute(a.cpu().numpy(), b.cpu().numpy(), *bench_ncc.DISPL, direction, bench_ncc.OVERLAP, kind="oracle")
    assert d.VHD_coords == want["coord"] and d.NCC_widths == want["NCC_widths"] and d.wRangeThrs == want["wRangeThr"]
    assert np.allclose(np.array(d.NCC_maxs, np.float32), want["NCC_maxs"], atol=2e-6, equal_nan=True)
    dj = jit[0, 1] - jit[0, 0] if direction == 1 else jit[1, 0] - jit[0, 0]
    nominal = (0, step) if direction == 1 else (step, 0)
    assert d.VHD_coords[0] == nominal[0] + int(dj[0]) and d.VHD_coords[1] == nominal[1] + int(dj[1])
    assert d.VHD_coords[2] == int(dj[2])                 # the D offset of the jittered cut
