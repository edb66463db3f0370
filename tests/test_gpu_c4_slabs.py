"""GPU: the 8-rank geometry of BASELINE config 4 (4096 x 4096 x 1024 volume, 63 x 63 x 127 PSF, 8 slabs along y) on ONE GPU.

The y x z cross-section is the full one -- 4096 rows cut into 8 slabs of 512 + 2 x 32 halo rows -> 576 = 9 * 64 local rows,
z = 1024, the 63 x 63 x 127 LsMakePSF PSF -- and x is reduced to 512 so that eight rank contexts and the un-sharded reference
context fit the card together (the full 4096-voxel x extent changes nothing about the halo geometry: x is never sharded; it
waits for the 8-GPU node).  The slabs run lock-step, exchanging exactly what eight ranks would send each other
(x-transformed halo rows, split x pass; tests/slab_util.py), and must reproduce the un-sharded result of the same library on
the same volume [decon.m:162-186; block geometry LsDeconv.m:341,401-403; split_stack.m:9-26]."""
import numpy as np
import pytest
import torch

from tests.rl_util import assert_close_device
from tests.slab_util import lockstep_iterate

pytestmark = pytest.mark.gpu

G, K, WORLD = (1024, 4096, 512), (127, 63, 63), 8


@pytest.fixture(scope="module")
def c4(dev):
    import bench
    from ipp_amd import slab
    psf = bench.make_psf(K)
    vol = bench.make_volume(G, dev, seed=404)
    slabs = [slab.SlabRL(G, psf, rank=r, world_size=WORLD, device=dev, flavour="fft", engine=2, volume=vol) for r in range(WORLD)]
    yield psf, vol, slabs
    del slabs
    torch.cuda.empty_cache()


def test_c4_geometry(c4):
    psf, vol, slabs = c4
    for r, s in enumerate(slabs):
        assert s.lshape == (1024, 576, 512) and s.h == 32 and s.n_loc == 512 and (s.y0, s.y1) == (512 * r, 512 * (r + 1))
        assert s.sharded and s.overlap and s.ctx.pair_layout   # fused iteration, split x pass, paired z pass (1024-point lines)
        assert s.neighbours() == ((r - 1) % WORLD, (r + 1) % WORLD)


def _exchange_real(slabs, vols):
    packed = [s.pack_halos(v) for s, v in zip(slabs, vols)]
    for s, v in zip(slabs, vols):
        lo, hi = s.neighbours()
        s.unpack_halos(v, packed[lo][0], packed[hi][1])


@pytest.mark.parametrize("row", [511, 512, 4095])
def test_c4_impulse_across_the_cut(c4, row):
    """A delta in the last row of slab 0 / the first row of slab 1 / the last row of the volume (the ring closes onto slab 0):
    the forward response is the PSF at deconFFT's placement on BOTH sides of the cut, the adjoint response its mirror image
    (decon.m:131-133,162-172)."""
    psf, vol, slabs = c4
    dev = vol.device
    at = (100, row, 77)
    shifts = [n // 2 - (n - k) // 2 for n, k in zip(G, K)]
    bg = 1e-3
    bls, outs = [], []
    for s in slabs:
        b = torch.full(s.lshape, bg, device=dev)
        if s.y0 <= row < s.y1:
            b[at[0], s.h + row - s.y0, at[2]] += 1.0
        bls.append(b)
        outs.append(torch.empty_like(b))
    _exchange_real(slabs, bls)
    for s, b, o in zip(slabs, bls, outs):
        s.ctx.forward_ratio(b, o)                            # o = b ./ max(conv(b), eps)  ->  conv = b ./ o
    conv = torch.cat([(b / o)[:, s.h:s.h + s.n_loc, :] for s, b, o in zip(slabs, bls, outs)], dim=1)
    rng = np.random.default_rng(row)
    js = [(0, 0, 0), (K[0] - 1, K[1] - 1, K[2] - 1), tuple(shifts)] + [tuple(int(rng.integers(0, k)) for k in K) for _ in range(200)]
    js += [(shifts[0], j, shifts[2]) for j in range(K[1])]   # the whole y line through the centre: every row of the halo
    for j in js:
        p = tuple((a + (jj - s)) % n for a, jj, s, n in zip(at, j, shifts, G))   # forward: sample j lands at at + (j - shift)
        assert float(conv[p]) - bg == pytest.approx(float(psf[j]), rel=2e-4, abs=1e-8), (j, p)
    del conv
    # adjoint: ones <- |1 .* conv_adj(delta)|
    deltas = []
    for s in slabs:
        d = torch.zeros(s.lshape, device=dev)
        if s.y0 <= row < s.y1:
            d[at[0], s.h + row - s.y0, at[2]] = 1.0
        deltas.append(d)
    _exchange_real(slabs, deltas)
    for s, d, o in zip(slabs, deltas, outs):
        o.fill_(1.0)
        s.ctx.adjoint_update(d, o)
    adj = torch.cat([o[:, s.h:s.h + s.n_loc, :] for s, o in zip(slabs, outs)], dim=1)
    for j in js:
        p = tuple((a - (jj - s)) % n for a, jj, s, n in zip(at, j, shifts, G))
        assert float(adj[p]) == pytest.approx(float(psf[j]), rel=2e-4, abs=1e-8), (j, p)


def test_c4_eight_slabs_equal_the_unsharded_volume(c4):
    """Two fused iterations on eight lock-step slabs == two fused iterations of one context on the whole (x-reduced) volume, under
    the three-bound metric of tests/rl_util.py; flux conservation and non-negativity of the whole ring."""
    from ipp_amd import capi, decon
    psf, vol, slabs = c4
    for s in slabs:                                          # (the impulse tests left their volumes in the contexts' buffers)
        s.bl.zero_()
        s.bl[:, s.h:s.h + s.n_loc, :] = vol[:, s.y0:s.y1, :]
        s._begun = False
    s0 = float(vol.double().sum())
    got = lockstep_iterate(slabs, 2)
    assert float(got.min()) >= 0.0 and bool(torch.isfinite(got).all())
    assert abs(float(got.double().sum()) - s0) / s0 < 1e-4
    one = decon.RLContext(G, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=vol.device)
    want = vol.clone()
    one.iterate(want, None, 2)
    assert float(want.max()) > float(vol.max())              # beads sharpen
    assert_close_device(got, want, what="8 slabs vs one context:")
