"""GPU: the pair-interleaved layout of the spectra around the z pass (fft_native.hip: k_y_pair, k_z_pair_pipe; reference chain
decon.m:162-172 fftn -> .* otf -> ifftn).  Every case runs on the paired layout, on the plain layout of the same library
(MI_FFT_NO_PAIR=1) and against a float64 reference (scipy's circular convolution / the RL oracle)."""
import os

import numpy as np
import pytest
import torch

from oracle import rl_oracle as R
from tests.rl_util import assert_close, asymmetric_psf

pytestmark = pytest.mark.gpu

# (z, y, x): z in 64..1152 (the paired z pass: 2^a, 3 * 2^a, 9 * 2^a), y a multiple of 16 -- powers of two below and above the fast path of k_y_pair
# (y >= 1024), 3 * 2^a and 9 * 2^a (generic path); x small
SHAPES = [(64, 16, 32), (64, 96, 16), (128, 32, 64), (256, 64, 16), (512, 32, 16), (128, 288, 16), (64, 1024, 16), (64, 2048, 16),
          (128, 4096, 16), (256, 1024, 16), (1024, 32, 16), (1024, 96, 32), (64, 160, 16), (128, 320, 32),
          (192, 32, 16), (384, 64, 16), (768, 32, 16), (576, 32, 16), (576, 96, 32), (1152, 16, 16)]   # z = 3 * 2^a, 9 * 2^a


def _contexts(shape, psf, psf_inv, boundary, monkeypatch):
    from ipp_amd import capi, decon
    dev = torch.device("cuda", 0)
    paired = decon.RLContext(shape, psf, psf_inv, boundary=boundary, engine=capi.ENGINE_FFT, device=dev)
    monkeypatch.setenv("MI_FFT_NO_PAIR", "1")
    plain = decon.RLContext(shape, psf, psf_inv, boundary=boundary, engine=capi.ENGINE_FFT, device=dev)
    monkeypatch.delenv("MI_FFT_NO_PAIR")
    assert paired.pair_layout and not plain.pair_layout
    return paired, plain


def _conv_pair(ctx, a, b):
    """(conv(a), conv_adj(b)) through the two half-steps of the context."""
    ra = torch.empty_like(a)
    ctx.forward_ratio(a, ra)  # a ./ max(conv(a), eps): positive data, so conv(a) = a ./ ra
    adj = torch.ones_like(b)
    ctx.adjoint_update(b, adj)  # |1 .* conv_adj(b)|
    return (a / ra).cpu().numpy(), adj.cpu().numpy()


@pytest.mark.parametrize("symmetric", [True, False])
@pytest.mark.parametrize("shape", SHAPES)
def test_paired_layout_equals_plain_layout_and_float64(dev, shape, symmetric, monkeypatch):
    from ipp_amd import capi
    kshape = (7, 5, 9)
    psf = R.gaussian_psf(kshape, (1.5, 1.0, 2.0)) if symmetric else asymmetric_psf(kshape, seed=sum(shape))
    paired, plain = _contexts(shape, psf, None, capi.BOUNDARY_CIRCULAR, monkeypatch)
    assert paired.otf_is_real == symmetric
    g = torch.Generator().manual_seed(sum(shape) + int(symmetric))
    a = (torch.rand(shape, generator=g) + 0.5).to(dev)
    b = (torch.rand(shape, generator=g) + 0.5).to(dev)
    fwd_p, adj_p = _conv_pair(paired, a, b)
    fwd_q, adj_q = _conv_pair(plain, a, b)
    assert_close(fwd_p, fwd_q.astype(np.float64), rel=2e-5, rel_l2=2e-6, pt_rel=2e-5)
    assert_close(adj_p, adj_q.astype(np.float64), rel=2e-5, rel_l2=2e-6, pt_rel=2e-5)
    # decon.m:162-172 in float64: real(ifftn(fftn(x) .* otf)) and its conjugate, the OTF placed as the reference places it
    otf = R.otf_from_psf(psf, shape)
    want_f = np.real(np.fft.ifftn(np.fft.fftn(a.cpu().numpy().astype(np.float64)) * otf))
    want_a = np.real(np.fft.ifftn(np.fft.fftn(b.cpu().numpy().astype(np.float64)) * np.conj(otf)))
    assert_close(fwd_p, want_f, rel=2e-5, rel_l2=2e-6, pt_rel=2e-5)
    assert_close(adj_p, want_a, rel=2e-5, rel_l2=2e-6, pt_rel=2e-5)


@pytest.mark.parametrize("shape", [(64, 32, 32), (128, 96, 16), (64, 1024, 16), (1024, 16, 32), (576, 32, 16), (384, 32, 32)])
def test_paired_layout_fused_iterations_match_the_oracle(dev, shape, monkeypatch):
    from ipp_amd import capi
    psf = asymmetric_psf((5, 7, 5), seed=11)
    paired, plain = _contexts(shape, psf, None, capi.BOUNDARY_CIRCULAR, monkeypatch)
    vol = R.bead_volume(shape, seed=5, psf=R.gaussian_psf((5, 7, 5), (1.0, 1.5, 1.0)))
    want = R.decon_fft(vol, psf, shape, 5, skip_edgetaper=True)
    a, b = torch.from_numpy(vol).to(dev), torch.from_numpy(vol).to(dev)
    paired.iterate(a, None, 5)
    plain.iterate(b, None, 5)
    assert_close(a.cpu().numpy(), want)
    assert_close(a.cpu().numpy(), b.cpu().numpy().astype(np.float64), rel=2e-5, rel_l2=5e-6, pt_rel=5e-5)


@pytest.mark.parametrize("vshape,kshape", [((50, 40, 60), (15, 9, 5)), ((100, 20, 30), (29, 5, 3)), ((200, 30, 16), (57, 3, 3))])
@pytest.mark.parametrize("boundary", [0, 1])
def test_paired_layout_on_padded_grids(dev, vshape, kshape, boundary, monkeypatch):
    """Zero / replicate rule: the transform grid (next supported extents of n + k - 1) has all-zero input planes and cropped
    output planes, which the paired passes skip like the plain ones."""
    from ipp_amd import capi, decon
    psf = asymmetric_psf(kshape, seed=3)
    inv = np.ascontiguousarray(psf[::-1, ::-1, ::-1])
    monkeypatch.setenv("MI_FFT_NATIVE_INFLATE", "100")  # (the cost model would hand grids this much larger than the volume to rocFFT)
    paired, plain = _contexts(vshape, psf, inv, boundary, monkeypatch)
    g = torch.Generator().manual_seed(7)
    a = (torch.rand(vshape, generator=g) + 0.5).to(dev)
    b = (torch.rand(vshape, generator=g) + 0.5).to(dev)
    fwd_p, adj_p = _conv_pair(paired, a, b)
    fwd_q, adj_q = _conv_pair(plain, a, b)
    assert_close(fwd_p, fwd_q.astype(np.float64), rel=2e-5, rel_l2=2e-6, pt_rel=2e-5)
    assert_close(adj_p, adj_q.astype(np.float64), rel=2e-5, rel_l2=2e-6, pt_rel=2e-5)
    direct = decon.RLContext(vshape, psf, inv, boundary=boundary, engine=capi.ENGINE_DIRECT, device=dev)
    fwd_d, adj_d = _conv_pair(direct, a, b)
    assert_close(fwd_p, fwd_d.astype(np.float64), rel=2e-5, rel_l2=2e-6, pt_rel=2e-5)
    assert_close(adj_p, adj_d.astype(np.float64), rel=2e-5, rel_l2=2e-6, pt_rel=2e-5)
    if boundary == 0:
        assert_close(fwd_p, R.convn_same(a.cpu().numpy(), psf).astype(np.float64), rel=2e-5, rel_l2=2e-6, pt_rel=2e-5)
