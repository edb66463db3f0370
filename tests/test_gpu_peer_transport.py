"""GPU: the copy-engine transport of the slab driver (slab.PeerLink on slab.HipPeer = one mi_peer_link of the C ABI: HIP IPC
memory handles, hipMemcpyPeerAsync on a stream of its own, sequence numbers in flag words of exported fine-grained memory) with TWO
processes that share the one GPU of the test box -- a rehearsal of the one-process-per-GPU layout: the handles cross a real process
boundary, the copies land in the other process' buffers, the launch streams wait for words the other process writes.  Every wait
carries its own timeout (MI_PEER_TIMEOUT_S, 30 s here) and the workers are ended after 120 s whatever happens.  (Over xGMI the same
calls address another device; that needs the 8-GPU node.)"""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import rl_oracle as R
from tests.rl_util import assert_close

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, flavour, engine, vol, psf, niter, out, zchunks=1):
    import torch.distributed as dist
    from ipp_amd import slab
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MI_FFT_NATIVE_INFLATE"] = "100"
    os.environ["MI_PEER_TIMEOUT_S"] = "30"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        drv = slab.SlabRL(vol.shape, psf, rank=rank, world_size=world, device=dev, flavour=flavour, engine=engine, volume=vol,
                          transport="peer", zchunks=zchunks)
        assert (drv.zb is not None) == (zchunks > 1 and drv.overlap)
        drv.run(niter)
        assert drv.link is not None and drv.link.n > drv.link.SETS
        # waits that have nothing new behind them -- the round-4 probe hung on 200 such waits on an interprocess event
        # (profiles/r04_slab_host_cost.txt); a wait of this link names a value: the same comparison however often it is made
        for _ in range(100):
            for d in (0, 1):
                if drv.link.src_mask >> d & 1:
                    drv.link.be.recv(drv.link.n, drv.link.C - 1, drv.link.C, d)
        torch.cuda.synchronize(dev)
        drv.link.check()
        mine = drv.interior().cpu().contiguous()
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        info = (drv.sharded, drv.overlap, drv.link.n)
        drv.close()
        if rank == 0:
            out.put((torch.cat(parts, dim=1).numpy(), info))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("flavour,engine,zchunks", [("fft", 2, 1), ("spatial", 2, 1), ("fft", 1, 1), ("fft", 2, 4)],
                         ids=["fused_overlapped_ring", "fused_zero_edges", "real_halos_direct_engine", "fused_ring_z_chunked"])
def test_two_processes_one_gpu_peer_copy_transport(dev, flavour, engine, zchunks):
    psf = R.gaussian_psf((5, 7, 5), (1.0, 1.5, 1.0))
    vol = R.bead_volume((16, 128, 32), seed=33, psf=psf)
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29800 + (os.getpid() % 150) + (0 if flavour == "fft" else 1) + 2 * engine + 5 * zchunks
    procs = [ctx.Process(target=_worker, args=(r, 2, port, flavour, engine, vol, psf, 4, out, zchunks)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        got, (sharded, overlap, n_exchanges) = out.get(timeout=120)
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    want = (R.decon_fft(vol, psf, vol.shape, 4, skip_edgetaper=True) if flavour == "fft"
            else R.decon_spatial(vol, psf, 4, skip_edgetaper=True))
    assert_close(got, want)
    assert sharded == (engine == 2)
    assert n_exchanges == (2 * 4 + 1 if sharded else 2 * 4)
    if engine == 2 and flavour == "fft" and not os.environ.get("MI_FFT_NO_PIPE"):
        assert overlap                                  # edge tiles first, copies issued, remaining tiles, then the wait


@pytest.mark.parametrize("world,zchunks", [(4, 1), (3, 2)], ids=["four_ranks", "three_ranks_z_chunked"])
def test_more_processes_one_gpu_peer_copy_ring(dev, world, zchunks):
    """The ring with inner ranks (two neighbours each, both directions in flight) and an odd rank count: `world` processes on the one
    GPU of the test box (at most 6 may use it), equal slabs of 32 / 42 rows, the fused overlapped iteration, result against the
    whole-volume oracle."""
    psf = R.gaussian_psf((5, 7, 5), (1.0, 1.5, 1.0))
    vol = R.bead_volume((16, 128 if world == 4 else 126, 32), seed=35, psf=psf)
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29960 + (os.getpid() % 30) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, "fft", 2, vol, psf, 3, out, zchunks)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got, (sharded, overlap, n_exchanges) = out.get(timeout=180)
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    assert_close(got, R.decon_fft(vol, psf, vol.shape, 3, skip_edgetaper=True))
    assert sharded and n_exchanges == 2 * 3 + 1


def test_a_wait_nobody_answers_ends_by_its_timeout(dev, monkeypatch):
    """A neighbour that never delivers (a rank that died) leaves an error behind, not a hung device: the one-lane wait kernel gives up
    after MI_PEER_TIMEOUT_S, counts itself in the link's status word, and the stream goes on."""
    import ctypes as C
    import time
    from ipp_amd import capi
    monkeypatch.setenv("MI_PEER_TIMEOUT_S", "1")
    L = capi.lib()
    link, hp, hf = C.c_void_p(), C.create_string_buffer(capi.IPC_HANDLE_BYTES), C.create_string_buffer(capi.IPC_HANDLE_BYTES)
    capi.check(L.mi_peer_link_create(dev.index, 4096, C.byref(link), hp, hf))
    try:
        assert any(hp.raw) and any(hf.raw)
        t = C.c_int(-1)
        capi.check(L.mi_peer_link_status(link, C.byref(t)))
        assert t.value == 0
        slot = C.c_void_p()
        s = C.c_void_p(capi.current_stream_ptr(dev))
        t0 = time.perf_counter()
        capi.check(L.mi_peer_link_recv(link, s, 1, 0, 1, 0, C.byref(slot)))      # exchange 1 of a slot nobody fills
        after = torch.ones(8, device=dev) * 2                                      # work behind the wait on the same stream
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        assert 0.9 < dt < 10.0 and slot.value and float(after.sum()) == 16.0
        capi.check(L.mi_peer_link_status(link, C.byref(t)))
        assert t.value == 1
        # invalid arguments are refused before anything is enqueued
        assert L.mi_peer_link_recv(link, s, 0, 0, 1, 0, C.byref(slot)) == capi.MI_ERR_INVALID
        assert L.mi_peer_link_send(link, s, 1, 2, 2, 0, 16, None, None) == capi.MI_ERR_INVALID
        assert L.mi_peer_link_send(link, s, 1, 0, 1, 4000, 200, None, None) == capi.MI_ERR_INVALID   # beyond the slot
    finally:
        L.mi_peer_link_destroy(link)
