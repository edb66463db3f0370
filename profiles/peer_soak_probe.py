"""Soak of the copy-engine link (slab.PeerLink on slab.HipPeer = mi_peer_link): W processes on the one GPU of the box in a closed
ring, N exchanges of a payload whose every float names (exchange, sender, edge); every arrival is compared on the device, mismatches
are counted without a host round trip per exchange, the link's timeout counter is read at the end.
    python profiles/peer_soak_probe.py [W=2] [N=20000] [floats=262144]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.multiprocessing as mp


class Ring:
    """What PeerLink needs of its driver: rank, world, neighbours (a closed ring: every rank has both)."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def neighbours(self):
        return (self.rank - 1) % self.world, (self.rank + 1) % self.world


def worker(rank, world, port, n_ex, nfloats, out):
    import torch.distributed as dist
    from ipp_amd import capi, slab
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MI_PEER_TIMEOUT_S"] = "20"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    link = slab.PeerLink(Ring(rank, world), nfloats, slab.HipPeer(dev))
    lo, hi = Ring(rank, world).neighbours()
    got = torch.empty(nfloats, dtype=torch.float32, device=dev)
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    L = capi.lib()

    def one(n):
        st = link.begin()
        for d in (0, 1):
            st[d].fill_(float((n % 4096) * 16 + rank * 2 + d))
        link.exchange()
        for d, src in ((0, lo), (1, hi)):       # slot 0: the previous rank's edge 0; slot 1: the next rank's edge 1
            p = link.receive(d)
            capi.check(L.mi_unpack_rows(dev.index, capi.current_stream_ptr(dev), p, nfloats, 1, 1, 0, 1, got.data_ptr()))
            bad.add_((got != float((n % 4096) * 16 + src * 2 + d)).sum())

    for n in range(1, 201):
        one(n)
    torch.cuda.synchronize(dev)
    dist.barrier()
    t0 = time.perf_counter()
    for n in range(201, 201 + n_ex):
        one(n)
        if n % 2000 == 0:
            torch.cuda.synchronize(dev)      # (bounds the run-ahead of the host; also a progress line for the log)
            if rank == 0:
                print(f"  exchange {n}: {(time.perf_counter() - t0) / (n - 200) * 1e6:.1f} us each so far", flush=True)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    link.check()
    res = (rank, int(bad.item()), link.be.timed_out(), dt / n_ex * 1e6)
    link.close()
    allr = [None] * world
    dist.all_gather_object(allr, res)
    if rank == 0:
        out.put(allr)
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n_ex = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    nfloats = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29700 + os.getpid() % 200
    procs = [ctx.Process(target=worker, args=(r, world, port, n_ex, nfloats, out)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = out.get(timeout=900)
        print(f"{world} processes on one GPU, closed ring, {n_ex} exchanges of {nfloats * 4 / 1e6:.2f} MB per edge (both edges of every rank), every float compared:")
        for rank, bad, tmo, us in res:
            print(f"  rank {rank}: {bad} mismatching floats, {tmo} waits ended by the timeout, {us:.1f} us per exchange")
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    sys.exit(0 if all(p.exitcode == 0 for p in procs) else 1)
