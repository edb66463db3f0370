"""Is a slab rank host-bound?    The HOST cost of enqueueing one sharded RL iteration -- the Python step
sequence of slab.SlabRL.iterate: edge tiles, pack, send, remaining tiles, wait, unpack, y, z, y, twice per iteration -- against the
DEVICE time of the same iteration, on the rank-local shapes of the 8-GPU runs:

    C3 at N = 8   global 2048 x 2048 x 512, 31 x 31 x 61 PSF  -> rank 2048 x 288 x 512   (256 rows + 2 x 15 halo rows -> 288)
    C4 at N = 8   global 4096 x 4096 x 1024, 63 x 63 x 127 PSF -> rank 4096 x 576 x 1024 (512 rows + 2 x 31 -> 576)

One GPU suffices: a self-ring (one process whose slab is its own neighbour: every kernel and every pack / unpack of a rank, no
transport) and two processes on one GPU with the copy-engine transport (slab.PeerLink on one mi_peer_link: IPC memory handles, peer
copies, sequence numbers in flag words; round 4: interprocess events and host sequence numbers).  "enqueue" = wall time of K iterations
issued back to back WITHOUT a synchronisation, per iteration (the launch queues take them all; since round 5 the issue loop never waits
for the neighbour on the host); "CPU" = the thread's CPU time over the same loop; "device" = the same K iterations timed to their end.
A rank is host-bound when its host work approaches its device time.

    python profiles/slab_host_cost.py > gpurun_out/r05_slab_host_cost.txt
"""
import os
import sys
import time

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

RANKS = {  # name: (slab of ONE rank as a global shape (z, y, x) for a ring of `world` such slabs, psf (z, y, x))
    "C3 at N=8": ((512, 256, 2048), (61, 31, 31)),
    "C4 at N=8": ((1024, 512, 4096), (127, 63, 63)),
}
K = 12


def measure(drv, dev):
    for _ in range(3):
        drv.iterate()
    torch.cuda.synchronize(dev)
    t0, c0 = time.perf_counter(), time.thread_time()
    for _ in range(K):
        drv.iterate()
    t_enq = (time.perf_counter() - t0) / K * 1e3
    t_cpu = (time.thread_time() - c0) / K * 1e3   # CPU time of this thread: what the host WORKS per iteration (waits for a neighbour sleep)
    torch.cuda.synchronize(dev)
    t_all = (time.perf_counter() - t0) / K * 1e3
    # device time alone: the queue is full when the clock starts only if the host is ahead; time K more with events
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(K):
        drv.iterate()
    ev1.record()
    torch.cuda.synchronize(dev)
    t_dev = ev0.elapsed_time(ev1) / K
    # the same issue sequence with an idle device in front of it (one iteration, then a synchronisation): what the host needs per
    # iteration when nothing pushes back -- K iterations ahead of the device the runtime makes the caller wait inside
    # hipMemcpyPeerAsync for its in-flight copies (1 ms per call, profiles/r05_slab_host_cost.txt), which is the device's pace, not host work
    t_idle = 0.0
    for _ in range(K):
        t0 = time.perf_counter()
        drv.iterate()
        t_idle += time.perf_counter() - t0
        torch.cuda.synchronize(dev)
    return t_enq, t_all, t_dev, t_cpu, t_idle / K * 1e3


def self_ring(name, zchunks):
    import bench
    from ipp_amd import slab
    dev = torch.device("cuda", 0)
    shape, kshape = RANKS[name]
    psf = bench.make_psf(kshape)
    drv = slab.SlabRL(shape, psf, rank=0, world_size=1, device=dev, flavour="fft", engine=2, seed=1, zchunks=zchunks)
    t_enq, t_all, t_dev, t_cpu, t_idle = measure(drv, dev)
    print(f"{name}  self-ring        zchunks {zchunks}: local {drv.lshape[2]} x {drv.lshape[1]} x {drv.lshape[0]}, fused {drv.sharded}, "
          f"split x pass {drv.overlap}:  enqueue {t_enq:6.3f} ms / iteration (CPU {t_cpu:6.3f} ms), device {t_dev:6.3f} ms, wall {t_all:6.3f} ms "
          f"-> host share {t_enq / t_dev:.2f}; issued to an idle device {t_idle:6.3f} ms", flush=True)
    drv.close()
    del drv
    torch.cuda.empty_cache()


def _peer_worker(rank, world, port, name, zchunks, out):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    from ipp_amd import slab
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        shape, kshape = RANKS[name]
        gshape = (shape[0], shape[1] * world, shape[2])  # two slabs of the rank's size
        psf = bench.make_psf(kshape)
        drv = slab.SlabRL(gshape, psf, rank=rank, world_size=world, device=dev, flavour="fft", engine=2, seed=1, transport="peer",
                          zchunks=zchunks)
        # host time inside each call of the link's backend (one C call each), summed over the measured iterations
        acc = {}
        be = drv.link.be if drv.link is not None else None
        if be is None:   # (the link is made by the first exchange)
            drv.iterate()
            be = drv.link.be

        def timed(name):
            fn = getattr(be, name)

            def wrapper(*a, **kw):
                t0 = time.perf_counter()
                r = fn(*a, **kw)
                c = acc.setdefault(name, [0, 0.0])
                c[0] += 1
                c[1] += time.perf_counter() - t0
                return r
            setattr(be, name, wrapper)
        for nm in ("begin", "send", "recv", "exchange"):
            timed(nm)
        res = measure(drv, dev)
        if rank == 0:
            iters = 3 + 2 * K
            print("    host time inside the link's C calls, rank 0: " + ", ".join(
                f"{nm} {c[0] / iters:.1f} calls / iteration x {c[1] / c[0] * 1e6:.0f} us" for nm, c in sorted(acc.items())), flush=True)
        dist.barrier()
        drv.close()
        if rank == 0:
            out.put((res, tuple(drv.lshape)))
    except Exception as e:   # (the parent must hear about it: the other rank would wait for this one until its time-out)
        import traceback
        out.put(("error", f"rank {rank}: {e!r}\n{traceback.format_exc()}"))
        os._exit(1)
    finally:
        dist.destroy_process_group()


def two_processes(name, zchunks):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29650 + (os.getpid() % 100) + zchunks
    procs = [ctx.Process(target=_peer_worker, args=(r, 2, port, name, zchunks, out)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        got = out.get(timeout=240)
        if got[0] == "error":
            raise RuntimeError(got[1])
        (t_enq, t_all, t_dev, t_cpu, t_idle), lshape = got
    finally:
        for p in procs:
            p.join(timeout=20)
            if p.is_alive():
                p.kill()
    print(f"{name}  2 procs, 1 GPU   zchunks {zchunks}: local {lshape[2]} x {lshape[1]} x {lshape[0]}, copy-engine transport:  issue loop "
          f"{t_enq:6.3f} ms / iteration per rank, of which CPU "
          f"{t_cpu:6.3f} ms; device (BOTH ranks) {t_dev:6.3f} ms, wall {t_all:6.3f} ms "
          f"-> host work / ONE rank's device time {t_cpu / (t_dev / 2):.2f}; issued to an idle device {t_idle:6.3f} ms / iteration "
          f"-> {t_idle / (t_dev / 2):.2f} of one rank's device time", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or list(RANKS)
    for name in which:
        for zc in (1, 4):
            self_ring(name, zc)
    for zc in (1, 4):
        two_processes("C3 at N=8", zc)
