"""Multi-GPU Richardson-Lucy: one volume cut into Y-slabs, one rank per GPU, halo exchange over RCCL.

The reference shards a volume into independent padded blocks (split_stack.m:9-26, LsDeconv.m:341,401-403,
647-654) that never talk to each other.  Here the blocks of one row of that grid stay coupled: every rank owns
``ny / world_size`` rows of the global volume plus ``h`` halo rows per side, and before each of the two
convolutions of an iteration the halo rows are refreshed from the neighbouring ranks (``ncclSend/ncclRecv``
through ``torch.distributed.batch_isend_irecv``; backend "nccl" is RCCL on ROCm).  The result equals the
single-GPU result on the whole volume up to fp32 rounding:

  flavour "fft"      deconFFT semantics (decon.m:127-204): circular on the GLOBAL shape, so the slab chain is a
                     ring (last rank <-> first rank); the PSF placement (incl. the even-shape one-voxel offset) is
                     taken from the global shape and handed to the local context (``mi_rl_create_ex``).
  flavour "spatial"  deconSpatial semantics (decon.m:26-124): zeros outside the GLOBAL volume, so the outer
                     halos of the first / last rank are zero-filled instead of exchanged.

When the local context runs the fused native FFT pipeline (``ctx.fuses``), the halo exchange moves from real space to the
pipeline's x-transformed buffer: the input of every convolution is kept there (the ratio never reaches HBM, ``bl`` is read
twice and written once per iteration, exactly like the single-GPU ``mi_rl_iterate``), the x transform of a row does not
depend on other rows, so each rank sends the x-transformed rows of its interior edges and receives its halo rows in that form
(same bytes as real rows).  The halo rows of ``bl`` are then never read again and hold meaningless values.

Y is sharded rather than Z because the halo is PSF-extent/2 rows of a 2048 x 512 plane instead of 30 of 64
planes ("halo inflation": 1.12x instead of 1.94x on config C3).  No collective is on the
data path; only the optional stop criterion all-reduces one double.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import capi
from .capi import BOUNDARY_CIRCULAR, BOUNDARY_ZERO, ENGINE_AUTO


def slab_rows(ny: int, world_size: int):
    """[y0, y1) per rank: ceil(ny / W) rows each, the last rank takes the remainder (split_stack.m:17-19 rule:
    ``min(ys + block.y - 1, stack_info.y)``)."""
    per = int(math.ceil(ny / world_size))
    return [(min(r * per, ny), min((r + 1) * per, ny)) for r in range(world_size)]


def psf_shift(n: int, k: int, flavour: str) -> int:
    """Placement of PSF sample j at offset (j - shift): deconFFT = ifftshift(zero-pad-centre) on the global
    extent (decon.m:131-133, 323-344); spatial = centred (convn 'same')."""
    if flavour == "fft":
        return n // 2 - (n - k) // 2
    return k - 1 - (k - 1 - k // 2)


class HipOps:
    """Device operations of the slab driver, all through the C ABI (no CPU fallback)."""

    def __init__(self, device):
        capi.require_gpu()
        self.device = torch.device(device)

    def make_ctx(self, local_shape_zyx, psf, boundary_xyz, shift_xyz, engine, psf_inv=None):
        from .decon import RLContext
        return RLContext(local_shape_zyx, psf, psf_inv, boundary=boundary_xyz, engine=engine, device=self.device,
                         shift_xyz=shift_xyz)

    def pack(self, vol, y0, rows):
        nz, ny, nx = vol.shape
        out = torch.empty((nz, rows, nx), dtype=vol.dtype, device=vol.device)
        capi.check(capi.lib().mi_pack_rows(vol.device.index, capi.current_stream_ptr(vol.device), vol.data_ptr(), nx, ny, nz,
                                           y0, rows, out.data_ptr()))
        return out

    def unpack(self, packed, vol, y0):
        nz, ny, nx = vol.shape
        capi.check(capi.lib().mi_unpack_rows(vol.device.index, capi.current_stream_ptr(vol.device), packed.data_ptr(), nx, ny,
                                             nz, y0, packed.shape[1], vol.data_ptr()))

    def zero_rows(self, vol, y0, rows):
        vol[:, y0:y0 + rows, :].zero_()

    # the copy-engine transport packs into its staging buffers and unpacks from raw receive pointers
    def peer_backend(self):
        return HipPeer(self.device)

    def halo_floats(self, ctx, lshape, h):
        # h rows of an X x Z plane as real rows; as x-transformed rows of the context's (possibly zero-padded) transform grid
        return max(h * lshape[0] * lshape[2], h * ctx.spectrum_row_floats if getattr(ctx, "fuses", 0) else 0)

    def pack_spec_into(self, ctx, y0, rows, out):
        ctx.spectrum_pack(y0, rows, out=out[:rows * ctx.spectrum_row_floats])

    def unpack_spec_ptr(self, ctx, ptr, y0, rows):
        ctx.spectrum_unpack_ptr(ptr, y0, rows)

    # z-chunked exchange of the fused pipeline: a chunk of planes keeps its place in the packed buffer
    def spec_planes(self, ctx, lshape):
        return int(lshape[0])

    def spec_granule(self, ctx):
        return int(ctx.z_granule)

    def spec_chunk_span(self, ctx, h, z0, z1, nz):
        per_plane = h * ctx.spectrum_row_floats // nz
        return z0 * per_plane, (z1 - z0) * per_plane

    def pack_spec_chunk(self, ctx, y0, rows, z0, z1, out_full):
        ctx.spectrum_rows_z(out_full, y0, rows, z0, z1, 0)

    def unpack_spec_chunk(self, ctx, src_full, y0, rows, z0, z1):
        """``src_full``: the whole packed buffer (tensor or raw device pointer), or None for zero rows"""
        ctx.spectrum_rows_z(src_full, y0, rows, z0, z1, 2 if src_full is None else 1)

    def stage(self, ctx, bl, update, stage, z0=0, z1=0, edge_rows=None):
        ctx.sharded_stage(bl, update, stage, z0, z1, edge_rows)

    def pack_into(self, vol, y0, rows, out):
        nz, ny, nx = vol.shape
        assert out.numel() >= nz * rows * nx
        capi.check(capi.lib().mi_pack_rows(vol.device.index, capi.current_stream_ptr(vol.device), vol.data_ptr(), nx, ny, nz,
                                           y0, rows, out.data_ptr()))

    def unpack_ptr(self, ptr, vol, y0, rows):
        nz, ny, nx = vol.shape
        capi.check(capi.lib().mi_unpack_rows(vol.device.index, capi.current_stream_ptr(vol.device), int(ptr), nx, ny, nz, y0, rows,
                                             vol.data_ptr()))

    # x-transformed rows of the fused pipeline (ctx.fuses)
    def pack_spec(self, ctx, y0, rows):
        return ctx.spectrum_pack(y0, rows)

    def unpack_spec(self, ctx, packed, y0, rows):
        ctx.spectrum_unpack(packed, y0, rows)


class HipPeer:
    """Device side of the copy-engine transport: one ``mi_peer_link`` of the C ABI (include/mi_lsdeconv.h, "copy-engine transport")
    -- receive slots and a page of flag words exported through HIP IPC memory handles, a copy stream, hipMemcpyPeerAsync, one-lane
    kernels that write / wait for sequence numbers.  One C call per step of an exchange."""

    def __init__(self, device):
        capi.require_gpu()
        self.dev = torch.device(device)
        self.di = self.dev.index
        self.L = capi.lib()
        self.link = C.c_void_p()

    def identity(self):
        """What names this device in EVERY process of the job: its UUID (a rank that masks its devices sees its GPU as ordinal 0)."""
        pr = torch.cuda.get_device_properties(self.dev)
        u = getattr(pr, "uuid", None)
        if u is not None:
            return ("uuid", str(u))
        bus = [getattr(pr, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
        return ("pci", tuple(bus)) if all(b is not None for b in bus) else ("ordinal", int(self.di))

    def resolve(self, ident):
        """The ordinal under which THIS process sees the device `ident` names; -1 when it does not (the runtime then resolves the
        mapped pointer instead of a named device)."""
        kind, val = ident
        if kind == "ordinal":
            return int(val)
        for i in range(torch.cuda.device_count()):
            pr = torch.cuda.get_device_properties(i)
            if kind == "uuid" and str(getattr(pr, "uuid", None)) == val:
                return i
            if kind == "pci" and tuple(getattr(pr, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")) == tuple(val):
                return i
        return -1

    def _s(self):
        return C.c_void_p(capi.current_stream_ptr(self.dev))

    def create(self, nbytes):
        """-> the handles (payload, flag page) the neighbours connect with"""
        hp, hf = C.create_string_buffer(capi.IPC_HANDLE_BYTES), C.create_string_buffer(capi.IPC_HANDLE_BYTES)
        capi.check(self.L.mi_peer_link_create(self.di, int(nbytes), C.byref(self.link), hp, hf))
        return hp.raw, hf.raw

    def connect(self, d, handles, ident):
        capi.check(self.L.mi_peer_link_connect(self.link, int(d), handles[0], handles[1], self.resolve(ident)))

    def staging(self, nfloats):
        return torch.empty(int(nfloats), dtype=torch.float32, device=self.dev)

    def begin(self, n, src_mask):
        capi.check(self.L.mi_peer_link_begin(self.link, self._s(), int(n), int(src_mask)))

    def send(self, n, k, chunks, first_byte, nbytes, up, dn):
        capi.check(self.L.mi_peer_link_send(self.link, self._s(), int(n), int(k), int(chunks), int(first_byte), int(nbytes),
                                            up.data_ptr() if up is not None else None, dn.data_ptr() if dn is not None else None))

    def recv(self, n, k, chunks, d):
        q = C.c_void_p()
        capi.check(self.L.mi_peer_link_recv(self.link, self._s(), int(n), int(k), int(chunks), int(d), C.byref(q)))
        return int(q.value)

    def exchange(self, n, src_mask, up, dn):
        """One exchange of one chunk in ONE call: both sends, then the waits.  -> (slot 0 or None, slot 1 or None)"""
        lo, hi = C.c_void_p(), C.c_void_p()
        capi.check(self.L.mi_peer_exchange(self.link, self._s(), int(n), int(src_mask), up.data_ptr() if up is not None else None,
                                           dn.data_ptr() if dn is not None else None, C.byref(lo), C.byref(hi)))
        return (int(lo.value) if lo.value else None), (int(hi.value) if hi.value else None)

    def timed_out(self):
        v = C.c_int()
        capi.check(self.L.mi_peer_link_status(self.link, C.byref(v)))
        return int(v.value)

    def disconnect(self):
        if self.link:
            self.L.mi_peer_link_disconnect(self.link)

    def destroy(self):
        if self.link:
            self.L.mi_peer_link_destroy(self.link)
            self.link = C.c_void_p()


class PeerLink:
    """Copy-engine transport of the halo exchange ("peer").  Every rank exports its receive slots (two directions x two sets that
    alternate between consecutive exchanges) and a page of flag words; a sender copies its packed rows straight into the
    neighbour's slot on a stream of its own -- SDMA engines, so the persistent x pass that runs meanwhile keeps every compute
    unit -- and then writes the sequence number of what it has delivered into the neighbour's arrival word; the receiver's launch
    stream waits until that word has reached the number it needs.  The receiver acknowledges exchange n when exchange n + 1 begins
    (a word of the sender's page), and the sender's copy stream waits for that before it overwrites the set at exchange n + 2.
    Nothing is handed over on the host after the handles have travelled once (``all_gather_object``); a wait names a VALUE, so a
    repeated or early wait is harmless (round 4 used interprocess events and host sequence numbers: csrc/peer.hip).

    Directed edges of rank r: d = 0 "up" (its last interior rows -> slot 0 of the next rank), d = 1 "down" (its first interior
    rows -> slot 1 of the previous rank).  ``backend``: HipPeer, or a host double with the same methods (tests/slab_util.py) so
    that the protocol runs on CPU ranks."""

    SETS = 2

    def __init__(self, drv, nfloats, backend, group=None, chunks=1):
        import torch.distributed as dist
        self.drv, self.group, self.n = drv, group, 0
        self.be = backend
        self.nfloats = int(nfloats)
        self.nbytes = 4 * self.nfloats
        self.C = max(1, int(chunks))   # an exchange may travel in C chunks (z chunks of the halo rows)
        handles = backend.create(self.nbytes)
        self.staging = [[backend.staging(self.nfloats) for _ in range(self.SETS)] for _ in range(2)]
        mine = {"handles": handles, "dev": backend.identity()}
        everyone = [None] * drv.world
        dist.all_gather_object(everyone, mine, group=group)
        lo, hi = drv.neighbours()
        self.has_dst = {0: hi is not None, 1: lo is not None}     # edge d has a receiver
        self.src_mask = (1 if lo is not None else 0) | (2 if hi is not None else 0)   # slot d has a sender
        for d, dst_rank in ((0, hi), (1, lo)):
            if dst_rank is not None:
                backend.connect(d, everyone[dst_rank]["handles"], everyone[dst_rank]["dev"])
        self._slots = (None, None)
        dist.barrier(group=group)

    def begin(self):
        """Next exchange: the staging buffers (edge 0, edge 1) to pack the rows into, on the launch stream."""
        self.n += 1
        st = self.n % self.SETS
        self.be.begin(self.n, self.src_mask)
        self._slots = (None, None)
        return [self.staging[d][st] if self.has_dst[d] else None for d in (0, 1)]

    def send(self, k=0, first_float=0, nfloats=None):
        """Chunk k of the rows is packed (launch stream): copy it into the neighbours' slots on the copy stream.  The chunk is the
        float range [first_float, first_float + nfloats) of the staging buffer and lands at the same place of the receive slot."""
        st = self.n % self.SETS
        nfloats = self.nfloats - first_float if nfloats is None else int(nfloats)
        self.be.send(self.n, k, self.C, 4 * first_float, 4 * nfloats, self.staging[0][st] if self.has_dst[0] else None,
                     self.staging[1][st] if self.has_dst[1] else None)

    def exchange(self):
        """send() of the whole buffer and receive() of both slots in one call of the backend (C == 1)."""
        st = self.n % self.SETS
        self._slots = self.be.exchange(self.n, self.src_mask, self.staging[0][st] if self.has_dst[0] else None,
                                       self.staging[1][st] if self.has_dst[1] else None)

    def receive(self, d, k=0):
        """The rows that arrived in slot d (None: global edge) as the backend's pointer TO THE WHOLE SLOT; the launch stream waits
        for chunk k of them."""
        if not (self.src_mask >> d & 1):
            return None
        if self.C == 1 and self._slots[d] is not None:   # exchange() has enqueued the wait already
            return self._slots[d]
        return self.be.recv(self.n, k, self.C, d)

    def release(self):
        """(kept for the callers' sequence: the acknowledgement of an exchange is written when the next one begins)"""

    def check(self):
        """Raises when a wait of this link ended by its timeout (synchronises with the device)."""
        t = self.be.timed_out()
        if t:
            raise TimeoutError(f"halo exchange: {t} wait(s) of rank {self.drv.rank} for a neighbour ended by the timeout (MI_PEER_TIMEOUT_S)")

    def close(self):
        import sys
        try:
            t = self.be.timed_out()
            self.be.disconnect()
            # nobody frees what it exported before every neighbour has unmapped it
            try:
                import torch.distributed as dist
                if dist.is_initialized():
                    dist.barrier(group=self.group)
            except Exception as e:
                sys.stderr.write(f"PeerLink.close: no barrier before the buffers are freed ({e!r})\n")
            self.be.destroy()
            if t:
                sys.stderr.write(f"PeerLink.close: {t} wait(s) for a neighbour had ended by the timeout\n")
        except Exception as e:
            sys.stderr.write(f"PeerLink.close: {e!r}\n")


class SlabRL:
    def __init__(self, global_shape_zyx, psf, rank=0, world_size=1, device=None, flavour="fft", engine=ENGINE_AUTO,
                 volume=None, seed=None, ops=None, group=None, transport="rccl", zchunks=1):
        if flavour not in ("fft", "spatial"):
            raise ValueError("flavour must be 'fft' (deconFFT) or 'spatial' (deconSpatial)")
        if transport not in ("rccl", "peer"):
            raise ValueError("transport must be 'rccl' (grouped send/recv of torch.distributed) or 'peer' (hipMemcpyPeerAsync)")
        self.rank, self.world, self.flavour, self.group = int(rank), int(world_size), flavour, group
        self.transport, self.link = transport, None
        self.gshape = tuple(int(s) for s in global_shape_zyx)
        self.psf = np.ascontiguousarray(psf, dtype=np.float32)
        gz, gy, gx = self.gshape
        kz, ky, kx = self.psf.shape
        self.ops = ops if ops is not None else HipOps(device)
        self.device = self.ops.device
        self.y0, self.y1 = slab_rows(gy, self.world)[self.rank]
        self.n_loc = self.y1 - self.y0
        sy = psf_shift(gy, ky, flavour)
        self.h = max(sy, ky - 1 - sy)  # forward reads [-(k-1-shift), +shift], the adjoint the mirror image
        if self.world > 1 and any(b - a < self.h for a, b in slab_rows(gy, self.world)):
            raise ValueError(f"slab of {self.n_loc} rows is thinner than the halo ({self.h}); use fewer ranks")
        rows = self.n_loc + 2 * self.h
        # local extent the FFT pipeline takes natively (2^a * {1,3,5,9} on this axis); the rows behind the upper halo stay zero
        self.rows = int(capi.lib().mi_fft_good_size(rows, 1)) if ops is None else rows
        bxz = BOUNDARY_CIRCULAR if flavour == "fft" else BOUNDARY_ZERO
        # y is "circular on the local extent": wrap-around only ever reaches halo / padding rows
        boundary_xyz = (bxz, BOUNDARY_CIRCULAR, bxz)
        shift_xyz = (psf_shift(gx, kx, flavour), sy, psf_shift(gz, kz, flavour))
        self.lshape = (gz, self.rows, gx)
        # deconSpatial convolves with psf.inv = the flipped PSF (LsDeconv.m:163, decon.m:64); for even extents that is not the
        # transpose of the forward operator, and the sharded axis only LOOKS circular to the context: hand the kernel over
        inv = None
        if flavour == "spatial" and any(k % 2 == 0 for k in self.psf.shape):
            inv = np.ascontiguousarray(self.psf[::-1, ::-1, ::-1])
        # a rank has its device to itself (one process per GPU): also the smaller spectrum arrays of many ranks are placed by trial
        # (fft_native.hip, NativeFft::init; the library's own limit of 6 GB keeps decwrap's concurrent block plans out)
        # -- for THIS plan only: the variable is read at plan creation and put back at once, later plans of the process (decon blocks,
        # other tests) keep the library's limit
        import os
        prev = os.environ.get("MI_FFT_PLACE_MIN_MB")
        if prev is None:
            os.environ["MI_FFT_PLACE_MIN_MB"] = "1024"
        try:
            self.ctx = self.ops.make_ctx(self.lshape, self.psf, boundary_xyz, shift_xyz, engine, inv)
        finally:
            if prev is None:
                os.environ.pop("MI_FFT_PLACE_MIN_MB", None)
        self.bl = torch.zeros(self.lshape, dtype=torch.float32, device=self.device)
        # fused pipeline: halos travel as x-transformed rows, no ratio volume exists
        self.sharded = bool(getattr(self.ctx, "fuses", 0))
        self._begun = False
        # rows sent to the neighbours: the first and the last h interior rows
        self.edge_rows = (self.h, 2 * self.h, self.n_loc, self.n_loc + self.h)
        self.overlap = (int(getattr(self.ctx, "fuses", 0)) == 2 and self.h > 0 and self.n_loc >= 2 * self.h)
        self.ratio = None if self.sharded else torch.zeros(self.lshape, dtype=torch.float32, device=self.device)
        # z-chunked exchange (fused pipeline with a split x pass): the halo rows travel in `zchunks` chunks of planes; a chunk is
        # sent as soon as the x tiles of its planes have run and the next step's y-forward pass starts on a chunk as soon as ITS
        # halos have landed -- the exchange hides behind almost the whole x pass and the y pass instead of the rest of the x pass
        self.zb, self._pending = None, None
        if int(zchunks) > 1 and self.overlap:
            g = self.ops.spec_granule(self.ctx)
            nzp = self.ops.spec_planes(self.ctx, self.lshape)
            if g > 0 and nzp >= 2 * g:
                c = min(int(zchunks), nzp // g)
                per = -(-nzp // c)
                per = -(-per // g) * g
                self.zb = [(z, min(z + per, nzp)) for z in range(0, nzp, per)]
                self.nzp = nzp
        if volume is not None:
            v = volume[:, self.y0:self.y1, :]
            v = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v
            self.bl[:, self.h:self.h + self.n_loc, :] = v.to(self.device)
        elif seed is not None:
            # this rank's rows of the seeded synthetic volume (background + sparse beads, bench.py recipe)
            g = torch.Generator(device=self.device).manual_seed(int(seed) + self.rank)
            part = torch.empty((gz, self.n_loc, gx), dtype=torch.float32, device=self.device).uniform_(0.01, 0.02, generator=g)
            idx = torch.randint(0, part.numel(), (max(1, part.numel() // 4096),), generator=g, device=self.device)
            amp = torch.empty(idx.numel(), dtype=torch.float32, device=self.device).uniform_(0.2, 1.0, generator=g)
            part.view(-1).index_put_((idx,), amp, accumulate=True)
            self.bl[:, self.h:self.h + self.n_loc, :] = part

    # ------------------------------------------------------------------ halo exchange
    def neighbours(self):
        """(rank whose top rows fill my lower halo, rank whose bottom rows fill my upper halo); None = global edge."""
        W, r, ring = self.world, self.rank, self.flavour == "fft"
        lo = r - 1 if r > 0 else (W - 1 if ring else None)
        hi = r + 1 if r < W - 1 else (0 if ring else None)
        return lo, hi

    def pack_halos(self, vol):
        """(rows for the upper neighbour's lower halo, rows for the lower neighbour's upper halo)."""
        h, n = self.h, self.n_loc
        return self.ops.pack(vol, n, h), self.ops.pack(vol, h, h)  # last h / first h interior rows

    def unpack_halos(self, vol, recv_lo, recv_hi):
        """Writes the received rows into the halos; None = global edge of the spatial flavour -> zeros."""
        h, n = self.h, self.n_loc
        if recv_lo is not None:
            self.ops.unpack(recv_lo, vol, 0)
        else:
            self.ops.zero_rows(vol, 0, h)
        if recv_hi is not None:
            self.ops.unpack(recv_hi, vol, h + n)
        else:
            self.ops.zero_rows(vol, h + n, h)

    def pack_spec_halos(self):
        h, n = self.h, self.n_loc
        return self.ops.pack_spec(self.ctx, n, h), self.ops.pack_spec(self.ctx, h, h)

    def unpack_spec_halos(self, recv_lo, recv_hi):
        h, n = self.h, self.n_loc
        self.ops.unpack_spec(self.ctx, recv_lo, 0, h)          # None = global edge of the spatial flavour -> zero rows
        self.ops.unpack_spec(self.ctx, recv_hi, h + n, h)

    def exchange_start(self, vol=None, finish_now=False):
        """Pack the rows the neighbours need and issue the sends / receives; returns the state ``exchange_finish`` needs.
        ``vol`` None: the rows of the context's x-transformed input buffer (fused pipeline)."""
        if self.h == 0:
            return None
        spec = vol is None
        lo_src, hi_src = self.neighbours()
        if self.transport == "peer" and self.world > 1:
            # copy engines: pack straight into the link's staging buffers, peer copies on the link's stream
            h, n = self.h, self.n_loc
            if self.link is None:
                self.link = PeerLink(self, self.ops.halo_floats(self.ctx, self.lshape, h), self.ops.peer_backend(), self.group)
            up, dn = self.link.begin()
            for buf, y0 in ((up, n), (dn, h)):          # last h / first h interior rows
                if buf is None:
                    continue
                if spec:
                    self.ops.pack_spec_into(self.ctx, y0, h, buf)
                else:
                    self.ops.pack_into(vol, y0, h, buf)
            if finish_now:
                self.link.exchange()                     # nothing runs between start and finish: sends and waits in ONE call of the C ABI
            else:
                self.link.send()                         # (the waits follow the x tiles that run beside the copies: exchange_finish)
            return ("peer", vol)
        send_up, send_dn = self.pack_spec_halos() if spec else self.pack_halos(vol)
        if self.world == 1:  # self-ring: my own rows wrap around
            return (vol, [], send_up if lo_src is not None else None, send_dn if hi_src is not None else None, False)
        import torch.distributed as dist
        # RCCL moves device buffers directly; a gloo group (CPU rehearsals of the multi-rank path) needs host staging
        staged = send_up.is_cuda and dist.get_backend(self.group) == "gloo"
        if staged:
            send_up, send_dn = send_up.cpu(), send_dn.cpu()
        recv_lo = torch.empty_like(send_up) if lo_src is not None else None
        recv_hi = torch.empty_like(send_dn) if hi_src is not None else None
        # message tags / issue order: "up" traffic (my bottom rows -> next rank) first, then "down"; with two ranks
        # on a ring both neighbours are the same peer and RCCL matches sends to receives by issue order
        ops = []
        if hi_src is not None:
            ops.append(dist.P2POp(dist.isend, send_up, hi_src, self.group, 1))
        if lo_src is not None:
            ops.append(dist.P2POp(dist.isend, send_dn, lo_src, self.group, 2))
        if lo_src is not None:
            ops.append(dist.P2POp(dist.irecv, recv_lo, lo_src, self.group, 1))
        if hi_src is not None:
            ops.append(dist.P2POp(dist.irecv, recv_hi, hi_src, self.group, 2))
        reqs = dist.batch_isend_irecv(ops)
        return (vol, reqs, recv_lo, recv_hi, staged, (send_up, send_dn))  # the send buffers stay alive until the wait

    def exchange_finish(self, state):
        if state is None:
            return
        if state[0] == "peer":
            vol, h, n = state[1], self.h, self.n_loc
            for d, y0 in ((0, 0), (1, h + n)):
                ptr = self.link.receive(d)               # the launch stream waits for the neighbour's copy
                if vol is None:
                    if ptr is None:
                        self.ops.unpack_spec(self.ctx, None, y0, h)   # global edge of the spatial flavour: zero rows
                    else:
                        self.ops.unpack_spec_ptr(self.ctx, ptr, y0, h)
                elif ptr is None:
                    self.ops.zero_rows(vol, y0, h)
                else:
                    self.ops.unpack_ptr(ptr, vol, y0, h)
            self.link.release()
            return
        vol, reqs, recv_lo, recv_hi, staged = state[:5]
        for req in reqs:
            req.wait()  # RCCL: makes the current stream wait for the transfer; gloo: blocks the host
        if staged:
            recv_lo = recv_lo.to(self.device) if recv_lo is not None else None
            recv_hi = recv_hi.to(self.device) if recv_hi is not None else None
        if vol is None:
            self.unpack_spec_halos(recv_lo, recv_hi)
        else:
            self.unpack_halos(vol, recv_lo, recv_hi)

    def exchange(self, vol=None):
        """Refresh the 2*h halo rows from the neighbouring slabs (ring for the circular flavour, zeros at the global
        edges for the spatial one): of ``vol`` in real space, or -- ``vol`` None, fused pipeline -- of the context's
        x-transformed input buffer."""
        self.exchange_finish(self.exchange_start(vol, finish_now=True))

    # ------------------------------------------------------------------ iteration
    def iterate(self):
        """One RL iteration (decon.m:61-79 / 162-186, lambda = 0) on the sharded volume."""
        if self.sharded and self.zb is not None:
            return self._iterate_zchunked()
        if self.sharded:
            if not self._begun:                  # x-forward of the start volume; later iterations get it from the update
                self.ctx.sharded_begin(self.bl)
                self.exchange()
                self._begun = True
            if self.overlap:
                # the x tiles that hold the rows to be sent run first; the transfer then overlaps with the rest of the x pass
                e = self.edge_rows
                self.ctx.sharded_ratio(self.bl, 1, e)
                st = self.exchange_start()
                self.ctx.sharded_ratio(self.bl, 2, e)
                self.exchange_finish(st)
                self.ctx.sharded_update(self.bl, True, 1, e)
                st = self.exchange_start()
                self.ctx.sharded_update(self.bl, True, 2, e)
                self.exchange_finish(st)
                return
            self.ctx.sharded_ratio(self.bl)
            self.exchange()
            self.ctx.sharded_update(self.bl, True)
            self.exchange()
            return
        self.exchange(self.bl)
        self.ctx.forward_ratio(self.bl, self.ratio)
        self.exchange(self.ratio)
        self.ctx.adjoint_update(self.ratio, self.bl)

    # ------------------------------------------------------------------ z-chunked exchange (fused pipeline)
    def _zx_begin(self):
        """Buffers of the next exchange: (kind, send_up, send_dn, recv_lo, recv_hi, per-chunk requests, staged)."""
        h = self.h
        lo_src, hi_src = self.neighbours()
        nfl = self.ops.halo_floats(self.ctx, self.lshape, h)
        if self.world > 1 and self.transport == "peer":
            if self.link is None:
                self.link = PeerLink(self, nfl, self.ops.peer_backend(), self.group, chunks=len(self.zb))
            up, dn = self.link.begin()
            return {"kind": "peer", "up": up, "dn": dn}
        mk = lambda: torch.empty(nfl, dtype=torch.float32, device=self.device)   # noqa: E731
        st = {"kind": "self" if self.world == 1 else "rccl", "up": mk() if hi_src is not None else None,
              "dn": mk() if lo_src is not None else None, "reqs": [None] * len(self.zb), "keep": []}
        if st["kind"] == "rccl":
            import torch.distributed as dist
            st["staged"] = self.device.type == "cuda" and dist.get_backend(self.group) == "gloo"
            host = "cpu" if st["staged"] else self.device
            st["recv_lo"] = torch.empty(nfl, dtype=torch.float32, device=host) if lo_src is not None else None
            st["recv_hi"] = torch.empty(nfl, dtype=torch.float32, device=host) if hi_src is not None else None
        return st

    def _zx_send(self, st, k):
        """Pack chunk k of the edge rows and send it off."""
        import torch.distributed as dist
        h, n = self.h, self.n_loc
        z0, z1 = self.zb[k]
        off, cnt = self.ops.spec_chunk_span(self.ctx, h, z0, z1, self.nzp)
        for buf, y0 in ((st["up"], n), (st["dn"], h)):
            if buf is not None:
                self.ops.pack_spec_chunk(self.ctx, y0, h, z0, z1, buf)
        if st["kind"] == "peer":
            self.link.send(k, off, cnt)
            return
        if st["kind"] == "self":
            return
        lo_src, hi_src = self.neighbours()
        up = st["up"][off:off + cnt] if st["up"] is not None else None
        dn = st["dn"][off:off + cnt] if st["dn"] is not None else None
        if st["staged"]:
            up, dn = (up.cpu() if up is not None else None), (dn.cpu() if dn is not None else None)
        ops = []
        if hi_src is not None:
            ops.append(dist.P2POp(dist.isend, up, hi_src, self.group, 1))
        if lo_src is not None:
            ops.append(dist.P2POp(dist.isend, dn, lo_src, self.group, 2))
        if lo_src is not None:
            ops.append(dist.P2POp(dist.irecv, st["recv_lo"][off:off + cnt], lo_src, self.group, 1))
        if hi_src is not None:
            ops.append(dist.P2POp(dist.irecv, st["recv_hi"][off:off + cnt], hi_src, self.group, 2))
        st["reqs"][k] = dist.batch_isend_irecv(ops)
        st["keep"].append((up, dn))                      # the send slices stay alive until the wait

    def _zx_recv(self, st, k):
        """Wait for chunk k of the halo rows and unpack it (the launch stream waits; the host only where the transport needs it)."""
        h, n = self.h, self.n_loc
        z0, z1 = self.zb[k]
        if st["kind"] == "peer":
            for d, y0 in ((0, 0), (1, h + n)):
                self.ops.unpack_spec_chunk(self.ctx, self.link.receive(d, k), y0, h, z0, z1)
            return
        if st["kind"] == "self":                         # self-ring: my own rows wrap around
            lo_src, hi_src = self.neighbours()
            self.ops.unpack_spec_chunk(self.ctx, st["up"] if lo_src is not None else None, 0, h, z0, z1)
            self.ops.unpack_spec_chunk(self.ctx, st["dn"] if hi_src is not None else None, h + n, h, z0, z1)
            return
        for req in st["reqs"][k]:
            req.wait()
        off, cnt = self.ops.spec_chunk_span(self.ctx, h, z0, z1, self.nzp)
        for key, y0 in (("recv_lo", 0), ("recv_hi", h + n)):
            src = st[key]
            if src is not None and st["staged"]:
                dev = st.setdefault(key + "_dev", torch.empty(src.numel(), dtype=torch.float32, device=self.device))
                dev[off:off + cnt].copy_(src[off:off + cnt])
                src = dev
            self.ops.unpack_spec_chunk(self.ctx, src, y0, h, z0, z1)

    def _zx_done(self, st):
        if st["kind"] == "peer":
            self.link.release()

    def drain(self):
        """Completes the exchange the last half-step left in flight (z-chunked mode): every rank calls it after its last iteration
        -- ``run`` and ``close`` do -- so that no transfer or acknowledgement is left dangling."""
        if self._pending is not None:
            for k in range(len(self.zb)):
                self._zx_recv(self._pending, k)
            self._zx_done(self._pending)
            self._pending = None                         # (the halo rows are in place: a later iterate goes on from here)

    def _iterate_zchunked(self):
        e, K = self.edge_rows, range(len(self.zb))
        if not self._begun:                              # x-forward of the start volume, its halo rows chunk by chunk
            self.ctx.sharded_begin(self.bl)
            st = self._zx_begin()
            for k in K:
                self._zx_send(st, k)
            self._pending, self._begun = st, True
        for update in (False, True):
            for k in K:                                  # y-forward on every chunk whose halo rows have landed
                if self._pending is not None:
                    self._zx_recv(self._pending, k)
                self.ops.stage(self.ctx, self.bl, update, 0, *self.zb[k])
            if self._pending is not None:
                self._zx_done(self._pending)
            self.ops.stage(self.ctx, self.bl, update, 1)                     # z pass + y-inverse
            st = self._zx_begin()
            for k in K:                                  # edge tiles of a chunk, then that chunk is on its way
                self.ops.stage(self.ctx, self.bl, update, 2, *self.zb[k], e)
                self._zx_send(st, k)
            self.ops.stage(self.ctx, self.bl, update, 3, 0, 0, e)            # everything else runs while the halos travel
            self._pending = st

    def run(self, niter, stop_criterion=0.0):
        """``niter`` iterations with the reference's stop test on the GLOBAL norm (decon.m:108-118)."""
        prev = self.norm2() if stop_criterion > 0 else 0.0
        done = 0
        for i in range(1, niter + 1):
            self.iterate()
            done = i
            if stop_criterion > 0:
                cur = self.norm2()
                rel = abs(prev - cur) / prev * 100.0
                prev = cur
                if i > 1 and rel <= stop_criterion:
                    break
        self.drain()
        if self.link is not None:
            self.link.check()                            # a neighbour that never delivered: an error, not silence
        return done

    def close(self):
        """Releases the copy-engine link (mapped peer memory, its own exported buffers); every rank calls it before the process
        group goes away."""
        self.drain()
        if self.link is not None:
            self.link.close()
            self.link = None

    def interior(self):
        return self.bl[:, self.h:self.h + self.n_loc, :]

    def norm2(self) -> float:
        import torch.distributed as dist
        s = (self.interior().double() ** 2).sum().reshape(1)
        if self.world > 1:
            dist.all_reduce(s, op=dist.ReduceOp.SUM, group=self.group)
        return float(s.sqrt().item())

    def gather(self):
        """The whole volume on every rank (tests / small volumes only)."""
        import torch.distributed as dist
        mine = self.interior().contiguous()
        if self.world == 1:
            return mine
        rows = [b - a for a, b in slab_rows(self.gshape[1], self.world)]
        parts = [torch.empty((self.gshape[0], n, self.gshape[2]), dtype=mine.dtype, device=mine.device) for n in rows]
        dist.all_gather(parts, mine, group=self.group) if len(set(rows)) == 1 else self._gather_uneven(parts, mine)
        return torch.cat(parts, dim=1)

    def _gather_uneven(self, parts, mine):
        import torch.distributed as dist
        for r in range(self.world):
            if r == self.rank:
                parts[r].copy_(mine)
            dist.broadcast(parts[r], src=r, group=self.group)
