"""Test doubles for the slab driver: a numpy convolution context (so the sharding / halo-exchange logic can run on
CPU ranks under gloo) and a lock-step driver that runs several slabs in one process."""
import numpy as np
import torch

from ipp_amd import capi


class NumpyCtx:
    """forward_ratio / adjoint_update with per-axis boundary + PSF placement, float64 FFTs (independent of the
    product's kernels and of oracle/rl_oracle.py's loops)."""
    engine = -1

    def __init__(self, lshape, psf, boundary_xyz, shift_xyz, psf_inv=None):
        self.n = tuple(lshape)
        k = psf.shape
        bnd = boundary_xyz[::-1]   # -> (z, y, x)
        shift = shift_xyz[::-1]
        self.pad = [0 if b == capi.BOUNDARY_CIRCULAR else kk for b, kk in zip(bnd, k)]
        F = [nn + 2 * p for nn, p in zip(self.n, self.pad)]
        img = np.zeros(F, np.float64)
        img[:k[0], :k[1], :k[2]] = psf
        img = np.roll(img, [-s for s in shift], axis=(0, 1, 2))
        self.otf = np.fft.fftn(img)
        self.otf_adj = self.otf.conj()
        if psf_inv is not None:            # an explicit adjoint kernel: an ordinary convolution at the forward placement
            img = np.zeros(F, np.float64)
            img[:k[0], :k[1], :k[2]] = psf_inv
            self.otf_adj = np.fft.fftn(np.roll(img, [-s for s in shift], axis=(0, 1, 2)))

    def _conv(self, a, adjoint):
        p = np.pad(a.astype(np.float64), [(q, q) for q in self.pad])
        o = self.otf_adj if adjoint else self.otf
        c = np.real(np.fft.ifftn(np.fft.fftn(p) * o))
        sl = tuple(slice(q, q + n) for q, n in zip(self.pad, self.n))
        return c[sl].astype(np.float32)

    # ---- the fused protocol of the native pipeline (ctx.fuses): the convolution input lives in an x-transformed buffer S
    # (here: numpy FFT along x of the padded array), halos are exchanged as rows of S
    fuses = False

    def _to_S(self, a):
        self.S = np.fft.fft(np.pad(a.astype(np.float64), [(q, q) for q in self.pad]), axis=2)

    def _conv_S(self, adjoint):
        o = self.otf_adj if adjoint else self.otf
        c = np.real(np.fft.ifftn(np.fft.fft2(self.S, axes=(0, 1)) * o))
        sl = tuple(slice(q, q + n) for q, n in zip(self.pad, self.n))
        return c[sl].astype(np.float32)

    def sharded_begin(self, bl):
        self._to_S(bl.numpy())

    def sharded_ratio(self, bl, part=0, edge_rows=None):
        if part == 2:  # the double computes everything in part 1
            return
        b = bl.numpy()
        self._to_S((b / np.maximum(self._conv_S(False), np.float32(2.0 ** -23))).astype(np.float32))

    def sharded_update(self, bl, more=True, part=0, edge_rows=None):
        if part == 2:
            return
        b = bl.numpy()
        bl.copy_(torch.from_numpy(np.abs(b * self._conv_S(True)).astype(np.float32)))
        if more:
            self._to_S(bl.numpy())

    # ---- the staged step of the z-chunked exchange (mi_rl_sharded_stage).  What each stage READS is what the real kernels read at
    # that point: stage 0 snapshots the planes it transforms (a chunk whose halo rows arrive later would be wrong), stages 2 / 3
    # reveal the new spectrum rows only where the real x pass has written them
    z_granule = 1
    _Sin = _Snew = None

    def sharded_stage(self, bl, update, stage, z0=0, z1=0, edge_rows=None):
        if stage == 0:
            if self._Sin is None:
                self._Sin = np.full_like(self.S, np.nan)
            self._Sin[z0:z1] = self.S[z0:z1]
            return
        if stage == 1:
            assert not np.isnan(self._Sin).any(), "a chunk of planes was never transformed"
            o = self.otf_adj if update else self.otf
            c = np.real(np.fft.ifftn(np.fft.fft2(self._Sin, axes=(0, 1)) * o))
            sl = tuple(slice(q, q + n) for q, n in zip(self.pad, self.n))
            self._c, self._Sin, self._Snew = c[sl].astype(np.float32), None, None
            return
        if self._Snew is None:                          # the epilogue + forward x transform of the whole slab, revealed piecewise
            b = bl.numpy()
            if update:
                bl.copy_(torch.from_numpy(np.abs(b * self._c).astype(np.float32)))
                new = bl.numpy()
            else:
                new = (b / np.maximum(self._c, np.float32(2.0 ** -23))).astype(np.float32)
            self._Snew = np.fft.fft(np.pad(new.astype(np.float64), [(q, q) for q in self.pad]), axis=2)
        a0, a1, b0, b1 = edge_rows
        edge = np.zeros(self.S.shape[1], bool)
        edge[a0:a1] = True
        edge[b0:b1] = True
        if stage == 2:
            self.S[z0:z1, edge, :] = self._Snew[z0:z1, edge, :]
        else:
            self.S[:, ~edge, :] = self._Snew[:, ~edge, :]

    def spectrum_pack(self, y0, rows):
        assert self.pad[1] == 0  # y is circular on the local extent
        return torch.view_as_real(torch.from_numpy(np.ascontiguousarray(self.S[:, y0:y0 + rows, :]))).contiguous()

    def spectrum_unpack(self, buf, y0, rows):
        self.S[:, y0:y0 + rows, :] = 0 if buf is None else torch.view_as_complex(buf).numpy()

    def forward_ratio(self, bl, ratio):
        b = bl.numpy()
        ratio.copy_(torch.from_numpy((b / np.maximum(self._conv(b, False), np.float32(2.0 ** -23))).astype(np.float32)))

    def adjoint_update(self, ratio, bl, lambda_=0.0, reg=None):
        b = bl.numpy()
        bl.copy_(torch.from_numpy(np.abs(b * self._conv(ratio.numpy(), True)).astype(np.float32)))


class ShmPeer:
    """Host double of slab.HipPeer for CPU ranks, same protocol (csrc/peer.hip): the exported "device" allocations are shared-memory
    files (a handle is the file's path, a slot pointer is (array, byte offset)), copies are host memcpys, the flag words are uint32
    in a second file -- a sender writes the sequence number behind its copy, a receiver polls for `word >= sequence` (on the host,
    bounded by a timeout), acknowledgements travel the same way.  What this tests is the link's protocol: sequence numbers,
    alternating buffer sets, acknowledgements before a set is overwritten, waits that are early, late or repeated, teardown."""

    ARR, ACK = 0, 2

    def __init__(self, timeout_s=60.0):
        self.timeout = float(timeout_s)
        self.files, self.peers, self.dst, self.timeouts = [], [], {}, 0

    def identity(self):
        return ("ordinal", 0)

    def _file(self, nbytes, fill):
        import os
        import tempfile
        fd, path = tempfile.mkstemp(prefix="mi_peer_test_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        os.ftruncate(fd, int(nbytes))
        os.close(fd)
        arr = np.memmap(path, dtype=np.uint8, mode="r+")
        arr[:] = fill
        self.files.append(path)
        return arr, path.encode()

    def create(self, nbytes):
        self.slot = (int(nbytes) + 255) & ~255
        self.payload, hp = self._file(4 * self.slot, 0xFF)          # (NaN pattern: stale reads would show)
        fl, hf = self._file(4096, 0)
        self.flags = fl.view(np.uint32)
        return hp, hf

    def connect(self, d, handles, ident):
        for i, (hp, _, _) in enumerate(self.peers):
            if hp == handles[0]:
                self.dst[d] = i
                return
        self.peers.append((handles[0], np.memmap(handles[0].decode(), dtype=np.uint8, mode="r+"),
                           np.memmap(handles[1].decode(), dtype=np.uint8, mode="r+").view(np.uint32)))
        self.dst[d] = len(self.peers) - 1

    def staging(self, nfloats):
        return torch.empty(int(nfloats), dtype=torch.float32)

    def _wait(self, word, want):
        import time
        t0 = time.monotonic()
        while int(self.flags[word]) < want:
            time.sleep(20e-6)
            if time.monotonic() - t0 > self.timeout:
                self.timeouts += 1
                return

    def begin(self, n, src_mask):
        for d in (0, 1):
            if n > 1 and (src_mask >> d & 1) and (1 - d) in self.dst:
                self.peers[self.dst[1 - d]][2][self.ACK + d] = n - 1

    def send(self, n, k, chunks, first_byte, nbytes, up, dn):
        st = n & 1
        for d, src in ((0, up), (1, dn)):
            if d not in self.dst:
                continue
            _, pay, fl = self.peers[self.dst[d]]
            if k == 0 and n > 2:
                self._wait(self.ACK + d, n - 2)
            off = (2 * d + st) * self.slot + first_byte
            pay[off:off + nbytes] = src.numpy().view(np.uint8).reshape(-1)[first_byte:first_byte + nbytes]
            fl[self.ARR + d] = (n - 1) * chunks + k + 1

    def recv(self, n, k, chunks, d):
        self._wait(self.ARR + d, (n - 1) * chunks + k + 1)
        return (self.payload, (2 * d + (n & 1)) * self.slot)

    def exchange(self, n, src_mask, up, dn):
        self.send(n, 0, 1, 0, 4 * (up if up is not None else dn).numel(), up, dn)
        return tuple(self.recv(n, 0, 1, d) if (src_mask >> d & 1) else None for d in (0, 1))

    def timed_out(self):
        return self.timeouts

    def disconnect(self):
        self.peers, self.dst = [], {}

    def destroy(self):
        import os
        for f in self.files:
            try:
                os.unlink(f)
            except OSError:
                pass
        self.files = []


def _ptr_tensor(ptr, nfloats):
    arr, off = ptr
    return torch.from_numpy(np.array(arr[off:off + 4 * nfloats]).view(np.float32))


class NumpyOps:
    device = torch.device("cpu")

    def __init__(self, fuses=0):
        self.fuses = int(fuses)

    def peer_backend(self):
        return ShmPeer()

    # z-chunked exchange: the double's packed rows are [z][rows][x] complex128 = four float32 words per sample
    def spec_planes(self, ctx, lshape):
        return int(ctx.otf.shape[0])

    def spec_granule(self, ctx):
        return 1

    def spec_chunk_span(self, ctx, h, z0, z1, nz):
        per_plane = 4 * h * ctx.otf.shape[2]
        return z0 * per_plane, (z1 - z0) * per_plane

    def pack_spec_chunk(self, ctx, y0, rows, z0, z1, out_full):
        off, cnt = self.spec_chunk_span(ctx, rows, z0, z1, 0)
        out_full[off:off + cnt].copy_(torch.from_numpy(np.ascontiguousarray(ctx.S[z0:z1, y0:y0 + rows, :]).view(np.float32).reshape(-1).copy()))

    def unpack_spec_chunk(self, ctx, src_full, y0, rows, z0, z1):
        if src_full is None:
            ctx.S[z0:z1, y0:y0 + rows, :] = 0
            return
        off, cnt = self.spec_chunk_span(ctx, rows, z0, z1, 0)
        if isinstance(src_full, tuple):                 # a receive slot of the ShmPeer double: (array, byte offset)
            arr, boff = src_full
            words = np.array(arr[boff + 4 * off:boff + 4 * (off + cnt)]).view(np.float32)
        else:
            words = src_full[off:off + cnt].numpy()
        ctx.S[z0:z1, y0:y0 + rows, :] = np.ascontiguousarray(words).view(np.complex128).reshape(z1 - z0, rows, ctx.S.shape[2])

    def stage(self, ctx, bl, update, stage, z0=0, z1=0, edge_rows=None):
        ctx.sharded_stage(bl, update, stage, z0, z1, edge_rows)

    def halo_floats(self, ctx, lshape, h):
        F = ctx.otf.shape                               # the double's spectra are complex128 on its padded grid
        return max(lshape[0] * h * lshape[2], 4 * F[0] * h * F[2])

    def pack_spec_into(self, ctx, y0, rows, out):
        t = self._spec_f32(ctx, y0, rows)
        out[:t.numel()].copy_(t)

    @staticmethod
    def _spec_f32(ctx, y0, rows):
        # the double keeps float64 spectra; the link moves float32 words, so a complex128 row travels as four of them
        return torch.from_numpy(np.ascontiguousarray(ctx.S[:, y0:y0 + rows, :]).view(np.float32).reshape(-1).copy())

    def unpack_spec_ptr(self, ctx, ptr, y0, rows):
        nz, _, nx = ctx.S.shape
        t = _ptr_tensor(ptr, 4 * nz * rows * nx)
        ctx.S[:, y0:y0 + rows, :] = t.numpy().view(np.complex128).reshape(nz, rows, nx)

    def pack_into(self, vol, y0, rows, out):
        t = vol[:, y0:y0 + rows, :].reshape(-1)
        out[:t.numel()].copy_(t)

    def unpack_ptr(self, ptr, vol, y0, rows):
        nz, _, nx = vol.shape
        vol[:, y0:y0 + rows, :] = _ptr_tensor(ptr, nz * rows * nx).reshape(nz, rows, nx)

    def make_ctx(self, lshape, psf, boundary_xyz, shift_xyz, engine, psf_inv=None):
        ctx = NumpyCtx(lshape, psf, boundary_xyz, shift_xyz, psf_inv)
        ctx.fuses = self.fuses
        return ctx

    def pack_spec(self, ctx, y0, rows):
        return ctx.spectrum_pack(y0, rows)

    def unpack_spec(self, ctx, packed, y0, rows):
        ctx.spectrum_unpack(packed, y0, rows)

    def pack(self, vol, y0, rows):
        return vol[:, y0:y0 + rows, :].contiguous()

    def unpack(self, packed, vol, y0):
        vol[:, y0:y0 + packed.shape[1], :] = packed

    def zero_rows(self, vol, y0, rows):
        vol[:, y0:y0 + rows, :] = 0


def lockstep_iterate_zchunked(slabs, niter):
    """All slabs of one volume in one process, z-chunked protocol: the same order of stages, packs and unpacks as
    SlabRL._iterate_zchunked on separate ranks -- the halo rows of a half-step are delivered chunk by chunk at the START of the
    next one, each chunk right before its planes are transformed along y."""
    s0 = slabs[0]
    assert all(s.zb is not None and s.zb == s0.zb for s in slabs)
    h = s0.h
    K = range(len(s0.zb))

    def new_bufs():
        return [[(torch.empty(s.ops.halo_floats(s.ctx, s.lshape, h), dtype=torch.float32, device=s.device)) for _ in range(2)] for s in slabs]

    def pack_chunk(bufs, k):
        for s, (up, dn) in zip(slabs, bufs):
            z0, z1 = s.zb[k]
            s.ops.pack_spec_chunk(s.ctx, s.n_loc, h, z0, z1, up)      # last h interior rows -> the next slab's lower halo
            s.ops.pack_spec_chunk(s.ctx, h, h, z0, z1, dn)            # first h interior rows -> the previous slab's upper halo

    def deliver_chunk(bufs, k):
        for s in slabs:
            lo, hi = s.neighbours()
            z0, z1 = s.zb[k]
            s.ops.unpack_spec_chunk(s.ctx, bufs[lo][0] if lo is not None else None, 0, h, z0, z1)
            s.ops.unpack_spec_chunk(s.ctx, bufs[hi][1] if hi is not None else None, h + s.n_loc, h, z0, z1)

    for s in slabs:
        s.ctx.sharded_begin(s.bl)
    pending = new_bufs()
    for k in K:
        pack_chunk(pending, k)
    for _ in range(niter):
        for update in (False, True):
            for k in K:
                deliver_chunk(pending, k)
                for s in slabs:
                    s.ops.stage(s.ctx, s.bl, update, 0, *s.zb[k])
            for s in slabs:
                s.ops.stage(s.ctx, s.bl, update, 1)
            nxt = new_bufs()
            for k in K:
                for s in slabs:
                    s.ops.stage(s.ctx, s.bl, update, 2, *s.zb[k], s.edge_rows)
                pack_chunk(nxt, k)
            for s in slabs:
                s.ops.stage(s.ctx, s.bl, update, 3, 0, 0, s.edge_rows)
            pending = nxt
    return torch.cat([s.interior() for s in slabs], dim=1)


def lockstep_iterate(slabs, niter):
    """Runs all slabs of one volume in one process: pack everywhere, deliver, compute -- the same four phases
    SlabRL.iterate() goes through on separate ranks."""
    def exchange(attr):
        packed = [s.pack_halos(getattr(s, attr)) for s in slabs]
        for s in slabs:
            lo, hi = s.neighbours()
            s.unpack_halos(getattr(s, attr), packed[lo][0] if lo is not None else None,
                           packed[hi][1] if hi is not None else None)

    def exchange_spec():
        packed = [s.pack_spec_halos() for s in slabs]
        for s in slabs:
            lo, hi = s.neighbours()
            s.unpack_spec_halos(packed[lo][0] if lo is not None else None, packed[hi][1] if hi is not None else None)

    if slabs[0].sharded:  # fused pipeline: halos travel as x-transformed rows
        for s in slabs:
            s.ctx.sharded_begin(s.bl)
        exchange_spec()
        def step(fn):
            if slabs[0].overlap:  # edge tiles, pack (= what the sends would carry), the other tiles, deliver
                for s in slabs:
                    fn(s, 1, s.edge_rows)
                packed = [s.pack_spec_halos() for s in slabs]
                for s in slabs:
                    fn(s, 2, s.edge_rows)
                for s in slabs:
                    lo, hi = s.neighbours()
                    s.unpack_spec_halos(packed[lo][0] if lo is not None else None, packed[hi][1] if hi is not None else None)
            else:
                for s in slabs:
                    fn(s, 0, None)
                exchange_spec()
        for _ in range(niter):
            step(lambda s, part, e: s.ctx.sharded_ratio(s.bl, part, e))
            step(lambda s, part, e: s.ctx.sharded_update(s.bl, True, part, e))
        return torch.cat([s.interior() for s in slabs], dim=1)
    for _ in range(niter):
        exchange("bl")
        for s in slabs:
            s.ctx.forward_ratio(s.bl, s.ratio)
        exchange("ratio")
        for s in slabs:
            s.ctx.adjoint_update(s.ratio, s.bl)
    return torch.cat([s.interior() for s in slabs], dim=1)
