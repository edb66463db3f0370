// Single-pass separable 3-D convolution with the RL epilogues (direct engine, rank-1 PSFs: BASELINE config 1's Gaussian and every
// other outer-product PSF).
//
// The reference applies every PSF as the dense tap loop (conv3d_gpu.cu:68-99: kx * ky * kz taps per voxel, every tap a global
// load; convn in decon.m:61,64).  A PSF that is an outer product a (x) b (x) c factors into three 1-D convolutions; run as three
// launches they move 24 B/voxel.  Here they are ONE pass of 8 B/voxel (+ the epilogue operand): the structure of k_gauss3d_fused
// (gauss3d.hip) with arbitrary taps, a window offset per axis (even extents, deconFFT's placement), one of the three boundary
// rules per axis and the RL epilogues of the direct engine.  A work-group owns a 64 x 16 (x, y) tile and marches along z: the
// boundary-resolved input patch of plane p + 1 travels into registers (16-byte loads where the patch lies inside the volume)
// while plane p is filtered along x and y in LDS; the xy-filtered plane joins an LDS ring of the last kz planes, one z-filtered
// plane leaves per step through the epilogue with 16-byte stores.  Every 1-D result is rounded to fp32, taps in window order.
#include "conv3d_direct.h"

namespace mi {
namespace {

constexpr int SF_TY = 16, SF_TX = 64, SF_THREADS = 256;
constexpr size_t kSepLdsMax = 150 * 1024;

// source index of grid coordinate i under a boundary rule (-1: the sample is zero)
__device__ __forceinline__ int bnd_index(int i, int n, int rule) {
    if (rule == MI_BOUNDARY_REPLICATE) return min(max(i, 0), n - 1);
    if (rule == MI_BOUNDARY_CIRCULAR) {
        i %= n;
        return i < 0 ? i + n : i;
    }
    return (i < 0 || i >= n) ? -1 : i;
}

struct SepGeom {
    int nx, ny, nz;
    int cx, cy, cz;     // window start offsets: out[i] = sum_t in[i - c + t] w[t]
    int bx, by, bz;     // boundary rule per axis
    int zchunk;
};

template <int EPI, int NPRE>
__global__ __launch_bounds__(SF_THREADS) void k_sep3d(const float* __restrict__ src, float* __restrict__ dst, ConvEpilogue epi, SepGeom g,
                                                      SepTaps tx, SepTaps ty, SepTaps tz) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nx = g.nx, ny = g.ny, nz = g.nz;
    const int cql = (g.cx + 3) / 4, cqr = (tx.n - 1 - g.cx + 3) / 4;  // x halo in float4 units, left / right
    const int segq = SF_TX / 4 + cql + cqr;                           // float4 per staged row
    const int seg = 4 * segq, rows_in = SF_TY + ty.n - 1;
    float* in = lds;                          // [rows_in][seg]
    float* xf = in + rows_in * seg;           // [rows_in][64]
    float* ring = xf + rows_in * SF_TX;       // [tz.n][16][64]
    const int tid = threadIdx.x, xq = tid & 15, rsub = tid >> 4;
    // XCD-aware tile order: a contiguous range of tiles per XCD keeps the halos of neighbouring tiles in one L2 (k_gauss3d_fused)
    const int gx = (nx + SF_TX - 1) / SF_TX, gy = (ny + SF_TY - 1) / SF_TY, gz = (nz + g.zchunk - 1) / g.zchunk;
    const int total = gx * gy * gz, per = (total + 7) / 8;
    const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (t >= total) return;
    const int bz = t / (gx * gy), by = (t - bz * gx * gy) / gx, bx = t - bz * gx * gy - by * gx;
    const int x0 = bx * SF_TX, y0 = by * SF_TY;
    const int za = bz * g.zchunk, zb = min(za + g.zchunk, nz);
    const int xoff = 4 * cql - g.cx;          // first tap of output x sits at staged column x + xoff
    const int xs0 = x0 - 4 * cql;             // grid x of staged column 0
    int slot = 0;
    float4 pre[NPRE];
    // the patch of walk position p (input plane p under the z rule), requested one plane ahead
    auto fetch = [&](int p) {
        const int zi = bnd_index(p, nz, g.bz);
        const float* plane = src + (size_t)max(zi, 0) * ny * nx;
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int it = tid + u * SF_THREADS;
            if (it < rows_in * segq) {
                const int r = it / segq, q = it - r * segq;
                const int yi = bnd_index(y0 - g.cy + r, ny, g.by);
                const int x = xs0 + 4 * q;
                float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (zi >= 0 && yi >= 0) {
                    const float* row = plane + (size_t)yi * nx;
                    if (x >= 0 && x + 3 < nx) {
                        v = *reinterpret_cast<const float4*>(row + x);
                    } else {
                        const int i0 = bnd_index(x, nx, g.bx), i1 = bnd_index(x + 1, nx, g.bx), i2 = bnd_index(x + 2, nx, g.bx),
                                  i3 = bnd_index(x + 3, nx, g.bx);
                        v = make_float4(i0 >= 0 ? row[i0] : 0.0f, i1 >= 0 ? row[i1] : 0.0f, i2 >= 0 ? row[i2] : 0.0f, i3 >= 0 ? row[i3] : 0.0f);
                    }
                }
                pre[u] = v;
            }
        }
    };
    const int p_first = za - g.cz, p_last = zb - 1 - g.cz + tz.n - 1;  // walk positions (grid z before the rule)
    fetch(p_first);
    for (int p = p_first; p <= p_last; ++p) {
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int it = tid + u * SF_THREADS;
            if (it < rows_in * segq) {
                const int r = it / segq, q = it - r * segq;
                *reinterpret_cast<float4*>(in + r * seg + 4 * q) = pre[u];
            }
        }
        __syncthreads();
        if (p < p_last) fetch(p + 1);
        for (int it = tid; it < rows_in * 16; it += SF_THREADS) {   // x filter: 4 outputs from kx + 3 staged samples
            const int r = it >> 4, q = it & 15;
            const float* a = in + r * seg + 4 * q + xoff;
            float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f, o3 = 0.0f;
            float v0 = a[0], v1 = a[1], v2 = a[2];
            for (int k = 0; k < tx.n; ++k) {
                const float v3 = a[k + 3], w = tx.w[k];
                o0 = fmaf(v0, w, o0);
                o1 = fmaf(v1, w, o1);
                o2 = fmaf(v2, w, o2);
                o3 = fmaf(v3, w, o3);
                v0 = v1; v1 = v2; v2 = v3;
            }
            *reinterpret_cast<float4*>(xf + r * SF_TX + 4 * q) = make_float4(o0, o1, o2, o3);
        }
        __syncthreads();
        {   // y filter of row rsub, columns 4 xq .. + 3 -> ring[slot]
            float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            for (int k = 0; k < ty.n; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(xf + (rsub + k) * SF_TX + 4 * xq);
                const float w = ty.w[k];
                acc.x = fmaf(v.x, w, acc.x); acc.y = fmaf(v.y, w, acc.y); acc.z = fmaf(v.z, w, acc.z); acc.w = fmaf(v.w, w, acc.w);
            }
            *reinterpret_cast<float4*>(ring + ((size_t)slot * SF_TY + rsub) * SF_TX + 4 * xq) = acc;
        }
        const int zo = p + g.cz - (tz.n - 1);  // output plane whose window [zo - cz, zo - cz + kz - 1] is now complete
        if (zo >= za) {
            float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            int rs = slot + 1;  // slot of the window's first plane (kz slots back, wrapping)
            if (rs >= tz.n) rs -= tz.n;
            for (int k = 0; k < tz.n; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(ring + ((size_t)rs * SF_TY + rsub) * SF_TX + 4 * xq);
                const float w = tz.w[k];
                acc.x = fmaf(v.x, w, acc.x); acc.y = fmaf(v.y, w, acc.y); acc.z = fmaf(v.z, w, acc.z); acc.w = fmaf(v.w, w, acc.w);
                if (++rs >= tz.n) rs = 0;
            }
            const int y = y0 + rsub, x = x0 + 4 * xq;
            if (y < ny && x < nx) {
                const size_t idx = ((size_t)zo * ny + y) * nx + x;
                float4 o = acc;
                if (EPI != EPI_NONE) {
                    const float4 a = *reinterpret_cast<const float4*>(epi.a + idx);
                    if (EPI == EPI_RATIO) {
                        o = make_float4(a.x / fmaxf(acc.x, kEpsSingle), a.y / fmaxf(acc.y, kEpsSingle), a.z / fmaxf(acc.z, kEpsSingle),
                                        a.w / fmaxf(acc.w, kEpsSingle));
                    } else if (EPI == EPI_UPDATE) {
                        o = make_float4(fabsf(a.x * acc.x), fabsf(a.y * acc.y), fabsf(a.z * acc.z), fabsf(a.w * acc.w));
                    } else {
                        const float4 b = *reinterpret_cast<const float4*>(epi.b + idx);
                        const float l = epi.lambda, m = 1.0f - epi.lambda;
                        o = make_float4(fabsf(a.x * acc.x * m + b.x * l), fabsf(a.y * acc.y * m + b.y * l), fabsf(a.z * acc.z * m + b.z * l),
                                        fabsf(a.w * acc.w * m + b.w * l));
                    }
                }
                *reinterpret_cast<float4*>(dst + idx) = o;
            }
        }
        if (++slot >= tz.n) slot = 0;
        // (the next plane's staging overwrites `in`, last read before the second barrier above; `xf` is rewritten only behind
        // the next first barrier, which every thread reaches after its y filter)
    }
}

// The same pass with the z window in REGISTERS (kz <= KZB taps): instead of a ring of the last kz xy-filtered planes in LDS -- 64 KB
// for 16 taps, 124 KB for 31: one work-group per CU, and kz 16-byte LDS reads per four outputs -- every thread keeps the kz partial
// sums of the output planes its four voxels are still collecting (acc[j]: output plane zo + j, zo the oldest one open).  An xy-filtered
// plane adds its term to all of them (tap kz - 1 - j), the oldest sum is complete and leaves through the epilogue, the others move
// down one slot.  The terms of an output arrive in window order, each one fmaf onto the sum so far -- the arithmetic of the ring
// version, bit for bit.  The register shifts cost VALU slots the pass has to spare; the LDS holds the patch and its x-filtered rows
// only (17 KB for 15 x 15 taps), so several work-groups share a CU and their barriers interleave.
template <int EPI, int NPRE, int KZB>
__global__ __launch_bounds__(SF_THREADS) void k_sep3d_acc(const float* __restrict__ src, float* __restrict__ dst, ConvEpilogue epi, SepGeom g,
                                                          SepTaps tx, SepTaps ty, SepTaps tzr) {
    // tzr: the z taps REVERSED and zero-filled up to KZB (compile-time indices: they live in scalar registers)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nx = g.nx, ny = g.ny, nz = g.nz;
    const int cql = (g.cx + 3) / 4, cqr = (tx.n - 1 - g.cx + 3) / 4;
    const int segq = SF_TX / 4 + cql + cqr;
    const int seg = 4 * segq, rows_in = SF_TY + ty.n - 1;
    const int xoff = 4 * cql - g.cx;          // first tap of output x sits at staged column x + xoff
    // The 1-D filters read LDS in 16-byte pieces, a whole piece of taps per trip, the next trip's operands requested before the
    // current one's products: with one tap per trip, its weight fetched from the argument block, every trip waited for an LDS
    // and a scalar-memory round trip (5 ms per iteration on a C2-sized volume with 9 x 9 x 15 taps, 1.3 TB/s).  The tap lists are
    // zero-filled to whole pieces (a zero weight leaves a sum as it is); what a zero weight multiplies is finite: staged samples
    // of a neighbouring row, or LDS that was cleared once.
    const int kyp = (ty.n + 3) & ~3, rows_pad = SF_TY + kyp - 1;     // y taps in fours; the rows the last four may touch
    const int nxc = (xoff + tx.n + 3 + 3) / 4;                        // 16-byte pieces of a row that four outputs draw on
    float* in = lds;                          // [rows_in][seg]
    float* xf = in + rows_in * seg;           // [rows_pad][64]  (rows >= rows_in stay zero)
    float* wx = xf + rows_pad * SF_TX;        // [4 nxc + 4]: wx[m] = tap m - 3 - xoff (0 outside the window)
    float* wy = wx + 4 * nxc + 4;             // [kyp]
    const int tid = threadIdx.x, xq = tid & 15, rsub = tid >> 4;
    const int gx = (nx + SF_TX - 1) / SF_TX, gy = (ny + SF_TY - 1) / SF_TY, gz = (nz + g.zchunk - 1) / g.zchunk;
    const int total = gx * gy * gz, per = (total + 7) / 8;
    const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (t >= total) return;
    for (int i = tid; i < rows_pad * SF_TX; i += SF_THREADS) xf[i] = 0.0f;
    for (int m = tid; m < 4 * nxc + 4; m += SF_THREADS) {
        const int k = m - 3 - xoff;
        wx[m] = (k >= 0 && k < tx.n) ? tx.w[k] : 0.0f;
    }
    for (int k = tid; k < kyp; k += SF_THREADS) wy[k] = k < ty.n ? ty.w[k] : 0.0f;
    const int bz = t / (gx * gy), by = (t - bz * gx * gy) / gx, bx = t - bz * gx * gy - by * gx;
    const int x0 = bx * SF_TX, y0 = by * SF_TY;
    const int za = bz * g.zchunk, zb = min(za + g.zchunk, nz);
    const int xs0 = x0 - 4 * cql;
    const int kz = tzr.n;
    // ---- the march.  Two patches are under way at any time (planes p + 1 and p + 2 while plane p is filtered: with one, a step
    // waited most of a memory round trip for its patch), and the epilogue operand of the NEXT step's output is requested before
    // them -- loads and stores of a wave retire in order, so an operand requested behind the patches could only be used once they
    // had landed.  For the waits to be counted no memory operation may sit in a branch: every load is unconditional -- an item
    // past the patch reads the last item again, a sample the boundary rule replaces comes from the nearest quad inside the
    // volume -- and the rule is applied when the registers are copied to LDS (item constants: the quad's offset inside a plane
    // with the rule in its two low bits -- 1: zero, 2 / 3: the quad's first / last sample four times; nx is a multiple of 4, so a
    // quad lies inside the row or outside, never across its end).
    const int n_items = rows_in * segq;
    int item_om[NPRE];
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
        const int it = min(tid + u * SF_THREADS, n_items - 1);
        const int r = it / segq, q = it - r * segq;
        const int yi = bnd_index(y0 - g.cy + r, ny, g.by);
        const int x = xs0 + 4 * q;
        int xi = x, mode = yi < 0 ? 1 : 0;
        if (x < 0 || x >= nx) {
            if (g.bx == MI_BOUNDARY_CIRCULAR) {
                xi = x % nx;
                if (xi < 0) xi += nx;
            } else {
                xi = x < 0 ? 0 : nx - 4;
                if (mode == 0) mode = g.bx == MI_BOUNDARY_REPLICATE ? (x < 0 ? 2 : 3) : 1;
            }
        }
        item_om[u] = (max(yi, 0) * nx + xi) | mode;   // (a plane has fewer than 2^31 samples: sep3d_launch)
    }
    auto fetch = [&](int p, float4 (&buf)[NPRE]) {
        const int zi = bnd_index(p, nz, g.bz);
        const float* plane = src + (size_t)max(zi, 0) * ny * nx;   // (a plane the z rule zeroes is read and dropped)
#pragma unroll
        for (int u = 0; u < NPRE; ++u) buf[u] = *reinterpret_cast<const float4*>(plane + (item_om[u] & ~3));
    };
    auto stage = [&](int p, const float4 (&buf)[NPRE]) {
        const bool zero_plane = bnd_index(p, nz, g.bz) < 0;
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int it = tid + u * SF_THREADS, mode = item_om[u] & 3;
            float4 v = buf[u];
            if (mode == 2) v = make_float4(v.x, v.x, v.x, v.x);
            if (mode == 3) v = make_float4(v.w, v.w, v.w, v.w);
            if (mode == 1 || zero_plane) v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (it < n_items) *reinterpret_cast<float4*>(in + 4 * it) = v;   // (item it = row it / segq, quad it % segq: rows are seg floats)
        }
    };
    // the lane's output voxels: row y0 + rsub, x0 + 4 xq .. + 3 (clamped for the operand loads; stores are guarded)
    const int oy = y0 + rsub, ox = x0 + 4 * xq;
    const bool o_live = oy < ny && ox < nx;
    const size_t o_base = (size_t)min(oy, ny - 1) * nx + min(ox, nx - 4);
    const size_t plane_sz = (size_t)ny * nx;
    auto operand = [&](const float* arr, int zo) {  // the epilogue operand of output plane zo (clamped into the chunk)
        return *reinterpret_cast<const float4*>(arr + (size_t)min(max(zo, za), zb - 1) * plane_sz + o_base);
    };
    float4 acc[KZB];
#pragma unroll
    for (int j = 0; j < KZB; ++j) acc[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int p_first = za - g.cz, p_last = zb - 1 - g.cz + kz - 1;
    const int zshift = g.cz - (kz - 1);   // output plane completed by walk position p: p + zshift
    // one step: plane p (in `buf`) is staged, the operand of the next step's output and the patch of plane p + 2 are requested
    // (into the operand registers of the next step and into `buf` itself), the plane is filtered, the oldest sum leaves
    auto step = [&](int p, float4 (&buf)[NPRE], const float4& a_cur, const float4& b_cur, float4& a_nxt, float4& b_nxt) {
        stage(p, buf);
        __syncthreads();
        if (EPI != EPI_NONE) a_nxt = operand(epi.a, p + 1 + zshift);
        if (EPI == EPI_UPDATE_REG) b_nxt = operand(epi.b, p + 1 + zshift);
        fetch(min(p + 2, p_last), buf);
        // x filter: sample j = 4 c + e of the piece row is term j - xoff - i of output i; wx is laid out so that its weight is
        // wx[4 c + e - i + 3].  The terms of an output arrive in window order.
        for (int it = tid; it < rows_in * 16; it += SF_THREADS) {
            const int r = it >> 4, q = it & 15;
            const float4* a = reinterpret_cast<const float4*>(in + r * seg + 4 * q);
            const float4* wq = reinterpret_cast<const float4*>(wx);
            float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f, o3 = 0.0f;
            float4 d = a[0], wa = wq[0], wb = wq[1];
            for (int c = 0; c < nxc; ++c) {
                const float4 dn = a[c + 1 < nxc ? c + 1 : c], wan = wb, wbn = wq[c + 2];   // (wx ends with a spare piece)
                o0 = fmaf(d.x, wa.w, o0); o0 = fmaf(d.y, wb.x, o0); o0 = fmaf(d.z, wb.y, o0); o0 = fmaf(d.w, wb.z, o0);
                o1 = fmaf(d.x, wa.z, o1); o1 = fmaf(d.y, wa.w, o1); o1 = fmaf(d.z, wb.x, o1); o1 = fmaf(d.w, wb.y, o1);
                o2 = fmaf(d.x, wa.y, o2); o2 = fmaf(d.y, wa.z, o2); o2 = fmaf(d.z, wa.w, o2); o2 = fmaf(d.w, wb.x, o2);
                o3 = fmaf(d.x, wa.x, o3); o3 = fmaf(d.y, wa.y, o3); o3 = fmaf(d.z, wa.z, o3); o3 = fmaf(d.w, wa.w, o3);
                d = dn; wa = wan; wb = wbn;
            }
            *reinterpret_cast<float4*>(xf + r * SF_TX + 4 * q) = make_float4(o0, o1, o2, o3);
        }
        __syncthreads();
        float4 yv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        {   // y filter, four taps per trip (their rows requested together, no second set: the other waves of the SIMD fill the gap)
            const float* col = xf + rsub * SF_TX + 4 * xq;
            const float4* wq = reinterpret_cast<const float4*>(wy);
            for (int k = 0; k < kyp; k += 4) {
                const float4 v0 = *reinterpret_cast<const float4*>(col + k * SF_TX), v1 = *reinterpret_cast<const float4*>(col + (k + 1) * SF_TX),
                             v2 = *reinterpret_cast<const float4*>(col + (k + 2) * SF_TX), v3 = *reinterpret_cast<const float4*>(col + (k + 3) * SF_TX),
                             w = wq[k >> 2];
                yv.x = fmaf(v0.x, w.x, yv.x); yv.y = fmaf(v0.y, w.x, yv.y); yv.z = fmaf(v0.z, w.x, yv.z); yv.w = fmaf(v0.w, w.x, yv.w);
                yv.x = fmaf(v1.x, w.y, yv.x); yv.y = fmaf(v1.y, w.y, yv.y); yv.z = fmaf(v1.z, w.y, yv.z); yv.w = fmaf(v1.w, w.y, yv.w);
                yv.x = fmaf(v2.x, w.z, yv.x); yv.y = fmaf(v2.y, w.z, yv.y); yv.z = fmaf(v2.z, w.z, yv.z); yv.w = fmaf(v2.w, w.z, yv.w);
                yv.x = fmaf(v3.x, w.w, yv.x); yv.y = fmaf(v3.y, w.w, yv.y); yv.z = fmaf(v3.z, w.w, yv.z); yv.w = fmaf(v3.w, w.w, yv.w);
            }
        }
        // plane p is term kz - 1 - j of the output plane behind acc[j] (tzr[j]; zero for j >= kz, where the sums stay zero)
#pragma unroll
        for (int j = 0; j < KZB; ++j) {
            const float w = tzr.w[j];
            acc[j].x = fmaf(yv.x, w, acc[j].x); acc[j].y = fmaf(yv.y, w, acc[j].y);
            acc[j].z = fmaf(yv.z, w, acc[j].z); acc[j].w = fmaf(yv.w, w, acc[j].w);
        }
        const int zo = p + zshift;
        if (zo >= za && zo < zb && o_live) {
            const float4 a0 = acc[0];
            const size_t idx = (size_t)zo * plane_sz + (size_t)oy * nx + ox;
            float4 o = a0;
            if (EPI != EPI_NONE) {
                const float4 a = a_cur;
                if (EPI == EPI_RATIO) {
                    o = make_float4(a.x / fmaxf(a0.x, kEpsSingle), a.y / fmaxf(a0.y, kEpsSingle), a.z / fmaxf(a0.z, kEpsSingle),
                                    a.w / fmaxf(a0.w, kEpsSingle));
                } else if (EPI == EPI_UPDATE) {
                    o = make_float4(fabsf(a.x * a0.x), fabsf(a.y * a0.y), fabsf(a.z * a0.z), fabsf(a.w * a0.w));
                } else {
                    const float4 b = b_cur;
                    const float l = epi.lambda, m = 1.0f - epi.lambda;
                    o = make_float4(fabsf(a.x * a0.x * m + b.x * l), fabsf(a.y * a0.y * m + b.y * l), fabsf(a.z * a0.z * m + b.z * l),
                                    fabsf(a.w * a0.w * m + b.w * l));
                }
            }
            *reinterpret_cast<float4*>(dst + idx) = o;
        }
#pragma unroll
        for (int j = 0; j + 1 < KZB; ++j) acc[j] = acc[j + 1];
        acc[KZB - 1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        // (`in` is rewritten behind the second barrier above, `xf` behind the next first barrier, which every thread reaches after
        // its y filter)
    };
    float4 bufX[NPRE], bufY[NPRE];
    float4 aX = make_float4(0.0f, 0.0f, 0.0f, 0.0f), aY = aX, bX = aX, bY = aX;
    if (EPI != EPI_NONE) aX = operand(epi.a, p_first + zshift);
    if (EPI == EPI_UPDATE_REG) bX = operand(epi.b, p_first + zshift);
    fetch(p_first, bufX);
    fetch(min(p_first + 1, p_last), bufY);
    // (two steps per trip: the register sets swap roles from step to step, and a copy from one to the other would wait for the
    // loads under way)
    // (an odd number of planes: the last step runs on the last plane once more and stores nothing -- a second step in a branch
    // would make the compiler wait for every load at the top of each trip, not knowing how many are under way)
    for (int p = p_first; p <= p_last; p += 2) {
        step(p, bufX, aX, bX, aY, bY);
        step(p + 1, bufY, aY, bY, aX, bX);
    }
}

size_t sep_lds_bytes(const int* k, const int* c) {
    const int cql = (c[0] + 3) / 4, cqr = (k[0] - 1 - c[0] + 3) / 4, seg = 4 * (SF_TX / 4 + cql + cqr), rows_in = SF_TY + k[1] - 1;
    return sizeof(float) * ((size_t)rows_in * seg + (size_t)rows_in * SF_TX + (size_t)k[2] * SF_TY * SF_TX);
}
size_t sep_lds_bytes_acc(const int* k, const int* c) {  // k_sep3d_acc: the patch and its x-filtered rows
    const int cql = (c[0] + 3) / 4, cqr = (k[0] - 1 - c[0] + 3) / 4, seg = 4 * (SF_TX / 4 + cql + cqr), rows_in = SF_TY + k[1] - 1;
    const int kyp = (k[1] + 3) & ~3, rows_pad = SF_TY + kyp - 1, nxc = (4 * cql - c[0] + k[0] + 3 + 3) / 4;
    return sizeof(float) * ((size_t)rows_in * seg + (size_t)rows_pad * SF_TX + 4 * nxc + 4 + kyp);
}
int sep_patch_quads(const int* k, const int* c) {
    const int cql = (c[0] + 3) / 4, cqr = (k[0] - 1 - c[0] + 3) / 4;
    return (SF_TY + k[1] - 1) * (SF_TX / 4 + cql + cqr);
}

}  // namespace

// whether the single-pass kernel takes this rank-1 convolution: rows of whole float4, tap counts within the argument block, the
// plane ring within the LDS, the staged patch within the prefetch registers
bool sep3d_fits(int nx, const int* k, const int* c) {
    for (int a = 0; a < 3; ++a)
        if (k[a] < 1 || k[a] > kSepMaxTaps || c[a] < 0 || c[a] >= k[a]) return false;
    return (nx % 4) == 0 && (k[2] <= 32 ? sep_lds_bytes_acc(k, c) <= 64 * 1024 : sep_lds_bytes(k, c) <= kSepLdsMax) &&
           sep_patch_quads(k, c) <= 6 * SF_THREADS;
}

int sep3d_launch(hipStream_t s, const float* in, float* out, int nx, int ny, int nz, const SepTaps taps[3], const int* offs, const int* bnd3,
                 int epi_kind, const ConvEpilogue& epi) {
    const int k[3] = {taps[0].n, taps[1].n, taps[2].n};
    MI_REQUIRE(sep3d_fits(nx, k, offs), "separable convolution: taps %d x %d x %d do not fit the single-pass kernel", k[0], k[1], k[2]);
    MI_REQUIRE(in != out, "separable convolution: input and output must differ");
    MI_REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)epi.a % 16) == 0 && ((uintptr_t)epi.b % 16) == 0,
               "separable convolution: buffers must be 16-byte aligned");
    MI_REQUIRE(epi_kind == EPI_NONE || epi_kind == EPI_RATIO || epi_kind == EPI_UPDATE || epi_kind == EPI_UPDATE_REG,
               "separable convolution: unknown epilogue %d", epi_kind);
    MI_REQUIRE(epi_kind == EPI_NONE || epi.a, "separable convolution: the epilogue needs its operand");
    MI_REQUIRE(epi_kind != EPI_UPDATE_REG || epi.b, "separable convolution: the regularised update needs its second operand");
    SepGeom g{nx, ny, nz, offs[0], offs[1], offs[2], bnd3[0], bnd3[1], bnd3[2], 0};
    // chunks along z: long enough that the kz - 1 planes of run-in are a small share, short enough to fill the device
    const int tiles_xy = ((nx + SF_TX - 1) / SF_TX) * ((ny + SF_TY - 1) / SF_TY);
    int zchunk = std::max(64, 8 * k[2]);
    while (zchunk > 2 * k[2] && zchunk > 16 && (size_t)tiles_xy * ((nz + zchunk - 1) / zchunk) < 1024) zchunk /= 2;
    g.zchunk = std::min(zchunk, nz);
    const int total = tiles_xy * ((nz + g.zchunk - 1) / g.zchunk);
    const bool wide = sep_patch_quads(k, offs) > 3 * SF_THREADS;
    const dim3 grid((unsigned)((total + 7) / 8 * 8));
    if (k[2] <= 32 && (size_t)ny * nx < ((size_t)1 << 31) && !MI_PROBE_ENV("MI_SEP_RING")) {  // z window in registers (k_sep3d_acc); MI_SEP_RING=1 (probes build): the LDS-ring kernel
        const size_t lds_a = sep_lds_bytes_acc(k, offs);
        SepTaps tzr{};
        tzr.n = k[2];
        for (int j = 0; j < k[2]; ++j) tzr.w[j] = taps[2].w[k[2] - 1 - j];
#define MI_SEPA3(E, NP, KB) hipLaunchKernelGGL((k_sep3d_acc<E, NP, KB>), grid, dim3(SF_THREADS), lds_a, s, in, out, epi, g, taps[0], taps[1], tzr)
#define MI_SEPA2(E, NP)                                                                     \
    do {                                                                                    \
        if (k[2] <= 8) MI_SEPA3(E, NP, 8);                                                  \
        else if (k[2] <= 16) MI_SEPA3(E, NP, 16);                                           \
        else MI_SEPA3(E, NP, 32);                                                           \
    } while (0)
#define MI_SEPA(E)                                                                          \
    case E:                                                                                 \
        if (wide) MI_SEPA2(E, 6);                                                           \
        else MI_SEPA2(E, 3);                                                                \
        break;
        switch (epi_kind) {
            MI_SEPA(EPI_NONE) MI_SEPA(EPI_RATIO) MI_SEPA(EPI_UPDATE) MI_SEPA(EPI_UPDATE_REG)
            default: break;
        }
#undef MI_SEPA
#undef MI_SEPA2
#undef MI_SEPA3
        return launch_check("k_sep3d_acc");
    }
    const size_t lds = sep_lds_bytes(k, offs);
    MI_REQUIRE(lds <= kSepLdsMax, "separable convolution: the plane ring of %d taps does not fit the LDS (planes of 2^31 samples and more take the ring kernel)", k[2]);
#define MI_SEP(E)                                                                                                                  \
    case E:                                                                                                                        \
        if (wide) {                                                                                                                \
            if (lds > 64 * 1024)                                                                                                   \
                MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sep3d<E, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL((k_sep3d<E, 6>), grid, dim3(SF_THREADS), lds, s, in, out, epi, g, taps[0], taps[1], taps[2]);       \
        } else {                                                                                                                   \
            if (lds > 64 * 1024)                                                                                                   \
                MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sep3d<E, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL((k_sep3d<E, 3>), grid, dim3(SF_THREADS), lds, s, in, out, epi, g, taps[0], taps[1], taps[2]);       \
        }                                                                                                                          \
        break;
    switch (epi_kind) {
        MI_SEP(EPI_NONE) MI_SEP(EPI_RATIO) MI_SEP(EPI_UPDATE) MI_SEP(EPI_UPDATE_REG)
        default: break;
    }
#undef MI_SEP
    return launch_check("k_sep3d");
}

}  // namespace mi
