// Hand-written FFT convolution pipeline for gfx950 (engine MI_ENGINE_FFT when every transform length is a
// power of two and no padding is needed; rocFFT remains the fallback for other shapes).
//
// Why: the rocFFT R2C/C2R route costs 12 transform kernels + multiply + epilogue per convolution
// (profiles/r01_bench_c3_rocfft_kernel_stats.csv: 62 ms per convolution on C3, its strided z pass alone
// 13.8 ms).  The RL iteration is HBM-bound, so the design goal is the minimum number of full-volume passes
// with every global access in >= 64-B contiguous segments:
//
//   real volume (Z,Y,X) is read as complex rows of Hx = X/2 samples z[n] = x[2n] + i x[2n+1]; a COMPLEX 3-D FFT
//   of size (Z, Y, Hx) is taken; the real-spectrum untangling, the OTF product and the re-tangling are done
//   point-wise on mirror pairs (k, -k) in the middle pass, so no (X/2+1)-wide array ever exists.
//
//   P1  x forward   rows -> LDS -> DIF FFT(Hx) -> spectrum S[z][px][y]   (transposed write: y fastest)
//   P2  y forward   contiguous columns of S, in place
//   P3  z forward + untangle * OTF (or conj) + retangle + z inverse, on mirror line pairs, S -> T
//   P4  y inverse   contiguous columns of T, in place
//   P5  x inverse   T[z][px][y] -> LDS -> DIT IFFT(Hx) -> real row -> fused RL epilogue -> out
//
// Forward transforms are decimation-in-frequency (natural in, bit-reversed out), inverse ones
// decimation-in-time (bit-reversed in, natural out), so no reordering pass exists: frequency-domain arrays
// simply live in bit-reversed positions (px, py, pz) and the OTF is pre-permuted once to match.
// Each transform runs inside LDS as "super-stages" of up to 4 fused radix-2 stages held in registers
// (16 points per lane), i.e. 3 LDS round trips for 512..4096 points.  LDS rows are padded (one slot per 32
// plus one per row) so that both the strided butterfly accesses and the transposed tile accesses are
// bank-conflict free for ds_read/write_b64.
#include <cmath>
#include <vector>

#include "fft_native.h"

namespace mi {
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ int phys(int i) { return i + (i >> 5); }
__host__ __device__ __forceinline__ int row_pitch(int n) { return n + (n >> 5) + 1; }

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a*conj(b)
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ unsigned brev_n(unsigned v, int bits) { return bits == 0 ? 0u : (__brev(v) >> (32 - bits)); }

// One super-stage: R = 2^LR points per lane, radix-2 stages s_hi..s_lo (forward, DIF) or s_lo..s_hi (inverse, DIT)
// on `batch` sequences of length N = 1 << logn stored at tile[b * pitch + phys(i)].  tw[e] = exp(-2 pi i e / N).
template <int LR, bool INVERSE>
__device__ __forceinline__ void super_stage(float2* tile, int batch, int pitch, int logn, int s_lo, const float2* __restrict__ tw) {
    constexpr int R = 1 << LR;
    const int groups = 1 << (logn - LR);
    const int h_lo = 1 << s_lo;
    const int total = batch * groups;
    for (int idx = threadIdx.x; idx < total; idx += kThreads) {
        const int b = idx >> (logn - LR), g = idx & (groups - 1);
        const int base = ((g >> s_lo) << (s_lo + LR)) | (g & (h_lo - 1));
        float2* row = tile + b * pitch;
        float2 v[R];
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = row[phys(base + j * h_lo)];
#pragma unroll
        for (int step = 0; step < LR; ++step) {
            const int bpos = INVERSE ? step : LR - 1 - step;  // local bit handled by this radix-2 stage
            const int s = s_lo + bpos;
            const int h = 1 << s;
            const int tshift = logn - 1 - s;  // exponent scale N / (2h)
#pragma unroll
            for (int j = 0; j < R; ++j) {
                if (j & (1 << bpos)) continue;
                const int jl = j & ((1 << bpos) - 1);
                const int e = ((base & (h - 1)) + jl * h_lo) << tshift;
                const float2 w = tw[e];
                const float2 a = v[j], c = v[j | (1 << bpos)];
                if (INVERSE) {
                    const float2 t = cmulc(c, w);
                    v[j] = cadd(a, t);
                    v[j | (1 << bpos)] = csub(a, t);
                } else {
                    v[j] = cadd(a, c);
                    v[j | (1 << bpos)] = cmul(csub(a, c), w);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < R; ++j) row[phys(base + j * h_lo)] = v[j];
    }
}

// full transform of `batch` LDS rows; callers place __syncthreads() before (data visible) -- one is issued after
// every super-stage here, so the tile is consistent on return
template <bool INVERSE>
__device__ void lds_fft(float2* tile, int batch, int pitch, int logn, const float2* __restrict__ tw) {
    // split logn into super-stages of at most 4 radix-2 stages; forward walks from the top stage down
    int sizes[4], ns = 0, rem = logn;
    while (rem > 0) {
        int r = rem >= 8 ? 4 : (rem > 4 ? (rem + 1) / 2 : rem);
        sizes[ns++] = r;
        rem -= r;
    }
    int s_next = INVERSE ? 0 : logn;
    for (int q = 0; q < ns; ++q) {
        const int r = sizes[q];
        const int s_lo = INVERSE ? s_next : s_next - r;
        switch (r) {
            case 1: super_stage<1, INVERSE>(tile, batch, pitch, logn, s_lo, tw); break;
            case 2: super_stage<2, INVERSE>(tile, batch, pitch, logn, s_lo, tw); break;
            case 3: super_stage<3, INVERSE>(tile, batch, pitch, logn, s_lo, tw); break;
            default: super_stage<4, INVERSE>(tile, batch, pitch, logn, s_lo, tw); break;
        }
        s_next = INVERSE ? s_next + r : s_next - r;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------- P1: x forward
// grid: (Y / TY) * Z tiles; tile = TY consecutive rows of one z-plane
__global__ __launch_bounds__(kThreads) void k_x_forward(const float* __restrict__ in, float2* __restrict__ S, NativeDims d,
                                                         const float2* __restrict__ tw) {
    extern __shared__ __attribute__((aligned(16))) float2 tile[];
    const int Hx = 1 << d.lhx, TY = d.ty, pitch = row_pitch(Hx);
    const int ytiles = d.ny / TY;
    const int z = blockIdx.x / ytiles, y0 = (blockIdx.x % ytiles) * TY;
    const float4* src = reinterpret_cast<const float4*>(in + ((size_t)z * d.ny + y0) * (size_t)(2 * Hx));
    const int quads = Hx / 2;  // float4 = 2 complex
    for (int i = threadIdx.x; i < TY * quads; i += kThreads) {
        const int r = i / quads, q = i - r * quads;
        const float4 v = src[(size_t)r * quads + q];
        float2* row = tile + r * pitch;
        row[phys(2 * q)] = make_float2(v.x, v.y);
        row[phys(2 * q + 1)] = make_float2(v.z, v.w);
    }
    __syncthreads();
    lds_fft<false>(tile, TY, pitch, d.lhx, tw);
    // transposed store: S[z][px][y0 + r], r fastest
    float2* dst = S + ((size_t)z * Hx) * d.ny + y0;
    for (int i = threadIdx.x; i < TY * Hx; i += kThreads) {
        const int px = i / TY, r = i - px * TY;
        dst[(size_t)px * d.ny + r] = tile[r * pitch + phys(px)];
    }
}

// ---------------------------------------------------------------------------------------------- P2 / P4: y passes
// contiguous columns: column c of the (Z * Hx) columns is buf[c * Y .. c * Y + Y)
template <bool INVERSE>
__global__ __launch_bounds__(kThreads) void k_y_pass(float2* __restrict__ buf, NativeDims d, const float2* __restrict__ tw) {
    extern __shared__ __attribute__((aligned(16))) float2 tile[];
    const int M = d.ny, TC = d.tc, pitch = row_pitch(M);
    float4* base = reinterpret_cast<float4*>(buf + (size_t)blockIdx.x * TC * M);
    const int quads = M / 2;
    for (int i = threadIdx.x; i < TC * quads; i += kThreads) {
        const int c = i / quads, q = i - c * quads;
        const float4 v = base[(size_t)c * quads + q];
        float2* row = tile + c * pitch;
        row[phys(2 * q)] = make_float2(v.x, v.y);
        row[phys(2 * q + 1)] = make_float2(v.z, v.w);
    }
    __syncthreads();
    lds_fft<INVERSE>(tile, TC, pitch, d.ly, tw);
    for (int i = threadIdx.x; i < TC * quads; i += kThreads) {
        const int c = i / quads, q = i - c * quads;
        const float2* row = tile + c * pitch;
        const float2 a = row[phys(2 * q)], b = row[phys(2 * q + 1)];
        base[(size_t)c * quads + q] = make_float4(a.x, a.y, b.x, b.y);
    }
}

// ---------------------------------------------------------------------------------------------- P3: z pass + OTF
// A-role planes: px even (xk = brev(px) < Hx/2) and px == 1 (xk == Hx/2).  For px in {0, 1} the mirror line lies in
// the same plane: every tile is processed in the A role (its mirror tile is only read) and only A is written, so each
// line is still written exactly once; otherwise both lines of a pair are written by the one tile that owns the pair.
// grid: (#A planes) * (Y / TL) tiles of TL consecutive py positions.
// OTF layout: G[(plane index)][py][pz] as float4 {Ga.re, Ga.im, Gb.re, Gb.im}, already scaled by 1/(Hx*Y*Z).
template <bool CONJ>
__global__ __launch_bounds__(kThreads) void k_z_conv(const float2* __restrict__ S, float2* __restrict__ T, const float4* __restrict__ G,
                                                      NativeDims d, const float2* __restrict__ tw) {
    extern __shared__ __attribute__((aligned(16))) float2 tile[];
    const int Hx = 1 << d.lhx, M = d.ny, L = d.nz, TL = d.tl, pitch = row_pitch(L);
    const int ytiles = M / TL;
    const int plane = blockIdx.x / ytiles;                 // 0 .. Hx/2
    const int py0 = (blockIdx.x % ytiles) * TL;
    const int px = plane == Hx / 2 ? 1 : 2 * plane;        // plane order: even px ascending, then px = 1
    const unsigned xk = brev_n((unsigned)px, d.lhx);
    const int pxB = (int)brev_n((Hx - xk) & (Hx - 1), d.lhx);
    // mirror block of py positions: blocks of TL aligned positions map to blocks (see file header)
    const unsigned ky0 = brev_n((unsigned)py0, d.ly);
    const int pyB_any = (int)brev_n((M - ky0) & (M - 1), d.ly);
    const int pyB0 = pyB_any & ~(TL - 1);
    const bool self_plane = (px == 0 || px == 1);
    float2* tA = tile;
    float2* tB = tile + TL * pitch;
    const size_t plane_stride = (size_t)M;              // S[z][px][py]: element (z, px, py) at ((z*Hx + px)*M + py)
    // load both tiles: lanes walk py fastest (TL * 8 B contiguous), then z
    for (int i = threadIdx.x; i < TL * L; i += kThreads) {
        const int z = i / TL, j = i - z * TL;
        const size_t zoff = (size_t)z * Hx;
        tA[j * pitch + phys(z)] = S[(zoff + px) * plane_stride + py0 + j];
        tB[j * pitch + phys(z)] = S[(zoff + pxB) * plane_stride + pyB0 + j];
    }
    __syncthreads();
    lds_fft<false>(tile, 2 * TL, pitch, d.lz, tw);
    // point-wise: element (line j, position pz) of A pairs with (line jB, position pzB) of B
    float sw, cw;
    sincospif(-2.0f * (float)xk / (float)(2 * Hx), &sw, &cw);  // w = exp(-2 pi i xk / Nx), Nx = 2 Hx
    const float2 w = make_float2(cw, sw);
    const float4* Gp = G + ((size_t)plane * M + py0) * L;
    for (int i = threadIdx.x; i < TL * L; i += kThreads) {
        const int j = i / L, pz = i - j * L;
        const unsigned ky = brev_n((unsigned)(py0 + j), d.ly);
        const int jB = (int)brev_n((M - ky) & (M - 1), d.ly) - pyB0;
        const unsigned kz = brev_n((unsigned)pz, d.lz);
        const int pzB = (int)brev_n((L - kz) & (L - 1), d.lz);
        const float2 a = tA[j * pitch + phys(pz)];
        const float2 bm = tB[jB * pitch + phys(pzB)];
        const float2 bc = cconj(bm);
        const float2 E = make_float2(0.5f * (a.x + bc.x), 0.5f * (a.y + bc.y));
        const float2 dlt = csub(a, bc);                          // a - conj(b)
        const float2 O = make_float2(0.5f * dlt.y, -0.5f * dlt.x);  // -i/2 * (a - conj(b))
        const float2 wO = cmul(w, O);
        const float2 Xa = cadd(E, wO), Xb = csub(E, wO);
        const float4 g = Gp[(size_t)j * L + pz];
        float2 Ga = make_float2(g.x, g.y), Gb = make_float2(g.z, g.w);
        if (CONJ) { Ga.y = -Ga.y; Gb.y = -Gb.y; }
        const float2 Ya = cmul(Xa, Ga), Yb = cmul(Xb, Gb);
        const float2 E2 = make_float2(0.5f * (Ya.x + Yb.x), 0.5f * (Ya.y + Yb.y));
        const float2 dY = csub(Ya, Yb);
        const float2 O2 = cmulc(make_float2(0.5f * dY.x, 0.5f * dY.y), w);  // (Ya - Yb) conj(w) / 2
        // Z'[k] = E' + i O' ; Z'[-k] = conj(E') + i conj(O')
        tA[j * pitch + phys(pz)] = make_float2(E2.x - O2.y, E2.y + O2.x);
        tB[jB * pitch + phys(pzB)] = make_float2(E2.x + O2.y, O2.x - E2.y);
    }
    __syncthreads();
    lds_fft<true>(tile, 2 * TL, pitch, d.lz, tw);
    for (int i = threadIdx.x; i < TL * L; i += kThreads) {
        const int z = i / TL, j = i - z * TL;
        const size_t zoff = (size_t)z * Hx;
        T[(zoff + px) * plane_stride + py0 + j] = tA[j * pitch + phys(z)];
        if (!self_plane) T[(zoff + pxB) * plane_stride + pyB0 + j] = tB[j * pitch + phys(z)];
    }
}

// ---------------------------------------------------------------------------------------------- P5: x inverse + epilogue
template <int EPI>
__global__ __launch_bounds__(kThreads) void k_x_inverse(const float2* __restrict__ T, float* __restrict__ out, ConvEpilogue e, NativeDims d,
                                                         const float2* __restrict__ tw) {
    extern __shared__ __attribute__((aligned(16))) float2 tile[];
    const int Hx = 1 << d.lhx, TY = d.ty, pitch = row_pitch(Hx);
    const int ytiles = d.ny / TY;
    const int z = blockIdx.x / ytiles, y0 = (blockIdx.x % ytiles) * TY;
    const float2* src = T + ((size_t)z * Hx) * d.ny + y0;
    for (int i = threadIdx.x; i < TY * Hx; i += kThreads) {
        const int px = i / TY, r = i - px * TY;
        tile[r * pitch + phys(px)] = src[(size_t)px * d.ny + r];
    }
    __syncthreads();
    lds_fft<true>(tile, TY, pitch, d.lhx, tw);
    const size_t row0 = ((size_t)z * d.ny + y0) * (size_t)(2 * Hx);
    const int quads = Hx / 2;
    float4* dst = reinterpret_cast<float4*>(out + row0);
    const float4* a4 = reinterpret_cast<const float4*>(e.a + row0);
    const float4* b4 = reinterpret_cast<const float4*>(e.b + row0);
    for (int i = threadIdx.x; i < TY * quads; i += kThreads) {
        const int r = i / quads, q = i - r * quads;
        const float2* row = tile + r * pitch;
        const float2 c0 = row[phys(2 * q)], c1 = row[phys(2 * q + 1)];
        float4 c = make_float4(c0.x, c0.y, c1.x, c1.y), o;
        const size_t gi = (size_t)r * quads + q;
        if (EPI == EPI_NONE) {
            o = c;
        } else {
            const float4 av = a4[gi];
            if (EPI == EPI_RATIO) {
                o = make_float4(av.x / fmaxf(c.x, kEpsSingle), av.y / fmaxf(c.y, kEpsSingle), av.z / fmaxf(c.z, kEpsSingle),
                                av.w / fmaxf(c.w, kEpsSingle));
            } else if (EPI == EPI_UPDATE) {
                o = make_float4(fabsf(av.x * c.x), fabsf(av.y * c.y), fabsf(av.z * c.z), fabsf(av.w * c.w));
            } else {
                const float4 bv = b4[gi];
                const float l = e.lambda, m = 1.0f - e.lambda;
                o = make_float4(fabsf(av.x * c.x * m + bv.x * l), fabsf(av.y * c.y * m + bv.y * l), fabsf(av.z * c.z * m + bv.z * l),
                                fabsf(av.w * c.w * m + bv.w * l));
            }
        }
        dst[gi] = o;
    }
}

// ---------------------------------------------------------------------------------------------- OTF repack
// From the R2C half spectrum H[kz][ky][kx], kx in [0, Hx] (rocFFT layout, unscaled or pre-scaled) to the pair
// layout of k_z_conv: G[plane][py][pz] = {H_full[xk], H_full[xk + Hx]} at (ky, kz) = (brev(py), brev(pz)).
__global__ __launch_bounds__(kThreads) void k_repack_otf(const float2* __restrict__ Hs, float4* __restrict__ G, NativeDims d, float scale) {
    const int Hx = 1 << d.lhx, M = d.ny, L = d.nz;
    const size_t total = (size_t)(Hx / 2 + 1) * M * L;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int pz = (int)(i % L);
        const size_t r = i / L;
        const int py = (int)(r % M), plane = (int)(r / M);
        const int px = plane == Hx / 2 ? 1 : 2 * plane;
        const int xk = (int)brev_n((unsigned)px, d.lhx), ky = (int)brev_n((unsigned)py, d.ly), kz = (int)brev_n((unsigned)pz, d.lz);
        const int W = Hx + 1;
        const float2 ga = Hs[((size_t)kz * M + ky) * W + xk];
        // H_full[xk + Hx, ky, kz] = conj(H[Hx - xk, -ky, -kz])
        const float2 gb = Hs[((size_t)((L - kz) & (L - 1)) * M + ((M - ky) & (M - 1))) * W + (Hx - xk)];
        G[i] = make_float4(ga.x * scale, ga.y * scale, gb.x * scale, -gb.y * scale);
    }
}

bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

}  // namespace

bool NativeFft::supported(const int F[3]) {
    // x: real length 2*Hx with 8 <= Hx <= 4096; y, z: 4 .. 4096; LDS tiles must fit
    return is_pow2(F[0]) && is_pow2(F[1]) && is_pow2(F[2]) && F[0] >= 16 && F[0] <= 8192 && F[1] >= 8 && F[1] <= 4096 && F[2] >= 8 &&
           F[2] <= 4096;
}

static size_t lds_bytes(int rows, int n) { return sizeof(float2) * (size_t)rows * row_pitch(n); }

int NativeFft::init(hipStream_t s, const int F[3], const float2* otf_half_spectrum, float scale) {
    dims.lhx = ilog2(F[0] / 2);
    dims.ly = ilog2(F[1]);
    dims.lz = ilog2(F[2]);
    dims.ny = F[1];
    dims.nz = F[2];
    const int Hx = F[0] / 2;
    const size_t budget = 68 * 1024;  // two work-groups per CU inside 160 KB
    auto fit = [&](int n, int maxrows, int mult) {
        int rows = maxrows;
        while (rows > 1 && (lds_bytes(rows * mult, n) > budget)) rows >>= 1;
        return rows;
    };
    dims.ty = std::min(fit(Hx, 16, 1), F[1]);
    dims.tc = std::min(fit(F[1], 16, 1), 1 << 30);
    dims.tl = std::min(fit(F[2], 16, 2), F[1]);
    while ((size_t)F[2] * Hx % dims.tc) dims.tc >>= 1;
    MI_REQUIRE(lds_bytes(dims.ty, Hx) <= 150 * 1024 && lds_bytes(dims.tc, F[1]) <= 150 * 1024 && lds_bytes(2 * dims.tl, F[2]) <= 150 * 1024,
               "native FFT: transform too long for LDS");
    n_cplx = (size_t)Hx * F[1] * F[2];
    MI_TRY(S.alloc(sizeof(float2) * n_cplx));
    MI_TRY(T.alloc(sizeof(float2) * n_cplx));
    MI_TRY(G.alloc(sizeof(float4) * (size_t)(Hx / 2 + 1) * F[1] * F[2]));
    // twiddle tables exp(-2 pi i e / N), e < N/2, in double on the host
    const int lens[3] = {Hx, F[1], F[2]};
    size_t off = 0, offs[3];
    for (int a = 0; a < 3; ++a) { offs[a] = off; off += (size_t)std::max(1, lens[a] / 2); }
    std::vector<float2> h(off);
    const double two_pi = 6.283185307179586476925286766559;
    for (int a = 0; a < 3; ++a)
        for (int e = 0; e < lens[a] / 2; ++e)
            h[offs[a] + e] = make_float2((float)std::cos(two_pi * e / lens[a]), (float)-std::sin(two_pi * e / lens[a]));
    MI_TRY(tw.alloc(sizeof(float2) * off));
    MI_HIP(hipMemcpyAsync(tw.p, h.data(), sizeof(float2) * off, hipMemcpyHostToDevice, s));
    tw_x = tw.as<float2>() + offs[0];
    tw_y = tw.as<float2>() + offs[1];
    tw_z = tw.as<float2>() + offs[2];
    const size_t total = (size_t)(Hx / 2 + 1) * F[1] * F[2];
    size_t blocks = (total + kThreads - 1) / kThreads;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_repack_otf, dim3((unsigned)blocks), dim3(kThreads), 0, s, otf_half_spectrum, G.as<float4>(), dims, scale);
    MI_TRY(launch_check("k_repack_otf"));
    MI_HIP(hipStreamSynchronize(s));  // host twiddle vector dies at scope exit
    // opt in to > 64 KB of dynamic LDS where a tile needs it
    const int big = 160 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_x_forward), hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_y_pass<false>), hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_y_pass<true>), hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_z_conv<false>), hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_z_conv<true>), hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_x_inverse<EPI_NONE>), hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_x_inverse<EPI_RATIO>), hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_x_inverse<EPI_UPDATE>), hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_x_inverse<EPI_UPDATE_REG>), hipFuncAttributeMaxDynamicSharedMemorySize, big);
    return MI_OK;
}

int NativeFft::conv(hipStream_t s, const float* in, bool conj_otf, float* out, int epi_kind, const ConvEpilogue& epi) {
    const int Hx = 1 << dims.lhx, M = dims.ny, L = dims.nz;
    MI_REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0 && (!epi.a || ((uintptr_t)epi.a % 16) == 0) &&
                   (!epi.b || ((uintptr_t)epi.b % 16) == 0),
               "native FFT: volume pointers must be 16-byte aligned");
    float2* Sp = S.as<float2>();
    float2* Tp = T.as<float2>();
    hipLaunchKernelGGL(k_x_forward, dim3((unsigned)((size_t)L * (M / dims.ty))), dim3(kThreads), lds_bytes(dims.ty, Hx), s, in, Sp, dims, tw_x);
    MI_TRY(launch_check("k_x_forward"));
    const unsigned ycols = (unsigned)((size_t)L * Hx / dims.tc);
    hipLaunchKernelGGL(k_y_pass<false>, dim3(ycols), dim3(kThreads), lds_bytes(dims.tc, M), s, Sp, dims, tw_y);
    MI_TRY(launch_check("k_y_pass<fwd>"));
    const unsigned ztiles = (unsigned)((size_t)(Hx / 2 + 1) * (M / dims.tl));
    if (conj_otf)
        hipLaunchKernelGGL(k_z_conv<true>, dim3(ztiles), dim3(kThreads), lds_bytes(2 * dims.tl, L), s, Sp, Tp, G.as<float4>(), dims, tw_z);
    else
        hipLaunchKernelGGL(k_z_conv<false>, dim3(ztiles), dim3(kThreads), lds_bytes(2 * dims.tl, L), s, Sp, Tp, G.as<float4>(), dims, tw_z);
    MI_TRY(launch_check("k_z_conv"));
    hipLaunchKernelGGL(k_y_pass<true>, dim3(ycols), dim3(kThreads), lds_bytes(dims.tc, M), s, Tp, dims, tw_y);
    MI_TRY(launch_check("k_y_pass<inv>"));
    const dim3 xg((unsigned)((size_t)L * (M / dims.ty)));
    const size_t xl = lds_bytes(dims.ty, Hx);
    switch (epi_kind) {
        case EPI_NONE: case EPI_TAPER_SHELL: hipLaunchKernelGGL(k_x_inverse<EPI_NONE>, xg, dim3(kThreads), xl, s, Tp, out, epi, dims, tw_x); break;
        case EPI_RATIO: hipLaunchKernelGGL(k_x_inverse<EPI_RATIO>, xg, dim3(kThreads), xl, s, Tp, out, epi, dims, tw_x); break;
        case EPI_UPDATE: hipLaunchKernelGGL(k_x_inverse<EPI_UPDATE>, xg, dim3(kThreads), xl, s, Tp, out, epi, dims, tw_x); break;
        case EPI_UPDATE_REG: hipLaunchKernelGGL(k_x_inverse<EPI_UPDATE_REG>, xg, dim3(kThreads), xl, s, Tp, out, epi, dims, tw_x); break;
        default: return fail(MI_ERR_INVALID, "native FFT: unknown epilogue %d", epi_kind);
    }
    return launch_check("k_x_inverse");
}

}  // namespace mi
