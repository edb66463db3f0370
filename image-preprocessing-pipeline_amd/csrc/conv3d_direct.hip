// Direct 3-D convolution, LDS-tiled, fp32 FMA (engine MI_ENGINE_DIRECT).
//
// Replaces conv3d_kernel (LsDeconvolveMultiGPU/conv3d_gpu.cu:68-99, one thread per voxel, every tap a
// global load) and MATLAB convn(...,'same') (decon.m:61,64,70) with one kernel parameterised by the
// boundary rule and a fused epilogue (RL ratio / RL update / edge-taper shell).
//
// Mapping (gfx950): work-group = 32x8 lanes = 4 waves; each lane owns 4 consecutive x outputs of one
// TWO y rows in TWO consecutive z planes, so a work-group produces a 128 x 16 x 2 tile.  For every input plane
// the tile (8+ky-1) x (128+kxp) is staged in LDS with the boundary rule applied once, together with the
// two kernel planes it meets; the taps of a row are consumed 4 at a time: one aligned ds_read_b128 slides
// an 8-float register window, two broadcast ds_read_b128 fetch the weights -> 32 v_fma per window read.
// The flipped PSF is pre-padded in x to a multiple of 4 with zero weights.  Work-groups are
// renumbered so that groups resident on one XCD walk z first: the kz input planes of a column are
// then shared through that XCD's L2 instead of being re-fetched.
#include "conv3d_direct.h"

namespace mi {
namespace {

constexpr int TX = 32, TY = 8, RX = 4;
constexpr int TILE_X = TX * RX;  // 128

__device__ __forceinline__ int wrap_index(int i, int n, int boundary) {
    if (boundary == MI_BOUNDARY_REPLICATE) return min(max(i, 0), n - 1);
    if (boundary == MI_BOUNDARY_CIRCULAR) {
        i %= n;
        return i < 0 ? i + n : i;
    }
    return (i < 0 || i >= n) ? -1 : i;  // zero
}

// kf[dz][dy][dx] = ker[kz-1-dz][ky-1-dy][kx-1-dx] * scale, x padded with zeros to kxp
__global__ void k_flip_pad_psf(const float* __restrict__ ker, float* __restrict__ kf, int kx, int ky, int kz, int kxp,
                               const float* __restrict__ sum, int normalise, int flip) {
    int total = kxp * ky * kz;
    float scale = 1.0f;
    if (normalise) scale = 1.0f / *sum;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int dx = i % kxp, r = i / kxp;
        int dy = r % ky, dz = r / ky;
        float v = 0.0f;
        if (dx < kx) {
            float k = flip ? ker[((size_t)(kz - 1 - dz) * ky + (ky - 1 - dy)) * kx + (kx - 1 - dx)]
                           : ker[((size_t)dz * ky + dy) * kx + dx];
            // edgetaper_3d.m:14 divides (psf ./ sum) rather than multiplying by a reciprocal
            v = normalise ? k / *sum : k * scale;
        }
        kf[i] = v;
    }
}

__global__ void k_sum_small(const float* __restrict__ x, int n, float* __restrict__ out) {
    // single work-group float sum in index order per lane + tree: matches sum(psf(:)) to ~1 ulp
    __shared__ float part[256];
    float acc = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) acc += x[i];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = part[0];
}

template <int EPI>
__device__ __forceinline__ float epilogue(float c, size_t idx, int x, int y, int z, const ConvEpilogue& e) {
    if (EPI == EPI_RATIO) return e.a[idx] / fmaxf(c, kEpsSingle);
    if (EPI == EPI_UPDATE) return fabsf(e.a[idx] * c);
    if (EPI == EPI_UPDATE_REG) return fabsf(e.a[idx] * c * (1.0f - e.lambda) + e.b[idx] * e.lambda);
    return c;
}

// Register blocking: each lane owns 4 x-outputs of TWO rows (ty, ty + TY) in TWO consecutive z planes = 16
// accumulators.  Every staged input plane feeds both z planes (with kernel planes dz and dz - 1); per 4 taps a lane
// issues 2 window reads + 2 broadcast weight reads (ds_read_b128) for 64 FMAs.  The two kernel planes of the current
// step sit in LDS next to the image tile, so the inner loop has no scalar-load latency; its LDS reads are issued one
// step ahead of the FMAs that use them.
constexpr int TZ = 2;
constexpr int RY = 2;
constexpr int TILE_Y = TY * RY;  // 16

#define MI_FMA16(U, W, LO, HI, A0, A1, A2, A3, B0, B1, B2, B3)                                                             \
    A0 = fmaf(U.x, LO.x, A0); A1 = fmaf(U.x, LO.y, A1); A2 = fmaf(U.x, LO.z, A2); A3 = fmaf(U.x, LO.w, A3);              \
    B0 = fmaf(W.x, LO.x, B0); B1 = fmaf(W.x, LO.y, B1); B2 = fmaf(W.x, LO.z, B2); B3 = fmaf(W.x, LO.w, B3);              \
    A0 = fmaf(U.y, LO.y, A0); A1 = fmaf(U.y, LO.z, A1); A2 = fmaf(U.y, LO.w, A2); A3 = fmaf(U.y, HI.x, A3);              \
    B0 = fmaf(W.y, LO.y, B0); B1 = fmaf(W.y, LO.z, B1); B2 = fmaf(W.y, LO.w, B2); B3 = fmaf(W.y, HI.x, B3);              \
    A0 = fmaf(U.z, LO.z, A0); A1 = fmaf(U.z, LO.w, A1); A2 = fmaf(U.z, HI.x, A2); A3 = fmaf(U.z, HI.y, A3);              \
    B0 = fmaf(W.z, LO.z, B0); B1 = fmaf(W.z, LO.w, B1); B2 = fmaf(W.z, HI.x, B2); B3 = fmaf(W.z, HI.y, B3);              \
    A0 = fmaf(U.w, LO.w, A0); A1 = fmaf(U.w, HI.x, A1); A2 = fmaf(U.w, HI.y, A2); A3 = fmaf(U.w, HI.z, A3);              \
    B0 = fmaf(W.w, LO.w, B0); B1 = fmaf(W.w, HI.x, B1); B2 = fmaf(W.w, HI.y, B2); B3 = fmaf(W.w, HI.z, B3);

template <int EPI>
__global__ __launch_bounds__(TX* TY) void k_conv3d_direct(const float* __restrict__ img, const float* __restrict__ kf,
                                                           float* __restrict__ out, ConvEpilogue epi, int nx, int ny, int nz,
                                                           int kx, int ky, int kz, int kxp, int cx, int cy, int cz, int bnd_x,
                                                           int bnd_y, int bnd_z, int gx, int gy, int gz) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    // XCD-aware renumbering: ids b, b+8, b+16.. share an XCD; give each XCD one contiguous range of
    // tiles and walk z fastest inside it.
    const unsigned total = (unsigned)gx * gy * gz;
    const unsigned chunk = (total + 7u) / 8u;
    const unsigned t = (blockIdx.x % 8u) * chunk + blockIdx.x / 8u;
    if (blockIdx.x / 8u >= chunk || t >= total) return;
    const int bz = (t % gz) * TZ;
    const int bx = (t / gz) % gx;
    const int by = t / ((unsigned)gz * gx);

    const int tx = threadIdx.x, ty = threadIdx.y;
    const int lane = ty * TX + tx;
    const int pitch = TILE_X + kxp;
    const int rows = TILE_Y + ky - 1;
    const int x0 = bx * TILE_X - cx, y0 = by * TILE_Y - cy;
    float* wA = tile + rows * pitch;  // kernel plane for output plane bz     (dz = step)
    float* wB = wA + ky * kxp;        // kernel plane for output plane bz + 1 (dz = step - 1)

    if (EPI == EPI_TAPER_SHELL) {
        // tile entirely on the plateau of the taper mask (mask == 1): nothing to blur (edgetaper_3d.m:44)
        int xa = bx * TILE_X, xb = min(xa + TILE_X, nx), ya = by * TILE_Y, yb = min(ya + TILE_Y, ny);
        if (xa >= epi.plat_lo[0] && xb <= epi.plat_hi[0] && ya >= epi.plat_lo[1] && yb <= epi.plat_hi[1] &&
            bz >= epi.plat_lo[2] && min(bz + TZ, nz) <= epi.plat_hi[2])
            return;
    }

    // a*: plane bz, b*: plane bz + 1; suffix 0..3: row ty, 4..7: row ty + TY
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    float a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f, b4 = 0.f, b5 = 0.f, b6 = 0.f, b7 = 0.f;
    const bool two = bz + 1 < nz;

    // input plane of step s: z_in = bz + s - cz; it meets output bz with dz = s and output bz + 1 with dz = s - 1
    for (int st = 0; st < kz + 1; ++st) {
        const bool useA = st < kz, useB = two && st >= 1;
        if (!useA && !useB) continue;
        const int gzi = wrap_index(bz + st - cz, nz, bnd_z);
        if (gzi < 0) continue;  // zero boundary: whole plane is zero (uniform branch)
        const float* plane = img + (size_t)gzi * ny * nx;
        __syncthreads();
        for (int i = lane; i < rows * pitch; i += TX * TY) {
            int r = i / pitch, c = i - r * pitch;
            float v = 0.0f;
            if (c < TILE_X + kx - 1) {
                int gyi = wrap_index(y0 + r, ny, bnd_y);
                int gxi = wrap_index(x0 + c, nx, bnd_x);
                if (gyi >= 0 && gxi >= 0) v = plane[(size_t)gyi * nx + gxi];
            }
            tile[i] = v;
        }
        {
            const float* srcA = kf + (size_t)(useA ? st : 0) * ky * kxp;
            const float* srcB = kf + (size_t)(useB ? st - 1 : 0) * ky * kxp;
            for (int i = lane; i < ky * kxp; i += TX * TY) {
                wA[i] = useA ? srcA[i] : 0.0f;
                wB[i] = useB ? srcB[i] : 0.0f;
            }
        }
        __syncthreads();
        for (int dy = 0; dy < ky; ++dy) {
            const float* r0 = tile + (ty + dy) * pitch + tx * RX;
            const float* r1 = r0 + TY * pitch;
            const float* wa = wA + dy * kxp;
            const float* wb = wB + dy * kxp;
            float4 lo0 = *reinterpret_cast<const float4*>(r0), hi0 = *reinterpret_cast<const float4*>(r0 + 4);
            float4 lo1 = *reinterpret_cast<const float4*>(r1), hi1 = *reinterpret_cast<const float4*>(r1 + 4);
            float4 u = *reinterpret_cast<const float4*>(wa);  // broadcast reads
            float4 w = *reinterpret_cast<const float4*>(wb);
            for (int d = 0; d < kxp; d += 4) {
                float4 hn0 = hi0, hn1 = hi1, u_n = u, w_n = w;
                if (d + 4 < kxp) {
                    hn0 = *reinterpret_cast<const float4*>(r0 + d + 8);
                    hn1 = *reinterpret_cast<const float4*>(r1 + d + 8);
                    u_n = *reinterpret_cast<const float4*>(wa + d + 4);
                    w_n = *reinterpret_cast<const float4*>(wb + d + 4);
                }
                MI_FMA16(u, w, lo0, hi0, a0, a1, a2, a3, b0, b1, b2, b3)
                MI_FMA16(u, w, lo1, hi1, a4, a5, a6, a7, b4, b5, b6, b7)
                lo0 = hi0; hi0 = hn0; lo1 = hi1; hi1 = hn1; u = u_n; w = w_n;
            }
        }
    }

    const int xs = bx * TILE_X + tx * RX;
    if (xs >= nx) return;
    const float ra[8] = {a0, a1, a2, a3, a4, a5, a6, a7}, rb[8] = {b0, b1, b2, b3, b4, b5, b6, b7};
#pragma unroll
    for (int p = 0; p < TZ; ++p) {
        const int z = bz + p;
        if (z >= nz) break;
#pragma unroll
        for (int q = 0; q < RY; ++q) {
            const int y = by * TILE_Y + ty + q * TY;
            if (y >= ny) continue;
            const size_t base = ((size_t)z * ny + y) * nx + xs;
#pragma unroll
            for (int j = 0; j < RX; ++j) {
                if (xs + j < nx) out[base + j] = epilogue<EPI>(p == 0 ? ra[4 * q + j] : rb[4 * q + j], base + j, xs + j, y, z, epi);
            }
        }
    }
}
#undef MI_FMA16

}  // namespace

// out[i] = ker[i] / sum(ker)  (edgetaper_3d.m:14), for engines that take the PSF as is
__global__ void k_normalise_psf(const float* __restrict__ ker, const float* __restrict__ sum, float* __restrict__ out, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = ker[i] / *sum;
}

int normalised_psf(hipStream_t s, const float* ker, int n, DevBuf& out) {
    MI_TRY(out.alloc(sizeof(float) * ((size_t)n + 4)));
    float* sum = out.as<float>() + n;
    hipLaunchKernelGGL(k_sum_small, dim3(1), dim3(256), 0, s, ker, n, sum);
    MI_TRY(launch_check("k_sum_small"));
    hipLaunchKernelGGL(k_normalise_psf, dim3(cdiv((size_t)n, 256)), dim3(256), 0, s, ker, sum, out.as<float>(), n);
    return launch_check("k_normalise_psf");
}

int conv_kernel_offset(int k, int boundary) {
    // conv3d_gpu centres at k/2 (conv3d_gpu.cu:77); convn 'same' keeps full[floor(k/2) : ...], i.e. the
    // window starts k-1-floor(k/2) before the output sample.  Identical for odd k.
    return boundary == MI_BOUNDARY_REPLICATE ? k / 2 : k - 1 - k / 2;
}

int direct_prepare_psf(hipStream_t s, const float* ker, int kx, int ky, int kz, bool normalise, bool flip, DevBuf& kf,
                       int* kxp_out) {
    const int kxp = (kx + 3) / 4 * 4;
    MI_TRY(kf.alloc(sizeof(float) * ((size_t)kxp * ky * kz + 4)));
    float* sum = kf.as<float>() + (size_t)kxp * ky * kz;
    if (normalise) {
        hipLaunchKernelGGL(k_sum_small, dim3(1), dim3(256), 0, s, ker, kx * ky * kz, sum);
        MI_TRY(launch_check("k_sum_small"));
    }
    hipLaunchKernelGGL(k_flip_pad_psf, dim3(cdiv((size_t)kxp * ky * kz, 256)), dim3(256), 0, s, ker, kf.as<float>(), kx, ky, kz,
                       kxp, sum, normalise ? 1 : 0, flip ? 1 : 0);
    MI_TRY(launch_check("k_flip_pad_psf"));
    *kxp_out = kxp;
    return MI_OK;
}

int direct_conv_launch(hipStream_t s, const float* img, const float* kf, float* out, int nx, int ny, int nz, int kx, int ky,
                       int kz, int kxp, int boundary, int epi_kind, const ConvEpilogue& epi, const int* offs, const int* bnd3) {
    const int gx = (nx + TILE_X - 1) / TILE_X, gy = (ny + TILE_Y - 1) / TILE_Y, gz = (nz + TZ - 1) / TZ;
    const size_t total = (size_t)gx * gy * gz;
    MI_REQUIRE(total < (1ull << 31) - 8, "conv3d: volume too large for one launch");
    const unsigned chunk = (unsigned)((total + 7) / 8);
    const size_t lds = sizeof(float) * ((size_t)(TILE_Y + ky - 1) * (TILE_X + kxp) + 2 * (size_t)ky * kxp);
    MI_REQUIRE(lds <= 160 * 1024, "conv3d: kernel %dx%d too large for the direct engine (LDS %zu B)", kx, ky, lds);
    const int cx = offs ? offs[0] : conv_kernel_offset(kx, boundary);
    const int cy = offs ? offs[1] : conv_kernel_offset(ky, boundary);
    const int cz = offs ? offs[2] : conv_kernel_offset(kz, boundary);
    const int bnd_x = bnd3 ? bnd3[0] : boundary, bnd_y = bnd3 ? bnd3[1] : boundary, bnd_z = bnd3 ? bnd3[2] : boundary;
    dim3 grid(chunk * 8), block(TX, TY);
#define MI_LAUNCH_CONV(E)                                                                                                    \
    do {                                                                                                                     \
        if (lds > 64 * 1024)                                                                                                 \
            MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3d_direct<E>),                                   \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                               \
        hipLaunchKernelGGL(k_conv3d_direct<E>, grid, block, lds, s, img, kf, out, epi, nx, ny, nz, kx, ky, kz, kxp, cx, cy, \
                           cz, bnd_x, bnd_y, bnd_z, gx, gy, gz);                                                                        \
    } while (0)
    switch (epi_kind) {
        case EPI_NONE: MI_LAUNCH_CONV(EPI_NONE); break;
        case EPI_RATIO: MI_LAUNCH_CONV(EPI_RATIO); break;
        case EPI_UPDATE: MI_LAUNCH_CONV(EPI_UPDATE); break;
        case EPI_UPDATE_REG: MI_LAUNCH_CONV(EPI_UPDATE_REG); break;
        case EPI_TAPER_SHELL: MI_LAUNCH_CONV(EPI_TAPER_SHELL); break;
        default: return fail(MI_ERR_INVALID, "conv3d: unknown epilogue %d", epi_kind);
    }
#undef MI_LAUNCH_CONV
    return launch_check("k_conv3d_direct");
}

}  // namespace mi
