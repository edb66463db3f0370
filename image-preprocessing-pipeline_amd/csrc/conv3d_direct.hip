// Direct 3-D convolution, LDS-tiled, fp32 FMA (engine MI_ENGINE_DIRECT).
//
// Replaces conv3d_kernel (LsDeconvolveMultiGPU/conv3d_gpu.cu:68-99, one thread per voxel, every tap a
// global load) and MATLAB convn(...,'same') (decon.m:61,64,70) with one kernel parameterised by the
// boundary rule and a fused epilogue (RL ratio / RL update / edge-taper shell).
//
// Mapping (gfx950): work-group = 32x8 lanes = 4 waves; each lane owns 4 consecutive x outputs of one
// (y, z) row, so a work-group produces a 128 x 8 x 1 tile.  For every kernel plane dz the input plane
// tile (8+ky-1) x (128+kxp) is staged in LDS with the boundary rule applied once; the taps of a row
// are then consumed 4 at a time: one aligned ds_read_b128 slides a 8-float register window and one
// s_load_dwordx4 fetches 4 wave-uniform weights -> 16 v_fma per LDS read, no per-tap global loads.
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
    const int bz = t % gz;
    const int bx = (t / gz) % gx;
    const int by = t / ((unsigned)gz * gx);

    const int tx = threadIdx.x, ty = threadIdx.y;
    const int lane = ty * TX + tx;
    const int pitch = TILE_X + kxp;
    const int rows = TY + ky - 1;
    const int x0 = bx * TILE_X - cx, y0 = by * TY - cy;

    if (EPI == EPI_TAPER_SHELL) {
        // tile entirely on the plateau of the taper mask (mask == 1): nothing to blur (edgetaper_3d.m:44)
        int xa = bx * TILE_X, xb = min(xa + TILE_X, nx), ya = by * TY, yb = min(ya + TY, ny);
        if (xa >= epi.plat_lo[0] && xb <= epi.plat_hi[0] && ya >= epi.plat_lo[1] && yb <= epi.plat_hi[1] &&
            bz >= epi.plat_lo[2] && bz < epi.plat_hi[2])
            return;
    }

    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;

    for (int dz = 0; dz < kz; ++dz) {
        const int gzi = wrap_index(bz + dz - cz, nz, bnd_z);
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
        __syncthreads();
        const float* wplane = kf + (size_t)dz * ky * kxp;
        for (int dy = 0; dy < ky; ++dy) {
            const float* row = tile + (ty + dy) * pitch + tx * RX;
            const float* w = wplane + dy * kxp;
            float4 lo = *reinterpret_cast<const float4*>(row);
            for (int d = 0; d < kxp; d += 4) {
                const float4 hi = *reinterpret_cast<const float4*>(row + d + 4);
                const float4 wv = *reinterpret_cast<const float4*>(w + d);
                acc0 = fmaf(wv.x, lo.x, acc0); acc1 = fmaf(wv.x, lo.y, acc1); acc2 = fmaf(wv.x, lo.z, acc2); acc3 = fmaf(wv.x, lo.w, acc3);
                acc0 = fmaf(wv.y, lo.y, acc0); acc1 = fmaf(wv.y, lo.z, acc1); acc2 = fmaf(wv.y, lo.w, acc2); acc3 = fmaf(wv.y, hi.x, acc3);
                acc0 = fmaf(wv.z, lo.z, acc0); acc1 = fmaf(wv.z, lo.w, acc1); acc2 = fmaf(wv.z, hi.x, acc2); acc3 = fmaf(wv.z, hi.y, acc3);
                acc0 = fmaf(wv.w, lo.w, acc0); acc1 = fmaf(wv.w, hi.x, acc1); acc2 = fmaf(wv.w, hi.y, acc2); acc3 = fmaf(wv.w, hi.z, acc3);
                lo = hi;
            }
        }
    }

    const int y = by * TY + ty, z = bz;
    const int xs = bx * TILE_X + tx * RX;
    if (y >= ny || xs >= nx) return;
    const size_t base = ((size_t)z * ny + y) * nx + xs;
    float r[4] = {acc0, acc1, acc2, acc3};
#pragma unroll
    for (int j = 0; j < RX; ++j) {
        if (xs + j < nx) out[base + j] = epilogue<EPI>(r[j], base + j, xs + j, y, z, epi);
    }
}

}  // namespace

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
    const int gx = (nx + TILE_X - 1) / TILE_X, gy = (ny + TY - 1) / TY, gz = nz;
    const size_t total = (size_t)gx * gy * gz;
    MI_REQUIRE(total < (1ull << 31) - 8, "conv3d: volume too large for one launch");
    const unsigned chunk = (unsigned)((total + 7) / 8);
    const size_t lds = sizeof(float) * (size_t)(TY + ky - 1) * (TILE_X + kxp);
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
