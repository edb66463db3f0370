// In-place separable 3-D Gaussian (replaces gauss3d_gpu: LsDeconvolveMultiGPU/gauss3d_gpu.cu:81-204,209-311).
//
// Same arithmetic as the reference: taps exp(-0.5 i^2 / sigma^2) built on the host (double exp, float
// store, double sum), filtered X, then Y, then Z with the index clamped to the volume, fp32 accumulate
// in tap order, every 1-D result rounded to fp32.  Execution: TWO volume passes instead of three plus a
// copy (the reference moves 32 B/voxel, this 16 B/voxel): x and y are fused in a kernel whose waves walk
// along y with an LDS ring of x-filtered rows; z is a second walk with a per-lane ring; the result
// lands back in `vol`, so no device-to-device copy is needed.  Taps travel in the kernel argument block.
#include <cmath>

#include "mi_internal.h"
#include "mi_lsdeconv.h"

namespace mi {
namespace {

constexpr int kMaxTaps = 51;  // MAX_KERNEL_SIZE, gauss3d_gpu.cu:77
struct Taps {
    float w[kMaxTaps];
    int n;
};

// make_gaussian_kernel, gauss3d_gpu.cu:81-90
void make_taps(float sigma, int ksize, Taps& t) {
    int r = ksize / 2;
    double sum = 0.0;
    float s2 = sigma * sigma;
    for (int i = -r; i <= r; ++i) {
        t.w[i + r] = static_cast<float>(std::exp(-0.5 * (i * i) / s2));
        sum += t.w[i + r];
    }
    for (int i = 0; i < ksize; ++i) t.w[i] = static_cast<float>(t.w[i] / sum);
    t.n = ksize;
}

// Pass 1 (x and y fused): a wave owns 64 consecutive x of one z-plane and walks along y.  Per step it stages one
// clamped row segment (64 + 2 rx samples) in LDS, filters it along x (the result is rounded to fp32 exactly like the
// reference's separate x pass), pushes it into a per-lane ring of the last ky x-filtered rows and emits one y-filtered
// row.  Every input row is read once and every output row written once (plus ry halo rows per y-chunk).
constexpr int GXY_WAVES = 4;
constexpr int GXY_YCHUNK = 256;

__global__ __launch_bounds__(64 * GXY_WAVES) void k_gauss_xy(const float* __restrict__ src, float* __restrict__ dst, int nx, int ny, int nz,
                                                              Taps tx, Taps ty) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rx = tx.n / 2, ry = ty.n / 2;
    const int seg = 64 + 2 * rx;
    float* rb = lds + wave * (seg + ty.n * 64);  // row buffer, then ring[ty.n][64]
    float* ring = rb + seg;
    const int x0 = blockIdx.x * 64;
    const int ya = blockIdx.y * GXY_YCHUNK, yb = min(ya + GXY_YCHUNK, ny);
    const int z = blockIdx.z * GXY_WAVES + wave;
    const bool zlive = z < nz;
    const float* plane = src + (size_t)(zlive ? z : 0) * ny * nx;
    float* oplane = dst + (size_t)(zlive ? z : 0) * ny * nx;
    int slot = 0;  // ring slot of walk position p
    for (int p = ya - ry; p < yb + ry; ++p) {
        const float* row = plane + (size_t)min(max(p, 0), ny - 1) * nx;
        // stage the clamped row segment: sample i is x = x0 - rx + i
        rb[lane] = row[min(max(x0 - rx + lane, 0), nx - 1)];
        if (lane + 64 < seg) rb[lane + 64] = row[min(max(x0 - rx + lane + 64, 0), nx - 1)];
        __syncthreads();
        float xf = 0.0f;
        for (int s = 0; s < tx.n; ++s) xf = fmaf(rb[lane + s], tx.w[s], xf);
        ring[slot * 64 + lane] = xf;  // private column of this lane: no barrier needed for the ring
        const int yo = p - ry;        // output row whose window [yo - ry, yo + ry] is now complete
        if (yo >= ya) {
            float acc = 0.0f;
            int rs = slot + 1;  // slot of walk position yo - ry = p - 2 ry  (ty.n = 2 ry + 1 slots back, wrapping)
            if (rs >= ty.n) rs -= ty.n;
            for (int s = 0; s < ty.n; ++s) {
                acc = fmaf(ring[rs * 64 + lane], ty.w[s], acc);
                if (++rs >= ty.n) rs = 0;
            }
            if (zlive && x0 + lane < nx) oplane[(size_t)yo * nx + x0 + lane] = acc;
        }
        if (++slot >= ty.n) slot = 0;
        __syncthreads();  // the row buffer is rewritten in the next step
    }
}

// Pass 2 (z): a lane owns one (x, y) column and walks along z with a private LDS ring of the last kz samples.
constexpr int GZ_ZCHUNK = 512;
__global__ __launch_bounds__(256) void k_gauss_z(const float* __restrict__ src, float* __restrict__ dst, int nx, int ny, int nz, Taps tz) {
    extern __shared__ float lds[];
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= nx) return;
    const int rz = tz.n / 2;
    const int za = blockIdx.z * GZ_ZCHUNK, zb = min(za + GZ_ZCHUNK, nz);
    const size_t col = (size_t)y * nx + x, pstride = (size_t)ny * nx;
    float* ring = lds + threadIdx.x;  // ring[slot * 256]
    int slot = 0;
    for (int p = za - rz; p < zb + rz; ++p) {
        ring[slot * 256] = src[col + (size_t)min(max(p, 0), nz - 1) * pstride];
        const int zo = p - rz;
        if (zo >= za) {
            float acc = 0.0f;
            int rs = slot + 1;
            if (rs >= tz.n) rs -= tz.n;
            for (int s = 0; s < tz.n; ++s) {
                acc = fmaf(ring[rs * 256], tz.w[s], acc);
                if (++rs >= tz.n) rs = 0;
            }
            dst[col + (size_t)zo * pstride] = acc;
        }
        if (++slot >= tz.n) slot = 0;
    }
}

}  // namespace

int gauss3d_async(hipStream_t s, float* vol, float* work, int nx, int ny, int nz, const float* sigma, const int* ksize) {
    MI_REQUIRE(vol && work && vol != work, "gauss3d_gpu: null or aliased buffers");
    MI_REQUIRE(nx > 0 && ny > 0 && nz > 0, "gauss3d_gpu: Input must be 3D.");
    MI_REQUIRE(((uintptr_t)vol % 16) == 0 && ((uintptr_t)work % 16) == 0, "gauss3d_gpu: buffers must be 16-byte aligned");
    int k[3];
    for (int a = 0; a < 3; ++a) {
        MI_REQUIRE(sigma[a] > 0.0f, "gauss3d_gpu: sigma must be positive");
        k[a] = ksize ? ksize[a] : 2 * (int)std::ceil(3.0 * (double)sigma[a]) + 1;  // gauss3d_gpu.cu:244-261
        MI_REQUIRE(k[a] >= 1 && k[a] <= kMaxTaps, "gauss3d_gpu: Kernel size exceeds MAX_KERNEL_SIZE (%d)", kMaxTaps);
    }
    Taps tx, ty, tz;
    make_taps(sigma[0], k[0], tx);
    make_taps(sigma[1], k[1], ty);
    make_taps(sigma[2], k[2], tz);
    // pass 1: vol -> work (x then y, each rounded to fp32 like the reference's separate passes); pass 2: work -> vol (z)
    const size_t lds_xy = sizeof(float) * GXY_WAVES * (size_t)(64 + 2 * (k[0] / 2) + k[1] * 64);
    hipLaunchKernelGGL(k_gauss_xy, dim3((nx + 63) / 64, (ny + GXY_YCHUNK - 1) / GXY_YCHUNK, (nz + GXY_WAVES - 1) / GXY_WAVES),
                       dim3(64 * GXY_WAVES), lds_xy, s, vol, work, nx, ny, nz, tx, ty);
    MI_TRY(launch_check("k_gauss_xy"));
    const size_t lds_z = sizeof(float) * 256 * (size_t)k[2];
    hipLaunchKernelGGL(k_gauss_z, dim3((nx + 255) / 256, ny, (nz + GZ_ZCHUNK - 1) / GZ_ZCHUNK), dim3(256), lds_z, s, work, vol, nx, ny, nz, tz);
    MI_TRY(launch_check("k_gauss_z"));
    return MI_OK;
}

}  // namespace mi

extern "C" int mi_gauss3d_inplace(int dev, void* stream, float* vol, float* work, int nx, int ny, int nz, const float* sigma,
                                  const int* ksize) {
    MI_TRY(mi::use_device(dev));
    MI_REQUIRE(sigma, "gauss3d_gpu: Usage: gauss3d_gpu(x, sigma [, kernel_size])");
    return mi::gauss3d_async(mi::as_stream(stream), vol, work, nx, ny, nz, sigma, ksize);
}
