// In-place separable 3-D Gaussian (replaces gauss3d_gpu: LsDeconvolveMultiGPU/gauss3d_gpu.cu:81-204,209-311).
//
// Same arithmetic as the reference: taps exp(-0.5 i^2 / sigma^2) built on the host (double exp, float
// store, double sum), three 1-D passes X, Y, Z with the index clamped to the volume, fp32 accumulate
// in tap order.  Differences in execution: the taps travel in the kernel argument block (wave-uniform
// scalar loads, no __constant__ upload + device sync per axis), and each lane produces 4 consecutive
// x outputs (16-B stores); for the Y/Z passes the 4 outputs share one float4 load per tap.
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

// AXIS 0: along x (scalar path, neighbours of the 4 outputs overlap); AXIS 1/2: along y/z (float4 per tap)
template <int AXIS>
__global__ __launch_bounds__(256) void k_gauss_axis(const float* __restrict__ src, float* __restrict__ dst, int nx, int ny, int nz,
                                                     Taps t) {
    const int nxq = (nx + 3) / 4;
    const size_t total = (size_t)nxq * ny * nz;
    const int r = t.n / 2;
    const bool vec_ok = (nx % 4) == 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int xq = (int)(i % nxq);
        const size_t rest = i / nxq;
        const int y = (int)(rest % ny), z = (int)(rest / ny);
        const int x = xq * 4;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (AXIS == 0) {
            const float* row = src + ((size_t)z * ny + y) * nx;
            for (int s = 0; s < t.n; ++s) {
                const float w = t.w[s];
                const int o = s - r;
                a0 = fmaf(row[min(max(x + o, 0), nx - 1)], w, a0);
                a1 = fmaf(row[min(max(x + 1 + o, 0), nx - 1)], w, a1);
                a2 = fmaf(row[min(max(x + 2 + o, 0), nx - 1)], w, a2);
                a3 = fmaf(row[min(max(x + 3 + o, 0), nx - 1)], w, a3);
            }
        } else {
            for (int s = 0; s < t.n; ++s) {
                const float w = t.w[s];
                const int o = s - r;
                const int yy = AXIS == 1 ? min(max(y + o, 0), ny - 1) : y;
                const int zz = AXIS == 2 ? min(max(z + o, 0), nz - 1) : z;
                const float* p = src + ((size_t)zz * ny + yy) * nx + x;
                if (vec_ok) {
                    const float4 v = *reinterpret_cast<const float4*>(p);
                    a0 = fmaf(v.x, w, a0); a1 = fmaf(v.y, w, a1); a2 = fmaf(v.z, w, a2); a3 = fmaf(v.w, w, a3);
                } else {
                    a0 = fmaf(p[0], w, a0);
                    if (x + 1 < nx) a1 = fmaf(p[1], w, a1);
                    if (x + 2 < nx) a2 = fmaf(p[2], w, a2);
                    if (x + 3 < nx) a3 = fmaf(p[3], w, a3);
                }
            }
        }
        float* q = dst + ((size_t)z * ny + y) * nx + x;
        if (vec_ok) {
            *reinterpret_cast<float4*>(q) = make_float4(a0, a1, a2, a3);
        } else {
            q[0] = a0;
            if (x + 1 < nx) q[1] = a1;
            if (x + 2 < nx) q[2] = a2;
            if (x + 3 < nx) q[3] = a3;
        }
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
    const size_t items = (size_t)((nx + 3) / 4) * ny * nz;
    size_t blocks = (items + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    float* src = vol;
    float* dst = work;
    for (int axis = 0; axis < 3; ++axis) {
        Taps t;
        make_taps(sigma[axis], k[axis], t);
        if (axis == 0) hipLaunchKernelGGL(k_gauss_axis<0>, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, nx, ny, nz, t);
        if (axis == 1) hipLaunchKernelGGL(k_gauss_axis<1>, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, nx, ny, nz, t);
        if (axis == 2) hipLaunchKernelGGL(k_gauss_axis<2>, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, nx, ny, nz, t);
        MI_TRY(launch_check("k_gauss_axis"));
        float* tmp = src; src = dst; dst = tmp;
    }
    // three passes leave the result in `work`; the reference copies it back too (gauss3d_gpu.cu:194-200)
    if (src != vol) MI_HIP(hipMemcpyAsync(vol, src, sizeof(float) * (size_t)nx * ny * nz, hipMemcpyDeviceToDevice, s));
    return MI_OK;
}

}  // namespace mi

extern "C" int mi_gauss3d_inplace(int dev, void* stream, float* vol, float* work, int nx, int ny, int nz, const float* sigma,
                                  const int* ksize) {
    MI_TRY(mi::use_device(dev));
    MI_REQUIRE(sigma, "gauss3d_gpu: Usage: gauss3d_gpu(x, sigma [, kernel_size])");
    return mi::gauss3d_async(mi::as_stream(stream), vol, work, nx, ny, nz, sigma, ksize);
}
