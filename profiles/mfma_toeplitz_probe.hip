// Measured prototype for the "MFMA implicit-GEMM direct engine" row: a zero-boundary 'same' 3-D convolution
// (convn, decon.m:61; conv3d_gpu.cu:68-99 is its replicate-boundary twin) with the taps applied through fp32 matrix cores.
//
// Formulation (banded Toeplitz along x): for a fixed (dz, dy) the 16 x 16 output tile D[y][x] gains  A . B  with
//   A[m][j] = in[z + dz - cz][y0 + m + dy - cy][x0 + j - cx],   B[j][n] = tap[dz][dy][j - n]  (0 outside [0, kx)),
// j = 0 .. 16 + kx - 2, taken four at a time by v_mfma_f32_16x16x4_f32 (exact fp32, 32 cycles per instruction and SIMD).  Only kx of
// the 16 + kx - 1 products of a column are non-zero: for kx = 15 the matrix cores spend 8 steps = 32 K-values on 15 useful ones (47 %).
// A work-group = 4 waves = a 64 x 64 (x, y) output tile of one z plane: wave w owns rows 16 w .. 16 w + 15 and four x tiles; the
// input plane tile and the zero-padded tap rows of the current dz are staged in LDS; per 4 MFMAs a lane reads 1 B value + 4 A values.
//
//   hipcc -O3 --offload-arch=gfx950 -Iinclude profiles/mfma_toeplitz_probe.hip -Limage-preprocessing-pipeline_amd -lmi_ipp \
//         -Wl,-rpath,$PWD/image-preprocessing-pipeline_amd -o /tmp/mfma_probe && /tmp/mfma_probe
// prints the time of this kernel and of the library's VALU direct engine (mi_conv3d, engine = direct) on the same volume and PSF
// (BASELINE config 2: 1024 x 1024 x 256, PSF 15 x 15 x 31), and the largest difference between the two results.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mi_lsdeconv.h"

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BX = 64, BY = 64, XT = BX / 16;

__global__ __launch_bounds__(256) void k_conv_mfma(const float* __restrict__ in, const float* __restrict__ taps /*[kz][ky][kx] correlation order*/,
                                                   float* __restrict__ out, int nx, int ny, int nz, int kx, int ky, int kz, int cx, int cy,
                                                   int cz, int pitch, int tpitch) {
    extern __shared__ float lds[];
    float* tile = lds;                                 // [(BY + ky - 1)][pitch]: columns x0 - cx ...
    float* trow = lds + (BY + ky - 1) * pitch;         // [ky][tpitch]: tap index i at trow[15 + i], zeros around
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = lane & 15, k = lane >> 4;
    const int x0 = blockIdx.x * BX, y0 = blockIdx.y * BY, z = blockIdx.z;
    const int S = (16 + kx - 1 + 3) / 4;
    f32x4 acc[XT];
    for (int t = 0; t < XT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int rows = BY + ky - 1, cols = BX + kx - 1 + 3;
    for (int dz = 0; dz < kz; ++dz) {
        __syncthreads();
        const int zi = z + dz - cz;
        for (int e = threadIdx.x; e < rows * cols; e += 256) {
            const int r = e / cols, c = e - r * cols;
            const int yi = y0 + r - cy, xi = x0 + c - cx;
            tile[r * pitch + c] = (zi >= 0 && zi < nz && yi >= 0 && yi < ny && xi >= 0 && xi < nx) ? in[((size_t)zi * ny + yi) * nx + xi] : 0.0f;
        }
        for (int e = threadIdx.x; e < ky * tpitch; e += 256) {
            const int dy = e / tpitch, i = e - dy * tpitch - 15;
            trow[e] = (i >= 0 && i < kx) ? taps[((size_t)dz * ky + dy) * kx + i] : 0.0f;
        }
        __syncthreads();
        for (int dy = 0; dy < ky; ++dy) {
            const float* arow = tile + (wave * 16 + m + dy) * pitch + k;
            const float* brow = trow + dy * tpitch + 15 + k - m;     // B[j][n] = tap[j - n], j = 4 s + k, n = lane & 15
            for (int s = 0; s < S; ++s) {
                const float b = brow[4 * s];
#pragma unroll
                for (int t = 0; t < XT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[16 * t + 4 * s], b, acc[t], 0, 0, 0);
            }
        }
    }
    // C/D: col = lane & 15 (x), row = (lane >> 4) * 4 + reg (y)
    for (int t = 0; t < XT; ++t)
        for (int r = 0; r < 4; ++r) {
            const int y = y0 + wave * 16 + (lane >> 4) * 4 + r, x = x0 + 16 * t + (lane & 15);
            if (y < ny && x < nx) out[((size_t)z * ny + y) * nx + x] = acc[t][r];
        }
}

int main(int argc, char** argv) {
    int nx = 1024, ny = 1024, nz = 256, kx = 15, ky = 15, kz = 31;
    if (argc > 3) { nx = atoi(argv[1]); ny = atoi(argv[2]); nz = atoi(argv[3]); }
    const size_t N = (size_t)nx * ny * nz, K = (size_t)kx * ky * kz;
    std::vector<float> hin(N), hk(K), hcorr(K);
    srand(7);
    for (auto& v : hin) v = (float)rand() / RAND_MAX;
    for (auto& v : hk) v = (float)rand() / RAND_MAX / K;
    // convolution kernel -> correlation taps (flip in all axes); convn 'same': window starts k - 1 - k/2 before the output sample
    for (int z = 0; z < kz; ++z)
        for (int y = 0; y < ky; ++y)
            for (int x = 0; x < kx; ++x) hcorr[((size_t)z * ky + y) * kx + x] = hk[((size_t)(kz - 1 - z) * ky + (ky - 1 - y)) * kx + (kx - 1 - x)];
    float *din, *dk, *dcorr, *o1, *o2;
    CK(hipMalloc(&din, 4 * N)); CK(hipMalloc(&o1, 4 * N)); CK(hipMalloc(&o2, 4 * N)); CK(hipMalloc(&dk, 4 * K)); CK(hipMalloc(&dcorr, 4 * K));
    CK(hipMemcpy(din, hin.data(), 4 * N, hipMemcpyHostToDevice));
    CK(hipMemcpy(dk, hk.data(), 4 * K, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcorr, hcorr.data(), 4 * K, hipMemcpyHostToDevice));
    const int cx = kx - 1 - kx / 2, cy = ky - 1 - ky / 2, cz = kz - 1 - kz / 2;
    const int pitch = (BX + kx - 1 + 3) | 1, tpitch = 15 + kx + 16 + 8;
    const size_t lds = 4 * ((size_t)(BY + ky - 1) * pitch + (size_t)ky * tpitch);
    dim3 grid((nx + BX - 1) / BX, (ny + BY - 1) / BY, nz);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms_mfma = 0, ms_valu = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_conv_mfma, grid, dim3(256), lds, 0, din, dcorr, o1, nx, ny, nz, kx, ky, kz, cx, cy, cz, pitch, tpitch);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_mfma, e0, e1));
        CK(hipEventRecord(e0));
        if (mi_conv3d(0, nullptr, din, dk, o2, nx, ny, nz, kx, ky, kz, MI_BOUNDARY_ZERO, MI_ENGINE_DIRECT) != 0) { printf("mi_conv3d: %s\n", mi_last_error()); return 1; }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_valu, e0, e1));
    }
    std::vector<float> a(N), b(N);
    CK(hipMemcpy(a.data(), o1, 4 * N, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), o2, 4 * N, hipMemcpyDeviceToHost));
    double worst = 0, top = 0;
    for (size_t i = 0; i < N; ++i) { worst = std::max(worst, (double)std::fabs(a[i] - b[i])); top = std::max(top, (double)std::fabs(b[i])); }
    const double flop = 2.0 * (double)N * (double)K;
    printf("volume %d x %d x %d, PSF %d x %d x %d (%zu taps), zero boundary\n", nx, ny, nz, kx, ky, kz, K);
    printf("banded-Toeplitz v_mfma_f32_16x16x4_f32 kernel: %.2f ms  (%.1f useful TFLOP/s)\n", ms_mfma, flop / ms_mfma / 1e9);
    printf("library direct engine (VALU, LDS-tiled)      : %.2f ms  (%.1f TFLOP/s)\n", ms_valu, flop / ms_valu / 1e9);
    printf("max |difference| = %.3e of max |result| %.3e\n", worst, top);
    return 0;
}
