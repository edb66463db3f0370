// Peak rate of v_fma_f64 on this device: 16 independent chains per lane, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 profiles/micro/fma64_peak.hip -o /tmp/fma64_peak && /tmp/fma64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_fma64(double* out, int iters, double a, double b) {
    double acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = threadIdx.x * 1e-9 + q;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = fma(acc[q], a, b);
    }
    double s = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += acc[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_fma32(float* out, int iters, float a, float b) {
    float acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = threadIdx.x * 1e-9f + q;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = fmaf(acc[q], a, b);
    }
    float s = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += acc[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    const int blocks = 256 * 8, iters = 20000;
    double* d;
    hipMalloc(&d, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wpb = 0; wpb < 2; ++wpb) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (wpb == 0) hipLaunchKernelGGL(k_fma64, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999999, 1e-7);
            else hipLaunchKernelGGL(k_fma32, dim3(blocks), dim3(256), 0, 0, (float*)d, iters, 0.999999f, 1e-7f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double fl = 2.0 * 16 * iters * (double)blocks * 256;
            printf("%s: %.3f ms, %.1f TFLOP/s\n", wpb == 0 ? "v_fma_f64" : "v_fma_f32", ms, fl / ms / 1e9);
        }
    }
    return 0;
}
