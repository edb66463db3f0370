// What the HBM system gives for the access pattern of the strided FFT passes, without any of their arithmetic: work-groups
// (256 lanes, 8 per CU) copy tiles of NZ segments of SEG bytes that lie PITCH bytes apart, from `nstreams` places per tile
// (the z pass: a line tile and its mirror partner).   build: hipcc --offload-arch=gfx950 -O3 -o profiles/build/strided_copy_probe profiles/strided_copy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NT = 256;
// one work-group copies a quarter (NZ / 4 segments per stream) of a tile; 8 work-groups per CU overlap loads and stores
template <int SEG, int NSTREAM, int NZ>
__global__ __launch_bounds__(NT) void k_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t plane_f4, size_t pitch_f4,
                                             int tiles_per_plane, int nplanes, int with_g, const float4* __restrict__ g, float* sink) {
    constexpr int LPS = SEG / 16, ZPI = NT / LPS, K = NZ / 4 / ZPI;
    const int t = blockIdx.x >> 2, zq = blockIdx.x & 3;
    const int z0 = zq * (NZ / 4) + threadIdx.x / LPS, l = threadIdx.x % LPS;
    const int plane = t / tiles_per_plane, yt = t - plane * tiles_per_plane;
    float4 v[NSTREAM][K];
    float acc = 0.0f;
#pragma unroll
    for (int st = 0; st < NSTREAM; ++st) {
        const int p = st == 0 ? plane : (nplanes - 1 - plane), y = st == 0 ? yt : (tiles_per_plane - 1 - yt);
        const float4* s = src + (size_t)p * plane_f4 + (size_t)y * LPS + (size_t)z0 * pitch_f4 + l;
#pragma unroll
        for (int k = 0; k < K; ++k) v[st][k] = s[(size_t)k * ZPI * pitch_f4];
    }
    if (with_g) {  // the OTF stream: contiguous, NSTREAM * NZ * SEG / 2 bytes per tile
        const float4* gp = g + ((size_t)blockIdx.x * (NSTREAM * NZ * LPS / 8)) + threadIdx.x;
#pragma unroll
        for (int k = 0; k < NSTREAM * NZ * LPS / 8 / NT; ++k) { const float4 q = gp[k * NT]; acc += q.x + q.w; }
    }
#pragma unroll
    for (int st = 0; st < NSTREAM; ++st) {
        const int p = st == 0 ? plane : (nplanes - 1 - plane), y = st == 0 ? yt : (tiles_per_plane - 1 - yt);
        float4* d = dst + (size_t)p * plane_f4 + (size_t)y * LPS + (size_t)z0 * pitch_f4 + l;
#pragma unroll
        for (int k = 0; k < K; ++k) d[(size_t)k * ZPI * pitch_f4] = v[st][k];
    }
    if (acc == 1.2345f) *sink = acc;
}

template <int SEG, int NSTREAM>
void run(const char* what, float4* a, float4* b, float4* g, float* sink, size_t total_bytes, int with_g, int pad) {
    constexpr int NZ = 512;
    const size_t pitch = 16384 + (size_t)pad;         // bytes between the segments of a tile (the row of 2048 complex samples + padding)
    const size_t plane = pitch * NZ;                  // one px plane
    const int nplanes = (int)(total_bytes / plane) - 1;
    const int tiles_per_plane = (int)(16384 / SEG);
    const int ntiles = nplanes * tiles_per_plane / NSTREAM;  // NSTREAM streams cover the planes from both ends
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_copy<SEG, NSTREAM, NZ>), dim3(ntiles * 4), dim3(NT), 0, 0, a, b, plane / 16, pitch / 16, tiles_per_plane, nplanes,
                           with_g, g, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = 2.0 * total_bytes + (with_g ? total_bytes / 2.0 : 0.0);
    printf("%-44s pad %4d  %6.3f ms  %5.2f TB/s\n", what, pad, best, bytes * (double)nplanes / (double)(nplanes + 1) / best / 1e9);
    fflush(stdout);
}

// read-only stream (what a reduction like the MIP pass of the NCC path can reach): UNR float4 per lane in flight
template <int UNR>
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ src, size_t n4, float* sink) {
    float acc = 0.0f;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (UNR - 1) * stride < n4; i += UNR * stride) {
        float4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc = fmaxf(acc, fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)));
    }
    if (acc == 1.2345f) *sink = acc;
}

// the same with 4-byte loads (one float per lane and instruction, as a kernel that keeps a column per lane does)
template <int UNR>
__global__ __launch_bounds__(256) void k_read32(const float* __restrict__ src, size_t n, float* sink) {
    float acc = 0.0f;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (UNR - 1) * stride < n; i += UNR * stride) {
        float v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc = fmaxf(acc, v[u]);
    }
    if (acc == 1.2345f) *sink = acc;
}

template <int UNR>
void run_read32(const float4* a, size_t bytes, int grid, float* sink) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_read32<UNR>), dim3(grid), dim3(256), 0, 0, reinterpret_cast<const float*>(a), bytes / 4, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    printf("read only, %d floats (4-byte loads) per lane in flight, grid %5d: %6.3f ms  %5.2f TB/s\n", UNR, grid, best, (double)bytes / best / 1e9);
    fflush(stdout);
}

template <int UNR>
void run_read(const float4* a, size_t bytes, int grid, float* sink) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_read<UNR>), dim3(grid), dim3(256), 0, 0, a, bytes / 16, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    printf("read only, %d float4 per lane in flight, grid %5d: %6.3f ms  %5.2f TB/s\n", UNR, grid, best, (double)bytes / best / 1e9);
    fflush(stdout);
}

// read-only, the shape of the MIP pass: a wave reads PIECE bytes of each of 16 rows (pitch 8 KB) of one slice (slices 16 MB apart,
// 4 waves of a work-group on 4 slices, 8 slices per wave); work-groups tile the columns and the row bands of a 307-row strip
template <int PIECE>
__global__ __launch_bounds__(256) void k_read_rows(const float* __restrict__ src, int ncolblocks, int nbands, float* sink) {
    constexpr int VEC = PIECE / 256;  // floats per lane
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = blockIdx.z, band = blockIdx.y, cb = blockIdx.x;
    const float* base = src + (size_t)tile * (32u << 22) + (size_t)band * 16 * 2048 + (size_t)cb * (PIECE / 4) + lane * VEC;
    float acc = 0.0f;
    for (int k = wave; k < 32; k += 4) {
        const float* p = base + (size_t)k * (1u << 22);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (VEC == 1) acc = fmaxf(acc, p[r * 2048]);
            else if constexpr (VEC == 2) { const float2 v = *reinterpret_cast<const float2*>(p + r * 2048); acc = fmaxf(acc, fmaxf(v.x, v.y)); }
            else { const float4 v = *reinterpret_cast<const float4*>(p + r * 2048); acc = fmaxf(acc, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w))); }
        }
    }
    if (acc == 1.2345f) *sink = acc;
}

template <int PIECE>
void run_rows(const float4* a, float* sink) {
    // 16 tiles of 2048 x 2048 x 32 floats (8.6 GB); the strip: 304 rows (19 bands) x all 2048 columns of every slice
    const int ncb = 8192 / PIECE, nbands = 19, ntiles = 16;
    const double bytes = (double)ntiles * 32 * nbands * 16 * 8192;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_read_rows<PIECE>), dim3(ncb, nbands, ntiles), dim3(256), 0, 0, reinterpret_cast<const float*>(a), ncb, nbands, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    printf("MIP-shaped read, %4d-byte pieces per row and wave: %6.3f ms  %5.2f TB/s\n", PIECE, best, bytes / best / 1e9);
    fflush(stdout);
}

int main() {
    const size_t total = (size_t)1024 * 512 * 2048 * 8;  // the C3 spectrum: 8.6 GB
    float4 *a, *b, *g;
    float* sink;
    CK(hipMalloc(&a, total));
    CK(hipMalloc(&b, total));
    CK(hipMalloc(&g, total / 2));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, total));
    CK(hipMemset(b, 0, total));
    CK(hipMemset(g, 0, total / 2));
    printf("-- MIP-shaped read-only pass (N-S strips of 16 tiles)\n");
    run_rows<256>(a, sink);
    run_rows<512>(a, sink);
    run_rows<1024>(a, sink);
    printf("-- read-only stream of 8.6 GB\n");
    for (int grid : {8192, 32768, 131072}) {
        run_read32<16>(a, total, grid, sink);
        run_read32<32>(a, total, grid, sink);
    }
    for (int grid : {2048, 8192, 32768}) {
        run_read<4>(a, total, grid, sink);
        run_read<8>(a, total, grid, sink);
        run_read<16>(a, total, grid, sink);
    }
    for (int with_g = 0; with_g < 2; ++with_g) {
        printf(with_g ? "-- with the contiguous OTF stream (4.3 GB)\n" : "-- spectrum in + out only (17.2 GB)\n");
        for (int pad : {0, 128, 256, 512, 1024, 2048}) {
            run<128, 2>("2 streams x 128-B segments (z pass, plain)", a, b, g, sink, total, with_g, pad);
            run<256, 1>("1 stream x 256-B segments", a, b, g, sink, total, with_g, pad);
            run<64, 2>("2 streams x 64-B segments", a, b, g, sink, total, with_g, pad);
        }
    }
    return 0;
}
