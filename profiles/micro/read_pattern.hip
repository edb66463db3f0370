// Read rate of the overlap views of config 5 (32 slices of 2048 x 2048 floats per tile; west-east view = 307 columns of every row,
// north-south view = 307 whole rows) with the running maxima of k_mips and nothing else, for two shapes of a wave's patch:
//   VEC = 1: 16 rows x 64 columns, one float per lane and row (k_mips);  VEC = 4: 16 rows x 256 columns, one float4 per lane and row.
//   hipcc --offload-arch=gfx950 -O3 profiles/micro/read_pattern.hip -o profiles/micro/read_pattern && profiles/micro/read_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int DK = 32, DI = 2048, DJ = 2048, OV = 307, ROWS = 16;

// WM (round 5): what a work-group WRITES when it has read its patches -- 0 nothing; 1: 16 KB contiguous, a slot of its own; 2: 64 pieces of
// 256 B at a pitch of 1228 B (the xy rows of a west-east patch); 3: the same 16 KB as 4096 atomic maxima; 4: 16 KB into a 1-MB region that
// stays in L2.  `wbuf` holds 128 KB per work-group.
template <int VEC, int DEPTH, int LDSKB, int NB, int WM = 0>
__global__ __launch_bounds__(256) void k_read(const float* __restrict__ vol, int i_lo, int n_i, int j_lo, int n_j, float* __restrict__ out, float* __restrict__ wbuf = nullptr) {
    __shared__ float pad[LDSKB * 256 + 1];  // (occupancy: LDSKB = 32 -> four work-groups per CU like k_mips)
    if (LDSKB > 0) pad[threadIdx.x] = 0.0f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const float* tile = vol + (size_t)blockIdx.z * DK * DI * DJ;
    const int jal = j_lo & ~(64 * VEC - 1);                       // blocks aligned to the tile rows
    const int j = jal + ((int)blockIdx.x * 64 + lane) * VEC;      // first column of the lane
    const bool live = j + VEC > j_lo && j < j_lo + n_j;           // (partial vectors at the view's edge are read whole)
    const int ib0 = i_lo + (int)blockIdx.y * NB * ROWS;
    float acc = 0.0f;
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    for (int b = 0; b < NB; ++b) {
        const int i0 = ib0 + b * ROWS;
        if (i0 >= i_lo + n_i) break;
        const int rows = min(ROWS, i_lo + n_i - i0);
        vec_t v[DEPTH][ROWS];
        auto load = [&](int k, vec_t (&dst)[ROWS]) {
            const float* p = tile + (size_t)k * DI * DJ + (size_t)i0 * DJ;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                if (live && r < rows) dst[r] = *reinterpret_cast<const vec_t*>(p + (size_t)r * DJ + j);
                else dst[r] = vec_t(0.0f);
            }
        };
        load(wave, v[0]);
        if (DEPTH > 1 && wave + 4 < DK) load(wave + 4, v[1 % DEPTH]);
#pragma unroll 1
        for (int k = wave; k < DK; k += 4 * DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (k + 4 * d >= DK) break;
#pragma unroll
                for (int r = 0; r < ROWS; ++r)
#pragma unroll
                    for (int c = 0; c < VEC; ++c) acc = fmaxf(acc, VEC == 1 ? ((float*)&v[d][r])[0] : v[d][r][c]);
                if (k + 4 * (d + DEPTH) < DK) load(k + 4 * (d + DEPTH), v[d]);
            }
        }
    }
    if (acc == 12345.0f) out[0] = acc + pad[threadIdx.x];  // (keeps the loads alive)
    if (WM) {
        const size_t wg = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        float* w = wbuf + (WM == 4 ? (wg & 7) : wg) * 32768;   // 128 KB per work-group (mode 2 spans 63 * 307 + 64 floats = 78 KB)
        if (WM == 1 || WM == 4) {
            for (int e = threadIdx.x; e < 4096; e += 256) w[e] = acc;
        } else if (WM == 2) {
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            for (int r = wave; r < 64; r += 4) w[r * 307 + lane] = acc;
        } else if (WM == 3) {
            for (int e = threadIdx.x; e < 4096; e += 256) atomicMax(reinterpret_cast<int*>(w) + e, __float_as_int(acc));
        }
    }
}

template <int VEC, int DEPTH, int LDSKB = 0, int NB = 4, int WM = 0>
void run(const char* name, const float* vol, int tiles, bool west_east, float* out, float* wbuf = nullptr) {
    const int i_lo = west_east ? 0 : DI - OV, n_i = west_east ? DI : OV, j_lo = west_east ? DJ - OV : 0, n_j = west_east ? OV : DJ;
    const int jal = j_lo & ~(64 * VEC - 1);
    const int cblocks = (j_lo + n_j - jal + 64 * VEC - 1) / (64 * VEC), bands = (n_i + NB * ROWS - 1) / (NB * ROWS);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_read<VEC, DEPTH, LDSKB, NB, WM>), dim3(cblocks, bands, tiles), dim3(256), 0, 0, vol, i_lo, n_i, j_lo, n_j, out, wbuf);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)tiles * DK * n_i * n_j * 4;
    printf("%-42s %s: %.3f ms, %.0f GB/s of the view's bytes\n", name, west_east ? "west-east  " : "north-south", best, bytes / best / 1e6);
}

int main() {
    const int tiles = 56;  // one group of config 5 reads 112 views; 56 tiles = 30 GB keep the allocation modest
    float *vol, *out;
    CK(hipMalloc(&vol, sizeof(float) * (size_t)tiles * DK * DI * DJ));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(vol, 0, sizeof(float) * (size_t)tiles * DK * DI * DJ));
    if (getenv("READ_PATTERN_WRITES")) {   // round 5: a streaming read with a few per cent of writes at the end of every work-group
        float* wbuf;
        const size_t wbytes = (size_t)131072 * (5 * 32 * tiles + 1);   // 5 x 32 (or 32 x 5) work-groups per tile, one slot to spare
        CK(hipMalloc(&wbuf, wbytes));
        CK(hipMemset(wbuf, 0, wbytes));
        for (int we = 1; we >= 0; --we) {
            run<1, 2, 32, 4, 0>("reads only", vol, tiles, we, out, wbuf);
            run<1, 2, 32, 4, 1>("+ 16 KB contiguous per work-group (3 %)", vol, tiles, we, out, wbuf);
            run<1, 2, 32, 4, 2>("+ 64 pieces of 256 B at a pitch of 1228 B", vol, tiles, we, out, wbuf);
            run<1, 2, 32, 4, 3>("+ 4096 atomic maxima (16 KB)", vol, tiles, we, out, wbuf);
            run<1, 2, 32, 4, 4>("+ 16 KB into a region that stays in L2", vol, tiles, we, out, wbuf);
            run<1, 2, 32, 4, 0>("reads only (again)", vol, tiles, we, out, wbuf);
        }
        return 0;
    }
    for (int we = 1; we >= 0; --we) {
        run<1, 1>("float per lane, 1 slice in flight", vol, tiles, we, out);
        run<1, 2>("float per lane, 2 slices", vol, tiles, we, out);
        run<1, 2, 32>("float per lane, 2 slices, 4 WG/CU", vol, tiles, we, out);
        run<1, 1, 32>("float per lane, 1 slice, 4 WG/CU", vol, tiles, we, out);
        run<1, 2, 48>("float per lane, 2 slices, 3 WG/CU", vol, tiles, we, out);
        run<1, 2, 32, 1>("float, 2 slices, 4 WG/CU, 1 band per WG", vol, tiles, we, out);
        run<1, 2, 32, 2>("float, 2 slices, 4 WG/CU, 2 bands", vol, tiles, we, out);
        run<1, 2, 32, 8>("float, 2 slices, 4 WG/CU, 8 bands", vol, tiles, we, out);
        run<1, 2, 32, 32>("float, 2 slices, 4 WG/CU, 32 bands", vol, tiles, we, out);
        run<2, 2>("float2 per lane, 2 slices", vol, tiles, we, out);
        run<4, 1>("float4 per lane, 1 slice", vol, tiles, we, out);
        run<4, 2>("float4 per lane, 2 slices", vol, tiles, we, out);
    }
    return 0;
}
