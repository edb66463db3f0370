// Shared core of the MIP-NCC translation units (ncc.hip: per-pair path + C ABI; ncc_lag.hip: batched lag-FFT pipeline):
// device kernels, summed-area-table window statistics, the bit-identical host logic of compute_funcs.cu:160-342,1294-1609
// and the per-pair working set.  Everything lives in an anonymous namespace: each including TU gets its own copy.
#pragma once
#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "mi_crossmips.h"
#include "mi_internal.h"

using namespace mi;

namespace mi {
// cumulative counters behind mi_ncc_stats (defined in ncc.hip): pairs finished by the batched pipeline, pairs finished by the
// per-pair (careful) path, entries recomputed in the two-pass form
void ncc_count(int which, long long n);
// Sample format of the tiles: float (iom::real_t, the reference's) or the 8 / 16-bit integers they were loaded from (k_mips_int), with
// the divisor that turns those into the reference's floats.  Tile pointers travel as `const float*` either way.
#ifndef MI_TILE_FMT_DEFINED
#define MI_TILE_FMT_DEFINED
struct TileFmt {
    int bytes = 4;  // 4: float; 2 / 1: the integer samples the tiles were loaded from
    float scale = 65535.0f;
};
#endif
}

namespace {

constexpr int TILE = 32;  // TILE_SIDE, compute_funcs.h:66
constexpr int MIP_KPW = 8;    // slices per wave whose column maxima stay in registers (stacks of up to 32 slices)
constexpr int MIP_NB = 4;     // row bands per work-group in that case
constexpr int MIP_ROWS = 16;
constexpr int NCC_THREADS = 256;

__device__ __forceinline__ void atomic_max_nonneg(float* addr, float v) {
    atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
}

// maximum of a non-negative value over the 64 lanes of a wave, valid in lane 63: DPP moves inside the VALU (quad permutes, row
// mirrors, row broadcasts) instead of six trips through the LDS crossbar
__device__ __forceinline__ float wave_max_nonneg(float v) {
    auto step = [](float x, auto ctrl, auto row_mask) {
        const int y = __builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, decltype(row_mask)::value, 0xf, false);
        return fmaxf(x, __int_as_float(y));  // lanes without a source keep 0 = the neutral element
    };
    using std::integral_constant;
    v = step(v, integral_constant<int, 0xB1>{}, integral_constant<int, 0xf>{});   // quad_perm [1,0,3,2]
    v = step(v, integral_constant<int, 0x4E>{}, integral_constant<int, 0xf>{});   // quad_perm [2,3,0,1]
    v = step(v, integral_constant<int, 0x141>{}, integral_constant<int, 0xf>{});  // row_half_mirror
    v = step(v, integral_constant<int, 0x140>{}, integral_constant<int, 0xf>{});  // row_mirror: every lane holds its row's max
    v = step(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});  // row_bcast:15 into rows 1 and 3
    v = step(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});  // row_bcast:31 into rows 2 and 3
    return v;
}

// Maxima of 16 per-lane values over the 64 lanes of a wave, all 16 rows at once (transpose reduction): in each of four steps a lane
// hands one half of its rows to the partner lane^m and keeps the other half, taking the maximum with what it receives -- 8 + 4 +
// 2 + 1 exchanges instead of 16 x 4 -- then two more exchanges merge the four 16-lane groups.  Returns, in EVERY lane, the maximum
// of row rows_of_lane(lane) = bit-reversal of (lane & 15).  Values >= 0.
__device__ __forceinline__ int row_of_lane(int lane) { return ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3); }
__device__ __forceinline__ float rows_max16(const float (&v)[MIP_ROWS], int lane) {
    float w[8];
    {   // partner lane ^ 1 (DPP quad_perm [1,0,3,2])
        const bool up = lane & 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float mine = up ? v[i + 8] : v[i], send = up ? v[i] : v[i + 8];
            const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0xB1, 0xf, 0xf, false));
            w[i] = fmaxf(mine, recv);
        }
    }
    {   // partner lane ^ 2 (DPP quad_perm [2,3,0,1])
        const bool up = lane & 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float mine = up ? w[i + 4] : w[i], send = up ? w[i] : w[i + 4];
            const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x4E, 0xf, 0xf, false));
            w[i] = fmaxf(mine, recv);
        }
    }
    {
        const bool up = lane & 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float mine = up ? w[i + 2] : w[i], send = up ? w[i] : w[i + 2];
            w[i] = fmaxf(mine, __shfl_xor(send, 4, 64));
        }
    }
    {
        const bool up = lane & 8;
        const float mine = up ? w[1] : w[0], send = up ? w[0] : w[1];
        w[0] = fmaxf(mine, __shfl_xor(send, 8, 64));
    }
    w[0] = fmaxf(w[0], __shfl_xor(w[0], 16, 64));
    w[0] = fmaxf(w[0], __shfl_xor(w[0], 32, 64));
    return w[0];
}

// view(k,i,j) = vol[k*slice + (i+i0)*pitch + (j+j0)]; blockIdx.z = 2 * pair + (0: A, 1: B).  A batch hands the tile pointers
// over as a device table (tab[2 * pair + which]); every output array of pair q sits q * pstride floats behind pair 0's.
// Since round 5 this is the pass for stacks DEEPER than 4 * MIP_KPW slices only (k_mips5 below takes the others): partial row / column
// maxima per band and column block in `yz_tmp` / `xz_tmp`, reduced by k_mips_yz / k_mips_xz.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_mips(const float* __restrict__ A, const float* __restrict__ B, const float* const* __restrict__ tab,
                                               size_t pstride, int dimk, int dimi_v, int dimj_v,
                                               size_t slice, int pitch, int ai0, int aj0, float* __restrict__ xy1, float* __restrict__ xz1,
                                               float* __restrict__ yz1, float* __restrict__ xy2, float* __restrict__ xz2,
                                               float* __restrict__ yz2, float* __restrict__ yz_tmp, float* __restrict__ xz_tmp, int knock,
                                               float* __restrict__ xyT, size_t tstride) {
    // xyT (batched pipeline, views whose rows are the long axis): a second, TRANSPOSED copy of the xy MIP -- [j][i], MIP `second` of
    // pair q at xyT + (2 q + second) * tstride -- so that the lag transform along i reads contiguous lines (as a strided gather
    // it took 0.46 instead of 0.24 ms per 56 pairs: every lane of a load another 128-byte line)
    // (knock: measurement aid, MI_NCC_MIPS_KNOCK -- 1: no xz maxima, 2: no yz maxima, 4: no xy store)
    // a work-group owns a 16-row x 64-column patch of the view; its four waves share the slices (wave w: k = w, w + 4, ...),
    // so a patch keeps four times as many loads in flight as one wave walking all slices
    extern __shared__ float xzp[];  // [band][MIP_ROWS][dimk] row maxima of the work-group's patches (xz_tmp != nullptr)
    const bool second = blockIdx.z & 1;
    const size_t poff = (size_t)(blockIdx.z >> 1) * pstride;
    const float* vol = tab ? tab[blockIdx.z] : (second ? B : A);
    if (!second) vol += (size_t)ai0 * pitch + aj0;
    float* xy = (second ? xy2 : xy1) + poff;
    float* xz = (second ? xz2 : xz1) + poff;
    (void)yz1; (void)yz2;  // written by k_mips_yz
    // (the wave index as a scalar: row addresses are then scalar bases + one per-lane offset instead of 16 address pairs)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // column blocks start on 256-byte boundaries of the TILE rows, not of the view (the first tile's view of a west-east pair
    // starts at column 1741 of 2048: blocks laid out from there shared their first and last 128-byte lines with their neighbours,
    // three lines fetched for two used); the first block is then the partial one
    const int jshift = second ? 0 : (aj0 & 63);
    const int j = (int)blockIdx.x * 64 + lane - jshift;
    const bool live = j >= 0 && j < dimj_v;
    // A work-group walks MIP_NB consecutive row bands (stacks of up to 4 * MIP_KPW slices): the column maxima of the slices a wave
    // owns stay with it across the bands (in LDS, one column per lane and slice of the wave) and go to yz_tmp ONCE per band group
    // -- a quarter of the partial maxima that k_mips_yz has to read back.  Deeper stacks write them per band (nb = 1 below).
    // NOTHING is stored to memory before the last slice has been requested: loads and stores of a wave return in order, so a
    // store in the middle of the stream -- the xy rows of a band, its xz maxima -- held up the loads requested behind it until
    // its acknowledgement came back, and the other three waves at the next barrier (switching the xy store off saved 11 % of
    // the pass although it is 3 % of its bytes: profiles/r03_mips_knock.txt).  The xy maxima of a band are merged across the
    // waves with LDS atomics, the xz maxima of all bands wait in LDS: the band loop has no barrier left.
    const bool keep = dimk <= 4 * MIP_KPW;
    const int nb = keep ? MIP_NB : 1;
    __shared__ float xyb[MIP_NB][MIP_ROWS][64];
    __shared__ float cacc[4][MIP_KPW][64];
    float* colacc = &cacc[wave][0][lane];
    if (keep) {
#pragma unroll
        for (int q = 0; q < MIP_KPW; ++q) colacc[q * 64] = 0.0f;
    }
    for (int e = threadIdx.x; e < MIP_NB * MIP_ROWS * 64; e += 256) (&xyb[0][0][0])[e] = 0.0f;
    const int ib0 = __builtin_amdgcn_readfirstlane((int)blockIdx.y * nb * MIP_ROWS);
    const int nbv = min(nb, (dimi_v - ib0 + MIP_ROWS - 1) / MIP_ROWS);  // bands of this work-group inside the view (>= 1)
    // the wave's slices of all its bands form one sequence: the next slice -- of this band or the first of the next -- is requested
    // before the current one is reduced
    float v[MIP_ROWS], vn[MIP_ROWS];
    // Every load is unconditional: lanes outside the view read its nearest column, rows past the last band its last row -- a
    // maximum does not change when a sample is taken twice (the partial results of such lanes are never stored).  Predicated
    // loads cost a branch around each of the 16 rows of a slice (`s_cbranch_execz`), in the inner loop of a memory-bound pass.
    const int jc = min(max(j, 0), dimj_v - 1);
    auto load_slice = [&](int bb, int k, float (&dst)[MIP_ROWS]) {
        const int i0 = ib0 + bb * MIP_ROWS, last = min(MIP_ROWS, dimi_v - i0) - 1;
        // (a GLOBAL pointer, said explicitly: the tile pointer comes out of a table in memory, and what the compiler cannot prove
        // global it reads with FLAT loads -- which count against the LDS counter as well and are waited for with vmcnt(0).
        // Measured on one box, same run: 1.87 / 1.81 ms with global loads and counted waits, 1.90 / 1.82 ms with FLAT loads -- the
        // pass is not bound by what a wave keeps in flight (three or four waves per SIMD are the optimum, five or two are slower))
        typedef const float __attribute__((address_space(1))) gfloat;
        gfloat* p = (gfloat*)(vol + (size_t)k * slice + (size_t)i0 * pitch);  // wave-uniform
#pragma unroll
        for (int r = 0; r < MIP_ROWS; ++r) dst[r] = (p + (size_t)min(r, last) * pitch)[jc];
    };
    if (wave < dimk) load_slice(0, wave, v);
    __syncthreads();  // (xyb is zero)
#pragma unroll 1
    for (int b = 0; b < nbv; ++b) {
        const int i0 = ib0 + b * MIP_ROWS;  // (scalar: so are the row addresses)
        const int rows = min(MIP_ROWS, dimi_v - i0);
        float best[MIP_ROWS];
#pragma unroll
        for (int r = 0; r < MIP_ROWS; ++r) best[r] = 0.0f;
        // one slice of the wave, held in `cur`; the wave's next one is requested into `nxt` first.  Returns the column maximum.
        // (The two buffers swap roles from slice to slice -- the loops below take two slices per trip.  With one loop body and a
        // copy nxt -> cur at its end the compiler had to wait for the slice it had just requested: `s_waitcnt vmcnt(0)` in every
        // step, no load in flight while a slice was reduced.)
        auto slice_step = [&](int k, float (&cur)[MIP_ROWS], float (&nxt)[MIP_ROWS]) {
            const bool wrap = k + 4 >= dimk;  // (wave-uniform)
            const int kn = __builtin_amdgcn_readfirstlane(wrap ? wave : k + 4), bn = __builtin_amdgcn_readfirstlane(wrap ? b + 1 : b);
            // (behind the work-group's last slice there is nothing to fetch: the first slice of its last band is read once more --
            // one slice in 128 -- so that the number of loads in flight is the same in every step)
            if (bn < nbv) load_slice(bn, kn, nxt);
            float colmax = 0.0f;
#pragma unroll
            for (int r = 0; r < MIP_ROWS; ++r) {
                best[r] = fmaxf(best[r], cur[r]);
                colmax = fmaxf(colmax, cur[r]);
            }
            // xz: the 16 row maxima over the 64 columns of the patch, per patch through LDS into xz_tmp[tile][column block][i][k]
            // (k_mips_xz takes the maximum over the column blocks); as atomics on the MIP they were 655 thousand single-lane
            // atomics per C5 pair, as 16 separate wave reductions 96 DPP steps per slice
            if (!(knock & 1)) {
                const float rowmax = rows_max16(cur, lane);
                const int r = row_of_lane(lane);
                if (lane < 16) {
                    if (xz_tmp) xzp[(b * MIP_ROWS + r) * dimk + k] = rowmax;
                    else if (r < rows) atomic_max_nonneg(&xz[(size_t)(i0 + r) * dimk + k], rowmax);
                }
            }
            return (knock & 2) ? 0.0f : colmax;
        };
        // yz: the column maxima of this band (group) go to yz_tmp[tile][group][k][j] (unit-stride stores); k_mips_yz takes the
        // maximum over the groups -- 2.6 million atomics per pair on the yz MIPs cost more than the whole streaming pass
        auto yz_out = [&](int k, int q, float colmax) {
            if (keep) colacc[q * 64] = fmaxf(colacc[q * 64], colmax);
            else if (live) yz_tmp[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * dimk + k) * dimj_v + j] = colmax;
        };
        int k = wave, q = 0;
#pragma unroll 1
        for (; k + 4 < dimk; k += 8, q += 2) {
            yz_out(k, q, slice_step(k, v, vn));
            yz_out(k + 4, q + 1, slice_step(k + 4, vn, v));
        }
        if (k < dimk) {  // an odd number of slices: the next band's first slice has arrived in the other buffer
            yz_out(k, q, slice_step(k, v, vn));
#pragma unroll
            for (int r = 0; r < MIP_ROWS; ++r) v[r] = vn[r];
        }
        // xy: maximum over the four waves' slices
#pragma unroll
        for (int r = 0; r < MIP_ROWS; ++r) atomic_max_nonneg(&xyb[b][r][lane], best[r]);
    }
    __syncthreads();
    if (wave < nbv && live && !(knock & 4)) {  // wave b stores band b
        const int i0 = ib0 + wave * MIP_ROWS, rows = min(MIP_ROWS, dimi_v - i0);
#pragma unroll
        for (int r = 0; r < MIP_ROWS; ++r)
            if (r < rows) xy[(size_t)(i0 + r) * dimj_v + j] = xyb[wave][r][lane];
    }
    if (xyT) {  // column c of the patch = 16 nbv consecutive floats of line j of the transposed copy: the lanes run along i
        float* t = xyT + (size_t)blockIdx.z * tstride;
        const int i = ib0 + lane;
        if (lane < nbv * MIP_ROWS && i < dimi_v)
            for (int c = wave; c < 64; c += 4) {
                const int jc_ = (int)blockIdx.x * 64 + c - jshift;
                if (jc_ >= 0 && jc_ < dimj_v) t[(size_t)jc_ * dimi_v + i] = xyb[lane >> 4][lane & 15][c];
            }
    }
    if (xz_tmp) {  // the rows of all bands * dimk: contiguous floats
        float* dst = xz_tmp + (((size_t)blockIdx.z * gridDim.x + blockIdx.x) * dimi_v + ib0) * dimk;
        const int total = min(nbv * MIP_ROWS, dimi_v - ib0) * dimk;
        for (int e = threadIdx.x; e < total; e += 256) dst[e] = xzp[e];
    }
    if (keep && live) {
#pragma unroll
        for (int q = 0; q < MIP_KPW; ++q) {
            const int k = wave + 4 * q;
            if (k < dimk) yz_tmp[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * dimk + k) * dimj_v + j] = colacc[q * 64];
        }
    }
}

// ---- k_mips5: the float pass for stacks of up to 4 * MIP_KPW slices (every C5-like stack), round-5 form of k_mips<true>.  Same
// patches, same slice pipeline (a work-group owns 16 rows x 64 columns x MIP_NB bands, its four waves share the slices, the next
// slice is requested before the current one is reduced, nothing is stored before the last load).  What changed:
//  * addressing: BUFFER loads -- the slice's first row is the descriptor base (wave-uniform, five scalar instructions per slice), the 16
//    row offsets of the wave's band wait in SGPRs (recomputed when the band changes, once per 8 slices) and go into the instruction's
//    scalar offset, the lane's column into its vector offset.  k_mips<true> rebuilt 16 64-bit row addresses per slice: 92 scalar
//    instructions per slice against 155 vector ones (profiles/r04_ncc_sq_counters.txt).
//  * no partial maxima in memory: the column maxima (yz) and row maxima (xz) a work-group has gathered in LDS are merged into the
//    MIPs themselves with integer atomic maxima (non-negative floats order like their bit patterns), 256 contiguous bytes per wave
//    instruction, ONCE per work-group -- 0.29 GB per 56-pair launch were written as yz_tmp / xz_tmp and read back by k_mips_yz /
//    k_mips_xz.  The MIPs must be zero when the pass starts (k_mips_zero).
//  * LDS images without bank conflicts: the row maxima at a stride of 33 words (16 lanes stored at a stride of dimk = 32 words: one
//    bank), the xy rows and column maxima at 65 (the transposed copy read them at a stride of 64 words: one bank for 64 lanes).
constexpr int MIP5_XS = 4 * MIP_KPW + 1;  // words per row of the xz image
__global__ __launch_bounds__(256) void k_mips_zero(size_t pstride, size_t n_xz, size_t n_yz, float* __restrict__ xz1, float* __restrict__ yz1,
                                                   float* __restrict__ xz2, float* __restrict__ yz2) {
    const size_t poff = (size_t)(blockIdx.y >> 1) * pstride;
    float* xz = ((blockIdx.y & 1) ? xz2 : xz1) + poff;
    float* yz = ((blockIdx.y & 1) ? yz2 : yz1) + poff;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n_xz + n_yz; e += (size_t)gridDim.x * 256) {
        if (e < n_xz) xz[e] = 0.0f;
        else yz[e - n_xz] = 0.0f;
    }
}

template <int WPE>  // waves per SIMD = work-groups per compute unit
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_mips5(
    const float* __restrict__ A, const float* __restrict__ B, const float* const* __restrict__ tab, size_t pstride, int dimk, int dimi_v, int dimj_v,
    size_t slice, int pitch, int ai0, int aj0, float* __restrict__ xy1, float* __restrict__ xz1, float* __restrict__ yz1, float* __restrict__ xy2,
    float* __restrict__ xz2, float* __restrict__ yz2, int knock, float* __restrict__ xyT, size_t tstride, int cblocks, int groups, int nviews, int remap) {
    // Patch of this work-group: column block bx, band group by, view bz, from a one-dimensional grid.  remap: work-groups b, b + 8,
    // b + 16, ... -- they run on one XCD and start together -- take the `cblocks` column blocks of ONE group of rows: their xy rows
    // are pieces of the same lines and of the same DRAM pages, and what one L2 collects it writes back as whole runs.
    int bx, by, bz;
    {
        const int id = (int)blockIdx.x;
        int rg;
        if (remap) {
            const int s_ = id >> 3;
            bx = s_ % cblocks;
            rg = (s_ / cblocks) * 8 + (id & 7);
        } else {
            bx = id % cblocks;
            rg = id / cblocks;
        }
        by = rg % groups;
        bz = rg / groups;
        if (bz >= nviews) return;   // (the grid is padded to whole groups of 8 row groups)
    }
    __shared__ float xyb[MIP_NB][MIP_ROWS][65];        // xy maxima of the bands (merged across the waves with LDS atomics)
    __shared__ float cacc[4][MIP_KPW][65];             // column maxima of the wave's slices over all bands: slice k = wave + 4 q
    __shared__ float xzp[MIP_NB * MIP_ROWS][MIP5_XS];  // row maxima over the patch's 64 columns
    const bool second = bz & 1;
    const size_t poff = (size_t)(bz >> 1) * pstride;
    const float* vol = tab ? tab[bz] : (second ? B : A);
    if (!second) vol += (size_t)ai0 * pitch + aj0;
    float* xy = (second ? xy2 : xy1) + poff;
    float* xz = (second ? xz2 : xz1) + poff;
    float* yz = (second ? yz2 : yz1) + poff;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int jshift = second ? 0 : (aj0 & 63);  // column blocks start on 256-byte boundaries of the TILE rows (see k_mips)
    const int j = bx * 64 + lane - jshift;
    const bool live = j >= 0 && j < dimj_v;
    float* colacc = &cacc[wave][0][lane];
#pragma unroll
    for (int q = 0; q < MIP_KPW; ++q) colacc[q * 65] = 0.0f;
    for (int e = threadIdx.x; e < MIP_NB * MIP_ROWS * 65; e += 256) (&xyb[0][0][0])[e] = 0.0f;
    const int ib0 = __builtin_amdgcn_readfirstlane(by * MIP_NB * MIP_ROWS);
    const int nbv = min(MIP_NB, (dimi_v - ib0 + MIP_ROWS - 1) / MIP_ROWS);  // bands of this work-group inside the view (>= 1)
    float v[MIP_ROWS], vn[MIP_ROWS];
    // every load is unconditional (see k_mips): lanes outside the view read its nearest column, rows past the last band its last row
    const int voff = 4 * min(max(j, 0), dimj_v - 1);
    const int pitch4 = 4 * pitch;
    int ro[MIP_ROWS];  // byte offsets of the 16 rows of the band being REQUESTED from the first row of its slice (scalars)
    auto set_band = [&](int bb) {
        const int last = min(MIP_ROWS, dimi_v - (ib0 + bb * MIP_ROWS)) - 1;
#pragma unroll
        for (int r = 0; r < MIP_ROWS; ++r) ro[r] = __builtin_amdgcn_readfirstlane(min(r, last) * pitch4);
    };
    auto load_slice = [&](int bb, int k, float (&dst)[MIP_ROWS]) {
        const float* p = vol + (size_t)k * slice + (size_t)(ib0 + bb * MIP_ROWS) * pitch;  // wave-uniform
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int r = 0; r < MIP_ROWS; ++r) dst[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, ro[r], 0));
    };
    set_band(0);
    if (wave < dimk) load_slice(0, wave, v);
    __syncthreads();  // (the LDS images are zero)
#pragma unroll 1
    for (int b = 0; b < nbv; ++b) {
        float best[MIP_ROWS];
#pragma unroll
        for (int r = 0; r < MIP_ROWS; ++r) best[r] = 0.0f;
        auto slice_step = [&](int k, int q, float (&cur)[MIP_ROWS], float (&nxt)[MIP_ROWS]) {
            const bool wrap = k + 4 >= dimk;  // (wave-uniform)
            const int kn = __builtin_amdgcn_readfirstlane(wrap ? wave : k + 4);
            // (behind the work-group's last slice there is nothing to fetch: the first slice of its last band is read once more --
            // one slice in 128 -- so that the number of loads in flight is the same in every step)
            const int bn = __builtin_amdgcn_readfirstlane(wrap ? min(b + 1, nbv - 1) : b);
            if (wrap) set_band(bn);
            load_slice(bn, kn, nxt);
            float colmax = 0.0f;
#pragma unroll
            for (int r = 0; r < MIP_ROWS; ++r) {
                best[r] = fmaxf(best[r], cur[r]);
                colmax = fmaxf(colmax, cur[r]);
            }
            if (!(knock & 1)) {
                const float rowmax = rows_max16(cur, lane);
                if (lane < 16) xzp[b * MIP_ROWS + row_of_lane(lane)][k] = rowmax;
            }
            if (!(knock & 2)) colacc[q * 65] = fmaxf(colacc[q * 65], colmax);
        };
        int k = wave, q = 0;
#pragma unroll 1
        for (; k + 4 < dimk; k += 8, q += 2) {
            slice_step(k, q, v, vn);
            slice_step(k + 4, q + 1, vn, v);
        }
        if (k < dimk) {  // an odd number of slices: the next band's first slice has arrived in the other buffer
            slice_step(k, q, v, vn);
#pragma unroll
            for (int r = 0; r < MIP_ROWS; ++r) v[r] = vn[r];
        }
        // xy: maximum over the four waves' slices
#pragma unroll
        for (int r = 0; r < MIP_ROWS; ++r) atomic_max_nonneg(&xyb[b][r][lane], best[r]);
    }
    __syncthreads();
    if (wave < nbv && live && !(knock & 4)) {  // wave b stores band b
        const int i0 = ib0 + wave * MIP_ROWS, rows = min(MIP_ROWS, dimi_v - i0);
#pragma unroll
        for (int r = 0; r < MIP_ROWS; ++r)
            if (r < rows) xy[(size_t)(i0 + r) * dimj_v + j] = xyb[wave][r][lane];
    }
    if (xyT) {  // column c of the patch = 16 nbv consecutive floats of line j of the transposed copy: the lanes run along i
        float* t = xyT + (size_t)bz * tstride;
        const int i = ib0 + lane;
        if (lane < nbv * MIP_ROWS && i < dimi_v)
            for (int c = wave; c < 64; c += 4) {
                const int jc_ = bx * 64 + c - jshift;
                if (jc_ >= 0 && jc_ < dimj_v) t[(size_t)jc_ * dimi_v + i] = xyb[lane >> 4][lane & 15][c];
            }
    }
    // xz[i][k] and yz[j][k]: a half-wave takes the dimk slices of one row / column -- 128 contiguous bytes of the MIP when dimk = 32
    {
        const int k = threadIdx.x & 31;
        const int rows = min(nbv * MIP_ROWS, dimi_v - ib0);
        if (k < dimk && !(knock & 8)) {
#pragma unroll
            for (int it = 0; it < MIP_NB * MIP_ROWS / 8; ++it) {
                const int r = (threadIdx.x >> 5) + 8 * it;
                if (r < rows) atomic_max_nonneg(&xz[(size_t)(ib0 + r) * dimk + k], xzp[r][k]);
            }
            const float* ck = &cacc[k & 3][k >> 2][0];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int c = (threadIdx.x >> 5) + 8 * it, jc_ = bx * 64 + c - jshift;
                if (jc_ >= 0 && jc_ < dimj_v) atomic_max_nonneg(&yz[(size_t)jc_ * dimk + k], ck[c]);
            }
        }
    }
}

// ---- 16-bit tiles.  TeraStitcher turns the 8 / 16-bit samples of its TIFF tiles into floats in [0, 1] when it loads them (value /
// 255 or / 65535, tiff2D.cpp:606-610) and compute_3_MIPs reads those; the division is monotonic, so the MIPs of the floats are the
// divided MIPs of the integers, bit for bit.  k_mips_int reads the tiles as they are stored -- half or a quarter of the bytes of the pass that
// dominates a batch -- and a lane takes TWO neighbouring columns as one packed 32-bit value: running maxima, column maxima and the
// transpose reduction of the row maxima work on both halves at once (v_pk_max_u16), i.e. the same instructions as for one float
// column move twice the voxels.  Only the results are divided (IEEE division: what numpy's float32 division gives the host mirror).
typedef unsigned short mi_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(mi_u16x2, a), __builtin_bit_cast(mi_u16x2, b)));
}
// rows_max16 for packed pairs: returns, in every lane, the packed maxima of row row_of_lane(lane) over the 64 lanes
__device__ __forceinline__ unsigned rows_max16_pk(const unsigned (&v)[MIP_ROWS], int lane) {
    unsigned w[8];
    {
        const bool up = lane & 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned mine = up ? v[i + 8] : v[i], send = up ? v[i] : v[i + 8];
            w[i] = pk_max_u16(mine, (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1, 0xf, 0xf, false));
        }
    }
    {
        const bool up = lane & 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned mine = up ? w[i + 4] : w[i], send = up ? w[i] : w[i + 4];
            w[i] = pk_max_u16(mine, (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, 0x4E, 0xf, 0xf, false));
        }
    }
    {
        const bool up = lane & 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned mine = up ? w[i + 2] : w[i], send = up ? w[i] : w[i + 2];
            w[i] = pk_max_u16(mine, (unsigned)__shfl_xor((int)send, 4, 64));
        }
    }
    {
        const bool up = lane & 8;
        const unsigned mine = up ? w[1] : w[0], send = up ? w[0] : w[1];
        w[0] = pk_max_u16(mine, (unsigned)__shfl_xor((int)send, 8, 64));
    }
    w[0] = pk_max_u16(w[0], (unsigned)__shfl_xor((int)w[0], 16, 64));
    w[0] = pk_max_u16(w[0], (unsigned)__shfl_xor((int)w[0], 32, 64));
    return w[0];
}

// The views as in k_mips (grid z = 2 * pair + tile, tab / A / B, ai0 / aj0 of the first tile).  BYTES = 2: a lane's word holds two
// columns, a work-group owns 16 rows x 128 columns per band and walks MIP_NB bands.  BYTES = 1: four columns per word, taken apart
// into two packed pairs (even / odd bytes) that go through the same packed instructions -- 16 rows x 256 columns per band, half
// as many bands per work-group (the xy maxima of its bands wait in LDS, one word per column).  Words are aligned to the TILE rows
// (dimj and the slice size a multiple of 4 / BYTES samples: 32-bit loads).  Stacks of up to 4 * MIP_KPW slices.
// `scale`: 65535 for 16-bit, 255 for 8-bit samples (tiff2D.cpp:606-610).
template <int BYTES>
struct IntTiles {
    static constexpr int P = BYTES == 2 ? 1 : 2;  // packed pairs per word
    static constexpr int C = 2 * P;               // columns per lane
    static constexpr int NB = MIP_NB / P;         // row bands per work-group
    static constexpr int W = 64 * C;              // columns per work-group
};
template <int BYTES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_mips_int(
    const unsigned char* __restrict__ A, const unsigned char* __restrict__ B, const unsigned char* const* __restrict__ tab, size_t pstride, int dimk,
    int dimi_v, int dimj_v, size_t slice, int pitch, int ai0, int aj0, float scale, float* __restrict__ xy1, float* __restrict__ xz1,
    float* __restrict__ yz1, float* __restrict__ xy2, float* __restrict__ xz2, float* __restrict__ yz2, float* __restrict__ xyT, size_t tstride,
    int cblocks, int groups, int nviews) {
    int bx, by, bz;   // column block, band group, view: the column blocks of one group of rows on ONE XCD (see k_mips5)
    {
        const int id = (int)blockIdx.x, s_ = id >> 3;
        bx = s_ % cblocks;
        const int rg = (s_ / cblocks) * 8 + (id & 7);
        by = rg % groups;
        bz = rg / groups;
        if (bz >= nviews) return;   // (the grid is padded to whole groups of 8 row groups)
    }
    // (round 5, as k_mips5: buffer loads with the band's row offsets in SGPRs; row and column maxima merged into the zeroed MIPs with
    // atomic maxima of the DIVIDED values -- the division is monotonic --, LDS images at odd strides)
    using G = IntTiles<BYTES>;
    constexpr int P = G::P, C = G::C, NB = G::NB;
    constexpr int XW = 64 * C + 1;                     // words per row of the xy image
    constexpr int CS = P * 64 + 1;                     // words per (wave, q) row of the column maxima
    __shared__ float xzp[NB * MIP_ROWS][MIP5_XS];      // row maxima, already divided
    __shared__ unsigned xyb[NB][MIP_ROWS][XW];         // xy maxima of the bands, one word per column (merged with LDS atomics)
    __shared__ unsigned cacc[4 * MIP_KPW][CS];         // packed column maxima of the wave's slices: row wave * MIP_KPW + q, word p * 64 + lane
    const bool second = bz & 1;
    const size_t poff = (size_t)(bz >> 1) * pstride;
    const unsigned char* vol = tab ? tab[bz] : (second ? B : A);
    if (!second) vol += (size_t)ai0 * pitch * BYTES;
    float* xy = (second ? xy2 : xy1) + poff;
    float* xz = (second ? xz2 : xz1) + poff;
    float* yz = (second ? yz2 : yz1) + poff;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int a0 = second ? 0 : aj0, alast = a0 + dimj_v - 1;          // first / last tile column of the view
    const int cj = (a0 & ~(G::W - 1)) + bx * G::W + C * lane;  // tile column of the lane's word
    const int jv0 = cj - a0;                                             // view column of its first sample
    // every load is unconditional (see k_mips): a word outside the view reads the nearest word inside, a word that straddles the
    // view's edge takes its nearest inside sample in place of the outside ones (byte permute with a lane-constant selector)
    const int cc = min(max(cj, a0 & ~(C - 1)), alast & ~(C - 1));
    unsigned sel;
    if (BYTES == 2) {
        sel = cc < a0 ? 0x03020302u : (cc + 1 > alast ? 0x01000100u : 0x03020100u);
    } else {
        const int f = max(0, a0 - cc), l = min(3, alast - cc);
        sel = 0;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) sel |= (unsigned)min(max(bb, f), l) << (8 * bb);
    }
    // column c of the lane's word sits in half h of packed pair p
    auto col_of = [](int p, int h) { return BYTES == 2 ? h : 2 * h + p; };
    unsigned* colacc = &cacc[wave * MIP_KPW][lane];  // [q][p]: q * CS + p * 64
#pragma unroll
    for (int q = 0; q < MIP_KPW; ++q)
#pragma unroll
        for (int p = 0; p < P; ++p) colacc[q * CS + p * 64] = 0u;
    for (int e = threadIdx.x; e < NB * MIP_ROWS * XW; e += 256) (&xyb[0][0][0])[e] = 0u;
    const int ib0 = __builtin_amdgcn_readfirstlane(by * NB * MIP_ROWS);
    const int nbv = min(NB, (dimi_v - ib0 + MIP_ROWS - 1) / MIP_ROWS);
    unsigned v[MIP_ROWS], vn[MIP_ROWS];
    const int voff = cc * BYTES, pitchb = pitch * BYTES;
    int ro[MIP_ROWS];  // byte offsets of the 16 rows of the band being requested from the first row of its slice (scalars)
    auto set_band = [&](int bb) {
        const int last = min(MIP_ROWS, dimi_v - (ib0 + bb * MIP_ROWS)) - 1;
#pragma unroll
        for (int r = 0; r < MIP_ROWS; ++r) ro[r] = __builtin_amdgcn_readfirstlane(min(r, last) * pitchb);
    };
    auto load_slice = [&](int bb, int k, unsigned (&dst)[MIP_ROWS]) {
        const unsigned char* p = vol + ((size_t)k * slice + (size_t)(ib0 + bb * MIP_ROWS) * pitch) * BYTES;  // wave-uniform
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int r = 0; r < MIP_ROWS; ++r) dst[r] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, ro[r], 0);
    };
    set_band(0);
    if (wave < dimk) load_slice(0, wave, v);
    __syncthreads();  // (the LDS images are zero)
#pragma unroll 1
    for (int b = 0; b < nbv; ++b) {
        unsigned best[P][MIP_ROWS];
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int r = 0; r < MIP_ROWS; ++r) best[p][r] = 0u;
        // one slice: the packed column maxima of its pairs go to colacc[q]
        auto slice_step = [&](int k, int q, unsigned (&cur)[MIP_ROWS], unsigned (&nxt)[MIP_ROWS]) {
            const bool wrap = k + 4 >= dimk;  // (wave-uniform)
            const int kn = __builtin_amdgcn_readfirstlane(wrap ? wave : k + 4);
            const int bn = __builtin_amdgcn_readfirstlane(wrap ? min(b + 1, nbv - 1) : b);
            if (wrap) set_band(bn);
            load_slice(bn, kn, nxt);
            unsigned colmax[P];
#pragma unroll
            for (int p = 0; p < P; ++p) colmax[p] = 0u;
#pragma unroll
            for (int r = 0; r < MIP_ROWS; ++r) {
                const unsigned w = __builtin_amdgcn_perm(cur[r], cur[r], sel);
                if (BYTES == 2) {
                    best[0][r] = pk_max_u16(best[0][r], w);
                    colmax[0] = pk_max_u16(colmax[0], w);
                    cur[r] = w;
                } else {
                    const unsigned e = w & 0x00ff00ffu, o = (w >> 8) & 0x00ff00ffu;
                    best[0][r] = pk_max_u16(best[0][r], e);
                    best[P - 1][r] = pk_max_u16(best[P - 1][r], o);
                    colmax[0] = pk_max_u16(colmax[0], e);
                    colmax[P - 1] = pk_max_u16(colmax[P - 1], o);
                    cur[r] = pk_max_u16(e, o);  // (the row maximum does not care which column)
                }
            }
            const unsigned rm = rows_max16_pk(cur, lane);
            if (lane < 16) xzp[b * MIP_ROWS + row_of_lane(lane)][k] = (float)max(rm & 0xffffu, rm >> 16) / scale;
#pragma unroll
            for (int p = 0; p < P; ++p) colacc[q * CS + p * 64] = pk_max_u16(colacc[q * CS + p * 64], colmax[p]);
        };
        int k = wave, q = 0;
#pragma unroll 1
        for (; k + 4 < dimk; k += 8, q += 2) {
            slice_step(k, q, v, vn);
            slice_step(k + 4, q + 1, vn, v);
        }
        if (k < dimk) {
            slice_step(k, q, v, vn);
#pragma unroll
            for (int r = 0; r < MIP_ROWS; ++r) v[r] = vn[r];
        }
#pragma unroll
        for (int r = 0; r < MIP_ROWS; ++r)
#pragma unroll
            for (int p = 0; p < P; ++p) {
                atomicMax(&xyb[b][r][C * lane + col_of(p, 0)], best[p][r] & 0xffffu);
                atomicMax(&xyb[b][r][C * lane + col_of(p, 1)], best[p][r] >> 16);
            }
    }
    __syncthreads();
    {   // 4 / NB waves share a band's rows
        constexpr int WPB = 4 / NB, RPW = MIP_ROWS / WPB;
        const int bnd = wave % NB, r0 = (wave / NB) * RPW;
        if (bnd < nbv) {
            const int i0 = ib0 + bnd * MIP_ROWS, rows = min(MIP_ROWS, dimi_v - i0);
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int r = r0 + rr;
                if (r < rows) {
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        if (jv0 + c >= 0 && jv0 + c < dimj_v) xy[(size_t)(i0 + r) * dimj_v + jv0 + c] = (float)xyb[bnd][r][C * lane + c] / scale;
                }
            }
        }
    }
    const int jv_first = (a0 & ~(G::W - 1)) + bx * G::W - a0;  // view column of the work-group's word 0, sample 0
    if (xyT) {  // the transposed copy (see k_mips): lanes along i, a wave per column
        float* t = xyT + (size_t)bz * tstride;
        const int i = ib0 + lane;
        if (lane < nbv * MIP_ROWS && i < dimi_v)
            for (int c = wave; c < G::W; c += 4) {
                const int jv = jv_first + c;
                if (jv >= 0 && jv < dimj_v) t[(size_t)jv * dimi_v + i] = (float)xyb[lane >> 4][lane & 15][c] / scale;
            }
    }
    {   // xz[i][k] and yz[j][k]: a half-wave takes the dimk slices of one row / column (see k_mips5)
        const int k = threadIdx.x & 31;
        const int rows = min(nbv * MIP_ROWS, dimi_v - ib0);
        if (k < dimk) {
#pragma unroll
            for (int it = 0; it < NB * MIP_ROWS / 8; ++it) {
                const int r = (threadIdx.x >> 5) + 8 * it;
                if (r < rows) atomic_max_nonneg(&xz[(size_t)(ib0 + r) * dimk + k], xzp[r][k]);
            }
            const unsigned* ck = &cacc[(k & 3) * MIP_KPW + (k >> 2)][0];
#pragma unroll 4
            for (int it = 0; it < G::W / 8; ++it) {
                const int c = (threadIdx.x >> 5) + 8 * it;   // column of the work-group: word c / C, sample c % C of it
                const int l = c / C, cs = c % C;
                const int p = BYTES == 2 ? 0 : (cs & 1), h = BYTES == 2 ? cs : (cs >> 1);
                const unsigned wv = ck[p * 64 + l];
                const int jv = jv_first + c;
                if (jv >= 0 && jv < dimj_v) atomic_max_nonneg(&yz[(size_t)jv * dimk + k], (float)(h ? (wv >> 16) : (wv & 0xffffu)) / scale);
            }
        }
    }
}

// yz[j][k] = max over the row bands of yz_tmp[tile][band][k][j]; one lane per (k, j), tile = blockIdx.y = 2 * pair + which
__global__ __launch_bounds__(256) void k_mips_yz(const float* __restrict__ yz_tmp, size_t pstride, int bands, int dimk, int dimj_v,
                                                  float* __restrict__ yz1, float* __restrict__ yz2) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= dimk * dimj_v) return;
    const int k = e / dimj_v, j = e - k * dimj_v;
    const float* p = yz_tmp + (size_t)blockIdx.y * bands * dimk * dimj_v + e;
    float m = 0.0f;
    for (int b = 0; b < bands; ++b) m = fmaxf(m, p[(size_t)b * dimk * dimj_v]);
    ((blockIdx.y & 1) ? yz2 : yz1)[(size_t)(blockIdx.y >> 1) * pstride + (size_t)j * dimk + k] = m;
}

// xz[i][k] = max over the column blocks of xz_tmp[tile][block][i][k]; one lane per (i, k), tile = blockIdx.y = 2 * pair + which
__global__ __launch_bounds__(256) void k_mips_xz(const float* __restrict__ xz_tmp, size_t pstride, int blocks, int dimk, int dimi_v,
                                                  float* __restrict__ xz1, float* __restrict__ xz2) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= dimk * dimi_v) return;
    const float* p = xz_tmp + (size_t)blockIdx.y * blocks * dimk * dimi_v + e;
    float m = 0.0f;
    for (int b = 0; b < blocks; ++b) m = fmaxf(m, p[(size_t)b * dimk * dimi_v]);
    ((blockIdx.y & 1) ? xz2 : xz1)[(size_t)(blockIdx.y >> 1) * pstride + e] = m;
}

// the six MIPs of `np` pairs (np == 1 and tab == nullptr: the pair A, B): k_mips + the reductions of its partial maxima.
// `tmp` must hold mips_tmp_floats(...) floats per pair.
// row bands per work-group of k_mips (see there) and the number of band groups of a view
inline int mips_band_group(int dimk) { return dimk <= 4 * MIP_KPW ? MIP_NB : 1; }
inline int mips_groups(int dimk, int dimi_v) {
    const int per = MIP_ROWS * mips_band_group(dimk);
    return (dimi_v + per - 1) / per;
}
inline size_t mips_tmp_floats(int dimk, int dimi_v, int dimj_v) {
    // (only the pass for stacks deeper than 4 * MIP_KPW slices still writes partial maxima: k_mips5 and k_mips_int merge theirs into the
    // MIPs.  A few floats all the same, so that a buffer exists.  Rows too long for k_mips5's 32-bit offsets -- 2^24 samples -- do not occur
    // with stacks this shallow in any caller's tiles; launch_mips refuses them.)
    if (dimk <= 4 * MIP_KPW) return 16;
    const size_t bands = std::max<size_t>(mips_groups(dimk, dimi_v), (dimi_v + 2 * MIP_ROWS - 1) / (2 * MIP_ROWS)), cblocks = (dimj_v + 63) / 64 + 1;
    return 2 * (bands * dimk * dimj_v + cblocks * (size_t)dimi_v * dimk);
}
// k_mips5: stacks whose xz maxima fit its LDS image, row offsets that fit the buffer instructions' 32-bit offsets
inline bool mips5_ok(int dimk, int pitch) { return dimk <= 4 * MIP_KPW && pitch < (1 << 24); }
inline bool mips_int_ok(int bytes, int dimk, int pitch, size_t slice) {
    const int per = 4 / bytes;  // samples per 32-bit word
    return (bytes == 1 || bytes == 2) && dimk <= 4 * MIP_KPW && pitch % per == 0 && slice % per == 0;
}
// launch geometry of the MIP kernel of a sample format: row bands per work-group, columns per work-group
inline int mips_fmt_bands(int bytes, int dimk) { return bytes == 4 ? mips_band_group(dimk) : (bytes == 2 ? IntTiles<2>::NB : IntTiles<1>::NB); }
inline int mips_fmt_width(int bytes) { return bytes == 4 ? 64 : (bytes == 2 ? IntTiles<2>::W : IntTiles<1>::W); }

// `beside_chain`: a lag chain of an earlier group runs while this pass does -- the pass then keeps TWO work-groups per compute
// unit instead of four (half of the LDS and of the registers stay free for the chain's work-groups, which otherwise wait for whole
// work-groups of the pass to retire and see none of the memory system: beside a pass at four the xy forward transform of 56 pairs
// took 2.3 ms instead of 0.24 -- profiles/r05_ncc_timeline_default.txt, r05_ncc_wpe.txt, r05_ncc_sched.txt: 5.8-5.9 ms per 112
// pairs at two, 6.1-6.3 at three, 6.3 at one)
int launch_mips(hipStream_t s, const float* A, const float* B, const float* const* tab, int np, size_t pstride, int dimk, int dimi_v, int dimj_v,
                size_t slice, int pitch, int ai0, int aj0, float* xy1, float* xz1, float* yz1, float* xy2, float* xz2, float* yz2, float* tmp,
                hipEvent_t xy_done = nullptr,  // recorded when the MIPs of k_mips5 / k_mips_int are final (deep stacks: the xy MIPs)
                TileFmt fmt = TileFmt(), float* xyT = nullptr, size_t tstride = 0, bool beside_chain = false) {
    const int nb = mips_fmt_bands(fmt.bytes, dimk), wcol = mips_fmt_width(fmt.bytes);
    const int bands = (dimi_v + MIP_ROWS * nb - 1) / (MIP_ROWS * nb);
    const int cblocks = (dimj_v + (aj0 & (wcol - 1)) + wcol - 1) / wcol;  // (band groups, column blocks aligned to the tile rows)
    const dim3 grid(cblocks, bands, 2 * np);   // (the deep-stack pass; k_mips5 and k_mips_int enumerate the same patches from a one-dimensional grid)
    const size_t n_xz = (size_t)dimi_v * dimk, n_yz = (size_t)dimj_v * dimk;
    auto zero_mips = [&]() {  // the passes that merge their row / column maxima into the MIPs with atomic maxima start from zero
        hipLaunchKernelGGL(k_mips_zero, dim3((unsigned)std::min<size_t>((n_xz + n_yz + 255) / 256, 64), 2 * np), dim3(256), 0, s, pstride, n_xz, n_yz, xz1,
                           yz1, xz2, yz2);
        return launch_check("k_mips_zero");
    };
    if (fmt.bytes != 4) {
        MI_REQUIRE(mips_int_ok(fmt.bytes, dimk, pitch, slice), "integer tiles: stacks of up to %d slices, rows of whole 32-bit words", 4 * MIP_KPW);
        const unsigned char* a8 = reinterpret_cast<const unsigned char*>(A);
        const unsigned char* b8 = reinterpret_cast<const unsigned char*>(B);
        const unsigned char* const* t8 = reinterpret_cast<const unsigned char* const*>(tab);
        MI_TRY(zero_mips());
        const dim3 grid_i((unsigned)((bands * 2 * np + 7) / 8 * 8 * cblocks));
        if (fmt.bytes == 2)
            hipLaunchKernelGGL(k_mips_int<2>, grid_i, dim3(256), 0, s, a8, b8, t8, pstride, dimk, dimi_v, dimj_v, slice, pitch, ai0, aj0, fmt.scale, xy1,
                               xz1, yz1, xy2, xz2, yz2, xyT, tstride, cblocks, bands, 2 * np);
        else
            hipLaunchKernelGGL(k_mips_int<1>, grid_i, dim3(256), 0, s, a8, b8, t8, pstride, dimk, dimi_v, dimj_v, slice, pitch, ai0, aj0, fmt.scale, xy1,
                               xz1, yz1, xy2, xz2, yz2, xyT, tstride, cblocks, bands, 2 * np);
        MI_TRY(launch_check("k_mips_int"));
        if (xy_done) MI_HIP(hipEventRecord(xy_done, s));
        return MI_OK;
    }
    static const int knock = [] {
        const char* e = MI_PROBE_ENV("MI_NCC_MIPS_KNOCK");
        return e ? std::atoi(e) : 0;
    }();
    if (mips5_ok(dimk, pitch)) {
        static const int wpe_env = [] { const char* e = MI_PROBE_ENV("MI_NCC_MIPS_WPE"); return e ? std::atoi(e) : 0; }();
        static const int wpe_beside = [] { const char* e = MI_PROBE_ENV("MI_NCC_MIPS_WPE_BESIDE"); return e ? std::atoi(e) : 0; }();
        const int wpe = beside_chain ? (wpe_beside ? wpe_beside : (wpe_env ? wpe_env : 2)) : (wpe_env ? wpe_env : 4);
        MI_TRY(zero_mips());
        static const int remap = [] { const char* e = MI_PROBE_ENV("MI_NCC_MIPS_NOREMAP"); return e && std::atoi(e) != 0 ? 0 : 1; }();
        const int row_groups = bands * 2 * np;
        const dim3 grid1((unsigned)((row_groups + 7) / 8 * 8 * cblocks));
#define MI_LAUNCH_MIPS5(W)                                                                                                                        \
    hipLaunchKernelGGL(k_mips5<W>, grid1, dim3(256), 0, s, A, B, tab, pstride, dimk, dimi_v, dimj_v, slice, pitch, ai0, aj0, xy1, xz1, yz1, xy2, xz2, yz2, \
                       knock, xyT, tstride, cblocks, bands, 2 * np, remap)
        if (wpe == 1) MI_LAUNCH_MIPS5(1);
        else if (wpe == 2) MI_LAUNCH_MIPS5(2);
        else if (wpe == 3) MI_LAUNCH_MIPS5(3);
        else MI_LAUNCH_MIPS5(4);
#undef MI_LAUNCH_MIPS5
        MI_TRY(launch_check("k_mips5"));
        if (xy_done) MI_HIP(hipEventRecord(xy_done, s));
        return MI_OK;
    }
    MI_REQUIRE(dimk > 4 * MIP_KPW, "compute_3_MIPs: rows of %d samples are too long for the MIP pass", pitch);
    // deeper stacks: the round-3 pass with its partial maxima and their reductions
    float* yz_tmp = tmp;
    float* xz_tmp = tmp + 2 * (size_t)np * bands * dimk * dimj_v;
    const size_t lds = sizeof(float) * MIP_ROWS * (size_t)dimk * mips_band_group(dimk);  // (k_mips: xzp)
    const bool via_lds = lds <= 32 * 1024;
    if (!via_lds) {  // very deep stacks: the xz MIPs are merged with atomicMax and must start at 0 (libcrossmips.cpp:319-337)
        MI_HIP(hipMemset2DAsync(xz1, sizeof(float) * (pstride ? pstride : 1), 0, sizeof(float) * (size_t)dimi_v * dimk, np, s));
        MI_HIP(hipMemset2DAsync(xz2, sizeof(float) * (pstride ? pstride : 1), 0, sizeof(float) * (size_t)dimi_v * dimk, np, s));
    }
    hipLaunchKernelGGL(k_mips, grid, dim3(256), via_lds ? lds : 0, s, A, B, tab, pstride, dimk, dimi_v, dimj_v, slice, pitch, ai0, aj0, xy1, xz1, yz1,
                       xy2, xz2, yz2, yz_tmp, via_lds ? xz_tmp : (float*)nullptr, knock, xyT, tstride);
    MI_TRY(launch_check("k_mips"));
    if (xy_done) MI_HIP(hipEventRecord(xy_done, s));
    hipLaunchKernelGGL(k_mips_yz, dim3((dimk * dimj_v + 255) / 256, 2 * np), dim3(256), 0, s, yz_tmp, pstride, bands, dimk, dimj_v, yz1, yz2);
    MI_TRY(launch_check("k_mips_yz"));
    if (via_lds) {
        hipLaunchKernelGGL(k_mips_xz, dim3((dimk * dimi_v + 255) / 256, 2 * np), dim3(256), 0, s, xz_tmp, pstride, cblocks, dimk, dimi_v, xz1, xz2);
        MI_TRY(launch_check("k_mips_xz"));
    }
    return MI_OK;
}

// one wave per tile (blockIdx.y = MIP of the plane): the tile is staged in LDS with coalesced loads, then lane 0 adds its 1024
// pixels in the reference's row-major order with a FLOAT running sum (bit-identical by construction)
__global__ __launch_bounds__(64) void k_tile_sums(const float* __restrict__ img1, const float* __restrict__ img2, size_t pstride, int height,
                                                  int width, float* __restrict__ ps1, float* __restrict__ ps2) {
    __shared__ float tile[TILE * TILE];
    const float* img = (blockIdx.y ? img2 : img1) + (size_t)blockIdx.z * pstride;
    float* ps = (blockIdx.y ? ps2 : ps1) + (size_t)blockIdx.z * pstride;
    const int pw = width / TILE, t = blockIdx.x;
    const int ti = t / pw, tj = t - ti * pw;
    const float* p = img + (size_t)ti * TILE * width + tj * TILE;
    const int lane = threadIdx.x, half = lane >> 5, col = lane & 31;
#pragma unroll
    for (int l = 0; l < TILE; l += 2) tile[(l + half) * TILE + col] = p[(size_t)(l + half) * width + col];
    __syncthreads();
    if (lane == 0) {
        float s = 0.0f;
#pragma unroll 32
        for (int i = 0; i < TILE * TILE; ++i) s += tile[i];
        ps[t] = s;
    }
}

template <int NT = NCC_THREADS>
__device__ __forceinline__ double block_sum(double v, double* sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
    for (int w = 0; w < NT / 64; ++w) r += sh[w];
    return r;
}

// sum of mip over rows [r0,r0+nr) x cols [c0,c0+nc) (per-lane partial): interior tiles from ps, border
// pixels directly (compute_funcs.cu:1186-1262); falls back to all pixels when ps == nullptr
// ------------------------------------------------------------------------------------------------ summed-area tables
// Everything of compute_NCC (compute_funcs.cu:1163-1292) except the cross term  sum f*t  depends on ONE image window:
//   mean'  = (sum of float tile sums inside the window + border pixels) / n        (the reference's means, :1186-1272)
//   F      = sum (f - mean')^2 = Q - 2 g' P + n g'^2   with  P = sum (f - c0), Q = sum (f - c0)^2, g' = mean' - c0
//   num    = sum f (t - tmean') = cross - tmean' * sum f
// so per MIP three fp64 summed-area tables (P, Q over the pixels shifted by the global mean c0, TS over the float tile sums)
// give all of them in O(1) per shift, and the NCC kernel only accumulates the cross term.
__device__ __forceinline__ double rect(const double* __restrict__ S, int w1, int r0, int c0, int nr, int nc) {
    return S[(size_t)(r0 + nr) * w1 + c0 + nc] - S[(size_t)r0 * w1 + c0 + nc] - S[(size_t)(r0 + nr) * w1 + c0] + S[(size_t)r0 * w1 + c0];
}

struct SatView {
    const double* P;   // (dimu+1) x (dimv+1): sum of (f - c0)
    const double* Q;   // same shape: sum of (f - c0)^2
    const double* TS;  // (ph+1) x (pw+1): sum of the float tile sums (nullptr when the MIP has no full tile)
    const double* c0;  // global mean of the MIP
    int w1;            // dimv + 1
    __device__ __forceinline__ double rectP(int r0, int c0_, int nr, int nc) const { return rect(P, w1, r0, c0_, nr, nc); }
    __device__ __forceinline__ double rectQ(int r0, int c0_, int nr, int nc) const { return rect(Q, w1, r0, c0_, nr, nc); }
};

// The same tables kept only where the window statistics ever look.  Shift (u, v) pairs the windows [max(u,0), dimu - max(-u,0)) x
// [max(v,0), ...): every row index a table is read at lies within E of an end of the axis (E = largest |shift|), the tile-aligned
// ones within E + 31.  Along the LONG axis of a MIP (2048 of 2048 x 307) that is a small band at either end: the tables are stored
// over logical coordinates (a = long axis, b = short axis) for a in [0, B] and [n_long - B, n_long] only -- 2 (B + 1) rows
// instead of n_long + 1 (C5: 216 of 2049) -- and are built from chunk-parallel column sums instead of two full-table passes.
struct BandView {
    const double* P;
    const double* Q;
    const double* TS;
    const double* c0;
    int n_long, n_short, B, long_is_u;  // B >= n_long: every row is kept
    __device__ __forceinline__ int row(int a) const { return (B >= n_long || a <= B) ? a : a - (n_long - B) + B + 1; }
    __device__ __forceinline__ double at(const double* S, int a, int b) const { return S[(size_t)row(a) * (n_short + 1) + b]; }
    __device__ __forceinline__ double rectS(const double* S, int r0, int c0_, int nr, int nc) const {
        const int a0 = long_is_u ? r0 : c0_, b0 = long_is_u ? c0_ : r0, na = long_is_u ? nr : nc, nb = long_is_u ? nc : nr;
        return at(S, a0 + na, b0 + nb) - at(S, a0, b0 + nb) - at(S, a0 + na, b0) + at(S, a0, b0);
    }
    __device__ __forceinline__ double rectP(int r0, int c0_, int nr, int nc) const { return rectS(P, r0, c0_, nr, nc); }
    __device__ __forceinline__ double rectQ(int r0, int c0_, int nr, int nc) const { return rectS(Q, r0, c0_, nr, nc); }
};

// window statistics of one MIP: mean' (reference flavour), sum f, sum (f - mean')^2
template <class View>
__device__ void window_stats(const View& sv, int dimu, int dimv, int r0, int c0, int nr, int nc, double* mean, double* sumf, double* ssd) {
    const double n = (double)nr * (double)nc, cm = *sv.c0;
    const double P = sv.rectP(r0, c0, nr, nc), Q = sv.rectQ(r0, c0, nr, nc);
    const double sf = P + n * cm;
    double m = sf / n;
    if (sv.TS && dimu >= TILE && dimv >= TILE) {
        int su = (r0 + TILE - 1) / TILE * TILE, tsv = (c0 + TILE - 1) / TILE * TILE;
        int eu = (r0 + nr) / TILE * TILE, ev = (c0 + nc) / TILE * TILE;
        if (su < eu && tsv < ev) {
            const int pw = dimv / TILE;
            const double tiles = rect(sv.TS, pw + 1, su / TILE, tsv / TILE, (eu - su) / TILE, (ev - tsv) / TILE);
            const double nreg = (double)(eu - su) * (double)(ev - tsv);
            const double region = sv.rectP(su, tsv, eu - su, ev - tsv) + nreg * cm;
            m = (tiles + (sf - region)) / n;  // float tile sums + border pixels, like the reference
        }
    }
    const double g = m - cm;
    *mean = m;
    *sumf = sf;
    *ssd = Q - 2.0 * g * P + n * g * g;
}

// pixel sums of both MIPs of a plane in MEAN_PARTS slices each (blockIdx.y = MIP), for the global means
constexpr int MEAN_PARTS = 64;
__global__ __launch_bounds__(256) void k_mip_partial(const float* __restrict__ m1, const float* __restrict__ m2, size_t pstride, size_t sstride,
                                                     size_t n, double* __restrict__ part) {
    __shared__ double sh[4];
    const float* m = (blockIdx.y ? m2 : m1) + (size_t)blockIdx.z * pstride;
    part += (size_t)blockIdx.z * sstride;
    const size_t per = (n + MEAN_PARTS - 1) / MEAN_PARTS, lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    double acc = 0.0;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) acc += (double)m[i];
    acc = block_sum<256>(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.y * MEAN_PARTS + blockIdx.x] = acc;
}

// global mean c0 of each of the two MIPs of a plane (blockIdx.x = 0 / 1) from the slices' sums (fixed order) and the table of
// its float tile sums
constexpr int kMeanLdsEntries = 4096;  // 32 KB of doubles: tile-sum tables of planes up to e.g. 2048 x 2016 pixels
__global__ __launch_bounds__(1024) void k_mip_mean(const double* __restrict__ part, size_t pstride, size_t sstride, int dimu, int dimv,
                                                   const float* __restrict__ ps1, const float* __restrict__ ps2, double* __restrict__ c0a,
                                                   double* __restrict__ c0b, double* __restrict__ ts1, double* __restrict__ ts2) {
    const size_t po = (size_t)blockIdx.z * pstride, so = (size_t)blockIdx.z * sstride;
    part += so; c0a += so; c0b += so;
    const float* ps = blockIdx.x ? ps2 : ps1;
    if (ps) ps += po;
    double* ts = (blockIdx.x ? ts2 : ts1) + so;
    if (threadIdx.x == 0) {
        double acc = 0.0;
        for (int k = 0; k < MEAN_PARTS; ++k) acc += part[blockIdx.x * MEAN_PARTS + k];
        *(blockIdx.x ? c0b : c0a) = acc / ((double)dimu * (double)dimv);
    }
    const int ph = dimu / TILE, pw = dimv / TILE;
    if (ps && ph * pw > 0) {
        // (ph+1) x (pw+1) inclusive table; a few hundred entries: one lane per row, then one per column.  The two running sums
        // are chains of dependent read-modify-writes: through global memory they cost ~1.5 us per step (64 steps: 0.1 ms for a
        // kernel that does nothing else), so the table is built in LDS when it fits and written out once
        const int w1 = pw + 1, n = (ph + 1) * w1;
        __shared__ double tl[kMeanLdsEntries];
        double* t = n <= kMeanLdsEntries ? tl : ts;
        for (int i = threadIdx.x; i < n; i += 1024) t[i] = 0.0;
        __syncthreads();
        for (int r = threadIdx.x; r < ph; r += 1024) {
            double run = 0.0;
            for (int c = 0; c < pw; ++c) { run += (double)ps[r * pw + c]; t[(r + 1) * w1 + c + 1] = run; }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < pw; c += 1024) {
            double run = 0.0;
            for (int r = 0; r < ph; ++r) { run += t[(r + 1) * w1 + c + 1]; t[(r + 1) * w1 + c + 1] = run; }
        }
        if (t != ts) {
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += 1024) ts[i] = tl[i];
        }
    }
}

__device__ __forceinline__ double wave_inclusive_scan(double v) {
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// row pass: one wave per (row, MIP): P/Q[(i+1)][j+1] = prefix along j of (f - c0), (f - c0)^2; row 0 and column 0 are zero
__global__ __launch_bounds__(64) void k_sat_rows(const float* __restrict__ m1, const float* __restrict__ m2, size_t pstride, size_t sstride,
                                                 int dimu, int dimv, const double* __restrict__ c0a, const double* __restrict__ c0b,
                                                 double* __restrict__ P1, double* __restrict__ Q1, double* __restrict__ P2,
                                                 double* __restrict__ Q2) {
    const size_t so = (size_t)blockIdx.z * sstride;
    const float* m = (blockIdx.y ? m2 : m1) + (size_t)blockIdx.z * pstride;
    double* P = (blockIdx.y ? P2 : P1) + so;
    double* Q = (blockIdx.y ? Q2 : Q1) + so;
    const double cm = (blockIdx.y ? c0b : c0a)[so];
    const int w1 = dimv + 1, lane = threadIdx.x;
    const int i = blockIdx.x;  // 0 .. dimu (row i of the table; table row 0 is all zero)
    if (i == 0) {
        for (int j = lane; j < w1; j += 64) { P[j] = 0.0; Q[j] = 0.0; }
        return;
    }
    const float* row = m + (size_t)(i - 1) * dimv;
    double cp = 0.0, cq = 0.0;
    if (lane == 0) { P[(size_t)i * w1] = 0.0; Q[(size_t)i * w1] = 0.0; }
    for (int j0 = 0; j0 < dimv; j0 += 64) {
        const int j = j0 + lane;
        const double g = j < dimv ? (double)row[j] - cm : 0.0;
        const double sp = wave_inclusive_scan(g) + cp, sq = wave_inclusive_scan(g * g) + cq;
        if (j < dimv) { P[(size_t)i * w1 + j + 1] = sp; Q[(size_t)i * w1 + j + 1] = sq; }
        cp = __shfl(sp, 63, 64);
        cq = __shfl(sq, 63, 64);
    }
}

// column pass, in place: a wave owns 64 neighbouring columns of one table (blockIdx.y) and walks down the rows with a running
// sum per lane -- every access is a 512-byte row segment; the loads of UNR rows are in flight together (they do not depend on
// the running sums).  The first version gave each wave ONE column (lanes = rows, stride of a whole table row between lanes):
// 88 us per C5 pair, more than the six MIPs.
__global__ __launch_bounds__(64) void k_sat_cols(size_t sstride, int dimu, int dimv, double* __restrict__ P1, double* __restrict__ Q1,
                                                 double* __restrict__ P2, double* __restrict__ Q2) {
    double* S = (blockIdx.y == 0 ? P1 : (blockIdx.y == 1 ? Q1 : (blockIdx.y == 2 ? P2 : Q2))) + (size_t)blockIdx.z * sstride;
    const int w1 = dimv + 1, j = blockIdx.x * 64 + threadIdx.x + 1;
    if (j > dimv) return;
    constexpr int UNR = 8;
    double run = 0.0;
    double* p = S + (size_t)w1 + j;  // table row 1
    int i = 1;
    for (; i + UNR - 1 <= dimu; i += UNR, p += (size_t)UNR * w1) {
        double v[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) v[q] = p[(size_t)q * w1];
#pragma unroll
        for (int q = 0; q < UNR; ++q) { run += v[q]; p[(size_t)q * w1] = run; }
    }
    for (; i <= dimu; ++i, p += w1) { run += *p; *p = run; }
}

// NCC cross terms, register-blocked.  A block is (4 u) x (8 v) shifts; a work-group takes a GROUP of up to 8 blocks that share
// u0 (a "row" of the shift map), a chunk of m2 rows and a segment of <= 512 m2 columns: wave w owns block w.  The m2 rows and
// the window of m1 rows all the group's blocks need (columns from the smallest v0 on, zero outside the MIP) are staged in LDS
// ONCE for the eight blocks with unit-stride loads -- staging, not the fp64 pipe, was the bound when every block staged its
// own window.  A lane then owns 4 neighbouring m2 columns of a row and, per u, the 11 m1 values its 8 shifts pair them with:
// thirteen 16-byte LDS reads feed 128 fp64 FMAs (exact fp32 products accumulated in fp64).  Zeros outside the MIP restrict
// every shift's sum to its own window.  Each wave reduces its own 32 sums (no work-group reduction).  Groups come from the
// regular grid (full maps: group = u-block, wave = v-block) or from a list (the "missing entries" of the neighbourhood
// refinement, gpu_NCC_miss).  The chunks' partial sums are added up in a fixed order by k_ncc_finish, which also applies the
// window statistics from the summed-area tables: deterministic, whatever the launch geometry.  Replaces gpu_NCC_map /
// gpu_NCC_miss (compute_funcs.cu:730-935).
constexpr int BU = 4, BV = 8, BC = 4, GW = 8, BLK_THREADS = 64 * GW;
// groups (list mode): 2 + GW ints each: {u0, number of blocks nb <= GW, v0[0] <= v0[1] <= ... (multiples of BV apart)}
__global__ __launch_bounds__(BLK_THREADS) void k_ncc_blk(const float* __restrict__ m1, const float* __restrict__ m2, int dimu, int dimv, int du,
                                                         int dv, int nvb, const int* __restrict__ groups, int n_groups, int rows_per_chunk,
                                                         int R, int seg_w, int nseg, int pitch1, double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int u0, nb, vmin, voff;  // voff: this wave's v0 - vmin
    if (groups) {
        const int* g = groups + (size_t)blockIdx.x * (2 + GW);
        u0 = g[0];
        nb = g[1];
        vmin = g[2];
        voff = g[2 + min(wave, nb - 1)] - vmin;
    } else {  // group = (u-block, run of GW v-blocks)
        const int nvg = (nvb + GW - 1) / GW, ub = (int)blockIdx.x / nvg, vg = (int)blockIdx.x - ub * nvg;
        u0 = ub * BU - du;
        nb = min(GW, nvb - vg * GW);
        vmin = vg * GW * BV - dv;
        voff = min(wave, nb - 1) * BV;
    }
    // chunk = (row chunk, column segment of seg_w m2 columns): wide MIPs are cut so that many rows fit one LDS stage
    const int rc = (int)blockIdx.y / nseg, sg = (int)blockIdx.y - rc * nseg;
    const int r_begin = rc * rows_per_chunk, r_end = min(dimu, r_begin + rows_per_chunk);
    const int cbeg = sg * seg_w, cw = min(seg_w, dimv - cbeg);  // m2 columns [cbeg, cbeg + cw)
    const int quads = (cw + BC - 1) / BC, pitch2 = seg_w;        // seg_w and pitch1 are multiples of BC
    float* l1 = lds;                            // R + BU - 1 rows of m1: l1[j][x] = m1[rb + u0 + j][cbeg + vmin + x]
    float* l2 = lds + (R + BU - 1) * pitch1;    // R rows of m2
    double acc[BU][BV];
#pragma unroll
    for (int a = 0; a < BU; ++a)
#pragma unroll
        for (int b = 0; b < BV; ++b) acc[a][b] = 0.0;
    for (int rb = r_begin; rb < r_end; rb += R) {
        const int nrow = min(R, r_end - rb);
        __syncthreads();
        {   // staging: a wave per row (no index division), four independent 64-column loads per lane in flight
            constexpr int UNR = 4;
            for (int j = wave; j < nrow + BU - 1; j += GW) {
                const int r1 = rb + u0 + j;
                const bool row_ok = r1 >= 0 && r1 < dimu;
                const float* src = m1 + (size_t)(row_ok ? r1 : 0) * dimv + cbeg + vmin;
                float* dst = l1 + j * pitch1;
                for (int x0 = lane; x0 < pitch1; x0 += 64 * UNR) {
                    float v[UNR];
#pragma unroll
                    for (int q = 0; q < UNR; ++q) {
                        const int x = x0 + 64 * q, c1 = cbeg + vmin + x;
                        v[q] = (row_ok && x < pitch1 && c1 >= 0 && c1 < dimv) ? src[x] : 0.0f;
                    }
#pragma unroll
                    for (int q = 0; q < UNR; ++q)
                        if (x0 + 64 * q < pitch1) dst[x0 + 64 * q] = v[q];
                }
            }
            for (int j = wave; j < nrow; j += GW) {
                const float* src = m2 + (size_t)(rb + j) * dimv + cbeg;
                float* dst = l2 + j * pitch2;
                for (int x0 = lane; x0 < pitch2; x0 += 64 * UNR) {
                    float v[UNR];
#pragma unroll
                    for (int q = 0; q < UNR; ++q) v[q] = x0 + 64 * q < cw ? src[x0 + 64 * q] : 0.0f;
#pragma unroll
                    for (int q = 0; q < UNR; ++q)
                        if (x0 + 64 * q < pitch2) dst[x0 + 64 * q] = v[q];
                }
            }
        }
        __syncthreads();
        if (wave < nb) {
            for (int it = lane; it < nrow * quads; it += 64) {
                const int rr = it / quads, c = (it - rr * quads) * BC;
                const float4 tq = *reinterpret_cast<const float4*>(l2 + rr * pitch2 + c);
                const double t[BC] = {(double)tq.x, (double)tq.y, (double)tq.z, (double)tq.w};
#pragma unroll
                for (int a = 0; a < BU; ++a) {
                    const float4* row = reinterpret_cast<const float4*>(l1 + (rr + a) * pitch1 + voff + c);
                    const float4 f0 = row[0], f1 = row[1], f2 = row[2];
                    const double f[12] = {(double)f0.x, (double)f0.y, (double)f0.z, (double)f0.w, (double)f1.x, (double)f1.y,
                                          (double)f1.z, (double)f1.w, (double)f2.x, (double)f2.y, (double)f2.z, (double)f2.w};
#pragma unroll
                    for (int b = 0; b < BV; ++b)
#pragma unroll
                        for (int x = 0; x < BC; ++x) acc[a][b] = fma(f[x + b], t[x], acc[a][b]);
                }
            }
        }
    }
    if (wave >= nb) return;
    // the wave's own 32 sums: shuffle tree, lane 0 stores
    double* dst = partial + (((size_t)blockIdx.y * n_groups + blockIdx.x) * GW + wave) * (BU * BV);
#pragma unroll
    for (int a = 0; a < BU; ++a)
#pragma unroll
        for (int b = 0; b < BV; ++b) {
            double v = acc[a][b];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) dst[a * BV + b] = v;
        }
}

// one lane per requested entry: cross term = sum of the chunks' partials (fixed order), then the NCC value of
// compute_NCC (compute_funcs.cu:1163-1292).  entries == nullptr: the full (2du+1) x (2dv+1) map in row-major order;
// else entry e = {u, v, slot in the partials of a chunk ((group * GW + wave) * 32 + a * 8 + b), output slot}
__global__ __launch_bounds__(256) void k_ncc_finish(const double* __restrict__ partial, int n_chunks, int n_groups,
                                                    const int* __restrict__ entries, int n_entries, int dimu, int dimv, int du, int dv,
                                                    SatView s1, SatView s2, float* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries) return;
    int u, v, pidx, slot;
    if (entries) {
        u = entries[4 * e];
        v = entries[4 * e + 1];
        pidx = entries[4 * e + 2];
        slot = entries[4 * e + 3];
    } else {
        const int W = 2 * dv + 1, iu = e / W, iv = e - iu * W;
        u = iu - du;
        v = iv - dv;
        const int nvb = (W + BV - 1) / BV, slots = (nvb + GW - 1) / GW * GW;  // block slots per u-block: its groups x GW
        pidx = (((iu / BU) * slots + iv / BV) * BU + iu % BU) * BV + iv % BV;
        slot = e;
    }
    const int nr = dimu - abs(u), nc = dimv - abs(v);
    if (nr <= 0 || nc <= 0) { out[slot] = __int_as_float(0x7fc00000); return; }  // reference: empty loops, 0/0
    double cr = 0.0;
#pragma unroll 8
    for (int ch = 0; ch < n_chunks; ++ch) cr += partial[(size_t)ch * n_groups * (GW * BU * BV) + pidx];  // fixed order
    double fm, sf, F1, tm, st, F2;
    window_stats(s1, dimu, dimv, max(u, 0), max(v, 0), nr, nc, &fm, &sf, &F1);
    window_stats(s2, dimu, dimv, max(-u, 0), max(-v, 0), nr, nc, &tm, &st, &F2);
    (void)fm; (void)st;
    const double num = cr - tm * sf;
    // a window without variance: the reference's two-pass sums are exactly 0 there and it returns 0/0 = NaN
    // (compute_funcs.cu:1277-1290); the table differences are exact for such data too, but the cross term is not
    out[slot] = (F1 > 0.0 && F2 > 0.0) ? (float)(num / sqrt(F1 * F2)) : __int_as_float(0x7fc00000);
}

// ------------------------------------------------------------------------------------------------ host logic
inline int imin(int a, int b) { return a < b ? a : b; }
inline int imax(int a, int b) { return a > b ? a : b; }

// compute_MAX_ind (compute_funcs.cu:1294-1305)
int argmax_first(const float* v, int len) {
    float best = v[0];
    int ind = 0;
    for (int i = 0; i < len; ++i)
        if (v[i] > best) { best = v[i]; ind = i; }
    return ind;
}

// half width of the peak at `ind` along one direction of the (2*wR1+1) x (2*wR2+1) window
// (compute_NCC_width, compute_funcs.cu:160-282).  `second_bound` is the limit of the slope-projection
// loops, which the reference takes from the horizontal range for both directions (:252,267); where it
// exceeds this direction's own range the reference indexes outside the window, so it is clamped.
//
// `tight` (optional) receives the smallest margin, in NCC units, by which any comparison or rounding step below was decided: the
// caller re-decides on exactly recomputed entries when it is below the resolution of its map values (mi::ncc_margin()).
inline void note_margin(float* tight, float a, float b) {
    if (!tight) return;
    const float d = std::fabs(a - b);  // NaN operands: the comparison is false whatever their last bits are
    if (d < *tight) *tight = d;
}
// q is about to be floored; dq/dNCC ~ scale: the margin is the distance to the next integer in NCC units
inline void note_rounding(float* tight, float q, float scale) {
    if (!tight || !(q == q) || !(scale > 0.0f)) return;
    const float f = std::fmin(q - std::floor(q), std::ceil(q) - q);
    const float d = (q == std::floor(q) ? 0.0f : f) / scale;
    if (d < *tight) *tight = d;
}
int peak_half_width(const mi_ncc_params& P, const float* M, int ind, int step, int range, int second_bound, float* tight = nullptr) {
#pragma clang fp contract(off)  // the float expression order below is part of the specification
    if (range < P.minDim_NCCmap) return P.INF_W;
    second_bound = imin(second_bound, range);
    const float thr = P.widthThr * M[ind];
    bool found = false;
    int w = 1;
    while (w <= range && !found) {
        note_margin(tight, M[ind - w * step], thr);
        if (M[ind - w * step] <= thr) found = true; else ++w;
    }
    found = false;
    while (w <= range && !found) {
        note_margin(tight, M[ind + w * step], thr);
        if (M[ind + w * step] <= thr) found = true; else ++w;
    }
    if (found) return w;
    float prec = M[ind - P.minPoints * step];
    int dist = P.minPoints + 1;
    while (dist <= second_bound && !found) {
        note_margin(tight, M[ind - dist * step], prec);
        if (M[ind - dist * step] >= prec) found = true;
        else { prec = M[ind - dist * step]; ++dist; }
    }
    if (dist < 2 * P.minPoints) w = P.INF_W;
    else {
        const float q = (float)((dist - 1) * (M[ind] - thr) / (M[ind] - prec));
        note_rounding(tight, q, (float)(dist - 1) / std::fabs(M[ind] - prec));
        w = (int)std::floor(q);
    }
    found = false;
    prec = M[ind + P.minPoints * step];
    dist = P.minPoints + 1;
    while (dist <= second_bound && !found) {
        note_margin(tight, M[ind + dist * step], prec);
        if (M[ind + dist * step] >= prec) found = true;
        else { prec = M[ind + dist * step]; ++dist; }
    }
    if (dist < 2 * P.minPoints) w = P.INF_W;
    else {
        const float q = (float)((dist - 1) * (M[ind] - thr) / (M[ind] - prec));
        note_rounding(tight, q, (float)(dist - 1) / std::fabs(M[ind] - prec));
        w = imin(imax(w, (int)std::floor(q)), P.INF_W - 1);
    }
    return w;
}

// compute_NCC_alignment (compute_funcs.cu:297-342)
void combine_axis(const mi_ncc_params& P, mi_ncc_descr* R, int ax, int d1, float p1, int w1, int d2, float p2, int w2, float* tight = nullptr) {
#pragma clang fp contract(off)
    if (w1 == 1) w1 = P.INF_W;
    if (w2 == 1) w2 = P.INF_W;
    if (w1 < P.INF_W) note_margin(tight, p1, P.maxThr);
    if (w2 < P.INF_W) note_margin(tight, p2, P.maxThr);
    const bool ok1 = p1 >= P.maxThr && w1 < P.INF_W, ok2 = p2 >= P.maxThr && w2 < P.INF_W;
    int d;
    float p;
    int w;
    if (ok1 && ok2) {
        if (std::abs(d1 - d2) < imin(w1, w2)) {
            const float mean = (p1 * d1 + p2 * d2) / (p1 + p2);
            if (d1 != d2) note_rounding(tight, mean + 0.5f, (float)std::abs(d1 - d2) / (p1 + p2));
            d = (int)std::floor((double)mean + 0.5);
            p = (p1 * p1 + p2 * p2) / (p1 + p2);
            w = imax(w1, w2);
        } else {
            if (tight) { const float m = std::fabs(p1 / w1 - p2 / w2) * (float)imin(w1, w2); if (m < *tight) *tight = m; }
            if (p1 / w1 > p2 / w2) { d = d1; p = p1; w = w1; }
            else { d = d2; p = p2; w = w2; }
        }
    } else if (ok1) { d = d1; p = p1; w = w1; }
    else if (ok2) { d = d2; p = p2; w = w2; }
    else { d = P.INV_COORD; p = P.UNR_NCC; w = P.INF_W; }
    R->coord[ax] = d;
    R->NCC_maxs[ax] = p;
    R->NCC_widths[ax] = w;
}

struct PlaneGeom {  // one of the three MIP planes
    int dimu, dimv;     // MIP extents
    int delayu, delayv; // search half ranges
    int wu, wv;         // window half extents (wRangeThr)
    size_t mip1, mip2, ps1, ps2, map;  // float offsets inside the workspace
    size_t sat;                        // double offset of this plane's summed-area tables inside Workspace::sat
    bool tiled;
};

// layout of one plane's tables (doubles): c0a, c0b | P1 | Q1 | P2 | Q2 | TS1 | TS2
struct SatLayout {
    size_t tab, ts, total;
    SatLayout(int dimu, int dimv) {
        tab = (size_t)(dimu + 1) * (dimv + 1);
        ts = (size_t)(dimu / TILE + 1) * (dimv / TILE + 1);
        total = 2 + 4 * tab + 2 * ts + 2 * MEAN_PARTS;  // c0a, c0b | P1 Q1 P2 Q2 | TS1 TS2 | partial pixel sums
    }
};

// builds tile sums (float, reference order), global means and the summed-area tables of both MIPs of a plane
// (of `np` pairs at once: pair q's MIPs / tile sums sit q * pstride floats, its tables q * sstride doubles behind pair 0's)
int prepare_plane(hipStream_t s, const float* m1, const float* m2, int dimu, int dimv, float* ps1, float* ps2, double* sat, SatView* v1,
                  SatView* v2, int np = 1, size_t pstride = 0, size_t sstride = 0) {
    const SatLayout L(dimu, dimv);
    const bool tiled = (dimu / TILE) * (dimv / TILE) > 0;
    double *c0a = sat, *c0b = sat + 1, *P1 = sat + 2, *Q1 = P1 + L.tab, *P2 = Q1 + L.tab, *Q2 = P2 + L.tab, *T1 = Q2 + L.tab, *T2 = T1 + L.ts;
    if (tiled) {
        const int nt = (dimu / TILE) * (dimv / TILE);
        hipLaunchKernelGGL(k_tile_sums, dim3(nt, 2, np), dim3(64), 0, s, m1, m2, pstride, dimu, dimv, ps1, ps2);
        MI_TRY(launch_check("k_tile_sums"));
    }
    double* part = T2 + L.ts;
    hipLaunchKernelGGL(k_mip_partial, dim3(MEAN_PARTS, 2, np), dim3(256), 0, s, m1, m2, pstride, sstride, (size_t)dimu * dimv, part);
    MI_TRY(launch_check("k_mip_partial"));
    hipLaunchKernelGGL(k_mip_mean, dim3(2, 1, np), dim3(1024), 0, s, part, pstride, sstride, dimu, dimv, tiled ? ps1 : nullptr,
                       tiled ? ps2 : nullptr, c0a, c0b, T1, T2);
    MI_TRY(launch_check("k_mip_mean"));
    hipLaunchKernelGGL(k_sat_rows, dim3(dimu + 1, 2, np), dim3(64), 0, s, m1, m2, pstride, sstride, dimu, dimv, c0a, c0b, P1, Q1, P2, Q2);
    MI_TRY(launch_check("k_sat_rows"));
    hipLaunchKernelGGL(k_sat_cols, dim3((dimv + 63) / 64, 4, np), dim3(64), 0, s, sstride, dimu, dimv, P1, Q1, P2, Q2);
    MI_TRY(launch_check("k_sat_cols"));
    *v1 = SatView{P1, Q1, tiled ? T1 : nullptr, c0a, dimv + 1};
    *v2 = SatView{P2, Q2, tiled ? T2 : nullptr, c0b, dimv + 1};
    return MI_OK;
}

// ------------------------------------------------------------------------------------------------ exact entries (two-pass form)
// Resolution of the map values the fast paths produce (summed-area tables + blocked / lag-transform cross terms): a decision
// whose operands are closer than this is re-taken on entries recomputed by k_ncc_exact.  MI_NCC_MARGIN overrides (tests).
inline float decision_margin() {
    static const float m = [] {
        const char* e = std::getenv("MI_NCC_MARGIN");
        return e ? (float)std::atof(e) : 4e-6f;
    }();
    return m;
}

// One work-group per listed entry {u, v, slot}: compute_NCC's own two-pass form (compute_funcs.cu:1163-1292) -- the window means
// first (the reference's flavour: float tile sums + border pixels, here from the tables, identical to 1e-16), then
//   num = sum f (t - tmean),  F1 = sum (f - fmean)^2,  F2 = sum (t - tmean)^2
// by direct fp64 summation over the window (no algebraic rearrangement, no cancellation), lanes and waves added in a fixed order.
__global__ __launch_bounds__(256) void k_ncc_exact(const float* __restrict__ m1, const float* __restrict__ m2, int dimu, int dimv, SatView s1, SatView s2,
                                                   const int* __restrict__ entries, float* __restrict__ out) {
    __shared__ double sh[4];
    const int u = entries[3 * blockIdx.x], v = entries[3 * blockIdx.x + 1], slot = entries[3 * blockIdx.x + 2];
    const int nr = dimu - abs(u), nc = dimv - abs(v);
    if (nr <= 0 || nc <= 0) { if (threadIdx.x == 0) out[slot] = __int_as_float(0x7fc00000); return; }
    const int r1 = max(u, 0), c1 = max(v, 0), r2 = max(-u, 0), c2 = max(-v, 0);
    double fm, sf, F1s, tm, st, F2s;
    window_stats(s1, dimu, dimv, r1, c1, nr, nc, &fm, &sf, &F1s);
    window_stats(s2, dimu, dimv, r2, c2, nr, nc, &tm, &st, &F2s);
    (void)sf; (void)st; (void)F1s; (void)F2s;
    double num = 0.0, F1 = 0.0, F2 = 0.0;
    const size_t n = (size_t)nr * nc;
    for (size_t e = threadIdx.x; e < n; e += 256) {
        const int i = (int)(e / nc), j = (int)(e - (size_t)i * nc);
        const double f = (double)m1[(size_t)(r1 + i) * dimv + c1 + j], t = (double)m2[(size_t)(r2 + i) * dimv + c2 + j];
        const double df = f - fm, dt = t - tm;
        num += f * dt;
        F1 += df * df;
        F2 += dt * dt;
    }
    num = block_sum<256>(num, sh);
    F1 = block_sum<256>(F1, sh);
    F2 = block_sum<256>(F2, sh);
    if (threadIdx.x == 0) out[slot] = (float)(num / sqrt(F1 * F2));  // flat window: 0 / 0 = NaN like the reference
}

// pinned host staging: pageable destinations make every small D2H / H2D a ~0.14 ms staged blit
struct PinnedBuf {
    void* p = nullptr;
    size_t bytes = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    int reserve(size_t n) {
        if (n <= bytes) return MI_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipHostMalloc(&p, n, hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; return fail(MI_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", n, hipGetErrorString(e)); }
        bytes = n;
        return MI_OK;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

struct Workspace {
    DevBuf buf;       // floats: MIPs | tile sums | maps | miss results
    DevBuf sat;       // doubles: per plane c0, P, Q, TS tables of both MIPs
    DevBuf mip_tmp;   // floats: per tile and row band, the column maxima the yz MIPs are reduced from
    SatView v1[3], v2[3];
    DevBuf list;      // ints: {u, v0, count, slot} groups of missing entries
    DevBuf partial[3]; // doubles: per plane, the row chunks' partial cross terms of a full map
    size_t floats = 0;
    int list_cap = 0;
    std::vector<int> host_groups, host_entries, host_slots;
    std::vector<long long> host_keys;
    PinnedBuf pin_groups, pin_res, pin_maps;
    DevBuf exact_list, exact_out;   // k_ncc_exact: {u, v, slot} entries and their values
    PinnedBuf pin_exact_in, pin_exact_out;
    int n_exact = 0;                // entries recomputed exactly for the current pair (statistics / tests)
};

// values[q] = the exact NCC of shift uv[2q], uv[2q+1] of a plane.  Synchronises `s`.
int exact_entries(hipStream_t s, const PlaneGeom& g, int plane, const float* d_base, Workspace& ws, const std::vector<int>& uv,
                  std::vector<float>& values) {
    const int n = (int)uv.size() / 2;
    values.assign(n, 0.0f);
    if (n == 0) return MI_OK;
    MI_TRY(ws.pin_exact_in.reserve(sizeof(int) * 3 * (size_t)n));
    MI_TRY(ws.pin_exact_out.reserve(sizeof(float) * (size_t)n));
    if (ws.exact_list.bytes < sizeof(int) * 3 * (size_t)n || ws.exact_out.bytes < sizeof(float) * (size_t)n) {
        MI_HIP(hipStreamSynchronize(s));
        MI_TRY(ws.exact_list.alloc(sizeof(int) * 3 * (size_t)n * 2));
        MI_TRY(ws.exact_out.alloc(sizeof(float) * (size_t)n * 2));
    }
    int* h = ws.pin_exact_in.as<int>();
    for (int q = 0; q < n; ++q) { h[3 * q] = uv[2 * q]; h[3 * q + 1] = uv[2 * q + 1]; h[3 * q + 2] = q; }
    MI_HIP(hipMemcpyAsync(ws.exact_list.p, h, sizeof(int) * 3 * (size_t)n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_ncc_exact, dim3(n), dim3(256), 0, s, d_base + g.mip1, d_base + g.mip2, g.dimu, g.dimv, ws.v1[plane], ws.v2[plane],
                       ws.exact_list.as<int>(), ws.exact_out.as<float>());
    MI_TRY(launch_check("k_ncc_exact"));
    MI_HIP(hipMemcpyAsync(ws.pin_exact_out.p, ws.exact_out.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    std::memcpy(values.data(), ws.pin_exact_out.p, sizeof(float) * (size_t)n);
    ws.n_exact += n;
    ncc_count(2, n);
    return MI_OK;
}

// compute_MAX_ind (compute_funcs.cu:1294-1305) decided on exact values: every entry within the decision margin of the largest
// value is recomputed by k_ncc_exact (once: `exact` remembers) before the first strict maximum is taken.  `to_uv(i, &u, &v)` maps
// an index of `v` to its shift.
template <class ToUV>
int argmax_exact(hipStream_t s, const PlaneGeom& g, int plane, const float* d_base, Workspace& ws, float* v, int len, std::vector<unsigned char>& exact,
                 ToUV to_uv, int* ind) {
    const float margin = decision_margin();
    for (int round = 0; round < 2; ++round) {
        const int best = argmax_first(v, len);
        if (!(v[best] == v[best])) { *ind = best; return MI_OK; }  // leading NaN: the reference keeps it whatever follows
        std::vector<int> uv, idx;
        int close = 0;
        for (int i = 0; i < len; ++i)
            if (v[i] == v[i] && v[best] - v[i] < margin) {
                ++close;
                if (!exact[i]) { int a, b; to_uv(i, &a, &b); uv.push_back(a); uv.push_back(b); idx.push_back(i); }
            }
        if (close < 2 || idx.empty()) { *ind = best; return MI_OK; }
        std::vector<float> val;
        MI_TRY(exact_entries(s, g, plane, d_base, ws, uv, val));
        for (size_t q = 0; q < idx.size(); ++q) { v[idx[q]] = val[q]; exact[idx[q]] = 1; }
    }
    *ind = argmax_first(v, len);
    return MI_OK;
}


// NCC values of one plane: the blocked cross terms + the finishing pass.  d_groups / d_entries == nullptr: the full map into
// d_out; else the listed groups {u0, nb, v0[GW]} and entries {u, v, partial index, output slot}.  `partial` grows as needed.
int ncc_launch(hipStream_t s, const float* m1, const float* m2, int dimu, int dimv, int delayu, int delayv, const SatView& v1, const SatView& v2,
               const int* d_groups, int n_groups, const int* d_entries, int n_entries, DevBuf& partial, float* d_out) {
    const int nvb = (2 * delayv + 1 + BV - 1) / BV;
    if (!d_groups) {
        n_groups = ((2 * delayu + 1 + BU - 1) / BU) * ((nvb + GW - 1) / GW);
        n_entries = (2 * delayu + 1) * (2 * delayv + 1);
    }
    if (n_groups <= 0 || n_entries <= 0) return MI_OK;
    // column segments of at most 512 m2 columns, row chunks so that about 768 work-groups of 8 waves exist
    const int nseg = (dimv + 511) / 512;
    const int seg_w = ((dimv + nseg - 1) / nseg + BC - 1) / BC * BC;
    int rchunks = (768 + n_groups * nseg - 1) / (n_groups * nseg);
    rchunks = imax(1, imin(rchunks, (dimu + 7) / 8));
    const int rows_per_chunk = (dimu + rchunks - 1) / rchunks;
    rchunks = (dimu + rows_per_chunk - 1) / rows_per_chunk;
    const int chunks = rchunks * nseg;
    // the m1 window of a group: its blocks start at most (GW - 1) * BV columns apart
    const int quads = seg_w / BC, pitch1 = seg_w + GW * BV, pitch2 = seg_w;
    // rows staged together: a few items per lane and stage, within 60 KB of LDS (two work-groups per CU)
    const int fit = ((60 * 1024) / (int)sizeof(float) - (BU - 1) * pitch1) / (pitch1 + pitch2);
    const int R = imax(1, imin(imin(rows_per_chunk, 1536 / quads), fit));
    const size_t lds = sizeof(float) * ((size_t)(R + BU - 1) * pitch1 + (size_t)R * pitch2);
    const size_t need = sizeof(double) * (size_t)chunks * n_groups * GW * BU * BV;
    if (partial.bytes < need) {
        MI_HIP(hipStreamSynchronize(s));  // an earlier launch of this stream may still read the old buffer
        MI_TRY(partial.alloc(need));
    }
    hipLaunchKernelGGL(k_ncc_blk, dim3(n_groups, chunks), dim3(BLK_THREADS), lds, s, m1, m2, dimu, dimv, delayu, delayv, nvb, d_groups, n_groups,
                       rows_per_chunk, R, seg_w, nseg, pitch1, partial.as<double>());
    MI_TRY(launch_check("k_ncc_blk"));
    hipLaunchKernelGGL(k_ncc_finish, dim3((n_entries + 255) / 256), dim3(256), 0, s, partial.as<double>(), chunks, n_groups, d_entries, n_entries,
                       dimu, dimv, delayu, delayv, v1, v2, d_out);
    return launch_check("k_ncc_finish");
}

// compute_Neighborhood (compute_funcs.cu:1324-1592): win = (2wu+1)x(2wv+1) window around the peak,
// re-centred up to maxIter times; entries exposed by a move are computed on the device.
int refine_neighbourhood(hipStream_t s, const mi_ncc_params& P, const float* map, const PlaneGeom& g, int plane, const float* d_base,
                         Workspace& ws,
                         std::vector<float>& win, int* du, int* dv, bool* failed, std::vector<unsigned char>* exact_out = nullptr) {
    const int H = 2 * g.wu + 1, W = 2 * g.wv + 1, Wm = 2 * g.delayv + 1, Hm = 2 * g.delayu + 1;
    std::vector<float> mapv(map, map + (size_t)Hm * Wm);
    std::vector<unsigned char> map_exact((size_t)Hm * Wm, 0);
    int ind_max = 0;
    MI_TRY(argmax_exact(s, g, plane, d_base, ws, mapv.data(), Hm * Wm, map_exact,
                        [&](int i, int* a, int* b) { *a = i / Wm - g.delayu; *b = i % Wm - g.delayv; }, &ind_max));
    map = mapv.data();
    const int initu = imin(imax(0, ind_max / Wm - g.wu), 2 * (g.delayu - g.wu));
    const int initv = imin(imax(0, ind_max % Wm - g.wv), 2 * (g.delayv - g.wv));
    MI_REQUIRE(initu >= 0 && initv >= 0, "CrossMIPs: negative index detected (initi)");
    win.assign((size_t)H * W, 0.0f);
    std::vector<unsigned char> win_exact((size_t)H * W, 0), old_exact;
    for (int r = 0; r < H; ++r) {
        std::memcpy(&win[(size_t)r * W], &map[(size_t)(initu + r) * Wm + initv], sizeof(float) * W);
        std::memcpy(&win_exact[(size_t)r * W], &map_exact[(size_t)(initu + r) * Wm + initv], W);
    }
    *du = initu - g.delayu + g.wu;
    *dv = initv - g.delayv + g.wv;
    ind_max = W * (ind_max / Wm - initu) + (ind_max % Wm - initv);
    const int ind_ref = W * g.wu + g.wv;
    std::vector<float> old;
    std::vector<int> miss;
    for (int it = 0; it < P.maxIter && ind_max != ind_ref; ++it) {
        const int deltau = ind_max / W - g.wu, deltav = ind_max % W - g.wv;
        old = win;
        old_exact = win_exact;
        *du += deltau;
        *dv += deltav;
        miss.clear();
        for (int r = 0; r < H; ++r)
            for (int c = 0; c < W; ++c) {
                const int sr = r + deltau, sc = c + deltav;
                if (sr >= 0 && sr < H && sc >= 0 && sc < W) {
                    win[(size_t)r * W + c] = old[(size_t)sr * W + sc];
                    win_exact[(size_t)r * W + c] = old_exact[(size_t)sr * W + sc];
                } else {
                    miss.push_back(r - g.wu + *du); miss.push_back(c - g.wv + *dv); miss.push_back(r * W + c);
                    win_exact[(size_t)r * W + c] = 0;
                }
            }
        const int n_miss = (int)miss.size() / 3;
        MI_REQUIRE(n_miss == H * W - (H - std::abs(deltau)) * (W - std::abs(deltav)), "CrossMIPs: incomplete NCC map in compute_Neighborhood");
        if (n_miss > 0) {
            // the missing entries are covered by 4 x 8 blocks of shifts anchored at the smallest missing (u, v); blocks of one
            // u-row form groups of up to GW blocks (one wave each) that share their staged rows
            int ub = miss[0], vb = miss[1];
            for (int q = 1; q < n_miss; ++q) { ub = imin(ub, miss[3 * q]); vb = imin(vb, miss[3 * q + 1]); }
            std::vector<long long>& keys = ws.host_keys;   // distinct blocks (bu << 20 | bv), sorted
            keys.clear();
            for (int q = 0; q < n_miss; ++q) keys.push_back((long long)((miss[3 * q] - ub) / BU) * (1 << 20) + (miss[3 * q + 1] - vb) / BV);
            std::sort(keys.begin(), keys.end());
            keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
            std::vector<int>& lst = ws.host_groups;        // groups {u0, nb, v0[GW]}
            lst.clear();
            std::vector<int>& slot_of = ws.host_slots;     // per block (in `keys` order): group * GW + wave
            slot_of.assign(keys.size(), 0);
            int n_groups = 0;
            for (size_t k = 0; k < keys.size();) {
                const int bu = (int)(keys[k] >> 20), bv0 = (int)(keys[k] & ((1 << 20) - 1));
                const size_t base = lst.size();
                lst.resize(base + 2 + GW, 0);
                int nb = 0;
                while (k < keys.size() && nb < GW && (int)(keys[k] >> 20) == bu && (int)(keys[k] & ((1 << 20) - 1)) - bv0 < GW) {
                    lst[base + 2 + nb] = vb + (int)(keys[k] & ((1 << 20) - 1)) * BV;
                    slot_of[k] = n_groups * GW + nb;
                    ++nb;
                    ++k;
                }
                lst[base] = ub + bu * BU;
                lst[base + 1] = nb;
                for (int w = nb; w < GW; ++w) lst[base + 2 + w] = lst[base + 2 + nb - 1];
                ++n_groups;
            }
            std::vector<int>& ent = ws.host_entries;       // entries {u, v, partial index, output slot}
            ent.clear();
            for (int q = 0; q < n_miss; ++q) {
                const int u = miss[3 * q], v = miss[3 * q + 1];
                const long long key = (long long)((u - ub) / BU) * (1 << 20) + (v - vb) / BV;
                const size_t k = std::lower_bound(keys.begin(), keys.end(), key) - keys.begin();
                ent.push_back(u); ent.push_back(v);
                ent.push_back((slot_of[k] * BU + (u - ub) % BU) * BV + (v - vb) % BV);
                ent.push_back(miss[3 * q + 2]);
            }
            const size_t ints = lst.size() + ent.size();
            if ((size_t)ws.list_cap < ints) {
                MI_HIP(hipStreamSynchronize(s));
                MI_TRY(ws.list.alloc(sizeof(int) * ints));
                ws.list_cap = (int)ints;
            }
            float* d_res = ws.buf.as<float>() + ws.floats;  // H*W result slots reserved behind the maps
            MI_TRY(ws.pin_groups.reserve(sizeof(int) * ints));
            MI_TRY(ws.pin_res.reserve(sizeof(float) * (size_t)H * W));
            std::memcpy(ws.pin_groups.p, lst.data(), sizeof(int) * lst.size());
            std::memcpy(ws.pin_groups.as<int>() + lst.size(), ent.data(), sizeof(int) * ent.size());
            MI_HIP(hipMemcpyAsync(ws.list.p, ws.pin_groups.p, sizeof(int) * ints, hipMemcpyHostToDevice, s));
            MI_TRY(ncc_launch(s, d_base + g.mip1, d_base + g.mip2, g.dimu, g.dimv, g.delayu, g.delayv, ws.v1[plane], ws.v2[plane],
                              ws.list.as<int>(), n_groups, ws.list.as<int>() + lst.size(), n_miss, ws.partial[plane], d_res));
            const float* res = ws.pin_res.as<float>();
            MI_HIP(hipMemcpyAsync(ws.pin_res.p, d_res, sizeof(float) * H * W, hipMemcpyDeviceToHost, s));
            MI_HIP(hipStreamSynchronize(s));
            for (int q = 0; q < n_miss; ++q) win[miss[3 * q + 2]] = res[miss[3 * q + 2]];
        }
        const int cu = *du, cv = *dv;
        MI_TRY(argmax_exact(s, g, plane, d_base, ws, win.data(), H * W, win_exact,
                            [&](int i, int* a, int* b) { *a = i / W - g.wu + cu; *b = i % W - g.wv + cv; }, &ind_max));
    }
    if (exact_out) *exact_out = win_exact;
    if (ind_ref != ind_max) {
        *du += ind_max / W - g.wu;
        *dv += ind_max % W - g.wv;
        *failed = true;
    }
    return MI_OK;
}

struct PairPlan {
    int dimk, dimi_v, dimj_v;
    int delayi, delayj, delayk;
    int ai0, aj0;
    PlaneGeom g[3];
    size_t total_floats, map_floats, map_begin, res_floats, sat_doubles;
};

int plan_pair(int dimk, int dimi, int dimj, int nk, int ni, int nj, int delayk, int delayi, int delayj, int side, mi_ncc_params* p,
              PairPlan& pl) {
    MI_REQUIRE(p, "CrossMIPs: missing configuration parameters");
    if (p->enhance) return fail(MI_ERR_UNSUPPORTED, "CrossMIPs: enhance is not built (PDAlgoMIPNCC.cpp:81 never enables it)");
    MI_REQUIRE(dimk > 0 && dimi > 0 && dimj > 0, "CrossMIPs: empty stack");
    MI_REQUIRE(nk == 0, "CrossMIPs: nk must be 0 (CrossMIPs.h: assumed 0 by the current implementation)");
    MI_REQUIRE(side == MI_NORTH_SOUTH || side == MI_WEST_EAST, "CrossMIPs: unexpected alignment configuration");
    MI_REQUIRE(ni >= 0 && nj >= 0 && ni < dimi && nj < dimj, "CrossMIPs: initial offsets outside the stack");
    MI_REQUIRE(delayi >= 0 && delayj >= 0 && delayk >= 0, "CrossMIPs: negative search range");
    MI_REQUIRE(!(p->wRangeThr_i > delayi || p->wRangeThr_j > delayj || p->wRangeThr_k > delayk),
               "CrossMIPs: one or more parameters: wRangeThr_i[=%d], wRangeThr_j[=%d], wRangeThr_k[=%d] are too large with respect to: "
               "delayi[=%d], delayj[=%d], delayik[=%d]",
               p->wRangeThr_i, p->wRangeThr_j, p->wRangeThr_k, delayi, delayj, delayk);
    MI_REQUIRE(p->wRangeThr_i >= 0 && p->wRangeThr_j >= 0 && p->wRangeThr_k >= 0 && p->minPoints >= 1 && p->maxIter >= 0,
               "CrossMIPs: invalid parameters");
    // libcrossmips.cpp:260-262,275-277
    delayi = imin(delayi, imax(0, dimi - ni - p->minDim_NCCsrc));
    delayj = imin(delayj, imax(0, dimj - nj - p->minDim_NCCsrc));
    delayk = imin(delayk, imax(0, dimk - nk - p->minDim_NCCsrc));
    p->wRangeThr_i = imin(p->wRangeThr_i, delayi);
    p->wRangeThr_j = imin(p->wRangeThr_j, delayj);
    p->wRangeThr_k = imin(p->wRangeThr_k, delayk);
    pl.dimk = dimk;
    pl.dimi_v = side == MI_NORTH_SOUTH ? dimi - ni : dimi;
    pl.dimj_v = side == MI_WEST_EAST ? dimj - nj : dimj;
    pl.ai0 = side == MI_NORTH_SOUTH ? ni : 0;
    pl.aj0 = side == MI_WEST_EAST ? nj : 0;
    pl.delayi = delayi; pl.delayj = delayj; pl.delayk = delayk;
    const int mu[3] = {pl.dimi_v, pl.dimi_v, pl.dimj_v}, mv[3] = {pl.dimj_v, dimk, dimk};
    const int du[3] = {delayi, delayi, delayj}, dv[3] = {delayj, delayk, delayk};
    const int wu[3] = {p->wRangeThr_i, p->wRangeThr_i, p->wRangeThr_j}, wv[3] = {p->wRangeThr_j, p->wRangeThr_k, p->wRangeThr_k};
    size_t off = 0;
    auto take = [&off](size_t n) { size_t o = off; off += (n + 3) / 4 * 4; return o; };
    for (int m = 0; m < 3; ++m) {
        PlaneGeom& g = pl.g[m];
        g.dimu = mu[m]; g.dimv = mv[m]; g.delayu = du[m]; g.delayv = dv[m]; g.wu = wu[m]; g.wv = wv[m];
        g.tiled = (g.dimu / TILE) * (g.dimv / TILE) > 0;
    }
    // MIPs first (one memset covers them), then tile sums, then maps (one D2H covers them)
    for (int m = 0; m < 3; ++m) pl.g[m].mip1 = take((size_t)pl.g[m].dimu * pl.g[m].dimv);
    for (int m = 0; m < 3; ++m) pl.g[m].mip2 = take((size_t)pl.g[m].dimu * pl.g[m].dimv);
    const size_t mip_end = off;
    for (int m = 0; m < 3; ++m) {
        const size_t nt = (size_t)(pl.g[m].dimu / TILE) * (pl.g[m].dimv / TILE);
        pl.g[m].ps1 = take(nt);
        pl.g[m].ps2 = take(nt);
    }
    pl.map_begin = off;
    size_t res = 0;
    for (int m = 0; m < 3; ++m) {
        pl.g[m].map = take((size_t)(2 * du[m] + 1) * (2 * dv[m] + 1));
        res = std::max(res, (size_t)(2 * wu[m] + 1) * (2 * wv[m] + 1));
    }
    pl.map_floats = off - pl.map_begin;
    pl.res_floats = res;
    pl.total_floats = off;  // miss results live behind this
    size_t soff = 0;
    for (int m = 0; m < 3; ++m) {
        pl.g[m].sat = soff;
        soff += SatLayout(pl.g[m].dimu, pl.g[m].dimv).total;
    }
    pl.sat_doubles = soff;
    (void)mip_end;
    return MI_OK;
}

// stage 1 of a pair: everything up to the D2H copy of the three NCC maps is enqueued on `s`, nothing is waited for
int pair_enqueue(hipStream_t s, const float* A, const float* B, int dimi, int dimj, const PairPlan& pl, Workspace& ws, TileFmt fmt = TileFmt()) {
    const size_t need = pl.total_floats + pl.res_floats;
    if (ws.buf.bytes < sizeof(float) * need) MI_TRY(ws.buf.alloc(sizeof(float) * need));
    ws.floats = pl.total_floats;
    float* base = ws.buf.as<float>();
    {
        const size_t tmp = sizeof(float) * mips_tmp_floats(pl.dimk, pl.dimi_v, pl.dimj_v);
        if (ws.mip_tmp.bytes < tmp) {
            MI_HIP(hipStreamSynchronize(s));
            MI_TRY(ws.mip_tmp.alloc(tmp));
        }
    }
    MI_TRY(launch_mips(s, A, B, nullptr, 1, 0, pl.dimk, pl.dimi_v, pl.dimj_v, (size_t)dimi * dimj, dimj, pl.ai0, pl.aj0, base + pl.g[0].mip1,
                       base + pl.g[1].mip1, base + pl.g[2].mip1, base + pl.g[0].mip2, base + pl.g[1].mip2, base + pl.g[2].mip2,
                       ws.mip_tmp.as<float>(), nullptr, fmt));
    if (ws.sat.bytes < sizeof(double) * pl.sat_doubles) MI_TRY(ws.sat.alloc(sizeof(double) * pl.sat_doubles));
    for (int m = 0; m < 3; ++m) {
        const PlaneGeom& g = pl.g[m];
        MI_TRY(prepare_plane(s, base + g.mip1, base + g.mip2, g.dimu, g.dimv, base + g.ps1, base + g.ps2, ws.sat.as<double>() + g.sat,
                             &ws.v1[m], &ws.v2[m]));
        MI_TRY(ncc_launch(s, base + g.mip1, base + g.mip2, g.dimu, g.dimv, g.delayu, g.delayv, ws.v1[m], ws.v2[m], nullptr, 0, nullptr, 0,
                          ws.partial[m], base + g.map));
    }
    MI_TRY(ws.pin_maps.reserve(sizeof(float) * pl.map_floats));
    MI_HIP(hipMemcpyAsync(ws.pin_maps.p, base + pl.map_begin, sizeof(float) * pl.map_floats, hipMemcpyDeviceToHost, s));
    return MI_OK;
}

// stage 2: wait for the maps, then the host-side logic (argmax, neighbourhood refinement with its small device
// launches on the same stream, widths, alignment)
int pair_finish(hipStream_t s, int ni, int nj, int side, mi_ncc_params* p, const PairPlan& pl, Workspace& ws, mi_ncc_descr* out) {
    MI_HIP(hipStreamSynchronize(s));
    float* base = ws.buf.as<float>();
    const float* host_maps = ws.pin_maps.as<float>();

    std::vector<float> win[3];
    std::vector<unsigned char> wex[3];
    int du[3], dv[3];
    bool failed[3] = {false, false, false};
    ws.n_exact = 0;
    for (int m = 0; m < 3; ++m)
        MI_TRY(refine_neighbourhood(s, *p, host_maps + (pl.g[m].map - pl.map_begin), pl.g[m], m, base, ws, win[m], &du[m], &dv[m],
                                    &failed[m], &wex[m]));
    // compute_Alignment (compute_funcs.cu:1597-1609).  Every comparison of the width / alignment rules reads entries of the row
    // and the column through the window centre only: when one of them is decided by less than the resolution of the map values
    // those entries are recomputed in the two-pass form and the rules run again on them.
    for (int pass = 0; pass < 2; ++pass) {
        int w1[3], w2[3];
        float peak[3];
        float tight = 3.0e38f;
        for (int m = 0; m < 3; ++m) {
            const PlaneGeom& g = pl.g[m];
            const int rowlen = 2 * g.wv + 1, c = g.wu * rowlen + g.wv;
            peak[m] = win[m][c];
            if (failed[m]) { w1[m] = w2[m] = p->INF_W; continue; }
            w2[m] = peak_half_width(*p, win[m].data(), c, 1, g.wv, g.wv, &tight);
            w1[m] = peak_half_width(*p, win[m].data(), c, rowlen, g.wu, g.wv, &tight);
        }
        combine_axis(*p, out, 0, du[0], peak[0], w1[0], du[1], peak[1], w1[1], &tight);  // V: xy rows, xz rows
        combine_axis(*p, out, 1, dv[0], peak[0], w2[0], du[2], peak[2], w1[2], &tight);  // H: xy cols, yz rows
        combine_axis(*p, out, 2, dv[1], peak[1], w2[1], dv[2], peak[2], w2[2], &tight);  // D: xz cols, yz cols
        if (pass == 1 || !(tight < decision_margin())) break;
        for (int m = 0; m < 3; ++m) {
            const PlaneGeom& g = pl.g[m];
            const int H = 2 * g.wu + 1, W = 2 * g.wv + 1;
            std::vector<int> uv, idx;
            for (int r = 0; r < H; ++r)
                for (int c = 0; c < W; ++c)
                    if ((r == g.wu || c == g.wv) && !wex[m][(size_t)r * W + c]) {
                        uv.push_back(r - g.wu + du[m]); uv.push_back(c - g.wv + dv[m]); idx.push_back(r * W + c);
                    }
            std::vector<float> val;
            MI_TRY(exact_entries(s, g, m, base, ws, uv, val));
            for (size_t q = 0; q < idx.size(); ++q) { win[m][idx[q]] = val[q]; wex[m][idx[q]] = 1; }
        }
    }
    if (side == MI_NORTH_SOUTH) out->coord[0] += ni; else out->coord[1] += nj;  // libcrossmips.cpp:483-486
    return MI_OK;
}

// a pair's working set between batch calls: workspace + stream, per device
struct PairSlot {
    int dev = 0;
    Workspace ws;
    hipStream_t s = nullptr;
    ~PairSlot() {
        if (s) {
            (void)hipSetDevice(dev);
            (void)hipStreamDestroy(s);
        }
    }
};
// never destroyed: at process exit the HIP runtime may already be gone
std::mutex& g_slot_mu = *new std::mutex;
std::vector<std::unique_ptr<PairSlot>>& g_slots = *new std::vector<std::unique_ptr<PairSlot>>;

std::unique_ptr<PairSlot> take_pair_slot(int dev) {
    {
        std::lock_guard<std::mutex> g(g_slot_mu);
        for (size_t i = 0; i < g_slots.size(); ++i)
            if (g_slots[i]->dev == dev) {
                std::unique_ptr<PairSlot> r = std::move(g_slots[i]);
                g_slots.erase(g_slots.begin() + i);
                return r;
            }
    }
    std::unique_ptr<PairSlot> r(new (std::nothrow) PairSlot);
    if (!r) return r;
    r->dev = dev;
    if (hipStreamCreateWithFlags(&r->s, hipStreamNonBlocking) != hipSuccess) r->s = nullptr;
    return r;
}

void give_pair_slot(std::unique_ptr<PairSlot> r) {
    if (!r) return;
    std::lock_guard<std::mutex> g(g_slot_mu);
    if (g_slots.size() < 16) g_slots.push_back(std::move(r));  // beyond that the slot is simply destroyed
}

}  // namespace
