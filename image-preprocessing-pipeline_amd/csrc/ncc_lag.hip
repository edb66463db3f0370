// Batched MIP-NCC: all pairs of a batch go through the device together, one stream synchronisation per batch.
//
// Replaces, for n pairs at once, libcrossmips.cpp:319-481 (6 MIPs, 3 NCC maps, 3 neighbourhood refinements) and the pair loop
// of StackStitcher.cpp:223-374 around it.  What differs from the per-pair path of ncc.hip / ncc_core.h:
//
//   * every kernel is launched once per (group of equal-geometry pairs, plane) with the pair index in the grid;
//   * the cross terms  sum f[i+u][j+v] t[i][j]  -- the only part of compute_NCC (compute_funcs.cu:1163-1292) that is not O(1) per
//     shift once the summed-area tables exist -- are no longer accumulated shift by shift (1.6e9 fp64 FMA for a 2048 x 307
//     MIP and 51 x 51 shifts) but through a LAG TRANSFORM along the long axis of the MIP, in fp64:
//       k_lag_fwd   column j of both MIPs -> one zero-padded complex FFT of length N >= n_long + E (f in the real, t in the
//                   imaginary part), untangled into F_j[k], T_j[k], k = 0 .. N/2;
//       k_lag_mac   per frequency k a direct correlation along the SHORT axis:  C_s[k] = sum_j F_{j+s}[k] conj(T_j[k])
//                   for every short-axis lag s in [-Es, Es];
//       k_lag_inv   per short-axis lag the inverse transform over k (two lags per complex FFT) -> cross[u][v] for EVERY long-axis
//                   lag, of which [-El, El] are kept.
//     That is ~10x fewer fp64 operations than the shift-by-shift sums and -- because every lag within E = delay + (re-centring
//     moves) * wRangeThr exists afterwards -- the neighbourhood refinement (compute_Neighborhood, :1324-1592) needs no further
//     pass over the MIPs: k_lag_refine does argmax, window extraction, the re-centring moves and the evaluation of the newly
//     exposed entries on the device, one work-group per (pair, plane), and only the final window + 4 ints per plane return.
//   * the host finishes with the unchanged bit-identical width / alignment rules (ncc_core.h).
//
// Accuracy: the transforms are fp64 with table twiddles; a cross term differs from the sequential fp64 sum by ~1e-15 of
// ||f|| ||t||, i.e. an NCC value by ~1e-13 -- six orders below the float the reference rounds to.  Decisions whose margin is
// below MI_NCC_MARGIN (4e-6: argmax runner-up on the device, threshold crossings / slope steps / rounding steps on the host) are
// not taken here: the pair is handed to the careful per-pair path (ncc.hip), which recomputes the entries involved with the
// reference's two-pass fp64 form.  The same happens when a re-centring move leaves the lag range that was transformed.
#include <cfloat>
#include <map>

#include "fft64_lds.h"
#include "ncc_core.h"
#include "ncc_lag.h"

namespace {

typedef double2 cplx;
using fft64::Plan;

__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }

// The transforms are those of fft64_lds.h: length N = 2^a * {1, 3, 9} (2048-row MIPs with 75 lags need N >= 2123: 2304 = 9 * 16 * 16),
// a thread owns a whole radix-16 / 9 / 8 butterfly, three LDS round trips for 2304 points under a conflict-free XOR image.
// First stage of a forward transform whose inputs come from load(i), i = logical index, instead of LDS (no fill pass).
template <int R, class Load>
__device__ __forceinline__ void first_stage_from(cplx* x, const Plan& pl, const cplx* __restrict__ tw, int first, int step, Load load) {
    const int nb = pl.N / R, q = pl.lr[0] == 0 ? (1 << pl.a) : (1 << pl.lq[0]);
    const cplx* stw = tw + pl.twoff[0];
    for (int t = first; t < nb; t += step) {
        cplx v[R];
#pragma unroll
        for (int m = 0; m < R; ++m) v[m] = load(t + m * q);
        fft64::butterfly<R, false>(v, stw, q, t);
        const int base = fft64::phys(pl, t);
#pragma unroll
        for (int m = 0; m < R; ++m) x[base ^ pl.pm[0][m]] = v[m];
    }
}
template <class Load>
__device__ __forceinline__ void first_stage_any(cplx* x, const Plan& pl, const cplx* __restrict__ tw, int first, int step, Load load) {
    switch (pl.radix[0]) {
        case 16: first_stage_from<16>(x, pl, tw, first, step, load); break;
        case 12: first_stage_from<12>(x, pl, tw, first, step, load); break;
        case 9: first_stage_from<9>(x, pl, tw, first, step, load); break;
        case 8: first_stage_from<8>(x, pl, tw, first, step, load); break;
        case 6: first_stage_from<6>(x, pl, tw, first, step, load); break;
        case 4: first_stage_from<4>(x, pl, tw, first, step, load); break;
        case 3: first_stage_from<3>(x, pl, tw, first, step, load); break;
        default: first_stage_from<2>(x, pl, tw, first, step, load); break;
    }
}

// Spectra of a plane: SP[((pair * ntile + tile) * n_short + j) * 2 KT + {0: F, 1: T} * KT + l] -- the KT frequency slots of a tile,
// for line j, F then T: one 128-byte line when KT = 4.  The forward transform of line j stores whole lines; the correlation
// kernels read a tile (all j of KT slots) as ONE contiguous block.

// forward lag transform of short-axis line j (blockIdx.x) of pair blockIdx.y: element i of the line = m[i * ls + j * ss]
template <int TH>
__global__ __launch_bounds__(TH) void k_lag_fwd(const float* __restrict__ m1, const float* __restrict__ m2, size_t pstride, int n_long, int n_short,
                                                 int ls, int ss, int KT, int ntile, const Plan* __restrict__ plp, const cplx* __restrict__ tw,
                                                 const int* __restrict__ slot_pos, const int* __restrict__ slot_neg, cplx* __restrict__ SP) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lag_lds[];
    cplx* x = reinterpret_cast<cplx*>(lag_lds);
    const Plan& pl = *plp;  // (in device memory, not a by-value argument: the stage loops index its tables with the stage number, and a
                            // by-value struct indexed that way is copied to scratch memory first)
    // gridDim.x = 8 * ceil(n_short / 8): work-groups b, b + 8, ... run on one XCD and take NEIGHBOURING lines j -- with the long
    // axis strided in memory (west-east pairs) 32 neighbouring lines share every 128-byte line they read, and spread over the
    // eight L2s each of them fetched it again (0.81 GB fetched for 0.28 GB of MIPs)
    const int N = pl.N, NK = N / 2 + 1, j = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (j >= n_short) return;
    const float* a = m1 + (size_t)blockIdx.y * pstride + (size_t)j * ss;
    const float* b = m2 + (size_t)blockIdx.y * pstride + (size_t)j * ss;
    // f in the real, t in the imaginary part; the zero padding never exists in memory
    // (unconditional loads through a clamped index: a load inside a predicated block is waited for on the spot, and the nine
    // inputs of a butterfly then arrive one memory latency after the other)
    cplx* twl = x + N;  // the later stages' twiddles (fft64::lds_twiddles)
    for (int e = threadIdx.x; e < fft64::lds_twiddles(pl); e += TH) twl[e] = tw[pl.twoff[1] + e];
    const cplx* tws = twl - (pl.nst > 1 ? pl.twoff[1] : 0);
    first_stage_any(x, pl, tw, threadIdx.x, TH, [&](int i) {
        const size_t o = (size_t)min(i, n_long - 1) * ls;
        const float fa = a[o], fb = b[o];
        return make_double2(i < n_long ? (double)fa : 0.0, i < n_long ? (double)fb : 0.0);
    });
    __syncthreads();
    for (int st = 1; st < pl.nst; ++st) {
        fft64::stage_any<false>(x, 0, 1, pl, st, tws, threadIdx.x, TH);
        __syncthreads();
    }
    // The N/2 + 1 frequencies 0 .. N/2 of a real signal's spectrum are kept in POSITION order ("slots": slot_pos = LDS slot of the
    // frequency, ascending in logical position; slot_neg = slot of the mirror frequency N - k): the untangling pass reads LDS
    // nearly contiguously, and the per-frequency correlation does not care about the order of its frequencies.
    cplx* row = SP + ((size_t)blockIdx.y * ntile * n_short + j) * 2 * KT;
    const size_t tstride = (size_t)n_short * 2 * KT;
    for (int sl = threadIdx.x; sl < NK; sl += TH) {
        const cplx zk = x[slot_pos[sl]], zn = x[slot_neg[sl]];
        const int tile = sl / KT, l = sl - tile * KT;
        cplx* o = row + (size_t)tile * tstride + l;
        o[0] = make_double2(0.5 * (zk.x + zn.x), 0.5 * (zk.y - zn.y));    // (Z[k] + conj Z[N-k]) / 2
        o[KT] = make_double2(0.5 * (zk.y + zn.y), -0.5 * (zk.x - zn.x));  // (Z[k] - conj Z[N-k]) / (2 i)
    }
}

// Correlation along the short axis through a second transform (planes whose short axis is long enough to pay for it): per
// frequency slot k,  C_s[k] = sum_j F_{j+s}[k] conj(T_j[k])  is the inverse transform of  F^[m] conj(T^[m])  on a zero-padded circle
// of M >= n_short + Es points.  A work-group takes the KT slots of a tile: 2 KT transforms of M points side by side in LDS (images
// M + 1 elements apart, so the 2 KT elements a wave writes for one t land in different banks), first stage straight from the
// tile in global memory, point-wise product in digit-reversed order, backward pass on KT images, last stage straight to CH.
// 3 x 5 M log2 M flops per slot instead of 8 n_short (2 Es + 1): config 5's xy plane 5 x fewer, and LDS-bound instead of fp64-bound.
template <int R, int KT2>
__device__ __forceinline__ void corr_first(cplx* arr, int AS, const cplx* __restrict__ src, int n_short, int nk, const Plan& pl,
                                           const cplx* __restrict__ tw, int first, int step) {
    const int nb = pl.N / R, q = pl.lr[0] == 0 ? (1 << pl.a) : (1 << pl.lq[0]);
    const cplx* stw = tw + pl.twoff[0];
    for (int e = first; e < nb * KT2; e += step) {
        const int t = e / KT2, f = e % KT2;  // f fastest: a wave reads whole 128-byte lines of the tile
        cplx v[R];
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int i = t + m * q;  // (clamped index + select: see k_lag_fwd; a tile is a few thousand elements: 32-bit offsets)
            const cplx g = src[min(i, n_short - 1) * KT2 + f];
            const bool on = i < n_short && (f % (KT2 / 2)) < nk;   // (slots past the spectrum's end in the last tile were never written)
            v[m] = make_double2(on ? g.x : 0.0, on ? g.y : 0.0);
        }
        fft64::butterfly<R, false>(v, stw, q, t);
        cplx* x = arr + (size_t)f * AS;
        const int base = fft64::phys(pl, t);
#pragma unroll
        for (int m = 0; m < R; ++m) x[base ^ pl.pm[0][m]] = v[m];
    }
}
template <int R>
__device__ __forceinline__ void corr_last(const cplx* arr, int AS, int nk, const Plan& pl, const cplx* __restrict__ tw, int Es, int nlp,
                                          cplx* __restrict__ dst, int first, int step) {
    const int nb = pl.N / R, q = pl.lr[0] == 0 ? (1 << pl.a) : (1 << pl.lq[0]), ls = pl.lslot[0], M = pl.N;
    const cplx* stw = tw + pl.twoff[0];
    const double inv = 1.0 / (double)M;
    for (int e = first; e < (nk << ls); e += step) {
        const int l = e >> ls, t = e & ((1 << ls) - 1);
        if (t >= nb) continue;
        bool need = false;
#pragma unroll
        for (int m = 0; m < R; ++m) need = need || t + m * q <= Es || t + m * q >= M - Es;
        if (!need) continue;
        const cplx* x = arr + (size_t)l * AS;
        const int base = fft64::phys(pl, t);
        cplx v[R];
#pragma unroll
        for (int m = 0; m < R; ++m) v[m] = x[base ^ pl.pm[0][m]];
        fft64::butterfly<R, true>(v, stw, q, t);
        cplx* o = dst + (size_t)l * nlp;
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int pos = t + m * q;  // lag s >= 0 at s, s < 0 at M + s
            if (pos <= Es) o[pos + Es] = make_double2(v[m].x * inv, -v[m].y * inv);
            else if (pos >= M - Es) o[pos - M + Es] = make_double2(v[m].x * inv, -v[m].y * inv);
        }
    }
}
template <int TH, int KT>
__global__ __launch_bounds__(TH) void k_lag_corr(const cplx* __restrict__ SP, int n_short, int NK, int Es, int nlp, int tiles_per_pair, const Plan* __restrict__ plp,
                                                  const cplx* __restrict__ tw, cplx* __restrict__ CH) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lag_lds[];
    cplx* arr = reinterpret_cast<cplx*>(lag_lds);
    const Plan& pl = *plp;
    constexpr int KT2 = 2 * KT;
    const int tile = blockIdx.x, pair = blockIdx.y, M = pl.N, AS = M + 1;
    const int k0 = tile * KT, nk = min(KT, NK - k0);
    if (nk <= 0) return;
    const cplx* src = SP + ((size_t)pair * tiles_per_pair + tile) * n_short * KT2;
    cplx* twl = arr + (size_t)KT2 * AS;
    for (int e = threadIdx.x; e < fft64::lds_twiddles(pl); e += TH) twl[e] = tw[pl.twoff[1] + e];
    const cplx* tws = twl - (pl.nst > 1 ? pl.twoff[1] : 0);
    switch (pl.radix[0]) {
        case 16: corr_first<16, KT2>(arr, AS, src, n_short, nk, pl, tw, threadIdx.x, TH); break;
        case 12: corr_first<12, KT2>(arr, AS, src, n_short, nk, pl, tw, threadIdx.x, TH); break;
        case 9: corr_first<9, KT2>(arr, AS, src, n_short, nk, pl, tw, threadIdx.x, TH); break;
        case 8: corr_first<8, KT2>(arr, AS, src, n_short, nk, pl, tw, threadIdx.x, TH); break;
        case 6: corr_first<6, KT2>(arr, AS, src, n_short, nk, pl, tw, threadIdx.x, TH); break;
        case 4: corr_first<4, KT2>(arr, AS, src, n_short, nk, pl, tw, threadIdx.x, TH); break;
        case 3: corr_first<3, KT2>(arr, AS, src, n_short, nk, pl, tw, threadIdx.x, TH); break;
        default: corr_first<2, KT2>(arr, AS, src, n_short, nk, pl, tw, threadIdx.x, TH); break;
    }
    __syncthreads();
    for (int st = 1; st < pl.nst; ++st) {
        fft64::stage_any<false>(arr, AS, KT2, pl, st, tws, threadIdx.x, TH);
        __syncthreads();
    }
    // conj(F^ conj(T^)) = conj(F^) T^, position by position (both spectra sit in the same digit-reversed, swizzled order), into F's image
    for (int l = 0; l < nk; ++l) {
        cplx* F = arr + (size_t)l * AS;
        const cplx* T = arr + (size_t)(KT + l) * AS;
        for (int p = threadIdx.x; p < M; p += TH) {
            const cplx f = F[p], t = T[p];
            F[p] = make_double2(f.x * t.x + f.y * t.y, f.x * t.y - f.y * t.x);
        }
    }
    __syncthreads();
    for (int st = pl.nst - 1; st >= 1; --st) {
        fft64::stage_any<true>(arr, AS, nk, pl, st, tws, threadIdx.x, TH);
        __syncthreads();
    }
    cplx* dst = CH + ((size_t)pair * NK + k0) * nlp;
    switch (pl.radix[0]) {
        case 16: corr_last<16>(arr, AS, nk, pl, tw, Es, nlp, dst, threadIdx.x, TH); break;
        case 12: corr_last<12>(arr, AS, nk, pl, tw, Es, nlp, dst, threadIdx.x, TH); break;
        case 9: corr_last<9>(arr, AS, nk, pl, tw, Es, nlp, dst, threadIdx.x, TH); break;
        case 8: corr_last<8>(arr, AS, nk, pl, tw, Es, nlp, dst, threadIdx.x, TH); break;
        case 6: corr_last<6>(arr, AS, nk, pl, tw, Es, nlp, dst, threadIdx.x, TH); break;
        case 4: corr_last<4>(arr, AS, nk, pl, tw, Es, nlp, dst, threadIdx.x, TH); break;
        case 3: corr_last<3>(arr, AS, nk, pl, tw, Es, nlp, dst, threadIdx.x, TH); break;
        default: corr_last<2>(arr, AS, nk, pl, tw, Es, nlp, dst, threadIdx.x, TH); break;
    }
}

// per frequency: C_s[k] = sum_j F_{j+s}[k] conj(T_j[k]),  s = -Es .. Es.  A TILE is KT frequencies of one pair; a thread takes one
// frequency, LB neighbouring lags and one of JP parts of the j range.  The LB samples of F a step needs sit in a ring of registers
// (two 16-byte LDS reads feed 4 LB FMAs; LB steps bring every sample back to its slot).
// The work-groups are persistent (tile blockIdx.x, + gridDim.x, ...) and carry the NEXT tile's spectra in registers (PF elements
// of F and of T per thread, requested right after the current tile reached LDS): work-groups of one CU start together and stay in
// step, so without this every fill phase -- 39 KB per tile on the xy plane of config 5 -- was a phase in which the CU computed
// nothing (600 -> 4xx us for that plane: profiles/r03_lag_shapes.txt).
template <int LB, int PF>
__global__ __launch_bounds__(256) void k_lag_mac(const cplx* __restrict__ SP, int n_short, int NK, int Es, int KT, int JP, int FW, int TW, int nlp,
                                                 int tiles_per_pair, int ntiles, cplx* __restrict__ CH) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lag_lds[];
    cplx* Fs = reinterpret_cast<cplx*>(lag_lds);  // KT rows of FW: PAD zeros | n_short samples | zeros
    cplx* Ts = Fs + (size_t)KT * FW;              // KT rows of TW
    cplx* red = Ts + (size_t)KT * TW;             // [jp - 1][vb][kl][LB]: the partial sums of the parts jp > 0
    const int PAD = Es + LB - 1;
    for (int e = threadIdx.x; e < KT * FW; e += blockDim.x) Fs[e] = make_double2(0.0, 0.0);
    const int nvb = nlp / LB;
    const int jlen = (n_short + JP - 1) / JP;
    const int item = threadIdx.x;  // (jp, vb, kl), kl fastest
    const int kl = item % KT, vb = (item / KT) % nvb, jp = item / (KT * nvb);
    const int v0 = -Es + LB * vb, jb = jp * jlen, je = min(n_short, jb + jlen);
    const cplx* fr = Fs + (size_t)kl * FW + PAD + v0;
    const cplx* tr = Ts + (size_t)kl * TW;
    const int nelem = n_short * KT;  // elements of a tile, kl fastest: 16 * KT contiguous bytes per line j
    const size_t tsz = (size_t)n_short * 2 * KT;  // a tile of SP: [j][F | T][KT]

    // Slot b of the persistent sequence (blockIdx.x, + gridDim.x, ...; both multiples of 16) -> tile (neighbouring tiles on one
    // XCD).  Tiles past the end and the padding tile of a pair are empty (nk <= 0).
    auto decode = [&](int b, int& tile, int& k0, int& nk) {
        tile = (b & ~15) | ((b & 7) << 1) | ((b >> 3) & 1);
        const int pair = tile / tiles_per_pair;
        k0 = (tile - pair * tiles_per_pair) * KT;
        nk = tile < ntiles ? min(KT, NK - k0) : 0;
    };
    const int nslots = (ntiles + 15) & ~15;
    int slot = blockIdx.x;
    __syncthreads();
    if (slot < nslots) {  // the first tile goes to LDS directly
        int tile, k0, nk;
        decode(slot, tile, k0, nk);
        const cplx* src = SP + (size_t)tile * tsz;
        for (int e = threadIdx.x; e < nelem; e += 256) {
            const int x = e / KT, l = e - x * KT;
            if (l < nk) {
                Fs[(size_t)l * FW + PAD + x] = src[(size_t)x * 2 * KT + l];
                Ts[(size_t)l * TW + x] = src[(size_t)x * 2 * KT + KT + l];
            }
        }
    }
    while (slot < nslots) {
        int tile, k0, nk;
        decode(slot, tile, k0, nk);
        const int pair = tile / tiles_per_pair;
        __syncthreads();  // the tile is in LDS
        const int next = slot + gridDim.x;
        const bool has_next = next < nslots;
        int ntile_ = 0, nk0 = 0, nnk = 0;
        if (has_next) decode(next, ntile_, nk0, nnk);
        const cplx* nsrc = SP + (size_t)ntile_ * tsz;
        cplx pf[PF > 0 ? PF : 1], pt[PF > 0 ? PF : 1];  // (PF = 0: rows too long for the registers -- the next tile is fetched behind this one)
        if (PF > 0 && has_next) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int e = threadIdx.x + i * 256;
                const int x = e / KT, l = e - x * KT;
                const bool ok = e < nelem && l < nnk;
                pf[i] = ok ? nsrc[(size_t)x * 2 * KT + l] : make_double2(0.0, 0.0);
                pt[i] = ok ? nsrc[(size_t)x * 2 * KT + KT + l] : make_double2(0.0, 0.0);
            }
        }

        const bool live = jp < JP && kl < nk;
        cplx acc[LB];
#pragma unroll
        for (int q = 0; q < LB; ++q) acc[q] = make_double2(0.0, 0.0);
        if (live) {
            const cplx* fp = fr + jb;
            const cplx* tp = tr + jb;
            cplx f[LB];
#pragma unroll
            for (int q = 0; q < LB - 1; ++q) f[q] = fp[q];
            auto steps = [&](int rem) {  // LB steps; rem < LB: only the first rem count (the others see t = 0)
#pragma unroll
                for (int jj = 0; jj < LB; ++jj) {
                    f[(jj + LB - 1) % LB] = fp[jj + LB - 1];  // (zeros past the row: FW leaves room for a whole trip)
                    cplx t = tp[jj];
                    if (jj >= rem) t = make_double2(0.0, 0.0);  // (the next part's samples, or what lies past the row)
#pragma unroll
                    for (int q = 0; q < LB; ++q) {
                        const cplx fv = f[(jj + q) % LB];
                        acc[q].x = fma(fv.x, t.x, fma(fv.y, t.y, acc[q].x));
                        acc[q].y = fma(fv.y, t.x, fma(-fv.x, t.y, acc[q].y));
                    }
                }
            };
            const int trips = (je - jb) / LB;
            for (int it = 0; it < trips; ++it) {
                steps(LB);
                fp += LB;
                tp += LB;
            }
            if (je - jb > trips * LB) steps(je - jb - trips * LB);
        }
        // the JP parts of a (kl, vb) are added in a fixed order
        if (live && jp > 0) {
#pragma unroll
            for (int q = 0; q < LB; ++q) red[(((size_t)(jp - 1) * nvb + vb) * KT + kl) * LB + q] = acc[q];
        }
        __syncthreads();  // (also: every read of this tile's spectra is done)
        if (live && jp == 0) {
            for (int p = 1; p < JP; ++p)
#pragma unroll
                for (int q = 0; q < LB; ++q) acc[q] = cadd(acc[q], red[(((size_t)(p - 1) * nvb + vb) * KT + kl) * LB + q]);
            cplx* dst = CH + ((size_t)pair * NK + k0 + kl) * nlp + LB * vb;
#pragma unroll
            for (int q = 0; q < LB; ++q) dst[q] = acc[q];
        }
        if (PF == 0 && has_next) {
            for (int e = threadIdx.x; e < nelem; e += 256) {
                const int x = e / KT, l = e - x * KT;
                if (l < nnk) {
                    Fs[(size_t)l * FW + PAD + x] = nsrc[(size_t)x * 2 * KT + l];
                    Ts[(size_t)l * TW + x] = nsrc[(size_t)x * 2 * KT + KT + l];
                }
            }
        }
        if (PF > 0 && has_next) {  // (every read of this tile's spectra lies before the barrier above; `red` is its own region)
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int e = threadIdx.x + i * 256;
                const int x = e / KT, l = e - x * KT;
                if (e < nelem && l < nnk) {
                    Fs[(size_t)l * FW + PAD + x] = pf[i];
                    Ts[(size_t)l * TW + x] = pt[i];
                }
            }
        }
        slot = next;
    }
}

// inverse lag transform of two short-axis lags (sa, sa + 1) of a pair: c_a[n] + i c_b[n] = FFT(conj C_a + i conj C_b) / N (both c
// real); the long-axis lags [-El, El] go to cross[(u + Eu) * (2 Ev + 1) + (v + Ev)].  The first stage gathers its inputs from CH
// (slot of frequency k: freq_slot[k]).  Work-groups b, b + 8, ... share an XCD and take the lag pairs of ONE pair after another: its
// block of CH (2 MB on config 5, read as 32-byte pieces of 1.6-KB rows) is then fetched once into that L2.
template <int TH>
__global__ __launch_bounds__(TH) void k_lag_inv(const cplx* __restrict__ CH, int NK, int nlp, const Plan* __restrict__ plp, int Es, int El, int long_is_u, int Eu, int Ev,
                                                 const cplx* __restrict__ tw, const int* __restrict__ freq_slot, int np, int lagpairs,
                                                 double* __restrict__ cross) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lag_lds[];
    cplx* x = reinterpret_cast<cplx*>(lag_lds);
    const Plan& pl = *plp;
    const int N = pl.N, nlag = 2 * Es + 1;
    const int slot = (int)(blockIdx.x >> 3), pair = 8 * (slot / lagpairs) + (int)(blockIdx.x & 7);
    if (pair >= np) return;
    const int sa = 2 * (slot % lagpairs), sb = sa + 1;
    const bool has_b = sb < nlag;
    const cplx* src = CH + (size_t)pair * NK * nlp;
    cplx* twl = x + N;
    for (int e = threadIdx.x; e < fft64::lds_twiddles(pl); e += TH) twl[e] = tw[pl.twoff[1] + e];
    const cplx* tws = twl - (pl.nst > 1 ? pl.twoff[1] : 0);
    first_stage_any(x, pl, tw, threadIdx.x, TH, [&](int kk) {
        const bool low = 2 * kk <= N;
        const int sl = freq_slot[low ? kk : N - kk];
        const cplx xa = src[(size_t)sl * nlp + sa];
        cplx xb = src[(size_t)sl * nlp + (has_b ? sb : sa)];
        if (!has_b) xb = make_double2(0.0, 0.0);
        return low ? make_double2(xa.x + xb.y, -xa.y + xb.x)   // conj(Xa) + i conj(Xb)
                   : make_double2(xa.x - xb.y, xa.y + xb.x);   // Xa + i Xb  (= conj X[N-k] terms)
    });
    __syncthreads();
    for (int st = 1; st < pl.nst; ++st) {
        fft64::stage_any<false>(x, 0, 1, pl, st, tws, threadIdx.x, TH);
        __syncthreads();
    }
    const double inv = 1.0 / (double)N;
    const int W = 2 * Ev + 1;
    double* out = cross + (size_t)pair * (2 * Eu + 1) * W;
    for (int e = threadIdx.x; e < 2 * El + 1; e += TH) {
        const int l = e - El;
        const cplx v = x[fft64::phys(pl, fft64::pos_of_freq(pl, (l + N) % N))];
        if (long_is_u) {
            out[(size_t)(l + Eu) * W + (sa - Es + Ev)] = v.x * inv;
            if (has_b) out[(size_t)(l + Eu) * W + (sb - Es + Ev)] = v.y * inv;
        } else {
            out[(size_t)(sa - Es + Eu) * W + (l + Ev)] = v.x * inv;
            if (has_b) out[(size_t)(sb - Es + Eu) * W + (l + Ev)] = v.y * inv;
        }
    }
}

// ------------------------------------------------------------------------------------------------ banded summed-area tables
// (BandView, ncc_core.h).  Logical coordinates: a = long axis, b = short axis, element (a, b) = m[a * ls + b * ss].
// Everything the refinement needs besides the cross terms, for the three planes of every pair of a group in TWO launches of
// small independent work-groups (round 3: six launches per plane -- tile sums, pixel sums, means, chunk sums, band columns, band
// rows -- 36 dependent launches per call of two groups, 1.3 ms of kernel time for 0.7 GB of reads):
//   k_plane_sums   float tile sums in the reference's order (seq_cpu_compute_partial_sums, compute_funcs.cu:474-500), and the
//                  column sums of every SEGMENT of rows: the kept rows in pieces of BAND_RC, the rows no window statistic reads
//                  (a in [B, n_long - B): most of a MIP) in chunks of BAND_CH;
//   k_band_tables  per (pair, plane, MIP) one work-group per PIECE of kept rows (its start = the segment sums before it, its rows
//                  walked by a thread per column, the prefix along b by wave scans in registers) and one for the tile-sum table.
// (One work-group per MIP with the pieces in sequence was built first: 0.2 ms alone, 0.9 ms beside the transform kernels, which
// left its 1024-thread work-groups waiting for whole compute units.)
// The tables hold sums of g = f - c0 and g^2.  c0 only has to lie near the mean (it keeps the sums small; every use of the tables
// adds it back exactly): it is the mean of 64 evenly spaced samples, which every wave that needs it takes itself -- the exact mean
// of round 3 cost a reduction over the whole MIP and a launch boundary before anything else could start.
constexpr int BAND_CH = 128;  // rows per chunk of the skipped rows
constexpr int BAND_RC = 16;   // kept rows per piece
struct TabPlane {
    int dimu, dimv, tiled, nt, pw;                       // MIP extents, full 32 x 32 tiles
    int n_long, n_short, ls, ss, B, rows, long_contig;   // band geometry (BandLayout)
    int ppb, nbands, nch, nseg, units;                   // pieces per band, bands, chunks of skipped rows, segments, waves of sums per MIP
    size_t mip1, mip2, ps1, ps2;                         // float offsets inside a pair's block of the MIP buffer
    size_t sat_off, tab, ts;                             // double offsets / sizes inside a pair's block of the table buffer
};
struct TabGeom {
    TabPlane p[3];
    int nplanes;
    size_t pstride, sstride;
};

struct __attribute__((packed, aligned(4))) F4u { float v[4]; };  // four floats at a 4-byte aligned address (rows of odd width)

// mean of 64 evenly spaced samples of a MIP of n pixels, the same bits in every lane of every wave that asks
__device__ __forceinline__ double sample_c0(const float* __restrict__ m, size_t n) {
    const int lane = threadIdx.x & 63;
    double v = (double)m[(size_t)lane * n / 64];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v * (1.0 / 64.0);
}

// inclusive prefix sum over the 64 lanes of a wave, in the VALU: four shifts inside the rows of 16 lanes, then the totals of
// rows 0 / 2 into rows 1 / 3 and of rows 0-1 into rows 2-3 (DPP moves of the two halves of the double; lanes without a source
// add 0).  (__shfl_up goes through the LDS crossbar: 12 dependent trips per scan, and a piece of the tables takes 32 scans.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_scan_dpp(double v) {
    v += dpp_take<0x111, 0xf>(v);  // row_shr:1
    v += dpp_take<0x112, 0xf>(v);  // row_shr:2
    v += dpp_take<0x114, 0xf>(v);  // row_shr:4
    v += dpp_take<0x118, 0xf>(v);  // row_shr:8
    v += dpp_take<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_take<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

// rows [a0, a1) of segment s: pieces of the lower band | chunks of the skipped rows | pieces of the upper band
__device__ __forceinline__ void seg_rows(const TabPlane& T, int s, int* a0, int* a1) {
    const bool full = T.nbands == 1;
    if (s < T.ppb) {
        *a0 = s * BAND_RC;
        *a1 = min(*a0 + BAND_RC, full ? T.n_long : T.B);
    } else if (s < T.ppb + T.nch) {
        const int skipped = T.n_long - 2 * T.B, ch = T.long_contig ? skipped : BAND_CH;
        *a0 = T.B + (s - T.ppb) * ch;
        *a1 = min(*a0 + ch, T.n_long - T.B);
    } else {
        *a0 = T.n_long - T.B + (s - T.ppb - T.nch) * BAND_RC;
        *a1 = min(*a0 + BAND_RC, T.n_long);
    }
}

// blockIdx.z = pair, blockIdx.y = 2 * plane + MIP.  blockIdx.x in [0, tb): tile sums (below).  blockIdx.x in [tb, ...): column
// sums of the segments, a wave per unit: (segment, strip of 64 columns) with a lane per column, or -- the skipped rows of a MIP
// whose long axis is the contiguous one -- (column b) with the lanes along a.  T[((2 MIP + {p, q}) * nseg + segment) * n_short + b].
__global__ __launch_bounds__(256) void k_plane_sums(TabGeom G, float* __restrict__ fbuf, double* __restrict__ sat, int tb, int knock) {
    const int plane = blockIdx.y >> 1, which = blockIdx.y & 1;
    if (knock && (((int)blockIdx.x < tb) ? (knock & 1) : (knock & 2))) return;  // (probe builds only: MI_NCC_TAB_KNOCK)
    const TabPlane& T = G.p[plane];
    float* base = fbuf + (size_t)blockIdx.z * G.pstride;
    const float* m = base + (which ? T.mip2 : T.mip1);
    if ((int)blockIdx.x < tb) {
        // a wave per (row of tiles ti, block of up to 64 tiles along the row): the 32 image rows of the band arrive as coalesced
        // 16-byte loads (the next row requested while the current one is added), cross LDS once -- tile t at 36 t floats, so
        // that the 16-byte reads of 16 lanes fall into different banks -- and lane t adds the 32 samples of its tile's row in
        // order into a FLOAT running sum like the reference does, bit-identical by construction
        __shared__ __attribute__((aligned(16))) float tl[4][64 * 36];
        const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int tjb = (T.pw + 63) / 64, unit = (int)blockIdx.x * 4 + wv;
        if (!T.tiled || unit >= (T.dimu / TILE) * tjb) return;
        const int ti = unit / tjb, tj0 = (unit - ti * tjb) * 64, ntw = min(64, T.pw - tj0), width = T.dimv, nf4 = 8 * ntw;
        const float* p = m + (size_t)ti * TILE * width + tj0 * TILE;
        float* my = tl[wv];
        F4u cur[8], nxt[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int f = min(lane + 64 * k, nf4 - 1);
            cur[k] = *reinterpret_cast<const F4u*>(p + 4 * f);
        }
        float s = 0.0f;
        for (int i = 0; i < TILE; ++i) {
            const float* pn = p + (size_t)min(i + 1, TILE - 1) * width;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int f = min(lane + 64 * k, nf4 - 1);
                nxt[k] = *reinterpret_cast<const F4u*>(pn + 4 * f);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int f = lane + 64 * k;
                if (f < nf4) *reinterpret_cast<float4*>(my + 36 * (f >> 3) + 4 * (f & 7)) = make_float4(cur[k].v[0], cur[k].v[1], cur[k].v[2], cur[k].v[3]);
            }
            __builtin_amdgcn_wave_barrier();  // (one wave: its LDS operations execute in order)
            if (lane < ntw) {
                float4 r[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) r[k] = *reinterpret_cast<const float4*>(my + 36 * lane + 4 * k);
#pragma unroll
                for (int k = 0; k < 8; ++k) { s += r[k].x; s += r[k].y; s += r[k].z; s += r[k].w; }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 8; ++k) cur[k] = nxt[k];
        }
        if (lane < ntw) (base + (which ? T.ps2 : T.ps1))[ti * T.pw + tj0 + lane] = s;
        return;
    }
    const int unit = ((int)blockIdx.x - tb) * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (unit >= T.units) return;
    const int n_short = T.n_short, ls = T.ls, ss = T.ss, strips = (n_short + 63) / 64, nbseg = T.nbands * T.ppb;
    const double cm = sample_c0(m, (size_t)T.dimu * T.dimv);
    double* Tp = sat + (size_t)blockIdx.z * G.sstride + T.sat_off + 2 + 4 * T.tab + 2 * T.ts + (size_t)(2 * which) * T.nseg * n_short;
    double* Tq = Tp + (size_t)T.nseg * n_short;
    double p = 0.0, q = 0.0;
    int a0, a1;
    if (T.long_contig && unit >= nbseg * strips) {  // one column over all skipped rows, the lanes along a
        const int b = unit - nbseg * strips;
        seg_rows(T, T.ppb, &a0, &a1);
        const float* col = m + (size_t)b * ss;
#pragma unroll 8
        for (int a = a0 + lane; a < a1; a += 64) {
            const double g = (double)col[(size_t)a * ls] - cm;
            p += g;
            q += g * g;
        }
        for (int off = 32; off > 0; off >>= 1) {
            p += __shfl_down(p, off, 64);
            q += __shfl_down(q, off, 64);
        }
        if (lane == 0) { Tp[(size_t)T.ppb * n_short + b] = p; Tq[(size_t)T.ppb * n_short + b] = q; }
        return;
    }
    int sg, strip;
    if (unit < nbseg * strips) {
        const int si = unit / strips;
        strip = unit - si * strips;
        sg = si < T.ppb ? si : T.ppb + T.nch + (si - T.ppb);
    } else {
        const int c = (unit - nbseg * strips) / strips;
        strip = unit - nbseg * strips - c * strips;
        sg = T.ppb + c;
    }
    const int b = strip * 64 + lane;
    if (b >= n_short) return;
    seg_rows(T, sg, &a0, &a1);
    const float* col = m + (size_t)b * ss;
#pragma unroll 16
    for (int a = a0; a < a1; ++a) {
        const double g = (double)col[(size_t)a * ls] - cm;
        p += g;
        q += g * g;
    }
    Tp[(size_t)sg * n_short + b] = p;
    Tq[(size_t)sg * n_short + b] = q;
}

// blockIdx.z = pair, blockIdx.y = 2 * plane + MIP, blockIdx.x = piece of BAND_RC kept rows, or (the last one) the tile-sum table.
// A piece: the thread of column b adds the segment sums that lie before the piece (in order), walks down its rows (running sums of
// g and g^2, all samples requested together); emitted row e then holds, per column, the sum over a' < a(e), and the prefix along b
// is a wave scan over the lanes = columns, the waves' totals meeting in LDS; columns beyond 256 in further rounds with a carry.
__global__ __launch_bounds__(256) void k_band_tables(TabGeom G, const float* __restrict__ fbuf, double* __restrict__ sat) {
    const int plane = blockIdx.y >> 1, which = blockIdx.y & 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const TabPlane& T = G.p[plane];
    const int npieces = T.nbands * T.ppb, pc = blockIdx.x;
    if (pc > npieces) return;
    const float* base = fbuf + (size_t)blockIdx.z * G.pstride;
    const float* m = base + (which ? T.mip2 : T.mip1);
    double* S = sat + (size_t)blockIdx.z * G.sstride + T.sat_off;
    const double cm = sample_c0(m, (size_t)T.dimu * T.dimv);
    if (pc == npieces) {  // c0 and the (ph + 1) x (pw + 1) inclusive table of the tile sums, one wave
        if (wave != 0) return;
        if (lane == 0) S[which] = cm;
        const int ph = T.dimu / TILE, pw = T.dimv / TILE;
        if (!T.tiled || ph * pw == 0) return;
        const float* ps = base + (which ? T.ps2 : T.ps1);
        double* ts = S + 2 + 4 * T.tab + which * T.ts;
        const int tw1 = pw + 1;
        for (int i = lane; i < tw1; i += 64) ts[i] = 0.0;
        for (int r = lane; r <= ph; r += 64) ts[(size_t)r * tw1] = 0.0;
        // The lanes run along the longer side of the tile grid, the shorter side is walked in sequence (its samples requested
        // eight steps at a time): cumulative sums along the lanes by a wave scan, along the walk in the lanes' registers.
        const bool lanes_cols = pw >= ph;
        const int nl = lanes_cols ? pw : ph, nsq = lanes_cols ? ph : pw;         // lanes' extent, steps
        const int lstep = lanes_cols ? 1 : pw, sstep = lanes_cols ? pw : 1;      // strides in ps
        const int olstep = lanes_cols ? 1 : tw1, osstep = lanes_cols ? tw1 : 1;  // strides in ts (entry (r + 1, c + 1))
        double acc[4] = {0.0, 0.0, 0.0, 0.0};  // (tile grids of up to 256 along the lanes: MIPs of up to 8192 pixels)
        for (int q0 = 0; q0 < nsq; q0 += 8) {
            float v[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const int l = min(kb * 64 + lane, nl - 1), q = min(q0 + u, nsq - 1);
                    v[u][kb] = kb * 64 < nl ? ps[(size_t)q * sstep + (size_t)l * lstep] : 0.0f;
                }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (q0 + u >= nsq) break;
                double carry_l = 0.0;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    if (kb * 64 >= nl) break;
                    const int l = kb * 64 + lane;
                    const double sc = wave_scan_dpp(l < nl ? (double)v[u][kb] : 0.0) + carry_l;
                    carry_l = __shfl(sc, 63, 64);
                    acc[kb] += sc;
                    if (l < nl) ts[(size_t)(q0 + u + 1) * osstep + (size_t)(l + 1) * olstep] = acc[kb];
                }
            }
        }
        return;
    }
    __shared__ double tot[2][BAND_RC][4];    // the waves' row totals (p, q)
    __shared__ double carry[2][BAND_RC];     // totals of the column rounds before this one
    const int n_long = T.n_long, n_short = T.n_short, ls = T.ls, ss = T.ss, B = T.B, w1 = n_short + 1;
    const bool full = T.nbands == 1;
    const int band = pc / T.ppb, k = pc - band * T.ppb, per_band = full ? n_long + 1 : B + 1;
    const int e0 = band * per_band + k * BAND_RC, ne = min(BAND_RC, per_band - k * BAND_RC);
    const int a_first = band == 0 ? k * BAND_RC : n_long - B + k * BAND_RC;
    const int nstart = band == 0 ? k : T.ppb + T.nch + k;
    double* P = S + 2 + (size_t)(2 * which) * T.tab;
    double* Q = P + T.tab;
    const double* Tp = S + 2 + 4 * T.tab + 2 * T.ts + (size_t)(2 * which) * T.nseg * n_short;
    const double* Tq = Tp + (size_t)T.nseg * n_short;
    if (tid < 2 * BAND_RC) carry[tid / BAND_RC][tid % BAND_RC] = 0.0;
    __syncthreads();
    for (int b0 = 0; b0 < n_short; b0 += 256) {
        const int b = b0 + tid, bc = min(b, n_short - 1);
        const float* col = m + (size_t)bc * ss;
        float v[BAND_RC];
#pragma unroll
        for (int r = 0; r < BAND_RC; ++r) v[r] = col[(size_t)min(a_first + r, n_long - 1) * ls];
        double rp = 0.0, rq = 0.0;
        for (int s0 = 0; s0 < nstart; s0 += 8) {  // (requested together, added in order)
            double tp[8], tq[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const size_t o = (size_t)min(s0 + c, nstart - 1) * n_short + bc;
                tp[c] = Tp[o];
                tq[c] = Tq[o];
            }
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (s0 + c < nstart) { rp += tp[c]; rq += tq[c]; }
        }
        if (b >= n_short) { rp = 0.0; rq = 0.0; }
        double ep[BAND_RC], eq[BAND_RC];
#pragma unroll
        for (int r = 0; r < BAND_RC; ++r) {
            ep[r] = wave_scan_dpp(rp);
            eq[r] = wave_scan_dpp(rq);
            if (lane == 63) { tot[0][r][wave] = ep[r]; tot[1][r][wave] = eq[r]; }
            if (b < n_short && a_first + r < n_long) {
                const double g = (double)v[r] - cm;
                rp += g;
                rq += g * g;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < BAND_RC; ++r) {
            double op = carry[0][r], oq = carry[1][r];
            for (int w = 0; w < wave; ++w) { op += tot[0][r][w]; oq += tot[1][r][w]; }
            if (r < ne && b < n_short) {
                P[(size_t)(e0 + r) * w1 + b + 1] = ep[r] + op;
                Q[(size_t)(e0 + r) * w1 + b + 1] = eq[r] + oq;
            }
        }
        if (b0 == 0 && tid < ne) { P[(size_t)(e0 + tid) * w1] = 0.0; Q[(size_t)(e0 + tid) * w1] = 0.0; }
        __syncthreads();
        if (tid < 2 * BAND_RC) {
            const int t = tid / BAND_RC, r = tid % BAND_RC;
            carry[t][r] += tot[t][r][0] + tot[t][r][1] + tot[t][r][2] + tot[t][r][3];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ refinement on the device
struct RefineGeom {
    int dimu, dimv, delayu, delayv, wu, wv, Eu, Ev, maxIter, tiled;
    size_t sstride, sat_off, tab, ts;  // doubles: per-pair stride of the table block, offset of this plane, table sizes
    int n_long, n_short, B, long_is_u;  // band geometry of the tables (BandView)
    float margin;
};

// first index of the strict maximum like compute_MAX_ind (compute_funcs.cu:1294-1305: a leading NaN stays the maximum, later NaNs never
// win) + the distance of the runner-up.  All threads return the same values.
__device__ int block_argmax(const float* __restrict__ arr, int len, float* red_v, int* red_i, float* gap) {
    float bv = -FLT_MAX;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < len; i += blockDim.x) {
        const float v = arr[i];
        if (v == v && (v > bv || (v == bv && i < bi) || bi == 0x7fffffff)) { bv = v; bi = i; }
    }
    auto reduce = [&](float& v, int& i) {
        red_v[threadIdx.x] = v;
        red_i[threadIdx.x] = i;
        __syncthreads();
        for (int off = blockDim.x >> 1; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) {
                const float ov = red_v[threadIdx.x + off];
                const int oi = red_i[threadIdx.x + off];
                const float mv = red_v[threadIdx.x];
                const int mi_ = red_i[threadIdx.x];
                if (oi != 0x7fffffff && (mi_ == 0x7fffffff || ov > mv || (ov == mv && oi < mi_))) { red_v[threadIdx.x] = ov; red_i[threadIdx.x] = oi; }
            }
            __syncthreads();
        }
        v = red_v[0];
        i = red_i[0];
        __syncthreads();
    };
    reduce(bv, bi);
    const float a0 = arr[0];
    const int ind = (a0 != a0 || bi == 0x7fffffff) ? 0 : bi;
    // runner-up: the largest value at any other index
    float sv = -FLT_MAX;
    int si = 0x7fffffff;
    for (int i = threadIdx.x; i < len; i += blockDim.x) {
        const float v = arr[i];
        if (i != ind && v == v && (si == 0x7fffffff || v > sv)) { sv = v; si = i; }
    }
    reduce(sv, si);
    *gap = (si == 0x7fffffff || a0 != a0 || bi == 0x7fffffff) ? FLT_MAX : bv - sv;
    return ind;
}

// one work-group per pair: the NCC map of a plane from the cross table + summed-area tables (compute_NCC, :1163-1292), then
// compute_Neighborhood (:1324-1592) entirely on the device.  Returns the final window, du, dv, failed and flags
// (1: an entry outside the transformed lag range was needed, 2: an argmax was decided by less than `margin`).
struct RefineAll {
    RefineGeom g[3];
    const double* cross[3];
    int chunk;  // pairs per plane in the output arrays: window of (plane, pair) at out_win + (plane * chunk + pair) * wcap
};
__global__ __launch_bounds__(256) void k_lag_refine(RefineAll A, const double* __restrict__ sat_base, int wcap, float* __restrict__ out_win,
                                                    int* __restrict__ out_int, float* __restrict__ out_map) {
    const RefineGeom& g = A.g[blockIdx.y];
    const double* cross = A.cross[blockIdx.y];
    out_win += (size_t)blockIdx.y * A.chunk * wcap;
    out_int += (size_t)blockIdx.y * A.chunk * 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lag_lds[];
    __shared__ float red_v[256];
    __shared__ int red_i[256];
    __shared__ int sh_flags;
    const int Hm = 2 * g.delayu + 1, Wm = 2 * g.delayv + 1, H = 2 * g.wu + 1, W = 2 * g.wv + 1;
    float* map = reinterpret_cast<float*>(lag_lds);
    float* win = map + Hm * Wm;
    float* alt = win + H * W;
    const int pair = blockIdx.x;
    const double* sat = sat_base + (size_t)pair * g.sstride + g.sat_off;
    const double *P1 = sat + 2, *Q1 = P1 + g.tab, *P2 = Q1 + g.tab, *Q2 = P2 + g.tab, *T1 = Q2 + g.tab, *T2 = T1 + g.ts;
    const BandView v1{P1, Q1, g.tiled ? T1 : nullptr, sat, g.n_long, g.n_short, g.B, g.long_is_u};
    const BandView v2{P2, Q2, g.tiled ? T2 : nullptr, sat + 1, g.n_long, g.n_short, g.B, g.long_is_u};
    const int CW = 2 * g.Ev + 1;
    const double* cr = cross + (size_t)pair * (2 * g.Eu + 1) * CW;
    if (threadIdx.x == 0) sh_flags = 0;
    __syncthreads();
    auto ncc_at = [&](int u, int v) -> float {
        const float nan = __int_as_float(0x7fc00000);
        const int nr = g.dimu - abs(u), nc = g.dimv - abs(v);
        if (nr <= 0 || nc <= 0) return nan;  // reference: empty loops, 0/0
        if (abs(u) > g.Eu || abs(v) > g.Ev) { atomicOr(&sh_flags, 1); return nan; }
        double fm, sf, F1, tm, st, F2;
        window_stats(v1, g.dimu, g.dimv, max(u, 0), max(v, 0), nr, nc, &fm, &sf, &F1);
        window_stats(v2, g.dimu, g.dimv, max(-u, 0), max(-v, 0), nr, nc, &tm, &st, &F2);
        const double num = cr[(size_t)(u + g.Eu) * CW + (v + g.Ev)] - tm * sf;
        return (F1 > 0.0 && F2 > 0.0) ? (float)(num / sqrt(F1 * F2)) : nan;
    };
    for (int e = threadIdx.x; e < Hm * Wm; e += blockDim.x) {
        const float val = ncc_at(e / Wm - g.delayu, e % Wm - g.delayv);
        map[e] = val;
        if (out_map) out_map[(size_t)pair * Hm * Wm + e] = val;
    }
    __syncthreads();
    float gap;
    int ind_max = block_argmax(map, Hm * Wm, red_v, red_i, &gap);
    int flags = gap < g.margin ? 2 : 0;
    const int initu = min(max(0, ind_max / Wm - g.wu), 2 * (g.delayu - g.wu));
    const int initv = min(max(0, ind_max % Wm - g.wv), 2 * (g.delayv - g.wv));
    for (int e = threadIdx.x; e < H * W; e += blockDim.x) win[e] = map[(initu + e / W) * Wm + initv + e % W];
    int du = initu - g.delayu + g.wu, dv = initv - g.delayv + g.wv;
    ind_max = W * (ind_max / Wm - initu) + (ind_max % Wm - initv);
    const int ind_ref = W * g.wu + g.wv;
    __syncthreads();
    for (int it = 0; it < g.maxIter && ind_max != ind_ref; ++it) {
        const int deltau = ind_max / W - g.wu, deltav = ind_max % W - g.wv;
        du += deltau;
        dv += deltav;
        for (int e = threadIdx.x; e < H * W; e += blockDim.x) {
            const int r = e / W, c = e - r * W, sr = r + deltau, sc = c + deltav;
            alt[e] = (sr >= 0 && sr < H && sc >= 0 && sc < W) ? win[sr * W + sc] : ncc_at(r - g.wu + du, c - g.wv + dv);
        }
        __syncthreads();
        float* t = win; win = alt; alt = t;
        ind_max = block_argmax(win, H * W, red_v, red_i, &gap);
        if (gap < g.margin) flags |= 2;
    }
    int failed = 0;
    if (ind_ref != ind_max) {
        du += ind_max / W - g.wu;
        dv += ind_max % W - g.wv;
        failed = 1;
    }
    for (int e = threadIdx.x; e < H * W; e += blockDim.x) out_win[(size_t)pair * wcap + e] = win[e];
    if (threadIdx.x == 0) {
        int* o = out_int + (size_t)pair * 4;
        o[0] = du;
        o[1] = dv;
        o[2] = failed;
        o[3] = flags | sh_flags;
    }
}

// ------------------------------------------------------------------------------------------------ host side
struct LagPlane {
    bool long_is_u;
    int n_long, n_short, ls, ss, El, Es, Eu, Ev;
    Plan fft;    // along the long axis
    Plan cfft;   // along the short axis (use_corr)
    bool use_corr;
    int KT, JP, FW, TW, LB, nlp, fft_threads;
    size_t lds_fft, lds_mac, lds_corr, lds_refine;
    bool ok;
};

// make_plan searches the LDS image of a length: once per length and process
const Plan& plan_for(int need) {
    static std::mutex& mu = *new std::mutex;
    static std::map<int, Plan>& plans = *new std::map<int, Plan>;  // keyed by the request; several requests may share a length
    std::lock_guard<std::mutex> lock(mu);
    auto it = plans.find(need);
    if (it == plans.end()) it = plans.emplace(need, fft64::make_plan(need)).first;
    return it->second;
}

LagPlane plan_lag_plane(const PlaneGeom& g, int maxIter) {
    LagPlane p{};
    p.long_is_u = g.dimu >= g.dimv;
    p.n_long = p.long_is_u ? g.dimu : g.dimv;
    p.n_short = p.long_is_u ? g.dimv : g.dimu;
    p.ls = p.long_is_u ? g.dimv : 1;
    p.ss = p.long_is_u ? 1 : g.dimv;
    const int dl = p.long_is_u ? g.delayu : g.delayv, ds = p.long_is_u ? g.delayv : g.delayu;
    const int wl = p.long_is_u ? g.wu : g.wv, wsh = p.long_is_u ? g.wv : g.wu;
    // every long-axis lag comes out of the inverse transform: keep all that the re-centring moves can reach; along the short axis a
    // lag costs a correlation: keep one move's worth (a second move beyond it is rare and sends the pair to the careful path)
    p.El = dl + maxIter * wl;
    p.Es = ds + (maxIter > 0 ? wsh : 0);
    p.Eu = p.long_is_u ? p.El : p.Es;
    p.Ev = p.long_is_u ? p.Es : p.El;
    p.ok = p.n_long + p.El <= 8192;
    if (!p.ok) return p;
    p.fft = plan_for(p.n_long + p.El);
    p.lds_fft = sizeof(double) * 2 * ((size_t)p.fft.N + fft64::lds_twiddles(p.fft));
    const int nlag = 2 * p.Es + 1;
    p.LB = 4;  // lags per thread of the direct correlation kernel (8 was measured: twice the registers, bank conflicts, no faster)
    p.nlp = (nlag + p.LB - 1) / p.LB * p.LB;
    // Short axes of 64 lines and more are correlated through a second transform (k_lag_corr), the others lag by lag (k_lag_mac):
    // 32-line planes (the xz / yz MIPs of thin stacks) have too few samples per frequency to fill a transform's work-group.
    p.use_corr = p.n_short >= 64;
    if (p.use_corr) {
        p.cfft = plan_for(p.n_short + p.Es);
        const size_t img = sizeof(double) * 2 * ((size_t)p.cfft.N + 1);
        p.KT = 2 * img * 4 <= 52 * 1024 ? 4 : (2 * img * 2 <= 80 * 1024 ? 2 : 1);  // (three work-groups per CU where the images allow)
        p.lds_corr = 2 * (size_t)p.KT * img + sizeof(double) * 2 * (size_t)fft64::lds_twiddles(p.cfft);
        p.JP = 1; p.FW = p.TW = 0;
        p.lds_mac = 0;
        p.ok = p.lds_corr <= 150 * 1024;
    } else {
        const int PAD = p.Es + p.LB - 1;
        // odd row lengths (in 16-byte slots): the KT rows a wave touches at once start in different banks, and a lag block is four slots
        p.FW = (p.n_short + 2 * PAD + p.LB) | 1;
        p.TW = (p.n_short + p.LB) | 1;
        const size_t row = sizeof(double) * 2 * (size_t)(p.FW + p.TW);
        // four frequencies per work-group: with rows one 16-byte slot apart and lanes ordered (frequency fastest, then lag block) the
        // 16-byte LDS reads of a wave are conflict-free; the j range is cut into JP parts so that all 256 threads have an item
        p.KT = (int)std::min<size_t>(4, (48 * 1024) / row);
        if (p.KT < 1) p.KT = 1;
        const int nvb = p.nlp / p.LB;
        while (p.KT > 1 && p.KT * nvb > 256) --p.KT;
        p.JP = std::max(1, std::min(8, 256 / (p.KT * nvb)));
        p.lds_mac = row * p.KT + sizeof(double) * 2 * p.LB * (size_t)(p.JP - 1) * nvb * p.KT;
        p.ok = p.lds_mac <= 150 * 1024 && nvb <= 256;
    }
    p.fft_threads = p.fft.N >= 1024 ? 256 : 128;
    const int Hm = 2 * g.delayu + 1, Wm = 2 * g.delayv + 1, H = 2 * g.wu + 1, W = 2 * g.wv + 1;
    p.lds_refine = sizeof(float) * ((size_t)Hm * Wm + 2 * (size_t)H * W);
    p.ok = p.ok && p.fft.N <= 8192 && p.lds_refine <= 120 * 1024;
    return p;
}

// Per (device, N), kept for the life of the process: the stages' twiddle tables in fp64 and the slot tables of the half spectrum
// (LDS slot of the s-th frequency in ascending logical position, LDS slot of its mirror frequency, slot of a frequency).
struct FftTables {
    const cplx* tw;
    const int *slot_pos, *slot_neg, *freq_slot;
    const Plan* plan;  // the plan itself, for the kernels
};
struct TwiddleKey { int dev, n; bool operator<(const TwiddleKey& o) const { return dev != o.dev ? dev < o.dev : n < o.n; } };
std::mutex& g_tw_mu = *new std::mutex;
std::map<TwiddleKey, FftTables>& g_tw = *new std::map<TwiddleKey, FftTables>;

int fft_tables(int dev, const Plan& pl, hipStream_t s, FftTables* out) {
    std::lock_guard<std::mutex> lock(g_tw_mu);
    auto it = g_tw.find(TwiddleKey{dev, pl.N});
    if (it != g_tw.end()) { *out = it->second; return MI_OK; }
    const size_t N = (size_t)pl.N, NK = N / 2 + 1;
    const std::vector<double> hs = fft64::make_twiddles(pl);
    // Which frequency sits in which slot is free (the correlation kernels do not care, the inverse transform looks slots up), so the
    // order is chosen for the untangling pass of k_lag_fwd: lane l of a wave reads Z[k] and Z[N - k] of slot 64 w + l with 16-byte
    // LDS reads, which are served in four groups of 16 lanes over 16 bank quads -- every group gets 16 frequencies whose Z[k] fall
    // into 16 different quads AND whose Z[N - k] do.  Frequencies are bucketed by the two quads; a group is a perfect matching
    // between the quads of the first read and those of the second (Kuhn's augmenting paths on a 16 x 16 graph), fullest buckets
    // first; what cannot be matched at the end fills the last slots.  (In position order the first read was 2-way, the second up
    // to 4-way conflicted: SQ_LDS_BANK_CONFLICT 14 % of the LDS cycles of the transform.)
    std::vector<int> p1(NK), p2(NK);
    std::vector<std::vector<int>> bucket(256);
    for (size_t k = 0; k < NK; ++k) {
        p1[k] = fft64::phys(pl, fft64::pos_of_freq(pl, (int)k));
        p2[k] = fft64::phys(pl, fft64::pos_of_freq(pl, (int)((N - k) % N)));
        bucket[(size_t)(p1[k] & 15) * 16 + (size_t)(p2[k] & 15)].push_back((int)k);
    }
    static const int lanes_of_group[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                               {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                               {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                               {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    std::vector<int> slot_freq(NK, -1);
    size_t placed = 0;
    for (size_t w = 0; w * 64 < NK && placed < NK; ++w)
        for (int g = 0; g < 4 && placed < NK; ++g) {
            int match_of_b2[16];
            for (int& m : match_of_b2) m = -1;
            // Kuhn: first-read quads in turn, their edges tried in the order of bucket size
            for (int a = 0; a < 16; ++a) {
                bool seen[16] = {false};
                struct Rec {
                    static bool go(int a_, const std::vector<std::vector<int>>& bk, int* mb, bool* sn) {
                        int ord[16];
                        for (int i = 0; i < 16; ++i) ord[i] = i;
                        std::sort(ord, ord + 16, [&](int x, int y) { return bk[(size_t)a_ * 16 + x].size() > bk[(size_t)a_ * 16 + y].size(); });
                        for (int oi = 0; oi < 16; ++oi) {
                            const int b = ord[oi];
                            if (bk[(size_t)a_ * 16 + b].empty() || sn[b]) continue;
                            sn[b] = true;
                            if (mb[b] < 0 || go(mb[b], bk, mb, sn)) { mb[b] = a_; return true; }
                        }
                        return false;
                    }
                };
                (void)Rec::go(a, bucket, match_of_b2, seen);
            }
            int li = 0;
            bool used_lane[16] = {false};
            for (int b = 0; b < 16; ++b) {
                if (match_of_b2[b] < 0) continue;
                std::vector<int>& bk = bucket[(size_t)match_of_b2[b] * 16 + b];
                const size_t sl = w * 64 + (size_t)lanes_of_group[g][li];
                if (sl >= NK) continue;  // (the last, partial wave)
                slot_freq[sl] = bk.back();
                bk.pop_back();
                used_lane[li++] = true;
                ++placed;
            }
            (void)used_lane;
            // lanes the matching left empty are filled at the end
        }
    {
        std::vector<int> rest;
        for (auto& bk : bucket) rest.insert(rest.end(), bk.begin(), bk.end());
        size_t ri = 0;
        for (size_t sl = 0; sl < NK; ++sl)
            if (slot_freq[sl] < 0) slot_freq[sl] = rest[ri++];
    }
    std::vector<int> tabs(3 * NK);
    for (size_t sl = 0; sl < NK; ++sl) {
        const size_t k = (size_t)slot_freq[sl];
        tabs[sl] = p1[k];
        tabs[NK + sl] = p2[k];
        tabs[2 * NK + k] = (int)sl;
    }
    void *d = nullptr, *di = nullptr, *dp = nullptr;
    MI_HIP(hipMalloc(&d, sizeof(double) * hs.size()));
    MI_HIP(hipMalloc(&di, sizeof(int) * 3 * NK));
    MI_HIP(hipMalloc(&dp, sizeof(Plan)));
    MI_HIP(hipMemcpyAsync(d, hs.data(), sizeof(double) * hs.size(), hipMemcpyHostToDevice, s));
    MI_HIP(hipMemcpyAsync(di, tabs.data(), sizeof(int) * 3 * NK, hipMemcpyHostToDevice, s));
    MI_HIP(hipMemcpyAsync(dp, &pl, sizeof(Plan), hipMemcpyHostToDevice, s));
    MI_HIP(hipStreamSynchronize(s));
    const int* ti = static_cast<const int*>(di);
    FftTables t{static_cast<const cplx*>(d), ti, ti + NK, ti + 2 * NK, static_cast<const Plan*>(dp)};
    g_tw[TwiddleKey{dev, pl.N}] = t;
    *out = t;
    return MI_OK;
}

// device + pinned buffers of the batched pipeline, kept between calls (one set per concurrent caller and device)
struct LagWorkspace {
    int dev = -1;
    DevBuf fbuf, sat, mip_tmp, xyT, SP[3], CH[3], cross[3], outw, outi, tab;  // (lag-transform scratch per plane: the planes' chains run side by side)
    PinnedBuf pin_tab, pin_w, pin_i;
    // Three streams PER DEVICE, shared by every group in flight: `sm` the MIP pass (k_mips: one HBM-bound streaming read of both
    // overlap views), `sa` the lag transform of the xy plane and the refinement of all planes, `sb` the tables of all planes and
    // the lag transforms of the two thin planes.  (A set of streams per workspace mapped onto the same hardware queues in a way
    // that put one group's MIP pass behind the other group's chain: profiles/r03_ncc_timeline.txt.)
    hipStream_t sm = nullptr, sa = nullptr, sb = nullptr;
    hipEvent_t ev_start = nullptr, ev_lag = nullptr, ev_done = nullptr, ev_side = nullptr;
    hipEvent_t ev_head_tab = nullptr;
    hipEvent_t ev_head = nullptr;  // per DEVICE like the streams (not owned): "the memory-bound head of the latest xy chain has run"
    std::vector<hipEvent_t> ev_mip, ev_mip_xy;  // per piece: all six MIPs of its pairs final / the xy MIPs final
    int wait_done() {
        MI_HIP(hipEventSynchronize(ev_done));
        return MI_OK;
    }
    ~LagWorkspace() {
        if (ev_start) (void)hipEventDestroy(ev_start);
        if (ev_lag) (void)hipEventDestroy(ev_lag);
        if (ev_done) (void)hipEventDestroy(ev_done);
        if (ev_side) (void)hipEventDestroy(ev_side);
        for (hipEvent_t e : ev_mip) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_mip_xy) (void)hipEventDestroy(e);
    }
    int streams(size_t pieces) {
        {
            static std::mutex mu;
            struct PerDev { hipStream_t s[3]; hipEvent_t head, head_tab; };
            static std::map<int, PerDev> per_dev;  // (never destroyed: the process' lifetime)
            std::lock_guard<std::mutex> lock(mu);
            auto it = per_dev.find(dev);
            if (it == per_dev.end()) {
                PerDev f{};
                int chains = 2;
                if (const char* e = MI_PROBE_ENV("MI_NCC_CHAIN_STREAMS")) chains = std::max(1, std::min(2, std::atoi(e)));
                // (stream priorities were measured: no effect on a call, profiles/r04_ncc_notes.txt)
                for (int i = 0; i < 1 + chains; ++i) MI_HIP(hipStreamCreateWithFlags(&f.s[i], hipStreamNonBlocking));
                for (int i = 1 + chains; i < 3; ++i) f.s[i] = f.s[chains];
                MI_HIP(hipEventCreateWithFlags(&f.head, hipEventDisableTiming));
                MI_HIP(hipEventCreateWithFlags(&f.head_tab, hipEventDisableTiming));
                it = per_dev.emplace(dev, f).first;
            }
            sm = it->second.s[0];
            sa = it->second.s[1];
            sb = it->second.s[2];
            ev_head = it->second.head;
            ev_head_tab = it->second.head_tab;
        }
        if (!ev_side) MI_HIP(hipEventCreateWithFlags(&ev_side, hipEventDisableTiming));
        if (!ev_start) MI_HIP(hipEventCreateWithFlags(&ev_start, hipEventDisableTiming));
        if (!ev_done) MI_HIP(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
        if (!ev_lag) MI_HIP(hipEventCreateWithFlags(&ev_lag, hipEventDisableTiming));
        while (ev_mip.size() < pieces) {
            hipEvent_t e = nullptr;
            MI_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ev_mip.push_back(e);
            hipEvent_t e2 = nullptr;
            MI_HIP(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
            ev_mip_xy.push_back(e2);
        }
        return MI_OK;
    }
};
std::mutex& g_lag_mu = *new std::mutex;
std::vector<std::unique_ptr<LagWorkspace>>& g_lag_ws = *new std::vector<std::unique_ptr<LagWorkspace>>;

std::unique_ptr<LagWorkspace> take_lag_ws(int dev) {
    {
        std::lock_guard<std::mutex> lock(g_lag_mu);
        for (size_t i = 0; i < g_lag_ws.size(); ++i)
            if (g_lag_ws[i]->dev == dev) {
                std::unique_ptr<LagWorkspace> r = std::move(g_lag_ws[i]);
                g_lag_ws.erase(g_lag_ws.begin() + i);
                return r;
            }
    }
    std::unique_ptr<LagWorkspace> r(new (std::nothrow) LagWorkspace);
    if (r) r->dev = dev;
    return r;
}

void give_lag_ws(std::unique_ptr<LagWorkspace> r) {
    if (!r) return;
    std::lock_guard<std::mutex> lock(g_lag_mu);
    if (g_lag_ws.size() < 4) g_lag_ws.push_back(std::move(r));
}

int grow(DevBuf& b, size_t bytes) { return b.bytes >= bytes ? MI_OK : b.alloc(bytes); }

// cross terms of `np` pairs of one plane through the lag transform (MIPs at m1 / m2 + q * pstride) into ws.cross
int lag_cross(int dev, hipStream_t s, const LagPlane& lp, const float* m1, const float* m2, size_t pstride, int np, LagWorkspace& ws, int m = 0,
              hipEvent_t after_fwd = nullptr) {
    const int N = lp.fft.N, NK = N / 2 + 1, nlag = 2 * lp.Es + 1, nlp = lp.nlp;
    const int tiles_per_pair = ((NK + lp.KT - 1) / lp.KT + 1) & ~1, ntiles = tiles_per_pair * np;  // (even: see k_lag_mac)
    FftTables ft;
    MI_TRY(fft_tables(dev, lp.fft, s, &ft));
    MI_TRY(grow(ws.SP[m], sizeof(double) * 2 * (size_t)ntiles * lp.n_short * 2 * lp.KT));
    MI_TRY(grow(ws.CH[m], sizeof(double) * 2 * (size_t)np * NK * nlp));
    MI_TRY(grow(ws.cross[m], sizeof(double) * (size_t)np * (2 * lp.Eu + 1) * (2 * lp.Ev + 1)));
    const bool big = lp.fft_threads == 256;
    {
        auto fwd = big ? k_lag_fwd<256> : k_lag_fwd<128>;
        if (lp.lds_fft > 64 * 1024)
            MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp.lds_fft));
        hipLaunchKernelGGL(fwd, dim3(8 * ((lp.n_short + 7) / 8), np), dim3(lp.fft_threads), lp.lds_fft, s, m1, m2, pstride, lp.n_long, lp.n_short, lp.ls,
                           lp.ss, lp.KT, tiles_per_pair, ft.plan, ft.tw, ft.slot_pos, ft.slot_neg, ws.SP[m].as<cplx>());
        MI_TRY(launch_check("k_lag_fwd"));
    }
    if (after_fwd) MI_HIP(hipEventRecord(after_fwd, s));
    if (lp.use_corr) {
        FftTables ct;
        MI_TRY(fft_tables(dev, lp.cfft, s, &ct));
        auto corr = lp.KT == 4 ? k_lag_corr<256, 4> : (lp.KT == 2 ? k_lag_corr<256, 2> : k_lag_corr<256, 1>);
        if (lp.lds_corr > 64 * 1024)
            MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(corr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp.lds_corr));
        hipLaunchKernelGGL(corr, dim3(tiles_per_pair, np), dim3(256), lp.lds_corr, s, ws.SP[m].as<cplx>(), lp.n_short, NK, lp.Es, nlp, tiles_per_pair,
                           ct.plan, ct.tw, ws.CH[m].as<cplx>());
        MI_TRY(launch_check("k_lag_corr"));
    } else {
        using MacFn = void (*)(const cplx*, int, int, int, int, int, int, int, int, int, int, cplx*);
        static const MacFn macs[] = {k_lag_mac<4, 0>, k_lag_mac<4, 1>, k_lag_mac<4, 2>, k_lag_mac<4, 3>, k_lag_mac<4, 4>, k_lag_mac<4, 5>, k_lag_mac<4, 6>};
        const int pfn = (lp.n_short * lp.KT + 255) / 256;  // float4 pairs per thread of a tile; beyond the table: no register prefetch
        MacFn mac = macs[pfn <= 6 ? pfn : 0];
        int cus = 256, per_cu = 1;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (lp.lds_mac > 64 * 1024)
            MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mac), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp.lds_mac));
        per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lp.lds_mac, 1)));
        const int grid = std::max(16, std::min((ntiles + 15) & ~15, (cus * per_cu) & ~15));
        hipLaunchKernelGGL(mac, dim3(grid), dim3(256), lp.lds_mac, s, ws.SP[m].as<cplx>(), lp.n_short, NK, lp.Es, lp.KT, lp.JP, lp.FW, lp.TW, nlp,
                           tiles_per_pair, ntiles, ws.CH[m].as<cplx>());
        MI_TRY(launch_check("k_lag_mac"));
    }
    {
        auto inv = big ? k_lag_inv<256> : k_lag_inv<128>;
        if (lp.lds_fft > 64 * 1024)
            MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(inv), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp.lds_fft));
        const int lagpairs = (nlag + 1) / 2;
        hipLaunchKernelGGL(inv, dim3(8 * ((np + 7) / 8) * lagpairs), dim3(lp.fft_threads), lp.lds_fft, s, ws.CH[m].as<cplx>(), NK, nlp, ft.plan, lp.Es,
                           lp.El, lp.long_is_u ? 1 : 0, lp.Eu, lp.Ev, ft.tw, ft.freq_slot, np, lagpairs, ws.cross[m].as<double>());
    }
    return launch_check("k_lag_inv");
}

// layout of one plane's banded tables (doubles): c0a, c0b | P1 | Q1 | P2 | Q2 | TS1 | TS2 | column sums of the row segments
struct BandLayout {
    int B, rows, ppb, nbands, nch, nseg;
    size_t tab, ts, total;
    BandLayout(const PlaneGeom& g, const LagPlane& lp) {
        B = lp.El + TILE;                                   // rows read: within E (+ 31 for the tile-aligned ones) of either end
        if (2 * (B + 1) >= lp.n_long + 1) B = lp.n_long;    // no gain: keep every row
        const bool full = B >= lp.n_long;
        rows = full ? lp.n_long + 1 : 2 * (B + 1);
        nbands = full ? 1 : 2;
        ppb = ((full ? lp.n_long + 1 : B + 1) + BAND_RC - 1) / BAND_RC;
        const int skipped = full ? 0 : lp.n_long - 2 * B;
        nch = lp.ls == 1 ? (skipped > 0 ? 1 : 0) : (skipped + BAND_CH - 1) / BAND_CH;  // (long axis contiguous: one sum per column)
        nseg = ppb + nch + (nbands == 2 ? ppb : 0);
        tab = (size_t)rows * (lp.n_short + 1);
        ts = (size_t)(g.dimu / TILE + 1) * (g.dimv / TILE + 1);
        total = 2 + 4 * tab + 2 * ts + 4 * (size_t)nseg * lp.n_short;
    }
};

TabPlane tab_plane(const PlaneGeom& g, const LagPlane& lp, size_t sat_off) {
    const BandLayout L(g, lp);
    TabPlane t{};
    t.dimu = g.dimu; t.dimv = g.dimv; t.tiled = g.tiled ? 1 : 0;
    t.pw = g.dimv / TILE; t.nt = (g.dimu / TILE) * t.pw;
    t.n_long = lp.n_long; t.n_short = lp.n_short; t.ls = lp.ls; t.ss = lp.ss; t.B = L.B; t.rows = L.rows; t.long_contig = lp.ls == 1 ? 1 : 0;
    t.ppb = L.ppb; t.nbands = L.nbands; t.nch = L.nch; t.nseg = L.nseg;
    const int strips = (lp.n_short + 63) / 64;
    t.units = L.nbands * L.ppb * strips + (t.long_contig ? (L.nch ? lp.n_short : 0) : L.nch * strips);
    t.mip1 = g.mip1; t.mip2 = g.mip2; t.ps1 = g.ps1; t.ps2 = g.ps2;
    t.sat_off = sat_off; t.tab = L.tab; t.ts = L.ts;
    return t;
}

// float tile sums (reference order), shift constants and the banded tables of both MIPs of `nplanes` planes, `np` pairs at once
int prepare_tables(hipStream_t s, const TabGeom& G, float* fbuf, double* sat, int np) {
    int tb = 1, ub = 0, pieces = 0;
    for (int m = 0; m < G.nplanes; ++m) {
        tb = std::max(tb, ((G.p[m].dimu / TILE) * ((G.p[m].pw + 63) / 64) + 3) / 4);
        ub = std::max(ub, (G.p[m].units + 3) / 4);
        pieces = std::max(pieces, G.p[m].nbands * G.p[m].ppb);
    }
    static const int knock = [] { const char* e = MI_PROBE_ENV("MI_NCC_TAB_KNOCK"); return e ? std::atoi(e) : 0; }();
    hipLaunchKernelGGL(k_plane_sums, dim3(tb + ub, 2 * G.nplanes, np), dim3(256), 0, s, G, fbuf, sat, tb, knock);
    MI_TRY(launch_check("k_plane_sums"));
    hipLaunchKernelGGL(k_band_tables, dim3(pieces + 1, 2 * G.nplanes, np), dim3(256), 0, s, G, fbuf, sat);
    return launch_check("k_band_tables");
}

RefineGeom refine_geom(const PlaneGeom& g, const LagPlane& lp, int maxIter, size_t sstride, size_t sat_off, float margin) {
    const BandLayout L(g, lp);
    RefineGeom r{};
    r.dimu = g.dimu; r.dimv = g.dimv; r.delayu = g.delayu; r.delayv = g.delayv; r.wu = g.wu; r.wv = g.wv;
    r.Eu = lp.Eu; r.Ev = lp.Ev; r.maxIter = maxIter; r.tiled = g.tiled ? 1 : 0;
    r.sstride = sstride; r.sat_off = sat_off; r.tab = L.tab; r.ts = L.ts; r.margin = margin;
    r.n_long = lp.n_long; r.n_short = lp.n_short; r.B = L.B; r.long_is_u = lp.long_is_u ? 1 : 0;
    return r;
}

}  // namespace

namespace mi {

float ncc_margin() { return decision_margin(); }

bool ncc_lag_supported(int dimk, int dimi, int dimj, int ni, int nj, int delayk, int delayi, int delayj, int side, const mi_ncc_params* p) {
    mi_ncc_params q = *p;
    PairPlan pl;
    if (plan_pair(dimk, dimi, dimj, 0, ni, nj, delayk, delayi, delayj, side, &q, pl) != MI_OK) return false;
    for (int m = 0; m < 3; ++m)
        if (!plan_lag_plane(pl.g[m], q.maxIter).ok) return false;
    return true;
}

// A group in flight: n pairs of ONE geometry (same side, nominal offsets and parameters) whose device stage has been enqueued.
struct LagJob {
    std::unique_ptr<LagWorkspace> ws;
    PairPlan pl;
    LagPlane lp[3];
    int n = 0, wcap = 1, ni = 0, nj = 0, side = 0;
    float margin = 0.0f;
    bool enqueued = false;  // ws->ev_done marks the end of this job's device stage
    // chains deferred (ncc_lag_enqueue_chains): what they need
    bool chains_pending = false;
    int dev = 0, chunk = 0, maxIter = 0;
    size_t pstride = 0, sstride = 0, sat_off[3] = {0, 0, 0}, tstride = 0;
    ~LagJob() {
        if (ws) {
            // (whatever the call that owned this job enqueued must not outlive the buffers' next user)
            if (enqueued && ws->ev_done) (void)ws->wait_done();
            else if (ws->sm) {
                (void)hipStreamSynchronize(ws->sm);
                (void)hipStreamSynchronize(ws->sa);
                (void)hipStreamSynchronize(ws->sb);
            }
            give_lag_ws(std::move(ws));
        }
    }
};

// Device stage of a group: enqueued behind the work `s` holds so far, on the device's MIP stream and its two chain streams.
// What overlaps: the MIP pass of the NEXT group with the chains of this one, and the xy plane's lag transform with the tables and
// the thin planes' transforms.
static int enqueue_chains(LagJob& job, int c0, int p0, int np, int pi, hipEvent_t gate, hipEvent_t gate_xy = nullptr);
static int close_job(LagJob& job);

// MI_NCC_GATE=1 (probe builds): a group's MIP pass starts only when the previous group's chain is past its tables and its xy forward
// transform.  Rounds 3-4 ran that way -- those kernels are bound by memory latency and waited for whole work-groups of the pass to
// retire before they found a compute unit (profiles/r03_ncc_timeline.txt).  Since round 5 a pass that runs beside a chain keeps two
// work-groups per compute unit (launch_mips), the chain's work-groups are resident next to them, and the gate costs more than it
// saves: 5.96 against 6.07 ms per 112 pairs (profiles/r05_ncc_sched.txt).
static bool mip_gate() {
    static const bool on = [] {
        const char* e = MI_PROBE_ENV("MI_NCC_GATE");
        return e ? std::atoi(e) != 0 : false;
    }();
    return on;
}

int ncc_lag_enqueue(int dev, hipStream_t s, int n, const float* const* a_ptrs, const float* const* b_ptrs, int dimk, int dimi, int dimj, int ni,
                    int nj, int delayk, int delayi, int delayj, int side, mi_ncc_params* params, LagJob** job_out, bool defer_chains, TileFmt fmt,
                    int groups_in_flight, bool chain_beside) {
    *job_out = nullptr;
    if (n <= 0) return MI_OK;
    std::unique_ptr<LagJob> job(new (std::nothrow) LagJob);
    if (!job) return fail(MI_ERR_NOMEM, "mi_ncc_mips_batch: out of host memory");
    PairPlan& pl = job->pl;
    for (int q = 0; q < n; ++q) MI_TRY(plan_pair(dimk, dimi, dimj, 0, ni, nj, delayk, delayi, delayj, side, &params[q], pl));
    const mi_ncc_params& P = params[0];
    LagPlane* lp = job->lp;
    for (int m = 0; m < 3; ++m) {
        lp[m] = plan_lag_plane(pl.g[m], P.maxIter);
        MI_REQUIRE(lp[m].ok, "mi_ncc_mips_batch: plane %d does not fit the lag transform", m);
    }
    job->ws = take_lag_ws(dev);
    if (!job->ws) return fail(MI_ERR_NOMEM, "mi_ncc_mips_batch: out of host memory");
    LagWorkspace& ws = *job->ws;
    job->n = n; job->ni = ni; job->nj = nj; job->side = side;

    // per-pair footprint -> chunk size
    size_t sat_off[3], sstride = 0;
    for (int m = 0; m < 3; ++m) {
        sat_off[m] = sstride;
        sstride += BandLayout(pl.g[m], lp[m]).total;
    }
    const size_t pstride = (pl.total_floats + 3) / 4 * 4;
    const size_t tmp_floats = mips_tmp_floats(pl.dimk, pl.dimi_v, pl.dimj_v);
    size_t spec = 0, crs = 0;
    int wcap = 1;
    for (int m = 0; m < 3; ++m) {
        const size_t NK = (size_t)lp[m].fft.N / 2 + 1, nlp = (size_t)lp[m].nlp, kt = (size_t)lp[m].KT;
        spec += 16 * (2 * (size_t)lp[m].n_short * kt * (((NK + kt - 1) / kt + 1) & ~(size_t)1) + NK * nlp);
        crs += 8 * (size_t)(2 * lp[m].Eu + 1) * (2 * lp[m].Ev + 1);
        wcap = std::max(wcap, (2 * pl.g[m].wu + 1) * (2 * pl.g[m].wv + 1));
    }
    job->wcap = wcap;
    const size_t per_pair = 4 * (pstride + tmp_floats) + 8 * sstride + spec + crs + 3 * 4 * (size_t)wcap + 64;
    // (every group of a batch is enqueued before the first is waited for, each with a workspace of its own: they share the budget)
    size_t budget = std::max((size_t)1 << 30, ((size_t)6 << 30) / (size_t)std::max(1, groups_in_flight));
    if (const char* e = std::getenv("MI_NCC_CHUNK_MB")) budget = (size_t)std::max(1, std::atoi(e)) << 20;
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, budget / per_pair));
    // (pieces of a chunk -- the MIP pass of piece i + 1 beside the chain of piece i -- were measured again on the round-4 chain: 6.6 / 6.9 /
    // 7.1 / 7.2 ms per 112 pairs for 1 / 2 / 3 / 4 pieces, uint16 tiles 4.6 / 5.6 / 5.9 / 6.0: more launches, more contention)
    int piece = chunk;
    if (const char* e = MI_PROBE_ENV("MI_NCC_PIECES")) piece = std::max(1, (chunk + std::max(1, std::atoi(e)) - 1) / std::max(1, std::atoi(e)));
    if (const char* e = MI_PROBE_ENV("MI_NCC_PIECES_BESIDE"))
        if (chain_beside) piece = std::max(1, (chunk + std::max(1, std::atoi(e)) - 1) / std::max(1, std::atoi(e)));

    MI_TRY(grow(ws.fbuf, 4 * pstride * chunk));
    MI_TRY(grow(ws.sat, 8 * sstride * chunk));
    MI_TRY(grow(ws.mip_tmp, 4 * tmp_floats * chunk));
    // xy MIPs whose long axis is the row index (west-east views) are also kept transposed, for the lag transform
    const size_t tstride = lp[0].ls == 1 ? 0 : ((size_t)pl.g[0].dimu * pl.g[0].dimv + 3) / 4 * 4;
    job->tstride = tstride;
    if (tstride) MI_TRY(grow(ws.xyT, 4 * 2 * tstride * chunk));
    MI_TRY(grow(ws.outw, 4 * (size_t)3 * wcap * chunk));
    MI_TRY(grow(ws.outi, sizeof(int) * 3 * 4 * chunk));
    MI_TRY(grow(ws.tab, sizeof(void*) * 2 * chunk));
    // host staging for ALL chunks: the single synchronisation comes after the last chunk
    MI_TRY(ws.pin_tab.reserve(sizeof(void*) * 2 * (size_t)n));
    MI_TRY(ws.pin_w.reserve(4 * (size_t)3 * wcap * n));
    MI_TRY(ws.pin_i.reserve(sizeof(int) * 3 * 4 * (size_t)n));
    MI_TRY(ws.streams((size_t)(chunk + piece - 1) / piece));
    const float** htab = ws.pin_tab.as<const float*>();
    for (int q = 0; q < n; ++q) {
        MI_REQUIRE(a_ptrs[q] && b_ptrs[q], "mi_ncc_mips_batch: null tile");
        htab[2 * q] = a_ptrs[q];
        htab[2 * q + 1] = b_ptrs[q];
    }
    job->margin = ncc_margin();
    job->dev = dev; job->chunk = chunk; job->maxIter = P.maxIter; job->pstride = pstride; job->sstride = sstride;
    for (int m = 0; m < 3; ++m) job->sat_off[m] = sat_off[m];
    const bool defer = defer_chains && chunk >= n && piece >= chunk;  // (one chunk, one piece: nothing of the chains is needed earlier)
    float* base0 = ws.fbuf.as<float>();
    hipStream_t sm = ws.sm;
    MI_HIP(hipEventRecord(ws.ev_start, s));
    MI_HIP(hipStreamWaitEvent(sm, ws.ev_start, 0));
    MI_HIP(hipStreamWaitEvent(ws.sa, ws.ev_start, 0));
    if (ws.sb != ws.sa) MI_HIP(hipStreamWaitEvent(ws.sb, ws.ev_start, 0));
    if (mip_gate()) {  // (never recorded yet: no wait)
        MI_HIP(hipStreamWaitEvent(sm, ws.ev_head, 0));
        MI_HIP(hipStreamWaitEvent(sm, ws.ev_head_tab, 0));
    }
    for (int c0 = 0; c0 < n; c0 += chunk) {
        const int nc = std::min(chunk, n - c0);
        if (c0 > 0) {  // the buffers of the previous chunk are free once its chains have run (the refinement on `sa` is their end)
            MI_HIP(hipEventRecord(ws.ev_lag, ws.sa));
            MI_HIP(hipStreamWaitEvent(sm, ws.ev_lag, 0));
        }
        for (int p0 = 0, pi = 0; p0 < nc; p0 += piece, ++pi) {
            const int np = std::min(piece, nc - p0);
            float* base = base0 + (size_t)p0 * pstride;
            const float** dtab = ws.tab.as<const float*>() + 2 * (size_t)p0;
            MI_HIP(hipMemcpyAsync(dtab, htab + 2 * (size_t)(c0 + p0), sizeof(void*) * 2 * np, hipMemcpyHostToDevice, sm));
            MI_TRY(launch_mips(sm, nullptr, nullptr, dtab, np, pstride, pl.dimk, pl.dimi_v, pl.dimj_v, (size_t)dimi * dimj, dimj, pl.ai0, pl.aj0,
                               base + pl.g[0].mip1, base + pl.g[1].mip1, base + pl.g[2].mip1, base + pl.g[0].mip2, base + pl.g[1].mip2,
                               base + pl.g[2].mip2, ws.mip_tmp.as<float>() + (size_t)p0 * tmp_floats, ws.ev_mip_xy[pi], fmt,
                               tstride ? ws.xyT.as<float>() + 2 * tstride * (size_t)p0 : nullptr, tstride, chain_beside || c0 > 0 || p0 > 0));
            MI_HIP(hipEventRecord(ws.ev_mip[pi], sm));
            if (defer) continue;
            MI_TRY(enqueue_chains(*job, c0, p0, np, pi, ws.ev_mip[pi], ws.ev_mip_xy[pi]));
        }
    }
    job->chains_pending = defer;
    if (!defer) MI_TRY(close_job(*job));
    *job_out = job.release();
    return MI_OK;
}

// the table / lag-transform / refinement chain of the pairs [p0, p0 + np) of a chunk behind `gate` (gate_xy: an earlier event
// that the xy plane's transform may start behind -- its MIPs come straight out of k_mips)
static int enqueue_chains(LagJob& job, int c0, int p0, int np, int pi, hipEvent_t gate, hipEvent_t gate_xy) {
    (void)pi;
    if (!gate_xy) gate_xy = gate;
    LagWorkspace& ws = *job.ws;
    const PairPlan& pl = job.pl;
    const LagPlane* lp = job.lp;
    const int n = job.n, wcap = job.wcap, chunk = job.chunk;
    const size_t pstride = job.pstride, sstride = job.sstride;
    float* base = ws.fbuf.as<float>() + (size_t)p0 * pstride;
    double* sat_p = ws.sat.as<double>() + (size_t)p0 * sstride;
    hipStream_t sa = ws.sa, sb = ws.sb;
    // `sb`: the tables of the three planes (two launches), then the lag transforms of the thin planes
    MI_HIP(hipStreamWaitEvent(sb, gate, 0));
    TabGeom G{};
    G.nplanes = 3; G.pstride = pstride; G.sstride = sstride;
    for (int m = 0; m < 3; ++m) G.p[m] = tab_plane(pl.g[m], lp[m], job.sat_off[m]);
    MI_TRY(prepare_tables(sb, G, base, sat_p, np));
    if (mip_gate()) MI_HIP(hipEventRecord(ws.ev_head_tab, sb));
    for (int m = 1; m < 3; ++m) MI_TRY(lag_cross(job.dev, sb, lp[m], base + pl.g[m].mip1, base + pl.g[m].mip2, pstride, np, ws, m));
    // `sa`: the xy plane's lag transform
    if (sa != sb) MI_HIP(hipStreamWaitEvent(sa, gate_xy, 0));
    if (job.tstride) {  // the transposed copies: line j of MIP 1 / 2 of pair q at xyT + (2 q) / (2 q + 1) * tstride + j * dimu
        LagPlane lt = lp[0];
        lt.ls = 1;
        lt.ss = lp[0].n_long;
        const float* t1 = ws.xyT.as<float>() + 2 * job.tstride * (size_t)p0;
        MI_TRY(lag_cross(job.dev, sa, lt, t1, t1 + job.tstride, 2 * job.tstride, np, ws, 0, mip_gate() ? ws.ev_head : nullptr));
    } else {
        MI_TRY(lag_cross(job.dev, sa, lp[0], base + pl.g[0].mip1, base + pl.g[0].mip2, pstride, np, ws, 0, mip_gate() ? ws.ev_head : nullptr));
    }
    if (sa != sb) {
        MI_HIP(hipEventRecord(ws.ev_side, sb));
        MI_HIP(hipStreamWaitEvent(sa, ws.ev_side, 0));
    }
    // the refinement of the three planes: one launch, one work-group per (pair, plane)
    RefineAll R{};
    size_t lds_refine = 0;
    for (int m = 0; m < 3; ++m) {
        R.g[m] = refine_geom(pl.g[m], lp[m], job.maxIter, sstride, job.sat_off[m], job.margin);
        R.cross[m] = ws.cross[m].as<double>();
        lds_refine = std::max(lds_refine, lp[m].lds_refine);
    }
    R.chunk = chunk;
    if (lds_refine > 64 * 1024)
        MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lag_refine), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_refine));
    float* ow = ws.outw.as<float>() + (size_t)p0 * wcap;
    int* oi = ws.outi.as<int>() + (size_t)p0 * 4;
    hipLaunchKernelGGL(k_lag_refine, dim3(np, 3), dim3(256), lds_refine, sa, R, sat_p, wcap, ow, oi, (float*)nullptr);
    MI_TRY(launch_check("k_lag_refine"));
    if (np == chunk && chunk == n) {  // the whole job at once: the three planes' results are one block on either side
        MI_HIP(hipMemcpyAsync(ws.pin_w.as<float>(), ws.outw.as<float>(), 4 * (size_t)3 * n * wcap, hipMemcpyDeviceToHost, sa));
        MI_HIP(hipMemcpyAsync(ws.pin_i.as<int>(), ws.outi.as<int>(), sizeof(int) * 4 * 3 * (size_t)n, hipMemcpyDeviceToHost, sa));
    } else {
        for (int m = 0; m < 3; ++m) {
            MI_HIP(hipMemcpyAsync(ws.pin_w.as<float>() + ((size_t)m * n + c0 + p0) * wcap, ow + (size_t)m * chunk * wcap, 4 * (size_t)np * wcap,
                                  hipMemcpyDeviceToHost, sa));
            MI_HIP(hipMemcpyAsync(ws.pin_i.as<int>() + ((size_t)m * n + c0 + p0) * 4, oi + (size_t)m * chunk * 4, sizeof(int) * 4 * np,
                                  hipMemcpyDeviceToHost, sa));
        }
    }
    return MI_OK;
}

// the end of the job: an event behind the refinement and its copies
static int close_job(LagJob& job) {
    LagWorkspace& ws = *job.ws;
    MI_HIP(hipEventRecord(ws.ev_done, ws.sa));  // (everything `sm` and `sb` were given lies before an event `sa` waited for)
    job.enqueued = true;
    return MI_OK;
}

// the chains of a job whose enqueue deferred them, behind `gate` (an event on the device's MIP stream: the batch records it
// after the LAST group's MIP pass, so that no chain kernel competes with a MIP pass for the memory system)
int ncc_lag_enqueue_chains(LagJob* job, hipEvent_t gate) {
    if (!job || !job->chains_pending) return MI_OK;
    MI_TRY(enqueue_chains(*job, 0, 0, job->n, 0, gate ? gate : job->ws->ev_mip[0]));
    job->chains_pending = false;
    return close_job(*job);
}

hipStream_t ncc_lag_mip_stream(LagJob* job) { return job ? job->ws->sm : nullptr; }

// Waits for the group's device stage, then the host rules.  careful[q] is set for pairs whose result was not taken here (see the
// header of this file); out[q] is then untouched.  Destroys the job.
int ncc_lag_finish(LagJob* job_in, mi_ncc_params* params, mi_ncc_descr* out, unsigned char* careful) {
    std::unique_ptr<LagJob> job(job_in);
    if (!job) return MI_OK;
    if (job->chains_pending) MI_TRY(ncc_lag_enqueue_chains(job.get(), nullptr));
    LagWorkspace& ws = *job->ws;
    const PairPlan& pl = job->pl;
    const int n = job->n, wcap = job->wcap;
    const float margin = job->margin;
    MI_TRY(ws.wait_done());

    // compute_Alignment (compute_funcs.cu:1597-1609) on the returned windows
    for (int q = 0; q < n; ++q) {
        const mi_ncc_params& Pq = params[q];
        int w1[3], w2[3], du[3], dv[3];
        float peak[3];
        float tight = FLT_MAX;  // smallest margin of any comparison the host rules made
        bool redo = false;
        for (int m = 0; m < 3; ++m) {
            const PlaneGeom& g = pl.g[m];
            const float* win = ws.pin_w.as<float>() + ((size_t)m * n + q) * wcap;
            const int* oi = ws.pin_i.as<int>() + ((size_t)m * n + q) * 4;
            du[m] = oi[0];
            dv[m] = oi[1];
            if (oi[3]) redo = true;
            const int rowlen = 2 * g.wv + 1, c = g.wu * rowlen + g.wv;
            peak[m] = win[c];
            if (oi[2]) { w1[m] = w2[m] = Pq.INF_W; continue; }
            w2[m] = peak_half_width(Pq, win, c, 1, g.wv, g.wv, &tight);
            w1[m] = peak_half_width(Pq, win, c, rowlen, g.wu, g.wv, &tight);
        }
        mi_ncc_descr r;
        combine_axis(Pq, &r, 0, du[0], peak[0], w1[0], du[1], peak[1], w1[1], &tight);  // V: xy rows, xz rows
        combine_axis(Pq, &r, 1, dv[0], peak[0], w2[0], du[2], peak[2], w1[2], &tight);  // H: xy cols, yz rows
        combine_axis(Pq, &r, 2, dv[1], peak[1], w2[1], dv[2], peak[2], w2[2], &tight);  // D: xz cols, yz cols
        if (job->side == MI_NORTH_SOUTH) r.coord[0] += job->ni; else r.coord[1] += job->nj;  // libcrossmips.cpp:483-486
        if (redo || tight < margin) { careful[q] = 1; continue; }
        careful[q] = 0;
        out[q] = r;
        ncc_count(0, 1);
    }
    return MI_OK;
}

void ncc_lag_abandon(LagJob* job) { delete job; }

int ncc_lag_group(int dev, hipStream_t s, int n, const float* const* a_ptrs, const float* const* b_ptrs, int dimk, int dimi, int dimj, int ni,
                  int nj, int delayk, int delayi, int delayj, int side, mi_ncc_params* params, mi_ncc_descr* out, unsigned char* careful, TileFmt fmt) {
    LagJob* job = nullptr;
    MI_TRY(ncc_lag_enqueue(dev, s, n, a_ptrs, b_ptrs, dimk, dimi, dimj, ni, nj, delayk, delayi, delayj, side, params, &job, false, fmt, 1, false));
    return ncc_lag_finish(job, params, out, careful);
}

// NCC map of one pair of MIPs through the lag transform (building block for the parity tests)
int ncc_lag_map(int dev, hipStream_t s, const float* mip1, const float* mip2, int dimu, int dimv, int delayu, int delayv, float* map) {
    PlaneGeom g{};
    g.dimu = dimu; g.dimv = dimv; g.delayu = delayu; g.delayv = delayv; g.wu = 0; g.wv = 0; g.sat = 0;
    g.tiled = (dimu / TILE) * (dimv / TILE) > 0;
    const LagPlane lp = plan_lag_plane(g, 0);
    MI_REQUIRE(lp.ok, "compute_NCC_map: extents do not fit the lag transform");
    std::unique_ptr<LagWorkspace> wsp = take_lag_ws(dev);
    if (!wsp) return fail(MI_ERR_NOMEM, "compute_NCC_map: out of host memory");
    LagWorkspace& ws = *wsp;
    struct Giver { std::unique_ptr<LagWorkspace>& p; ~Giver() { give_lag_ws(std::move(p)); } } giver{wsp};
    const size_t nt = (size_t)(dimu / TILE) * (dimv / TILE), npx = ((size_t)dimu * dimv + 3) / 4 * 4;
    const BandLayout L(g, lp);
    // the kernels address a pair's arrays as offsets into ONE block: mip1 | mip2 | tile sums 1 | tile sums 2
    g.mip1 = 0; g.mip2 = npx; g.ps1 = 2 * npx; g.ps2 = 2 * npx + nt;
    MI_TRY(grow(ws.fbuf, 4 * (2 * npx + 2 * nt + 4)));
    MI_TRY(grow(ws.sat, 8 * L.total));
    MI_TRY(grow(ws.outw, 4 * 4));
    MI_TRY(grow(ws.outi, sizeof(int) * 4));
    float* base = ws.fbuf.as<float>();
    MI_HIP(hipMemcpyAsync(base, mip1, 4 * (size_t)dimu * dimv, hipMemcpyDeviceToDevice, s));
    MI_HIP(hipMemcpyAsync(base + npx, mip2, 4 * (size_t)dimu * dimv, hipMemcpyDeviceToDevice, s));
    TabGeom G{};
    G.nplanes = 1; G.pstride = 0; G.sstride = L.total;
    G.p[0] = tab_plane(g, lp, 0);
    MI_TRY(prepare_tables(s, G, base, ws.sat.as<double>(), 1));
    MI_TRY(lag_cross(dev, s, lp, base, base + npx, 0, 1, ws));
    RefineAll R{};
    R.g[0] = refine_geom(g, lp, 0, L.total, 0, 0.0f);
    R.cross[0] = ws.cross[0].as<double>();
    R.chunk = 1;
    if (lp.lds_refine > 64 * 1024)
        MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lag_refine), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp.lds_refine));
    hipLaunchKernelGGL(k_lag_refine, dim3(1, 1), dim3(256), lp.lds_refine, s, R, ws.sat.as<double>(), 1, ws.outw.as<float>(), ws.outi.as<int>(), map);
    MI_TRY(launch_check("k_lag_refine"));
    MI_HIP(hipStreamSynchronize(s));
    return MI_OK;
}

// average duration of ONE k_mips launch over n pairs (HIP events on `s` around `reps` launches; bench.py's roofline hook)
int ncc_time_mips(int dev, hipStream_t s, int n, const float* const* a_ptrs, const float* const* b_ptrs, int dimk, int dimi, int dimj, int ni, int nj,
                  int side, int reps, float* ms, TileFmt fmt) {
    MI_REQUIRE(n > 0 && reps > 0 && ms && a_ptrs && b_ptrs, "mi_ncc_time_mips: invalid arguments");
    MI_REQUIRE(side == MI_NORTH_SOUTH || side == MI_WEST_EAST, "CrossMIPs: unexpected alignment configuration");
    MI_REQUIRE(fmt.bytes == 4 || mips_int_ok(fmt.bytes, dimk, dimj, (size_t)dimi * dimj),
               "mi_ncc_time_mips: integer tiles need rows of whole 32-bit words and at most %d slices", 4 * MIP_KPW);
    const int dimi_v = side == MI_NORTH_SOUTH ? dimi - ni : dimi, dimj_v = side == MI_WEST_EAST ? dimj - nj : dimj;
    MI_REQUIRE(dimi_v > 0 && dimj_v > 0 && dimk > 0, "mi_ncc_time_mips: empty view");
    MI_REQUIRE(fmt.bytes != 4 || dimk > 4 * MIP_KPW || mips5_ok(dimk, dimj), "mi_ncc_time_mips: rows of %d samples are too long for the MIP pass", dimj);
    const size_t xy = (size_t)dimi_v * dimj_v, xz = (size_t)dimi_v * dimk, yz = (size_t)dimj_v * dimk;
    const size_t pstride = (2 * (xy + xz + yz) + 3) / 4 * 4, tmpf = mips_tmp_floats(dimk, dimi_v, dimj_v);
    DevBuf out, tmp, tab;
    MI_TRY(out.alloc(4 * pstride * n));
    MI_TRY(tmp.alloc(4 * tmpf * n));
    MI_TRY(tab.alloc(sizeof(void*) * 2 * n));
    std::vector<const float*> h(2 * (size_t)n);
    for (int q = 0; q < n; ++q) { h[2 * q] = a_ptrs[q]; h[2 * q + 1] = b_ptrs[q]; }
    MI_HIP(hipMemcpyAsync(tab.p, h.data(), sizeof(void*) * 2 * n, hipMemcpyHostToDevice, s));
    MI_HIP(hipStreamSynchronize(s));
    float* o = out.as<float>();
    const int nb = mips_fmt_bands(fmt.bytes, dimk);
    const int bands = (dimi_v + MIP_ROWS * nb - 1) / (MIP_ROWS * nb), cblocks = (dimj_v + ((side == MI_WEST_EAST ? nj : 0) & 63) + 63) / 64;
    const size_t lds = sizeof(float) * MIP_ROWS * (size_t)dimk * nb;
    MI_REQUIRE(lds <= 32 * 1024, "mi_ncc_time_mips: stack too deep for the timed variant");
    float* xz_tmp = tmp.as<float>() + 2 * (size_t)n * bands * dimk * dimj_v;
    struct Events {  // (destroyed on every path out of this function)
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } evs;
    MI_HIP(hipEventCreate(&evs.a));
    MI_HIP(hipEventCreate(&evs.b));
    hipEvent_t e0 = evs.a, e1 = evs.b;
    const char* ke = MI_PROBE_ENV("MI_NCC_MIPS_KNOCK");  // (measurement aid, see k_mips)
    const int knock = ke ? std::atoi(ke) : 0;
    for (int r = -1; r < reps; ++r) {  // r = -1: warm-up
        if (r == 0) MI_HIP(hipEventRecord(e0, s));
        if (fmt.bytes != 4) {
            const int aj0 = side == MI_WEST_EAST ? nj : 0, wcol = mips_fmt_width(fmt.bytes);
            const int cb = (dimj_v + (aj0 & (wcol - 1)) + wcol - 1) / wcol;
            const dim3 grid((unsigned)((bands * 2 * n + 7) / 8 * 8 * cb));
            const unsigned char* const* t8 = tab.as<const unsigned char*>();
            float* o2 = o + xy + xz + yz;   // (the maxima merge into whatever the MIPs hold: the timed launches need no zeroing)
            if (fmt.bytes == 2)
                hipLaunchKernelGGL(k_mips_int<2>, grid, dim3(256), 0, s, (const unsigned char*)nullptr, (const unsigned char*)nullptr, t8, pstride, dimk,
                                   dimi_v, dimj_v, (size_t)dimi * dimj, dimj, side == MI_NORTH_SOUTH ? ni : 0, aj0, fmt.scale, o, o + xy, o + xy + xz, o2,
                                   o2 + xy, o2 + xy + xz, (float*)nullptr, (size_t)0, cb, bands, 2 * n);
            else
                hipLaunchKernelGGL(k_mips_int<1>, grid, dim3(256), 0, s, (const unsigned char*)nullptr, (const unsigned char*)nullptr, t8, pstride, dimk,
                                   dimi_v, dimj_v, (size_t)dimi * dimj, dimj, side == MI_NORTH_SOUTH ? ni : 0, aj0, fmt.scale, o, o + xy, o + xy + xz, o2,
                                   o2 + xy, o2 + xy + xz, (float*)nullptr, (size_t)0, cb, bands, 2 * n);
            continue;
        }
        if (mips5_ok(dimk, dimj)) {  // (its maxima merge into whatever the MIPs hold: the timed launches need no zeroing)
            const char* we = MI_PROBE_ENV("MI_NCC_MIPS_WPE");
            const char* nr = MI_PROBE_ENV("MI_NCC_MIPS_NOREMAP");
            const int row_groups = bands * 2 * n, remap = nr && std::atoi(nr) != 0 ? 0 : 1;
            hipLaunchKernelGGL(HIP_KERNEL_NAME(we && std::atoi(we) == 3 ? k_mips5<3> : k_mips5<4>), dim3((unsigned)((row_groups + 7) / 8 * 8 * cblocks)), dim3(256), 0, s,
                               (const float*)nullptr, (const float*)nullptr, tab.as<const float*>(), pstride, dimk, dimi_v, dimj_v, (size_t)dimi * dimj, dimj,
                               side == MI_NORTH_SOUTH ? ni : 0, side == MI_WEST_EAST ? nj : 0, o, o + xy, o + xy + xz, o + xy + xz + yz, o + 2 * xy + xz + yz,
                               o + 2 * xy + 2 * xz + yz, knock, (float*)nullptr, (size_t)0, cblocks, bands, 2 * n, remap);
            continue;
        }
        hipLaunchKernelGGL(k_mips, dim3(cblocks, bands, 2 * n), dim3(256), lds, s, (const float*)nullptr, (const float*)nullptr, tab.as<const float*>(),
                           pstride, dimk, dimi_v, dimj_v, (size_t)dimi * dimj, dimj, side == MI_NORTH_SOUTH ? ni : 0, side == MI_WEST_EAST ? nj : 0, o,
                           o + xy, o + xy + xz, o + xy + xz + yz, o + 2 * xy + xz + yz, o + 2 * xy + 2 * xz + yz, tmp.as<float>(), xz_tmp, knock, (float*)nullptr,
                           (size_t)0);
    }
    MI_HIP(hipEventRecord(e1, s));
    MI_HIP(hipEventSynchronize(e1));
    float total = 0.0f;
    MI_HIP(hipEventElapsedTime(&total, e0, e1));
    *ms = total / (float)reps;
    return launch_check("k_mips");
}

void ncc_lag_drop_cached(int dev) {
    std::vector<std::unique_ptr<LagWorkspace>> drop;
    std::lock_guard<std::mutex> lock(g_lag_mu);
    for (size_t i = 0; i < g_lag_ws.size();)
        if (dev < 0 || g_lag_ws[i]->dev == dev) { drop.push_back(std::move(g_lag_ws[i])); g_lag_ws.erase(g_lag_ws.begin() + i); }
        else ++i;
}

}  // namespace mi
