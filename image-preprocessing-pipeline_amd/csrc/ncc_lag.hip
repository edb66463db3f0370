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

#include "ncc_core.h"
#include "ncc_lag.h"

namespace {

typedef double2 cplx;

__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }

// Transform length N = 2^a * 3^b (b <= 2): the radices of the decimation-in-frequency stages, in order (4s, then a 2, then 3s).
// 2048-row MIPs with 75 lags need N >= 2123: 2304 = 4^4 * 9 instead of 4096 cuts every lag kernel by 44 %.
struct FftPlan {
    int N, nstages;
    int radix[14];
    int twoff[14];  // start of stage s in the per-stage twiddle table (see fft_tables)
};

// In-place decimation-in-frequency transform of x[0 .. N) in LDS.  Natural order in, digit-reversed order out: with position
// digits d_1 d_2 ... (most significant first, radices r_1 r_2 ...) the frequency is k = d_1 + r_1 (d_2 + r_2 (...)).
// tw: per-stage tables -- stage s (radix r, q = L / r butterflies per group) holds exp(-2 pi i m t (N / L) / N) at
// twoff[s] + (m - 1) q + t, m = 1 .. r - 1: the look-ups of consecutive lanes are consecutive entries (from the one table
// exp(-2 pi i n / N) the middle stages touched up to 64 cache lines per load instruction).
__device__ __forceinline__ int pos_of_freq(int k, const FftPlan& pl) {
    int p = 0, rem = pl.N;
    for (int s = 0; s < pl.nstages; ++s) {
        const int r = pl.radix[s];
        rem /= r;
        p += (k % r) * rem;
        k /= r;
    }
    return p;
}

// BF = butterflies per thread and step: their LDS reads and table loads are issued before any store
template <int BF>
__device__ __forceinline__ void fft_dif(cplx* __restrict__ x, const FftPlan& pl, const cplx* __restrict__ tw) {
    const int N = pl.N;
    int L = N;
    for (int st = 0; st < pl.nstages; ++st) {
        const int r = pl.radix[st], q = L / r, nb = N / r;
        const cplx* stw = tw + pl.twoff[st];
        for (int i0 = threadIdx.x; i0 < nb; i0 += BF * blockDim.x) {
            cplx a[BF][4], w[BF][3];
            int off[BF];  // (offsets, not pointers: a pointer that may be null is no longer known to point into LDS -> FLAT accesses)
            bool live[BF];
#pragma unroll
            for (int u = 0; u < BF; ++u) {
                const int idx = i0 + u * blockDim.x;
                live[u] = idx < nb;
                const int g = idx / q, t = idx - g * q;
                off[u] = g * L + t;
                if (live[u]) {
                    a[u][0] = x[off[u]];
                    a[u][1] = x[off[u] + q];
                    if (r > 2) a[u][2] = x[off[u] + 2 * q];
                    if (r > 3) a[u][3] = x[off[u] + 3 * q];
                    w[u][0] = stw[t];
                    if (r > 2) w[u][1] = stw[q + t];
                    if (r > 3) w[u][2] = stw[2 * q + t];
                }
            }
#pragma unroll
            for (int u = 0; u < BF; ++u) {
                if (!live[u]) continue;
                cplx* pu = x + off[u];
                if (r == 4) {
                    const cplx s02 = cadd(a[u][0], a[u][2]), d02 = csub(a[u][0], a[u][2]), s13 = cadd(a[u][1], a[u][3]), d13 = csub(a[u][1], a[u][3]);
                    const cplx y1 = make_double2(d02.x + d13.y, d02.y - d13.x);  // d02 - i d13
                    const cplx y3 = make_double2(d02.x - d13.y, d02.y + d13.x);  // d02 + i d13
                    pu[0] = cadd(s02, s13);
                    pu[q] = cmul(y1, w[u][0]);
                    pu[2 * q] = cmul(csub(s02, s13), w[u][1]);
                    pu[3 * q] = cmul(y3, w[u][2]);
                } else if (r == 3) {
                    const cplx t1 = cadd(a[u][1], a[u][2]), dd = csub(a[u][1], a[u][2]);
                    const cplx t2 = make_double2(a[u][0].x - 0.5 * t1.x, a[u][0].y - 0.5 * t1.y);
                    const double h = 0.86602540378443864676;  // sqrt(3) / 2
                    const cplx sv = make_double2(h * dd.x, h * dd.y);
                    pu[0] = cadd(a[u][0], t1);
                    pu[q] = cmul(make_double2(t2.x + sv.y, t2.y - sv.x), w[u][0]);      // t2 - i sv
                    pu[2 * q] = cmul(make_double2(t2.x - sv.y, t2.y + sv.x), w[u][1]);  // t2 + i sv
                } else {
                    pu[0] = cadd(a[u][0], a[u][1]);
                    pu[q] = cmul(csub(a[u][0], a[u][1]), w[u][0]);
                }
            }
        }
        __syncthreads();
        L = q;
    }
}

// forward lag transform of short-axis line j (blockIdx.x) of pair blockIdx.y: element i of the line = m[i * ls + j * ss]
template <int TH, int BF>
__global__ __launch_bounds__(TH) void k_lag_fwd(const float* __restrict__ m1, const float* __restrict__ m2, size_t pstride, int n_long, int n_short,
                                                 int ls, int ss, int nkp, FftPlan pl, const cplx* __restrict__ tw, const int* __restrict__ slot_pos,
                                                 const int* __restrict__ slot_neg, cplx* __restrict__ SF, cplx* __restrict__ ST) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lag_lds[];
    cplx* x = reinterpret_cast<cplx*>(lag_lds);
    // gridDim.x = 8 * ceil(n_short / 8): work-groups b, b + 8, ... run on one XCD and take NEIGHBOURING lines j -- with the long
    // axis strided in memory (west-east pairs) 32 neighbouring lines share every 128-byte line they read, and spread over the
    // eight L2s each of them fetched it again (0.81 GB fetched for 0.28 GB of MIPs)
    const int N = pl.N, NK = N / 2 + 1, j = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (j >= n_short) return;
    const float* a = m1 + (size_t)blockIdx.y * pstride + (size_t)j * ss;
    const float* b = m2 + (size_t)blockIdx.y * pstride + (size_t)j * ss;
    for (int i = threadIdx.x; i < N; i += blockDim.x)
        x[i] = i < n_long ? make_double2((double)a[(size_t)i * ls], (double)b[(size_t)i * ls]) : make_double2(0.0, 0.0);
    __syncthreads();
    fft_dif<BF>(x, pl, tw);
    cplx* of = SF + ((size_t)blockIdx.y * n_short + j) * nkp;  // (rows of nkp >= NK spectra: whole 128-byte lines per eight)
    cplx* ot = ST + ((size_t)blockIdx.y * n_short + j) * nkp;
    // The N/2 + 1 frequencies 0 .. N/2 of a real signal's spectrum are kept in POSITION order ("slots": slot_pos ascending; slot_neg =
    // position of the mirror frequency N - k): the untangling pass then reads LDS in nearly contiguous order -- in frequency
    // order consecutive lanes sit N/4 elements apart, a 16-way bank conflict -- and the per-frequency correlation does not care
    // about the order of its frequencies.
    for (int sl = threadIdx.x; sl < NK; sl += blockDim.x) {
        const cplx zk = x[slot_pos[sl]], zn = x[slot_neg[sl]];
        of[sl] = make_double2(0.5 * (zk.x + zn.x), 0.5 * (zk.y - zn.y));   // (Z[k] + conj Z[N-k]) / 2
        ot[sl] = make_double2(0.5 * (zk.y + zn.y), -0.5 * (zk.x - zn.x));  // (Z[k] - conj Z[N-k]) / (2 i)
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
__global__ __launch_bounds__(256) void k_lag_mac(const cplx* __restrict__ SF, const cplx* __restrict__ ST, int n_short, int NK, int Es, int KT, int JP,
                                                 int FW, int TW, int nlp, int nkp, int tiles_per_pair, int ntiles, cplx* __restrict__ CH) {
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

    // Slot b of the persistent sequence (blockIdx.x, + gridDim.x, ...; both multiples of 16) -> tile: the two tiles that share the
    // 128-byte lines of the spectra (k0 = 8 m and 8 m + 4; tiles_per_pair is even) go to work-groups b and b + 8, which run on ONE
    // XCD at about the same time -- taken in order they sat on different XCDs and every L2 fetched whole lines for half of
    // each (1.75 GB fetched for 0.63 GB of spectra).  Tiles past the end and the padding tile of a pair are empty (nk <= 0).
    auto decode = [&](int b, int& pair, int& k0, int& nk) {
        const int tile = (b & ~15) | ((b & 7) << 1) | ((b >> 3) & 1);
        pair = tile / tiles_per_pair;
        k0 = (tile - pair * tiles_per_pair) * KT;
        nk = tile < ntiles ? min(KT, NK - k0) : 0;
    };
    const int nslots = (ntiles + 15) & ~15;
    int slot = blockIdx.x;
    __syncthreads();
    if (slot < nslots) {  // the first tile goes to LDS directly
        int pair, k0, nk;
        decode(slot, pair, k0, nk);
        const size_t pbase = (size_t)pair * n_short;
        for (int e = threadIdx.x; e < nelem; e += 256) {
            const int x = e / KT, l = e - x * KT;
            if (l < nk) {
                Fs[(size_t)l * FW + PAD + x] = SF[(pbase + x) * nkp + k0 + l];
                Ts[(size_t)l * TW + x] = ST[(pbase + x) * nkp + k0 + l];
            }
        }
    }
    while (slot < nslots) {
        int pair, k0, nk;
        decode(slot, pair, k0, nk);
        __syncthreads();  // the tile is in LDS
        const int next = slot + gridDim.x;
        const bool has_next = next < nslots;
        int npair = 0, nk0 = 0, nnk = 0;
        if (has_next) decode(next, npair, nk0, nnk);
        cplx pf[PF > 0 ? PF : 1], pt[PF > 0 ? PF : 1];  // (PF = 0: rows too long for the registers -- the next tile is fetched behind this one)
        if (PF > 0 && has_next) {
            const size_t pbase = (size_t)npair * n_short;
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int e = threadIdx.x + i * 256;
                const int x = e / KT, l = e - x * KT;
                const bool ok = e < nelem && l < nnk;
                pf[i] = ok ? SF[(pbase + x) * nkp + nk0 + l] : make_double2(0.0, 0.0);
                pt[i] = ok ? ST[(pbase + x) * nkp + nk0 + l] : make_double2(0.0, 0.0);
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
            const size_t pbase = (size_t)npair * n_short;
            for (int e = threadIdx.x; e < nelem; e += 256) {
                const int x = e / KT, l = e - x * KT;
                if (l < nnk) {
                    Fs[(size_t)l * FW + PAD + x] = SF[(pbase + x) * nkp + nk0 + l];
                    Ts[(size_t)l * TW + x] = ST[(pbase + x) * nkp + nk0 + l];
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

// inverse lag transform of two short-axis lags (2 * blockIdx.x, + 1) of pair blockIdx.y: c_a[n] + i c_b[n] = FFT(conj C_a + i conj C_b) / N
// (both c real); the long-axis lags [-El, El] go to cross[(u + Eu) * (2 Ev + 1) + (v + Ev)]
template <int TH, int BF>
__global__ __launch_bounds__(TH) void k_lag_inv(const cplx* __restrict__ CH, int NK, int nlp, FftPlan pl, int Es, int El, int long_is_u, int Eu, int Ev,
                                                 const cplx* __restrict__ tw, const int* __restrict__ slot_freq, double* __restrict__ cross) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lag_lds[];
    cplx* x = reinterpret_cast<cplx*>(lag_lds);
    const int N = pl.N, nlag = 2 * Es + 1;
    const int sa = 2 * blockIdx.x, sb = sa + 1;
    const bool has_b = sb < nlag;
    const cplx* src = CH + (size_t)blockIdx.y * NK * nlp;
    for (int sl = threadIdx.x; sl < NK; sl += blockDim.x) {
        const int k = slot_freq[sl];
        const cplx xa = src[(size_t)sl * nlp + sa];
        const cplx xb = has_b ? src[(size_t)sl * nlp + sb] : make_double2(0.0, 0.0);
        x[k] = make_double2(xa.x + xb.y, -xa.y + xb.x);                              // conj(Xa) + i conj(Xb)
        if (k > 0 && k < N / 2) x[N - k] = make_double2(xa.x - xb.y, xa.y + xb.x);   // Xa + i Xb  (= conj X[N-k] terms)
    }
    __syncthreads();
    fft_dif<BF>(x, pl, tw);
    const double inv = 1.0 / (double)N;
    const int W = 2 * Ev + 1;
    double* out = cross + (size_t)blockIdx.y * (2 * Eu + 1) * W;
    for (int e = threadIdx.x; e < 2 * El + 1; e += blockDim.x) {
        const int l = e - El;
        const cplx v = x[pos_of_freq((l + N) % N, pl)];
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
constexpr int BAND_CH = 128;  // rows per chunk of the column sums

// chunk column sums of (f - c0) and (f - c0)^2: T[((2 z + {0,1}) * nch + chunk) * n_short + b], z = blockIdx.z = 2 * pair + MIP.
// LONG_CONTIG = false: rows a are n_short apart in memory, a wave owns 64 neighbouring columns b (coalesced rows);
// LONG_CONTIG = true : the long axis is the contiguous one, a wave owns ONE b and strides along a.
template <bool LONG_CONTIG>
__global__ __launch_bounds__(64) void k_band_chunksum(const float* __restrict__ m1, const float* __restrict__ m2, size_t pstride, size_t sstride,
                                                      const double* __restrict__ c0a, int n_long, int n_short, int ls, int ss, int nch,
                                                      double* __restrict__ T) {
    const int z = blockIdx.z, pair = z >> 1, which = z & 1, ca = blockIdx.y;
    const float* m = (which ? m2 : m1) + (size_t)pair * pstride;
    const double cm = c0a[(size_t)pair * sstride + which];
    double* Tp = T + (size_t)pair * sstride + ((size_t)(2 * which) * nch + ca) * n_short;
    double* Tq = Tp + (size_t)nch * n_short;
    const int a0 = ca * BAND_CH, a1 = min(n_long, a0 + BAND_CH);
    double p = 0.0, q = 0.0;
    if (!LONG_CONTIG) {
        const int b = blockIdx.x * 64 + threadIdx.x;
        if (b >= n_short) return;
        const float* col = m + (size_t)b * ss;
#pragma unroll 4
        for (int a = a0; a < a1; ++a) {
            const double g = (double)col[(size_t)a * ls] - cm;
            p += g;
            q += g * g;
        }
        Tp[b] = p;
        Tq[b] = q;
    } else {
        const int b = blockIdx.x;
        const float* row = m + (size_t)b * ss;
        for (int a = a0 + threadIdx.x; a < a1; a += 64) {
            const double g = (double)row[(size_t)a * ls] - cm;
            p += g;
            q += g * g;
        }
        for (int off = 32; off > 0; off >>= 1) {
            p += __shfl_down(p, off, 64);
            q += __shfl_down(q, off, 64);
        }
        if (threadIdx.x == 0) { Tp[b] = p; Tq[b] = q; }
    }
}

// column running sums at the band rows: tab[row(a)][b + 1] = sum_{a' < a} g[a'][b]  (P and Q of one MIP); blockIdx.y = band
// (0: rows [0, B], 1: rows [n_long - B, n_long]), a lane owns one column b
__global__ __launch_bounds__(64) void k_band_cols(const float* __restrict__ m1, const float* __restrict__ m2, size_t pstride, size_t sstride,
                                                  const double* __restrict__ c0a, int n_long, int n_short, int ls, int ss, int nch, int B, size_t tab,
                                                  const double* __restrict__ T, double* __restrict__ P1) {
    const int z = blockIdx.z, pair = z >> 1, which = z & 1, band = blockIdx.y;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= n_short) return;
    const bool full = B >= n_long;
    if (full && band == 1) return;
    const float* col = (which ? m2 : m1) + (size_t)pair * pstride + (size_t)b * ss;
    const double cm = c0a[(size_t)pair * sstride + which];
    double* P = P1 + (size_t)pair * sstride + (size_t)(2 * which) * tab;  // P1 | Q1 | P2 | Q2
    double* Q = P + tab;
    const int w1 = n_short + 1;
    auto rowof = [&](int a) { return (full || a <= B) ? a : a - (n_long - B) + B + 1; };
    double p = 0.0, q = 0.0;
    int a = 0;
    if (band == 1) {  // everything below the band: whole chunks from their sums, the rest row by row
        const int a_start = n_long - B, cs = a_start / BAND_CH;
        const double* Tp = T + (size_t)pair * sstride + (size_t)(2 * which) * nch * n_short;
        const double* Tq = Tp + (size_t)nch * n_short;
        for (int c = 0; c < cs; ++c) { p += Tp[(size_t)c * n_short + b]; q += Tq[(size_t)c * n_short + b]; }
        for (a = cs * BAND_CH; a < a_start; ++a) {
            const double g = (double)col[(size_t)a * ls] - cm;
            p += g;
            q += g * g;
        }
    }
    const int a_end = band == 0 ? min(n_long, full ? n_long : B) : n_long;
    P[(size_t)rowof(a) * w1 + b + 1] = p;
    Q[(size_t)rowof(a) * w1 + b + 1] = q;
    for (; a < a_end; ++a) {
        const double g = (double)col[(size_t)a * ls] - cm;
        p += g;
        q += g * g;
        P[(size_t)rowof(a + 1) * w1 + b + 1] = p;
        Q[(size_t)rowof(a + 1) * w1 + b + 1] = q;
    }
}

// prefix along b of every kept row, in place; column 0 = 0.  One wave per (row, table); blockIdx.y = table (P1, Q1, P2, Q2)
__global__ __launch_bounds__(64) void k_band_rows(size_t sstride, int n_short, size_t tab, double* __restrict__ P1) {
    double* S = P1 + (size_t)blockIdx.z * sstride + (size_t)blockIdx.y * tab + (size_t)blockIdx.x * (n_short + 1);
    const int lane = threadIdx.x;
    double carry = 0.0;
    if (lane == 0) S[0] = 0.0;
    for (int b0 = 0; b0 < n_short; b0 += 64) {
        const int b = b0 + lane;
        const double v = b < n_short ? S[b + 1] : 0.0;
        const double sc = wave_inclusive_scan(v) + carry;
        if (b < n_short) S[b + 1] = sc;
        carry = __shfl(sc, 63, 64);
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
__global__ __launch_bounds__(256) void k_lag_refine(RefineGeom g, const double* __restrict__ sat_base, const double* __restrict__ cross, int wcap,
                                                    float* __restrict__ out_win, int* __restrict__ out_int, float* __restrict__ out_map) {
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
    FftPlan fft;
    int KT, JP, FW, TW, LB, nlp, fft_threads;
    size_t lds_fft, lds_mac, lds_refine;
    bool ok;
};

// smallest N = 2^a * {1, 3, 9} >= need (a >= 2), as radix-4 stages, then a radix-2 stage, then the radix-3 stages
FftPlan make_fft_plan(int need) {
    long best = 0;
    int best_a = 0, best_b = 0;
    for (int b = 0; b <= 2; ++b) {
        long n = b == 0 ? 1 : (b == 1 ? 3 : 9);
        int a = 0;
        while (a < 2 || n < need) { n *= 2; ++a; }
        if (best == 0 || n < best) { best = n; best_a = a; best_b = b; }
    }
    FftPlan pl{};
    pl.N = (int)best;
    int a = best_a;
    while (a >= 2) { pl.radix[pl.nstages++] = 4; a -= 2; }
    if (a == 1) pl.radix[pl.nstages++] = 2;
    for (int b = 0; b < best_b; ++b) pl.radix[pl.nstages++] = 3;
    int off = 0, L = pl.N;
    for (int st = 0; st < pl.nstages; ++st) {
        pl.twoff[st] = off;
        L /= pl.radix[st];           // q of the stage
        off += (pl.radix[st] - 1) * L;
    }
    return pl;
}

int host_pos_of_freq(int k, const FftPlan& pl) {
    int p = 0, rem = pl.N;
    for (int s = 0; s < pl.nstages; ++s) {
        const int r = pl.radix[s];
        rem /= r;
        p += (k % r) * rem;
        k /= r;
    }
    return p;
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
    p.fft = make_fft_plan(p.n_long + p.El);
    p.lds_fft = sizeof(double) * 2 * (size_t)p.fft.N;
    const int nlag = 2 * p.Es + 1;
    p.LB = 4;  // lags per thread of the correlation kernel (8 was measured: twice the registers, bank conflicts, no faster)
    p.nlp = (nlag + p.LB - 1) / p.LB * p.LB;
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
    p.fft_threads = p.fft.N >= 2048 ? 512 : 256;
    p.lds_mac = row * p.KT + sizeof(double) * 2 * p.LB * (size_t)(p.JP - 1) * nvb * p.KT;  // (xy plane of config 5: 52 KB, three per CU)
    const int Hm = 2 * g.delayu + 1, Wm = 2 * g.delayv + 1, H = 2 * g.wu + 1, W = 2 * g.wv + 1;
    p.lds_refine = sizeof(float) * ((size_t)Hm * Wm + 2 * (size_t)H * W);
    p.ok = p.fft.N <= 8192 && p.lds_mac <= 150 * 1024 && nvb <= 256 && p.lds_refine <= 120 * 1024;
    return p;
}

// Per (device, N), kept for the life of the process: exp(-2 pi i n / N) in fp64, and the slot tables of the half spectrum
// (position of slot s in ascending order, position of its mirror frequency, its frequency).
struct FftTables {
    const cplx* tw;
    const int *slot_pos, *slot_neg, *slot_freq;
};
struct TwiddleKey { int dev, n; bool operator<(const TwiddleKey& o) const { return dev != o.dev ? dev < o.dev : n < o.n; } };
std::mutex& g_tw_mu = *new std::mutex;
std::map<TwiddleKey, FftTables>& g_tw = *new std::map<TwiddleKey, FftTables>;

int fft_tables(int dev, const FftPlan& pl, hipStream_t s, FftTables* out) {
    std::lock_guard<std::mutex> lock(g_tw_mu);
    auto it = g_tw.find(TwiddleKey{dev, pl.N});
    if (it != g_tw.end()) { *out = it->second; return MI_OK; }
    const size_t N = (size_t)pl.N, NK = N / 2 + 1;
    std::vector<double> h(2 * N);
    const long double step = 2.0L * 3.14159265358979323846264338327950288L / (long double)N;
    for (size_t n = 0; n < N; ++n) {
        h[2 * n] = (double)cosl(step * (long double)n);
        h[2 * n + 1] = (double)-sinl(step * (long double)n);
    }
    if (N % 4 == 0) {  // exact at the quarter points
        h[2 * (N / 4)] = 0.0; h[2 * (N / 4) + 1] = -1.0;
        h[2 * (N / 2)] = -1.0; h[2 * (N / 2) + 1] = 0.0;
        h[2 * (3 * N / 4)] = 0.0; h[2 * (3 * N / 4) + 1] = 1.0;
    }
    std::vector<std::pair<int, int>> order(NK);  // (position, frequency)
    for (size_t k = 0; k < NK; ++k) order[k] = {host_pos_of_freq((int)k, pl), (int)k};
    std::sort(order.begin(), order.end());
    std::vector<int> tabs(3 * NK);
    for (size_t sl = 0; sl < NK; ++sl) {
        tabs[sl] = order[sl].first;
        tabs[NK + sl] = host_pos_of_freq((int)((N - (size_t)order[sl].second) % N), pl);
        tabs[2 * NK + sl] = order[sl].second;
    }
    // per-stage tables: copies of the entries above, so every twiddle keeps its value
    std::vector<double> hs;
    {
        size_t L = N;
        for (int st = 0; st < pl.nstages; ++st) {
            const size_t r = (size_t)pl.radix[st], q = L / r, tstep = N / L;
            for (size_t m = 1; m < r; ++m)
                for (size_t t = 0; t < q; ++t) {
                    const size_t n = m * t * tstep;  // < N
                    hs.push_back(h[2 * n]);
                    hs.push_back(h[2 * n + 1]);
                }
            L = q;
        }
    }
    void *d = nullptr, *di = nullptr;
    MI_HIP(hipMalloc(&d, sizeof(double) * hs.size()));
    MI_HIP(hipMalloc(&di, sizeof(int) * 3 * NK));
    MI_HIP(hipMemcpyAsync(d, hs.data(), sizeof(double) * hs.size(), hipMemcpyHostToDevice, s));
    MI_HIP(hipMemcpyAsync(di, tabs.data(), sizeof(int) * 3 * NK, hipMemcpyHostToDevice, s));
    MI_HIP(hipStreamSynchronize(s));
    const int* ti = static_cast<const int*>(di);
    FftTables t{static_cast<const cplx*>(d), ti, ti + NK, ti + 2 * NK};
    g_tw[TwiddleKey{dev, pl.N}] = t;
    *out = t;
    return MI_OK;
}

// device + pinned buffers of the batched pipeline, kept between calls (one set per concurrent caller and device)
struct LagWorkspace {
    int dev = -1;
    DevBuf fbuf, sat, mip_tmp, SF[3], ST[3], CH[3], cross[3], outw, outi, tab;  // (lag-transform scratch per plane: the planes' chains run side by side)
    PinnedBuf pin_tab, pin_w, pin_i;
    // Two streams PER DEVICE, shared by every group in flight: the MIP pass (k_mips: one HBM-bound streaming read of both overlap
    // views) of piece i + 1 runs on `sm` while the table / lag-transform / refinement chain (fp64 and LDS work on a few MB) of
    // piece i runs on `sl`.  (A pair of streams per workspace mapped onto the same hardware queues in a way that put one
    // group's MIP pass behind the other group's chain: 1.3 of 4.8 ms overlapped, profiles/r03_ncc_timeline.txt.)
    hipStream_t sm = nullptr, sl[3] = {nullptr, nullptr, nullptr};  // (sl[m]: the chain of plane m; MI_NCC_CHAIN_STREAMS=1: one for all)
    hipStream_t sx = nullptr;  // MI_NCC_SPLIT_XY=1: the lag transform of the xy plane beside that plane's tables (default: behind them;
                               // measured equal -- the runtime maps the extra stream onto the hardware queue of the tables)
    hipEvent_t ev_start = nullptr, ev_lag = nullptr, ev_done = nullptr, ev_plane[2] = {nullptr, nullptr}, ev_x = nullptr;
    hipEvent_t ev_head_tab = nullptr;
    hipEvent_t ev_head = nullptr;  // per DEVICE like the streams (not owned): "the memory-bound head of the latest xy chain has run"
    std::vector<hipEvent_t> ev_mip, ev_mip_xy;  // per piece: all six MIPs of its pairs final / the xy MIPs final
    // The end of a job's device stage: ev_done on the xy plane's stream and one event on each of the other two.  The HOST waits for the
    // three; the xy stream does not wait for the others -- the streams belong to the device, and the next group's xy chain sat
    // behind that wait until this group's xz and yz chains had finished (0.35 ms of a 112-pair call: profiles/r03_ncc_timeline.txt).
    int wait_done() {
        MI_HIP(hipEventSynchronize(ev_done));
        for (int m = 1; m < 3; ++m)
            if (sl[m] != sl[0]) MI_HIP(hipEventSynchronize(ev_plane[m - 1]));
        return MI_OK;
    }
    ~LagWorkspace() {
        if (ev_start) (void)hipEventDestroy(ev_start);
        if (ev_lag) (void)hipEventDestroy(ev_lag);
        if (ev_done) (void)hipEventDestroy(ev_done);
        if (ev_x) (void)hipEventDestroy(ev_x);
        for (hipEvent_t e : ev_plane)
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_mip) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_mip_xy) (void)hipEventDestroy(e);
    }
    int streams(size_t pieces) {
        {
            static std::mutex mu;
            struct Four { hipStream_t s[5]; hipEvent_t head, head_tab; };
            static std::map<int, Four> per_dev;  // (never destroyed: the process' lifetime)
            std::lock_guard<std::mutex> lock(mu);
            auto it = per_dev.find(dev);
            if (it == per_dev.end()) {
                Four f{};
                int chains = 3;
                if (const char* e = std::getenv("MI_NCC_CHAIN_STREAMS")) chains = std::max(1, std::min(3, std::atoi(e)));
                for (int i = 0; i < 1 + chains; ++i) MI_HIP(hipStreamCreateWithFlags(&f.s[i], hipStreamNonBlocking));
                for (int i = 1 + chains; i < 4; ++i) f.s[i] = f.s[chains];
                f.s[4] = f.s[1];
                const char* sp = std::getenv("MI_NCC_SPLIT_XY");
                if (chains >= 2 && sp && std::atoi(sp) != 0) MI_HIP(hipStreamCreateWithFlags(&f.s[4], hipStreamNonBlocking));
                MI_HIP(hipEventCreateWithFlags(&f.head, hipEventDisableTiming));
                MI_HIP(hipEventCreateWithFlags(&f.head_tab, hipEventDisableTiming));
                it = per_dev.emplace(dev, f).first;
            }
            sm = it->second.s[0];
            for (int m = 0; m < 3; ++m) sl[m] = it->second.s[1 + m];
            sx = it->second.s[4];
            ev_head = it->second.head;
            ev_head_tab = it->second.head_tab;
        }
        if (!ev_x) MI_HIP(hipEventCreateWithFlags(&ev_x, hipEventDisableTiming));
        for (hipEvent_t& e : ev_plane)
            if (!e) MI_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
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

// The work-group shapes of the two transform kernels: threads and butterflies per thread and step.
using FwdFn = void (*)(const float*, const float*, size_t, int, int, int, int, int, FftPlan, const cplx*, const int*, const int*, cplx*, cplx*);
using InvFn = void (*)(const cplx*, int, int, FftPlan, int, int, int, int, int, const cplx*, const int*, double*);
struct FftShape {
    int threads, bf;
    FwdFn fwd_fn;
    InvFn inv_fn;
    const void *fwd, *inv;
};
template <int TH, int BF>
FftShape make_shape() {
    FftShape s{TH, BF, k_lag_fwd<TH, BF>, k_lag_inv<TH, BF>, nullptr, nullptr};
    s.fwd = reinterpret_cast<const void*>(s.fwd_fn);
    s.inv = reinterpret_cast<const void*>(s.inv_fn);
    return s;
}
const FftShape& fft_shape(const LagPlane& lp) {
    static const FftShape shapes[] = {make_shape<512, 2>(), make_shape<512, 1>(), make_shape<256, 2>(), make_shape<256, 1>(), make_shape<256, 3>(),
                                      make_shape<192, 3>(), make_shape<384, 3>(), make_shape<128, 3>(), make_shape<128, 5>()};
    static const int forced = [] {  // MI_NCC_FFT_SHAPE=<threads>,<butterflies>: measurement aid
        const char* e = std::getenv("MI_NCC_FFT_SHAPE");
        int t = 0, b = 0;
        if (!e || sscanf(e, "%d,%d", &t, &b) != 2) return -1;
        for (size_t i = 0; i < sizeof(shapes) / sizeof(shapes[0]); ++i)
            if (shapes[i].threads == t && shapes[i].bf == b) return (int)i;
        return -1;
    }();
    if (forced >= 0) return shapes[forced];
    // one butterfly per thread and step keeps the kernels under 64 registers: four work-groups of 512 per CU instead of two
    // (forward transform of the xy plane 765 -> 500 us: profiles/r03_lag_shapes.txt)
    return lp.fft_threads == 512 ? shapes[1] : shapes[3];
}

// cross terms of `np` pairs of one plane through the lag transform (MIPs at m1 / m2 + q * pstride) into ws.cross
int lag_cross(int dev, hipStream_t s, const LagPlane& lp, const float* m1, const float* m2, size_t pstride, int np, LagWorkspace& ws, int m = 0,
              hipEvent_t after_fwd = nullptr) {
    const int N = lp.fft.N, NK = N / 2 + 1, nlag = 2 * lp.Es + 1, nlp = lp.nlp, NKP = (NK + 7) & ~7;
    FftTables ft;
    MI_TRY(fft_tables(dev, lp.fft, s, &ft));
    const cplx* tw = ft.tw;
    MI_TRY(grow(ws.SF[m], sizeof(double) * 2 * (size_t)np * lp.n_short * NKP));
    MI_TRY(grow(ws.ST[m], sizeof(double) * 2 * (size_t)np * lp.n_short * NKP));
    MI_TRY(grow(ws.CH[m], sizeof(double) * 2 * (size_t)np * NK * nlp));
    MI_TRY(grow(ws.cross[m], sizeof(double) * (size_t)np * (2 * lp.Eu + 1) * (2 * lp.Ev + 1)));
    const FftShape& sh = fft_shape(lp);
    if (lp.lds_fft > 64 * 1024) MI_HIP(hipFuncSetAttribute(sh.fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp.lds_fft));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(sh.fwd_fn), dim3(8 * ((lp.n_short + 7) / 8), np), dim3(sh.threads), lp.lds_fft, s, m1, m2, pstride, lp.n_long,
                       lp.n_short, lp.ls, lp.ss, NKP, lp.fft, tw, ft.slot_pos, ft.slot_neg, ws.SF[m].as<cplx>(), ws.ST[m].as<cplx>());
    MI_TRY(launch_check("k_lag_fwd"));
    if (after_fwd) MI_HIP(hipEventRecord(after_fwd, s));
    {
        using MacFn = void (*)(const cplx*, const cplx*, int, int, int, int, int, int, int, int, int, int, int, cplx*);
        static const MacFn macs[] = {k_lag_mac<4, 0>, k_lag_mac<4, 1>, k_lag_mac<4, 2>, k_lag_mac<4, 3>, k_lag_mac<4, 4>, k_lag_mac<4, 5>, k_lag_mac<4, 6>};
        const int pfn = (lp.n_short * lp.KT + 255) / 256;  // float4 pairs per thread of a tile; beyond the table: no register prefetch
        MacFn mac = macs[pfn <= 6 ? pfn : 0];
        const int tiles_per_pair = ((NK + lp.KT - 1) / lp.KT + 1) & ~1, ntiles = tiles_per_pair * np;  // (even: see k_lag_mac)
        int cus = 256, per_cu = 1;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (lp.lds_mac > 64 * 1024)
            MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mac), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp.lds_mac));
        per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lp.lds_mac, 1)));
        const int grid = std::max(16, std::min((ntiles + 15) & ~15, (cus * per_cu) & ~15));
        hipLaunchKernelGGL(mac, dim3(grid), dim3(256), lp.lds_mac, s, ws.SF[m].as<cplx>(), ws.ST[m].as<cplx>(), lp.n_short, NK, lp.Es, lp.KT, lp.JP,
                           lp.FW, lp.TW, nlp, NKP, tiles_per_pair, ntiles, ws.CH[m].as<cplx>());
    }
    MI_TRY(launch_check("k_lag_mac"));
    if (lp.lds_fft > 64 * 1024) MI_HIP(hipFuncSetAttribute(sh.inv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp.lds_fft));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(sh.inv_fn), dim3((nlag + 1) / 2, np), dim3(sh.threads), lp.lds_fft, s, ws.CH[m].as<cplx>(), NK, nlp, lp.fft, lp.Es, lp.El,
                       lp.long_is_u ? 1 : 0, lp.Eu, lp.Ev, tw, ft.slot_freq, ws.cross[m].as<double>());
    return launch_check("k_lag_inv");
}

// layout of one plane's banded tables (doubles): c0a, c0b | P1 | Q1 | P2 | Q2 | TS1 | TS2 | partial pixel sums | chunk column sums
struct BandLayout {
    int B, rows, nch;
    size_t tab, ts, total;
    BandLayout(const PlaneGeom& g, const LagPlane& lp) {
        B = lp.El + TILE;                                   // rows read: within E (+ 31 for the tile-aligned ones) of either end
        if (2 * (B + 1) >= lp.n_long + 1) B = lp.n_long;    // no gain: keep every row
        rows = B >= lp.n_long ? lp.n_long + 1 : 2 * (B + 1);
        nch = (lp.n_long + BAND_CH - 1) / BAND_CH;
        tab = (size_t)rows * (lp.n_short + 1);
        ts = (size_t)(g.dimu / TILE + 1) * (g.dimv / TILE + 1);
        total = 2 + 4 * tab + 2 * ts + 2 * MEAN_PARTS + 4 * (size_t)nch * lp.n_short;
    }
};

// float tile sums (reference order), global means and the banded tables of both MIPs of a plane, `np` pairs at once
int prepare_plane_band(hipStream_t s, const float* m1, const float* m2, const PlaneGeom& g, const LagPlane& lp, float* ps1, float* ps2, double* sat,
                       int np, size_t pstride, size_t sstride) {
    const BandLayout L(g, lp);
    double *c0a = sat, *c0b = sat + 1, *P1 = sat + 2, *T1 = P1 + 4 * L.tab, *T2 = T1 + L.ts, *part = T2 + L.ts, *chunks = part + 2 * MEAN_PARTS;
    if (g.tiled) {
        const int nt = (g.dimu / TILE) * (g.dimv / TILE);
        hipLaunchKernelGGL(k_tile_sums, dim3(nt, 2, np), dim3(64), 0, s, m1, m2, pstride, g.dimu, g.dimv, ps1, ps2);
        MI_TRY(launch_check("k_tile_sums"));
    }
    hipLaunchKernelGGL(k_mip_partial, dim3(MEAN_PARTS, 2, np), dim3(256), 0, s, m1, m2, pstride, sstride, (size_t)g.dimu * g.dimv, part);
    MI_TRY(launch_check("k_mip_partial"));
    hipLaunchKernelGGL(k_mip_mean, dim3(2, 1, np), dim3(1024), 0, s, part, pstride, sstride, g.dimu, g.dimv, g.tiled ? ps1 : nullptr,
                       g.tiled ? ps2 : nullptr, c0a, c0b, T1, T2);
    MI_TRY(launch_check("k_mip_mean"));
    const int cblocks = (lp.n_short + 63) / 64;
    if (L.B < lp.n_long) {  // the lower band starts from the chunk sums
        if (lp.long_is_u)
            hipLaunchKernelGGL(k_band_chunksum<false>, dim3(cblocks, L.nch, 2 * np), dim3(64), 0, s, m1, m2, pstride, sstride, c0a, lp.n_long, lp.n_short,
                               lp.ls, lp.ss, L.nch, chunks);
        else
            hipLaunchKernelGGL(k_band_chunksum<true>, dim3(lp.n_short, L.nch, 2 * np), dim3(64), 0, s, m1, m2, pstride, sstride, c0a, lp.n_long, lp.n_short,
                               lp.ls, lp.ss, L.nch, chunks);
        MI_TRY(launch_check("k_band_chunksum"));
    }
    hipLaunchKernelGGL(k_band_cols, dim3(cblocks, 2, 2 * np), dim3(64), 0, s, m1, m2, pstride, sstride, c0a, lp.n_long, lp.n_short, lp.ls, lp.ss, L.nch,
                       L.B, L.tab, chunks, P1);
    MI_TRY(launch_check("k_band_cols"));
    hipLaunchKernelGGL(k_band_rows, dim3(L.rows, 4, np), dim3(64), 0, s, sstride, lp.n_short, L.tab, P1);
    return launch_check("k_band_rows");
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
    size_t pstride = 0, sstride = 0, sat_off[3] = {0, 0, 0};
    ~LagJob() {
        if (ws) {
            // (whatever the call that owned this job enqueued must not outlive the buffers' next user)
            if (enqueued && ws->ev_done) (void)ws->wait_done();
            else if (ws->sm) {
                (void)hipStreamSynchronize(ws->sm);
                for (hipStream_t st : ws->sl) (void)hipStreamSynchronize(st);
                if (ws->sx) (void)hipStreamSynchronize(ws->sx);
            }
            give_lag_ws(std::move(ws));
        }
    }
};

// Device stage of a group: enqueued behind the work `s` holds so far, on the device's MIP stream and its three chain streams.
// What overlaps: the MIP pass of the NEXT group with the chains of this one, and the three planes' chains with each other.
// (MI_NCC_PIECES > 1 cuts a group into pieces that are pipelined the same way; measured, it loses: the chain is a dozen
// latency-bound launches whose cost hardly depends on the number of pairs, so pieces multiply it -- 9.3 / 10.2 / 12.3 ms per 112
// pairs for 1 / 2 / 4 pieces.)
static int enqueue_chains(LagJob& job, int c0, int p0, int np, int pi, hipEvent_t gate, hipEvent_t gate_xy = nullptr);
static int close_job(LagJob& job);

// MI_NCC_GATE (default 1): a group's MIP pass starts only when the previous group's xy chain is past its tables and its forward lag
// transform.  Those are bound by memory latency and run three times longer beside a MIP pass (which they slow down in turn); what
// is left of the chain -- the fp64 correlation, the inverse transform, the refinement -- is compute-bound and shares the device well.
static bool xy_tables_aside() {
    static const bool on = [] {
        const char* e = std::getenv("MI_NCC_XY_TABLES_ASIDE");
        return e ? std::atoi(e) != 0 : true;
    }();
    return on;
}
static bool mip_gate() {
    static const bool on = [] {
        const char* e = std::getenv("MI_NCC_GATE");
        return e ? std::atoi(e) != 0 : true;
    }();
    return on;
}

int ncc_lag_enqueue(int dev, hipStream_t s, int n, const float* const* a_ptrs, const float* const* b_ptrs, int dimk, int dimi, int dimj, int ni,
                    int nj, int delayk, int delayi, int delayj, int side, mi_ncc_params* params, LagJob** job_out, bool defer_chains, TileFmt fmt) {
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
        const size_t NK = (size_t)lp[m].fft.N / 2 + 1, nlp = (size_t)lp[m].nlp;
        spec = std::max(spec, 16 * (2 * (size_t)lp[m].n_short * ((NK + 7) & ~(size_t)7) + NK * nlp));
        crs = std::max(crs, 8 * (size_t)(2 * lp[m].Eu + 1) * (2 * lp[m].Ev + 1));
        wcap = std::max(wcap, (2 * pl.g[m].wu + 1) * (2 * pl.g[m].wv + 1));
    }
    job->wcap = wcap;
    const size_t per_pair = 4 * (pstride + tmp_floats) + 8 * sstride + spec + crs + 3 * 4 * (size_t)wcap + 64;
    size_t budget = (size_t)6 << 30;
    if (const char* e = std::getenv("MI_NCC_CHUNK_MB")) budget = (size_t)std::max(1, std::atoi(e)) << 20;
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, budget / per_pair));
    int pieces = 1;
    if (const char* e = std::getenv("MI_NCC_PIECES")) pieces = std::max(1, std::min(64, std::atoi(e)));
    const int piece = std::max(1, (chunk + pieces - 1) / pieces);

    MI_TRY(grow(ws.fbuf, 4 * pstride * chunk));
    MI_TRY(grow(ws.sat, 8 * sstride * chunk));
    MI_TRY(grow(ws.mip_tmp, 4 * tmp_floats * chunk));
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
    for (int m = 0; m < 3; ++m)
        if (m == 0 || ws.sl[m] != ws.sl[m - 1]) MI_HIP(hipStreamWaitEvent(ws.sl[m], ws.ev_start, 0));
    if (mip_gate()) {  // (never recorded yet: no wait)
        MI_HIP(hipStreamWaitEvent(sm, ws.ev_head, 0));
        MI_HIP(hipStreamWaitEvent(sm, ws.ev_head_tab, 0));
    }
    for (int c0 = 0; c0 < n; c0 += chunk) {
        const int nc = std::min(chunk, n - c0);
        if (c0 > 0) {  // the buffers of the previous chunk are free once its chains have run
            for (int m = 0; m < 3; ++m) {
                MI_HIP(hipEventRecord(ws.ev_lag, ws.sl[m]));
                MI_HIP(hipStreamWaitEvent(sm, ws.ev_lag, 0));
            }
        }
        for (int p0 = 0, pi = 0; p0 < nc; p0 += piece, ++pi) {
            const int np = std::min(piece, nc - p0);
            float* base = base0 + (size_t)p0 * pstride;
            const float** dtab = ws.tab.as<const float*>() + 2 * (size_t)p0;
            MI_HIP(hipMemcpyAsync(dtab, htab + 2 * (size_t)(c0 + p0), sizeof(void*) * 2 * np, hipMemcpyHostToDevice, sm));
            MI_TRY(launch_mips(sm, nullptr, nullptr, dtab, np, pstride, pl.dimk, pl.dimi_v, pl.dimj_v, (size_t)dimi * dimj, dimj, pl.ai0, pl.aj0,
                               base + pl.g[0].mip1, base + pl.g[1].mip1, base + pl.g[2].mip1, base + pl.g[0].mip2, base + pl.g[1].mip2,
                               base + pl.g[2].mip2, ws.mip_tmp.as<float>() + (size_t)p0 * tmp_floats, ws.ev_mip_xy[pi], fmt));
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

// the table / lag-transform / refinement chains of the pairs [p0, p0 + np) of a chunk, one plane per chain stream, behind `gate`
// (gate_xy: an earlier event that the xy plane's chain may start behind -- its MIPs come straight out of k_mips)
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
    // the three planes' chains are independent (own MIPs, own tables, own lag-transform scratch): each is a dozen small
    // dependent launches, so they run side by side on their own streams
    for (int m = 0; m < 3; ++m) {
        hipStream_t sl = ws.sl[m];
        hipEvent_t gm = m == 0 ? gate_xy : gate;
        if (m == 0 || sl != ws.sl[m - 1]) MI_HIP(hipStreamWaitEvent(sl, gm, 0));
        const PlaneGeom& g = pl.g[m];
        // the tables (tile sums, means, banded summed-area tables) and the lag transform of a plane read the same MIPs and meet
        // only in the refinement.  For the xy plane, whose chain is the longest, the tables go to the stream of the xz plane (ahead
        // of its chain): half a dozen small launches that fit beside the transform and the correlation.  (The device offers this
        // process three hardware queues besides the default stream's: the MIP stream, the xy chain, everything else.)
        const bool aside = m == 0 && xy_tables_aside() && ws.sl[1] != sl;
        hipStream_t st = aside ? ws.sl[1] : sl;
        hipStream_t sxm = (m == 0 && !aside && ws.sx != sl) ? ws.sx : sl;
        if (st != sl) MI_HIP(hipStreamWaitEvent(st, gm, 0));
        if (sxm != sl) MI_HIP(hipStreamWaitEvent(sxm, gm, 0));
        MI_TRY(prepare_plane_band(st, base + g.mip1, base + g.mip2, g, lp[m], base + g.ps1, base + g.ps2, sat_p + job.sat_off[m], np, pstride,
                                  sstride));
        if (st != sl) {
            MI_HIP(hipEventRecord(ws.ev_x, st));
            if (mip_gate()) MI_HIP(hipEventRecord(ws.ev_head_tab, st));
        }
        MI_TRY(lag_cross(job.dev, sxm, lp[m], base + g.mip1, base + g.mip2, pstride, np, ws, m, m == 0 && sxm == sl && mip_gate() ? ws.ev_head : nullptr));
        if (st != sl) MI_HIP(hipStreamWaitEvent(sl, ws.ev_x, 0));
        if (sxm != sl) {
            MI_HIP(hipEventRecord(ws.ev_x, sxm));
            MI_HIP(hipStreamWaitEvent(sl, ws.ev_x, 0));
        }
        const RefineGeom rg = refine_geom(g, lp[m], job.maxIter, sstride, job.sat_off[m], job.margin);
        if (lp[m].lds_refine > 64 * 1024)
            MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lag_refine), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lp[m].lds_refine));
        float* ow = ws.outw.as<float>() + ((size_t)m * chunk + p0) * wcap;
        int* oi = ws.outi.as<int>() + ((size_t)m * chunk + p0) * 4;
        hipLaunchKernelGGL(k_lag_refine, dim3(np), dim3(256), lp[m].lds_refine, sl, rg, sat_p, ws.cross[m].as<double>(), wcap, ow, oi,
                           (float*)nullptr);
        MI_TRY(launch_check("k_lag_refine"));
        MI_HIP(hipMemcpyAsync(ws.pin_w.as<float>() + ((size_t)m * n + c0 + p0) * wcap, ow, 4 * (size_t)np * wcap, hipMemcpyDeviceToHost, sl));
        MI_HIP(hipMemcpyAsync(ws.pin_i.as<int>() + ((size_t)m * n + c0 + p0) * 4, oi, sizeof(int) * 4 * np, hipMemcpyDeviceToHost, sl));
    }
    return MI_OK;
}

// the end of the job: an event on each plane's stream (LagWorkspace::wait_done)
static int close_job(LagJob& job) {
    LagWorkspace& ws = *job.ws;
    for (int m = 1; m < 3; ++m) {
        if (ws.sl[m] == ws.sl[0]) continue;
        MI_HIP(hipEventRecord(ws.ev_plane[m - 1], ws.sl[m]));
    }
    MI_HIP(hipEventRecord(ws.ev_done, ws.sl[0]));  // (everything `sm` was given lies before the last event a chain waited for)
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
    MI_TRY(ncc_lag_enqueue(dev, s, n, a_ptrs, b_ptrs, dimk, dimi, dimj, ni, nj, delayk, delayi, delayj, side, params, &job, false, fmt));
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
    const int nt = (dimu / TILE) * (dimv / TILE);
    const BandLayout L(g, lp);
    DevBuf ps;
    MI_TRY(ps.alloc(sizeof(float) * 2 * (size_t)(nt > 0 ? nt : 1)));
    MI_TRY(grow(ws.sat, 8 * L.total));
    MI_TRY(grow(ws.outw, 4 * 4));
    MI_TRY(grow(ws.outi, sizeof(int) * 4));
    MI_TRY(prepare_plane_band(s, mip1, mip2, g, lp, ps.as<float>(), ps.as<float>() + (nt > 0 ? nt : 0), ws.sat.as<double>(), 1, 0, L.total));
    MI_TRY(lag_cross(dev, s, lp, mip1, mip2, 0, 1, ws));
    const RefineGeom rg = refine_geom(g, lp, 0, L.total, 0, 0.0f);
    if (lp.lds_refine > 64 * 1024)
        MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lag_refine), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp.lds_refine));
    hipLaunchKernelGGL(k_lag_refine, dim3(1), dim3(256), lp.lds_refine, s, rg, ws.sat.as<double>(), ws.cross[0].as<double>(), 1, ws.outw.as<float>(),
                       ws.outi.as<int>(), map);
    MI_TRY(launch_check("k_lag_refine"));
    MI_HIP(hipStreamSynchronize(s));
    return MI_OK;
}

// average duration of ONE k_mips launch over n pairs (HIP events on `s` around `reps` launches; bench.py's roofline hook)
int ncc_time_mips(int dev, hipStream_t s, int n, const float* const* a_ptrs, const float* const* b_ptrs, int dimk, int dimi, int dimj, int ni, int nj,
                  int side, int reps, float* ms, TileFmt fmt) {
    MI_REQUIRE(n > 0 && reps > 0 && ms && a_ptrs && b_ptrs, "mi_ncc_time_mips: invalid arguments");
    MI_REQUIRE(fmt.bytes == 4 || mips_int_ok(fmt.bytes, dimk, dimj, (size_t)dimi * dimj),
               "mi_ncc_time_mips: integer tiles need rows of whole 32-bit words and at most %d slices", 4 * MIP_KPW);
    const int dimi_v = side == MI_NORTH_SOUTH ? dimi - ni : dimi, dimj_v = side == MI_WEST_EAST ? dimj - nj : dimj;
    MI_REQUIRE(dimi_v > 0 && dimj_v > 0 && dimk > 0, "mi_ncc_time_mips: empty view");
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
    const int bands = (dimi_v + MIP_ROWS * nb - 1) / (MIP_ROWS * nb), cblocks = (dimj_v + 63) / 64;
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
    const char* ke = std::getenv("MI_NCC_MIPS_KNOCK");  // (measurement aid, see k_mips)
    const int knock = ke ? std::atoi(ke) : 0;
    for (int r = -1; r < reps; ++r) {  // r = -1: warm-up
        if (r == 0) MI_HIP(hipEventRecord(e0, s));
        if (fmt.bytes != 4) {
            const int aj0 = side == MI_WEST_EAST ? nj : 0, wcol = mips_fmt_width(fmt.bytes);
            const dim3 grid((dimj_v + (aj0 & (wcol - 1)) + wcol - 1) / wcol, bands, 2 * n);
            const unsigned char* const* t8 = tab.as<const unsigned char*>();
            if (fmt.bytes == 2)
                hipLaunchKernelGGL(k_mips_int<2>, grid, dim3(256), lds, s, (const unsigned char*)nullptr, (const unsigned char*)nullptr, t8, pstride, dimk,
                                   dimi_v, dimj_v, (size_t)dimi * dimj, dimj, side == MI_NORTH_SOUTH ? ni : 0, aj0, fmt.scale, o, o + xy + xz + yz,
                                   tmp.as<float>(), xz_tmp);
            else
                hipLaunchKernelGGL(k_mips_int<1>, grid, dim3(256), lds, s, (const unsigned char*)nullptr, (const unsigned char*)nullptr, t8, pstride, dimk,
                                   dimi_v, dimj_v, (size_t)dimi * dimj, dimj, side == MI_NORTH_SOUTH ? ni : 0, aj0, fmt.scale, o, o + xy + xz + yz,
                                   tmp.as<float>(), xz_tmp);
            continue;
        }
        hipLaunchKernelGGL(HIP_KERNEL_NAME(dimk <= 4 * MIP_KPW ? k_mips<true> : k_mips<false>), dim3(cblocks, bands, 2 * n), dim3(256), lds, s, (const float*)nullptr, (const float*)nullptr, tab.as<const float*>(),
                           pstride, dimk, dimi_v, dimj_v, (size_t)dimi * dimj, dimj, side == MI_NORTH_SOUTH ? ni : 0, side == MI_WEST_EAST ? nj : 0, o,
                           o + xy, o + xy + xz, o + xy + xz + yz, o + 2 * xy + xz + yz, o + 2 * xy + 2 * xz + yz, tmp.as<float>(), xz_tmp, knock);
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
