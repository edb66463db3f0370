// Hand-written FFT convolution pipeline for gfx950 (engine MI_ENGINE_FFT whenever every transform length is 2^a * {1,3,9};
// rocFFT remains the fallback for other shapes).
//
// Why: the rocFFT R2C/C2R route costs 12 transform kernels + multiply + epilogue per convolution
// (profiles/r01_bench_c3_rocfft_kernel_stats.csv: 62 ms per convolution on C3, its strided z pass alone
// 13.8 ms).  The RL iteration is HBM-bound, so the design goal is the minimum number of full-volume passes
// with every global access in >= 64-B contiguous segments:
//
//   real volume (Z,Y,X) is read as complex rows of Hx = X/2 samples z[n] = x[2n] + i x[2n+1]; a COMPLEX 3-D FFT
//   of size (Z, Y, Hx) is taken; the real-spectrum untangling, the OTF product and the re-tangling are done
//   point-wise on mirror pairs (k, -k) in the middle pass, so no (X/2+1)-wide array ever exists.
//
//   P1  x forward   rows -> LDS -> DIF FFT(Hx) -> S[z][px][y]            (transposed write: y fastest, px stride 16 KB)
//   P2  y forward   whole contiguous columns: S[z][px][.] -> T[px][z][.]   (the re-layout is free: 16-KB chunks)
//   P3  z forward + untangle * OTF (or conj / explicit adjoint OTF) + retangle + z inverse, on mirror line pairs,
//                   T -> S, both [px][z][py] (z stride 16 KB instead of 16 MB in the [z][px][py] layout)
//   P4  y inverse   S[px][z][.] -> T[z][px][.]
//   P5  x inverse   T[z][px][y] -> LDS -> DIT IFFT(Hx) -> real row -> fused RL epilogue -> out
//   P5+P1 fused     ... -> epilogue result stays in LDS -> DIF FFT(Hx) -> S of the NEXT convolution (8 passes / RL iteration)
//
// Forward transforms are decimation-in-frequency (natural in, permuted out), inverse ones decimation-in-time
// (permuted in, natural out), so no reordering pass exists: frequency-domain arrays simply live in permuted
// positions (px, py, pz) and the OTF is built by the pipeline itself in that order (k_z_conv<BUILD>).
// Each transform runs inside LDS as super-stages of 3 fused radix-2 stages held in registers (8 points per lane);
// see "LDS image", "super-stage chains" and the pipelined kernels below for how the LDS, VALU and HBM phases are
// kept conflict-free, short and overlapped.
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "fft_native.h"

namespace mi {
namespace {

#ifndef MI_FFT_UNROLL
#define MI_FFT_UNROLL 4
#endif
constexpr int kThreadsXZ = 1024;  // strided passes: one ~140-KB work-group of 16 waves per CU
constexpr int kWavesXZ = 4;       // waves per SIMD the register budget is sized for (128 VGPRs)
constexpr int kThreadsY = 512;    // contiguous pass: two work-groups per CU
#ifndef MI_YCUT
#define MI_YCUT 1
#endif
constexpr int kYCut = MI_YCUT;    // super-stage cut of the y kernels (seg_r); -DMI_YCUT=0: the round-4 cut, for A/B builds

// ------------------------------------------------------------------------------------------------ LDS image
// Element i (8 B) of a row sits at slot i ^ G(bits 4..7 of i) ^ rmask(row).  DS traffic is banked per instruction
// (MI355X_MICROARCH.md "LDS"): ds_read_b64 serves 32 lanes per LDS cycle from 64 dword banks, i.e. it is conflict-free when
// the 32 slots differ mod 32; ds_write_b64 serves 16 lanes from 32 dword banks (slots must differ mod 16).  G is GF(2)-linear
// with columns (15, 13, 25, 16) for bits 4..7: with it every butterfly pattern of the super-stage chains below ((0,3), (3,3),
// (3,2), (3,1) and all S_LO >= 5), the stride-2 row accesses and -- together with rmask -- the transposed tile accesses are
// conflict-free under both rules.  (An additive pad of one slot per 32 leaves 2- and 4-way conflicts on the stages with
// 0 < S_LO < 5: 43 % of the LDS cycles of the x pass were conflict cycles, profiles/r01_sq_counters_padded_layout.txt.)
__host__ __device__ constexpr int swz_g(int t) { return ((t & 1) ? 15 : 0) ^ ((t & 2) ? 13 : 0) ^ ((t & 4) ? 25 : 0) ^ ((t & 8) ? 16 : 0); }
__host__ __device__ constexpr int swz_c(int i) { return i ^ swz_g((i >> 4) & 15); }
__host__ __device__ constexpr unsigned long long swz_table_hi() {  // G restricted to bits 5..7, 8 entries of 5 bits
    unsigned long long v = 0;
    for (int t = 0; t < 8; ++t) v |= (unsigned long long)swz_g(2 * t) << (5 * t);
    return v;
}
__device__ __forceinline__ int phys(int i) {
    constexpr unsigned long long T = swz_table_hi();
    const int hi = (int)((T >> (5 * ((i >> 5) & 7))) & 31ull);
    return i ^ hi ^ (__builtin_amdgcn_sbfe(i, 4, 1) & 15);
}
// Rows: the transposed accesses of the x and z passes put `hp` row pairs x (32 / hp) consecutive elements into one lane group
// (16 / hp for a store); row 2 rp (+1) is XOR-ed with a mask that spreads the rp bits over the banks the elements leave free.
__device__ __forceinline__ int rmask(int row, int hp) {
    const int rp = (row >> 1) & (hp - 1);
    const int s = hp == 8 ? 1 : hp == 4 ? 2 : hp == 2 ? 3 : 0;
    return (rp << s) ^ ((rp & 1) << 4);
}
// rows start on a multiple of 32 slots, so that only the masks decide the banks
__host__ __device__ constexpr int row_pitch(int n) { return (n + 31) & ~31; }

__device__ __forceinline__ int launder(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
// Work-group barrier that orders LDS traffic only.  Nothing in these kernels communicates through global memory inside a
// launch, so the barrier must not drain the vector-memory queue: global loads issued before an FFT phase (the next tile,
// the epilogue operand) stay in flight across the phase's barriers and are waited for at their first use.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// Wave-level ordering of LDS traffic: DS operations of one wave execute in order, so data a wave wrote is visible to its own
// later reads (any lane) without a barrier; the fence only keeps the compiler from reordering them.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// between two phases on a tile: rows private to their owner waves need no work-group barrier
__device__ __forceinline__ void stage_sync(bool priv) {
    if (priv) wave_lds_fence();
    else lds_barrier();
}

// 1 / max(c, eps) of the RL ratio step (decon.m:164) with the hardware reciprocal (1 ulp): the IEEE division sequence costs
// 11 VALU instructions per value, a sixth of the fused x pass, for a difference far inside the fp32 noise of the transforms
__device__ __forceinline__ float rcp_eps(float c) { return __builtin_amdgcn_rcpf(fmaxf(c, kEpsSingle)); }

// Complex arithmetic on packed pairs: written on 2-vectors with explicit lane shuffles so that every complex product becomes
// v_pk_mul_f32 + v_pk_fma_f32 (lane selects and the swapped / negated twiddle are operand modifiers or hoisted set-up);
// from the scalar formulas the compiler emits one v_pk_mul + two half-used v_pk_fma + a move per product.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f V2(float2 a) { return v2f{a.x, a.y}; }
__device__ __forceinline__ float2 F2(v2f a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    const v2f A = V2(a), B = V2(b);
    const v2f axx = __builtin_shufflevector(A, A, 0, 0), ayy = __builtin_shufflevector(A, A, 1, 1);
    const v2f bs = {-B.y, B.x};
    return F2(__builtin_elementwise_fma(ayy, bs, axx * B));
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {  // a * conj(b)
    const v2f A = V2(a), B = V2(b);
    const v2f axx = __builtin_shufflevector(A, A, 0, 0), ayy = __builtin_shufflevector(A, A, 1, 1);
    const v2f bc = {B.x, -B.y}, bsw = {B.y, B.x};
    return F2(__builtin_elementwise_fma(ayy, bsw, axx * bc));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return F2(V2(a) + V2(b)); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return F2(V2(a) - V2(b)); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ unsigned brev_n(unsigned v, int bits) { return bits == 0 ? 0u : (__brev(v) >> (32 - bits)); }

// An axis of length N = r3 * 2^l2 (r3 in {1, 3, 9}) is transformed by one radix-r3 DIF stage followed by r3 power-of-two
// sub-transforms; position p = k1 * 2^l2 + p' then holds frequency k1 + r3 * brev(p').
__device__ __forceinline__ int pos2freq(int p, int l2, int r3) {
    const int k1 = p >> l2, pp = p & ((1 << l2) - 1);
    return k1 + r3 * (int)brev_n((unsigned)pp, l2);
}
__device__ __forceinline__ int freq2pos(int k, int l2, int r3) {
    const int k2 = k / r3, k1 = k - k2 * r3;
    return (k1 << l2) + (int)brev_n((unsigned)k2, l2);
}
__device__ __forceinline__ int mirror_pos(int p, int n, int l2, int r3) {
    const int k = pos2freq(p, l2, r3);
    return freq2pos(k == 0 ? 0 : n - k, l2, r3);
}
__device__ __forceinline__ int y_pos2freq(int p, const NativeDims& d) { return pos2freq(p, d.ly2, d.r3); }
__device__ __forceinline__ int y_mirror_pos(int p, const NativeDims& d) { return mirror_pos(p, d.ny, d.ly2, d.r3); }

// Padded mode (PadWindow::on): the transform grid is larger than the caller's volume.  Source sample of grid coordinate g
// on axis a (-1: zero): zero rule = the data sits at [o, o + n); replicate rule = clamped samples inside the window [0, w).
__device__ __forceinline__ int pad_src(const PadWindow& p, int a, int g) {
    if (p.rep[a]) return g < p.w[a] ? min(max(g - p.o[a], 0), p.n[a] - 1) : -1;
    const int s = g - p.o[a];
    return (s >= 0 && s < p.n[a]) ? s : -1;
}
// Output sample of grid coordinate g (-1: the grid point is not part of the cropped result)
__device__ __forceinline__ int pad_dst(const PadWindow& p, int a, int g) {
    const int s = g - p.o[a];
    return (s >= 0 && s < p.n[a]) ? s : -1;
}

// exp(-2 pi i m / 2^(bpos+1)), m < 2^bpos, bpos <= 3: the part of a butterfly twiddle that depends only on the
// register index, as compile-time constants (cos/sin of multiples of 2 pi / 16)
__device__ __forceinline__ constexpr float c16(int k) {
    constexpr float c[8] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                            0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f};
    return c[k];
}
__device__ __forceinline__ constexpr float s16(int k) {
    constexpr float sn[8] = {0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f,
                             1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f};
    return sn[k];
}

// ------------------------------------------------------------------------------------------------ super-stage chains
// The log2(N) radix-2 stages of a transform are cut, bottom-up, into super-stages of 3 stages (8 points per lane in
// registers; a remainder of 4 becomes 2 + 2 -- one stage of 16 points for 1024-point transforms --, a remainder of 1 or 2 sits at the top): seg_r(logn, s) is the length of the
// super-stage that starts at stage s.  The same cut serves both directions (forward walks it top-down, inverse bottom-up),
// and all its (S_LO, LR) pairs below stage 5 are among the conflict-free patterns of the swizzle.
__host__ __device__ constexpr int seg_r(int logn, int s, int cut = 0) {
    const int rem = logn - s;
    if (logn == 4) return s == 0 ? 3 : 1;
    // 1024 points as 8 x 8 x 16 -- three LDS round trips instead of the four of 8 x 8 x 4 x 4 (round 4; C3: ratio launch of the x
    // pass 5.07 -> 4.76 ms, the z pass of 1024-point lines 6.15 -> 5.71 ms, the y passes of C2 0.426 -> 0.416 ms).  The sixteen-point
    // butterfly reads its fifteen twiddles where it uses them (butterflies): held together they spilled.
    if (logn == 10 && rem == 4) return 4;
    // cut 1 (the y kernels: 512 threads, 256 registers to spend): 2048 points as 16 x 16 x 8 and 4096 as 16 x 16 x 16 -- three round
    // trips instead of four (round 5; C3: y passes 3.06 / 3.10 -> 2.99 / 2.93 ms).  The x kernels keep 8 x 8 x 8 x 4 for 2048 points:
    // at their 128 registers the sixteen-point butterflies cost more than the round trip (C4-shaped rank: x pass 7.0 / 7.9 ms
    // against 7.7 / 8.6 with 16 x 16 x 8 and 9.5 / 10.6 with 8 x 16 x 16, profiles/r05_fft_cut_2048.txt).
    if (cut == 1 && logn == 11) return s < 8 ? 4 : 3;
    if (cut == 1 && logn == 12) return 4;
    return rem >= 5 ? 3 : rem == 4 ? 2 : rem;  // rem in {1, 2, 3}: all of it
}
// start of the super-stage that ends at stage `top` (exclusive)
__host__ __device__ constexpr int seg_below(int logn, int top, int cut = 0) {
    int s = 0;
    while (s + seg_r(logn, s, cut) < top) s += seg_r(logn, s, cut);
    return s;
}
// LDS twiddle tables: every super-stage with S_LO > 0 owns a packed table of 2^S_LO entries, exp(-2 pi i m / 2^(S_LO+LR))
// (stride-1 look-ups: no bank conflicts, and no vector-memory loads inside the FFT phases -- those would drain the prefetch
// queue, vmcnt being in order); the tables lie one after the other, bottom-up.  tw_off: offset of the table of stage s.
// Powers kept per lane-twiddle index: all R - 1 of them while the table stays small (stage <= 6), else only the first (the
// others are derived by multiplications).
__host__ __device__ constexpr int tw_powers(int s, int r) { return s <= 6 ? (1 << r) - 1 : 1; }
__host__ __device__ constexpr int tw_off(int logn, int s, int cut = 0) {
    int off = 0, t = 0;
    while (t < s) {
        if (t > 0) off += tw_powers(t, seg_r(logn, t, cut)) << t;
        t += seg_r(logn, t, cut);
    }
    return off;
}
__host__ __device__ constexpr int chain_entries(int logn, int cut = 0) { return tw_off(logn, logn, cut); }
// gather the tables from the global table tw[e] = exp(-2 pi i e / 2^LOGN), e < 2^(LOGN-1): entry [p - 1][m] of the super-stage
// at S is exp(-2 pi i m p / 2^(S+r)), the p-th power of the lane twiddle of group element m
template <int LOGN, int NT, int S = 0, int CUT = 0>
__device__ __forceinline__ void fill_chain_tw(float2* twl, const float2* __restrict__ tw) {
    if constexpr (S < LOGN) {
        constexpr int r = seg_r(LOGN, S, CUT);
        if constexpr (S > 0) {
            constexpr int np = tw_powers(S, r);
            for (int i = threadIdx.x; i < (np << S); i += NT) {
                const int p = (i >> S) + 1, m = i & ((1 << S) - 1);
                const int e = (m * p) << (LOGN - S - r);  // < 2^LOGN
                const float2 t = tw[e & ((1 << (LOGN - 1)) - 1)];
                twl[tw_off(LOGN, S, CUT) + i] = (e >> (LOGN - 1)) ? make_float2(-t.x, -t.y) : t;  // exp(-i(x + pi)) = -exp(-ix)
            }
        }
        fill_chain_tw<LOGN, NT, S + r, CUT>(twl, tw);
    }
}
// LDS layout behind the tile of an axis kernel: [chain tables][radix-3/9 table: exp(-2 pi i n2 / N), n2 < 2^L2]
template <int L2, int R3, int CUT = 0>
struct TwLds {
    static constexpr int r3 = chain_entries(L2, CUT);
    static constexpr int total = r3 + (R3 > 1 ? (1 << L2) : 0);
    // tw: global table of the axis ([sub/2 power-of-two part][full circle of N when R3 > 1])
    template <int NT>
    static __device__ __forceinline__ void fill(float2* twl, const float2* __restrict__ tw) {
        fill_chain_tw<L2, NT, 0, CUT>(twl, tw);
        if constexpr (R3 > 1) {
            const float2* twM = tw + (1 << L2) / 2;
            for (int n2 = threadIdx.x; n2 < (1 << L2); n2 += NT) twl[r3 + n2] = twM[n2];
        }
    }
};
// twiddle entries in LDS for an axis of length n = r3 * 2^l2
__host__ __device__ constexpr int axis_tw_entries(int n) {
    int r3 = 1, l2 = 0;
    while (n % 3 == 0) { n /= 3; r3 *= 3; }
    while (n % 5 == 0) { n /= 5; r3 *= 5; }
    while ((1 << l2) < n) ++l2;
    return chain_entries(l2) + (r3 > 1 ? (1 << l2) : 0);
}
#ifndef MI_Y_TILE_CAP
#define MI_Y_TILE_CAP 16
#endif
constexpr int kLdsOneWg = 156 * 1024;  // one work-group per CU (160 KB LDS)
constexpr int kLdsTwoWg = 78 * 1024;   // two work-groups per CU
constexpr size_t kSpecGapBytes = 4224;  // bytes between the end of S and the start of T (NativeFft::init)
constexpr int kRowPadBytes = 4224;     // padding behind the rows of the spectrum arrays ...
constexpr size_t kPadRowBytes = 8192;  // ... that are at least this long (NativeFft::init)
constexpr int kPairLines = 8;          // lines per block of the pair-interleaved z-side layout (8 A + 8 B lines = 128 bytes)
// rows of an x tile / line pairs of a z tile: 16 (full 128-B lines in the transposed layouts) while tile + tables fit one
// work-group per CU; columns of a y tile: two work-groups per CU
__host__ __device__ constexpr int x_tile_rows(int hx) {
    int rows = 16;
    while (rows > 2 && 8 * (rows * row_pitch(hx) + axis_tw_entries(hx)) > kLdsOneWg) rows >>= 1;
    return rows;
}
__host__ __device__ constexpr int z_tile_lines(int nz) {
    int tl = 16;
    while (tl > 2 && 8 * (2 * tl * row_pitch(nz) + axis_tw_entries(nz)) > kLdsOneWg) tl >>= 1;
    return tl;
}
__host__ __device__ constexpr int y_tile_cols(int ny) {
    int tc = MI_Y_TILE_CAP;
    while (tc > 1 && 8 * (tc * row_pitch(ny) + axis_tw_entries(ny)) > kLdsTwoWg) tc >>= 1;
    return tc;
}

// Sequences of a tile: `batch` = rows * R3 power-of-two sub-transforms of length 2^LOGN; sequence b = row b / R3, sub-block
// b % R3 (elements [sub << LOGN, (sub + 1) << LOGN) of the row).  PRIV: the rows are dealt to the waves (row r belongs to
// wave r mod NW) and every phase between two tile-wide barriers touches a row only through its owner, so the super-stages of
// a chain follow each other without work-group barriers and the waves drift apart (LDS and VALU phases of different waves
// overlap).  Otherwise the sequences are split over all lanes and a barrier follows every super-stage.
template <int LOGN, int LR, int NT, int R3>
struct SeqMap {
    static constexpr int GL = LOGN - LR, NW = NT / 64;
    int total, first, step;
    int wave;
    bool PRIV;
    __device__ __forceinline__ SeqMap(int batch, bool priv) : PRIV(priv) {
        if (PRIV) {
            wave = threadIdx.x >> 6;
            total = ((batch / R3) / NW * R3) << GL;  // rows % NW == 0 (checked by the caller)
            first = threadIdx.x & 63;
            step = 64;
        } else {
            wave = 0;
            total = batch << GL;
            first = threadIdx.x;
            step = NT;
        }
    }
    // work item u -> (row, sub-block, group g)
    __device__ __forceinline__ void at(int u, int& row, int& sub, int& g) const {
        int bl = u >> GL;
        if (GL >= 6) bl = __builtin_amdgcn_readfirstlane(bl);  // 64 consecutive items of a wave share the sequence: SALU row math
        g = u & ((1 << GL) - 1);
        if (R3 == 1) {
            sub = 0;
            row = PRIV ? bl * NW + wave : bl;
        } else {
            const int rl = bl / R3;
            sub = bl - rl * R3;
            row = PRIV ? rl * NW + wave : rl;
        }
    }
};

// multiplication by exp(-2 pi i k16 / 16) (forward) or its conjugate (inverse): compile-time constants; -i / +i are swaps
template <bool CONJ>
__device__ __forceinline__ float2 mul_c16(float2 a, int k16) {
    if (k16 == 0) return a;
    if (k16 == 4) return CONJ ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
    const float2 c = make_float2(c16(k16), CONJ ? s16(k16) : -s16(k16));
    return cmul(a, c);
}
__host__ __device__ constexpr int bit_rev(int j, int bits) {
    int r = 0;
    for (int b = 0; b < bits; ++b) r |= ((j >> b) & 1) << (bits - 1 - b);
    return r;
}

// The LR radix-2 stages of a super-stage on the R = 2^LR points of one lane, as one radix-R butterfly: the stages only carry
// their compile-time constants exp(-2 pi i jl / 2^(bpos+1)); the lane-dependent part of all twiddles on the path of register j
// collapses to ONE factor w^rev(j), w = exp(-2 pi i m / 2^(S_LO+LR)), applied after the stages (forward, DIF) or, conjugated,
// before them (inverse, DIT) -- R - 1 complex products instead of LR * R / 2.  twl: the super-stage's table, [p - 1][m] = w^p
// for the tw_powers() powers it keeps (unused when S_LO == 0).
template <int LR, int S_LO, bool INVERSE>
__device__ __forceinline__ void butterflies(float2 (&v)[1 << LR], const float2* twl, int m) {
    constexpr int R = 1 << LR, NP = tw_powers(S_LO, LR);
    float2 w[NP == R - 1 ? 1 : R];
    // (all powers in the table: each is read where it is used -- sixteen points per lane and their fifteen twiddles at once do not
    // fit the 128 registers of the strided passes)
    auto tw_of = [&](int j) { return twl[((bit_rev(j, LR) - 1) << S_LO) + m]; };
    if constexpr (S_LO > 0) {
        if constexpr (NP != R - 1) {
            w[1] = twl[m];
#pragma unroll
            for (int p = 2; p < R; ++p) w[p] = (p & 1) ? cmul(w[p - 1], w[1]) : cmul(w[p / 2], w[p / 2]);
        }
        if constexpr (INVERSE) {
#pragma unroll
            for (int j = 1; j < R; ++j) v[j] = cmulc(v[j], NP == R - 1 ? tw_of(j) : w[NP == R - 1 ? 0 : bit_rev(j, LR)]);
        }
    }
#pragma unroll
    for (int step = 0; step < LR; ++step) {
        const int bpos = INVERSE ? step : LR - 1 - step;  // local bit handled by this radix-2 stage
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (j & (1 << bpos)) continue;
            const int jl = j & ((1 << bpos) - 1);
            const int k16 = jl * (8 >> bpos);  // jl / 2^(bpos+1) turns = k16 / 16
            const float2 a = v[j], c = v[j | (1 << bpos)];
            if (INVERSE) {
                const float2 t = mul_c16<true>(c, k16);
                v[j] = cadd(a, t);
                v[j | (1 << bpos)] = csub(a, t);
            } else {
                v[j] = cadd(a, c);
                v[j | (1 << bpos)] = mul_c16<false>(csub(a, c), k16);
            }
        }
    }
    if constexpr (S_LO > 0 && !INVERSE) {
#pragma unroll
        for (int j = 1; j < R; ++j) v[j] = cmul(v[j], NP == R - 1 ? tw_of(j) : w[NP == R - 1 ? 0 : bit_rev(j, LR)]);
    }
}

// element of register 0 of group g: the LR-bit register field is inserted at bit S_LO (register j: | (j << S_LO)).  The map is
// a bit permutation, hence OR/XOR-linear: p0(g1 | g2) = p0(g1) | p0(g2) for disjoint g1, g2.
template <int LR, int S_LO>
__host__ __device__ constexpr int group_elem(int g) { return ((g >> S_LO) << (S_LO + LR)) | (g & ((1 << S_LO) - 1)); }

// One super-stage, everything about the transform compile-time: R = 2^LR points per lane, radix-2 stages
// S_LO+LR-1..S_LO (forward, DIF) or S_LO..S_LO+LR-1 (inverse, DIT) on the sequences of the tile.
// twl: the super-stage's LDS table of lane twiddles and their powers (see butterflies).
template <int LOGN, int LR, int S_LO, bool INVERSE, int NT, int R3>
__device__ __forceinline__ void super_stage(float2* tile, int batch, int pitch, int hp, bool priv, const float2* twl) {
    constexpr int R = 1 << LR, H_LO = 1 << S_LO, GL = LOGN - LR, NW = NT / 64;
    if constexpr (GL >= 6) {
        // The group index of a lane is (lane part) | (step part): `coop` lanes work on one sequence (the wave's 64 when rows are
        // private, else min(NT, groups)), so everything that depends on the sequence and on the step is wave-uniform (SALU)
        // and, the swizzle being XOR-linear, a lane's slots are its own constants XOR one scalar per step.
        constexpr int G = 1 << GL;
        const int coop = priv ? 64 : (NT < G ? NT : G);           // power of two >= 64
        const int lid = priv ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
        const int gl = lid & (coop - 1);
        const int a_lane = phys(group_elem<LR, S_LO>(gl));
        const int m_lane = gl & (H_LO - 1);
        // sequences: private rows -> those of the wave's rows; else sequence (lid / coop) + kb * (NT / coop)
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int b0 = priv ? 0 : __builtin_amdgcn_readfirstlane(lid / coop);
        const int bstep = priv ? 1 : NT / coop;
        const int nseq = priv ? (batch / R3) / NW * R3 : batch;
        for (int bl = b0; bl < nseq; bl += bstep) {
            const int rl = bl / R3, sub = bl - rl * R3, rowi = priv ? rl * NW + wave : rl;
            float2* row = tile + rowi * pitch;
            const int s_seq = swz_c(sub << LOGN) ^ rmask(rowi, hp);
#pragma unroll 1
            for (int gk = 0; gk < G; gk += coop) {
                const int a0 = a_lane ^ (s_seq ^ swz_c(group_elem<LR, S_LO>(gk)));
                float2 v[R];
#pragma unroll
                for (int j = 0; j < R; ++j) v[j] = H_LO >= 256 ? row[a0 + j * H_LO] : row[a0 ^ swz_c(j << S_LO)];
                butterflies<LR, S_LO, INVERSE>(v, twl, m_lane | (gk & (H_LO - 1)));
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    if (H_LO >= 256) row[a0 + j * H_LO] = v[j];
                    else row[a0 ^ swz_c(j << S_LO)] = v[j];
                }
            }
        }
        return;
    }
    const SeqMap<LOGN, LR, NT, R3> map(batch, priv);
#pragma unroll 1
    for (int u = map.first; u < map.total; u += map.step) {
        int rowi, sub, g;
        map.at(u, rowi, sub, g);
        const int m = g & (H_LO - 1);
        const int p0 = (sub << LOGN) | group_elem<LR, S_LO>(g);  // element of register 0; register j: p0 | (j << S_LO)
        float2* row = tile + rowi * pitch;
        // slot(p0 | J) = slot(p0) ^ swz_c(J) (the swizzle is linear and the j field of p0 is zero); for H_LO >= 256 the j
        // field lies above the swizzled bits and the R slots are slot(p0) + j * H_LO: immediate offsets
        const int a0 = phys(p0) ^ rmask(rowi, hp);
        float2 v[R];
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = H_LO >= 256 ? row[a0 + j * H_LO] : row[a0 ^ swz_c(j << S_LO)];
        butterflies<LR, S_LO, INVERSE>(v, twl, m);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (H_LO >= 256) row[a0 + j * H_LO] = v[j];
            else row[a0 ^ swz_c(j << S_LO)] = v[j];
        }
    }
}

// full transform of the tile's sequences as the chain of super-stages; twl: the axis' LDS tables.  The caller synchronises
// before (tile and tables filled): with a work-group barrier, or -- PRIV, and the rows were filled by their owners -- not at
// all.  On return the tile is consistent for the work-group (!PRIV) or for each row's owner (PRIV).
template <int LOGN, bool INVERSE, int NT, int R3 = 1, int DONE = 0, int STOP = LOGN, int CUT = 0>
__device__ __forceinline__ void lds_fft(float2* tile, int batch, int pitch, int hp, bool priv, const float2* twl) {
    if constexpr (DONE < STOP) {
        constexpr int s_lo = INVERSE ? DONE : seg_below(LOGN, LOGN - DONE, CUT);  // forward: top stages first; inverse: bottom first
        constexpr int r = INVERSE ? seg_r(LOGN, DONE, CUT) : LOGN - DONE - s_lo;
        super_stage<LOGN, r, s_lo, INVERSE, NT, R3>(tile, batch, pitch, hp, priv, twl + tw_off(LOGN, s_lo, CUT));
        stage_sync(priv);
        lds_fft<LOGN, INVERSE, NT, R3, DONE + r, STOP, CUT>(tile, batch, pitch, hp, priv, twl);
    }
}

// 3-point DFT in place (forward: exp(-2 pi i /3); inverse: conjugate)
template <bool INVERSE>
__device__ __forceinline__ void dft3(float2& a, float2& b, float2& c) {
    const float hs = 0.86602540378443865f;  // sqrt(3)/2
    const float2 t1 = cadd(b, c);
    const float2 t2 = make_float2(a.x - 0.5f * t1.x, a.y - 0.5f * t1.y);
    const float2 dd = csub(b, c);
    // forward: -i * hs * (b - c) ; inverse: +i * hs * (b - c)
    const float2 t3 = INVERSE ? make_float2(-hs * dd.y, hs * dd.x) : make_float2(hs * dd.y, -hs * dd.x);
    a = cadd(a, t1);
    b = cadd(t2, t3);
    c = csub(t2, t3);
}

// 5-point DFT in place (forward: exp(-2 pi i / 5); inverse: conjugate), in the usual sum / difference form
template <bool INVERSE>
__device__ __forceinline__ void dft5(float2 (&v)[5]) {
    const float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;  // cos(2 pi / 5), cos(4 pi / 5)
    const float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;   // sin(2 pi / 5), sin(4 pi / 5)
    const float2 a1 = cadd(v[1], v[4]), a2 = cadd(v[2], v[3]), b1 = csub(v[1], v[4]), b2 = csub(v[2], v[3]);
    const float2 x0 = v[0];
    const float2 p1 = make_float2(x0.x + c1 * a1.x + c2 * a2.x, x0.y + c1 * a1.y + c2 * a2.y);
    const float2 p2 = make_float2(x0.x + c2 * a1.x + c1 * a2.x, x0.y + c2 * a1.y + c1 * a2.y);
    const float2 q1 = make_float2(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y);
    const float2 q2 = make_float2(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y);
    // forward: X[k] = p -+ i q ... with -i q = (q.y, -q.x); inverse: +i q = (-q.y, q.x)
    const float2 iq1 = INVERSE ? make_float2(-q1.y, q1.x) : make_float2(q1.y, -q1.x);
    const float2 iq2 = INVERSE ? make_float2(-q2.y, q2.x) : make_float2(q2.y, -q2.x);
    v[0] = cadd(x0, cadd(a1, a2));
    v[1] = cadd(p1, iq1);
    v[4] = csub(p1, iq1);
    v[2] = cadd(p2, iq2);
    v[3] = csub(p2, iq2);
}

// radix-R3 stage of the y transform on `cols` LDS rows of length M = R3 * Msub: forward = DIF first stage
// (DFT over n1 of x[n1 * Msub + n2], times W_M^(n2 k1), stored at k1 * Msub + n2); inverse = its exact reverse.
// tw3[n2] = exp(-2 pi i n2 / M), n2 < Msub (LDS); the twiddles W_M^(n2 q), q < R3, are its powers.
template <int R3, bool INVERSE, int NT>
__device__ __forceinline__ void radix3_stage(float2* tile, int cols, int pitch, int hp, bool PRIV, int msub, const float2* tw3) {
    constexpr int NW = NT / 64;
    const int wave = threadIdx.x >> 6;
    const int total = PRIV ? (cols / NW) * msub : cols * msub;
    for (int idx = PRIV ? (threadIdx.x & 63) : threadIdx.x; idx < total; idx += PRIV ? 64 : NT) {
        int cl = idx / msub;
        if (msub >= 64) cl = __builtin_amdgcn_readfirstlane(cl);  // msub is a power of two: a wave's 64 items share the row
        const int n2 = idx - cl * msub;
        const int c = PRIV ? cl * NW + wave : cl;
        float2* row = tile + c * pitch;
        float2 v[R3];
        // slot of element q * msub + n2: n2 < msub and the multiples of msub occupy disjoint bits and the swizzle is XOR-linear,
        // so it is the slot of n2 XOR a constant (no per-q address registers)
        const int s0 = phys(n2) ^ rmask(c, hp);
#pragma unroll
        for (int q = 0; q < R3; ++q) v[q] = row[s0 ^ swz_c(q * msub)];
        float2 wq[R3];  // wq[q] = w1^q, by squaring / one multiplication from lower powers (depth <= 3)
        wq[1] = tw3[n2];
#pragma unroll
        for (int q = 2; q < R3; ++q) wq[q] = (q & 1) ? cmul(wq[q - 1], wq[1]) : cmul(wq[q / 2], wq[q / 2]);
        if (INVERSE) {
#pragma unroll
            for (int q = 1; q < R3; ++q) v[q] = cmulc(v[q], wq[q]);
        }
        if constexpr (R3 == 3) {
            dft3<INVERSE>(v[0], v[1], v[2]);
        } else if constexpr (R3 == 5) {
            dft5<INVERSE>(v);
        } else {  // 9 = 3 x 3: index n = 3 n1 + n2' , k = k1' + 3 k2'
            // DIF order for the forward transform, reversed for the inverse (which takes k-ordered input)
            if constexpr (!INVERSE) {
#pragma unroll
                for (int r = 0; r < 3; ++r) dft3<false>(v[r], v[r + 3], v[r + 6]);       // over n1 (stride 3): -> A[n2'][k1'] at r + 3 k1'
                const float c9[3] = {1.0f, 0.76604444311897801f, 0.17364817766693033f};   // cos(2 pi {0,1,2}/9)
                const float s9[3] = {0.0f, 0.64278760968653933f, 0.98480775301220802f};   // sin(2 pi {0,1,2}/9)
                const float c94 = -0.93969262078590843f, s94 = 0.34202014332566871f;      // 4/9 turn
                v[4] = cmul(v[4], make_float2(c9[1], -s9[1]));   // n2'=1,k1'=1: W9^1
                v[7] = cmul(v[7], make_float2(c9[2], -s9[2]));   // n2'=1,k1'=2: W9^2
                v[5] = cmul(v[5], make_float2(c9[2], -s9[2]));   // n2'=2,k1'=1: W9^2
                v[8] = cmul(v[8], make_float2(c94, -s94));       // n2'=2,k1'=2: W9^4
                // over n2' for each k1': inputs v[0 + 3k1'], v[1 + 3k1'], v[2 + 3k1'] -> X[k1' + 3 k2'] for k2' = 0,1,2
#pragma unroll
                for (int k1 = 0; k1 < 3; ++k1) dft3<false>(v[3 * k1], v[3 * k1 + 1], v[3 * k1 + 2]);
                // now v[3 k1' + k2'] = X[k1' + 3 k2'] : reorder to k order
                float2 t[9];
#pragma unroll
                for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
                    for (int k2 = 0; k2 < 3; ++k2) t[k1 + 3 * k2] = v[3 * k1 + k2];
#pragma unroll
                for (int q = 0; q < 9; ++q) v[q] = t[q];
            } else {
                float2 t[9];
#pragma unroll
                for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
                    for (int k2 = 0; k2 < 3; ++k2) t[3 * k1 + k2] = v[k1 + 3 * k2];
#pragma unroll
                for (int q = 0; q < 9; ++q) v[q] = t[q];
#pragma unroll
                for (int k1 = 0; k1 < 3; ++k1) dft3<true>(v[3 * k1], v[3 * k1 + 1], v[3 * k1 + 2]);
                const float c9[3] = {1.0f, 0.76604444311897801f, 0.17364817766693033f};
                const float s9[3] = {0.0f, 0.64278760968653933f, 0.98480775301220802f};
                const float c94 = -0.93969262078590843f, s94 = 0.34202014332566871f;
                v[4] = cmul(v[4], make_float2(c9[1], s9[1]));
                v[7] = cmul(v[7], make_float2(c9[2], s9[2]));
                v[5] = cmul(v[5], make_float2(c9[2], s9[2]));
                v[8] = cmul(v[8], make_float2(c94, s94));
#pragma unroll
                for (int r = 0; r < 3; ++r) dft3<true>(v[r], v[r + 3], v[r + 6]);
            }
        }
        if (!INVERSE) {
#pragma unroll
            for (int q = 1; q < R3; ++q) v[q] = cmul(v[q], wq[q]);
        }
#pragma unroll
        for (int q = 0; q < R3; ++q) row[s0 ^ swz_c(q * msub)] = v[q];
    }
}

// slot of element e in row `row` of a tile (in float2 units from the tile start)
__device__ __forceinline__ int cell(int row, int pitch, int hp, int e) { return row * pitch + (phys(e) ^ rmask(row, hp)); }

// ---------------------------------------------------------------------------------------------- P1: x forward
// grid: (Y / TY) * Z tiles; tile = TY consecutive rows of one z-plane
template <int LHX2, int R3>
__global__ __launch_bounds__(kThreadsXZ, kWavesXZ) void k_x_forward(const float* __restrict__ in, float2* __restrict__ S, NativeDims d,
                                                         const float2* __restrict__ tw, PadWindow pw) {
    extern __shared__ __attribute__((aligned(16))) float2 tile[];
    constexpr int Hx = R3 << LHX2, NW = kThreadsXZ / 64;
    const int TY = d.ty, hp = TY / 2, pitch = row_pitch(Hx);
    const int ytiles = d.ny / TY;
    const int z = blockIdx.x / ytiles, y0 = (blockIdx.x % ytiles) * TY;
    const int rowq = d.xrow / 2;
    float4* dst = reinterpret_cast<float4*>(S + ((size_t)z * Hx) * d.xrow + y0);
    if (pw.on) {
        // staged load with the boundary rule; a tile that lies entirely in the zero padding transforms to zeros
        const int sz = pad_src(pw, 2, z);
        bool live = false;
        for (int r = 0; r < TY; ++r) live = live || pad_src(pw, 1, y0 + r) >= 0;
        if (sz < 0 || !live) {
            if (z >= d.z_in_hi) return;  // the y pass does not read these planes
            for (int i = threadIdx.x; i < hp * Hx; i += kThreadsXZ) {
                const int px = i / hp, rp = i - px * hp;
                dst[(size_t)px * rowq + rp] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            return;
        }
        for (int i = threadIdx.x; i < TY * Hx; i += kThreadsXZ) {
            const int r = i / Hx, q = i - r * Hx;
            const int sy = pad_src(pw, 1, y0 + r);
            float2 v = make_float2(0.0f, 0.0f);
            if (sy >= 0) {
                const float* row = in + ((size_t)sz * pw.n[1] + sy) * (size_t)pw.n[0];
                const int s0 = pad_src(pw, 0, 2 * q), s1 = pad_src(pw, 0, 2 * q + 1);
                if (s0 >= 0) v.x = row[s0];
                if (s1 >= 0) v.y = row[s1];
            }
            tile[cell(r, pitch, hp, q)] = v;
        }
    } else {
        const float4* src = reinterpret_cast<const float4*>(in + ((size_t)z * d.ny + y0) * (size_t)(2 * Hx));
        const int quads = Hx / 2;  // float4 = 2 complex
        for (int i = threadIdx.x; i < TY * quads; i += kThreadsXZ) {
            const int r = i / quads, q = i - r * quads;
            const float4 v = src[(size_t)r * quads + q];
            const int c0 = cell(r, pitch, hp, 2 * q);  // elements 2q, 2q + 1 are slot neighbours (same bits 4..7)
            tile[c0] = make_float2(v.x, v.y);
            tile[c0 ^ 1] = make_float2(v.z, v.w);
        }
    }
    using TW = TwLds<LHX2, R3>;
    float2* twl = tile + TY * pitch;
    TW::template fill<kThreadsXZ>(twl, tw);
    lds_barrier();
    const bool priv = (TY % NW) == 0;
    if constexpr (R3 > 1) {
        radix3_stage<R3, false, kThreadsXZ>(tile, TY, pitch, hp, priv, 1 << LHX2, twl + TW::r3);
        stage_sync(priv);
    }
    lds_fft<LHX2, false, kThreadsXZ, R3>(tile, TY * R3, pitch, hp, priv, twl);
    if (priv) lds_barrier();
    // transposed store: S[z][px][y0 + r], r fastest; one float4 = rows (2 rp, 2 rp + 1) of one px
#pragma unroll MI_FFT_UNROLL
    for (int i = threadIdx.x; i < hp * Hx; i += kThreadsXZ) {
        const int px = i / hp, rp = i - px * hp;
        const int c0 = cell(2 * rp, pitch, hp, px);
        const float2 a = tile[c0], b = tile[c0 + pitch];
        dst[(size_t)px * rowq + rp] = make_float4(a.x, a.y, b.x, b.y);
    }
}

// ---------------------------------------------------------------------------------------------- P2 / P4: y passes
// whole contiguous columns.  Forward: column (z, px) of src[z][px][.] -> dst[px][z][.]; inverse: the way back.
template <int LY2, int R3, bool INVERSE>
__global__ __launch_bounds__(kThreadsY, 4) void k_y_pass(const float2* __restrict__ src, float2* __restrict__ dst, NativeDims d,
                                                      const float2* __restrict__ tw) {
    extern __shared__ __attribute__((aligned(16))) float2 tile[];
    constexpr int M = R3 << LY2, NW = kThreadsY / 64;
    constexpr int pitch = row_pitch(M), quads = M / 2;
    constexpr int TCC = y_tile_cols(M);  // the tile height the host normally picks: compile-time item decomposition
    const int TC = d.tc, Hx = d.hx, L = d.nz;
    const size_t c0 = (size_t)blockIdx.x * TC + (INVERSE ? (size_t)0 : (size_t)d.yz0 * Hx);  // (forward: columns (z, px), z slowest)
    // padded grids: forward, the columns of all-zero input planes are neither read nor produced (the z pass knows they are
    // zero); inverse, only the planes that survive the crop are transformed
    if (!INVERSE) {
        if ((int)(c0 / Hx) >= d.z_in_hi) return;
    } else if (L % TC == 0) {
        const int z_first = (int)(c0 % L);
        if (z_first >= d.z_out_hi || z_first + TC <= d.z_out_lo) return;
    }
    // row pitches: the x side ([z][px][py]) may carry padding behind every row (NativeDims::xrow)
    const size_t src_pitch = INVERSE ? (size_t)M : (size_t)d.xrow, dst_pitch = INVERSE ? (size_t)d.xrow : (size_t)M;
    const float4* base = reinterpret_cast<const float4*>(src + c0 * src_pitch);
    // columns dealt to the waves when there are enough of them: then the fill, the transform and the drain of a column all
    // belong to one wave and the kernel has no work-group barrier besides the one behind the table fill
    const bool priv = (TC % NW) == 0;
    using TW = TwLds<LY2, R3, kYCut>;
    float2* twl = tile + TC * pitch;
    TW::template fill<kThreadsY>(twl, tw);
    // destination of source column sc: forward [z][px] -> [px][z], inverse [px][z] -> [z][px]
    auto dest_col = [&](size_t sc) {
        if (INVERSE) { const size_t px = sc / L, z = sc - px * L; return z * Hx + px; }
        const size_t z = sc / Hx, px = sc - z * Hx;
        return px * L + z;
    };
    // fast path: float4 item k of a lane is quad tid + (k NT mod quads) of column (k NT) / quads -- the column is a
    // compile-time number (its addresses are scalar), the slot is the lane's constant XOR a compile-time constant
    constexpr bool FAST_OK = (quads % kThreadsY == 0) && ((TCC * quads) % kThreadsY == 0);
    constexpr int NIT = FAST_OK ? TCC * quads / kThreadsY : 1;
    const bool fast = FAST_OK && TC == TCC && !priv;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_items = priv ? (TC / NW) * quads : TC * quads, first = priv ? lane : threadIdx.x, step = priv ? 64 : kThreadsY;
    if (fast) {
        const int s_lane = phys(2 * (int)threadIdx.x);
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int c = (k * kThreadsY) / quads, qk = (k * kThreadsY) % quads;
            const float4 v = base[(size_t)c * (src_pitch / 2) + qk + threadIdx.x];
            const int s0 = c * pitch + (s_lane ^ swz_c(2 * qk));
            tile[s0] = make_float2(v.x, v.y);
            tile[s0 ^ 1] = make_float2(v.z, v.w);
        }
    } else {
#pragma unroll MI_FFT_UNROLL
        for (int i = first; i < n_items; i += step) {
            const int cl = i / quads, q = i - cl * quads;
            const int c = priv ? cl * NW + wave : cl;
            const float4 v = base[(size_t)c * (src_pitch / 2) + q];
            const int s0 = c * pitch + phys(2 * q);
            tile[s0] = make_float2(v.x, v.y);
            tile[s0 ^ 1] = make_float2(v.z, v.w);
        }
    }
    lds_barrier();
    if constexpr (!INVERSE && R3 > 1) {
        radix3_stage<R3, false, kThreadsY>(tile, TC, pitch, 1, priv, 1 << LY2, twl + TW::r3);
        stage_sync(priv);
    }
    lds_fft<LY2, INVERSE, kThreadsY, R3, 0, LY2, kYCut>(tile, TC * R3, pitch, 1, priv, twl);
    if constexpr (INVERSE && R3 > 1) {
        radix3_stage<R3, true, kThreadsY>(tile, TC, pitch, 1, priv, 1 << LY2, twl + TW::r3);
        stage_sync(priv);
    }
    if (fast) {
        const int s_lane = phys(2 * (int)threadIdx.x);
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int c = (k * kThreadsY) / quads, qk = (k * kThreadsY) % quads;
            const int s0 = c * pitch + (s_lane ^ swz_c(2 * qk));
            const float2 a = tile[s0], b = tile[s0 ^ 1];
            float4* dcol = reinterpret_cast<float4*>(dst + dest_col(c0 + c) * dst_pitch);  // scalar
            if (!INVERSE || 2 * (qk + (int)threadIdx.x) < d.y_out_hi) dcol[qk + threadIdx.x] = make_float4(a.x, a.y, b.x, b.y);
        }
    } else {
#pragma unroll MI_FFT_UNROLL
        for (int i = first; i < n_items; i += step) {
            const int cl = i / quads, q = i - cl * quads;
            const int c = priv ? cl * NW + wave : cl;
            const int s0 = c * pitch + phys(2 * q);
            const float2 a = tile[s0], b = tile[s0 ^ 1];
            if (!INVERSE || 2 * q < d.y_out_hi) reinterpret_cast<float4*>(dst + dest_col(c0 + c) * dst_pitch)[q] = make_float4(a.x, a.y, b.x, b.y);
        }
    }
}

// ---------------------------------------------------------------------------------------------- P2 / P4, pair-interleaved
// The y passes on the pair-interleaved layout of the z side (NativeDims::paired): row (xk, z), xk <= Hx/2, holds 2 M samples,
// for every block of 8 y positions the 8 lines of plane xk ("A") followed by their 8 mirror partners from plane Hx - xk ("B", in
// partner order: B slot j is the line the z pass pairs with A slot j).  A work-group takes TC/2 z planes x {A, B} of one xk, so
// that it reads and writes whole rows although each plane only owns every other 64 bytes.  Planes 0 and Hx/2 are their own
// partners: their lines are stored twice (as A of their block and as B of the mirror block).
template <int LY2, int R3, bool INVERSE>
__global__ __launch_bounds__(kThreadsY, 4) void k_y_pair(const float2* __restrict__ src, float2* __restrict__ dst, NativeDims d,
                                                      const float2* __restrict__ tw) {
    extern __shared__ __attribute__((aligned(16))) float2 tile[];
    constexpr int M = R3 << LY2, NW = kThreadsY / 64;
    constexpr int pitch = row_pitch(M), quads = M / 2;
    const int TC = d.tc, Hx = d.hx, L = d.nz;
    const int zper = TC / 2, zblocks = L / zper, nxk = d.xkn;  // (a launch covers the planes xk0 .. xk0 + xkn - 1: all, or a chunk)
    // work-groups in flight read neighbouring memory (reads wait, writes do not): forward, the planes of one z pair on the x
    // side; inverse, consecutive rows of one xk on the z side
    const int xkl = INVERSE ? blockIdx.x / zblocks : blockIdx.x % nxk;
    const int xk = d.xk0 + xkl;
    const int z0 = (INVERSE ? blockIdx.x - xkl * zblocks : blockIdx.x / nxk) * zper + (INVERSE ? 0 : d.yz0);
    if (!INVERSE) {
        if (z0 >= d.z_in_hi) return;  // all-zero input planes of a padded grid: neither read nor produced
    } else if (z0 >= d.z_out_hi || z0 + zper <= d.z_out_lo) {
        return;                       // planes the crop drops
    }
    const int pxA = freq2pos(xk, d.lhx2, d.r3x), pxB = freq2pos(xk == 0 ? 0 : Hx - xk, d.lhx2, d.r3x);
    const bool self = pxA == pxB;
    // LDS row c of the tile: side c & 1, plane z0 + (c >> 1)
    // x side ([z][px][py], whole columns): float4 q of column c = positions 2 q, 2 q + 1
    auto x_item = [&](int i, int& c, int& q, size_t& g) {
        c = i / quads;
        q = i - c * quads;
        g = (((size_t)(z0 + (c >> 1)) * Hx + ((c & 1) ? pxB : pxA)) * d.xrow) / 2 + q;
    };
    // z side, by float4 f of row (xk, z0 + zi): block f >> 3; f & 7 < 4: lines 2 (f & 3), + 1 of the block from the A column,
    // else the partners of those two lines from the B column -- the mirrors of neighbouring positions are neighbours (they
    // differ by M/2 in frequency), so both sides read or write one LDS slot pair
    auto z_item = [&](int i, int& c, int& s0, size_t& g) {
        const int zi = i / M, f = i - zi * M;
        const int py = ((f >> 3) << 3) + 2 * (f & 3), side = (f >> 2) & 1;
        c = 2 * zi + side;
        // (inverse: B columns are transformed as they lie and leave row-reversed, see x_slots; forward: the mirror map)
        s0 = c * pitch + phys(!INVERSE && side ? mirror_pos(py, M, LY2, R3) : py);
        g = ((size_t)xk * L + z0 + zi) * (size_t)(M + d.zpad) + f;
    };
    // General path, B columns.  Forward: the transform of the column is stored through the mirror map of the frequency positions
    // (B slot of position p <- position mirror(p)).  Inverse: the B slots are loaded in position order -- the array at position
    // p is X_B[-k(p)], whose inverse transform is the column ROW-REVERSED -- and row n is stored from LDS position -n mod M:
    // no mirror arithmetic and, for every radix, contiguous LDS traffic where the mirror map scatters (y = 9 * 64: inverse
    // pass 1.16 -> 0.97 ms; the forward pass is faster with the mirror map, 0.97 against 1.14 ms).  (The fast path below uses
    // the mirror map in both directions: for power-of-two columns it is XOR-linear.)
    auto x_slots = [&](int c, int q, int& s_lo, int& s_hi) {  // LDS slots of rows 2 q and 2 q + 1 of column c
        if (INVERSE && (c & 1)) {
            s_lo = c * pitch + phys(q == 0 ? 0 : M - 2 * q);
            s_hi = c * pitch + phys(M - 2 * q - 1);
        } else {
            s_lo = c * pitch + phys(2 * q);
            s_hi = s_lo ^ 1;
        }
    };
    const bool priv = (TC % NW) == 0;  // (the transform only: fill and drain cross the columns)
    using TW = TwLds<LY2, R3, kYCut>;
    float2* twl = tile + TC * pitch;
    TW::template fill<kThreadsY>(twl, tw);
    const int n_items = TC * quads;
    // fast path (power-of-two columns of at least 2 NT samples): item k of a lane is float4 tid + k NT of the tile on either
    // side, so columns, rows and the high position bits are compile-time numbers and -- the swizzle being XOR-linear -- a slot
    // is a lane constant XOR a compile-time constant.  x side: position 2 tid + (2 k NT mod M).  z side: the lane's block
    // position py_l = 8 (tid >> 3) + 2 (tid & 3) plus f0 = k NT mod M; the mirror of f0 + py_l is (py_l ^ (NT - 1)) + [mirror
    // of the high bits] unless f0 = 0, when it is the mirror of py_l inside the first NT positions.
    constexpr int TCC = y_tile_cols(M);
    constexpr bool FAST_OK = R3 == 1 && quads % kThreadsY == 0;
    constexpr int NIT = FAST_OK ? TCC * quads / kThreadsY : 1;
    const bool fast = FAST_OK && TC == TCC;
    constexpr int WHI = FAST_OK ? LY2 - 9 : 0;  // position bits above the lane's 9 (kThreadsY = 512)
    static_assert(kThreadsY == 512, "the fast path of k_y_pair counts on 512 lanes");
    struct ZLane { int a, b0, b1, side; };
    auto z_lane = [&]() {
        const int tid = launder(threadIdx.x);
        const int py_l = ((tid >> 3) << 3) + 2 * (tid & 3), side = (tid >> 2) & 1;
        return ZLane{phys(py_l), phys(mirror_pos(py_l, M, LY2, R3)), phys(py_l ^ 511), side};
    };
    auto z_slot_fast = [&](const ZLane& zl, int k) {  // k: compile-time after unrolling
        const int zi = (k * kThreadsY) / M, f0 = (k * kThreadsY) % M;
        const int flo = (int)brev_n((unsigned)f0, LY2);                          // the low WHI frequency bits
        const int mhi = flo ? (int)brev_n((unsigned)((1 << WHI) - flo), LY2) : 0;  // position bits of their negative
        const int sa = zl.a ^ swz_c(f0), sb = flo ? (zl.b1 ^ swz_c(mhi)) : zl.b0;
        return (2 * zi) * pitch + (zl.side ? pitch + sb : sa);
    };
    if (fast) {
        if (INVERSE) {
            const ZLane zl = z_lane();
            const float4* rowp = reinterpret_cast<const float4*>(src) + ((size_t)xk * L + z0) * (size_t)(M + d.zpad) + threadIdx.x;
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const float4 v = rowp[(size_t)((k * kThreadsY) / M) * (M + d.zpad) + (k * kThreadsY) % M];
                const int s0 = z_slot_fast(zl, k);
                tile[s0] = make_float2(v.x, v.y);
                tile[s0 ^ 1] = make_float2(v.z, v.w);
            }
        } else {
            const int s_lane = phys(2 * (int)threadIdx.x);
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int c = (k * kThreadsY) / quads, qk = (k * kThreadsY) % quads;
                const size_t col = (((size_t)(z0 + (c >> 1)) * Hx + ((c & 1) ? pxB : pxA)) * d.xrow) / 2;  // scalar
                const float4 v = reinterpret_cast<const float4*>(src)[col + qk + threadIdx.x];
                const int s0 = c * pitch + (s_lane ^ swz_c(2 * qk));
                tile[s0] = make_float2(v.x, v.y);
                tile[s0 ^ 1] = make_float2(v.z, v.w);
            }
        }
    } else {
#pragma unroll MI_FFT_UNROLL
    for (int i = threadIdx.x; i < n_items; i += kThreadsY) {
        int c, s0, s1;
        size_t g;
        if (INVERSE) {
            z_item(i, c, s0, g);
            s1 = s0 ^ 1;
        } else {
            int q;
            x_item(i, c, q, g);
            x_slots(c, q, s0, s1);
        }
        const float4 v = reinterpret_cast<const float4*>(src)[g];
        tile[s0] = make_float2(v.x, v.y);
        tile[s1] = make_float2(v.z, v.w);
    }
    }
    lds_barrier();
    if constexpr (!INVERSE && R3 > 1) {
        radix3_stage<R3, false, kThreadsY>(tile, TC, pitch, 1, priv, 1 << LY2, twl + TW::r3);
        stage_sync(priv);
    }
    lds_fft<LY2, INVERSE, kThreadsY, R3, 0, LY2, kYCut>(tile, TC * R3, pitch, 1, priv, twl);
    if constexpr (INVERSE && R3 > 1) {
        radix3_stage<R3, true, kThreadsY>(tile, TC, pitch, 1, priv, 1 << LY2, twl + TW::r3);
        stage_sync(priv);
    }
    if (priv) lds_barrier();
    if (fast) {
        if (INVERSE) {
            const int s_lane = phys(2 * (int)threadIdx.x);
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int c = (k * kThreadsY) / quads, qk = (k * kThreadsY) % quads;
                const int z = z0 + (c >> 1);
                // (a plane that is its own partner is written once, from its A copy)
                if ((self && (c & 1)) || 2 * (qk + (int)threadIdx.x) >= d.y_out_hi || z < d.z_out_lo || z >= d.z_out_hi) continue;
                const size_t col = (((size_t)z * Hx + ((c & 1) ? pxB : pxA)) * d.xrow) / 2;  // scalar
                const int s0 = c * pitch + (s_lane ^ swz_c(2 * qk));
                const float2 a = tile[s0], b = tile[s0 ^ 1];
                reinterpret_cast<float4*>(dst)[col + qk + threadIdx.x] = make_float4(a.x, a.y, b.x, b.y);
            }
        } else {
            const ZLane zl = z_lane();
            float4* rowp = reinterpret_cast<float4*>(dst) + ((size_t)xk * L + z0) * (size_t)(M + d.zpad) + threadIdx.x;
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int s0 = z_slot_fast(zl, k);
                const float2 a = tile[s0], b = tile[s0 ^ 1];
                rowp[(size_t)((k * kThreadsY) / M) * (M + d.zpad) + (k * kThreadsY) % M] = make_float4(a.x, a.y, b.x, b.y);
            }
        }
        return;
    }
#pragma unroll MI_FFT_UNROLL
    for (int i = threadIdx.x; i < n_items; i += kThreadsY) {
        int c, s0, s1;
        size_t g;
        if (INVERSE) {
            int q;
            x_item(i, c, q, g);
            const int z = z0 + (c >> 1);
            // (a plane that is its own partner is written once, from its A copy)
            if ((self && (c & 1)) || 2 * q >= d.y_out_hi || z < d.z_out_lo || z >= d.z_out_hi) continue;
            x_slots(c, q, s0, s1);
        } else {
            z_item(i, c, s0, g);
            s1 = s0 ^ 1;
        }
        const float2 a = tile[s0], b = tile[s1];
        reinterpret_cast<float4*>(dst)[g] = make_float4(a.x, a.y, b.x, b.y);
    }
}

// ---------------------------------------------------------------------------------------------- P3: z pass + OTF
// One tile = the TL lines (py0 .. py0 + TL) of plane xk ("A", rows 0 .. TL-1 of the LDS tile) and their mirror lines in plane
// Hx - xk ("B", rows TL .. 2 TL - 1): xk runs over 0 .. Hx/2, one representative of every mirror pair of planes.  For
// xk in {0, Hx/2} the mirror line lies in the same plane: every tile is processed in the A role (its mirror tile is only read)
// and only A is written, so each line is still written exactly once; otherwise both lines of a pair are written by the one
// tile that owns the pair.  grid: (Hx/2 + 1) * (Y / TL) tiles.
// OTF layout: G[xk][py][pz] as float4 {Ga.re, Ga.im, Gb.re, Gb.im}, already scaled by 2/(X*Y*Z).
// BUILD: instead of multiplying, the untangled spectrum of the (real) input -- a placed PSF -- is stored as the OTF in that
// same layout, scaled: the pipeline builds its own OTF with the transform it will later apply.
template <int LZ2, int R3, bool BUILD>
__global__ __launch_bounds__(kThreadsXZ, kWavesXZ) void k_z_conv(const float2* __restrict__ S, float2* __restrict__ T, const float4* __restrict__ G,
                                                      NativeDims d, const float2* __restrict__ tw, int conj_otf, float4* __restrict__ Gout,
                                                      float scale) {
    extern __shared__ __attribute__((aligned(16))) float2 tile[];
    constexpr int L = R3 << LZ2, NW = kThreadsXZ / 64;
    const int Hx = d.hx, M = d.ny, TL = d.tl, hp = TL / 2, pitch = row_pitch(L);
    const int ytiles = M / TL;
    const int plane = blockIdx.x / ytiles;
    const int py0 = (blockIdx.x % ytiles) * TL;
    const int xk = plane;
    const int px = freq2pos(xk, d.lhx2, d.r3x);
    const int pxB = freq2pos(xk == 0 ? 0 : Hx - xk, d.lhx2, d.r3x);
    // mirror block of py positions: an aligned block of TL positions maps onto an aligned block (within one power-of-two
    // sub-block: low bits of the frequency fixed -> low bits of its negative fixed)
    const int pyB_any = y_mirror_pos(py0, d);
    const int pyB0 = pyB_any & ~(TL - 1);
    const bool self_plane = (px == pxB);  // xk == 0 or xk == Hx/2
    // layout [px][z][py]: element (px, z, py) at ((px * L + z) * M + py); one float4 = lines (2 jp, 2 jp + 1)
    const float4* sA = reinterpret_cast<const float4*>(S + (size_t)px * L * M + py0);
    const float4* sB = reinterpret_cast<const float4*>(S + (size_t)pxB * L * M + pyB0);
    const int rowq = M / 2;
#pragma unroll MI_FFT_UNROLL
    for (int i = threadIdx.x; i < hp * L; i += kThreadsXZ) {
        const int z = i / hp, jp = i - z * hp;
        float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = a;
        if (z < d.z_in_hi) { a = sA[(size_t)z * rowq + jp]; b = sB[(size_t)z * rowq + jp]; }  // planes beyond: all zero, not stored
        const int cA = cell(2 * jp, pitch, hp, z), cB = cA + TL * pitch;  // rows TL + 2 jp carry the same mask
        tile[cA] = make_float2(a.x, a.y);
        tile[cA + pitch] = make_float2(a.z, a.w);
        tile[cB] = make_float2(b.x, b.y);
        tile[cB + pitch] = make_float2(b.z, b.w);
    }
    const float4* Gp = G + ((size_t)plane * M + py0) * L;
    using TW = TwLds<LZ2, R3>;
    float2* twl = tile + 2 * TL * pitch;
    TW::template fill<kThreadsXZ>(twl, tw);
    lds_barrier();
    const bool priv = ((2 * TL) % NW) == 0;
    if (!(d.dbg & 1)) {
        if constexpr (R3 > 1) {
            radix3_stage<R3, false, kThreadsXZ>(tile, 2 * TL, pitch, hp, priv, 1 << LZ2, twl + TW::r3);
            stage_sync(priv);
        }
        lds_fft<LZ2, false, kThreadsXZ, R3>(tile, 2 * TL * R3, pitch, hp, priv, twl);
    }
    if (priv) lds_barrier();  // the point-wise step pairs rows of different owners
    // point-wise: element (line j, position pz) of A pairs with (line jB, position pzB) of B
    float sw, cw;
    sincospif(-2.0f * (float)xk / (float)(2 * Hx), &sw, &cw);  // w = exp(-2 pi i xk / Nx), Nx = 2 Hx
    const float2 w = make_float2(cw, sw);
    const int n_it = (TL * L + kThreadsXZ - 1) / kThreadsXZ;
#pragma unroll 1
    for (int q = 0; q < n_it; ++q) {
        const int i = threadIdx.x + q * kThreadsXZ;
        if (i >= TL * L || (d.dbg & 2)) break;
        float4 g = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if constexpr (!BUILD) g = Gp[i];  // Gp[(size_t)j * L + pz] with i = j * L + pz
        int j = i / L;
        if (L % 64 == 0) j = __builtin_amdgcn_readfirstlane(j);  // a wave's 64 items share the line: scalar mirror math
        const int pz = i - j * L;
        const int jB = y_mirror_pos(py0 + j, d) - pyB0;
        const int pzB = mirror_pos(pz, L, LZ2, R3);
        const int cA = cell(j, pitch, hp, pz), cB = cell(TL + jB, pitch, hp, pzB);
        const float2 a = tile[cA];
        const float2 bm = tile[cB];
        const float2 bc = cconj(bm);
        const float2 E = make_float2(0.5f * (a.x + bc.x), 0.5f * (a.y + bc.y));
        const float2 dlt = csub(a, bc);                          // a - conj(b)
        const float2 O = make_float2(0.5f * dlt.y, -0.5f * dlt.x);  // -i/2 * (a - conj(b))
        const float2 wO = cmul(w, O);
        const float2 Xa = cadd(E, wO), Xb = csub(E, wO);
        if constexpr (BUILD) {
            Gout[((size_t)plane * M + py0) * L + i] = make_float4(Xa.x * scale, Xa.y * scale, Xb.x * scale, Xb.y * scale);
            continue;
        }
        float2 Ga = make_float2(g.x, g.y), Gb = make_float2(g.z, g.w);
        if (conj_otf) { Ga.y = -Ga.y; Gb.y = -Gb.y; }
        const float2 Ya = cmul(Xa, Ga), Yb = cmul(Xb, Gb);
        const float2 E2 = make_float2(0.5f * (Ya.x + Yb.x), 0.5f * (Ya.y + Yb.y));
        const float2 dY = csub(Ya, Yb);
        const float2 O2 = cmulc(make_float2(0.5f * dY.x, 0.5f * dY.y), w);  // (Ya - Yb) conj(w) / 2
        // Z'[k] = E' + i O' ; Z'[-k] = conj(E') + i conj(O')
        tile[cA] = make_float2(E2.x - O2.y, E2.y + O2.x);
        tile[cB] = make_float2(E2.x + O2.y, O2.x - E2.y);
    }
    if constexpr (BUILD) return;
    lds_barrier();
    if (!(d.dbg & 4)) {
        lds_fft<LZ2, true, kThreadsXZ, R3>(tile, 2 * TL * R3, pitch, hp, priv, twl);
        if constexpr (R3 > 1) {
            radix3_stage<R3, true, kThreadsXZ>(tile, 2 * TL, pitch, hp, priv, 1 << LZ2, twl + TW::r3);
            stage_sync(priv);
        }
    }
    if (priv) lds_barrier();
    float4* dA = reinterpret_cast<float4*>(T + (size_t)px * L * M + py0);
    float4* dB = reinterpret_cast<float4*>(T + (size_t)pxB * L * M + pyB0);
#pragma unroll MI_FFT_UNROLL
    for (int i = threadIdx.x; i < hp * L; i += kThreadsXZ) {
        const int z = i / hp, jp = i - z * hp;
        if (z < d.z_out_lo || z >= d.z_out_hi) continue;  // planes the crop drops
        const int cA = cell(2 * jp, pitch, hp, z), cB = cA + TL * pitch;
        const float2 a0 = tile[cA], a1 = tile[cA + pitch];
        dA[(size_t)z * rowq + jp] = make_float4(a0.x, a0.y, a1.x, a1.y);
        if (!self_plane) {
            const float2 b0 = tile[cB], b1 = tile[cB + pitch];
            dB[(size_t)z * rowq + jp] = make_float4(b0.x, b0.y, b1.x, b1.y);
        }
    }
}

// ---------------------------------------------------------------------------------------------- P3, pipelined
// The z pass as a persistent kernel (one work-group per CU): the OTF of the current tile is requested before the forward
// transform and the next tile's lines before the inverse transform, both into registers, so HBM stays busy during the FFT
// phases; stores drain behind.
// REALG: the OTF of a PSF that is mirror-symmetric about its centre sample is a real function times the phase ramp of the
// centre's offset from the grid origin: G holds the two real factors of a pair (float2 instead of float4: 4 instead of 8 B per
// voxel of OTF traffic, a sixth of this pass) and the ramp exp(-2 pi i (kx dx/Fx + ky dy/Fy + kz dz/Fz)) is put back from three
// small per-axis tables (x and y: scalar loads, z: one look-up per lane and tile).
constexpr bool z_pipe_even(int L) {
    // the real form needs line-uniform phases per item: either the lines divide the work-group evenly, or every wave owns one
    // pair of lines (the WP layout of k_z_conv_pipe)
    return (L % 64 == 0) && ((z_tile_lines(L) * L) % kThreadsXZ == 0) && ((kThreadsXZ % L == 0) || z_tile_lines(L) == kThreadsXZ / 64);
}
struct RealOtf {
    const float2* g;     // [xk][py][pz] {Ra, Rb}
    const float2* ph_x;  // by xk
    const float2* ph_y;  // by ky
    const float2* ph_z;  // by kz
};

template <int LZ2, int R3, bool REALG>
__global__ __launch_bounds__(kThreadsXZ, kWavesXZ) void k_z_conv_pipe(const float2* __restrict__ S, float2* __restrict__ T, const float4* __restrict__ G,
                                                           NativeDims d, const float2* __restrict__ tw, int conj_otf, int ntiles, RealOtf ro) {
    extern __shared__ __attribute__((aligned(16))) float2 tile[];
    constexpr int L = R3 << LZ2, NW = kThreadsXZ / 64;
    constexpr int TL = z_tile_lines(L), hp = TL / 2, pitch = row_pitch(L);
    constexpr int NA = hp * L;                                   // float4 of the A lines (and of the B lines) of a tile
    constexpr int NPA = (NA + kThreadsXZ - 1) / kThreadsXZ;
    constexpr int NG = TL * L;                                   // OTF float4 of a tile = point-wise items
    constexpr int NPG = (NG + kThreadsXZ - 1) / kThreadsXZ;
    constexpr int P = kThreadsXZ / hp;                           // transposed view: item k of a lane is position z0 + k * P
    constexpr bool PRIV = ((2 * TL) % NW) == 0;
    // WP: with one A line and one B line per wave, the B lines are stored so that LDS row TL + j holds the MIRROR PARTNER of A
    // line j -- both rows of a pair then belong to wave j and the point-wise step needs no work-group barrier either: forward
    // transforms, point-wise product and inverse transforms of a pair run back to back inside its wave (3 barriers per tile
    // instead of 5, all of them around the transposed fill and drain)
    constexpr bool WP = PRIV && TL == NW && (L % 64 == 0) && (NG % kThreadsXZ == 0);
    // point-wise view: item k of a lane is element pz0 of line j0 + k * JS when the lines divide the work-group evenly
    constexpr bool EVEN = (kThreadsXZ % L == 0) && (L % 64 == 0) && (NG % kThreadsXZ == 0) && ((kThreadsXZ / L) % 2 == 0 || kThreadsXZ == L);
    constexpr int JS = kThreadsXZ / (L > 0 ? L : 1);
    const int Hx = d.hx, M = d.ny;
    const int ytiles = M / TL, rowq = M / 2;
    // lane constants (tile-invariant; recomputed per phase from a laundered thread index so that they do not occupy registers
    // across the FFT phases): the swizzle is XOR-linear, so item k's slot is item 0's slot XOR a constant
    struct FView { int row, slot, z0, jp, pz; size_t off; };  // transposed view: item k = position z0 + k * P, line pair jp
    auto f_view = [&]() {
        const int tid = launder(threadIdx.x);
        const int z0 = tid / hp, jp = tid - z0 * hp;
        const int pz = phys(z0);
        return FView{(2 * jp) * pitch, pz ^ rmask(2 * jp, hp), z0, jp, pz, (size_t)z0 * rowq + jp};
    };
    // WP: A line index (0..TL-1) whose mirror partner is B line jb, 4 bits each (wave-uniform, recomputed per tile)
    auto partner_table = [&](const auto& w) {
        unsigned long long tab = 0;
        for (int jb = 0; jb < TL; ++jb) tab |= (unsigned long long)((y_mirror_pos(w.pyB0 + jb, d) - w.py0) & 15) << (4 * jb);
        return tab;
    };
    // WP: LDS cells of the B lines (2 jp, 2 jp + 1) at position slot `pzs` (unmasked)
    auto b_cells = [&](unsigned long long tab, int jp, int pzs, int& c0, int& c1) {
        const int r0 = TL + (int)((tab >> (8 * jp)) & 15), r1 = TL + (int)((tab >> (8 * jp + 4)) & 15);
        c0 = r0 * pitch + (pzs ^ rmask(r0, hp));
        c1 = r1 * pitch + (pzs ^ rmask(r1, hp));
    };
    float4 preA[NPA], preB[NPA];
    struct Where { int plane, py0, px, pxB, pyB0; };
    auto where = [&](int t) {
        Where w;
        w.plane = t / ytiles;
        w.py0 = (t - w.plane * ytiles) * TL;
        w.px = freq2pos(w.plane, d.lhx2, d.r3x);
        w.pxB = freq2pos(w.plane == 0 ? 0 : Hx - w.plane, d.lhx2, d.r3x);
        w.pyB0 = y_mirror_pos(w.py0, d) & ~(TL - 1);
        return w;
    };
    auto load_S = [&](int t) {
        const Where w = where(t);
        const FView fv = f_view();
        const float4* sA = reinterpret_cast<const float4*>(S + (size_t)w.px * L * M + w.py0) + fv.off;
        const float4* sB = reinterpret_cast<const float4*>(S + (size_t)w.pxB * L * M + w.pyB0) + fv.off;
#pragma unroll
        for (int k = 0; k < NPA; ++k) {
            if (NA % kThreadsXZ == 0 || (int)threadIdx.x + k * kThreadsXZ < NA) {
                float4 va = make_float4(0.0f, 0.0f, 0.0f, 0.0f), vb = va;
                if (fv.z0 + k * P < d.z_in_hi) {  // planes beyond: all-zero input of a padded grid, never stored
                    va = sA[(size_t)(k * P) * rowq];
                    vb = sB[(size_t)(k * P) * rowq];
                }
                preA[k] = va;
                preB[k] = vb;
            }
        }
    };
    using TW = TwLds<LZ2, R3>;
    float2* twl = tile + 2 * TL * pitch;
    TW::template fill<kThreadsXZ>(twl, tw);
    int t = blockIdx.x;
    if (t < ntiles) load_S(t);
    for (; t < ntiles; t += gridDim.x) {
        const Where w = where(t);
        unsigned long long ptab = 0;
        if constexpr (WP) ptab = partner_table(w);
        {
        const FView fv = f_view();
#pragma unroll
        for (int k = 0; k < NPA; ++k) {
            if (NA % kThreadsXZ == 0 || (int)threadIdx.x + k * kThreadsXZ < NA) {
                const int cA = fv.row + (fv.slot ^ swz_c(k * P));
                int cB0 = cA + TL * pitch, cB1 = cB0 + pitch;  // rows TL + 2 jp (+1) carry the same mask
                if constexpr (WP) b_cells(ptab, fv.jp, fv.pz ^ swz_c(k * P), cB0, cB1);
                tile[cA] = make_float2(preA[k].x, preA[k].y);
                tile[cA + pitch] = make_float2(preA[k].z, preA[k].w);
                tile[cB0] = make_float2(preB[k].x, preB[k].y);
                tile[cB1] = make_float2(preB[k].z, preB[k].w);
            }
        }
        }
        const size_t g0 = ((size_t)w.plane * M + w.py0) * L;
        float4 gv[REALG ? 1 : NPG];
        float2 gr[REALG ? NPG : 1];
        float2 ph_xz = make_float2(1.0f, 0.0f), ph_yk[REALG ? NPG : 1];
        auto load_G = [&]() {
            // REALG: phase of (this tile's xk) x (this lane's kz), and of the ky of the line of every item (wave-uniform); requested
            // here, together with the OTF, so that they have arrived long before the point-wise step
            // (WP: item k of a lane is position lane + 64 k of its wave's line: one ky per wave, one kz per item)
            if constexpr (REALG) {
                const int tid = launder(threadIdx.x);
                if constexpr (WP) {
                    ph_xz = cmul(ro.ph_x[w.plane], ro.ph_y[y_pos2freq(w.py0 + __builtin_amdgcn_readfirstlane(tid >> 6), d)]);
    #pragma unroll
                    for (int k = 0; k < NPG; ++k) ph_yk[k] = ro.ph_z[pos2freq((tid & 63) + 64 * k, LZ2, R3)];
                } else {
                    ph_xz = cmul(ro.ph_x[w.plane], ro.ph_z[pos2freq(tid % L, LZ2, R3)]);
                    const int j0e = __builtin_amdgcn_readfirstlane(tid / L);
    #pragma unroll
                    for (int k = 0; k < NPG; ++k) ph_yk[k] = ro.ph_y[y_pos2freq(w.py0 + j0e + k * JS, d)];
                }
            }
            {
                const int tid = launder(threadIdx.x);
                // WP: the OTF entries of line `wave`, positions lane + 64 k
                const size_t gl = WP ? g0 + (size_t)(tid >> 6) * L + (tid & 63) : g0 + tid;
    #pragma unroll
                for (int k = 0; k < NPG; ++k) {
                    if (NG % kThreadsXZ == 0 || tid + k * kThreadsXZ < NG) {
                        if constexpr (REALG) gr[k] = ro.g[gl + (WP ? 64 : kThreadsXZ) * k];
                        else gv[k] = G[gl + (WP ? 64 : kThreadsXZ) * k];
                    }
                }
            }
        };
        // (a sixteen-point top stage -- the FIRST of the forward transform -- and the OTF registers do not fit 128 registers together:
        // the OTF is then requested behind that stage)
        constexpr bool LATE_G = R3 == 1 && LZ2 - seg_below(LZ2, LZ2) == 4;
        if (R3 != 9 && !LATE_G) load_G();  // (radix-9 lines: requested behind the 9-point stage, which needs the registers)
        lds_barrier();
        if constexpr (R3 > 1) {
            radix3_stage<R3, false, kThreadsXZ>(tile, 2 * TL, pitch, hp, PRIV, 1 << LZ2, twl + TW::r3);
            stage_sync(PRIV);
        }
        if (R3 == 9) load_G();
        if constexpr (LATE_G) {
            lds_fft<LZ2, false, kThreadsXZ, R3, 0, 4>(tile, 2 * TL * R3, pitch, hp, PRIV, twl);
            load_G();
            lds_fft<LZ2, false, kThreadsXZ, R3, 4, LZ2>(tile, 2 * TL * R3, pitch, hp, PRIV, twl);
        } else {
            lds_fft<LZ2, false, kThreadsXZ, R3>(tile, 2 * TL * R3, pitch, hp, PRIV, twl);
        }
        if (PRIV && !WP) lds_barrier();  // the point-wise step pairs rows of different owners
        float sw, cw;
        sincospif(-2.0f * (float)w.plane / (float)(2 * Hx), &sw, &cw);  // exp(-2 pi i xk / Nx), Nx = 2 Hx
        const float2 wx = make_float2(cw, sw);
        const int tid = launder(threadIdx.x);
        const int pz0 = tid % L, j0 = __builtin_amdgcn_readfirstlane(tid / L);
        const int pA = phys(WP ? (tid & 63) : pz0), pB = phys(mirror_pos(pz0, L, LZ2, R3));
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int k = 0; k < NPG; ++k) {
            const int i = tid + k * kThreadsXZ;
            if (NG % kThreadsXZ == 0 || i < NG) {
                int cA, cB;
                if constexpr (WP) {  // lines A[wave] and its partner in row TL + wave, position lane + 64 k
                    cA = wv * pitch + (pA ^ swz_c(64 * k) ^ rmask(wv, hp));
                    cB = (TL + wv) * pitch + (phys(mirror_pos((tid & 63) + 64 * k, L, LZ2, R3)) ^ rmask(TL + wv, hp));
                } else if (EVEN) {
                    const int j = j0 + k * JS;                                // scalar: the line is shared by the wave
                    const int jB = y_mirror_pos(w.py0 + j, d) - w.pyB0;       // scalar mirror math
                    cA = j * pitch + (pA ^ rmask(j, hp));
                    cB = (TL + jB) * pitch + (pB ^ rmask(TL + jB, hp));
                } else {
                    int j = i / L;
                    if (L % 64 == 0) j = __builtin_amdgcn_readfirstlane(j);
                    const int pz = i - j * L;
                    const int jB = y_mirror_pos(w.py0 + j, d) - w.pyB0;
                    cA = cell(j, pitch, hp, pz);
                    cB = cell(TL + jB, pitch, hp, mirror_pos(pz, L, LZ2, R3));
                }
                const float2 a = tile[cA];
                const float2 bc = cconj(tile[cB]);
                const float2 E = make_float2(0.5f * (a.x + bc.x), 0.5f * (a.y + bc.y));
                const float2 dlt = csub(a, bc);
                const float2 O = make_float2(0.5f * dlt.y, -0.5f * dlt.x);  // -i/2 * (a - conj(b))
                const float2 wO = cmul(wx, O);
                const float2 Xa = cadd(E, wO), Xb = csub(E, wO);
                float2 Ya, Yb;
                if constexpr (REALG) {
                    float2 P = cmul(ph_xz, ph_yk[k]);                         // (REALG requires the EVEN item layout)
                    if (conj_otf) P.y = -P.y;
                    const float2 XaP = cmul(Xa, P), XbP = cmul(Xb, P);
                    Ya = make_float2(XaP.x * gr[k].x, XaP.y * gr[k].x);
                    Yb = make_float2(XbP.x * gr[k].y, XbP.y * gr[k].y);
                } else {
                    float2 Ga = make_float2(gv[k].x, gv[k].y), Gb = make_float2(gv[k].z, gv[k].w);
                    if (conj_otf) { Ga.y = -Ga.y; Gb.y = -Gb.y; }
                    Ya = cmul(Xa, Ga);
                    Yb = cmul(Xb, Gb);
                }
                const float2 E2 = make_float2(0.5f * (Ya.x + Yb.x), 0.5f * (Ya.y + Yb.y));
                const float2 dY = csub(Ya, Yb);
                const float2 O2 = cmulc(make_float2(0.5f * dY.x, 0.5f * dY.y), wx);
                tile[cA] = make_float2(E2.x - O2.y, E2.y + O2.x);
                tile[cB] = make_float2(E2.x + O2.y, O2.x - E2.y);
            }
        }
        const int tn = t + gridDim.x;
        if (R3 != 9 && tn < ntiles) load_S(tn);
        if (WP) wave_lds_fence();
        else lds_barrier();
        lds_fft<LZ2, true, kThreadsXZ, R3>(tile, 2 * TL * R3, pitch, hp, PRIV, twl);
        if constexpr (R3 > 1) {
            radix3_stage<R3, true, kThreadsXZ>(tile, 2 * TL, pitch, hp, PRIV, 1 << LZ2, twl + TW::r3);
            stage_sync(PRIV);
        }
        if (R3 == 9 && tn < ntiles) load_S(tn);
        if (PRIV) lds_barrier();
        const bool self_plane = (w.px == w.pxB);
        const FView fv = f_view();
        float4* dA = reinterpret_cast<float4*>(T + (size_t)w.px * L * M + w.py0) + fv.off;
        float4* dB = reinterpret_cast<float4*>(T + (size_t)w.pxB * L * M + w.pyB0) + fv.off;
#pragma unroll
        for (int k = 0; k < NPA; ++k) {
            const int zk = fv.z0 + k * P;
            if ((NA % kThreadsXZ == 0 || (int)threadIdx.x + k * kThreadsXZ < NA) && zk >= d.z_out_lo && zk < d.z_out_hi) {
                const int cA = fv.row + (fv.slot ^ swz_c(k * P));
                int cB0 = cA + TL * pitch, cB1 = cB0 + pitch;
                if constexpr (WP) b_cells(ptab, fv.jp, fv.pz ^ swz_c(k * P), cB0, cB1);
                const float2 a0 = tile[cA], a1 = tile[cA + pitch];
                dA[(size_t)(k * P) * rowq] = make_float4(a0.x, a0.y, a1.x, a1.y);
                if (!self_plane) {
                    const float2 b0 = tile[cB0], b1 = tile[cB1];
                    dB[(size_t)(k * P) * rowq] = make_float4(b0.x, b0.y, b1.x, b1.y);
                }
            }
        }
        lds_barrier();  // the tile is free for the next fill
    }
}

// ---------------------------------------------------------------------------------------------- P3, pair-interleaved layout
// The spectra around the z pass as [xk][z][ty][side][TL]: the TL A lines of a tile and, right behind them, their TL mirror
// partners (in partner order), so that a tile of only TL = 8 line pairs still moves whole 128-byte segments and two 8-wave
// work-groups with a 64-KB tile each share a CU: one transforms while the other waits for HBM.  Replaces the same chain as
// k_z_conv_pipe (decon.m:162-172: the z part of fftn, .* otf, the z part of ifftn); same OTF array, same point-wise step.
//   NT = 512 (lines of up to 576 points): a wave owns one A line and its partner -- forward transform, point-wise step and
//     inverse transform of the pair run inside the wave, the only work-group barriers surround the transposed fill and drain;
//     for 2^a lines of 256 / 512 points the fill and the drain ARE the top super-stage (on the registers of the global access).
//   NT = 1024 (768, 1152 points): a wave owns one line; the point-wise step sits between two barriers.  (1024-point lines run on
//     NT = 512 with 246 registers: see the launch.)
//   Lines of 3 * 2^a / 9 * 2^a points carry the radix-3 / 9 stage in front (behind, inverse) of the power-of-two chain.
template <int LZ2, int R3, bool REALG, int NT, int TL, bool PHL = true, bool TOPON = true>
__global__ __launch_bounds__(NT, (NT == 512 && (R3 << LZ2) > 576) ? 2 : kWavesXZ) void k_z_pair_pipe(const float2* __restrict__ S, float2* __restrict__ T, const float4* __restrict__ G,
                                                              NativeDims d, const float2* __restrict__ tw, int conj_otf, int ntiles, RealOtf ro,
                                                              int* __restrict__ tile_ctr) {
    extern __shared__ __attribute__((aligned(16))) float2 tile[];
    __shared__ int s_next_tile;  // tiles from a device counter when tile_ctr != nullptr (see k_x_fused_pipe)
    constexpr int L = R3 << LZ2, NW = NT / 64, hp = TL, pitch = row_pitch(L);
    // WP: every wave owns one A line and its partner (rows wave, TL + wave), so the point-wise step is wave-private too;
    // else (1024-point lines: 16 waves on 16 rows) a wave owns ONE row, the point-wise step of line `wave % TL` is shared by the
    // owners of its two rows -- half of the positions each -- and sits between two work-group barriers
    constexpr bool WP = TL == NW;
    static_assert((WP || (NW == 2 * TL && L % 128 == 0)) && L % 64 == 0 && (TL * L) % NT == 0, "one or two waves per line pair");
    constexpr int NPA = TL * L / NT;  // float4 (two neighbouring lines at one z) per lane and tile
    constexpr int P = NT / TL;        // item k of a lane: position z0 + k * P
    constexpr int NPG = TL * L / NT;  // point-wise items (mirror pairs) per lane
    // the top super-stage of the chain (stages TOPS .. LZ2-1) works on elements z0 + k * 2^TOPS: exactly the items of a lane
    constexpr int TOPS = seg_below(LZ2, LZ2), TOPR = LZ2 - TOPS;
    constexpr bool TOPREG = TOPON && R3 == 1 && (1 << TOPS) == P && (1 << TOPR) == NPA;
    const int Hx = d.hx, M = d.ny, ytiles = M / TL;
    const size_t ZR = (size_t)(M + d.zpad);  // float4 per row (xk, z) of the paired layout
    struct FView { int row, slot, z0; size_t off; };
    auto f_view = [&]() {
        const int tid = launder(threadIdx.x);
        const int z0 = tid / TL, jq = tid - z0 * TL;  // float4 jq of the segment: lines 2 jq, 2 jq + 1 (rows TL.. = B side)
        return FView{(2 * jq) * pitch, phys(z0) ^ rmask(2 * jq, hp), z0, (size_t)z0 * ZR + jq};
    };
    float4 pre[NPA];
    auto load_S = [&](int t) {
        const int pl = t / ytiles, ty = t - pl * ytiles, plane = d.xk0 + pl;  // (tiles of the planes xk0 ..: all, or a chunk)
        const FView fv = f_view();
        const float4* sp = reinterpret_cast<const float4*>(S) + (size_t)plane * L * ZR + (size_t)ty * TL + fv.off;
#pragma unroll
        for (int k = 0; k < NPA; ++k) pre[k] = (fv.z0 + k * P < d.z_in_hi) ? sp[(size_t)(k * P) * ZR] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    };
    using TW = TwLds<LZ2, R3>;
    float2* twl = tile + 2 * TL * pitch;
    TW::template fill<NT>(twl, tw);
    // REALG: the z ramp by POSITION, behind the twiddle tables (stride-1 look-ups in the point-wise step; PHL = false when that
    // table would cost the second work-group of the CU: the ramp then comes from global memory, by frequency)
    float2* phl = twl + TW::total;
    if constexpr (REALG && PHL) {
        for (int p = threadIdx.x; p < L; p += NT) phl[p] = ro.ph_z[pos2freq(p, LZ2, R3)];
    }
    const bool dyn = tile_ctr != nullptr;
    int t = blockIdx.x, tn = t + (int)gridDim.x;
    if (dyn) {
        if (threadIdx.x == 0) s_next_tile = atomicAdd(tile_ctr, 2);
        lds_barrier();
        t = __builtin_amdgcn_readfirstlane(s_next_tile);
        tn = t + 1;
    }
    if (t < ntiles) load_S(t);
    lds_barrier();  // the tables: the first tile's top super-stage reads them before any other barrier
    for (; t < ntiles;) {
        int fetched = 0;
        if (dyn && threadIdx.x == 0) fetched = atomicAdd(tile_ctr, 1);  // the tile after the next one
        const int pl_ = t / ytiles, py0 = (t - pl_ * ytiles) * TL, plane = d.xk0 + pl_;
        {
            const FView fv = f_view();
            if constexpr (TOPREG) {  // the top super-stage on the registers the loads arrived in
                float2 v[NPA], u[NPA];
#pragma unroll
                for (int k = 0; k < NPA; ++k) { v[k] = make_float2(pre[k].x, pre[k].y); u[k] = make_float2(pre[k].z, pre[k].w); }
                butterflies<TOPR, TOPS, false>(v, twl + tw_off(LZ2, TOPS), fv.z0);
                butterflies<TOPR, TOPS, false>(u, twl + tw_off(LZ2, TOPS), fv.z0);
#pragma unroll
                for (int k = 0; k < NPA; ++k) {
                    const int c = fv.row + (fv.slot ^ swz_c(k * P));
                    tile[c] = v[k];
                    tile[c + pitch] = u[k];
                }
            } else {
#pragma unroll
            for (int k = 0; k < NPA; ++k) {
                const int c = fv.row + (fv.slot ^ swz_c(k * P));
                tile[c] = make_float2(pre[k].x, pre[k].y);
                tile[c + pitch] = make_float2(pre[k].z, pre[k].w);
            }
            }
        }
        const size_t g0 = ((size_t)plane * M + py0) * L;
        float4 gv[REALG ? 1 : NPG];
        float2 gr[REALG ? NPG : 1];
        float2 ph_xy = make_float2(1.0f, 0.0f);
        auto load_G = [&]() {
            const int tid = launder(threadIdx.x);
            const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
            const int line = WP ? wv : wv % TL, p0 = WP ? 0 : (wv / TL) * (L / 2);
            if constexpr (REALG) {
                ph_xy = cmul(ro.ph_x[plane], ro.ph_y[y_pos2freq(py0 + line, d)]);
            }
            const size_t gl = g0 + (size_t)line * L + p0 + (tid & 63);
#pragma unroll
            for (int k = 0; k < NPG; ++k) {
                if constexpr (REALG) gr[k] = ro.g[gl + 64 * k];
                else gv[k] = G[gl + 64 * k];
            }
        };
        // (complex OTF and a sixteen-point top stage, the first of the forward transform: see k_z_conv_pipe)
        constexpr bool LATE_G = !REALG && R3 == 1 && !TOPREG && LZ2 - seg_below(LZ2, LZ2) == 4;
        if (R3 != 9 && !LATE_G) load_G();  // (radix-9 lines: requested behind the 9-point stage, which needs the registers)
        lds_barrier();
        if constexpr (R3 > 1) {
            radix3_stage<R3, false, NT>(tile, 2 * TL, pitch, hp, true, 1 << LZ2, twl + TW::r3);
            wave_lds_fence();
        }
        if (R3 == 9) load_G();
        if constexpr (LATE_G) {
            lds_fft<LZ2, false, NT, R3, 0, 4>(tile, 2 * TL * R3, pitch, hp, true, twl);
            load_G();
            lds_fft<LZ2, false, NT, R3, 4, LZ2>(tile, 2 * TL * R3, pitch, hp, true, twl);
        } else {
            lds_fft<LZ2, false, NT, R3, TOPREG ? TOPR : 0>(tile, 2 * TL * R3, pitch, hp, true, twl);
        }
        if constexpr (!WP) lds_barrier();
        float sw, cw;
        sincospif(-2.0f * (float)plane / (float)(2 * Hx), &sw, &cw);  // exp(-2 pi i xk / Nx), Nx = 2 Hx
        const float2 wx = make_float2(cw, sw);
        {
            const int tid = launder(threadIdx.x);
            const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
            // item k: position lane + 64 k of the wave's A line and its mirror in the partner line.  The mirror of a position
            // whose high bits 64 k are not zero is (lane ^ 63) + [mirror of the high bits alone], else the mirror of `lane` among
            // the first 64 positions: lane constants XOR compile-time numbers, like every slot here
            const int lane = tid & 63;
            const int line = WP ? wv : wv % TL, p0 = WP ? 0 : (wv / TL) * (L / 2);  // (scalar)
            const int pA = phys(lane) ^ rmask(line, hp), pB1 = phys(lane ^ 63) ^ rmask(TL + line, hp);
            const int pB0 = phys(mirror_pos(lane, L, LZ2, R3)) ^ rmask(TL + line, hp);
#pragma unroll
            for (int k = 0; k < NPG; ++k) {
                const int hb = 64 * k + p0;  // the position bits above the lane's six
                const int cA = line * pitch + (pA ^ swz_c(hb));
                int cB;
                if constexpr (R3 == 1) {
                    const int flo = (int)brev_n((unsigned)hb, LZ2);
                    const int mhi = flo ? (int)brev_n((unsigned)((1 << (LZ2 - 6)) - flo), LZ2) : 0;
                    cB = (TL + line) * pitch + (flo ? (pB1 ^ swz_c(mhi)) : pB0);
                } else {  // (3 * 2^a, 9 * 2^a: the mirror map is not XOR-linear)
                    cB = (TL + line) * pitch + (phys(mirror_pos(lane + hb, L, LZ2, R3)) ^ rmask(TL + line, hp));
                }
                const float2 a = tile[cA];
                const float2 bc = cconj(tile[cB]);
                const float2 E = make_float2(0.5f * (a.x + bc.x), 0.5f * (a.y + bc.y));
                const float2 dlt = csub(a, bc);
                const float2 O = make_float2(0.5f * dlt.y, -0.5f * dlt.x);
                const float2 wO = cmul(wx, O);
                const float2 Xa = cadd(E, wO), Xb = csub(E, wO);
                float2 Ya, Yb;
                if constexpr (REALG) {
                    float2 Pq = cmul(ph_xy, PHL ? phl[lane + hb] : ro.ph_z[pos2freq(lane + hb, LZ2, R3)]);
                    if (conj_otf) Pq.y = -Pq.y;
                    const float2 XaP = cmul(Xa, Pq), XbP = cmul(Xb, Pq);
                    Ya = make_float2(XaP.x * gr[k].x, XaP.y * gr[k].x);
                    Yb = make_float2(XbP.x * gr[k].y, XbP.y * gr[k].y);
                } else {
                    float2 Ga = make_float2(gv[k].x, gv[k].y), Gb = make_float2(gv[k].z, gv[k].w);
                    if (conj_otf) { Ga.y = -Ga.y; Gb.y = -Gb.y; }
                    Ya = cmul(Xa, Ga);
                    Yb = cmul(Xb, Gb);
                }
                const float2 E2 = make_float2(0.5f * (Ya.x + Yb.x), 0.5f * (Ya.y + Yb.y));
                const float2 dY = csub(Ya, Yb);
                const float2 O2 = cmulc(make_float2(0.5f * dY.x, 0.5f * dY.y), wx);
                tile[cA] = make_float2(E2.x - O2.y, E2.y + O2.x);
                tile[cB] = make_float2(E2.x + O2.y, O2.x - E2.y);
            }
        }
        // (requesting them right after the fill, a whole tile ahead, gains nothing with two work-groups per CU: 4.72 vs 4.68 ms; with
        // the one 1024-thread work-group of 1024-point lines it LOSES -- 6.28 against 5.90 ms on 1024 x 576 x 4096, A / B in one
        // process, the loads unconditional and behind the OTF loads so that every wait stays counted; a 512-thread variant with a
        // line pair per wave measured 6.00 with the 4 x 4 top stages -- with the 16-point top stage on the registers of the global
        // access it became the kept form, see the launch: round 4, profiles/zpass_ab.py)
        if (R3 != 9 && tn < ntiles) load_S(tn);
        if constexpr (WP) wave_lds_fence();
        else lds_barrier();
        lds_fft<LZ2, true, NT, R3, 0, TOPREG ? TOPS : LZ2>(tile, 2 * TL * R3, pitch, hp, true, twl);
        if constexpr (R3 > 1) {
            radix3_stage<R3, true, NT>(tile, 2 * TL, pitch, hp, true, 1 << LZ2, twl + TW::r3);
            wave_lds_fence();
        }
        if (R3 == 9 && tn < ntiles) load_S(tn);
        lds_barrier();
        {
            const FView fv = f_view();
            float4* dp = reinterpret_cast<float4*>(T) + (size_t)plane * L * ZR + (size_t)(py0 / TL) * TL + fv.off;
#pragma unroll
            for (int k = 0; k < NPA; ++k) {
                const int zk = fv.z0 + k * P;
                if constexpr (!TOPREG) {
                if (zk >= d.z_out_lo && zk < d.z_out_hi) {
                    const int c = fv.row + (fv.slot ^ swz_c(k * P));
                    const float2 a0 = tile[c], a1 = tile[c + pitch];
                    dp[(size_t)(k * P) * ZR] = make_float4(a0.x, a0.y, a1.x, a1.y);
                }
                }
            }
            if constexpr (TOPREG) {
                float2 v[NPA], u[NPA];
#pragma unroll
                for (int k = 0; k < NPA; ++k) {
                    const int c = fv.row + (fv.slot ^ swz_c(k * P));
                    v[k] = tile[c];
                    u[k] = tile[c + pitch];
                }
                butterflies<TOPR, TOPS, true>(v, twl + tw_off(LZ2, TOPS), fv.z0);
                butterflies<TOPR, TOPS, true>(u, twl + tw_off(LZ2, TOPS), fv.z0);
#pragma unroll
                for (int k = 0; k < NPA; ++k) {
                    const int zk = fv.z0 + k * P;
                    if (zk >= d.z_out_lo && zk < d.z_out_hi) dp[(size_t)(k * P) * ZR] = make_float4(v[k].x, v[k].y, u[k].x, u[k].y);
                }
            }
        }
        if (dyn && threadIdx.x == 0) s_next_tile = fetched;
        lds_barrier();
        t = tn;
        tn = dyn ? __builtin_amdgcn_readfirstlane(s_next_tile) : tn + (int)gridDim.x;
    }
}

// ---------------------------------------------------------------------------------------------- P5: x inverse + epilogue
// FUSE: the epilogue result stays in LDS and is transformed forward again into S_next (the P1 of the NEXT
// convolution): the ratio never touches HBM, and bl is read once and written once per iteration.
template <int LHX2, int R3, bool FUSE>
__global__ __launch_bounds__(kThreadsXZ, kWavesXZ) void k_x_inverse(const float2* __restrict__ T, float* __restrict__ out, ConvEpilogue e, NativeDims d,
                                                         const float2* __restrict__ tw, float2* __restrict__ S_next, int EPI, PadWindow pw) {
    extern __shared__ __attribute__((aligned(16))) float2 tile[];
    constexpr int Hx = R3 << LHX2, NW = kThreadsXZ / 64;
    const int TY = d.ty, hp = TY / 2, pitch = row_pitch(Hx);
    const int ytiles = d.ny / TY;
    const int z = blockIdx.x / ytiles, y0 = (blockIdx.x % ytiles) * TY;
    const float4* src = reinterpret_cast<const float4*>(T + ((size_t)z * Hx) * d.xrow + y0);
    const int rowq = d.xrow / 2;
    int oz = z;
    if (pw.on) {
        // rows outside the cropped result are never stored: a tile without any is skipped (fused: its part of the next
        // convolution's input is the zero padding)
        oz = pad_dst(pw, 2, z);
        bool live = false;
        for (int r = 0; r < TY; ++r) live = live || pad_dst(pw, 1, y0 + r) >= 0;
        if (oz < 0 || !live) {
            if (FUSE && z < d.z_in_hi) {  // planes beyond are never read by the next y pass
                float4* sdst = reinterpret_cast<float4*>(S_next + ((size_t)z * Hx) * d.xrow + y0);
                for (int i = threadIdx.x; i < hp * Hx; i += kThreadsXZ) {
                    const int px = i / hp, rp = i - px * hp;
                    sdst[(size_t)px * rowq + rp] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
            }
            return;
        }
    }
#pragma unroll MI_FFT_UNROLL
    for (int i = threadIdx.x; i < hp * Hx; i += kThreadsXZ) {
        const int px = i / hp, rp = i - px * hp;
        const float4 v = src[(size_t)px * rowq + rp];
        const int c0 = cell(2 * rp, pitch, hp, px);
        tile[c0] = make_float2(v.x, v.y);
        tile[c0 + pitch] = make_float2(v.z, v.w);
    }
    using TW = TwLds<LHX2, R3>;
    float2* twl = tile + TY * pitch;
    TW::template fill<kThreadsXZ>(twl, tw);
    lds_barrier();
    // rows dealt to the waves: the inverse transform, the epilogue and the forward transform of a row all belong to its owner
    const bool priv = (TY % NW) == 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (!(d.dbg & 8)) {
        lds_fft<LHX2, true, kThreadsXZ, R3>(tile, TY * R3, pitch, hp, priv, twl);
        if constexpr (R3 > 1) {
            radix3_stage<R3, true, kThreadsXZ>(tile, TY, pitch, hp, priv, 1 << LHX2, twl + TW::r3);
            stage_sync(priv);
        }
    }
    if (pw.on) {
        // crop + epilogue on the caller's (unpadded) volume; fused: the zero padding of the next input is re-created
        const float l = e.lambda, m = 1.0f - e.lambda;
        const int n_items = priv ? (TY / NW) * Hx : TY * Hx;
        for (int i = priv ? lane : threadIdx.x; i < n_items; i += priv ? 64 : kThreadsXZ) {
            const int rl = i / Hx, q = i - rl * Hx;
            const int r = priv ? rl * NW + wave : rl;
            const int oy = pad_dst(pw, 1, y0 + r);
            float2* cl = tile + cell(r, pitch, hp, q);
            const float2 c = *cl;
            float2 o = make_float2(0.0f, 0.0f);
            if (oy >= 0) {
                const size_t rbase = ((size_t)oz * pw.n[1] + oy) * (size_t)pw.n[0];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int ox = pad_dst(pw, 0, 2 * q + h);
                    if (ox < 0) continue;
                    const float cv = h ? c.y : c.x;
                    const size_t gi = rbase + ox;
                    float v;
                    if (EPI == EPI_NONE) v = cv;
                    else if (EPI == EPI_RATIO) v = e.a[gi] * rcp_eps(cv);
                    else if (EPI == EPI_UPDATE) v = fabsf(e.a[gi] * cv);
                    else v = fabsf(e.a[gi] * cv * m + e.b[gi] * l);
                    if (!FUSE || out != nullptr) out[gi] = v;
                    if (h) o.y = v; else o.x = v;
                }
            }
            if (FUSE) *cl = o;
        }
    } else {
        const size_t row0 = ((size_t)z * d.ny + y0) * (size_t)(2 * Hx);
        const int quads = Hx / 2;
        float4* dst = reinterpret_cast<float4*>(out + row0);
        const float4* a4 = reinterpret_cast<const float4*>(e.a + row0);
        const float4* b4 = reinterpret_cast<const float4*>(e.b + row0);
        const int n_items = priv ? (TY / NW) * quads : TY * quads;
        for (int i = priv ? lane : threadIdx.x; i < n_items; i += priv ? 64 : kThreadsXZ) {
            const int rl = i / quads, q = i - rl * quads;
            const int r = priv ? rl * NW + wave : rl;
            const int c0i = cell(r, pitch, hp, 2 * q);
            const float2 c0 = tile[c0i], c1 = tile[c0i ^ 1];
            float4 c = make_float4(c0.x, c0.y, c1.x, c1.y), o;
            const size_t gi = (size_t)r * quads + q;
            if (EPI == EPI_NONE) {
                o = c;
            } else {
                const float4 av = a4[gi];
                if (EPI == EPI_RATIO) {
                    o = make_float4(av.x * rcp_eps(c.x), av.y * rcp_eps(c.y), av.z * rcp_eps(c.z), av.w * rcp_eps(c.w));
                } else if (EPI == EPI_UPDATE) {
                    o = make_float4(fabsf(av.x * c.x), fabsf(av.y * c.y), fabsf(av.z * c.z), fabsf(av.w * c.w));
                } else {
                    const float4 bv = b4[gi];
                    const float l = e.lambda, m = 1.0f - e.lambda;
                    o = make_float4(fabsf(av.x * c.x * m + bv.x * l), fabsf(av.y * c.y * m + bv.y * l), fabsf(av.z * c.z * m + bv.z * l),
                                    fabsf(av.w * c.w * m + bv.w * l));
                }
            }
            if (!FUSE || out != nullptr) dst[gi] = o;
            if (FUSE) {
                tile[c0i] = make_float2(o.x, o.y);
                tile[c0i ^ 1] = make_float2(o.z, o.w);
            }
        }
    }
    if (FUSE) {
        stage_sync(priv);
        if (!(d.dbg & 16)) {
            if constexpr (R3 > 1) {
                radix3_stage<R3, false, kThreadsXZ>(tile, TY, pitch, hp, priv, 1 << LHX2, twl + TW::r3);
                stage_sync(priv);
            }
            lds_fft<LHX2, false, kThreadsXZ, R3>(tile, TY * R3, pitch, hp, priv, twl);
        }
        if (priv) lds_barrier();
        float4* sdst = reinterpret_cast<float4*>(S_next + ((size_t)z * Hx) * d.xrow + y0);
#pragma unroll MI_FFT_UNROLL
        for (int i = threadIdx.x; i < hp * Hx; i += kThreadsXZ) {
            const int px = i / hp, rp = i - px * hp;
            const int c0 = cell(2 * rp, pitch, hp, px);
            const float2 a = tile[c0], b = tile[c0 + pitch];
            sdst[(size_t)px * rowq + rp] = make_float4(a.x, a.y, b.x, b.y);
        }
    }
}

// ---------------------------------------------------------------------------------------------- P5 + P1, pipelined
// The fused x pass as a persistent kernel: one work-group per CU walks over tiles and keeps HBM busy during the FFT phases --
// the epilogue operand of the current tile is requested before the inverse transform and the next tile's spectrum before the
// forward transform, both into registers (8 float4 each for a 16 x 1024 tile); stores drain behind.
// Unpadded volumes only (the padded mode keeps k_x_inverse).
// MODE 0: the fused pass.  MODE 1: forward only -- the rows of the real volume `e.a` are transformed into S_next (k_x_forward as a
// persistent kernel: the next tile's rows travel during the transform and the store of the current one).  MODE 2: inverse only --
// T -> epilogue (none / ratio / update) -> out, nothing is transformed forward (k_x_inverse without the regularised epilogues).
template <int LHX2, int R3, int MODE = 0>
__global__ __launch_bounds__(kThreadsXZ, kWavesXZ) void k_x_fused_pipe(const float2* __restrict__ T, float* __restrict__ out, ConvEpilogue e, NativeDims d,
                                                            const float2* __restrict__ tw, float2* __restrict__ S_next, int EPI, int ntiles,
                                                            TileSelect sel, PadWindow pw, int* __restrict__ tile_ctr) {
    extern __shared__ __attribute__((aligned(16))) float2 tile[];
    // Tile hand-out.  tile_ctr == nullptr: work-group b takes tiles b, b + grid, b + 2 grid, ...  Otherwise every tile comes from a
    // device counter (zeroed by the host; a work-group takes two numbers when it starts, then one atomicAdd per tile): a work-group
    // whose CU was busy with something else when the launch began -- a collective's kernels during a halo exchange -- then simply
    // takes fewer tiles, or none, instead of leaving a fixed share as the tail of the pass.  A number is fetched a whole tile
    // ahead (requested at the top of a tile, published through LDS behind the tile's last barrier), so its latency never sits on
    // the tile's chain; only the first fetch of a work-group is waited for.
    __shared__ int s_next_tile;
    constexpr int Hx = R3 << LHX2, NW = kThreadsXZ / 64;
    constexpr int TY = x_tile_rows(Hx), hp = TY / 2, quads = Hx / 2;
    constexpr int NQ = hp * Hx;  // float4 per tile, in the transposed (T / S) and in the row (bl) view alike
    constexpr int NPF = (NQ + kThreadsXZ - 1) / kThreadsXZ;
    constexpr int pitch = row_pitch(Hx);
    constexpr int P = kThreadsXZ / hp;  // item j of a lane in the transposed view: column px0 + j * P, row pair rp
    // rows dealt to the waves: the inverse transform, the epilogue and the forward transform of a row all belong to its owner
    // and run without work-group barriers; only the transposed fill and drain are tile-wide
    constexpr bool PRIV = (TY % NW == 0) && (NQ % kThreadsXZ == 0) && (quads % 64 == 0);
    const int ytiles = d.ny / TY, rowq = d.xrow / 2;
    // Lane constants of the two views (tile-invariant, a handful of registers).  The swizzle is XOR-linear, so the slot of
    // item j is the slot of item 0 XOR a compile-time constant: px0 < P and j * P (2 * lane < 128 and the multiples of 128 of
    // the row view) occupy disjoint bits.
    // (They are recomputed at the start of every phase from a laundered thread index: kept live across the FFT phases they
    // push the kernel over its 128 VGPRs, and every spilled dword costs ~0.5 GB of scratch traffic per launch.)
    struct TView { int row, slot; size_t off; };   // transposed view: item j = column px0 + j * P, row pair rp
    auto t_view = [&]() {
        const int tid = launder(threadIdx.x);
        const int px0 = tid / hp, rp = tid - px0 * hp;
        return TView{(2 * rp) * pitch, phys(px0) ^ rmask(2 * rp, hp), (size_t)px0 * rowq + rp};
    };
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // row view, item j: PRIV: float4 u = lane + 64 j of the wave's rows -> row rl * NW + wave, quad q; else float4 tid + j * NT
    struct RView { int tid, lane, slot; };
    auto r_view = [&]() {
        const int tid = launder(threadIdx.x);
        return RView{tid, tid & 63, phys(2 * (tid & 63))};
    };
    auto r_item = [&](const RView& rv, int j, int& i, int& c, int& r, int& q) {
        if (PRIV) {
            const int rl = (64 * j) / quads, q0 = (64 * j) % quads;  // compile-time after unrolling
            r = rl * NW + wave;                                        // scalar
            q = q0 + rv.lane;
            i = r * quads + q;
            c = r * pitch + (rv.slot ^ swz_c(2 * q0) ^ rmask(r, hp));
        } else {
            i = rv.tid + j * kThreadsXZ;
            r = i / quads;
            q = i - r * quads;
            c = cell(r, pitch, hp, 2 * q);
        }
    };
    // Padded grids (zero rule, data at the origin, nx a multiple of 4): row r of a tile is row y0 + r of the caller's volume when
    // that is < ny, its quads q < nx / 4 hold data; everything else is padding (epilogue result 0).  Only the live tiles are
    // enumerated (mode 3); the tiles of live planes that lie entirely in the y padding are zero-filled first.
    const int data_quads = pw.on ? pw.n[0] / 4 : quads;
    if (pw.on && MODE != 2) {  // (the inverse-only mode writes no spectrum)
        const int nty = sel.n0, nzl = pw.n[2];
        for (int u = blockIdx.x; u < d.z_in_hi * ytiles; u += gridDim.x) {  // every tile the next y pass reads ...
            const int z = u / ytiles, ty = u - z * ytiles;
            if (z < nzl && ty < nty) continue;                               // ... that the loop below does not produce
            float4* sdst = reinterpret_cast<float4*>(S_next + ((size_t)z * Hx) * d.xrow + ty * TY);
            for (int i = threadIdx.x; i < hp * Hx; i += kThreadsXZ) {
                const int px = i / hp, rp = i - px * hp;
                sdst[(size_t)px * rowq + rp] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
    }
    float4 pre[NPF];
    // tile number -> (z, first row): all y tiles of a plane, or only / all but the tiles of two row ranges (the slab driver
    // sends the edge rows off while the rest of the pass runs)
    auto tile_zy = [&](int t, int& z, int& y0) {
        const int per = sel.mode == 0 ? ytiles : (sel.mode == 1 ? sel.n0 + sel.n1 : sel.mode == 3 ? sel.n0 : ytiles - sel.n0 - sel.n1);
        const int zl = t / per;
        z = sel.z0 + zl;
        int ty = t - zl * per;
        if (sel.mode == 1) {
            ty = ty < sel.n0 ? sel.lo0 + ty : sel.lo1 + (ty - sel.n0);
        } else if (sel.mode == 2) {
            if (ty >= sel.lo0) ty += sel.n0;
            if (ty >= sel.lo1) ty += sel.n1;
        }
        y0 = ty * TY;
    };
    auto tile_base = [&](int t) { int z, y0; tile_zy(t, z, y0); return ((size_t)z * Hx) * d.xrow + y0; };
    auto load_T = [&](int t) {
        const TView tv = t_view();
        const float4* src = reinterpret_cast<const float4*>(T + tile_base(t)) + tv.off;
#pragma unroll
        for (int j = 0; j < NPF; ++j)
            if (NQ % kThreadsXZ == 0 || (int)threadIdx.x + j * kThreadsXZ < NQ) pre[j] = src[(size_t)(j * P) * rowq];
    };
    using TW = TwLds<LHX2, R3>;
    float2* twl = tile + TY * pitch;
    TW::template fill<kThreadsXZ>(twl, tw);
    // MODE 1: the rows of a tile in the row view (float4 j of a lane as in r_item), requested one tile ahead into `pre`
    auto load_rows = [&](int t) {
        int z, y0;
        tile_zy(t, z, y0);
        // (padded grids: rows and quads beyond the caller's volume are zero, as in the epilogue below)
        const size_t row0 = pw.on ? ((size_t)z * pw.n[1] + y0) * (size_t)data_quads : ((size_t)z * d.ny + y0) * (size_t)quads;
        const int rows_live = pw.on ? pw.n[1] - y0 : TY;
        const float4* src4 = reinterpret_cast<const float4*>(e.a);
        const RView rv = r_view();
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            int i, c, r, q;
            r_item(rv, j, i, c, r, q);
            if (NQ % kThreadsXZ == 0 || i < NQ) {
                const bool live = r < rows_live && q < data_quads;
                pre[j] = live ? src4[pw.on ? row0 + (size_t)r * data_quads + q : row0 + i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
    };
    const bool dyn = tile_ctr != nullptr;
    int t = blockIdx.x, tn = t + (int)gridDim.x;
    if (dyn) {
        if (threadIdx.x == 0) s_next_tile = atomicAdd(tile_ctr, 2);
        lds_barrier();
        t = __builtin_amdgcn_readfirstlane(s_next_tile);
        tn = t + 1;
        lds_barrier();  // (everybody has read the slot before the first tile's owner of lane 0 overwrites it)
    }
    if (t < ntiles) {
        if constexpr (MODE == 1) load_rows(t);
        else load_T(t);
    }
    if constexpr (MODE == 1) lds_barrier();  // the tables (the other modes meet a barrier before their first transform)
    for (; t < ntiles;) {
        int fetched = 0;
        if (dyn && threadIdx.x == 0) fetched = atomicAdd(tile_ctr, 1);  // the tile after the next one
        // behind the last barrier of a tile: t <- tn, tn <- the fetched number (or the static successor)
        auto advance = [&]() {
            t = tn;
            tn = dyn ? __builtin_amdgcn_readfirstlane(s_next_tile) : tn + (int)gridDim.x;
        };
        if constexpr (MODE != 1) {
            const TView tv = t_view();
#pragma unroll
            for (int j = 0; j < NPF; ++j) {
                if (NQ % kThreadsXZ == 0 || (int)threadIdx.x + j * kThreadsXZ < NQ) {
                    const int c0 = tv.row + (tv.slot ^ swz_c(j * P));
                    tile[c0] = make_float2(pre[j].x, pre[j].y);
                    tile[c0 + pitch] = make_float2(pre[j].z, pre[j].w);
                }
            }
        }
        // rows of this tile in the real volume: contiguous TY * 2 Hx floats
        int z, y0;
        tile_zy(t, z, y0);
        // float4 index of (row r, quad q) of this tile in the caller's volume: rows are 2 Hx floats apart, or nx on a padded grid
        const size_t row0 = pw.on ? ((size_t)z * pw.n[1] + y0) * (size_t)data_quads : ((size_t)z * d.ny + y0) * (size_t)quads;
        const int rows_live = pw.on ? pw.n[1] - y0 : TY;  // rows of the tile that exist in the caller's volume
        auto g_index = [&](int i, int r, int q) { return pw.on ? row0 + (size_t)r * data_quads + q : row0 + i; };
        const float4* a4 = reinterpret_cast<const float4*>(e.a);
        float4 av[NPF];
        auto load_a = [&]() {
            {
                const RView rv = r_view();
    #pragma unroll
                for (int j = 0; j < NPF; ++j) {
                    int i, c, r, q;
                    r_item(rv, j, i, c, r, q);
                    if (NQ % kThreadsXZ == 0 || i < NQ) {
                        const bool live = r < rows_live && q < data_quads && !(MODE == 2 && EPI == EPI_NONE);  // (no operand then)
                        av[j] = live ? a4[g_index(i, r, q)] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    }
                }
            }
        };
        if constexpr (MODE == 1) {
#pragma unroll
            for (int j = 0; j < NPF; ++j) av[j] = pre[j];
            if (tn < ntiles) load_rows(tn);
        } else {
        // (the inverse-only launch with a sixteen-point top stage: the operand is requested in front of that stage -- held across
        // the whole transform it does not fit the 128 registers beside the stage's sixteen points)
        constexpr int TOP_LO = seg_below(LHX2, LHX2);
        constexpr bool LATE_A = MODE == 2 && R3 == 1 && LHX2 - TOP_LO == 4;
        if (R3 != 9 && !LATE_A) load_a();  // (radix-9 rows: requested behind the 9-point stage, which needs the registers)
        lds_barrier();
        if constexpr (LATE_A) {
            lds_fft<LHX2, true, kThreadsXZ, R3, 0, TOP_LO>(tile, TY * R3, pitch, hp, PRIV, twl);
            load_a();
            lds_fft<LHX2, true, kThreadsXZ, R3, TOP_LO, LHX2>(tile, TY * R3, pitch, hp, PRIV, twl);
        } else {
            lds_fft<LHX2, true, kThreadsXZ, R3>(tile, TY * R3, pitch, hp, PRIV, twl);
        }
        if constexpr (R3 > 1) {
            radix3_stage<R3, true, kThreadsXZ>(tile, TY, pitch, hp, PRIV, 1 << LHX2, twl + TW::r3);
            stage_sync(PRIV);
        }
        if (R3 == 9) load_a();
        }
        float4* dst = reinterpret_cast<float4*>(out);
        const RView rv = r_view();
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            int i, s0, r, q;
            r_item(rv, j, i, s0, r, q);  // elements 2q and 2q + 1 are slot neighbours
            if (NQ % kThreadsXZ == 0 || i < NQ) {
                const float4 a = av[j];
                if constexpr (MODE == 1) {  // the volume's rows, as they are
                    tile[s0] = make_float2(a.x, a.y);
                    tile[s0 ^ 1] = make_float2(a.z, a.w);
                    continue;
                }
                const float2 c0 = tile[s0], c1 = tile[s0 ^ 1];
                const bool live = r < rows_live && q < data_quads;
                float4 o;
                if (MODE == 2 && EPI == EPI_NONE)
                    o = make_float4(c0.x, c0.y, c1.x, c1.y);
                else if (EPI == EPI_RATIO)
                    o = make_float4(a.x * rcp_eps(c0.x), a.y * rcp_eps(c0.y), a.z * rcp_eps(c1.x), a.w * rcp_eps(c1.y));
                else
                    o = make_float4(fabsf(a.x * c0.x), fabsf(a.y * c0.y), fabsf(a.z * c1.x), fabsf(a.w * c1.y));
                if (!live) o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // the zero padding of the next convolution's input
                if (out != nullptr && live) dst[g_index(i, r, q)] = o;
                if constexpr (MODE != 2) {
                    tile[s0] = make_float2(o.x, o.y);
                    tile[s0 ^ 1] = make_float2(o.z, o.w);
                }
            }
        }
        // (radix-9 rows: the 9-point stage needs the registers, so the next tile is requested behind it)
        if constexpr (MODE == 2) {  // nothing goes forward: the tile is free once everybody has read its rows
            if (tn < ntiles) load_T(tn);
            if (dyn && threadIdx.x == 0) s_next_tile = fetched;
            lds_barrier();
            advance();
            continue;
        }
        if (MODE == 0 && R3 != 9 && tn < ntiles) load_T(tn);
        stage_sync(PRIV);
        if constexpr (R3 > 1) {
            radix3_stage<R3, false, kThreadsXZ>(tile, TY, pitch, hp, PRIV, 1 << LHX2, twl + TW::r3);
            stage_sync(PRIV);
        }
        if (MODE == 0 && R3 == 9 && tn < ntiles) load_T(tn);
        lds_fft<LHX2, false, kThreadsXZ, R3>(tile, TY * R3, pitch, hp, PRIV, twl);
        if (PRIV) lds_barrier();  // rows complete for everybody before the transposed drain
        const TView tv = t_view();
        float4* sdst = reinterpret_cast<float4*>(S_next + tile_base(t)) + tv.off;
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            if (NQ % kThreadsXZ == 0 || (int)threadIdx.x + j * kThreadsXZ < NQ) {
                const int c0 = tv.row + (tv.slot ^ swz_c(j * P));
                const float2 a = tile[c0], b = tile[c0 + pitch];
                sdst[(size_t)(j * P) * rowq] = make_float4(a.x, a.y, b.x, b.y);
            }
        }
        if (dyn && threadIdx.x == 0) s_next_tile = fetched;
        lds_barrier();  // the tile is free for the next fill
        advance();
    }
}

// rows [y0, y0 + rows) of the x-transformed buffer S[z][px][py] <-> a contiguous buffer [z * Hx + px][rows] (halo exchange of
// the sharded iteration); dir 0: pack, 1: unpack, 2: zero-fill
__global__ __launch_bounds__(256) void k_spectrum_rows(float2* __restrict__ S, float2* __restrict__ buf, size_t lines, int M, int y0, int rows,
                                                       int dir) {
    const size_t total = lines * (size_t)rows;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t line = i / rows;
        const int j = (int)(i - line * rows);
        float2* cell = S + line * M + (y0 + j);
        if (dir == 0) buf[i] = *cell;
        else if (dir == 1) *cell = buf[i];
        else *cell = make_float2(0.0f, 0.0f);
    }
}

// Complex pair OTF -> real pair OTF: Gr = Re(G * conj(P)) with P the phase ramp of the PSF centre's offset; the largest
// imaginary part that is dropped and the largest magnitude are returned (float bits, atomic max) for the host's decision.
__global__ __launch_bounds__(256) void k_g_to_real(const float4* __restrict__ G, float2* __restrict__ Gr, NativeDims d, RealOtf ro,
                                                    size_t total, unsigned* __restrict__ stats) {
    const int M = d.ny, L = d.nz;
    float max_im = 0.0f, max_re = 0.0f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int pz = (int)(i % L);
        const size_t r = i / L;
        const int py = (int)(r % M), xk = (int)(r / M);
        const float2 P = cmul(cmul(ro.ph_x[xk], ro.ph_y[y_pos2freq(py, d)]), ro.ph_z[pos2freq(pz, d.lz2, d.r3z)]);
        const float4 g = G[i];
        const float2 a = cmulc(make_float2(g.x, g.y), P), b = cmulc(make_float2(g.z, g.w), P);
        Gr[i] = make_float2(a.x, b.x);
        max_im = fmaxf(max_im, fmaxf(fabsf(a.y), fabsf(b.y)));
        max_re = fmaxf(max_re, fmaxf(fabsf(a.x), fabsf(b.x)));
    }
    atomicMax(&stats[0], __float_as_uint(max_im));  // non-negative floats order like their bit patterns
    atomicMax(&stats[1], __float_as_uint(max_re));
}

bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

}  // namespace

// An axis length n = r3 * 2^l2 with r3 in {1, 3, 9}: powers of two from 8 to 4096, or 3 * / 9 * (32 .. 512).  (A radix-3/9
// factor matters most on y, the axis the slab driver shards, where slab + halos is rarely a power of two; on x and z it
// keeps zero-padded deconFFT shapes close to the 7-smooth ones.)
static bool split_axis(int n, int* r3, int* l2, bool five = false) {
    for (int r : {1, 3, 5, 9}) {
        if (n % r || (r == 5 && !five)) continue;
        const int m = n / r;
        if (!is_pow2(m)) continue;
        const int l = ilog2(m);
        if (r == 1 ? (l >= 3 && l <= 12) : (l >= 5 && l <= (r == 5 ? 8 : 9))) { *r3 = r; *l2 = l; return true; }
    }
    return false;
}

static const int kMaxZ = 2304;  // 2 * TL >= 4 rows of the z pass must fit the LDS tile

bool NativeFft::supported(const int F[3]) {
    // x: real length 2 * Hx, the transform runs on Hx complex points
    int r3, l2;
    return F[0] % 2 == 0 && split_axis(F[0] / 2, &r3, &l2) && split_axis(F[1], &r3, &l2, true) && split_axis(F[2], &r3, &l2) && F[2] <= kMaxZ;
}

int NativeFft::good_size(int n, int axis) {
    int r3, l2;
    if (axis == 0) {
        for (int h = n < 16 ? 8 : (n + 1) / 2;; ++h)
            if (split_axis(h, &r3, &l2)) return 2 * h;
    }
    for (int m = n < 8 ? 8 : n;; ++m)
        if (split_axis(m, &r3, &l2, axis == 1)) return (axis == 2 && m > kMaxZ) ? 0 : m;
}

// LDS of an axis kernel: the tile, then the twiddle tables (both chains and the radix-3/9 table: what the fused kernels of
// the axis use, an upper bound for the others; see TwLds)
static size_t lds_bytes(int rows, int n) { return sizeof(float2) * ((size_t)rows * row_pitch(n) + axis_tw_entries(n)); }

namespace {
thread_local int tl_no_placement_trial = 0;
}
NoPlacementTrial::NoPlacementTrial() { ++tl_no_placement_trial; }
NoPlacementTrial::~NoPlacementTrial() { --tl_no_placement_trial; }

int NativeFft::init(hipStream_t s, const int F[3], bool explicit_adjoint) {
    MI_REQUIRE(supported(F), "native FFT: unsupported shape %d x %d x %d", F[0], F[1], F[2]);
    const int Hx = F[0] / 2;
    split_axis(Hx, &dims.r3x, &dims.lhx2);
    split_axis(F[1], &dims.r3, &dims.ly2, true);  // (the y axis also takes 5 * 2^a: 320 rows of a slab rank instead of 384)
    split_axis(F[2], &dims.r3z, &dims.lz2);
    dims.hx = Hx;
    dims.ny = F[1];
    dims.nz = F[2];
    dims.ty = std::min(x_tile_rows(Hx), F[1]);
    dims.tc = y_tile_cols(F[1]);
    dims.tl = std::min(z_tile_lines(F[2]), F[1]);
    dims.z_in_hi = F[2];
    dims.z_out_lo = 0;
    dims.z_out_hi = F[2];
    dims.y_out_hi = F[1];
    dims.xk0 = 0;
    dims.xkn = Hx / 2 + 1;
    dims.yz0 = 0;
    dims.dbg = 0;
    if (const char* e = MI_PROBE_ENV("MI_FFT_ZDBG")) dims.dbg = atoi(e);  // phase knock-out for timing experiments
    // tuning overrides (experiments only): MI_FFT_TY / MI_FFT_TC / MI_FFT_TL
    if (const char* e = MI_PROBE_ENV("MI_FFT_TY")) dims.ty = std::max(2, std::min(atoi(e), F[1]));
    if (const char* e = MI_PROBE_ENV("MI_FFT_TC")) dims.tc = std::max(1, atoi(e));
    if (const char* e = MI_PROBE_ENV("MI_FFT_TL")) dims.tl = std::max(2, std::min(atoi(e), F[1]));
    while ((size_t)F[2] * Hx % dims.tc) dims.tc >>= 1;
    // tiles are whole float4 groups of rows / lines and must divide y; z tiles of TL positions must map onto aligned
    // mirror blocks, which holds for TL <= 2^ly2 (positions inside one power-of-two sub-block mirror inside one)
    MI_REQUIRE(dims.ty >= 2 && dims.ty % 2 == 0 && F[1] % dims.ty == 0, "native FFT: x tile of %d rows does not divide y = %d", dims.ty, F[1]);
    MI_REQUIRE(dims.tl >= 2 && is_pow2(dims.tl) && dims.tl <= (1 << dims.ly2), "native FFT: z tile of %d lines does not fit y = %d", dims.tl, F[1]);
    MI_REQUIRE(lds_bytes(dims.ty, Hx) <= 160 * 1024 && lds_bytes(dims.tc, F[1]) <= 160 * 1024 &&
                   lds_bytes(2 * dims.tl, F[2]) <= 160 * 1024,
               "native FFT: transform too long for LDS");
    // pair-interleaved z-side layout (k_y_pair / k_z_pair_pipe): z a power of two the paired z pass takes, whole blocks of
    // kPairLines lines, an even number of columns per y tile; MI_FFT_NO_PAIR=1 keeps the plain layout (A/B measurements)
    const bool z_pairs = dims.r3z == 1 ? (dims.lz2 >= 6 && dims.lz2 <= 10)
                         : dims.r3z == 3 ? (dims.lz2 >= 6 && dims.lz2 <= 8) : (dims.r3z == 9 && dims.lz2 >= 6 && dims.lz2 <= 7);
    dims.paired = z_pairs && F[1] % (2 * kPairLines) == 0 && dims.tc >= 2 && dims.tc % 2 == 0 &&
                  F[2] % (dims.tc / 2) == 0 && dims.dbg == 0 && std::getenv("MI_FFT_NO_PAIR") == nullptr &&
                  std::getenv("MI_FFT_NO_PIPE") == nullptr && MI_PROBE_ENV("MI_FFT_TL") == nullptr;
    n_cplx = (size_t)Hx * F[1] * F[2];
    // (the two planes that are their own mirror partners are stored twice in the paired layout)
    // Rows an exact power of two apart camp on few HBM channels: behind every row of the x side ([z][px][.]) and of the paired z
    // side ([xk][z][.]) that is at least 8 KB long lie 4 KB + 128 B of padding.  C3 (profiles/zpad_probe.py): z pass 4.75 -> 4.2 ms
    // with any odd multiple of 128 B behind the z rows (64-byte offsets break the 128-byte lines: 6.2 ms); with 4 KB + 128 B
    // on both sides the y passes drop from 3.2-3.35 to 2.85-3.35 ms and the x pass from 5.1 / 6.0-7.2 to 4.75 / 5.7-6.8 ms.
    const int pad_x = (size_t)F[1] * sizeof(float2) >= kPadRowBytes ? kRowPadBytes : 0;
    const int pad_z = (size_t)F[1] * 2 * sizeof(float2) >= kPadRowBytes ? kRowPadBytes : 0;
    dims.zpad = dims.paired ? pad_z / 16 : 0;
    dims.xrow = F[1] + pad_x / 8;
    if (const char* e = MI_PROBE_ENV("MI_FFT_ZPAD")) dims.zpad = dims.paired ? std::max(0, atoi(e)) : 0;   // float4 per row
    if (const char* e = MI_PROBE_ENV("MI_FFT_XPAD")) dims.xrow = F[1] + 2 * std::max(0, atoi(e));          // float4 per row
    const size_t n_x = (size_t)Hx * F[2] * dims.xrow;
    const size_t n_buf = std::max(n_x, dims.paired ? (size_t)(Hx / 2 + 1) * F[2] * 2 * (size_t)(F[1] + dims.zpad) : n_cplx);
    // one allocation for both arrays: their distance -- which decides how the strided streams of a pass that reads one and
    // writes the other fall onto the HBM channels -- is then the same in every context instead of whatever the driver returns
    // (measuring the passes for six distances at plan time did not pay: in a process where the y passes run in their slow mode
    // they do so for every distance tried)
    size_t gap = kSpecGapBytes;
    if (const char* e = MI_PROBE_ENV("MI_FFT_STGAP")) gap = (size_t)atoll(e) & ~(size_t)127;
    // Probe builds: MI_FFT_VMM=<order>[,<chunk MB>] backs the spectrum arrays with the virtual-memory API instead of one
    // hipMalloc -- physical chunks created one by one and mapped into a reserved range in a chosen ORDER (0 as created, 1 reversed,
    // 2 bit-reversed, 3 shuffled) -- to see whether the library can choose how physical memory meets the power-of-two row pitch
    // (profiles/r04_placement_vmm.txt).
    int vmm_order = -1, vmm_chunk_mb = 0;
    if (const char* e = MI_PROBE_ENV("MI_FFT_VMM")) {
        if (sscanf(e, "%d,%d", &vmm_order, &vmm_chunk_mb) < 1) vmm_order = -1;
    }
    if (vmm_order >= 0) {
        MI_TRY(vmm.alloc(sizeof(float2) * 2 * n_buf + gap, (size_t)std::max(0, vmm_chunk_mb) << 20, vmm_order));
        S.p = vmm.va;  // (not the pool's: handed back by ~NativeFft)
        S.bytes = sizeof(float2) * 2 * n_buf + gap;
    } else {
        MI_TRY(S.alloc(sizeof(float2) * 2 * n_buf + gap));
    }
    t_spec = S.as<float2>() + n_buf + gap / sizeof(float2);
    spec_bytes = sizeof(float2) * n_buf;
    MI_TRY(G.alloc(sizeof(float4) * (size_t)(Hx / 2 + 1) * F[1] * F[2]));
    // twiddle tables exp(-2 pi i e / N) in double on the host, per axis: e < sub/2 for the power-of-two sub-transform
    // (sub = 2^l2), followed by the full circle e < n of the radix-3/9 stage when the axis has one
    const int lens[3] = {Hx, F[1], F[2]};
    const int subs[3] = {1 << dims.lhx2, 1 << dims.ly2, 1 << dims.lz2};
    size_t off = 0, offs[3];
    for (int a = 0; a < 3; ++a) { offs[a] = off; off += (size_t)std::max(1, subs[a] / 2) + (lens[a] != subs[a] ? (size_t)lens[a] : 0); }
    std::vector<float2> h(off);
    const double two_pi = 6.283185307179586476925286766559;
    for (int a = 0; a < 3; ++a) {
        for (int e = 0; e < subs[a] / 2; ++e)
            h[offs[a] + e] = make_float2((float)std::cos(two_pi * e / subs[a]), (float)-std::sin(two_pi * e / subs[a]));
        if (lens[a] != subs[a])
            for (int e = 0; e < lens[a]; ++e)
                h[offs[a] + subs[a] / 2 + e] = make_float2((float)std::cos(two_pi * e / lens[a]), (float)-std::sin(two_pi * e / lens[a]));
    }
    MI_TRY(tw.alloc(sizeof(float2) * off));
    MI_HIP(hipMemcpyAsync(tw.p, h.data(), sizeof(float2) * off, hipMemcpyHostToDevice, s));
    tw_x = tw.as<float2>() + offs[0];
    tw_y = tw.as<float2>() + offs[1];
    tw_z = tw.as<float2>() + offs[2];
    {
        int devid = 0, cus = 0;
        MI_HIP(hipGetDevice(&devid));
        MI_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, devid));
        n_cu = cus > 0 ? cus : 256;
    }
    have_adj = explicit_adjoint;
    if (have_adj) MI_TRY(G_adj.alloc(G.bytes));  // explicit adjoint kernel (psf_inv of the 'same'-convolution flavour) instead of conj(OTF)
    MI_HIP(hipStreamSynchronize(s));  // host twiddle vector dies at scope exit
    // ---- where the spectrum arrays lie.  A strided pass runs at one of two speeds depending on the PHYSICAL memory behind the array
    // it reads and the array it writes: K buffers of one array's size allocated side by side fall into groups (runs of ~32 GB on one
    // box: the size of an HBM stack), and the forward y pass of C3 takes 3.10 ms between two buffers of one group, 2.98 ms across
    // groups; the update launch of the x pass takes 6.1 instead of 5.5 ms when the array it writes shares a group with the volume
    // (profiles/r04_spectrum_halves.txt).  Two arrays carved out of ONE allocation -- rounds 1-3 -- mostly share a group: the slow
    // placement of those rounds, and what a fresh process' first allocation regularly gets.  So large arrays are placed by trial:
    // up to MI_FFT_PLACE_CANDIDATES (6) buffers are allocated side by side (as many as the free memory allows beside 24 GB for the
    // caller), the passes are timed on every ordered pair (S read by the four y passes, T read by the two z and the two x passes of
    // an iteration: cost = 4 y(S -> T) + 3 update(T -> S), on a stand-in volume, contents do not matter), the best pair stays, the
    // rest goes back to the driver.  ~0.5 s and, for a moment, the candidates' memory at plan creation; arrays of
    // MI_FFT_PLACE_MIN_MB (6144, both together) and more -- smaller plans are not tried: decwrap creates its block plans, 3-4 GB
    // each, on several workers per device while others compute, and every released candidate is a device-wide synchronisation
    // (slab.SlabRL lowers the limit for its rank, which has its device to itself).
    size_t place_min = (size_t)6 << 30;
    if (const char* e = std::getenv("MI_FFT_PLACE_MIN_MB")) place_min = (size_t)std::max(0LL, atoll(e)) << 20;
    if (vmm_order < 0 && S.bytes >= place_min && tl_no_placement_trial == 0) {
        int tries = 6;
        if (const char* e = std::getenv("MI_FFT_PLACE_CANDIDATES")) tries = std::max(1, std::min(8, atoi(e)));
        // one trial at a time per device (plans created concurrently -- decwrap's workers with a large --block-size-max -- would each
        // hold their candidates and push each other out of memory); what the pool keeps cached goes back to the driver first: the
        // candidates are allocated behind the pool's back and get none of its trim-on-failure
        static std::mutex trial_mu[16];
        int dev_id = 0;
        MI_HIP(hipGetDevice(&dev_id));
        std::lock_guard<std::mutex> trial_lock(trial_mu[dev_id & 15]);
        (void)mi_release_cached_memory(dev_id);
        size_t free_b = 0, total_b = 0;
        MI_HIP(hipMemGetInfo(&free_b, &total_b));
        const size_t vol_bytes = sizeof(float) * 2 * (size_t)Hx * F[1] * F[2];
        const size_t half = sizeof(float2) * n_buf, keep = ((size_t)24 << 30) + vol_bytes;
        while (tries > 1 && (size_t)tries * half + keep > free_b) --tries;
        if (tries > 1) {
            struct TrialEvents {   // (destroyed on every path out of the trial)
                hipEvent_t a = nullptr, b = nullptr;
                ~TrialEvents() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
            } tev;
            MI_HIP(hipEventCreate(&tev.a));
            MI_HIP(hipEventCreate(&tev.b));
            const hipEvent_t e0 = tev.a, e1 = tev.b;
            (void)hipFree(S.p);   // (the single allocation made above makes room for the candidates)
            S.p = nullptr;
            const size_t block_bytes = S.bytes;
            S.bytes = 0;
            t_spec = nullptr;
            void* xtmp = nullptr;
            if (hipMalloc(&xtmp, vol_bytes) != hipSuccess) { (void)hipGetLastError(); xtmp = nullptr; }
            std::vector<void*> cand;
            for (int i = 0; i < tries; ++i) {
                void* q = nullptr;
                if (hipMalloc(&q, half) != hipSuccess) { (void)hipGetLastError(); break; }
                cand.push_back(q);
            }
            const int K = (int)cand.size();
            int rc = MI_OK, bi = -1, bj = -1, kept_idx = -1;
            float best = 0.0f;
            std::vector<float> ms, tyv((size_t)K * K, 0.0f);
            auto timed = [&](auto&& launch, float* out) {   // (two launches, the second counts)
                for (int rep = 0; rep < 2 && rc == MI_OK; ++rep) {
                    (void)hipEventRecord(e0, s);
                    rc = launch();
                    (void)hipEventRecord(e1, s);
                    if (rc == MI_OK && hipEventSynchronize(e1) != hipSuccess) rc = fail(MI_ERR_HIP, "native FFT: placement trial failed");
                    if (rc == MI_OK) (void)hipEventElapsedTime(out, e0, e1);
                }
            };
            for (int i = 0; i < K && rc == MI_OK; ++i)
                for (int j = 0; j < K && rc == MI_OK; ++j) {
                    if (i == j) continue;
                    S.p = cand[i];
                    t_spec = static_cast<float2*>(cand[j]);
                    float ty = 0.0f, tx = 0.0f;
                    timed([&] { return y_pass(s, false, dims.paired != 0); }, &ty);
                    if (xtmp) {
                        ConvEpilogue ep;
                        ep.a = static_cast<const float*>(xtmp);
                        timed([&] { return x_inverse(s, static_cast<float*>(xtmp), EPI_UPDATE, ep, true); }, &tx);
                    }
                    tyv[(size_t)i * K + j] = ty;
                    const float cost = 4.0f * ty + 3.0f * tx;
                    if (bi < 0 || cost < best) { best = cost; bi = i; bj = j; kept_idx = (int)ms.size(); }
                    ms.push_back(cost);
                }
            if (xtmp) (void)hipFree(xtmp);
            // a second buffer for S stays until the first call that brings the caller's volume: the update launch is slow when S
            // shares a region with THAT volume, which nothing here can know (NativeFft::iterate settles it: settle_s)
            int bk = -1;
            size_t alt_min = (size_t)8 << 30;   // (MI_FFT_PLACE_ALT_MIN_MB: the smallest array that keeps a second buffer for S)
            if (const char* e = std::getenv("MI_FFT_PLACE_ALT_MIN_MB")) alt_min = (size_t)std::max(0LL, atoll(e)) << 20;
            if (rc == MI_OK && bi >= 0 && half >= alt_min)
                for (int k = 0; k < K; ++k)
                    if (k != bi && k != bj && tyv[(size_t)k * K + bj] <= 1.03f * tyv[(size_t)bi * K + bj] + 0.02f &&
                        (bk < 0 || tyv[(size_t)k * K + bj] < tyv[(size_t)bk * K + bj]))
                        bk = k;
            for (int i = 0; i < K; ++i)
                if (rc != MI_OK || bi < 0 || (i != bi && i != bj && i != bk)) (void)hipFree(cand[i]);
            if (bk >= 0) { S_alt.p = cand[bk]; S_alt.bytes = half; }
            S.p = nullptr;
            t_spec = nullptr;
            if (rc != MI_OK) return rc;
            if (bi < 0) {   // (fewer than two candidates: back to the single allocation)
                MI_TRY(S.alloc(block_bytes));
                t_spec = S.as<float2>() + n_buf + gap / sizeof(float2);
            } else {
                S.p = cand[bi];
                S.bytes = half;
                T2.p = cand[bj];
                T2.bytes = half;
                t_spec = T2.as<float2>();
                placement_ms = ms;
                placement_kept = kept_idx;   // (index in the list of ordered pairs (i, j), i != j, i slowest)
                if (std::getenv("MI_FFT_PLACE_LOG")) {   // (diagnostics on stderr)
                    float worst = best;
                    for (float v : ms) worst = std::max(worst, v);
                    std::fprintf(stderr, "native FFT: 2 x %.1f GB placed on buffers %d (S) and %d (T) of %d: 4 y + 3 update %.2f ms (pairs from %.2f to %.2f)\n",
                                 (double)half / 1e9, bi, bj, K, (double)best, (double)best, (double)worst);
                }
            }
        }
    }
    return MI_OK;
}

// ---- launch helpers: the kernels are templated on (log2 of the power-of-two part, radix-3/9 factor) of their axis;
// key = l2 * 16 + r3
#define MI_AXIS_CASES(M) M(3, 1) M(4, 1) M(5, 1) M(6, 1) M(7, 1) M(8, 1) M(9, 1) M(10, 1) M(11, 1) M(12, 1) \
    M(5, 3) M(6, 3) M(7, 3) M(8, 3) M(9, 3) M(5, 9) M(6, 9) M(7, 9) M(8, 9) M(9, 9)
// y: also 5 * 2^a (only the y kernels are built for it)
#define MI_Y_CASES(M) MI_AXIS_CASES(M) M(5, 5) M(6, 5) M(7, 5) M(8, 5)
// z: lengths up to kMaxZ
#define MI_Z_CASES(M) M(3, 1) M(4, 1) M(5, 1) M(6, 1) M(7, 1) M(8, 1) M(9, 1) M(10, 1) M(11, 1) \
    M(5, 3) M(6, 3) M(7, 3) M(8, 3) M(9, 3) M(5, 9) M(6, 9) M(7, 9) M(8, 9)

template <class K, class... Args>
static int launch_lds(K kernel, unsigned grid, int threads, size_t lds, hipStream_t s, const char* name, Args... args) {
    if (lds > 64 * 1024)
        MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, s, args...);
    return launch_check(name);
}

void NativeFft::set_window(const int n[3], const int o[3], const int rep[3], const int k[3]) {
    pw.on = 1;
    if (S_alt.p && alt_phase < 3) {   // (padded grids do not settle S on their update launches: the second buffer is not needed)
        if (alt_phase == 1 || alt_phase == 2) { /* nothing to undo: S.p is whichever buffer the last launch wrote */ }
        (void)hipFree(S_alt.p);
        S_alt.p = nullptr;
        S_alt.bytes = 0;
        alt_phase = 3;
    }
    for (int a = 0; a < 3; ++a) {
        pw.n[a] = n[a];
        pw.o[a] = o[a];
        pw.rep[a] = rep[a];
        pw.w[a] = n[a] + k[a] - 1;
    }
    // what the padding leaves to prune (MI_FFT_NO_PRUNE=1 keeps the full passes, for A/B measurements)
    if (std::getenv("MI_FFT_NO_PRUNE") != nullptr) return;
    const int in_hi = rep[2] ? pw.w[2] : o[2] + n[2];
    dims.z_in_hi = std::min(in_hi, dims.nz);
    dims.z_out_lo = o[2];
    dims.z_out_hi = std::min(o[2] + n[2], dims.nz);
    const int yh = ((o[1] + n[1] + dims.ty - 1) / dims.ty) * dims.ty;  // the x pass reads whole tiles of rows
    dims.y_out_hi = std::min(yh, dims.ny);
}

bool NativeFft::can_fuse() const { return !pw.on || !(pw.rep[0] || pw.rep[1] || pw.rep[2]); }

// the fused x pass runs as the persistent pipelined kernel, which can also process a subset of its tiles
bool NativeFft::pipe_ok() const {
    static const bool no_pipe = std::getenv("MI_FFT_NO_PIPE") != nullptr;
    return dims.dbg == 0 && !no_pipe && dims.ty == x_tile_rows(dims.hx);
}
bool NativeFft::splits() const { return !pw.on && pipe_ok(); }

// tiles of the fused x pass that hold rows of [a0, a1) or [b0, b1) (a before b): mode 1 = only those, 2 = all the others
TileSelect NativeFft::edge_tiles(int mode, int a0, int a1, int b0, int b1) const {
    TileSelect t{};
    const int ty = dims.ty;
    t.mode = mode;
    t.lo0 = a0 / ty;
    t.n0 = (a1 + ty - 1) / ty - t.lo0;
    t.lo1 = std::max(b0 / ty, t.lo0 + t.n0);   // overlapping ranges: the second one starts behind the first
    t.n1 = std::max((b1 + ty - 1) / ty - t.lo1, 0);
    return t;
}

// Grid of a persistent x launch and, when its tiles are handed out dynamically, the armed counter (see k_x_fused_pipe).
// Tiles come from the counter by default: compute units do not all run at the same speed, and the static stride left the
// slowest one as the tail (C3: paired z pass 4.36 -> 3.94 ms, fused x pass 5.13 / 5.71 -> 5.06 / 5.61 ms; part 2 of a slab rank's
// x pass at N = 8: 0.77 -> 0.66 ms; profiles/r03_overlap_probe.txt).  `overlapped`: the launch runs beside a halo exchange
// (part 2 of a sharded step) and follows mi_rl_set_overlap: `free_cus` compute units are left to the collective's kernels.
// MI_X_DYN=0|1 / MI_X_FREE_CUS=<k> override for every launch (A/B measurements).
int NativeFft::persistent_grid(hipStream_t s, int ntiles, bool overlapped, unsigned* grid, int** ctr_out) {
    static const char* env_dyn = std::getenv("MI_X_DYN");
    static const char* env_free = MI_PROBE_ENV("MI_X_FREE_CUS");
    const bool dyn = env_dyn ? atoi(env_dyn) != 0 : (overlapped ? overlap_dynamic : x_dynamic);
    const int free_cus = env_free ? atoi(env_free) : (overlapped ? overlap_free_cus : 0);
    const int cus = std::max(1, n_cu - std::max(0, free_cus));
    *grid = (unsigned)std::min(ntiles, cus);
    *ctr_out = nullptr;
    if (dyn) {
        if (!ctr.p) MI_TRY(ctr.alloc(256));
        // one counter per launch in flight would be needed if two dynamic launches of one context could overlap; they cannot:
        // every launch of a context goes to the caller's stream
        MI_HIP(hipMemsetAsync(ctr.p, 0, sizeof(int), s));
        *ctr_out = ctr.as<int>();
    }
    return MI_OK;
}

int NativeFft::x_forward(hipStream_t s, const float* in) {
    const PadWindow pw = this->pw;
    const int Hx = dims.hx, M = dims.ny, L = dims.nz;
    // persistent kernel with prefetch: unpadded grids, and padded ones under the conditions of the fused pass (zero rule, data at
    // the origin, whole float4 rows): only the tiles that hold rows of the volume are transformed, the others zero-filled
    const bool pad_pipe = pw.on && can_fuse() && pipe_ok() && pw.o[0] == 0 && pw.o[1] == 0 && pw.o[2] == 0 && pw.n[0] % 4 == 0;
    if ((splits() || pad_pipe) && ((uintptr_t)in % 16) == 0 && std::getenv("MI_FFT_NO_XPIPE") == nullptr) {
        ConvEpilogue e;
        e.a = in;
        TileSelect sel{};
        int per = M / dims.ty, planes = L;
        if (pad_pipe) {
            sel.mode = 3;
            sel.n0 = (pw.n[1] + dims.ty - 1) / dims.ty;
            per = sel.n0;
            planes = pw.n[2];
        }
        const int ntiles = planes * per;
        unsigned grid = 0;
        int* ctr_p = nullptr;
        MI_TRY(persistent_grid(s, ntiles, false, &grid, &ctr_p));
        const NativeDims d = dims;
        int rc = MI_ERR_INVALID;
#define MI_XF(LG, R) case LG * 16 + R: rc = launch_lds(k_x_fused_pipe<LG, R, 1>, grid, kThreadsXZ, lds_bytes(dims.ty, Hx), s, "k_x_fused_pipe<forward>", (const float2*)nullptr, (float*)nullptr, e, d, tw_x, S.as<float2>(), (int)EPI_NONE, ntiles, sel, pw, ctr_p); break;
        switch (dims.lhx2 * 16 + dims.r3x) { MI_AXIS_CASES(MI_XF) default: return fail(MI_ERR_UNSUPPORTED, "native FFT: x length %d", 2 * Hx); }
#undef MI_XF
        return rc;
    }
    const unsigned xtiles = (unsigned)((size_t)L * (M / dims.ty));
    const size_t xl = lds_bytes(dims.ty, Hx);
    const NativeDims d = dims;
    float2* Sp = S.as<float2>();
    const float2* twx = tw_x;
    int rc = MI_ERR_INVALID;
#define MI_X(LG, R) case LG * 16 + R: rc = launch_lds(k_x_forward<LG, R>, xtiles, kThreadsXZ, xl, s, "k_x_forward", in, Sp, d, twx, pw); break;
    switch (dims.lhx2 * 16 + dims.r3x) { MI_AXIS_CASES(MI_X) default: return fail(MI_ERR_UNSUPPORTED, "native FFT: x length %d", 2 * Hx); }
#undef MI_X
    return rc;
}

int NativeFft::y_pass(hipStream_t s, bool inverse, bool paired, const float2* src_o, float2* dst_o, int xk0, int xkn) {
    const int Hx = dims.hx, M = dims.ny, L = dims.nz;
    if (xkn < 0) xkn = Hx / 2 + 1;
    MI_REQUIRE(paired || (xk0 == 0 && xkn == Hx / 2 + 1), "native FFT: only the paired y pass runs on a chunk of planes");
    const unsigned ycols = paired ? (unsigned)((size_t)xkn * (L / (dims.tc / 2))) : (unsigned)((size_t)L * Hx / dims.tc);
    const size_t yl = lds_bytes(dims.tc, M);
    NativeDims d = dims;
    d.xk0 = xk0;
    d.xkn = xkn;
    const float2* src = src_o ? src_o : S.as<float2>();
    float2* dst = dst_o ? dst_o : t_spec;
    const float2* twy = tw_y;
    int rc = MI_ERR_INVALID;
#define MI_Y(LG, R)                                                                                                            \
    case LG * 16 + R:                                                                                                          \
        if (paired)                                                                                                            \
            rc = inverse ? launch_lds(k_y_pair<LG, R, true>, ycols, kThreadsY, yl, s, "k_y_pair<inv>", src, dst, d, twy)        \
                         : launch_lds(k_y_pair<LG, R, false>, ycols, kThreadsY, yl, s, "k_y_pair<fwd>", src, dst, d, twy);      \
        else                                                                                                                   \
            rc = inverse ? launch_lds(k_y_pass<LG, R, true>, ycols, kThreadsY, yl, s, "k_y_pass<inv>", src, dst, d, twy)        \
                         : launch_lds(k_y_pass<LG, R, false>, ycols, kThreadsY, yl, s, "k_y_pass<fwd>", src, dst, d, twy);      \
        break;
    switch (dims.ly2 * 16 + dims.r3) { MI_Y_CASES(MI_Y) default: return fail(MI_ERR_UNSUPPORTED, "native FFT: y length %d", M); }
#undef MI_Y
    return rc;
}

int NativeFft::z_conv(hipStream_t s, bool conj_otf, const float2* src_o, float2* dst_o, int xk0, int xkn) {
    const int Hx = dims.hx, M = dims.ny, L = dims.nz;
    if (xkn < 0) xkn = Hx / 2 + 1;
    MI_REQUIRE(dims.paired || (xk0 == 0 && xkn == Hx / 2 + 1), "native FFT: only the paired z pass runs on a chunk of planes");
    const unsigned ztiles = (unsigned)((size_t)(Hx / 2 + 1) * (M / dims.tl));
    const size_t zl = lds_bytes(2 * dims.tl, L);
    NativeDims d = dims;
    d.xk0 = xk0;
    d.xkn = xkn;
    const float2* Tp = src_o ? src_o : t_spec;
    float2* Sp = dst_o ? dst_o : S.as<float2>();
    const bool adj_slot = conj_otf && have_adj;
    const float4* Gp = adj_slot ? G_adj.as<float4>() : G.as<float4>();
    const float2* twz = tw_z;
    const int cj = (conj_otf && !have_adj) ? 1 : 0;
    int rc = MI_ERR_INVALID;
    if (dims.paired) {
        const int ntiles = xkn * (M / kPairLines);
        const bool phl = !(dims.lz2 == 6 && dims.r3z == 9);  // (576-point lines: the LDS phase table would cost the second work-group)
        const size_t lds = lds_bytes(2 * kPairLines, L) + (real_otf && phl ? sizeof(float2) * (size_t)L : 0);
        const int per_cu = std::max(1, std::min(2, (int)(kLdsOneWg / lds)));  // 8 waves of 128 registers each: two fit a CU
        const unsigned grid = (unsigned)std::min(ntiles, per_cu * n_cu);
        int* ctr_p = nullptr;
        {
            static const char* env_dyn = std::getenv("MI_Z_DYN");
            if (env_dyn ? atoi(env_dyn) != 0 : z_dynamic) {
                if (!ctr.p) MI_TRY(ctr.alloc(256));
                ctr_p = ctr.as<int>() + 16 + 4 * (ctr_slot & 7);
                MI_HIP(hipMemsetAsync(ctr_p, 0, sizeof(int), s));
            }
        }
        RealOtf ro{};
        if (real_otf) {
            ro.g = adj_slot ? Gr_adj.as<float2>() : Gr.as<float2>();
            ro.ph_x = ph.as<float2>();
            ro.ph_y = ro.ph_x + (Hx / 2 + 1);
            ro.ph_z = ro.ph_y + M;
        }
#define MI_ZQ(LG, R, NTH, PH)                                                                                                            \
    case LG * 16 + R:                                                                                                                    \
        rc = real_otf ? launch_lds(k_z_pair_pipe<LG, R, true, NTH, kPairLines, PH>, grid, NTH, lds, s, "k_z_pair_pipe<real OTF>", Tp, Sp, Gp, \
                                   d, twz, cj, ntiles, ro, ctr_p)                                                                       \
                      : launch_lds(k_z_pair_pipe<LG, R, false, NTH, kPairLines, PH>, grid, NTH, lds, s, "k_z_pair_pipe", Tp, Sp, Gp, d, \
                                   twz, cj, ntiles, ro, ctr_p);                                                                         \
        break;
        // (lines of up to 576 points: 8 waves on a 64-KB tile, two work-groups per CU; 768 and 1152 points: 16 waves, one line per wave;
        // 1024 points: 8 waves again, a line pair per wave -- with the 16-point top stage a lane's 16 float4 are exactly that stage's
        // points of two lines, so it runs on the registers of the global access like the 512-point pass (246 registers, one
        // work-group per CU): 5.60 -> 5.11 ms on 1024 x 576 x 4096 against the 16-wave form, A / B in one process)
        switch (dims.lz2 * 16 + dims.r3z) {
            MI_ZQ(6, 1, 512, true) MI_ZQ(7, 1, 512, true) MI_ZQ(8, 1, 512, true) MI_ZQ(9, 1, 512, true) MI_ZQ(10, 1, 512, true)
            MI_ZQ(6, 3, 512, true) MI_ZQ(7, 3, 512, true) MI_ZQ(8, 3, 1024, true)
            MI_ZQ(6, 9, 512, false) MI_ZQ(7, 9, 1024, true)
            default: return fail(MI_ERR_UNSUPPORTED, "native FFT: paired z length %d", L);
        }
#undef MI_ZQ
        return rc;
    }
    if (z_pipelined()) {
        const int ntiles = (int)ztiles;
        const unsigned grid = (unsigned)std::min(ntiles, n_cu);
        RealOtf ro{};
        if (real_otf) {
            ro.g = adj_slot ? Gr_adj.as<float2>() : Gr.as<float2>();
            ro.ph_x = ph.as<float2>();
            ro.ph_y = ro.ph_x + (Hx / 2 + 1);
            ro.ph_z = ro.ph_y + M;
        }
#define MI_ZP(LG, R)                                                                                                                     \
    case LG * 16 + R:                                                                                                                    \
        if constexpr (z_pipe_even(R << LG)) {                                                                                            \
            if (real_otf) {                                                                                                              \
                rc = launch_lds(k_z_conv_pipe<LG, R, true>, grid, kThreadsXZ, zl, s, "k_z_conv_pipe<real OTF>", Tp, Sp, Gp, d, twz, cj, ntiles, ro); \
                break;                                                                                                                   \
            }                                                                                                                            \
        }                                                                                                                                \
        rc = launch_lds(k_z_conv_pipe<LG, R, false>, grid, kThreadsXZ, zl, s, "k_z_conv_pipe", Tp, Sp, Gp, d, twz, cj, ntiles, ro);      \
        break;
        switch (dims.lz2 * 16 + dims.r3z) { MI_Z_CASES(MI_ZP) default: return fail(MI_ERR_UNSUPPORTED, "native FFT: z length %d", L); }
#undef MI_ZP
        return rc;
    }
    MI_REQUIRE(!real_otf, "native FFT: the real OTF form needs the pipelined z pass");
#define MI_Z(LG, R)                                                                                                                      \
    case LG * 16 + R:                                                                                                                    \
        rc = launch_lds(k_z_conv<LG, R, false>, ztiles, kThreadsXZ, zl, s, "k_z_conv", Tp, Sp, Gp, d, twz, cj, (float4*)nullptr, 0.0f); \
        break;
    switch (dims.lz2 * 16 + dims.r3z) { MI_Z_CASES(MI_Z) default: return fail(MI_ERR_UNSUPPORTED, "native FFT: z length %d", L); }
#undef MI_Z
    return rc;
}

bool NativeFft::z_pipelined() const {
    static const bool no_pipe = std::getenv("MI_FFT_NO_PIPE") != nullptr;
    return dims.paired || (dims.dbg == 0 && !no_pipe && dims.tl == z_tile_lines(dims.nz));
}

// Tries the real form of the OTF(s): `delta` = offset (x, y, z) of the PSF's centre sample from the grid origin.  Keeps the
// complex form when the PSF is not mirror-symmetric about that sample (the imaginary parts left after removing the phase ramp
// exceed the rounding noise of the transform) or when the z pass of this shape cannot take it.
int NativeFft::try_real_otf(hipStream_t s, const int delta[3]) {
    const bool off = std::getenv("MI_FFT_COMPLEX_OTF") != nullptr;
    const int Hx = dims.hx, M = dims.ny, L = dims.nz;
    bool even = dims.paired != 0;
#define MI_EV(LG, R) case LG * 16 + R: even = even || z_pipe_even(R << LG); break;
    switch (dims.lz2 * 16 + dims.r3z) { MI_Z_CASES(MI_EV) default: break; }
#undef MI_EV
    if (off || !z_pipelined() || !even) return MI_OK;
    // phase tables exp(-2 pi i (k delta mod F) / F) in double on the host: x by xk <= Hx/2 (F = 2 Hx), y by ky, z by kz
    const int nx = Hx / 2 + 1;
    std::vector<float2> h((size_t)nx + M + L);
    const double two_pi = 6.283185307179586476925286766559;
    auto fill = [&](float2* dst, int n, long long F, int dl) {
        for (int k = 0; k < n; ++k) {
            const long long t = (((long long)k * dl) % F + F) % F;
            dst[k] = make_float2((float)std::cos(two_pi * (double)t / (double)F), (float)-std::sin(two_pi * (double)t / (double)F));
        }
    };
    fill(h.data(), nx, 2LL * Hx, delta[0]);
    fill(h.data() + nx, M, M, delta[1]);
    fill(h.data() + nx + M, L, L, delta[2]);
    MI_TRY(ph.alloc(sizeof(float2) * h.size()));
    MI_HIP(hipMemcpyAsync(ph.p, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice, s));
    RealOtf ro{};
    ro.ph_x = ph.as<float2>();
    ro.ph_y = ro.ph_x + nx;
    ro.ph_z = ro.ph_y + M;
    const size_t total = (size_t)nx * M * L;
    DevBuf st;
    MI_TRY(st.alloc(4 * sizeof(unsigned)));
    MI_HIP(hipMemsetAsync(st.p, 0, 4 * sizeof(unsigned), s));
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    for (int slot = 0; slot < (have_adj ? 2 : 1); ++slot) {
        DevBuf& dst = slot ? Gr_adj : Gr;
        MI_TRY(dst.alloc(sizeof(float2) * total));
        hipLaunchKernelGGL(k_g_to_real, dim3((unsigned)blocks), dim3(256), 0, s, (slot ? G_adj : G).as<float4>(), dst.as<float2>(), dims, ro,
                           total, st.as<unsigned>() + 2 * slot);
        MI_TRY(launch_check("k_g_to_real"));
    }
    float hs[4] = {0, 0, 0, 0};
    MI_HIP(hipMemcpyAsync(hs, st.p, sizeof(hs), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    bool ok = hs[0] <= 4e-6f * hs[1] && (!have_adj || hs[2] <= 4e-6f * hs[3]);
    if (ok) {
        real_otf = true;
        G.release();
        G_adj.release();
    } else {
        Gr.release();
        Gr_adj.release();
        ph.release();
    }
    return MI_OK;
}

// OTF of the placed kernel volume `placed` (shape F, real; may be the T buffer itself): forward x, y and z transforms, then
// the untangled spectrum is stored in the z pass' pair layout, times `scale`.
int NativeFft::build_otf(hipStream_t s, const float* placed, bool adjoint_slot, float scale) {
    MI_REQUIRE(!adjoint_slot || have_adj, "native FFT: no adjoint OTF slot");
    return spectrum(s, placed, adjoint_slot ? G_adj.as<float4>() : G.as<float4>(), scale);
}

// untangled half spectrum of a real F volume in the OTF layout (pairs (X[k], X[mirror k]) per point-wise item of the z pass)
int NativeFft::spectrum(hipStream_t s, const float* vol, float4* Gp, float scale) {
    MI_REQUIRE(!pw.on, "native FFT: spectra are taken on the unpadded grid (before the pad window is set)");
    MI_TRY(x_forward(s, vol));
    MI_TRY(y_pass(s, false, false));  // (k_z_conv<build> reads the plain [px][z][py] layout)
    const int Hx = dims.hx, M = dims.ny, L = dims.nz;
    const unsigned ztiles = (unsigned)((size_t)(Hx / 2 + 1) * (M / dims.tl));
    const size_t zl = lds_bytes(2 * dims.tl, L);
    const NativeDims d = dims;
    const float2* Tp = t_spec;
    float2* Sp = S.as<float2>();
    const float2* twz = tw_z;
    int rc = MI_ERR_INVALID;
#define MI_Z(LG, R)                                                                                                                       \
    case LG * 16 + R:                                                                                                                     \
        rc = launch_lds(k_z_conv<LG, R, true>, ztiles, kThreadsXZ, zl, s, "k_z_conv<build>", Tp, Sp, (const float4*)nullptr, d, twz, 0, Gp, scale); \
        break;
    switch (dims.lz2 * 16 + dims.r3z) { MI_Z_CASES(MI_Z) default: return fail(MI_ERR_UNSUPPORTED, "native FFT: z length %d", L); }
#undef MI_Z
    return rc;
}

// P2, P3, P4: S[z][px][py] -> T[z][px][py] (x still transformed), multiplied by the OTF or its conjugate
int NativeFft::middle(hipStream_t s, bool conj_otf) {
    MI_TRY(y_pass(s, false, dims.paired != 0));
    MI_TRY(z_conv(s, conj_otf));
    return y_pass(s, true, dims.paired != 0);
}

NativeFft::~NativeFft() {
    for (auto& e : alt_ev)
        if (e) (void)hipEventDestroy(e);
    if (vmm.va) {  // the spectrum arrays are a mapped range, not a pool block
        S.p = nullptr;
        S.bytes = 0;
        vmm.release();
    }
}

int VmmRange::alloc(size_t n, size_t chunk_bytes, int order) {
    release();
    int dev = 0;
    MI_HIP(hipGetDevice(&dev));
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    MI_HIP(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (gran == 0) gran = (size_t)2 << 20;
    chunk = std::max(gran, (chunk_bytes + gran - 1) / gran * gran);
    const size_t nchunks = (n + chunk - 1) / chunk;
    bytes = nchunks * chunk;
    MI_HIP(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
    h.assign(nchunks, hipMemGenericAllocationHandle_t{});
    for (size_t i = 0; i < nchunks; ++i) {
        hipError_t e = hipMemCreate(&h[i], chunk, &prop, 0);
        if (e != hipSuccess) {
            h.resize(i);
            release();
            return fail(MI_ERR_NOMEM, "hipMemCreate(%zu bytes) failed: %s", chunk, hipGetErrorString(e));
        }
    }
    // physical chunk perm[i] backs slot i of the range
    std::vector<size_t> perm(nchunks);
    for (size_t i = 0; i < nchunks; ++i) perm[i] = i;
    if (order == 1) std::reverse(perm.begin(), perm.end());
    if (order == 2) {  // bit-reversed positions (of the next power of two), the gaps closed
        size_t bits = 0;
        while (((size_t)1 << bits) < nchunks) ++bits;
        std::vector<std::pair<size_t, size_t>> key(nchunks);
        for (size_t i = 0; i < nchunks; ++i) {
            size_t r = 0;
            for (size_t b = 0; b < bits; ++b) r |= ((i >> b) & 1) << (bits - 1 - b);
            key[i] = {r, i};
        }
        std::sort(key.begin(), key.end());
        for (size_t i = 0; i < nchunks; ++i) perm[i] = key[i].second;
    }
    if (order == 3) {
        unsigned long long rng = 0x9E3779B97F4A7C15ull;
        for (size_t i = nchunks; i > 1; --i) {
            rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
            std::swap(perm[i - 1], perm[(size_t)(rng % i)]);
        }
    }
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (size_t i = 0; i < nchunks; ++i) MI_HIP(hipMemMap(static_cast<char*>(va) + i * chunk, chunk, 0, h[perm[i]], 0));
    MI_HIP(hipMemSetAccess(va, bytes, &acc, 1));
    mapped = true;
    return MI_OK;
}

void VmmRange::release() {
    if (va) {
        if (mapped) (void)hipMemUnmap(va, bytes);
        for (auto& hh : h) (void)hipMemRelease(hh);
        (void)hipMemAddressFree(va, bytes);
    }
    h.clear();
    va = nullptr;
    bytes = 0;
    mapped = false;
}

// P5 (+ P1 of the next convolution when fuse_forward): T -> out (may be null when fused) [-> S]
int NativeFft::x_inverse(hipStream_t s, float* out, int epi_kind, const ConvEpilogue& epi, bool fuse_forward, const TileSelect* part) {
    const int Hx = dims.hx, M = dims.ny, L = dims.nz;
    const unsigned xtiles = (unsigned)((size_t)L * (M / dims.ty));
    const size_t xl = lds_bytes(dims.ty, Hx);
    const NativeDims d = dims;
    const float2* Tp = t_spec;
    float2* Sp = S.as<float2>();
    const float2* twx = tw_x;
    const int ek = epi_kind == EPI_TAPER_SHELL ? EPI_NONE : epi_kind;
    MI_REQUIRE(ek == EPI_NONE || ek == EPI_RATIO || ek == EPI_UPDATE || ek == EPI_UPDATE_REG, "native FFT: unknown epilogue %d", epi_kind);
    MI_REQUIRE(!fuse_forward || ek == EPI_RATIO || ek == EPI_UPDATE, "native FFT: only the plain RL epilogues fuse");
    MI_REQUIRE(!fuse_forward || can_fuse(), "native FFT: a replicate-padded axis cannot fuse consecutive convolutions");
    const PadWindow w = pw;
    int rc = MI_ERR_INVALID;
    // padded grids go through the persistent kernel too when every padded axis follows the zero rule with the data at the origin
    // and the caller's rows are whole float4 groups
    const bool pad_pipe = pw.on && can_fuse() && pipe_ok() && pw.o[0] == 0 && pw.o[1] == 0 && pw.o[2] == 0 && pw.n[0] % 4 == 0 &&
                          ((uintptr_t)epi.a % 16) == 0 && ((uintptr_t)out % 16) == 0 && !(part && part->mode != 0);
    if (fuse_forward && (splits() || pad_pipe)) {
        TileSelect sel{};
        int per = M / dims.ty, planes = L;
        if (pad_pipe) {  // only the tiles that hold rows of the caller's volume
            sel.mode = 3;
            sel.n0 = (pw.n[1] + dims.ty - 1) / dims.ty;
            per = sel.n0;
            planes = pw.n[2];
        } else if (part && part->mode != 0) {
            sel = *part;
            per = sel.mode == 1 ? sel.n0 + sel.n1 : per - sel.n0 - sel.n1;
            if (sel.nz > 0) {
                MI_REQUIRE(sel.z0 >= 0 && sel.z0 + sel.nz <= L, "native FFT: plane range [%d, %d) outside [0, %d)", sel.z0, sel.z0 + sel.nz, L);
                planes = sel.nz;
            }
        }
        const int ntiles = planes * per;
        if (ntiles <= 0) return MI_OK;
        unsigned grid = 0;
        int* ctr_p = nullptr;
        MI_TRY(persistent_grid(s, ntiles, sel.mode == 2, &grid, &ctr_p));
#define MI_XP(LG, R) case LG * 16 + R: rc = launch_lds(k_x_fused_pipe<LG, R>, grid, kThreadsXZ, xl, s, "k_x_fused_pipe", Tp, out, epi, d, twx, Sp, ek, ntiles, sel, w, ctr_p); break;
        switch (dims.lhx2 * 16 + dims.r3x) { MI_AXIS_CASES(MI_XP) default: return fail(MI_ERR_UNSUPPORTED, "native FFT: x length %d", 2 * Hx); }
#undef MI_XP
        return rc;
    }
    MI_REQUIRE(!part || part->mode == 0, "native FFT: this kernel cannot run a subset of its tiles");
    if (!fuse_forward && (splits() || pad_pipe) && (ek == EPI_NONE || ek == EPI_RATIO || ek == EPI_UPDATE) && epi_kind != EPI_TAPER_SHELL &&
        out != nullptr && ((uintptr_t)out % 16) == 0 && ((uintptr_t)epi.a % 16) == 0 && std::getenv("MI_FFT_NO_XPIPE") == nullptr) {
        TileSelect sel{};
        int per = M / dims.ty, planes = L;
        if (pad_pipe) {  // only the tiles that hold rows of the caller's volume
            sel.mode = 3;
            sel.n0 = (pw.n[1] + dims.ty - 1) / dims.ty;
            per = sel.n0;
            planes = pw.n[2];
        }
        const int ntiles = planes * per;
        if (ntiles <= 0) return MI_OK;
        unsigned grid = 0;
        int* ctr_p = nullptr;
        MI_TRY(persistent_grid(s, ntiles, false, &grid, &ctr_p));
#define MI_XO(LG, R) case LG * 16 + R: rc = launch_lds(k_x_fused_pipe<LG, R, 2>, grid, kThreadsXZ, xl, s, "k_x_fused_pipe<inverse>", Tp, out, epi, d, twx, (float2*)nullptr, ek, ntiles, sel, w, ctr_p); break;
        switch (dims.lhx2 * 16 + dims.r3x) { MI_AXIS_CASES(MI_XO) default: return fail(MI_ERR_UNSUPPORTED, "native FFT: x length %d", 2 * Hx); }
#undef MI_XO
        return rc;
    }
#define MI_XI(LG, R)                                                                                                               \
    case LG * 16 + R:                                                                                                              \
        rc = fuse_forward ? launch_lds(k_x_inverse<LG, R, true>, xtiles, kThreadsXZ, xl, s, "k_x_inverse<fused>", Tp, out, epi, d, twx, Sp, ek, w) \
                          : launch_lds(k_x_inverse<LG, R, false>, xtiles, kThreadsXZ, xl, s, "k_x_inverse", Tp, out, epi, d, twx, Sp, ek, w);    \
        break;
    switch (dims.lhx2 * 16 + dims.r3x) { MI_AXIS_CASES(MI_XI) default: return fail(MI_ERR_UNSUPPORTED, "native FFT: x length %d", 2 * Hx); }
#undef MI_XI
    return rc;
}

int NativeFft::spectrum_rows(hipStream_t s, int y0, int rows, float2* buf, int dir, int z0, int nzc) {
    MI_REQUIRE(y0 >= 0 && rows > 0 && y0 + rows <= dims.ny, "spectrum rows [%d, %d) outside [0, %d)", y0, y0 + rows, dims.ny);
    MI_REQUIRE(dir == 2 || buf, "spectrum rows: null buffer");
    if (nzc <= 0) { z0 = 0; nzc = dims.nz; }
    MI_REQUIRE(z0 >= 0 && z0 + nzc <= dims.nz, "spectrum rows: planes [%d, %d) outside [0, %d)", z0, z0 + nzc, dims.nz);
    // lines (z, px) of the chunk: they keep their place in S and in the packed buffer [z * Hx + px][rows]
    const size_t line0 = (size_t)z0 * dims.hx, lines = (size_t)nzc * dims.hx, total = lines * (size_t)rows;
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_spectrum_rows, dim3((unsigned)blocks), dim3(256), 0, s, S.as<float2>() + line0 * dims.xrow,
                       buf ? buf + line0 * (size_t)rows : buf, lines, dims.xrow, y0, rows, dir);
    return launch_check("k_spectrum_rows");
}

// forward y pass of a range of z planes (the z-chunked halo exchange: a chunk's columns are transformed as soon as its halo rows
// have landed, while the later chunks still travel)
int NativeFft::y_forward_planes(hipStream_t s, int z0, int nzc) {
    const int Hx = dims.hx, M = dims.ny, L = dims.nz, gran = y_z_granule();
    MI_REQUIRE(z0 >= 0 && nzc > 0 && z0 + nzc <= L && z0 % gran == 0 && (nzc % gran == 0 || z0 + nzc == L),
               "native FFT: plane range [%d, %d) must be cut at multiples of %d", z0, z0 + nzc, gran);
    const bool paired = dims.paired != 0;
    if (!paired) MI_REQUIRE(((size_t)nzc * Hx) % dims.tc == 0 && ((size_t)z0 * Hx) % dims.tc == 0, "native FFT: plane range does not hold whole y tiles");
    const unsigned ycols = paired ? (unsigned)((size_t)(Hx / 2 + 1) * ((nzc + gran - 1) / gran)) : (unsigned)((size_t)nzc * Hx / dims.tc);
    const size_t yl = lds_bytes(dims.tc, M);
    NativeDims d = dims;
    d.yz0 = z0;
    d.z_in_hi = std::min(dims.z_in_hi, z0 + nzc);   // (work-groups of the last, partial granule stop here)
    const float2* src = S.as<float2>();
    float2* dst = t_spec;
    const float2* twy = tw_y;
    int rc = MI_ERR_INVALID;
#define MI_YZ(LG, R)                                                                                                     \
    case LG * 16 + R:                                                                                                    \
        rc = paired ? launch_lds(k_y_pair<LG, R, false>, ycols, kThreadsY, yl, s, "k_y_pair<fwd>", src, dst, d, twy)      \
                    : launch_lds(k_y_pass<LG, R, false>, ycols, kThreadsY, yl, s, "k_y_pass<fwd>", src, dst, d, twy);     \
        break;
    switch (dims.ly2 * 16 + dims.r3) { MI_Y_CASES(MI_YZ) default: return fail(MI_ERR_UNSUPPORTED, "native FFT: y length %d", M); }
#undef MI_YZ
    return rc;
}

// Average duration (ms) of one launch of a single pass, measured with HIP events on `s` (bench.py's roofline leg).
// which: 0 x forward, 1 y forward, 2 z convolution, 3 y inverse, 4 fused x inverse+ratio+forward, 5 fused x inverse+update+
// forward (this one overwrites bl with |bl .* c| of whatever the buffers hold).  The buffers
// keep whatever the previous convolution left in them; `bl` is only read.
int NativeFft::time_pass(hipStream_t s, int which, const float* bl, int reps, float* avg_ms) {
    MI_REQUIRE(reps > 0 && avg_ms && which >= 0 && which <= 5, "time_pass: bad arguments");
    hipEvent_t e0, e1;
    MI_HIP(hipEventCreate(&e0));
    MI_HIP(hipEventCreate(&e1));
    int rc = MI_OK;
    ConvEpilogue e;
    e.a = bl;
    for (int r = -1; r < reps && rc == MI_OK; ++r) {  // r == -1: warm-up launch
        if (r == 0) (void)hipEventRecord(e0, s);
        switch (which) {
            case 0: rc = x_forward(s, bl); break;
            case 4: rc = x_inverse(s, nullptr, EPI_RATIO, e, true); break;
            case 5: rc = x_inverse(s, const_cast<float*>(bl), EPI_UPDATE, e, true); break;
            // (blocked middle: the whole chain is quoted as pass 1, passes 2 and 3 do not exist on their own)
            case 1: rc = y_pass(s, false, dims.paired != 0); break;
            case 2: rc = z_conv(s, false); break;
            default: rc = y_pass(s, true, dims.paired != 0); break;
        }
    }
    (void)hipEventRecord(e1, s);
    hipError_t he = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc == MI_OK && he != hipSuccess) rc = fail(MI_ERR_HIP, "time_pass: %s", hipGetErrorString(he));
    *avg_ms = ms / (float)reps;
    return rc;
}

int NativeFft::time_between(hipStream_t s, int which, const float2* src, float2* dst, float* bl, int reps, float* avg_ms) {
    hipEvent_t e0, e1;
    MI_HIP(hipEventCreate(&e0));
    MI_HIP(hipEventCreate(&e1));
    int rc = MI_OK;
    void* const s_own = S.p;
    float2* const t_own = t_spec;
    ConvEpilogue ep;
    ep.a = bl;
    for (int r = -1; r < reps && rc == MI_OK; ++r) {
        if (r == 0) (void)hipEventRecord(e0, s);
        if (which == 0) {
            rc = y_pass(s, false, dims.paired != 0, src, dst);
        } else if (which == 2) {   // the forward x pass reads the volume and writes S
            S.p = dst;
            rc = x_forward(s, bl);
            S.p = s_own;
        } else if (which == 3) {   // the z pass reads T (and the OTF), writes S
            rc = z_conv(s, false, src, dst);
        } else if (which == 4) {   // the z pass on the context's own arrays with `src` standing in for the (real) OTF
            void* const g_own = Gr.p;
            if (!g_own) { rc = fail(MI_ERR_UNSUPPORTED, "time_between: no real OTF"); break; }
            Gr.p = const_cast<float2*>(src);
            rc = z_conv(s, false);
            Gr.p = g_own;
        } else {   // the update launch reads T and writes S: the two buffers stand in for them
            t_spec = const_cast<float2*>(src);
            S.p = dst;
            rc = x_inverse(s, bl, EPI_UPDATE, ep, true);
            S.p = s_own;
            t_spec = t_own;
        }
    }
    (void)hipEventRecord(e1, s);
    hipError_t he = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc == MI_OK && he != hipSuccess) rc = fail(MI_ERR_HIP, "time_between: %s", hipGetErrorString(he));
    *avg_ms = ms / (float)reps;
    return rc;
}

static int check_aligned(const void* p, const char* what) {
    MI_REQUIRE(p == nullptr || ((uintptr_t)p % 16) == 0, "native FFT: %s must be 16-byte aligned", what);
    return MI_OK;
}

int NativeFft::conv(hipStream_t s, const float* in, bool conj_otf, float* out, int epi_kind, const ConvEpilogue& epi) {
    if (!pw.on) {  // 16-byte accesses on the caller's volumes; the padded mode reads and writes them element-wise
        MI_TRY(check_aligned(in, "input"));
        MI_TRY(check_aligned(out, "output"));
        MI_TRY(check_aligned(epi.a, "epilogue operand"));
        MI_TRY(check_aligned(epi.b, "epilogue operand"));
    }
    MI_TRY(release_spare());
    MI_TRY(x_forward(s, in));
    MI_TRY(middle(s, conj_otf));
    return x_inverse(s, out, epi_kind, epi, false);
}

// n whole RL iterations (decon.m:162-186 with lambda = 0) in 8 passes each: the x passes of consecutive
// convolutions are fused, so per iteration bl is read twice and written once and the ratio never exists in HBM.
// Which of the two buffers kept for S goes with the CALLER's volume is settled on the first update launches of the fused loop
// themselves: the update launch is slow when the array it writes shares a memory region with the volume it rewrites, and only
// that launch shows it (the ratio launch and the forward x pass, which only read the volume, do not).  The first update launch
// that is followed by another iteration is timed writing the first buffer, the second one writing the other buffer -- every x
// launch writes S completely and the passes before it have consumed the old contents, so the buffer can change from one x launch
// to the next -- and the next x launch already goes to the faster of the two; the loser returns to the driver.
void NativeFft::settle_before_update() {
    if (alt_phase == 1) std::swap(S.p, S_alt.p);   // (the second buffer's turn)
}

int NativeFft::settle_decide(hipStream_t s) {
    (void)s;
    float t[2] = {0.0f, 0.0f};
    hipError_t he = hipEventSynchronize(alt_ev[3]);
    if (he == hipSuccess) he = hipEventElapsedTime(&t[0], alt_ev[0], alt_ev[1]);
    if (he == hipSuccess) he = hipEventElapsedTime(&t[1], alt_ev[2], alt_ev[3]);
    for (auto& e : alt_ev) { (void)hipEventDestroy(e); e = nullptr; }
    // now S.p is the second buffer, S_alt.p the first
    if (he != hipSuccess || t[0] <= 1.02f * t[1]) std::swap(S.p, S_alt.p);
    if (std::getenv("MI_FFT_PLACE_LOG"))
        std::fprintf(stderr, "native FFT: S settled on the %s buffer (update launch %.3f / %.3f ms with this volume)\n",
                     (he != hipSuccess || t[0] <= 1.02f * t[1]) ? "first" : "second", (double)t[0], (double)t[1]);
    (void)hipFree(S_alt.p);   // (waits for the device: the passes that still read it have run by then)
    S_alt.p = nullptr;
    S_alt.bytes = 0;
    alt_phase = 3;
    return he == hipSuccess ? MI_OK : fail(MI_ERR_HIP, "native FFT: settling S: %s", hipGetErrorString(he));
}

int NativeFft::release_spare() {
    if (!S_alt.p || alt_phase >= 3) return MI_OK;
    if (alt_phase == 2) return settle_decide(nullptr);   // (both update launches have been timed: keep the faster buffer)
    // phase 0 / 1: S.p is the first buffer, the second was never (or not yet) written by a launch whose output is still needed
    for (auto& e : alt_ev)
        if (e) { (void)hipEventDestroy(e); e = nullptr; }
    (void)hipFree(S_alt.p);   // (waits for the device)
    S_alt.p = nullptr;
    S_alt.bytes = 0;
    alt_phase = 3;
    return MI_OK;
}

int NativeFft::iterate(hipStream_t s, float* bl, int n_iters) {
    if (!pw.on) MI_TRY(check_aligned(bl, "bl"));
    MI_REQUIRE(can_fuse(), "native FFT: a replicate-padded axis cannot fuse consecutive convolutions");
    if (n_iters <= 0) return MI_OK;
    ConvEpilogue e;
    e.a = bl;
    if (S_alt.p && alt_phase == 2) MI_TRY(settle_decide(s));   // (see settle_before_update)
    MI_TRY(x_forward(s, bl));
    for (int it = 0; it < n_iters; ++it) {
        MI_TRY(middle(s, false));
        if (S_alt.p && alt_phase == 2) MI_TRY(settle_decide(s));
        MI_TRY(x_inverse(s, nullptr, EPI_RATIO, e, true));   // ratio = bl ./ max(c, eps) -> S, not stored
        MI_TRY(middle(s, true));
        const bool fuse = it + 1 < n_iters, timed = S_alt.p != nullptr && alt_phase < 2 && fuse && !pw.on;
        if (timed) {
            settle_before_update();
            if (!alt_ev[0]) for (auto& ev : alt_ev) MI_HIP(hipEventCreate(&ev));
            (void)hipEventRecord(alt_ev[2 * alt_phase], s);
        }
        MI_TRY(x_inverse(s, bl, EPI_UPDATE, e, fuse));  // bl = |bl .* a| (-> S for the next iteration)
        if (timed) {
            (void)hipEventRecord(alt_ev[2 * alt_phase + 1], s);
            ++alt_phase;
        }
    }
    return MI_OK;
}

}  // namespace mi
