// filter_subband_3d_z (LsDeconvolveMultiGPU/filter_subband_3d_z.m:1-123): log1p, db9 wavelet decomposition of every XZ slice
// ('sym' extension), Gaussian notch along z on the H sub-bands (high-pass in x, low-pass in z -> stripes running along z),
// reconstruction, expm1.  The reference loops over the Y slices (:23-27); the slices are independent and X is the unit-stride
// axis of the block, so here every pass runs over the whole [Z][Y][X] volume at once: the wavelet passes along z move
// unit-stride rows of x, the passes along x read a row with stride 2.  All passes are bandwidth-bound element-wise stencils.
//
//   analysis   out[i] = sum_t F[t] in[sym(2 i + 1 - t)]        (dwt2 'sym': extension by lf - 1, valid convolution, even
//                                                               1-based samples; floor((n + lf - 1) / 2) coefficients)
//   synthesis  out[j] = sum_k a[k] Lo_R[j + lf - 2 - 2 k] + d[k] Hi_R[j + lf - 2 - 2 k]   (dyadup, full convolution, centre)
//   notch      H <- real(ifft(fft(H, z) .* g)), g(k) = 1 - exp(-k^2 / (2 (sigma / n)^2))  (:92-123): evaluated as
//              H - (1/n) sum_k (1 - g(k)) F_k e^{2 pi i k z / n} over the bins whose weight is not zero in single precision --
//              with sigma / n << 1 that is the mean along z only
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

#include "mi_internal.h"
#include "mi_lsdeconv.h"

namespace mi {
namespace {

constexpr int kThreads = 256;
constexpr int LF = 18;  // db9

// Lo_R = sqrt(2) * dbwavf('db9'): the extremal-phase Daubechies filter with 9 vanishing moments (published table; the tests
// recompute it by spectral factorisation of the half-band polynomial, agreement 3e-11)
const double kLoR[LF] = {3.80779473638783381e-02,  2.43834674612590230e-01,  6.04823123690111153e-01,  6.57288078051299962e-01,
                         1.33197385825007591e-01,  -2.93273783279174305e-01, -9.68407832229760124e-02, 1.48540749338105876e-01,
                         3.07256814793338794e-02,  -6.76328290613307237e-02, 2.50947114831909018e-04,  2.23616621236789742e-02,
                         -4.72320475775138936e-03, -4.28150368246343286e-03, 1.84764688305622871e-03,  2.30385763523196190e-04,
                         -2.51963188942710503e-04, 3.93473203162716365e-05};

struct Filters {
    float lo_d[LF], hi_d[LF], lo_r[LF], hi_r[LF];
};

Filters make_filters() {
    Filters f;
    for (int t = 0; t < LF; ++t) {
        f.lo_r[t] = (float)kLoR[t];
        const double q = kLoR[LF - 1 - t] * ((t & 1) ? -1.0 : 1.0);  // qmf: reversed, the even (1-based) entries negated
        f.hi_r[t] = (float)q;
    }
    for (int t = 0; t < LF; ++t) {
        f.lo_d[t] = f.lo_r[LF - 1 - t];
        f.hi_d[t] = f.hi_r[LF - 1 - t];
    }
    return f;
}

__device__ __forceinline__ int sym_index(int j, int n) {  // half-point symmetric extension, any distance
    const int p = 2 * n;
    j %= p;
    if (j < 0) j += p;
    return j < n ? j : p - 1 - j;
}

inline unsigned grid_for(size_t n_items) {
    const size_t b = (n_items + kThreads - 1) / kThreads, cap = 256 * 32;
    return static_cast<unsigned>(b < 1 ? 1 : (b > cap ? cap : b));
}

// analysis along z: in [n_valid][rows] (rows = Y * nx, unit stride) -> lo, hi [m][rows].  One thread per column and z chunk,
// sliding an 18-sample window down the column: every input is read (and its log1p taken) once per chunk, two new samples per
// output.  Planes at and beyond n_valid are the zero padding to the even extent n (:50-51); LOG1P: the input is the raw block.
// With w[q] = x[2 i - 16 + q]: out[i] = sum_t F_D[t] x[2 i + 1 - t] = sum_q F_R[q] w[q]  (F_D is F_R reversed)
template <int V> struct VecOf;
template <> struct VecOf<1> { using type = float; };
template <> struct VecOf<4> { using type = float4; };
__device__ __forceinline__ float vget(const float& v, int) { return v; }
__device__ __forceinline__ float& vref(float& v, int) { return v; }
__device__ __forceinline__ float vget(const float4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }
__device__ __forceinline__ float& vref(float4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }

// V = 4: a lane owns four neighbouring columns (16-byte accesses; rows % 4 == 0 and aligned bases), which keeps enough bytes
// in flight per lane for the two loads per step to cover the HBM latency
template <bool LOG1P, int V>
__global__ __launch_bounds__(kThreads) void k_dwt_z(const float* __restrict__ in, float* __restrict__ lo, float* __restrict__ hi,
                                                   size_t rows, int n, int n_valid, int m, int chunk, Filters f) {
    using T = typename VecOf<V>::type;
    const size_t r = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (r >= rows) return;
    const int i0 = blockIdx.y * chunk, i1 = min(m, i0 + chunk);
    auto raw = [&](int j) {  // zero padding: log1p(0) = 0
        const int jj = sym_index(j, n);
        return jj < n_valid ? *reinterpret_cast<const T*>(in + (size_t)jj * rows + r) : T();
    };
    auto xf = [&](T v) {
        if (LOG1P) {
#pragma unroll
            for (int c = 0; c < V; ++c) vref(v, c) = log1pf(vget(v, c));
        }
        return v;
    };
    T w[LF];
#pragma unroll
    for (int q = 2; q < LF; ++q) w[q] = xf(raw(2 * i0 - 16 + q - 2));  // the first step shifts these into place
    T n0 = raw(2 * i0), n1 = raw(2 * i0 + 1);  // the two new samples of a step are requested one step ahead
    for (int i = i0; i < i1; ++i) {
#pragma unroll
        for (int q = 0; q < LF - 2; ++q) w[q] = w[q + 2];
        w[LF - 2] = xf(n0);
        w[LF - 1] = xf(n1);
        if (i + 1 < i1) {
            n0 = raw(2 * i + 2);
            n1 = raw(2 * i + 3);
        }
        T al = T(), ah = T();
#pragma unroll
        for (int q = 0; q < LF; ++q) {
#pragma unroll
            for (int c = 0; c < V; ++c) {
                vref(al, c) += f.lo_r[q] * vget(w[q], c);
                vref(ah, c) += f.hi_r[q] * vget(w[q], c);
            }
        }
        *reinterpret_cast<T*>(lo + (size_t)i * rows + r) = al;
        *reinterpret_cast<T*>(hi + (size_t)i * rows + r) = ah;
    }
}

constexpr int kTileX = kThreads;       // pairs of the x synthesis per work-group (one per lane)
constexpr int kTileA = 2 * kThreads;   // outputs of the x analysis per work-group (two per lane)

// analysis along x: in [lines][n_pitch] -> lo, hi [lines][m].  One work-group per line segment: the 2 * kTileA + 16 inputs of
// kTileA outputs are staged in LDS with unit-stride loads (columns at and beyond n_valid are the zero padding); a lane takes
// the 20 inputs of its two outputs as five 16-byte LDS reads
__global__ __launch_bounds__(kThreads) void k_dwt_x(const float* __restrict__ in, float* __restrict__ lo, float* __restrict__ hi,
                                                   int n, int n_valid, int n_pitch, int m, int tiles, Filters f) {
    __shared__ __attribute__((aligned(16))) float seg[2 * kTileA + 20];
    const size_t line = blockIdx.x / tiles;
    const int i0 = (int)(blockIdx.x - line * tiles) * kTileA;
    const float* row = in + line * (size_t)n_pitch;
    const int j0 = 2 * i0 - 16;                                          // seg[q] = x[j0 + q]
    if (j0 >= 0 && j0 + 2 * kTileA + 20 <= n_valid) {                    // interior segment: no reflection, no padding
        for (int q = threadIdx.x; q < 2 * kTileA + 20; q += kThreads) seg[q] = row[j0 + q];
    } else {
        for (int q = threadIdx.x; q < 2 * kTileA + 20; q += kThreads) {
            const int j = sym_index(j0 + q, n);
            seg[q] = j < n_valid ? row[j] : 0.0f;
        }
    }
    __syncthreads();
    const int i = i0 + 2 * threadIdx.x;
    if (i >= m) return;
    float w[20];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(&seg[4 * threadIdx.x + 4 * q]);
        w[4 * q] = v.x;
        w[4 * q + 1] = v.y;
        w[4 * q + 2] = v.z;
        w[4 * q + 3] = v.w;
    }
    float al0 = 0.0f, ah0 = 0.0f, al1 = 0.0f, ah1 = 0.0f;
#pragma unroll
    for (int q = 0; q < LF; ++q) {
        al0 += f.lo_r[q] * w[q];
        ah0 += f.hi_r[q] * w[q];
        al1 += f.lo_r[q] * w[q + 2];
        ah1 += f.hi_r[q] * w[q + 2];
    }
    const size_t o = line * (size_t)m + i;
    lo[o] = al0;
    hi[o] = ah0;
    if (i + 1 < m) {
        lo[o + 1] = al1;
        hi[o + 1] = ah1;
    }
}

// synthesis along x: a, d [lines][m] -> out [lines][s].  Both outputs of a pair (2 p, 2 p + 1) use a[p .. p + 8], d[p .. p + 8]
// (taps 2 (8 - c) and 2 (8 - c) + 1 for k = p + c); coefficients beyond m do not exist (zero)
__global__ __launch_bounds__(kThreads) void k_idwt_x(const float* __restrict__ a, const float* __restrict__ d, float* __restrict__ out,
                                                    int m, int s, int tiles, Filters f) {
    __shared__ float sa[kTileX + 8], sd[kTileX + 8];
    const size_t line = blockIdx.x / tiles;
    const int p0 = (int)(blockIdx.x - line * tiles) * kTileX;
    for (int q = threadIdx.x; q < kTileX + 8; q += kThreads) {
        const int k = p0 + q;
        sa[q] = k < m ? a[line * (size_t)m + k] : 0.0f;
        sd[q] = k < m ? d[line * (size_t)m + k] : 0.0f;
    }
    __syncthreads();
    const int p = p0 + threadIdx.x;
    if (2 * p >= s) return;
    float e = 0.0f, o = 0.0f;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const float va = sa[threadIdx.x + c], vd = sd[threadIdx.x + c];
        e += va * f.lo_r[2 * (8 - c)] + vd * f.hi_r[2 * (8 - c)];
        o += va * f.lo_r[2 * (8 - c) + 1] + vd * f.hi_r[2 * (8 - c) + 1];
    }
    float* dst = out + line * (size_t)s + 2 * p;
    dst[0] = e;
    if (2 * p + 1 < s) dst[1] = o;
}

// synthesis along z: a, d [m][rows] -> out [s][rows_out]; one thread per output column and z chunk, a 9-coefficient window of
// a and of d slides down the column.  EXPM1: the final level writes expm1 into the block, whose rows are narrower than the
// even-padded working rows when nx is odd (crop, :86-88)
template <bool EXPM1>
__global__ __launch_bounds__(kThreads) void k_idwt_z(const float* __restrict__ a, const float* __restrict__ d, float* __restrict__ out,
                                                    int ny, int nx_work, int nx_out, int m, int s, int chunk, Filters f) {
    const size_t rows_out = (size_t)ny * nx_out, rows = (size_t)ny * nx_work;
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows_out) return;
    const size_t y = r / nx_out, x = r - y * nx_out;
    const size_t rw = y * nx_work + x;
    const int p0 = blockIdx.y * chunk, p1 = min((s + 1) / 2, p0 + chunk);
    float wa[9], wd[9];
#pragma unroll
    for (int c = 1; c < 9; ++c) {
        const int k = p0 + c - 1;
        wa[c] = k < m ? a[(size_t)k * rows + rw] : 0.0f;
        wd[c] = k < m ? d[(size_t)k * rows + rw] : 0.0f;
    }
    auto coef = [&](const float* __restrict__ src, int k) { return k < m ? src[(size_t)k * rows + rw] : 0.0f; };
    float na = coef(a, p0 + 8), nd = coef(d, p0 + 8);  // requested one step ahead
    for (int p = p0; p < p1; ++p) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            wa[c] = wa[c + 1];
            wd[c] = wd[c + 1];
        }
        wa[8] = na;
        wd[8] = nd;
        if (p + 1 < p1) {
            na = coef(a, p + 9);
            nd = coef(d, p + 9);
        }
        float e = 0.0f, o = 0.0f;
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            e += wa[c] * f.lo_r[2 * (8 - c)] + wd[c] * f.hi_r[2 * (8 - c)];
            o += wa[c] * f.lo_r[2 * (8 - c) + 1] + wd[c] * f.hi_r[2 * (8 - c) + 1];
        }
        out[(size_t)(2 * p) * rows_out + r] = EXPM1 ? expm1f(e) : e;
        if (2 * p + 1 < s) out[(size_t)(2 * p + 1) * rows_out + r] = EXPM1 ? expm1f(o) : o;
    }
}

// z chunking of the column kernels: enough threads for the device when a level has few columns
inline int z_chunks(size_t columns, int steps) {
    const size_t want = 256 * 2048;  // lanes in flight on 256 CUs
    size_t c = columns >= want ? 1 : (want + columns - 1) / columns;
    const size_t most = (size_t)std::max(1, steps / 16);
    return (int)std::min(c, most);
}

// notch along z on H [n][rows]: one frequency bin per launch, one thread per column.  F_k in double; the bin's share of the
// filtered-out part is w / n * (Re - Im)(F_k e^{+i theta z}): the reference multiplies the spectrum by complex(g, g) =
// g (1 + i) and keeps the real part (:112-114).  For bins that come in conjugate pairs of equal weight (every bin of an even
// length) the Im parts cancel.  MODE 0: H -= share (a single bin: nothing else reads H afterwards); 1: corr = share;
// 2: corr += share (several bins: all of them transform the ORIGINAL H, the sum is subtracted at the end)
template <int MODE>
__global__ __launch_bounds__(kThreads) void k_notch_bin(float* __restrict__ H, float* __restrict__ corr, size_t rows, int n, int k, double w) {
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (size_t)gridDim.x * blockDim.x) {
        double re = 0.0, im = 0.0;
        for (int z = 0; z < n; ++z) {
            double sn = 0.0, cs = 1.0;
            if (k) sincospi(2.0 * (double)(((long long)k * z) % n) / (double)n, &sn, &cs);
            const double v = (double)H[(size_t)z * rows + r];
            re += v * cs;  // F_k = sum v e^{-i theta}
            im -= v * sn;
        }
        const double sc = w / (double)n;
        for (int z = 0; z < n; ++z) {
            double sn = 0.0, cs = 1.0;
            if (k) sincospi(2.0 * (double)(((long long)k * z) % n) / (double)n, &sn, &cs);
            const float share = (float)(sc * ((re * cs - im * sn) - (re * sn + im * cs)));
            const size_t i = (size_t)z * rows + r;
            if (MODE == 0) H[i] -= share;
            else if (MODE == 1) corr[i] = share;
            else corr[i] += share;
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_subtract(float* __restrict__ H, const float* __restrict__ corr, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) H[i] -= corr[i];
}

// `corr`: scratch of n * rows floats, used when the notch spans more than one bin
int notch_z(hipStream_t s, float* H, float* corr, size_t rows, int n, double sigma) {
    // g as the reference builds it (:117-123): single precision, x = (0:n-1) - floor(n/2), then fftshift = circular shift by
    // floor(n/2): bin k holds x = ((k - floor(n/2)) mod n) - floor(n/2).  Even n: x = k below n/2, k - n above (zero at DC).
    // Odd n: x = k + 1 below floor(n/2), k - (n - 1) above -- the zero of the notch lands on the last bin (frequency -1), a
    // quirk of fftshift on odd lengths that is kept.  Weight of the subtracted part = 1 - g
    sigma = std::max(sigma, (double)kEpsSingle);
    const float two_s2 = (float)(2.0 * sigma * sigma);
    std::vector<std::pair<int, double>> bins;
    for (int k = 0; k < n; ++k) {
        const int h = n / 2, xk = ((k - h) % n + n) % n - h;
        const float x = (float)xk;
        const float g = 1.0f - expf(-(x * x) / two_s2);
        const double w = 1.0 - (double)g;
        if (w != 0.0) bins.emplace_back(k, w);
    }
    if (bins.empty()) return MI_OK;
    const dim3 grid(grid_for(rows)), block(kThreads);
    if (bins.size() == 1) {
        hipLaunchKernelGGL(k_notch_bin<0>, grid, block, 0, s, H, corr, rows, n, bins[0].first, bins[0].second);
        return launch_check("k_notch_bin");
    }
    for (size_t b = 0; b < bins.size(); ++b) {
        if (b == 0) hipLaunchKernelGGL(k_notch_bin<1>, grid, block, 0, s, H, corr, rows, n, bins[b].first, bins[b].second);
        else hipLaunchKernelGGL(k_notch_bin<2>, grid, block, 0, s, H, corr, rows, n, bins[b].first, bins[b].second);
        MI_TRY(launch_check("k_notch_bin"));
    }
    hipLaunchKernelGGL(k_subtract, dim3(grid_for(rows * (size_t)n)), block, 0, s, H, corr, rows * (size_t)n);
    return launch_check("k_subtract");
}

int wmaxlev(int a, int b) {  // fix(log2(min(size) / (lf - 1)))
    const int m = std::min(a, b);
    int lev = 0;
    while ((LF - 1) * (2 << lev) <= m) ++lev;
    return lev;
}

}  // namespace
}  // namespace mi

using namespace mi;

extern "C" int mi_destripe_max_levels(int nx, int nz) { return wmaxlev(nx + (nx & 1), nz + (nz & 1)); }

extern "C" int mi_destripe_z(int dev, void* stream, float* bl, int nx, int ny, int nz, float sigma, int levels) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(bl, "filter_subband_3d_z: null pointer");
    MI_REQUIRE(nx > 0 && ny > 0 && nz > 0, "filter_subband_3d_z: bl must be 3D and non-empty");
    MI_REQUIRE(levels >= 0, "filter_subband_3d_z: levels must be >= 0");
    hipStream_t s = as_stream(stream);
    const int px = nx + (nx & 1), pz = nz + (nz & 1);  // even extents (:50-51), zeros appended
    if (levels == 0) levels = wmaxlev(px, pz);         // :54-56
    const Filters f = make_filters();
    std::vector<int> sx(levels + 1), sz(levels + 1);
    sx[0] = px;
    sz[0] = pz;
    for (int l = 1; l <= levels; ++l) {
        sx[l] = (sx[l - 1] + LF - 1) / 2;
        sz[l] = (sz[l - 1] + LF - 1) / 2;
    }
    const size_t Y = (size_t)ny;
    if (levels == 0) {  // wavedec2 with zero levels: the transform is the identity, and so is the filter
        return MI_OK;
    }
    // per level: A, H, V, D [sz_l][Y][sx_l]; the z-analysis halves ZL, ZH [sz_l][Y][sx_{l-1}] are shared by all levels
    std::vector<DevBuf> A(levels + 1), H(levels + 1), V(levels + 1), D(levels + 1);
    DevBuf ZL, ZH;
    MI_TRY(ZL.alloc(sizeof(float) * (size_t)sz[1] * Y * sx[0]));
    MI_TRY(ZH.alloc(sizeof(float) * (size_t)sz[1] * Y * sx[0]));
    for (int l = 1; l <= levels; ++l) {
        const size_t bytes = sizeof(float) * (size_t)sz[l] * Y * sx[l];
        MI_TRY(A[l].alloc(bytes));
        MI_TRY(H[l].alloc(bytes));
        MI_TRY(V[l].alloc(bytes));
        MI_TRY(D[l].alloc(bytes));
    }
    // ---- decomposition (wavedec2: dim 2 = z first, then dim 1 = x; dwt2.m)
    for (int l = 1; l <= levels; ++l) {
        const int nzl = sz[l - 1], nxl = sx[l - 1], mz = sz[l], mx = sx[l];
        const size_t lines = (size_t)mz * Y;
        // level 1 reads the block itself: rows are nx wide (no x padding stored), planes beyond nz are the zero padding
        const int pitch = l == 1 ? nx : nxl;
        const size_t cols = Y * (size_t)pitch;
        const int zc = z_chunks(cols, mz), zchunk = (mz + zc - 1) / zc;
        const float* zin = l == 1 ? bl : A[l - 1].as<float>();
        const bool vec4 = cols % 4 == 0 && ((uintptr_t)zin % 16) == 0;  // pool blocks are 256-byte aligned
        const size_t lanes = vec4 ? cols / 4 : cols;
        const dim3 zgrid((unsigned)((lanes + kThreads - 1) / kThreads), (unsigned)((mz + zchunk - 1) / zchunk));
        const int zvalid = l == 1 ? nz : nzl;
#define MI_DWT_Z(LOG, VW) \
    hipLaunchKernelGGL((k_dwt_z<LOG, VW>), zgrid, dim3(kThreads), 0, s, zin, ZL.as<float>(), ZH.as<float>(), cols, nzl, zvalid, mz, zchunk, f)
        if (l == 1) { if (vec4) MI_DWT_Z(true, 4); else MI_DWT_Z(true, 1); }
        else { if (vec4) MI_DWT_Z(false, 4); else MI_DWT_Z(false, 1); }
#undef MI_DWT_Z
        MI_TRY(launch_check("k_dwt_z"));
        const int tiles = (mx + kTileA - 1) / kTileA;
        MI_REQUIRE(lines * (size_t)tiles < 0x7fffffffull, "filter_subband_3d_z: block too large for one launch");
        const dim3 xgrid((unsigned)(lines * tiles));
        hipLaunchKernelGGL(k_dwt_x, xgrid, dim3(kThreads), 0, s, ZL.as<float>(), A[l].as<float>(), H[l].as<float>(), nxl, pitch, pitch, mx,
                           tiles, f);
        MI_TRY(launch_check("k_dwt_x"));
        hipLaunchKernelGGL(k_dwt_x, xgrid, dim3(kThreads), 0, s, ZH.as<float>(), V[l].as<float>(), D[l].as<float>(), nxl, pitch, pitch, mx,
                           tiles, f);
        MI_TRY(launch_check("k_dwt_x"));
        // horizontal details of this level: notch along z with sigma / size(H, 2) (:69-73)
        MI_TRY(notch_z(s, H[l].as<float>(), ZL.as<float>(), Y * mx, mz, (double)sigma / (double)mz));  // ZL is free again
    }
    // ---- reconstruction (waverec2), coarsest level first
    for (int l = levels; l >= 1; --l) {
        const int mz = sz[l], mx = sx[l], sxo = sx[l - 1], szo = sz[l - 1];
        const size_t lines = (size_t)mz * Y;
        const int tiles = ((sxo + 1) / 2 + kTileX - 1) / kTileX;
        const dim3 xgrid((unsigned)(lines * tiles));
        hipLaunchKernelGGL(k_idwt_x, xgrid, dim3(kThreads), 0, s, A[l].as<float>(), H[l].as<float>(), ZL.as<float>(), mx, sxo, tiles, f);
        MI_TRY(launch_check("k_idwt_x"));
        hipLaunchKernelGGL(k_idwt_x, xgrid, dim3(kThreads), 0, s, V[l].as<float>(), D[l].as<float>(), ZH.as<float>(), mx, sxo, tiles, f);
        MI_TRY(launch_check("k_idwt_x"));
        const int s_out = l == 1 ? nz : szo, nx_out = l == 1 ? nx : sxo;  // level 1: crop the even padding (:86-88), expm1 (:30)
        const size_t cols = Y * (size_t)nx_out;
        const int pairs = (s_out + 1) / 2, zc = z_chunks(cols, pairs), zchunk = (pairs + zc - 1) / zc;
        const dim3 zgrid((unsigned)((cols + kThreads - 1) / kThreads), (unsigned)((pairs + zchunk - 1) / zchunk));
        if (l == 1)
            hipLaunchKernelGGL(k_idwt_z<true>, zgrid, dim3(kThreads), 0, s, ZL.as<float>(), ZH.as<float>(), bl, ny, sxo, nx_out, mz, s_out,
                               zchunk, f);
        else
            hipLaunchKernelGGL(k_idwt_z<false>, zgrid, dim3(kThreads), 0, s, ZL.as<float>(), ZH.as<float>(), A[l - 1].as<float>(), ny, sxo,
                               nx_out, mz, s_out, zchunk, f);
        MI_TRY(launch_check("k_idwt_z"));
    }
    MI_HIP(hipStreamSynchronize(s));  // the work buffers die here
    return MI_OK;
}
