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

#include <cmath>
#include <utility>
#include <vector>

#include "mi_internal.h"
#include "mi_lsdeconv.h"

namespace mi {
namespace {

constexpr int kThreads = 256;
constexpr int LF = 18;  // db9

// Lo_R = sqrt(2) * dbwavf('db9'): the extremal-phase Daubechies filter with 9 vanishing moments (published table; recomputed
// by spectral factorisation in oracle/destripe_oracle.py::db_filters, agreement 3e-11)
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

// analysis along z: in [nz_in][rows] (rows = Y * nx, unit stride) -> lo, hi [mz][rows].  n_valid <= nz_in: planes at and beyond
// n_valid are the zero padding to an even extent (:50-51); LOG1P: the input is the raw block (level 1)
template <bool LOG1P>
__global__ __launch_bounds__(kThreads) void k_dwt_z(const float* __restrict__ in, float* __restrict__ lo, float* __restrict__ hi,
                                                   size_t rows, int n, int n_valid, int m, Filters f) {
    const size_t total = rows * (size_t)m;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx / rows);
        const size_t r = idx - (size_t)i * rows;
        float al = 0.0f, ah = 0.0f;
#pragma unroll
        for (int t = 0; t < LF; ++t) {
            const int j = sym_index(2 * i + 1 - t, n);
            float v = 0.0f;
            if (j < n_valid) {
                v = in[(size_t)j * rows + r];
                if (LOG1P) v = log1pf(v);
            }
            al += f.lo_d[t] * v;
            ah += f.hi_d[t] * v;
        }
        lo[idx] = al;
        hi[idx] = ah;
    }
}

// analysis along x: in [lines][n] -> lo, hi [lines][m]; columns at and beyond n_valid (of the stored row pitch n_pitch) are zero
// padding.  LOG1P never applies here (z runs first)
__global__ __launch_bounds__(kThreads) void k_dwt_x(const float* __restrict__ in, float* __restrict__ lo, float* __restrict__ hi,
                                                   size_t lines, int n, int n_valid, int n_pitch, int m, Filters f) {
    const size_t total = lines * (size_t)m;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t line = idx / m;
        const int i = (int)(idx - line * m);
        const float* row = in + line * n_pitch;
        float al = 0.0f, ah = 0.0f;
#pragma unroll
        for (int t = 0; t < LF; ++t) {
            const int j = sym_index(2 * i + 1 - t, n);
            const float v = j < n_valid ? row[j] : 0.0f;
            al += f.lo_d[t] * v;
            ah += f.hi_d[t] * v;
        }
        lo[idx] = al;
        hi[idx] = ah;
    }
}

// synthesis along x: a, d [lines][m] -> out [lines][s]
__global__ __launch_bounds__(kThreads) void k_idwt_x(const float* __restrict__ a, const float* __restrict__ d, float* __restrict__ out,
                                                    size_t lines, int m, int s, Filters f) {
    const size_t total = lines * (size_t)s;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t line = idx / s;
        const int j = (int)(idx - line * s);
        const float* ra = a + line * m;
        const float* rd = d + line * m;
        // taps t = j + LF - 2 - 2k in [0, LF): k from ceil((j - 1) / 2) to floor((j + LF - 2) / 2)
        const int k0 = j <= 0 ? 0 : j / 2, k1 = min(m - 1, (j + LF - 2) / 2);
        float acc = 0.0f;
        for (int k = k0; k <= k1; ++k) {
            const int t = j + LF - 2 - 2 * k;
            acc += ra[k] * f.lo_r[t] + rd[k] * f.hi_r[t];
        }
        out[idx] = acc;
    }
}

// synthesis along z: a, d [m][rows] -> out [s_out][rows_out]; EXPM1: the final level writes expm1 into the block, whose rows are
// narrower than the even-padded working rows when nx is odd (crop, :86-88)
template <bool EXPM1>
__global__ __launch_bounds__(kThreads) void k_idwt_z(const float* __restrict__ a, const float* __restrict__ d, float* __restrict__ out,
                                                    int ny, int nx_work, int nx_out, int m, int s, Filters f) {
    const size_t rows_out = (size_t)ny * nx_out, rows = (size_t)ny * nx_work;
    const size_t total = rows_out * (size_t)s;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx / rows_out);
        const size_t r = idx - (size_t)j * rows_out;
        const size_t y = r / nx_out, x = r - y * nx_out;
        const size_t rw = y * nx_work + x;
        const int k0 = j <= 0 ? 0 : j / 2, k1 = min(m - 1, (j + LF - 2) / 2);
        float acc = 0.0f;
        for (int k = k0; k <= k1; ++k) {
            const int t = j + LF - 2 - 2 * k;
            acc += a[(size_t)k * rows + rw] * f.lo_r[t] + d[(size_t)k * rows + rw] * f.hi_r[t];
        }
        out[idx] = EXPM1 ? expm1f(acc) : acc;
    }
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
        if (l == 1) {
            // the block's rows are nx wide (no x padding stored): z analysis over [nz][Y * nx], the pad column appears in x
            hipLaunchKernelGGL(k_dwt_z<true>, dim3(grid_for((size_t)mz * Y * nx)), dim3(kThreads), 0, s, bl, ZL.as<float>(), ZH.as<float>(),
                               Y * nx, nzl, nz, mz, f);
        } else {
            hipLaunchKernelGGL(k_dwt_z<false>, dim3(grid_for((size_t)mz * Y * nxl)), dim3(kThreads), 0, s, A[l - 1].as<float>(),
                               ZL.as<float>(), ZH.as<float>(), Y * nxl, nzl, nzl, mz, f);
        }
        MI_TRY(launch_check("k_dwt_z"));
        const int valid = l == 1 ? nx : nxl, pitch = l == 1 ? nx : nxl;
        hipLaunchKernelGGL(k_dwt_x, dim3(grid_for(lines * mx)), dim3(kThreads), 0, s, ZL.as<float>(), A[l].as<float>(), H[l].as<float>(),
                           lines, nxl, valid, pitch, mx, f);
        MI_TRY(launch_check("k_dwt_x"));
        hipLaunchKernelGGL(k_dwt_x, dim3(grid_for(lines * mx)), dim3(kThreads), 0, s, ZH.as<float>(), V[l].as<float>(), D[l].as<float>(),
                           lines, nxl, valid, pitch, mx, f);
        MI_TRY(launch_check("k_dwt_x"));
        // horizontal details of this level: notch along z with sigma / size(H, 2) (:69-73)
        MI_TRY(notch_z(s, H[l].as<float>(), ZL.as<float>(), Y * mx, mz, (double)sigma / (double)mz));  // ZL is free again
    }
    // ---- reconstruction (waverec2), coarsest level first
    for (int l = levels; l >= 1; --l) {
        const int mz = sz[l], mx = sx[l], sxo = sx[l - 1], szo = sz[l - 1];
        const size_t lines = (size_t)mz * Y;
        hipLaunchKernelGGL(k_idwt_x, dim3(grid_for(lines * sxo)), dim3(kThreads), 0, s, A[l].as<float>(), H[l].as<float>(), ZL.as<float>(),
                           lines, mx, sxo, f);
        MI_TRY(launch_check("k_idwt_x"));
        hipLaunchKernelGGL(k_idwt_x, dim3(grid_for(lines * sxo)), dim3(kThreads), 0, s, V[l].as<float>(), D[l].as<float>(), ZH.as<float>(),
                           lines, mx, sxo, f);
        MI_TRY(launch_check("k_idwt_x"));
        if (l == 1) {  // into the block: crop the even padding (:86-88), expm1 (:30)
            hipLaunchKernelGGL(k_idwt_z<true>, dim3(grid_for((size_t)nz * Y * nx)), dim3(kThreads), 0, s, ZL.as<float>(), ZH.as<float>(), bl,
                               ny, sxo, nx, mz, nz, f);
        } else {
            hipLaunchKernelGGL(k_idwt_z<false>, dim3(grid_for((size_t)szo * Y * sxo)), dim3(kThreads), 0, s, ZL.as<float>(), ZH.as<float>(),
                               A[l - 1].as<float>(), ny, sxo, sxo, mz, szo, f);
        }
        MI_TRY(launch_check("k_idwt_z"));
    }
    MI_HIP(hipStreamSynchronize(s));  // the work buffers die here
    return MI_OK;
}
