// rocFFT convolution engine (MI_ENGINE_FFT): c = IFFT3(FFT3(x) .* OTF) on R2C half spectra.
//
// Replaces the reference's full complex fftn / .* / ifftn / real chain (decon.m:162-172: four C2C
// transforms per iteration on complex single buffers plus ~8 element-wise passes and two conj(otf)
// materialisations, :168,174) with: R2C -> one fused multiply (conj folded in as a flag, 1/N folded
// into the OTF) -> C2R -> one fused epilogue (RL ratio or RL update).  The same engine serves
//   * deconFFT semantics: circular on fft_shape, PSF placed as ifftshift(zero-pad-centre(psf))
//     (decon.m:131-133, supplements/otf_gpu.cu:36-67) -- including the one-voxel offset that
//     placement has for even fft_shape;
//   * convn(...,'same') / conv3d_gpu semantics for large PSFs: the volume is staged into a padded
//     buffer (zeros or clamped samples) large enough that circular wrap never reaches the output.
#include <algorithm>
#include <cmath>
#include <mutex>
#include <vector>

#include <cstdlib>

#include "fftconv.h"

namespace mi {

#define MI_FFT(call)                                                                                     \
    do {                                                                                                 \
        rocfft_status st_ = (call);                                                                      \
        if (st_ != rocfft_status_success)                                                                \
            return ::mi::fail(MI_ERR_FFT, "%s:%d: %s failed with rocfft_status %d", __FILE__, __LINE__,  \
                              #call, (int)st_);                                                          \
    } while (0)

int rocfft_global_setup() {
    static std::once_flag once;
    static rocfft_status st = rocfft_status_success;
    std::call_once(once, [] { st = rocfft_setup(); });
    if (st != rocfft_status_success) return fail(MI_ERR_FFT, "rocfft_setup failed with status %d", (int)st);
    return MI_OK;
}

namespace {

constexpr int kThreads = 256;
inline unsigned stream_grid(size_t n_items) {
    size_t b = (n_items + kThreads - 1) / kThreads;
    const size_t cap = 256 * 16;
    return static_cast<unsigned>(b < 1 ? 1 : (b > cap ? cap : b));
}

// circular kernel image: dst[p] = psf[j] where p == (j - shift) mod F per axis, else 0.  The image is a few thousand samples
// in a grid of up to billions: one memset, then one lane per PSF sample (k <= F: the targets are distinct)
__global__ __launch_bounds__(kThreads) void k_scatter_psf(const float* __restrict__ psf, float* __restrict__ dst, int kx, int ky, int kz,
                                                           int Fx, int Fy, int Fz, int sx, int sy, int sz) {
    const int total = kx * ky * kz;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int jx = i % kx, r = i / kx, jy = r % ky, jz = r / ky;
        const int x = ((jx - sx) % Fx + Fx) % Fx, y = ((jy - sy) % Fy + Fy) % Fy, z = ((jz - sz) % Fz + Fz) % Fz;
        dst[((size_t)z * Fy + y) * Fx + x] = psf[i];
    }
}

int place_psf(hipStream_t s, const float* psf, float* dst, int kx, int ky, int kz, int Fx, int Fy, int Fz, int sx, int sy, int sz) {
    MI_HIP(hipMemsetAsync(dst, 0, sizeof(float) * (size_t)Fx * Fy * Fz, s));
    hipLaunchKernelGGL(k_scatter_psf, dim3(stream_grid((size_t)kx * ky * kz)), dim3(kThreads), 0, s, psf, dst, kx, ky, kz, Fx, Fy, Fz, sx, sy, sz);
    return launch_check("k_scatter_psf");
}

__global__ __launch_bounds__(kThreads) void k_scale_c(float2* __restrict__ a, size_t n, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float2 v = a[i];
        a[i] = make_float2(v.x * scale, v.y * scale);
    }
}

// spec .*= otf  or  spec .*= conj(otf)
template <bool CONJ>
__global__ __launch_bounds__(kThreads) void k_mul_otf(float2* __restrict__ spec, const float2* __restrict__ otf, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float2 a = spec[i], b = otf[i];
        const float bi = CONJ ? -b.y : b.y;
        spec[i] = make_float2(a.x * b.x - a.y * bi, a.x * bi + a.y * b.x);
    }
}

// padded staging: dst (F) <- src (n) at offset o; zero rule: zeros elsewhere; replicate rule: the
// window [0, n + k - 1) holds clamped samples, zeros beyond
__global__ __launch_bounds__(kThreads) void k_stage(const float* __restrict__ src, float* __restrict__ dst, int nx, int ny, int nz,
                                                     int Fx, int Fy, int Fz, int ox, int oy, int oz, int kx, int ky, int kz,
                                                     int rep_x, int rep_y, int rep_z) {
    const size_t total = (size_t)Fx * Fy * Fz;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Fx);
        const size_t r = i / Fx;
        const int y = (int)(r % Fy), z = (int)(r / Fy);
        int sx = x - ox, sy = y - oy, sz = z - oz;
        // per axis: replicate rule = clamped samples inside [0, n + k - 1), zero rule = zeros outside [0, n)
        bool ok = true;
        if (rep_x) { ok = ok && x < nx + kx - 1; sx = min(max(sx, 0), nx - 1); } else ok = ok && sx >= 0 && sx < nx;
        if (rep_y) { ok = ok && y < ny + ky - 1; sy = min(max(sy, 0), ny - 1); } else ok = ok && sy >= 0 && sy < ny;
        if (rep_z) { ok = ok && z < nz + kz - 1; sz = min(max(sz, 0), nz - 1); } else ok = ok && sz >= 0 && sz < nz;
        const float v = ok ? src[((size_t)sz * ny + sy) * nx + sx] : 0.0f;
        dst[i] = v;
    }
}

template <int EPI>
__device__ __forceinline__ float fft_epi(float c, size_t idx, const ConvEpilogue& e) {
    if (EPI == EPI_RATIO) return e.a[idx] / fmaxf(c, kEpsSingle);
    if (EPI == EPI_UPDATE) return fabsf(e.a[idx] * c);
    if (EPI == EPI_UPDATE_REG) return fabsf(e.a[idx] * c * (1.0f - e.lambda) + e.b[idx] * e.lambda);
    return c;
}

// out (n) = epilogue(c (F) cropped at o)
template <int EPI>
__global__ __launch_bounds__(kThreads) void k_fft_epilogue(const float* __restrict__ c, float* __restrict__ out, ConvEpilogue e, int nx,
                                                            int ny, int nz, int Fx, int Fy, int ox, int oy, int oz) {
    const size_t total = (size_t)nx * ny * nz;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % nx);
        const size_t r = i / nx;
        const int y = (int)(r % ny), z = (int)(r / ny);
        out[i] = fft_epi<EPI>(c[((size_t)(z + oz) * Fy + (y + oy)) * Fx + (x + ox)], i, e);
    }
}

// same-shape fast path: 16 B per lane
template <int EPI>
__global__ __launch_bounds__(kThreads) void k_fft_epilogue_flat(const float* __restrict__ c, float* __restrict__ out, ConvEpilogue e,
                                                                 size_t n) {
    const size_t n4 = n / 4;
    const float4* c4 = reinterpret_cast<const float4*>(c);
    const float4* a4 = reinterpret_cast<const float4*>(e.a);
    const float4* b4 = reinterpret_cast<const float4*>(e.b);
    float4* o4 = reinterpret_cast<float4*>(out);
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < n4; i += stride) {
        const float4 cv = c4[i];
        float4 r;
        if (EPI == EPI_NONE) {
            r = cv;
        } else {
            const float4 av = a4[i];
            if (EPI == EPI_RATIO) {
                r = make_float4(av.x / fmaxf(cv.x, kEpsSingle), av.y / fmaxf(cv.y, kEpsSingle), av.z / fmaxf(cv.z, kEpsSingle),
                                av.w / fmaxf(cv.w, kEpsSingle));
            } else if (EPI == EPI_UPDATE) {
                r = make_float4(fabsf(av.x * cv.x), fabsf(av.y * cv.y), fabsf(av.z * cv.z), fabsf(av.w * cv.w));
            } else {
                const float4 bv = b4[i];
                const float l = e.lambda, m = 1.0f - e.lambda;
                r = make_float4(fabsf(av.x * cv.x * m + bv.x * l), fabsf(av.y * cv.y * m + bv.y * l), fabsf(av.z * cv.z * m + bv.z * l),
                                fabsf(av.w * cv.w * m + bv.w * l));
            }
        }
        o4[i] = r;
    }
    for (size_t i = n4 * 4 + tid; i < n; i += stride) out[i] = fft_epi<EPI>(c[i], i, e);
}

}  // namespace

int build_otf(hipStream_t s, rocfft_plan fwd, rocfft_execution_info info, const float* psf, const AxisPlan ax[3], float* real_scratch,
              float* otf, float scale) {
    const size_t n_spec = (size_t)(ax[0].F / 2 + 1) * ax[1].F * ax[2].F;
    MI_TRY(place_psf(s, psf, real_scratch, ax[0].k, ax[1].k, ax[2].k,
                       ax[0].F, ax[1].F, ax[2].F, ax[0].shift, ax[1].shift, ax[2].shift));
    void* in[1] = {real_scratch};
    void* out[1] = {otf};
    MI_FFT(rocfft_execute(fwd, in, out, info));
    if (scale != 1.0f) {
        hipLaunchKernelGGL(k_scale_c, dim3(stream_grid(n_spec)), dim3(kThreads), 0, s, reinterpret_cast<float2*>(otf), n_spec, scale);
        MI_TRY(launch_check("k_scale_c"));
    }
    return MI_OK;
}

// Transform lengths: circular axes keep their extent; padded axes get a 7-smooth length for rocFFT, or 2^a * {1,3,9} for the
// hand-written pipeline, which is three times as fast per point (measured: 8-10 ps against 28-30 ps per grid point and
// convolution) -- taken unless it inflates the padded volume by more than MI_FFT_NATIVE_INFLATE (default 2.2) over the 7-smooth
// one.  MI_FFT_ROCFFT=1 forces rocFFT.
bool choose_fft_lengths(const int need[3], const int bnd[3], int F[3]) {
    const char* force = std::getenv("MI_FFT_ROCFFT");
    int Fn[3];
    double vs = 1.0, vn = 1.0;
    for (int d = 0; d < 3; ++d) {
        const bool circ = bnd[d] == MI_BOUNDARY_CIRCULAR;
        F[d] = circ ? need[d] : mi_next_fast_len(need[d]);
        Fn[d] = circ ? need[d] : NativeFft::good_size(need[d], d);
        vs *= F[d];
        vn *= Fn[d];
    }
    double limit = 2.2;
    if (const char* e = std::getenv("MI_FFT_NATIVE_INFLATE")) limit = atof(e);
    const bool use_native = !(force && force[0] == '1') && Fn[0] > 0 && Fn[1] > 0 && Fn[2] > 0 && NativeFft::supported(Fn) && vn <= limit * vs;
    if (use_native)
        for (int d = 0; d < 3; ++d) F[d] = Fn[d];
    return use_native;
}

// (rocFFT plans are destroyed with their engine, not kept for the next engine of the same lengths -- creating the pair takes 0.7 s at
// decwrap's block sizes -- because live rocFFT plans are not independent of each other in ROCm 7.2: with the plans of a
// 256 x 16 x 64 grid alive, a new engine on 32 x 128 x 8 returns values 6 % off (profiles/r05_rocfft_coexistence.txt; alone it is
// exact).  A table of idle plans turned that into a failure of consecutive calls; callers keep away from the rocFFT route instead
// (lsdeconv.block_fft_shape), and an engine on it checks its transforms at creation: verify_rocfft_engine.)
FftEngine::~FftEngine() {
    delete native;
    if (info) rocfft_execution_info_destroy(info);
    if (fwd) rocfft_plan_destroy(fwd);
    if (inv) rocfft_plan_destroy(inv);
}

static int make_plans(hipStream_t s, const size_t lengths[3], rocfft_plan* fwd, rocfft_plan* inv, rocfft_execution_info* info,
                      DevBuf& work) {
    MI_TRY(rocfft_global_setup());
    MI_FFT(rocfft_plan_create(fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward, rocfft_precision_single, 3,
                              lengths, 1, nullptr));
    size_t wf = 0, wi = 0;
    MI_FFT(rocfft_plan_get_work_buffer_size(*fwd, &wf));
    if (inv) {
        MI_FFT(rocfft_plan_create(inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, rocfft_precision_single, 3,
                                  lengths, 1, nullptr));
        MI_FFT(rocfft_plan_get_work_buffer_size(*inv, &wi));
    }
    MI_FFT(rocfft_execution_info_create(info));
    const size_t w = wf > wi ? wf : wi;
    if (w) {
        MI_TRY(work.alloc(w));
        MI_FFT(rocfft_execution_info_set_work_buffer(*info, work.p, w));
    }
    MI_FFT(rocfft_execution_info_set_stream(*info, s));
    return MI_OK;
}

// Does the rocFFT plan pair transform?  Live rocFFT plans are not independent of each other in ROCm 7.2 (a new plan can return
// values several per cent off while certain other plans exist: profiles/r05_rocfft_coexistence.txt), and nothing in this library
// can prevent that -- so every engine on the rocFFT route proves itself once, at creation, on the data it has anyway:
//   forward: a few bins of the OTF it has just built against the direct sum over the PSF's samples (k^3 terms per bin, on the host);
//   inverse: the inverse transform of that OTF must give back the placed PSF (read at a few of its samples and at empty positions).
// A wrong transform fails the creation with MI_ERR_FFT instead of deconvolving with it.  Cost: one inverse transform, a few
// 8-byte reads and a synchronisation per engine; rocFFT plan creation itself takes tens to hundreds of milliseconds.
static int verify_rocfft_engine(hipStream_t s, FftEngine& e, const float* psf, const float* otf_dev, float scale) {
    const AxisPlan* ax = e.ax;
    const int kx = ax[0].k, ky = ax[1].k, kz = ax[2].k, Fx = ax[0].F, Fy = ax[1].F, Fz = ax[2].F, Hx = Fx / 2 + 1;
    const size_t nk = (size_t)kx * ky * kz;
    std::vector<float> hp(nk);
    MI_HIP(hipMemcpyAsync(hp.data(), psf, sizeof(float) * nk, hipMemcpyDeviceToHost, s));
    constexpr int NB = 6;
    int bins[NB][3];
    float2 got[NB];
    unsigned lcg = 12345u + (unsigned)Fx * 31u + (unsigned)Fy * 17u + (unsigned)Fz;
    auto next = [&](int m) { lcg = lcg * 1664525u + 1013904223u; return (int)((lcg >> 8) % (unsigned)m); };
    for (int b = 0; b < NB; ++b) {
        bins[b][0] = next(Hx);
        bins[b][1] = next(Fy);
        bins[b][2] = next(Fz);
        const size_t idx = ((size_t)bins[b][2] * Fy + bins[b][1]) * Hx + bins[b][0];
        MI_HIP(hipMemcpyAsync(&got[b], reinterpret_cast<const float2*>(otf_dev) + idx, sizeof(float2), hipMemcpyDeviceToHost, s));
    }
    // inverse: spec <- OTF (the real inverse may overwrite its input), real <- inv(spec)
    MI_HIP(hipMemcpyAsync(e.spec.p, otf_dev, sizeof(float2) * e.n_spec, hipMemcpyDeviceToDevice, s));
    void* iin[1] = {e.spec.p};
    void* iout[1] = {e.real.p};
    MI_FFT(rocfft_execute(e.inv, iin, iout, e.info));
    constexpr int NS = 8;
    size_t pos[NS];
    int tap[NS];   // index into the PSF, or -1: a position no sample lands on
    float back[NS];
    auto placed = [&](int jx, int jy, int jz) {
        const int x = ((jx - ax[0].shift) % Fx + Fx) % Fx, y = ((jy - ax[1].shift) % Fy + Fy) % Fy, z = ((jz - ax[2].shift) % Fz + Fz) % Fz;
        return ((size_t)z * Fy + y) * Fx + x;
    };
    for (int q = 0; q < NS; ++q) {
        if (q < 6) {
            const int jx = q == 0 ? kx / 2 : next(kx), jy = q == 0 ? ky / 2 : next(ky), jz = q == 0 ? kz / 2 : next(kz);
            tap[q] = (jz * ky + jy) * kx + jx;
            pos[q] = placed(jx, jy, jz);
        } else {   // the position half a grid away from the centre sample: empty whenever F > 2 k - 1, else checked as what it holds
            const int x = (((kx / 2 - ax[0].shift) + Fx / 2) % Fx + Fx) % Fx, y = (((ky / 2 - ax[1].shift) + (q == 6 ? Fy / 2 : 0)) % Fy + Fy) % Fy,
                      z = (((kz / 2 - ax[2].shift) + (q == 7 ? Fz / 2 : 0)) % Fz + Fz) % Fz;
            pos[q] = ((size_t)z * Fy + y) * Fx + x;
            tap[q] = -1;
            for (size_t i = 0; i < nk && tap[q] < 0; ++i) {
                const int jx = (int)(i % kx), r = (int)(i / kx), jy = r % ky, jz = r / ky;
                if (placed(jx, jy, jz) == pos[q]) tap[q] = (int)i;
            }
        }
        MI_HIP(hipMemcpyAsync(&back[q], e.real.as<float>() + pos[q], sizeof(float), hipMemcpyDeviceToHost, s));
    }
    MI_HIP(hipStreamSynchronize(s));
    double sum_abs = 0.0, max_abs = 0.0;
    for (float v : hp) { sum_abs += std::fabs((double)v); max_abs = std::max(max_abs, std::fabs((double)v)); }
    const double two_pi = 6.283185307179586476925286766559;
    for (int b = 0; b < NB; ++b) {
        double re = 0.0, im = 0.0;
        for (size_t i = 0; i < nk; ++i) {
            const int jx = (int)(i % kx), r = (int)(i / kx), jy = r % ky, jz = r / ky;
            const double ph = -two_pi * ((double)bins[b][0] * (jx - ax[0].shift) / Fx + (double)bins[b][1] * (jy - ax[1].shift) / Fy +
                                         (double)bins[b][2] * (jz - ax[2].shift) / Fz);
            re += hp[i] * std::cos(ph);
            im += hp[i] * std::sin(ph);
        }
        const double err = std::hypot(got[b].x - re * scale, got[b].y - im * scale);
        if (!(err <= 2e-4 * sum_abs * scale + 1e-30))
            return fail(MI_ERR_FFT, "rocFFT forward transform of the %d x %d x %d grid is wrong: OTF bin (%d, %d, %d) is (%g, %g), the PSF's direct sum gives (%g, %g) "
                        "-- live rocFFT plans of other lengths can do this (profiles/r05_rocfft_coexistence.txt): destroy the other FFT contexts of this "
                        "process, or choose a shape the hand-written pipeline takes (mi_fft_good_size)",
                        Fx, Fy, Fz, bins[b][0], bins[b][1], bins[b][2], (double)got[b].x, (double)got[b].y, re * scale, im * scale);
    }
    const double n_all = (double)Fx * Fy * Fz * scale;   // the inverse is unnormalised: real = N * scale * placed PSF
    for (int q = 0; q < NS; ++q) {
        const double want = tap[q] >= 0 ? (double)hp[(size_t)tap[q]] * n_all : 0.0;
        if (!(std::fabs((double)back[q] - want) <= 2e-4 * sum_abs * n_all + 1e-30))
            return fail(MI_ERR_FFT, "rocFFT inverse transform of the %d x %d x %d grid is wrong: sample %d of the placed PSF comes back as %g instead of %g "
                        "-- live rocFFT plans of other lengths can do this (profiles/r05_rocfft_coexistence.txt): destroy the other FFT contexts of this "
                        "process, or choose a shape the hand-written pipeline takes (mi_fft_good_size)", Fx, Fy, Fz, q, (double)back[q], want);
    }
    (void)max_abs;
    return MI_OK;
}

int FftEngine::set_psf(hipStream_t s, const float* psf) {
    MI_REQUIRE(native && !padded && !native->real_otf, "FFT engine: set_psf needs the hand-written pipeline on an unpadded grid");
    MI_TRY(place_psf(s, psf, native->scratch(), ax[0].k, ax[1].k, ax[2].k,
                       ax[0].F, ax[1].F, ax[2].F, ax[0].shift, ax[1].shift, ax[2].shift));
    const float nscale = 2.0f / (float)((double)ax[0].F * ax[1].F * ax[2].F);
    return native->build_otf(s, native->scratch(), false, nscale);
}

int FftEngine::init(hipStream_t s, const int n[3], const int k[3], const int bnd[3], const int shift[3], const float* psf,
                    const float* psf_inv, bool need_adjoint, bool fixed_psf) {
    padded = false;
    bool conj_ok = true;
    int F[3], need[3];
    for (int d = 0; d < 3; ++d) {
        AxisPlan& a = ax[d];
        a.n = n[d];
        a.k = k[d];
        a.boundary = bnd[d];
        a.shift = shift[d];
        const int off = k[d] - 1 - shift[d];  // window start offset of the forward convolution
        MI_REQUIRE(shift[d] >= 0 && off >= 0, "FFT engine: PSF placement shift %d outside [0, %d) on axis %d", shift[d], k[d], d);
        a.o = 0;
        if (bnd[d] == MI_BOUNDARY_CIRCULAR) {
            need[d] = n[d];
        } else if (bnd[d] == MI_BOUNDARY_REPLICATE) {
            need[d] = n[d] + k[d] - 1;
            a.o = off;
        } else {
            need[d] = n[d] + std::max(off, shift[d]);
        }
        // conj(OTF) is the adjoint the caller wants when the axis is circular (deconFFT: decon.m:168) or the
        // placement is symmetric (odd extent, centred)
        if (bnd[d] != MI_BOUNDARY_CIRCULAR && 2 * shift[d] != k[d] - 1) conj_ok = false;
    }
    const bool use_native = choose_fft_lengths(need, bnd, F);
    for (int d = 0; d < 3; ++d) {
        AxisPlan& a = ax[d];
        a.F = F[d];
        MI_REQUIRE(F[d] >= k[d], "FFT shape %d smaller than the PSF extent %d on axis %d", F[d], k[d], d);
        if (a.F != a.n || a.o != 0) padded = true;
    }
    n_real = (size_t)F[0] * F[1] * F[2];
    n_spec = (size_t)(F[0] / 2 + 1) * F[1] * F[2];
    // deconFFT never sees psf_inv (decon.m:18): its adjoint is conj(otf).  For 'same'-convolution semantics
    // an explicit psf_inv is just another kernel with the same placement rule.
    have_adj = need_adjoint && psf_inv != nullptr;
    if (need_adjoint && !have_adj)
        MI_REQUIRE(conj_ok, "FFT engine: the implied adjoint (flipped PSF) needs odd PSF extents on non-circular axes");
    if (use_native) {
        // the hand-written pipeline transforms the placed PSF itself: no rocFFT plan, no half-spectrum buffers
        native = new (std::nothrow) NativeFft;
        if (!native) return fail(MI_ERR_NOMEM, "FFT engine: out of host memory");
        MI_SPAN_BEGIN(sp0, "FftEngine: native init");
        MI_TRY(native->init(s, F, have_adj));
        MI_SPAN_END(sp0);
        MI_SPAN_BEGIN(sp1, "FftEngine: native OTF (enqueue)");
        // 1/(Fx Fy Fz) of the unnormalised inverse transform, times 2 for the half-length complex packing of x
        const float nscale = 2.0f / (float)((double)F[0] * F[1] * F[2]);
        for (int slot = 0; slot < (have_adj ? 2 : 1); ++slot) {
            MI_TRY(place_psf(s, slot ? psf_inv : psf, native->scratch(), ax[0].k,
                               ax[1].k, ax[2].k, ax[0].F, ax[1].F, ax[2].F, ax[0].shift, ax[1].shift, ax[2].shift));
            MI_TRY(native->build_otf(s, native->scratch(), slot == 1, nscale));
        }
        MI_SPAN_END(sp1);
        // PSFs of odd extents that are mirror-symmetric about their centre sample (every LsMakePSF PSF) have a real OTF up to the
        // phase ramp of the centre's offset from the grid origin: sample j sits at j - shift, the centre at (k-1)/2 - shift
        if (fixed_psf && (ax[0].k & 1) && (ax[1].k & 1) && (ax[2].k & 1)) {
            const int delta[3] = {(ax[0].k - 1) / 2 - ax[0].shift, (ax[1].k - 1) / 2 - ax[1].shift, (ax[2].k - 1) / 2 - ax[2].shift};
            MI_SPAN_BEGIN(sp2, "FftEngine: native real-OTF test");
            MI_TRY(native->try_real_otf(s, delta));
            MI_SPAN_END(sp2);
        }
        if (padded) {  // the x passes pad and crop on the fly: no staging volume
            int nn[3], oo[3], rep[3], kk[3];
            for (int d = 0; d < 3; ++d) {
                nn[d] = ax[d].n;
                oo[d] = ax[d].o;
                rep[d] = ax[d].boundary == MI_BOUNDARY_REPLICATE;
                kk[d] = ax[d].k;
            }
            native->set_window(nn, oo, rep, kk);
        }
        return MI_OK;
    }
    const size_t lengths[3] = {(size_t)F[0], (size_t)F[1], (size_t)F[2]};  // rocFFT: fastest dimension first
    MI_SPAN_BEGIN(sp3, "FftEngine: rocFFT plans");
    MI_TRY(make_plans(s, lengths, &fwd, &inv, &info, work));
    MI_SPAN_END(sp3);
    MI_SPAN_BEGIN(sp4, "FftEngine: rocFFT buffers + OTF (enqueue)");
    MI_TRY(real.alloc(sizeof(float) * n_real));
    MI_TRY(spec.alloc(sizeof(float) * 2 * n_spec));
    MI_TRY(otf.alloc(sizeof(float) * 2 * n_spec));
    const float scale = 1.0f / (float)((double)F[0] * F[1] * F[2]);
    MI_TRY(build_otf(s, fwd, info, psf, ax, real.as<float>(), otf.as<float>(), scale));
    MI_SPAN_END(sp4);
    if (!std::getenv("MI_FFT_NO_VERIFY")) MI_TRY(verify_rocfft_engine(s, *this, psf, otf.as<float>(), scale));
    if (have_adj) {
        MI_TRY(otf_adj.alloc(sizeof(float) * 2 * n_spec));
        MI_TRY(build_otf(s, fwd, info, psf_inv, ax, real.as<float>(), otf_adj.as<float>(), scale));
    }
    return MI_OK;
}

int FftEngine::conv(hipStream_t s, const float* in, bool adjoint, float* out, int epi_kind, const ConvEpilogue& epi) {
    if (native) return native->conv(s, in, adjoint, out, epi_kind, epi);
    MI_FFT(rocfft_execution_info_set_stream(info, s));
    const float* src = in;
    if (padded) {
        hipLaunchKernelGGL(k_stage, dim3(stream_grid(n_real)), dim3(kThreads), 0, s, in, real.as<float>(), ax[0].n, ax[1].n, ax[2].n,
                           ax[0].F, ax[1].F, ax[2].F, ax[0].o, ax[1].o, ax[2].o, ax[0].k, ax[1].k, ax[2].k,
                           ax[0].boundary == MI_BOUNDARY_REPLICATE, ax[1].boundary == MI_BOUNDARY_REPLICATE,
                           ax[2].boundary == MI_BOUNDARY_REPLICATE);
        MI_TRY(launch_check("k_stage"));
        src = real.as<float>();
    }
    void* fin[1] = {const_cast<float*>(src)};
    void* fout[1] = {spec.p};
    MI_FFT(rocfft_execute(fwd, fin, fout, info));
    const float2* o = reinterpret_cast<const float2*>(adjoint && have_adj ? otf_adj.p : otf.p);
    if (adjoint && !have_adj)
        hipLaunchKernelGGL(k_mul_otf<true>, dim3(stream_grid(n_spec)), dim3(kThreads), 0, s, spec.as<float2>(), o, n_spec);
    else
        hipLaunchKernelGGL(k_mul_otf<false>, dim3(stream_grid(n_spec)), dim3(kThreads), 0, s, spec.as<float2>(), o, n_spec);
    MI_TRY(launch_check("k_mul_otf"));
    void* iin[1] = {spec.p};
    void* iout[1] = {real.p};
    MI_FFT(rocfft_execute(inv, iin, iout, info));
    const size_t n_out = (size_t)ax[0].n * ax[1].n * ax[2].n;
    const bool flat = !padded && ((uintptr_t)out % 16) == 0 && (!epi.a || ((uintptr_t)epi.a % 16) == 0) &&
                      (!epi.b || ((uintptr_t)epi.b % 16) == 0);
#define MI_EPI(E)                                                                                                                  \
    do {                                                                                                                           \
        if (flat)                                                                                                                  \
            hipLaunchKernelGGL(k_fft_epilogue_flat<E>, dim3(stream_grid(n_out / 4 + 1)), dim3(kThreads), 0, s, real.as<float>(), out, \
                               epi, n_out);                                                                                        \
        else                                                                                                                       \
            hipLaunchKernelGGL(k_fft_epilogue<E>, dim3(stream_grid(n_out)), dim3(kThreads), 0, s, real.as<float>(), out, epi,      \
                               ax[0].n, ax[1].n, ax[2].n, ax[0].F, ax[1].F, ax[0].o, ax[1].o, ax[2].o);                           \
    } while (0)
    switch (epi_kind) {
        case EPI_NONE: case EPI_TAPER_SHELL: MI_EPI(EPI_NONE); break;
        case EPI_RATIO: MI_EPI(EPI_RATIO); break;
        case EPI_UPDATE: MI_EPI(EPI_UPDATE); break;
        case EPI_UPDATE_REG: MI_EPI(EPI_UPDATE_REG); break;
        default: return fail(MI_ERR_INVALID, "fft conv: unknown epilogue %d", epi_kind);
    }
#undef MI_EPI
    return launch_check("k_fft_epilogue");
}

}  // namespace mi

using namespace mi;

extern "C" int mi_otf(int dev, void* stream, const float* psf, int kx, int ky, int kz, float* otf, int fx, int fy, int fz, float scale) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(psf && otf, "otf_gpu: null pointer");
    MI_REQUIRE(kx > 0 && ky > 0 && kz > 0 && fx >= kx && fy >= ky && fz >= kz, "otf_gpu: fft_shape must be >= psf size in every dimension");
    hipStream_t s = as_stream(stream);
    AxisPlan ax[3];
    const int k[3] = {kx, ky, kz}, F[3] = {fx, fy, fz};
    for (int d = 0; d < 3; ++d) {
        ax[d].n = ax[d].F = F[d];
        ax[d].k = k[d];
        ax[d].shift = F[d] / 2 - (F[d] - k[d]) / 2;  // otf_gpu.cu:121-123 pre-pad + ifftshift (:36-67)
    }
    rocfft_plan fwd = nullptr;
    rocfft_execution_info info = nullptr;
    DevBuf work, real;
    const size_t lengths[3] = {(size_t)fx, (size_t)fy, (size_t)fz};
    int rc = make_plans(s, lengths, &fwd, nullptr, &info, work);
    if (rc == MI_OK) rc = real.alloc(sizeof(float) * (size_t)fx * fy * fz);
    if (rc == MI_OK) rc = build_otf(s, fwd, info, psf, ax, real.as<float>(), otf, scale);
    hipError_t e = hipStreamSynchronize(s);
    if (info) rocfft_execution_info_destroy(info);
    if (fwd) rocfft_plan_destroy(fwd);
    if (rc == MI_OK && e != hipSuccess) rc = fail(MI_ERR_HIP, "mi_otf: %s", hipGetErrorString(e));
    return rc;
}
