// Richardson-Lucy context and loops (replaces decon.m:1-204: decon / deconSpatial / deconFFT).
//
// One iteration = two fused passes regardless of engine:
//   forward_ratio : ratio = bl ./ max(conv(bl, psf), eps)       (decon.m:61-63 / 162-167)
//   adjoint_update: bl    = abs(bl .* conv(ratio, psf_inv))     (decon.m:64,76,79 / 169-186)
// The numerator is the running estimate (no stored observation) and abs() follows the product,
// exactly as the reference does.  The regularisation schedule, the Gaussian pre-smooth (5 taps on
// the GPU path), the Tikhonov blend and the ||bl||_2 stop test follow decon.m:41-59,67-79,108-118.
#include <chrono>
#include <cmath>
#include <cstring>
#include <new>
#include <utility>
#include <vector>

#include "fftconv.h"

using namespace mi;

struct mi_rl_ctx {
    int dev = 0;
    int n[3] = {0, 0, 0};
    int k[3] = {0, 0, 0};
    int bnd[3] = {MI_BOUNDARY_ZERO, MI_BOUNDARY_ZERO, MI_BOUNDARY_ZERO};
    int engine = MI_ENGINE_DIRECT;
    // direct engine: tap tables + window offsets for the forward and adjoint kernels
    DevBuf kf_fwd, kf_adj;
    int kxp = 0;
    int off_fwd[3] = {0, 0, 0}, off_adj[3] = {0, 0, 0};
    // separable fast path of the direct engine: a PSF that is an outer product a (x) b (x) c (a Gaussian: BASELINE config 1) is
    // applied as three 1-D convolutions -- kx + ky + kz instead of kx * ky * kz taps per voxel; the RL epilogue rides on the last
    bool separable = false;
    DevBuf sep_fwd[3], sep_adj[3];  // 1-D tap tables per axis
    int sep_kxp[3] = {0, 0, 0};
    DevBuf sep_t0, sep_t1;          // intermediate volumes of the three-launch route (allocated with its first convolution)
    SepTaps sep_tf[3], sep_ta[3];   // the same taps in window order for the single-pass kernel (sep3d.hip)
    bool sep_single = false;        // ... which takes this PSF (tap counts, LDS ring of kz planes, whole float4 rows)
    FftEngine* fft = nullptr;
    ~mi_rl_ctx() { delete fft; }
};

namespace {
// Rank-1 test on the host: with the largest sample p0 at (z0, y0, x0), psf is separable iff psf[z][y][x] = a[z] b[y] c[x] / p0^2
// for the lines a, b, c through that sample, to fp32 rounding.  factors[0..2] = taps along x, y, z (product = the PSF).
bool factor_rank1(const std::vector<float>& p, int kx, int ky, int kz, std::vector<float> (&factors)[3]) {
    size_t best = 0;
    for (size_t i = 1; i < p.size(); ++i)
        if (std::fabs(p[i]) > std::fabs(p[best])) best = i;
    const double p0 = p[best];
    if (!(std::fabs(p0) > 0.0) || kx * ky * kz == 1) return false;
    const int z0 = (int)(best / ((size_t)ky * kx)), y0 = (int)((best / kx) % ky), x0 = (int)(best % kx);
    auto at = [&](int z, int y, int x) { return (double)p[((size_t)z * ky + y) * kx + x]; };
    // every sample within a few fp32 roundings of the product of its three line samples RELATIVE TO THE SAMPLE (four roundings:
    // its own and those of the three line samples), with an absolute floor of 1e-9 of the peak for samples that underflow; the
    // flux the rank-1 form takes from or adds to the PSF is then below 4e-7 of its own (an absolute bound of 4e-7 of the peak
    // per sample allowed up to K * 4e-7 * peak)
    for (int z = 0; z < kz; ++z)
        for (int y = 0; y < ky; ++y)
            for (int x = 0; x < kx; ++x) {
                const double v = at(z, y, x), r = std::fabs(v - at(z, y0, x0) * at(z0, y, x0) * at(z0, y0, x) / (p0 * p0));
                if (r > 4e-7 * std::fabs(v) + 1e-9 * std::fabs(p0)) return false;
            }
    factors[0].resize(kx);
    factors[1].resize(ky);
    factors[2].resize(kz);
    for (int x = 0; x < kx; ++x) factors[0][x] = (float)at(z0, y0, x);
    for (int y = 0; y < ky; ++y) factors[1][y] = (float)(at(z0, y, x0) / p0);
    for (int z = 0; z < kz; ++z) factors[2][z] = (float)(at(z, y0, x0) / p0);
    return true;
}

// the three 1-D tap tables of a rank-1 kernel for the direct engine
int prepare_separable(hipStream_t s, const std::vector<float> (&f)[3], bool flip, DevBuf (&tabs)[3], int (&kxp)[3]) {
    for (int d = 0; d < 3; ++d) {
        DevBuf line;
        const int n = (int)f[d].size();
        MI_TRY(line.alloc(sizeof(float) * (size_t)n));
        MI_HIP(hipMemcpyAsync(line.p, f[d].data(), sizeof(float) * (size_t)n, hipMemcpyHostToDevice, s));
        MI_TRY(direct_prepare_psf(s, line.as<float>(), d == 0 ? n : 1, d == 1 ? n : 1, d == 2 ? n : 1, false, flip, tabs[d], &kxp[d]));
        MI_HIP(hipStreamSynchronize(s));  // `line` and the host vector die at scope exit
    }
    return MI_OK;
}
}  // namespace

extern "C" int mi_engine_select(int nx, int ny, int nz, int kx, int ky, int kz, int boundary) {
    // direct: 2*K flop per voxel at ~60 TFLOP/s sustained fp32.  FFT, per grid point and convolution (measured):
    // ~9.5 ps through the hand-written pipeline on a padded grid (8.2 ps on a circular one), ~29 ps through rocFFT.
    const double K = (double)kx * ky * kz;
    const double N = (double)nx * ny * nz;
    const int n[3] = {nx, ny, nz}, k[3] = {kx, ky, kz}, bnd[3] = {boundary, boundary, boundary};
    int need[3], F[3];
    for (int d = 0; d < 3; ++d) {
        need[d] = n[d];
        if (boundary == MI_BOUNDARY_ZERO) need[d] = n[d] + k[d] / 2;
        if (boundary == MI_BOUNDARY_REPLICATE) need[d] = n[d] + k[d] - 1;
    }
    const bool native = choose_fft_lengths(need, bnd, F);
    const double Nf = (double)F[0] * F[1] * F[2];
    const double t_direct = N * 2.0 * K / 60e12;
    const double t_fft = Nf * (native ? (boundary == MI_BOUNDARY_CIRCULAR ? 8.2e-12 : 9.5e-12) : 29e-12) + 60e-6;
    // without an explicit adjoint kernel the flipped PSF is conj(OTF), which needs centred odd extents off the circular rule
    const bool odd = (kx & 1) && (ky & 1) && (kz & 1);
    return ((odd || boundary == MI_BOUNDARY_CIRCULAR) && t_fft < t_direct) ? MI_ENGINE_FFT : MI_ENGINE_DIRECT;
}

// default PSF placement shift of one axis: sample j of the PSF sits at index (j - shift) of the circular kernel
static int default_shift(int n, int k, int boundary) {
    if (boundary == MI_BOUNDARY_CIRCULAR) return n / 2 - (n - k) / 2;  // ifftshift(zero-pad-centre(psf)), decon.m:131-133
    return k - 1 - conv_kernel_offset(k, boundary);
}

// psf(end:-1:1,end:-1:1,end:-1:1) of a contiguous array = its samples in reverse linear order (LsDeconv.m:163)
__global__ void k_reverse(const float* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[n - 1 - i];
}

static int rl_create(int dev, void* stream, int nx, int ny, int nz, const float* psf, const float* psf_inv, int kx, int ky, int kz,
                     const int* boundary_xyz, const int* shift_xyz, int engine, bool fixed_psf, mi_rl_ctx** out) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(out, "mi_rl_create: null ctx pointer");
    *out = nullptr;
    MI_REQUIRE(psf && boundary_xyz, "mi_rl_create: null psf / boundary");
    MI_REQUIRE(nx > 0 && ny > 0 && nz > 0 && kx > 0 && ky > 0 && kz > 0, "mi_rl_create: bl and psf must be 3D and non-empty");
    const int n[3] = {nx, ny, nz}, k[3] = {kx, ky, kz};
    int shift[3];
    bool uniform = true;
    for (int d = 0; d < 3; ++d) {
        MI_REQUIRE(boundary_xyz[d] >= MI_BOUNDARY_ZERO && boundary_xyz[d] <= MI_BOUNDARY_CIRCULAR, "mi_rl_create: unknown boundary rule %d",
                   boundary_xyz[d]);
        if (boundary_xyz[d] == MI_BOUNDARY_CIRCULAR)
            MI_REQUIRE(k[d] <= n[d], "pad_block_to_fft_shape: psf larger than FFT shape on axis %d", d);
        shift[d] = (shift_xyz && shift_xyz[d] >= 0) ? shift_xyz[d] : default_shift(n[d], k[d], boundary_xyz[d]);
        MI_REQUIRE(shift[d] >= 0 && shift[d] < k[d], "mi_rl_create: PSF shift %d outside [0,%d) on axis %d", shift[d], k[d], d);
        uniform = uniform && boundary_xyz[d] == boundary_xyz[0];
    }
    if (engine == MI_ENGINE_AUTO) engine = mi_engine_select(nx, ny, nz, kx, ky, kz, uniform ? boundary_xyz[0] : MI_BOUNDARY_ZERO);
    MI_REQUIRE(engine == MI_ENGINE_DIRECT || engine == MI_ENGINE_FFT, "mi_rl_create: engine %d not available", engine);
    hipStream_t s = as_stream(stream);
    // psf_inv == NULL stands for the flipped PSF of LsDeconv.m:163, which decon.m:64 convolves with convn(., 'same'): the same
    // centre as the forward convolution.  On a circular axis (deconFFT: conj(otf), decon.m:168) and on every axis whose placement
    // is mirror-symmetric (k - 1 - shift == shift: odd extents, centred) that is the transpose of the forward operator, which both
    // engines get for free; on a non-circular axis of EVEN extent it is one sample off the transpose, so the flipped kernel is
    // materialised and takes the explicit-psf_inv path.
    DevBuf flipped;
    if (!psf_inv) {
        bool off_transpose = false, circular_skew = false;
        for (int d = 0; d < 3; ++d) {
            const bool skew = (k[d] - 1 - shift[d]) != shift[d];
            off_transpose = off_transpose || (skew && boundary_xyz[d] != MI_BOUNDARY_CIRCULAR);
            circular_skew = circular_skew || (skew && boundary_xyz[d] == MI_BOUNDARY_CIRCULAR);
        }
        if (off_transpose) {
            MI_REQUIRE(!circular_skew, "mi_rl_create: psf_inv == NULL is ambiguous for even PSF extents on a mix of circular and "
                                       "non-circular axes; pass the adjoint kernel explicitly");
            const int nk = kx * ky * kz;
            MI_TRY(flipped.alloc(sizeof(float) * (size_t)nk));
            k_reverse<<<cdiv(nk, 256), 256, 0, s>>>(psf, flipped.as<float>(), nk);
            MI_TRY(launch_check("k_reverse"));
            psf_inv = flipped.as<float>();
        }
    }
    mi_rl_ctx* c = new (std::nothrow) mi_rl_ctx;
    if (!c) return fail(MI_ERR_NOMEM, "mi_rl_create: out of host memory");
    c->dev = dev;
    c->engine = engine;
    for (int d = 0; d < 3; ++d) {
        c->n[d] = n[d];
        c->k[d] = k[d];
        c->bnd[d] = boundary_xyz[d];
        // forward: out[x] = sum_j in[x - j + shift] psf[j]  -> window starts k-1-shift before x (flipped taps)
        // adjoint (conj OTF): out[x] = sum_j in[x + j - shift] psf[j] -> window starts shift before x (plain taps);
        // with an explicit psf_inv it is an ordinary convolution with that kernel
        c->off_fwd[d] = k[d] - 1 - shift[d];
        c->off_adj[d] = psf_inv ? k[d] - 1 - shift[d] : shift[d];
    }
    int rc = MI_OK;
    if (engine == MI_ENGINE_DIRECT && !std::getenv("MI_NO_SEPARABLE")) {
        // rank-1 PSF (and adjoint kernel): three 1-D passes instead of the dense taps
        const size_t nk = (size_t)kx * ky * kz;
        std::vector<float> hp(nk), hi(psf_inv ? nk : 0);
        std::vector<float> ff[3], fi[3];
        hipError_t e = hipMemcpyAsync(hp.data(), psf, sizeof(float) * nk, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && psf_inv) e = hipMemcpyAsync(hi.data(), psf_inv, sizeof(float) * nk, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) rc = fail(MI_ERR_HIP, "mi_rl_create: %s", hipGetErrorString(e));
        if (rc == MI_OK && factor_rank1(hp, kx, ky, kz, ff) && (!psf_inv || factor_rank1(hi, kx, ky, kz, fi))) {
            rc = prepare_separable(s, ff, /*flip=*/true, c->sep_fwd, c->sep_kxp);
            // adjoint taps: flip(psf_inv); without psf_inv the transpose of the forward operator = the plain taps of psf
            if (rc == MI_OK) rc = psf_inv ? prepare_separable(s, fi, true, c->sep_adj, c->sep_kxp) : prepare_separable(s, ff, false, c->sep_adj, c->sep_kxp);
            c->separable = rc == MI_OK;
            if (c->separable) {  // window-order taps: forward = the flipped lines; adjoint = flip(psf_inv), or the plain lines of psf
                bool fits = true;
                for (int d = 0; d < 3; ++d) {
                    const std::vector<float>& fl = ff[d];
                    const std::vector<float>& al = psf_inv ? fi[d] : ff[d];
                    const int nt = (int)fl.size();
                    fits = fits && nt <= kSepMaxTaps;
                    if (!fits) break;
                    c->sep_tf[d].n = c->sep_ta[d].n = nt;
                    for (int t = 0; t < nt; ++t) {
                        c->sep_tf[d].w[t] = fl[nt - 1 - t];
                        c->sep_ta[d].w[t] = psf_inv ? al[nt - 1 - t] : al[t];
                    }
                }
                c->sep_single = fits && !std::getenv("MI_NO_SEP_SINGLE") && sep3d_fits(nx, c->k, c->off_fwd) && sep3d_fits(nx, c->k, c->off_adj);
            }
        }
    }
    if (rc == MI_OK && engine == MI_ENGINE_DIRECT) {
        rc = direct_prepare_psf(s, psf, kx, ky, kz, false, /*flip=*/true, c->kf_fwd, &c->kxp);
        // adjoint taps: flip(psf_inv); with psf_inv = flip(psf) (LsDeconv.m:163) that is psf itself
        if (rc == MI_OK)
            rc = psf_inv ? direct_prepare_psf(s, psf_inv, kx, ky, kz, false, true, c->kf_adj, &c->kxp)
                         : direct_prepare_psf(s, psf, kx, ky, kz, false, false, c->kf_adj, &c->kxp);
    } else {
        c->fft = new (std::nothrow) FftEngine;
        if (!c->fft) rc = fail(MI_ERR_NOMEM, "mi_rl_create: out of host memory");
        if (rc == MI_OK) rc = c->fft->init(s, n, k, c->bnd, shift, psf, psf_inv, true, fixed_psf);
    }
    if (rc == MI_OK) {
        MI_SPAN_BEGIN(spw, "rl_create: final wait for the stream");
        hipError_t e = hipStreamSynchronize(s);
        MI_SPAN_END(spw);
        if (e != hipSuccess) rc = fail(MI_ERR_HIP, "mi_rl_create: %s", hipGetErrorString(e));
    }
    if (rc != MI_OK) {
        delete c;
        return rc;
    }
    *out = c;
    return MI_OK;
}

extern "C" int mi_rl_create_ex(int dev, void* stream, int nx, int ny, int nz, const float* psf, const float* psf_inv, int kx, int ky,
                               int kz, const int* boundary_xyz, const int* shift_xyz, int engine, mi_rl_ctx** out) {
    return rl_create(dev, stream, nx, ny, nz, psf, psf_inv, kx, ky, kz, boundary_xyz, shift_xyz, engine, true, out);
}

extern "C" int mi_rl_create(int dev, void* stream, int nx, int ny, int nz, const float* psf, const float* psf_inv, int kx, int ky,
                            int kz, int boundary, int engine, mi_rl_ctx** out) {
    const int b[3] = {boundary, boundary, boundary};
    // deconFFT takes only psf.psf (decon.m:18): the circular flavour ignores psf_inv
    return mi_rl_create_ex(dev, stream, nx, ny, nz, psf, boundary == MI_BOUNDARY_CIRCULAR ? nullptr : psf_inv, kx, ky, kz, b, nullptr,
                           engine, out);
}

extern "C" int mi_rl_destroy(mi_rl_ctx* ctx) {
    if (!ctx) return MI_OK;
    (void)hipSetDevice(ctx->dev);
    delete ctx;
    return MI_OK;
}

extern "C" int mi_rl_engine(const mi_rl_ctx* ctx) { return ctx ? ctx->engine : MI_ERR_INVALID; }

extern "C" size_t mi_rl_device_bytes(const mi_rl_ctx* ctx) {
    if (!ctx) return 0;
    return ctx->kf_fwd.bytes + ctx->kf_adj.bytes + ctx->sep_t0.bytes + ctx->sep_t1.bytes + (ctx->fft ? ctx->fft->device_bytes() : 0);
}

extern "C" int mi_rl_separable(const mi_rl_ctx* ctx) { return ctx && ctx->separable ? (ctx->sep_single ? 2 : 1) : 0; }
extern "C" int mi_rl_pair_layout(const mi_rl_ctx* ctx) {
    return ctx && ctx->engine == MI_ENGINE_FFT && ctx->fft && ctx->fft->native && ctx->fft->native->dims.paired ? 1 : 0;
}

static int ctx_conv(mi_rl_ctx* c, hipStream_t s, const float* in, bool adjoint, float* out, int epi_kind, const ConvEpilogue& epi) {
    if (c->engine == MI_ENGINE_DIRECT && c->separable && c->sep_single && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0 &&
        ((uintptr_t)epi.a % 16) == 0 && ((uintptr_t)epi.b % 16) == 0 && epi_kind != EPI_TAPER_SHELL)
        return sep3d_launch(s, in, out, c->n[0], c->n[1], c->n[2], adjoint ? c->sep_ta : c->sep_tf, adjoint ? c->off_adj : c->off_fwd, c->bnd,
                            epi_kind, epi);
    if (c->engine == MI_ENGINE_DIRECT && c->separable) {
        const size_t bytes = sizeof(float) * (size_t)c->n[0] * c->n[1] * c->n[2];
        if (!c->sep_t0.p) MI_TRY(c->sep_t0.alloc(bytes));
        if (!c->sep_t1.p) MI_TRY(c->sep_t1.alloc(bytes));
        const int* off = adjoint ? c->off_adj : c->off_fwd;
        DevBuf* tab = adjoint ? c->sep_adj : c->sep_fwd;
        const float* src = in;
        float* dsts[3] = {c->sep_t0.as<float>(), c->sep_t1.as<float>(), out};
        for (int d = 0; d < 3; ++d) {  // x, then y, then z; window offsets of the other axes are 0 for a 1-tap kernel
            const int o[3] = {d == 0 ? off[0] : 0, d == 1 ? off[1] : 0, d == 2 ? off[2] : 0};
            MI_TRY(direct_conv_launch(s, src, tab[d].as<float>(), dsts[d], c->n[0], c->n[1], c->n[2], d == 0 ? c->k[0] : 1, d == 1 ? c->k[1] : 1,
                                      d == 2 ? c->k[2] : 1, c->sep_kxp[d], c->bnd[0], d == 2 ? epi_kind : EPI_NONE, d == 2 ? epi : ConvEpilogue(), o,
                                      c->bnd));
            src = dsts[d];
        }
        return MI_OK;
    }
    if (c->engine == MI_ENGINE_DIRECT)
        return direct_conv_launch(s, in, adjoint ? c->kf_adj.as<float>() : c->kf_fwd.as<float>(), out, c->n[0], c->n[1], c->n[2], c->k[0],
                                  c->k[1], c->k[2], c->kxp, c->bnd[0], epi_kind, epi, adjoint ? c->off_adj : c->off_fwd, c->bnd);
    return c->fft->conv(s, in, adjoint, out, epi_kind, epi);
}

extern "C" int mi_rl_forward_ratio(mi_rl_ctx* ctx, void* stream, const float* bl, float* ratio) {
    MI_REQUIRE(ctx && bl && ratio && bl != ratio, "mi_rl_forward_ratio: null or aliased pointers");
    MI_TRY(use_device(ctx->dev));
    ConvEpilogue e;
    e.a = bl;
    return ctx_conv(ctx, as_stream(stream), bl, false, ratio, EPI_RATIO, e);
}

extern "C" int mi_rl_adjoint_update(mi_rl_ctx* ctx, void* stream, const float* ratio, float* bl, float lambda, const float* reg) {
    MI_REQUIRE(ctx && bl && ratio && bl != ratio, "mi_rl_adjoint_update: null or aliased pointers");
    MI_TRY(use_device(ctx->dev));
    ConvEpilogue e;
    e.a = bl;
    e.b = reg;
    e.lambda = lambda;
    const bool with_reg = lambda > 0.0f && reg != nullptr;
    return ctx_conv(ctx, as_stream(stream), ratio, true, bl, with_reg ? EPI_UPDATE_REG : EPI_UPDATE, e);
}

extern "C" int mi_rl_iterate(mi_rl_ctx* ctx, void* stream, float* bl, float* ratio, int n_iters) {
    MI_REQUIRE(ctx && bl, "mi_rl_iterate: null pointer");
    MI_REQUIRE(n_iters >= 0, "mi_rl_iterate: negative iteration count");
    MI_TRY(use_device(ctx->dev));
    if (ctx->engine == MI_ENGINE_FFT && ctx->fft->native && ctx->fft->native->can_fuse())
        return ctx->fft->native->iterate(as_stream(stream), bl, n_iters);
    MI_REQUIRE(ratio && ratio != bl, "mi_rl_iterate: this engine needs a ratio scratch volume");
    for (int i = 0; i < n_iters; ++i) {
        MI_TRY(mi_rl_forward_ratio(ctx, stream, bl, ratio));
        MI_TRY(mi_rl_adjoint_update(ctx, stream, ratio, bl, 0.0f, nullptr));
    }
    return MI_OK;
}

extern "C" int mi_fft_good_size(int n, int axis) {
    // extents the hand-written FFT pipeline takes without falling back to rocFFT: 2^a * {1, 3, 9} per axis (x: even, the
    // transform runs on x/2 complex points); 0 when no such extent exists (z beyond the LDS tile)
    if (axis < 0 || axis > 2 || n < 1) return 0;
    return NativeFft::good_size(n, axis);
}

// ---- sharded fused iteration (slab driver): mi_rl_iterate cut where the y halos of a convolution's input are refreshed
static int sharded_native(mi_rl_ctx* ctx, NativeFft** out) {
    MI_REQUIRE(ctx, "mi_rl_sharded: null context");
    MI_TRY(use_device(ctx->dev));
    if (!(ctx->engine == MI_ENGINE_FFT && ctx->fft->native && ctx->fft->native->can_fuse()))
        return fail(MI_ERR_UNSUPPORTED, "mi_rl_sharded: only the native FFT pipeline fuses consecutive convolutions");
    *out = ctx->fft->native;
    return ctx->fft->native->release_spare();   // (only the fused loop of mi_rl_iterate settles the spare S array: here it would stay for good)
}

extern "C" int mi_rl_fuses(mi_rl_ctx* ctx) {
    if (!(ctx && ctx->engine == MI_ENGINE_FFT && ctx->fft->native && ctx->fft->native->can_fuse())) return 0;
    return ctx->fft->native->splits() ? 2 : 1;
}

extern "C" int mi_rl_otf_is_real(mi_rl_ctx* ctx) {
    return ctx && ctx->engine == MI_ENGINE_FFT && ctx->fft->native && ctx->fft->native->real_otf ? 1 : 0;
}

extern "C" int mi_rl_sharded_begin(mi_rl_ctx* ctx, void* stream, const float* bl) {
    NativeFft* nf = nullptr;
    MI_TRY(sharded_native(ctx, &nf));
    MI_REQUIRE(bl, "mi_rl_sharded_begin: null pointer");
    return nf->x_forward(as_stream(stream), bl);
}

// part 0: the whole step; 1: the y/z passes and the x tiles holding rows of [a0,a1) or [b0,b1); 2: the remaining x tiles
static int sharded_step(mi_rl_ctx* ctx, void* stream, float* bl, bool update, int more, int part, const int* edges) {
    NativeFft* nf = nullptr;
    MI_TRY(sharded_native(ctx, &nf));
    MI_REQUIRE(bl, "mi_rl_sharded: null pointer");
    MI_REQUIRE(part >= 0 && part <= 2, "mi_rl_sharded: part must be 0, 1 or 2");
    TileSelect sel{};
    if (part != 0) {
        MI_REQUIRE(nf->splits(), "mi_rl_sharded: this context cannot split the x pass (mi_rl_fuses() != 2)");
        MI_REQUIRE(edges && edges[0] >= 0 && edges[0] < edges[1] && edges[1] <= edges[2] && edges[2] < edges[3] && edges[3] <= ctx->n[1],
                   "mi_rl_sharded: edge row ranges must be ordered and inside the volume");
        sel = nf->edge_tiles(part, edges[0], edges[1], edges[2], edges[3]);
    }
    ConvEpilogue e;
    e.a = bl;
    if (part != 2) MI_TRY(nf->middle(as_stream(stream), update));
    return update ? nf->x_inverse(as_stream(stream), bl, EPI_UPDATE, e, more != 0, &sel)
                  : nf->x_inverse(as_stream(stream), nullptr, EPI_RATIO, e, true, &sel);
}

extern "C" int mi_rl_sharded_ratio(mi_rl_ctx* ctx, void* stream, const float* bl, int part, const int* edge_rows) {
    return sharded_step(ctx, stream, const_cast<float*>(bl), false, 1, part, edge_rows);
}

extern "C" int mi_rl_sharded_update(mi_rl_ctx* ctx, void* stream, float* bl, int more, int part, const int* edge_rows) {
    MI_REQUIRE(part == 0 || more, "mi_rl_sharded_update: a split step must produce the next input (more != 0)");
    return sharded_step(ctx, stream, bl, true, more, part, edge_rows);
}

extern "C" int mi_rl_set_overlap(mi_rl_ctx* ctx, int free_cus, int dynamic_tiles) {
    NativeFft* nf = nullptr;
    MI_TRY(sharded_native(ctx, &nf));
    MI_REQUIRE(free_cus >= 0 && free_cus < nf->n_cu, "mi_rl_set_overlap: free_cus must be in [0, %d)", nf->n_cu);
    nf->overlap_free_cus = free_cus;
    nf->overlap_dynamic = dynamic_tiles != 0;
    return MI_OK;
}

// A stand-in for a collective's kernels: `busy_wgs` small work-groups that hold their compute units for `ticks` of the 100-MHz
// wall clock (and touch `buf` so that they are not optimised away)
__global__ __launch_bounds__(256) void k_busy(long long ticks, int* __restrict__ buf) {
    extern __shared__ int hold[];  // (dynamic LDS: what keeps a 140-KB work-group of the x pass off this compute unit)
    hold[threadIdx.x] = (int)threadIdx.x;
    const long long t0 = wall_clock64();
    int spins = 0;
    while (wall_clock64() - t0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        ++spins;
    }
    if (threadIdx.x == 0 && buf) buf[blockIdx.x] = spins + hold[(spins + 1) & 255];
}

extern "C" int mi_rl_overlap_probe(mi_rl_ctx* ctx, void* stream, float* bl, const int* edge_rows, int busy_wgs, float busy_us, int reps,
                                   float* out_ms) {
    NativeFft* nf = nullptr;
    MI_TRY(sharded_native(ctx, &nf));
    MI_REQUIRE(bl && edge_rows && out_ms && reps > 0 && busy_wgs >= 0 && busy_wgs <= 1024 && busy_us >= 0.0f && busy_us <= 50000.0f,
               "mi_rl_overlap_probe: bad arguments");
    MI_REQUIRE(nf->splits(), "mi_rl_overlap_probe: this context cannot split the x pass");
    hipStream_t s = as_stream(stream), side = nullptr;
    const TileSelect sel = nf->edge_tiles(2, edge_rows[0], edge_rows[1], edge_rows[2], edge_rows[3]);
    ConvEpilogue e;
    e.a = bl;
    DevBuf sink;
    MI_TRY(sink.alloc(sizeof(int) * 1024));
    MI_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t ev[4];
    for (auto& x : ev) MI_HIP(hipEventCreate(&x));
    int rc = MI_OK;
    float sum_x = 0.0f, sum_all = 0.0f;
    for (int r = -1; r < reps && rc == MI_OK; ++r) {  // r == -1: warm-up
        (void)hipStreamSynchronize(s);
        (void)hipEventRecord(ev[0], s);
        if (busy_wgs > 0) {
            (void)hipStreamWaitEvent(side, ev[0], 0);
            // 48 KB of LDS per stand-in work-group: it cannot share a compute unit with a work-group of the x pass (~140 of 160 KB),
            // like a collective's kernel with its staging buffers (a stand-in without LDS simply co-resides: measured, no effect)
            hipLaunchKernelGGL(k_busy, dim3((unsigned)busy_wgs), dim3(256), 48 * 1024, side, (long long)(busy_us * 100.0f), sink.as<int>());
            rc = launch_check("k_busy");
            (void)hipEventRecord(ev[3], side);
            // the stand-in must be resident before the pass is launched, like a collective that was issued first
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 150.0) {}
        }
        (void)hipEventRecord(ev[1], s);
        if (rc == MI_OK) rc = nf->x_inverse(s, nullptr, EPI_RATIO, e, true, &sel);
        (void)hipEventRecord(ev[2], s);
        if (busy_wgs > 0) (void)hipStreamWaitEvent(s, ev[3], 0);
        (void)hipEventRecord(ev[3], s);
        hipError_t he = hipEventSynchronize(ev[3]);
        float a = 0.0f, b = 0.0f;
        if (he == hipSuccess) he = hipEventElapsedTime(&a, ev[1], ev[2]);
        if (he == hipSuccess) he = hipEventElapsedTime(&b, ev[0], ev[3]);
        if (rc == MI_OK && he != hipSuccess) rc = fail(MI_ERR_HIP, "mi_rl_overlap_probe: %s", hipGetErrorString(he));
        if (r >= 0) { sum_x += a; sum_all += b; }
    }
    (void)hipStreamSynchronize(side);
    for (auto& x : ev) (void)hipEventDestroy(x);
    (void)hipStreamDestroy(side);
    out_ms[0] = sum_x / (float)reps;    // launch -> end of the x launch (part 2)
    out_ms[1] = sum_all / (float)reps;  // stand-in issued -> both finished
    return rc;
}

extern "C" int mi_rl_spectrum_rows(mi_rl_ctx* ctx, void* stream, int y0, int rows, float* buf, int dir) {
    NativeFft* nf = nullptr;
    MI_TRY(sharded_native(ctx, &nf));
    MI_REQUIRE(dir >= 0 && dir <= 2, "mi_rl_spectrum_rows: dir must be 0 (pack), 1 (unpack) or 2 (zero)");
    return nf->spectrum_rows(as_stream(stream), y0, rows, reinterpret_cast<float2*>(buf), dir);
}

extern "C" int mi_rl_spectrum_rows_z(mi_rl_ctx* ctx, void* stream, int y0, int rows, int z0, int z1, float* buf, int dir) {
    NativeFft* nf = nullptr;
    MI_TRY(sharded_native(ctx, &nf));
    MI_REQUIRE(dir >= 0 && dir <= 2, "mi_rl_spectrum_rows_z: dir must be 0 (pack), 1 (unpack) or 2 (zero)");
    MI_REQUIRE(z0 >= 0 && z0 < z1 && z1 <= ctx->n[2], "mi_rl_spectrum_rows_z: planes [%d, %d) outside [0, %d)", z0, z1, ctx->n[2]);
    return nf->spectrum_rows(as_stream(stream), y0, rows, reinterpret_cast<float2*>(buf), dir, z0, z1 - z0);
}

extern "C" int mi_rl_z_granule(mi_rl_ctx* ctx) {
    NativeFft* nf = nullptr;
    if (sharded_native(ctx, &nf) != MI_OK || !nf->splits()) return 0;
    return nf->y_z_granule();
}

// The sharded step cut into the stages of the z-chunked halo exchange (see include/mi_lsdeconv.h)
extern "C" int mi_rl_sharded_stage(mi_rl_ctx* ctx, void* stream, float* bl, int update, int stage, int z0, int z1, const int* edges) {
    NativeFft* nf = nullptr;
    MI_TRY(sharded_native(ctx, &nf));
    MI_REQUIRE(stage >= 0 && stage <= 3, "mi_rl_sharded_stage: stage must be 0 .. 3");
    MI_REQUIRE(nf->splits(), "mi_rl_sharded_stage: this context cannot split the x pass (mi_rl_fuses() != 2)");
    hipStream_t s = as_stream(stream);
    if (stage == 0) {
        MI_REQUIRE(z0 >= 0 && z0 < z1 && z1 <= ctx->n[2], "mi_rl_sharded_stage: planes [%d, %d) outside [0, %d)", z0, z1, ctx->n[2]);
        return nf->y_forward_planes(s, z0, z1 - z0);
    }
    if (stage == 1) {
        MI_TRY(nf->z_conv(s, update != 0));
        return nf->y_pass(s, true, nf->dims.paired != 0);
    }
    MI_REQUIRE(bl, "mi_rl_sharded_stage: null pointer");
    MI_REQUIRE(edges && edges[0] >= 0 && edges[0] < edges[1] && edges[1] <= edges[2] && edges[2] < edges[3] && edges[3] <= ctx->n[1],
               "mi_rl_sharded_stage: edge row ranges must be ordered and inside the volume");
    TileSelect sel = nf->edge_tiles(stage == 2 ? 1 : 2, edges[0], edges[1], edges[2], edges[3]);
    if (stage == 2) {
        MI_REQUIRE(z0 >= 0 && z0 < z1 && z1 <= ctx->n[2], "mi_rl_sharded_stage: planes [%d, %d) outside [0, %d)", z0, z1, ctx->n[2]);
        sel.z0 = z0;
        sel.nz = z1 - z0;
    }
    ConvEpilogue e;
    e.a = bl;
    return update ? nf->x_inverse(s, bl, EPI_UPDATE, e, true, &sel) : nf->x_inverse(s, nullptr, EPI_RATIO, e, true, &sel);
}

extern "C" size_t mi_rl_spectrum_row_floats(mi_rl_ctx* ctx) {
    return mi_rl_fuses(ctx) ? ctx->fft->native->spectrum_row_floats() : 0;
}

extern "C" int mi_rl_time_pass(mi_rl_ctx* ctx, void* stream, int which, const float* bl, int reps, float* avg_ms) {
    MI_REQUIRE(ctx && bl && avg_ms, "mi_rl_time_pass: null pointer");
    MI_TRY(use_device(ctx->dev));
    if (!(ctx->engine == MI_ENGINE_FFT && ctx->fft->native && !ctx->fft->padded))
        return fail(MI_ERR_UNSUPPORTED, "mi_rl_time_pass: only the unpadded native FFT pipeline exposes its passes");
    return ctx->fft->native->time_pass(as_stream(stream), which, bl, reps, avg_ms);
}

extern "C" size_t mi_rl_fft_spectrum_bytes(mi_rl_ctx* ctx) {
    if (!(ctx && ctx->engine == MI_ENGINE_FFT && ctx->fft && ctx->fft->native)) return 0;
    return ctx->fft->native->spectrum_bytes();
}

extern "C" int mi_rl_time_between(mi_rl_ctx* ctx, void* stream, int which, const void* src, void* dst, float* bl, int reps, float* avg_ms) {
    MI_REQUIRE(ctx && src && dst && avg_ms && reps > 0 && (which == 0 || which == 3 || which == 4 || ((which == 1 || which == 2) && bl)), "mi_rl_time_between: bad arguments");
    MI_TRY(use_device(ctx->dev));
    if (!(ctx->engine == MI_ENGINE_FFT && ctx->fft->native && !ctx->fft->padded))
        return fail(MI_ERR_UNSUPPORTED, "mi_rl_time_between: only the unpadded native FFT pipeline");
    return ctx->fft->native->time_between(as_stream(stream), which, static_cast<const float2*>(src), static_cast<float2*>(dst), bl, reps, avg_ms);
}

extern "C" int mi_rl_fft_placement(mi_rl_ctx* ctx, float* cost_ms, int cap, int* n, int* kept) {
    MI_REQUIRE(ctx && n && kept && (cost_ms || cap <= 0), "mi_rl_fft_placement: null pointer");
    *n = 0;
    *kept = -1;
    if (!(ctx->engine == MI_ENGINE_FFT && ctx->fft && ctx->fft->native)) return MI_OK;
    const auto& ms = ctx->fft->native->placement_ms;
    *n = (int)ms.size();
    *kept = ctx->fft->native->placement_kept;
    for (int i = 0; i < *n && i < cap; ++i) cost_ms[i] = ms[i];
    return MI_OK;
}

// ------------------------------------------------------------------------------------------------
extern "C" int mi_conv3d(int dev, void* stream, const float* img, const float* ker, float* out, int nx, int ny, int nz, int kx, int ky,
                         int kz, int boundary, int engine) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(img && ker && out && img != out, "conv3d: null or aliased pointers");
    MI_REQUIRE(nx > 0 && ny > 0 && nz > 0, "conv3d_gpu:Input: Image must be 3D gpuArray single.");
    MI_REQUIRE(kx > 0 && ky > 0 && kz > 0, "conv3d_gpu:Kernel: Kernel must be 3D gpuArray single.");
    MI_REQUIRE(boundary >= MI_BOUNDARY_ZERO && boundary <= MI_BOUNDARY_CIRCULAR, "conv3d: unknown boundary rule %d", boundary);
    hipStream_t s = as_stream(stream);
    if (engine == MI_ENGINE_AUTO) engine = mi_engine_select(nx, ny, nz, kx, ky, kz, boundary);
    ConvEpilogue e;
    int rc;
    if (engine == MI_ENGINE_DIRECT) {
        DevBuf kf;
        int kxp = 0;
        rc = direct_prepare_psf(s, ker, kx, ky, kz, false, true, kf, &kxp);
        if (rc == MI_OK) rc = direct_conv_launch(s, img, kf.as<float>(), out, nx, ny, nz, kx, ky, kz, kxp, boundary, EPI_NONE, e);
        hipError_t he = hipStreamSynchronize(s);  // kf dies at scope exit
        if (rc == MI_OK && he != hipSuccess) rc = fail(MI_ERR_HIP, "conv3d: %s", hipGetErrorString(he));
        return rc;
    }
    MI_REQUIRE(engine == MI_ENGINE_FFT, "conv3d: engine %d not available", engine);
    FftEngine fe;
    const int n[3] = {nx, ny, nz}, k[3] = {kx, ky, kz}, b[3] = {boundary, boundary, boundary};
    int shift[3];
    for (int d = 0; d < 3; ++d) {
        MI_REQUIRE(boundary != MI_BOUNDARY_CIRCULAR || k[d] <= n[d], "conv3d: kernel larger than the circular shape on axis %d", d);
        // a plain circular convolution is centred like convn (no deconFFT placement quirk)
        shift[d] = k[d] - 1 - conv_kernel_offset(k[d], boundary);
    }
    rc = fe.init(s, n, k, b, shift, ker, nullptr, /*need_adjoint=*/false);
    if (rc == MI_OK) rc = fe.conv(s, img, false, out, EPI_NONE, e);
    hipError_t he = hipStreamSynchronize(s);
    if (rc == MI_OK && he != hipSuccess) rc = fail(MI_ERR_HIP, "conv3d: %s", hipGetErrorString(he));
    return rc;
}

extern "C" int mi_conv3d_replicate(int dev, void* stream, const float* img, const float* ker, float* out, int nx, int ny, int nz, int kx,
                                   int ky, int kz) {
    return mi_conv3d(dev, stream, img, ker, out, nx, ny, nz, kx, ky, kz, MI_BOUNDARY_REPLICATE, MI_ENGINE_DIRECT);
}

// ------------------------------------------------------------------------------------------------
namespace {

bool regularization_time(int i, int niter, int interval) {  // decon.m:54-55, i is 1-based
    const bool apply = interval > 0 && interval < niter;
    return apply && i > 1 && i < niter && (i % interval) == 0;
}

int host_norm(hipStream_t s, const float* x, size_t n, double* d_scratch, double* out) {
    MI_TRY(sumsq_async(s, x, n, d_scratch));
    double h = 0.0;
    MI_HIP(hipMemcpyAsync(&h, d_scratch, sizeof(double), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    *out = std::sqrt(h);
    return MI_OK;
}

// the loop shared by deconSpatial and deconFFT once bl has its working shape
int rl_iterate(mi_rl_ctx* ctx, hipStream_t s, float* bl, float* ratio, float* reg, double* d_scratch, int nx, int ny, int nz,
               const mi_rl_options& o, double delta_prev, int* iters_done) {
    const size_t N = (size_t)nx * ny * nz;
    const float sig[3] = {0.5f, 0.5f, 0.5f};
    const int k3[3] = {3, 3, 3};
    int done = 0;
    // the estimate ping-pongs between the caller's array and the scratch volume: the single-pass Gaussian of the regularisation
    // step writes into the other buffer (8 B/voxel, no copy back), which then IS the estimate; the buffer it leaves is the
    // scratch of the following steps.  One copy at the end when the estimate sits in the scratch volume.
    float* const home = bl;
    for (int i = 1; i <= o.niter;) {
        const bool reg_time = regularization_time(i, o.niter, o.regularize_interval);
        int span = 1;
        if (reg_time) {
            MI_REQUIRE(ratio, "decon: the regularisation step needs the scratch volume");
            bool fused = false;
            MI_TRY(gauss3d_to(s, bl, ratio, nx, ny, nz, sig, o.gauss_taps == 3 ? k3 : nullptr, &fused));  // decon.m:57-59
            if (fused) std::swap(bl, ratio);
            if (o.lambda > 0.0f) {
                MI_TRY(mi_rl_forward_ratio(ctx, s, bl, ratio));
                MI_TRY(mi_rl_reg_term(ctx->dev, s, bl, reg, nx, ny, nz));
                MI_TRY(mi_rl_adjoint_update(ctx, s, ratio, bl, o.lambda, reg));
            } else {
                // without the Tikhonov term the smoothed estimate simply goes through a plain iteration (fused on the native pipeline)
                MI_TRY(mi_rl_iterate(ctx, s, bl, ratio, 1));
            }
        } else {
            // run of plain iterations up to the next regularisation step: one call, so an engine that can fuse
            // consecutive iterations (native FFT pipeline) does; the stop test needs the norm after every iteration
            if (o.stop_criterion <= 0.0f)
                while (i + span <= o.niter && !regularization_time(i + span, o.niter, o.regularize_interval)) ++span;
            MI_TRY(mi_rl_iterate(ctx, s, bl, ratio, span));
        }
        i += span;
        done = i - 1;
        if (o.stop_criterion > 0.0f) {  // decon.m:108-118
            double cur = 0.0;
            MI_TRY(host_norm(s, bl, N, d_scratch, &cur));
            const double rel = std::fabs(delta_prev - cur) / delta_prev * 100.0;
            delta_prev = cur;
            if (done > 1 && rel <= (double)o.stop_criterion) break;
        }
    }
    if (bl != home) MI_HIP(hipMemcpyAsync(home, bl, sizeof(float) * N, hipMemcpyDeviceToDevice, s));
    if (iters_done) *iters_done = done;
    return MI_OK;
}

int check_options(const mi_rl_options* o) {
    MI_REQUIRE(o, "decon: null options");
    MI_REQUIRE(o->niter >= 0, "decon: niter must be >= 0");
    MI_REQUIRE(o->lambda >= 0.0f && o->lambda < 1.0f, "decon: lambda must be in [0,1)");
    MI_REQUIRE(o->gauss_taps == 0 || o->gauss_taps == 3 || o->gauss_taps == 5, "decon: gauss_taps must be 0, 3 or 5");
    MI_REQUIRE(o->psf_grid[0] >= 0 && o->psf_grid[1] >= 0 && o->psf_grid[2] >= 0, "decon: psf_grid extents must be >= 0");
    return MI_OK;
}

}  // namespace

// keep_ctx / keep_taper (a deconvolution plan): the RL context and the taper's FFT engine are created into / reused from them
static int rl_spatial_impl(int dev, void* stream, float* bl, const float* psf, const float* psf_inv, int nx, int ny, int nz, int kx, int ky,
                           int kz, const mi_rl_options* opt, int* iters_done, mi_rl_ctx** keep_ctx, TaperKeep** keep_taper) {
    MI_TRY(use_device(dev));
    MI_TRY(check_options(opt));
    MI_REQUIRE(bl && psf, "deconSpatial: null pointer");
    MI_REQUIRE(nx > 0 && ny > 0 && nz > 0 && kx > 0 && ky > 0 && kz > 0, "deconSpatial: bl and psf must be 3D and non-empty");
    hipStream_t s = as_stream(stream);
    const size_t N = (size_t)nx * ny * nz;
    const bool need_reg = opt->lambda > 0.0f && opt->regularize_interval > 0 && opt->regularize_interval < opt->niter;
    // the scratch volume serves the edge taper's blur, the regularisation step's Gaussian and the ratio of engines that do not
    // fuse an iteration: a fused loop without regularisation never touches it and does not allocate it (8.6 GB on config C3)
    const bool reg_sched = opt->regularize_interval > 0 && opt->regularize_interval < opt->niter;
    DevBuf ratio, reg, scratch;
    if (need_reg) MI_TRY(reg.alloc(sizeof(float) * N));
    MI_TRY(scratch.alloc(sizeof(double)));
    double delta_prev = 0.0;
    if (opt->stop_criterion > 0.0f) MI_TRY(host_norm(s, bl, N, scratch.as<double>(), &delta_prev));  // before the taper (decon.m:46-50)
    if (!opt->skip_edgetaper) {
        MI_TRY(ratio.alloc(sizeof(float) * N));
        MI_TRY(edgetaper_async(s, bl, ratio.as<float>(), psf, nx, ny, nz, kx, ky, kz, keep_taper));
    }
    mi_rl_ctx* ctx = keep_ctx ? *keep_ctx : nullptr;
    if (!ctx) {
        NoPlacementTrial as_they_come;   // (a handful of iterations per context: see fft_native.h)
        MI_TRY(mi_rl_create(dev, stream, nx, ny, nz, psf, psf_inv, kx, ky, kz, MI_BOUNDARY_ZERO, opt->engine, &ctx));
    }
    if (keep_ctx) *keep_ctx = ctx;
    if (reg_sched || !mi_rl_fuses(ctx)) {
        if (!ratio.p) MI_TRY(ratio.alloc(sizeof(float) * N));
    } else {
        MI_HIP(hipStreamSynchronize(s));  // the taper may still be reading it
        ratio.release();
    }
    int rc = rl_iterate(ctx, s, bl, ratio.as<float>(), reg.as<float>(), scratch.as<double>(), nx, ny, nz, *opt, delta_prev, iters_done);
    hipError_t e = hipStreamSynchronize(s);
    if (!keep_ctx) mi_rl_destroy(ctx);
    if (rc == MI_OK && e != hipSuccess) rc = fail(MI_ERR_HIP, "deconSpatial: %s", hipGetErrorString(e));
    return rc;
}

extern "C" int mi_rl_spatial(int dev, void* stream, float* bl, const float* psf, const float* psf_inv, int nx, int ny, int nz, int kx,
                             int ky, int kz, const mi_rl_options* opt, int* iters_done) {
    return rl_spatial_impl(dev, stream, bl, psf, psf_inv, nx, ny, nz, kx, ky, kz, opt, iters_done, nullptr, nullptr);
}

static int rl_fft_impl(int dev, void* stream, float* bl, const float* psf, int nx, int ny, int nz, int kx, int ky, int kz, int fx, int fy,
                       int fz, const mi_rl_options* opt, int* iters_done, mi_rl_ctx** keep_ctx, TaperKeep** keep_taper) {
    MI_TRY(use_device(dev));
    MI_TRY(check_options(opt));
    MI_REQUIRE(bl && psf, "deconFFT: null pointer");
    MI_REQUIRE(nx > 0 && ny > 0 && nz > 0 && kx > 0 && ky > 0 && kz > 0, "deconFFT: bl and psf must be 3D and non-empty");
    MI_REQUIRE(fx >= nx && fy >= ny && fz >= nz, "pad_block_to_fft_shape: bl [%d %d %d] is larger than FFT shape [%d %d %d], cannot pad",
               nx, ny, nz, fx, fy, fz);
    MI_REQUIRE(fx >= kx && fy >= ky && fz >= kz, "pad_block_to_fft_shape: psf [%d %d %d] is larger than FFT shape [%d %d %d], cannot pad",
               kx, ky, kz, fx, fy, fz);
    hipStream_t s = as_stream(stream);
    const size_t NF = (size_t)fx * fy * fz;
    const bool padded = fx != nx || fy != ny || fz != nz;
    const bool need_reg = opt->lambda > 0.0f && opt->regularize_interval > 0 && opt->regularize_interval < opt->niter;
    const bool reg_sched = opt->regularize_interval > 0 && opt->regularize_interval < opt->niter;
    DevBuf ratio, reg, scratch, blF;
    MI_SPAN_BEGIN(sp0, "deconFFT: allocations");
    if (need_reg) MI_TRY(reg.alloc(sizeof(float) * NF));
    MI_TRY(scratch.alloc(sizeof(double)));
    if (!opt->skip_edgetaper) MI_TRY(ratio.alloc(sizeof(float) * NF));
    MI_SPAN_END(sp0);
    if (!opt->skip_edgetaper) {  // decon.m:143
        MI_SPAN_BEGIN(sp1, "deconFFT: edgetaper (ends with a wait)");
        MI_TRY(edgetaper_async(s, bl, ratio.as<float>(), psf, nx, ny, nz, kx, ky, kz, keep_taper));
        MI_SPAN_END(sp1);
    }
    float* work_bl = bl;
    if (padded) {  // decon.m:144
        MI_SPAN_BEGIN(sp2, "deconFFT: padded copy (alloc + enqueue)");
        MI_TRY(blF.alloc(sizeof(float) * NF));
        MI_TRY(mi_pad_center(dev, stream, bl, nx, ny, nz, blF.as<float>(), fx, fy, fz));
        work_bl = blF.as<float>();
        MI_SPAN_END(sp2);
    }
    double delta_prev = 0.0;
    if (opt->stop_criterion > 0.0f) MI_TRY(host_norm(s, work_bl, NF, scratch.as<double>(), &delta_prev));  // decon.m:145-147
    mi_rl_ctx* ctx = keep_ctx ? *keep_ctx : nullptr;
    const int engine = opt->engine == MI_ENGINE_DIRECT ? MI_ENGINE_DIRECT : MI_ENGINE_FFT;
    if (!ctx) {
        MI_SPAN_BEGIN(sp3, "deconFFT: mi_rl_create (a new shape)");
        NoPlacementTrial as_they_come;   // (a handful of iterations per context: see fft_native.h)
        // where the PSF's samples land on the circular grid: ifftshift(zero-pad-centre(psf)) on the grid opt->psf_grid names
        // (default: the FFT shape; see mi_rl_options)
        const int circ[3] = {MI_BOUNDARY_CIRCULAR, MI_BOUNDARY_CIRCULAR, MI_BOUNDARY_CIRCULAR};
        const int F[3] = {fx, fy, fz}, K[3] = {kx, ky, kz};
        int shift[3];
        for (int d = 0; d < 3; ++d) {
            const int g = opt->psf_grid[d] > 0 ? opt->psf_grid[d] : F[d];
            MI_REQUIRE(g >= K[d], "deconFFT: psf_grid extent %d smaller than the PSF extent %d on axis %d", g, K[d], d);
            shift[d] = g / 2 - (g - K[d]) / 2;
        }
        MI_TRY(mi_rl_create_ex(dev, stream, fx, fy, fz, psf, nullptr, kx, ky, kz, circ, shift, engine, &ctx));
        MI_SPAN_END(sp3);
    }
    if (keep_ctx) *keep_ctx = ctx;
    MI_SPAN_BEGIN(sp4, "deconFFT: wait + release of the taper's work");
    if (reg_sched || !mi_rl_fuses(ctx)) {  // (see rl_spatial_impl)
        if (!ratio.p) MI_TRY(ratio.alloc(sizeof(float) * NF));
    } else {
        MI_HIP(hipStreamSynchronize(s));
        ratio.release();
    }
    MI_SPAN_END(sp4);
    MI_SPAN_BEGIN(sp5, "deconFFT: rl_iterate (enqueue)");
    int rc = rl_iterate(ctx, s, work_bl, ratio.as<float>(), reg.as<float>(), scratch.as<double>(), fx, fy, fz, *opt, delta_prev, iters_done);
    if (rc == MI_OK && padded) rc = mi_crop_center(dev, stream, work_bl, fx, fy, fz, bl, nx, ny, nz);  // decon.m:203
    MI_SPAN_END(sp5);
    MI_SPAN_BEGIN(sp6, "deconFFT: final wait for the stream");
    hipError_t e = hipStreamSynchronize(s);
    MI_SPAN_END(sp6);
    if (!keep_ctx) mi_rl_destroy(ctx);
    if (rc == MI_OK && e != hipSuccess) rc = fail(MI_ERR_HIP, "deconFFT: %s", hipGetErrorString(e));
    return rc;
}

extern "C" int mi_rl_fft(int dev, void* stream, float* bl, const float* psf, int nx, int ny, int nz, int kx, int ky, int kz, int fx,
                         int fy, int fz, const mi_rl_options* opt, int* iters_done) {
    return rl_fft_impl(dev, stream, bl, psf, nx, ny, nz, kx, ky, kz, fx, fy, fz, opt, iters_done, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------------ deconFFT_Wiener
namespace {

constexpr int kThreads = 256;
inline unsigned stream_grid(size_t n_items) {
    const size_t b = (n_items + kThreads - 1) / kThreads, cap = 256 * 16;  // grid-stride beyond 16 work-groups per CU
    return static_cast<unsigned>(b < 1 ? 1 : (b > cap ? cap : b));
}

// otf_new = F{Y} conj(F{X}) / max(|F{X}|^2, eps) (decon.m:282-288), element-wise on the untangled half spectra; `scale` folds in
// the 2 / (Fx Fy Fz) of the pipeline's unnormalised inverse
__global__ __launch_bounds__(kThreads) void k_wiener_otf(const float2* __restrict__ fy, const float2* __restrict__ fx, float2* __restrict__ otf,
                                                       size_t n, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float2 y = fy[i], x = fx[i];
        const float den = fmaxf(x.x * x.x + x.y * x.y, kEpsSingle);
        const float re = y.x * x.x + y.y * x.y, im = y.y * x.x - y.x * x.y;  // y * conj(x)
        otf[i] = make_float2(re / den * scale, im / den * scale);
    }
}

__global__ __launch_bounds__(kThreads) void k_delta(float* __restrict__ v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) v[i] = i == 0 ? 1.0f : 0.0f;
}

// psf = max(full(centre box), 0); psf /= sum(psf) when the sum is positive (decon.m:293-301).  One work-group: a PSF is small.
__global__ __launch_bounds__(256) void k_wiener_psf(const float* __restrict__ full, float* __restrict__ psf, int fx, int fy, int kx, int ky,
                                                     int kz, int cx, int cy, int cz) {
    __shared__ float part[256];
    const int n = kx * ky * kz;
    float acc = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int x = i % kx, r = i / kx, y = r % ky, z = r / ky;
        const float v = fmaxf(full[((size_t)(cz + z) * fy + (cy + y)) * fx + (cx + x)], 0.0f);
        psf[i] = v;
        acc += v;
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    const float sum = part[0];
    if (sum > 0.0f)
        for (int i = threadIdx.x; i < n; i += 256) psf[i] = psf[i] / sum;
}

}  // namespace

extern "C" int mi_rl_fft_wiener(int dev, void* stream, float* bl, float* psf, int nx, int ny, int nz, int kx, int ky, int kz, int fx,
                                int fy, int fz, const mi_rl_options* opt, int* iters_done) {
    MI_TRY(use_device(dev));
    MI_TRY(check_options(opt));
    MI_REQUIRE(bl && psf, "deconFFT_Wiener: null pointer");
    MI_REQUIRE(nx > 0 && ny > 0 && nz > 0 && kx > 0 && ky > 0 && kz > 0, "deconFFT_Wiener: bl and psf must be 3D and non-empty");
    MI_REQUIRE(fx >= nx && fy >= ny && fz >= nz, "pad_block_to_fft_shape: bl [%d %d %d] is larger than FFT shape [%d %d %d], cannot pad",
               nx, ny, nz, fx, fy, fz);
    MI_REQUIRE(fx >= kx && fy >= ky && fz >= kz, "pad_block_to_fft_shape: psf [%d %d %d] is larger than FFT shape [%d %d %d], cannot pad",
               kx, ky, kz, fx, fy, fz);
    // the spectra F{Y}, F{X} live in the hand-written pipeline's own layout; rocFFT-only shapes are not served
    if (!(mi_fft_good_size(fx, 0) == fx && mi_fft_good_size(fy, 1) == fy && mi_fft_good_size(fz, 2) == fz))
        return fail(MI_ERR_UNSUPPORTED, "deconFFT_Wiener: fft_shape [%d %d %d] must be made of extents mi_fft_good_size accepts (2^a, 3 * 2^a or 9 * 2^a; x even)", fx, fy, fz);
    hipStream_t s = as_stream(stream);
    const size_t NF = (size_t)fx * fy * fz;
    const bool padded = fx != nx || fy != ny || fz != nz;
    const int nk = kx * ky * kz;
    DevBuf ratio, reg, scratch, blF, psf_cur, FY, FX;
    MI_TRY(ratio.alloc(sizeof(float) * NF));
    if (opt->lambda > 0.0f && opt->regularize_interval > 0) MI_TRY(reg.alloc(sizeof(float) * NF));
    MI_TRY(scratch.alloc(sizeof(double)));
    MI_TRY(psf_cur.alloc(sizeof(float) * nk));
    MI_HIP(hipMemcpyAsync(psf_cur.p, psf, sizeof(float) * nk, hipMemcpyDeviceToDevice, s));
    if (!opt->skip_edgetaper) MI_TRY(edgetaper_async(s, bl, ratio.as<float>(), psf, nx, ny, nz, kx, ky, kz));  // decon.m:212
    float* work_bl = bl;
    if (padded) {  // decon.m:213
        MI_TRY(blF.alloc(sizeof(float) * NF));
        MI_TRY(mi_pad_center(dev, stream, bl, nx, ny, nz, blF.as<float>(), fx, fy, fz));
        work_bl = blF.as<float>();
    }
    double delta_prev = 0.0;
    if (opt->stop_criterion > 0.0f) MI_TRY(host_norm(s, work_bl, NF, scratch.as<double>(), &delta_prev));  // decon.m:215
    mi_rl_ctx* ctx = nullptr;
    const int b[3] = {MI_BOUNDARY_CIRCULAR, MI_BOUNDARY_CIRCULAR, MI_BOUNDARY_CIRCULAR};
    {
        NoPlacementTrial as_they_come;
        MI_TRY(rl_create(dev, stream, fx, fy, fz, psf, nullptr, kx, ky, kz, b, nullptr, MI_ENGINE_FFT, /*fixed_psf=*/false, &ctx));
    }
    struct Guard { mi_rl_ctx* c; hipStream_t s; ~Guard() { (void)hipStreamSynchronize(s); mi_rl_destroy(c); } } guard{ctx, s};
    FftEngine* fe = ctx->fft;
    MI_REQUIRE(fe->native, "deconFFT_Wiener: the hand-written FFT pipeline refused fft_shape [%d %d %d]", fx, fy, fz);
    NativeFft* nf = fe->native;
    const size_t items = nf->otf_items();
    MI_TRY(FY.alloc(sizeof(float4) * items));
    MI_TRY(FX.alloc(sizeof(float4) * items));
    float4 *fy_p = FY.as<float4>(), *fx_p = FX.as<float4>();
    const float nscale = 2.0f / (float)((double)fx * fy * fz);
    const float sig[3] = {0.5f, 0.5f, 0.5f};
    const int k3[3] = {3, 3, 3};
    const int cx = (fx - kx) / 2, cy = (fy - ky) / 2, cz = (fz - kz) / 2;  // floor((fft_shape - psf_sz) / 2) + 1, 0-based (decon.m:236)
    ConvEpilogue none;
    int done = 0;
    for (int i = 1; i <= opt->niter; ++i) {
        if (i > 1) MI_TRY(fe->set_psf(s, psf_cur.as<float>()));  // decon.m:243-245 (iteration 1: the OTF mi_rl_create built)
        const bool reg_i = opt->regularize_interval > 0 && (i % opt->regularize_interval) == 0;
        // F{Y} (decon.m:248-256) only feeds the Wiener quotient here: the RL step transforms bl itself
        if (i > 1 && reg_i) MI_TRY(gauss3d_async(s, work_bl, ratio.as<float>(), fx, fy, fz, sig, opt->gauss_taps == 3 ? k3 : nullptr));
        if (i < opt->niter && (i == 1 || reg_i)) MI_TRY(nf->spectrum(s, work_bl, fy_p, 1.0f));
        if (reg_i && opt->lambda > 0.0f && i < opt->niter) {  // decon.m:273-275
            MI_TRY(mi_rl_forward_ratio(ctx, s, work_bl, ratio.as<float>()));
            MI_TRY(mi_rl_reg_term(dev, s, work_bl, reg.as<float>(), fx, fy, fz));
            MI_TRY(mi_rl_adjoint_update(ctx, s, ratio.as<float>(), work_bl, opt->lambda, reg.as<float>()));
        } else {
            MI_TRY(mi_rl_iterate(ctx, s, work_bl, ratio.as<float>(), 1));
        }
        if (i < opt->niter) {  // decon.m:281-307
            MI_TRY(nf->spectrum(s, work_bl, fx_p, 1.0f));
            hipLaunchKernelGGL(k_wiener_otf, dim3(stream_grid(2 * items)), dim3(kThreads), 0, s, reinterpret_cast<const float2*>(fy_p),
                               reinterpret_cast<const float2*>(fx_p), reinterpret_cast<float2*>(nf->otf()), 2 * items, nscale);
            MI_TRY(launch_check("k_wiener_otf"));
            std::swap(fy_p, fx_p);  // F{X} is the next iteration's F{Y}
            // real(ifftn(otf_new)): the response of the new OTF to a unit impulse at the origin
            hipLaunchKernelGGL(k_delta, dim3(stream_grid(NF)), dim3(kThreads), 0, s, ratio.as<float>(), NF);
            MI_TRY(launch_check("k_delta"));
            MI_TRY(fe->conv(s, ratio.as<float>(), false, ratio.as<float>(), EPI_NONE, none));
            hipLaunchKernelGGL(k_wiener_psf, dim3(1), dim3(256), 0, s, ratio.as<float>(), psf_cur.as<float>(), fx, fy, kx, ky, kz, cx, cy, cz);
            MI_TRY(launch_check("k_wiener_psf"));
        }
        done = i;
        if (opt->stop_criterion > 0.0f) {  // decon.m:310-317: no i > 1 guard in this variant
            double cur = 0.0;
            MI_TRY(host_norm(s, work_bl, NF, scratch.as<double>(), &cur));
            if (std::fabs(delta_prev - cur) / delta_prev * 100.0 <= (double)opt->stop_criterion) break;
            delta_prev = cur;
        }
    }
    if (iters_done) *iters_done = done;
    if (padded) MI_TRY(mi_crop_center(dev, stream, work_bl, fx, fy, fz, bl, nx, ny, nz));  // decon.m:320
    MI_HIP(hipMemcpyAsync(psf, psf_cur.p, sizeof(float) * nk, hipMemcpyDeviceToDevice, s));
    MI_HIP(hipStreamSynchronize(s));
    return MI_OK;
}

extern "C" int mi_decon(int dev, void* stream, float* bl, const float* psf, const float* psf_inv, int nx, int ny, int nz, int kx, int ky,
                        int kz, const mi_rl_options* opt, int use_fft, const int* fft_shape_xyz, int adaptive_psf, int* iters_done) {
    if (use_fft) {
        int f[3] = {nx, ny, nz};
        if (fft_shape_xyz) { f[0] = fft_shape_xyz[0]; f[1] = fft_shape_xyz[1]; f[2] = fft_shape_xyz[2]; }
        if (adaptive_psf) {  // decon.m:15-16; the caller's PSF stays as it was (the refined one is deconFFT_Wiener's local)
            MI_REQUIRE(psf && kx > 0 && ky > 0 && kz > 0, "deconFFT_Wiener: null pointer");
            MI_TRY(use_device(dev));
            DevBuf p;
            MI_TRY(p.alloc(sizeof(float) * (size_t)kx * ky * kz));
            MI_HIP(hipMemcpyAsync(p.p, psf, sizeof(float) * (size_t)kx * ky * kz, hipMemcpyDeviceToDevice, as_stream(stream)));
            return mi_rl_fft_wiener(dev, stream, bl, p.as<float>(), nx, ny, nz, kx, ky, kz, f[0], f[1], f[2], opt, iters_done);
        }
        return mi_rl_fft(dev, stream, bl, psf, nx, ny, nz, kx, ky, kz, f[0], f[1], f[2], opt, iters_done);
    }
    return mi_rl_spatial(dev, stream, bl, psf, psf_inv, nx, ny, nz, kx, ky, kz, opt, iters_done);  // adaptive_psf: FFT path only (decon.m:14-22)
}

// ------------------------------------------------------------------------------------------------ deconvolution plans
// The blocks of a volume share shape and PSF (LsDeconv.m:620-668): a plan keeps what `decon` would rebuild for every block --
// the RL context (OTF, twiddles, scratch) and the FFT engine of edgetaper_3d's blur -- and rebuilds it only when shape, PSF or
// engine change.  One plan per worker thread (it is not re-entrant).
struct mi_decon_plan {
    int dev = 0;
    int key[15] = {0};          // nx ny nz kx ky kz use_fft fx fy fz engine has_inv psf_grid[3]
    std::vector<float> psf, psf_inv;
    mi_rl_ctx* ctx = nullptr;
    TaperKeep* taper = nullptr;
    void drop() {
        if (ctx) mi_rl_destroy(ctx);
        taper_keep_free(taper);
        ctx = nullptr;
        taper = nullptr;
    }
};

extern "C" int mi_decon_plan_create(int dev, mi_decon_plan** out) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(out, "mi_decon_plan_create: null pointer");
    *out = new (std::nothrow) mi_decon_plan;
    if (!*out) return fail(MI_ERR_NOMEM, "mi_decon_plan_create: out of host memory");
    (*out)->dev = dev;
    return MI_OK;
}

extern "C" int mi_decon_plan_destroy(mi_decon_plan* plan) {
    if (!plan) return MI_OK;
    (void)hipSetDevice(plan->dev);
    (void)hipDeviceSynchronize();
    plan->drop();
    delete plan;
    return MI_OK;
}

extern "C" int mi_decon_plan_run(mi_decon_plan* plan, void* stream, float* bl, const float* psf, const float* psf_inv, int nx, int ny, int nz,
                                 int kx, int ky, int kz, const mi_rl_options* opt, int use_fft, const int* fft_shape_xyz, int adaptive_psf,
                                 int* iters_done) {
    MI_REQUIRE(plan && opt && psf, "mi_decon_plan_run: null pointer");
    const int dev = plan->dev;
    if (adaptive_psf && use_fft)  // the Wiener variant rebuilds its OTF every iteration: nothing to keep
        return mi_decon(dev, stream, bl, psf, psf_inv, nx, ny, nz, kx, ky, kz, opt, use_fft, fft_shape_xyz, adaptive_psf, iters_done);
    MI_TRY(use_device(dev));
    MI_REQUIRE(kx > 0 && ky > 0 && kz > 0, "decon: psf must be 3D and non-empty");
    int f[3] = {nx, ny, nz};
    if (use_fft && fft_shape_xyz) { f[0] = fft_shape_xyz[0]; f[1] = fft_shape_xyz[1]; f[2] = fft_shape_xyz[2]; }
    const int key[15] = {nx, ny, nz, kx, ky, kz, use_fft ? 1 : 0, f[0], f[1], f[2], opt->engine, (!use_fft && psf_inv) ? 1 : 0,
                         use_fft ? opt->psf_grid[0] : 0, use_fft ? opt->psf_grid[1] : 0, use_fft ? opt->psf_grid[2] : 0};
    // the PSFs are small: compare their values with the ones the kept objects were built from
    hipStream_t s = as_stream(stream);
    const size_t nk = (size_t)kx * ky * kz;
    std::vector<float> h(nk), hi;
    MI_SPAN_BEGIN(sp0, "plan_run: psf fetch + wait for the stream");
    MI_HIP(hipMemcpyAsync(h.data(), psf, sizeof(float) * nk, hipMemcpyDeviceToHost, s));
    if (key[11]) {
        hi.resize(nk);
        MI_HIP(hipMemcpyAsync(hi.data(), psf_inv, sizeof(float) * nk, hipMemcpyDeviceToHost, s));
    }
    MI_HIP(hipStreamSynchronize(s));
    MI_SPAN_END(sp0);
    if (std::memcmp(key, plan->key, sizeof(key)) != 0 || h != plan->psf || hi != plan->psf_inv) {
        MI_SPAN_BEGIN(sp1, "plan_run: drop of the kept objects");
        MI_HIP(hipStreamSynchronize(s));
        plan->drop();
        std::memcpy(plan->key, key, sizeof(key));
        plan->psf.swap(h);
        plan->psf_inv.swap(hi);
        MI_SPAN_END(sp1);
    }
    int rc = use_fft ? rl_fft_impl(dev, stream, bl, psf, nx, ny, nz, kx, ky, kz, f[0], f[1], f[2], opt, iters_done, &plan->ctx, &plan->taper)
                     : rl_spatial_impl(dev, stream, bl, psf, psf_inv, nx, ny, nz, kx, ky, kz, opt, iters_done, &plan->ctx, &plan->taper);
    if (rc != MI_OK) {  // whatever was half built is not trusted
        (void)hipStreamSynchronize(s);
        plan->drop();
        plan->key[0] = 0;
    }
    return rc;
}
