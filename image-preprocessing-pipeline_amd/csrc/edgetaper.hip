// Edge taper (replaces edgetaper_3d.m:13-44 + make_taper.m:13-35 + the conv3d_gpu call inside).
//
// blur = conv3d_replicate(bl, psf / sum(psf)) is only needed where the separable taper mask is < 1,
// i.e. in a border shell of width max(8, round(k/2)) per axis: the direct engine skips every tile
// that lies on the mask plateau, and the blend touches only shell voxels.  For C3/C4-sized PSFs this
// removes 85-90 % of the reference's work for this step.
#include <cmath>
#include <cstdlib>
#include <new>
#include <vector>

#include "fftconv.h"

namespace mi {
namespace {

// make_taper.m:13-35
std::vector<float> make_taper(int dimsz, int taper_width) {
    int w = std::min(taper_width, dimsz / 2);
    std::vector<float> t;
    if (w <= 0) return std::vector<float>(dimsz, 1.0f);
    std::vector<double> ramp(w + 1);
    // MATLAB linspace(0,1,w+1): d1 + (0:n1)*(d2-d1)/n1 with the last sample pinned to d2
    for (int i = 0; i <= w; ++i) ramp[i] = (i == w) ? 1.0 : ((double)i * 1.0) / (double)w;
    for (int i = 0; i <= w; ++i) t.push_back((float)ramp[i]);
    if (2 * w < dimsz)
        for (int i = 0; i < dimsz - 2 * w; ++i) t.push_back(1.0f);
    for (int i = w - 1; i >= 0; --i) t.push_back((float)ramp[i]);
    if ((int)t.size() > dimsz) t.resize(dimsz);
    while ((int)t.size() < dimsz) t.push_back(1.0f);
    return t;
}

int matlab_round_half(int k) {  // round(k/2) with half away from zero, k > 0
    return (k + 1) / 2;
}

__global__ __launch_bounds__(256) void k_taper_blend(float* __restrict__ bl, const float* __restrict__ blur,
                                                      const float* __restrict__ tx, const float* __restrict__ ty,
                                                      const float* __restrict__ tz, int nx, int ny, int nz) {
    const size_t total = (size_t)nx * ny * nz;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % nx);
        const size_t r = i / nx;
        const int y = (int)(r % ny), z = (int)(r / ny);
        const float m = (tx[x] * ty[y]) * tz[z];  // mask built x, then y, then z (edgetaper_3d.m:30-39)
        if (m != 1.0f) bl[i] = m * bl[i] + (1.0f - m) * blur[i];
    }
}

}  // namespace

int edgetaper_async(hipStream_t s, float* bl, float* work, const float* psf, int nx, int ny, int nz, int kx, int ky, int kz,
                    FftEngine** keep) {
    MI_REQUIRE(bl && work && psf && bl != work, "edgetaper_3d: null or aliased buffers");
    MI_REQUIRE(nx > 0 && ny > 0 && nz > 0 && kx > 0 && ky > 0 && kz > 0, "edgetaper_3d: bl and psf must be 3D");
    const int n[3] = {nx, ny, nz}, k[3] = {kx, ky, kz};
    std::vector<float> taper[3];
    ConvEpilogue epi;
    size_t off[3], tot = 0;
    for (int d = 0; d < 3; ++d) {
        taper[d] = make_taper(n[d], std::max(8, matlab_round_half(k[d])));  // edgetaper_3d.m:32
        int lo = 0, hi = 0;
        for (int i = 0; i < n[d]; ++i)
            if (taper[d][i] == 1.0f) { lo = i; break; }
        for (int i = n[d] - 1; i >= 0; --i)
            if (taper[d][i] == 1.0f) { hi = i + 1; break; }
        bool contiguous = hi > lo;
        for (int i = lo; i < hi; ++i) contiguous = contiguous && taper[d][i] == 1.0f;
        epi.plat_lo[d] = contiguous ? lo : 0;
        epi.plat_hi[d] = contiguous ? hi : 0;
        off[d] = tot;
        tot += n[d];
    }
    DevBuf dtaper, kf;
    MI_TRY(dtaper.alloc(sizeof(float) * tot));
    std::vector<float> host(tot);
    for (int d = 0; d < 3; ++d) std::copy(taper[d].begin(), taper[d].end(), host.begin() + off[d]);
    MI_HIP(hipMemcpyAsync(dtaper.p, host.data(), sizeof(float) * tot, hipMemcpyHostToDevice, s));
    // blur = conv3d_gpu(bl, psf / sum): direct engine on the shell tiles only, or one FFT convolution of the replicate-padded
    // volume (same result within fp32 rounding).  Measured on C3 (31x31x61 PSF): 980 ms on the shell, 73 ms through the
    // hand-written FFT pipeline incl. building its OTF (24 ps per grid point), 2.8 s through rocFFT incl. plan creation.
    double shell = 1.0;
    int need[3], F[3];
    const int bnd_rep[3] = {MI_BOUNDARY_REPLICATE, MI_BOUNDARY_REPLICATE, MI_BOUNDARY_REPLICATE};
    for (int d = 0; d < 3; ++d) {
        shell *= (double)std::max(0, epi.plat_hi[d] - epi.plat_lo[d]) / n[d];
        need[d] = n[d] + k[d] - 1;
    }
    const bool native_grid = choose_fft_lengths(need, bnd_rep, F);
    const double nf = (double)F[0] * F[1] * F[2];
    shell = 1.0 - shell;
    const double taps = (double)kx * ky * kz, nvox = (double)nx * ny * nz;
    const double t_direct = shell * nvox * 2.0 * taps / 50e12;
    const double t_fft = native_grid ? nf * 25e-12 + 0.02 : nf * 220.0 / 4e12 + 2.5;
    const bool odd = (kx & 1) && (ky & 1) && (kz & 1);
    bool use_fft = odd && t_fft < t_direct;
    if (const char* f = std::getenv("MI_EDGETAPER_ENGINE")) use_fft = odd && f[0] == 'f';  // tests / experiments: "fft" | "direct"
    if (use_fft) {
        // `keep` (a deconvolution plan): the engine -- the OTF of the normalised PSF on the replicate-padded grid -- outlives
        // the call and serves the next block of the same shape and PSF
        FftEngine local, *fe = &local;
        bool built = false;
        if (keep && *keep) fe = *keep;
        else {
            if (keep) {
                fe = new (std::nothrow) FftEngine;
                if (!fe) return fail(MI_ERR_NOMEM, "edgetaper_3d: out of host memory");
            }
            DevBuf pn;
            int rc = normalised_psf(s, psf, kx * ky * kz, pn);
            const int bnd[3] = {MI_BOUNDARY_REPLICATE, MI_BOUNDARY_REPLICATE, MI_BOUNDARY_REPLICATE};
            int shift[3];
            for (int d = 0; d < 3; ++d) shift[d] = k[d] - 1 - conv_kernel_offset(k[d], MI_BOUNDARY_REPLICATE);
            if (rc == MI_OK) rc = fe->init(s, n, k, bnd, shift, pn.as<float>(), nullptr, false);
            hipError_t e = hipStreamSynchronize(s);  // pn dies here
            if (rc == MI_OK && e != hipSuccess) rc = fail(MI_ERR_HIP, "edgetaper_3d: %s", hipGetErrorString(e));
            if (rc != MI_OK) {
                if (keep) delete fe;
                return rc;
            }
            if (keep) *keep = fe;
            built = true;
        }
        (void)built;
        MI_TRY(fe->conv(s, bl, false, work, EPI_NONE, ConvEpilogue()));
        if (!keep) MI_HIP(hipStreamSynchronize(s));  // the local engine's buffers die at scope exit
    } else {
        int kxp = 0;
        MI_TRY(direct_prepare_psf(s, psf, kx, ky, kz, /*normalise=*/true, /*flip=*/true, kf, &kxp));
        MI_TRY(direct_conv_launch(s, bl, kf.as<float>(), work, nx, ny, nz, kx, ky, kz, kxp, MI_BOUNDARY_REPLICATE, EPI_TAPER_SHELL, epi));
    }
    const size_t total = (size_t)nx * ny * nz;
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    const float* t = dtaper.as<float>();
    hipLaunchKernelGGL(k_taper_blend, dim3((unsigned)blocks), dim3(256), 0, s, bl, work, t + off[0], t + off[1], t + off[2], nx, ny, nz);
    MI_TRY(launch_check("k_taper_blend"));
    // host vector / DevBufs die at scope exit: the H2D copy source must outlive the copy
    MI_HIP(hipStreamSynchronize(s));
    return MI_OK;
}

}  // namespace mi

extern "C" int mi_edgetaper3d(int dev, void* stream, float* bl, float* work, const float* psf, int nx, int ny, int nz, int kx,
                              int ky, int kz) {
    MI_TRY(mi::use_device(dev));
    return mi::edgetaper_async(mi::as_stream(stream), bl, work, psf, nx, ny, nz, kx, ky, kz, nullptr);
}
