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

// one work-group per row (y, z): no 64-bit index arithmetic per voxel; rows on the plateau of both other axes only visit their
// two tapered ends (x_lo, x_hi: first and one-past-last x with tx == 1)
__global__ __launch_bounds__(256) void k_taper_blend(float* __restrict__ bl, const float* __restrict__ blur,
                                                      const float* __restrict__ tx, const float* __restrict__ ty,
                                                      const float* __restrict__ tz, int nx, int ny, int nz, int x_lo, int x_hi) {
    const int y = blockIdx.x, z = blockIdx.y;
    const float myz = ty[y], mz = tz[z];
    const bool plateau = myz == 1.0f && mz == 1.0f;
    const size_t row = ((size_t)z * ny + y) * (size_t)nx;
    for (int x = threadIdx.x; x < nx; x += 256) {
        if (plateau && x >= x_lo && x < x_hi) {  // skip to the upper end
            x += ((x_hi - x + 255) / 256) * 256 - 256;
            continue;
        }
        const float m = (tx[x] * myz) * mz;  // mask built x, then y, then z (edgetaper_3d.m:30-39)
        if (m != 1.0f) bl[row + x] = m * bl[row + x] + (1.0f - m) * blur[row + x];
    }
}

// dst (grid F, x fastest) = src[clamp(origin + q)] for q < extent per axis, 0 beyond: the replicate-padded neighbourhood of a face slab
__global__ __launch_bounds__(256) void k_pack_clamped(const float* __restrict__ src, int nx, int ny, int nz, float* __restrict__ dst, int Fx, int Fy,
                                                      int Fz, int ox, int oy, int oz, int ex, int ey, int ez) {
    const size_t total = (size_t)Fx * Fy * Fz;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Fx);
        const size_t r = i / Fx;
        const int y = (int)(r % Fy), z = (int)(r / Fy);
        float v = 0.0f;
        if (x < ex && y < ey && z < ez) {
            const int sx = min(max(ox + x, 0), nx - 1), sy = min(max(oy + y, 0), ny - 1), sz = min(max(oz + z, 0), nz - 1);
            v = src[((size_t)sz * ny + sy) * nx + sx];
        }
        dst[i] = v;
    }
}

// work[box] = grid[(p - box.lo) + c]: the owned region of a slab out of its convolved grid
__global__ __launch_bounds__(256) void k_unpack_box(const float* __restrict__ grid, int Fx, int Fy, float* __restrict__ work, int nx, int ny, int x0,
                                                    int y0, int z0, int bx, int by, int bz, int cx, int cy, int cz) {
    const size_t total = (size_t)bx * by * bz;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % bx);
        const size_t r = i / bx;
        const int y = (int)(r % by), z = (int)(r / by);
        work[((size_t)(z0 + z) * ny + y0 + y) * nx + x0 + x] = grid[((size_t)(z + cz) * Fy + y + cy) * Fx + x + cx];
    }
}

unsigned stream_blocks(size_t n) {
    size_t b = (n + 255) / 256;
    return (unsigned)(b > 256 * 32 ? 256 * 32 : (b < 1 ? 1 : b));
}

}  // namespace

// what a deconvolution plan keeps of the taper between blocks of one shape and PSF: the engine of the whole replicate-padded
// block, or the three engines of the face slabs (x, y, z)
struct TaperKeep {
    FftEngine* full = nullptr;
    FftEngine* slab[3] = {nullptr, nullptr, nullptr};
    ~TaperKeep() {
        delete full;
        for (FftEngine* e : slab) delete e;
    }
};
void taper_keep_free(TaperKeep* k) { delete k; }

namespace {

// The blur is only needed where the mask is below 1: the shell of the block.  Shell = six face slabs, cut so that they do not
// overlap (x slabs: full y, z; y slabs: x plateau only; z slabs: x and y plateau): each pair of opposite slabs goes through ONE
// circular FFT engine on a grid just large enough for the slab plus the PSF's reach -- its input is the clamped (= replicate
// padded, conv3d_gpu.cu:82-91) neighbourhood of the slab gathered from the block, its output lands in `work` at the slab's
// voxels.  On config C3 that is 1.3 G grid points at the circular pipeline's 8 ps instead of 3.1 G at the padded one's 24 ps.
struct SlabPlan {
    int lo[3], hi[3];      // plateau [lo, hi) per axis: the shell is [0, lo) and [hi, n)
    int F[3][3];           // grid of the slabs of axis d
    int thick[3];          // owned thickness of the thicker of the two slabs of axis d
    double points;         // grid points of all six convolutions
    bool ok;
};

SlabPlan plan_slabs(const int n[3], const int k[3], const ConvEpilogue& epi) {
    SlabPlan sp{};
    sp.ok = true;
    sp.points = 0.0;
    for (int d = 0; d < 3; ++d) {
        sp.lo[d] = epi.plat_lo[d];
        sp.hi[d] = epi.plat_hi[d];
        sp.thick[d] = std::max(sp.lo[d], n[d] - sp.hi[d]);
        if (!(sp.hi[d] > sp.lo[d]) || (k[d] & 1) == 0) sp.ok = false;  // no plateau on this axis: the shell is the whole block
    }
    if (!sp.ok) return sp;
    for (int d = 0; d < 3; ++d) {
        if (sp.thick[d] == 0) { sp.F[d][0] = sp.F[d][1] = sp.F[d][2] = 0; continue; }
        for (int a = 0; a < 3; ++a) {
            // owned extent along a: the slab's thickness on its own axis; on the others the whole block (x slabs), or the plateau
            // of the axes already covered by earlier slabs
            int owned = a == d ? sp.thick[d] : (a < d ? sp.hi[a] - sp.lo[a] : n[a]);
            const int e = owned + k[a] - 1;
            const int g = mi_fft_good_size(e, a);
            if (g <= 0) { sp.ok = false; return sp; }
            sp.F[d][a] = g;
        }
        sp.points += 2.0 * (double)sp.F[d][0] * sp.F[d][1] * sp.F[d][2];
    }
    return sp;
}

int edgetaper_slabs(hipStream_t s, const float* bl, float* work, const float* psf_norm, const int n[3], const int k[3], const SlabPlan& sp,
                    TaperKeep* keep) {
    const int circ[3] = {MI_BOUNDARY_CIRCULAR, MI_BOUNDARY_CIRCULAR, MI_BOUNDARY_CIRCULAR};
    int shift[3], c[3];
    for (int a = 0; a < 3; ++a) {
        shift[a] = k[a] - 1 - conv_kernel_offset(k[a], MI_BOUNDARY_REPLICATE);  // sample j acts at offset j - shift
        c[a] = k[a] / 2;                                                         // reach towards lower indices (odd k)
    }
    DevBuf gin, gout;
    for (int d = 0; d < 3; ++d) {
        if (sp.thick[d] == 0) continue;
        const int* F = sp.F[d];
        const size_t G = (size_t)F[0] * F[1] * F[2];
        if (gin.bytes < sizeof(float) * G) {
            MI_SPAN_BEGIN(spa, "edgetaper: slab buffers (wait + alloc)");
            MI_HIP(hipStreamSynchronize(s));
            MI_TRY(gin.alloc(sizeof(float) * G));
            MI_TRY(gout.alloc(sizeof(float) * G));
            MI_SPAN_END(spa);
        }
        FftEngine local, *fe = &local;
        if (keep) {
            if (!keep->slab[d]) {
                keep->slab[d] = new (std::nothrow) FftEngine;
                if (!keep->slab[d]) return fail(MI_ERR_NOMEM, "edgetaper_3d: out of host memory");
                MI_SPAN_BEGIN(spi, "edgetaper: slab engine init (a new shape)");
                int rc = keep->slab[d]->init(s, F, k, circ, shift, psf_norm, nullptr, false);
                MI_SPAN_END(spi);
                if (rc != MI_OK) { delete keep->slab[d]; keep->slab[d] = nullptr; return rc; }
            }
            fe = keep->slab[d];
        } else {
            MI_TRY(local.init(s, F, k, circ, shift, psf_norm, nullptr, false));
        }
        for (int side = 0; side < 2; ++side) {
            int b0[3], b1[3];  // owned box
            for (int a = 0; a < 3; ++a) {
                if (a == d) { b0[a] = side == 0 ? 0 : sp.hi[a]; b1[a] = side == 0 ? sp.lo[a] : n[a]; }
                else if (a < d) { b0[a] = sp.lo[a]; b1[a] = sp.hi[a]; }
                else { b0[a] = 0; b1[a] = n[a]; }
            }
            if (b1[d] <= b0[d]) continue;
            const int ex = b1[0] - b0[0] + k[0] - 1, ey = b1[1] - b0[1] + k[1] - 1, ez = b1[2] - b0[2] + k[2] - 1;
            hipLaunchKernelGGL(k_pack_clamped, dim3(stream_blocks(G)), dim3(256), 0, s, bl, n[0], n[1], n[2], gin.as<float>(), F[0], F[1], F[2],
                               b0[0] - c[0], b0[1] - c[1], b0[2] - c[2], ex, ey, ez);
            MI_TRY(launch_check("k_pack_clamped"));
            MI_TRY(fe->conv(s, gin.as<float>(), false, gout.as<float>(), EPI_NONE, ConvEpilogue()));
            const size_t box = (size_t)(b1[0] - b0[0]) * (b1[1] - b0[1]) * (b1[2] - b0[2]);
            hipLaunchKernelGGL(k_unpack_box, dim3(stream_blocks(box)), dim3(256), 0, s, gout.as<float>(), F[0], F[1], work, n[0], n[1], b0[0], b0[1],
                               b0[2], b1[0] - b0[0], b1[1] - b0[1], b1[2] - b0[2], c[0], c[1], c[2]);
            MI_TRY(launch_check("k_unpack_box"));
        }
        if (!keep) MI_HIP(hipStreamSynchronize(s));  // the local engine's buffers die at scope exit
    }
    MI_HIP(hipStreamSynchronize(s));  // gin / gout die at scope exit
    return MI_OK;
}

}  // namespace

int edgetaper_async(hipStream_t s, float* bl, float* work, const float* psf, int nx, int ny, int nz, int kx, int ky, int kz,
                    TaperKeep** keep) {
    MI_REQUIRE(bl && work && psf && bl != work, "edgetaper_3d: null or aliased buffers");
    NoPlacementTrial as_they_come;   // (the blur's engines run once per block: fft_native.h)
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
    MI_SPAN_BEGIN(spb, "edgetaper: blur (engine build, enqueue, waits)");
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
    // (the direct engine skips whole 128 x 16 x 2 tiles on the plateau: a tile that the shell only touches is computed in full, so
    // for blocks a few tiles wide most of the volume counts -- 512 columns with a 9-sample taper: half of them)
    double inner_tiles = 1.0;
    {
        const int tile[3] = {128, 16, 2};
        for (int d = 0; d < 3; ++d) {
            const int t0 = (epi.plat_lo[d] + tile[d] - 1) / tile[d], t1 = epi.plat_hi[d] / tile[d], nt = (n[d] + tile[d] - 1) / tile[d];
            inner_tiles *= (double)std::max(0, t1 - t0) / nt;
        }
    }
    const double t_direct = (1.0 - inner_tiles) * nvox * 2.0 * taps / 50e12;
    const double t_fft = native_grid ? nf * 25e-12 + 0.02 : nf * 220.0 / 4e12 + 2.5;
    const bool odd = (kx & 1) && (ky & 1) && (kz & 1);
    bool use_fft = odd && t_fft < t_direct;
    // six face slabs through circular engines instead of the whole replicate-padded block
    const SlabPlan sp = plan_slabs(n, k, epi);
    const double t_slabs = sp.ok ? sp.points * 9e-12 + nvox * shell * 8.0 / 3e12 + 0.005 : 1e30;
    bool use_slabs = odd && sp.ok && t_slabs < t_fft && t_slabs < t_direct;
    if (const char* f = std::getenv("MI_EDGETAPER_ENGINE")) {  // tests / experiments: "fft" | "direct" | "slabs"
        use_fft = odd && f[0] == 'f';
        use_slabs = odd && sp.ok && f[0] == 's';
    }
    MI_SPAN_BEGIN(spr, use_slabs ? "edgetaper route: slabs" : use_fft ? "edgetaper route: whole-block FFT" : "edgetaper route: direct shell");
    if (use_slabs) {
        TaperKeep local_keep;
        TaperKeep* kp = nullptr;
        if (keep) {
            if (!*keep) *keep = new (std::nothrow) TaperKeep;
            if (!*keep) return fail(MI_ERR_NOMEM, "edgetaper_3d: out of host memory");
            kp = *keep;
        }
        DevBuf pn;
        MI_TRY(normalised_psf(s, psf, kx * ky * kz, pn));
        int rc = edgetaper_slabs(s, bl, work, pn.as<float>(), n, k, sp, kp);
        hipError_t e = hipStreamSynchronize(s);  // pn dies here
        if (rc == MI_OK && e != hipSuccess) rc = fail(MI_ERR_HIP, "edgetaper_3d: %s", hipGetErrorString(e));
        MI_TRY(rc);
        (void)local_keep;
    } else if (use_fft) {
        // `keep` (a deconvolution plan): the engine -- the OTF of the normalised PSF on the replicate-padded grid -- outlives
        // the call and serves the next block of the same shape and PSF
        FftEngine local, *fe = &local;
        bool built = false;
        if (keep && !*keep) {
            *keep = new (std::nothrow) TaperKeep;
            if (!*keep) return fail(MI_ERR_NOMEM, "edgetaper_3d: out of host memory");
        }
        if (keep && (*keep)->full) fe = (*keep)->full;
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
            if (keep) (*keep)->full = fe;
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
    MI_SPAN_END(spr);
    MI_SPAN_END(spb);
    MI_SPAN_BEGIN(spw, "edgetaper: blend + final wait");
    const float* t = dtaper.as<float>();
    int x_lo = 0, x_hi = nx;  // plateau of the x taper (host copy of the vectors: `taper`)
    while (x_lo < nx && taper[0][x_lo] != 1.0f) ++x_lo;
    while (x_hi > x_lo && taper[0][x_hi - 1] != 1.0f) --x_hi;
    hipLaunchKernelGGL(k_taper_blend, dim3((unsigned)ny, (unsigned)nz), dim3(256), 0, s, bl, work, t + off[0], t + off[1], t + off[2], nx, ny, nz,
                       x_lo, x_hi);
    MI_TRY(launch_check("k_taper_blend"));
    // host vector / DevBufs die at scope exit: the H2D copy source must outlive the copy
    MI_HIP(hipStreamSynchronize(s));
    MI_SPAN_END(spw);
    return MI_OK;
}

}  // namespace mi

extern "C" int mi_edgetaper3d(int dev, void* stream, float* bl, float* work, const float* psf, int nx, int ny, int nz, int kx,
                              int ky, int kz) {
    MI_TRY(mi::use_device(dev));
    return mi::edgetaper_async(mi::as_stream(stream), bl, work, psf, nx, ny, nz, kx, ky, kz, nullptr);
}
