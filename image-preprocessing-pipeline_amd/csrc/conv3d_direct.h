// Internal interface of the direct convolution engine (conv3d_direct.hip).
#pragma once
#include "mi_internal.h"
#include "mi_lsdeconv.h"

namespace mi {

enum ConvEpi {
    EPI_NONE = 0,        // out = c
    EPI_RATIO = 1,       // out = a ./ max(c, eps)                      (decon.m:62-63)
    EPI_UPDATE = 2,      // out = abs(a .* c)                           (decon.m:76,79)
    EPI_UPDATE_REG = 3,  // out = abs(a .* c .* (1-lambda) + b .* lambda) (decon.m:71,79)
    EPI_TAPER_SHELL = 4  // out = c, tiles on the taper plateau skipped (edgetaper_3d.m:44)
};

struct ConvEpilogue {
    const float* a = nullptr;
    const float* b = nullptr;
    float lambda = 0.0f;
    int plat_lo[3] = {0, 0, 0};  // [lo, hi) per axis (x, y, z) where the taper is exactly 1
    int plat_hi[3] = {0, 0, 0};
};

// offset of the kernel window start relative to the output sample, per boundary rule
int conv_kernel_offset(int k, int boundary);
// builds the x-padded tap table of the direct engine on the device: flip=true turns a convolution
// kernel into the correlation taps the engine consumes (optionally sum-normalised)
int direct_prepare_psf(hipStream_t s, const float* ker, int kx, int ky, int kz, bool normalise, bool flip, DevBuf& kf,
                       int* kxp_out);
// device copy of ker / sum(ker) (n taps)
int normalised_psf(hipStream_t s, const float* ker, int n, DevBuf& out);
// offs (optional) = window start offsets {cx, cy, cz}; default conv_kernel_offset(k, boundary)
// bnd3 (optional) = per-axis boundary rules {x, y, z}; default `boundary` on every axis
int direct_conv_launch(hipStream_t s, const float* img, const float* kf, float* out, int nx, int ny, int nz, int kx, int ky,
                       int kz, int kxp, int boundary, int epi_kind, const ConvEpilogue& epi, const int* offs = nullptr,
                       const int* bnd3 = nullptr);
// sep3d.hip: a rank-1 kernel a (x) b (x) c as ONE pass (8 B/voxel + epilogue operand) instead of three launches of the dense kernel.
// taps[axis] in WINDOW order: out[i] = sum_t in[rule(i - offs[axis] + t)] * taps[axis].w[t], each 1-D result rounded to fp32.
constexpr int kSepMaxTaps = 51;
struct SepTaps {
    float w[kSepMaxTaps];
    int n;
};
bool sep3d_fits(int nx, const int* k, const int* offs);
int sep3d_launch(hipStream_t s, const float* in, float* out, int nx, int ny, int nz, const SepTaps taps[3], const int* offs, const int* bnd3,
                 int epi_kind, const ConvEpilogue& epi);
int gauss3d_async(hipStream_t s, float* vol, float* work, int nx, int ny, int nz, const float* sigma, const int* ksize);
// dst = G(src) in ONE pass when the filter fits the fused kernel (*fused = true); else the two-pass route, which uses dst as
// its intermediate and leaves the result in src (*fused = false)
int gauss3d_to(hipStream_t s, float* src, float* dst, int nx, int ny, int nz, const float* sigma, const int* ksize, bool* fused);
struct TaperKeep;  // the FFT engines of edgetaper_3d's blur (edgetaper.hip)
void taper_keep_free(TaperKeep* k);
// keep != nullptr: the engines of the blur are created into / reused from *keep (see mi_decon_plan); the caller owns it and frees it
// with taper_keep_free
int edgetaper_async(hipStream_t s, float* bl, float* work, const float* psf, int nx, int ny, int nz, int kx, int ky, int kz,
                    TaperKeep** keep = nullptr);

// common.hip
int sumsq_async(hipStream_t s, const float* x, size_t n, double* d_out);

}  // namespace mi
