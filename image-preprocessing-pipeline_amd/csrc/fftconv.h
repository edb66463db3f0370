// Internal interface of the rocFFT convolution engine (fftconv.hip).
#pragma once
#include <rocfft/rocfft.h>

#include "conv3d_direct.h"
#include "fft_native.h"

namespace mi {

// Where the PSF sample j of an axis lands in the length-F circular kernel: index (j - shift) mod F.
//   exact "same" convolution (zero / replicate rules): shift = k - 1 - window_offset
//   deconFFT flavour (decon.m:131-133): centred zero-pad then ifftshift -> shift = floor(F/2) - floor((F-k)/2)
struct AxisPlan {
    int n = 0;      // data extent
    int k = 0;      // PSF extent
    int F = 0;      // transform length
    int o = 0;      // offset of the data inside the padded array (replicate: window offset)
    int shift = 0;  // PSF placement shift
    int boundary = MI_BOUNDARY_ZERO;
};

struct FftEngine {
    AxisPlan ax[3];  // x, y, z
    bool padded = false;  // F != n or o != 0: inputs are staged through `real`
    rocfft_plan fwd = nullptr, inv = nullptr;
    rocfft_execution_info info = nullptr;
    DevBuf work, spec, real, otf, otf_adj;
    size_t n_real = 0, n_spec = 0;  // element counts (floats / complex)
    bool have_adj = false;
    NativeFft* native = nullptr;  // hand-written pipeline, when the shape allows it

    ~FftEngine();
    // bnd/shift per axis (x, y, z): boundary rule and PSF placement shift (see AxisPlan)
    // fixed_psf = false: the PSF will be replaced with set_psf (deconFFT_Wiener), so the OTF keeps its general complex form
    int init(hipStream_t s, const int n[3], const int k[3], const int bnd[3], const int shift[3], const float* psf,
             const float* psf_inv, bool need_adjoint, bool fixed_psf = true);
    // native pipeline only: rebuild the forward OTF from a new PSF of the same extents and placement (device pointer)
    int set_psf(hipStream_t s, const float* psf);
    // c = conv(in, psf or its adjoint), then the epilogue of `epi_kind` into out (shape n)
    int conv(hipStream_t s, const float* in, bool adjoint, float* out, int epi_kind, const ConvEpilogue& epi);
    size_t device_bytes() const {
        return work.bytes + spec.bytes + real.bytes + otf.bytes + otf_adj.bytes + (native ? native->device_bytes() : 0);
    }
};

// writes the half-spectrum OTF of `psf` placed per `ax` (scaled) into `otf` using plan `fwd`
int build_otf(hipStream_t s, rocfft_plan fwd, rocfft_execution_info info, const float* psf, const AxisPlan ax[3], float* real_scratch,
              float* otf, float scale);
int rocfft_global_setup();
// transform lengths for the extents `need` (x, y, z) under the per-axis boundary rules; true: the hand-written pipeline takes them
bool choose_fft_lengths(const int need[3], const int bnd[3], int F[3]);

}  // namespace mi
