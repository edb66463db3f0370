// Internal interface of the hand-written FFT convolution pipeline (fft_native.hip).
#pragma once
#include "conv3d_direct.h"

namespace mi {

struct NativeDims {
    int lhx, lz;      // log2 of Hx = X/2 and of Z
    int ly2, r3;      // Y = r3 * 2^ly2, r3 in {1, 3, 9}
    int ny, nz;
    int ty, tc, tl;   // rows per x tile, columns per y tile, lines per z tile (A and B tiles each)
    int dbg;          // timing experiments only: knocks out phases of the z pass (results are then wrong)
};

struct NativeFft {
    NativeDims dims{};
    DevBuf S, T, G, tw;
    const float2* tw_x = nullptr;
    const float2* tw_y = nullptr;
    const float2* tw_z = nullptr;
    size_t n_cplx = 0;

    static bool supported(const int F[3]);
    // smallest supported y extent >= n (2^a, 3*2^a or 9*2^a)
    static int good_size_y(int n);
    // otf_half_spectrum: R2C layout [Z][Y][X/2+1]; it is multiplied by `scale` while being repacked
    int init(hipStream_t s, const int F[3], const float2* otf_half_spectrum, float scale);
    int conv(hipStream_t s, const float* in, bool conj_otf, float* out, int epi_kind, const ConvEpilogue& epi);
    // n fused RL iterations on bl in place (lambda = 0, no regularisation step in between)
    int iterate(hipStream_t s, float* bl, int n_iters);
    int time_pass(hipStream_t s, int which, const float* bl, int reps, float* avg_ms);
    int x_forward(hipStream_t s, const float* in);
    int middle(hipStream_t s, bool conj_otf);
    int x_inverse(hipStream_t s, float* out, int epi_kind, const ConvEpilogue& epi, bool fuse_forward);
    size_t device_bytes() const { return S.bytes + T.bytes + G.bytes + tw.bytes; }
};

}  // namespace mi
