// Internal interface of the hand-written FFT convolution pipeline (fft_native.hip).
#pragma once
#include <vector>
#include <algorithm>
#include "conv3d_direct.h"

namespace mi {

struct NativeDims {
    // every axis length is r3 * 2^l2 with r3 in {1, 3, 9}: x (Hx = X/2 complex points), y, z
    int lhx2, r3x;
    int ly2, r3;
    int lz2, r3z;
    int hx, ny, nz;
    int ty, tc, tl;   // rows per x tile, columns per y tile, lines per z tile (A and B tiles each)
    // padded grids: planes z >= z_in_hi of a convolution's input are all zero (never stored, never loaded); of its result only
    // planes [z_out_lo, z_out_hi) and rows < y_out_hi are ever read.  Whole grid when nothing is padded.
    int z_in_hi, z_out_lo, z_out_hi, y_out_hi;
    int dbg;          // timing experiments only: knocks out phases of the z pass (results are then wrong)
    int paired;       // spectra around the z pass in the pair-interleaved layout (k_y_pair, k_z_pair_pipe)
    int zpad;         // paired layout: float4 of padding behind every row (xk, z)
    int xrow;         // x side: complex samples from one row (z, px) to the next (ny + padding)
    int xk0, xkn;     // paired layout: planes xk0 .. xk0 + xkn - 1 of a y / z launch (all of them, or one chunk of the blocked chain)
    int yz0;          // forward y pass: first z plane of the launch (a z chunk of the sharded step; multiple of the planes per work-group)
};

// Padded mode: the caller's volume (extents n) sits at offset o inside the transform grid; the x passes apply the boundary
// rule while loading (zero rule: zeros outside the data; replicate rule: clamped samples inside the window [0, w), zeros beyond)
// and crop while storing, so no padded copy of the volume exists.
struct PadWindow {
    int on = 0;
    int n[3] = {0, 0, 0};
    int o[3] = {0, 0, 0};
    int rep[3] = {0, 0, 0};
    int w[3] = {0, 0, 0};
};

// Subset of the y tiles of the fused x pass: mode 0 all, 1 only the tiles [lo0, lo0+n0) and [lo1, lo1+n1), 2 all the others
struct TileSelect {
    int mode = 0, lo0 = 0, n0 = 0, lo1 = 0, n1 = 0;
    int z0 = 0, nz = 0;  // only the planes [z0, z0 + nz) (nz = 0: all): the z-chunked halo exchange runs the edge tiles chunk by chunk
};

// a device address range backed by physical chunks that were created one by one and mapped in a chosen order (HIP virtual memory)
struct VmmRange {
    void* va = nullptr;
    size_t bytes = 0, chunk = 0;
    bool mapped = false;
    std::vector<hipMemGenericAllocationHandle_t> h;
    int alloc(size_t n, size_t chunk_bytes, int order);
    void release();
    ~VmmRange() { release(); }
};

// While one of these exists on a thread, the contexts that thread creates take their spectrum arrays as they come instead of placing
// them by trial (NativeFft::init): the whole-loop entry points (mi_decon, mi_decon_plan_run, mi_rl_fft) run a handful of iterations per
// context, the trial costs 0.5 s, holds six candidates of the arrays, is serialised per device and trims the pool -- with five
// decwrap workers on blocks of 1024^3 it made 27 blocks take 14.5 s (profiles/r05_decwrap_scale.txt).  Contexts made with
// mi_rl_create (a bench, a slab rank: thousands of iterations on one placement) are placed by trial as before.
struct NoPlacementTrial {
    NoPlacementTrial();
    ~NoPlacementTrial();
    NoPlacementTrial(const NoPlacementTrial&) = delete;
    NoPlacementTrial& operator=(const NoPlacementTrial&) = delete;
};

struct NativeFft {
    NativeDims dims{};
    PadWindow pw{};
    DevBuf S, G, G_adj, tw;  // S: both spectrum arrays, S first
    float2* t_spec = nullptr;  // the second spectrum array T (inside S's allocation, or T2 when the arrays were placed by trial)
    DevBuf T2, S_alt;  // S_alt: a second buffer for S, kept until the caller's volume is known (settle_s)
    DevBuf Gr, Gr_adj, ph;  // real form of the OTF(s) + phase tables (symmetric PSFs), see try_real_otf
    bool real_otf = false;
    bool have_adj = false;  // adjoint = second OTF (G_adj) instead of conj(G)
    const float2* tw_x = nullptr;
    const float2* tw_y = nullptr;
    const float2* tw_z = nullptr;
    size_t n_cplx = 0;
    int n_cu = 256;  // persistent kernels launch one work-group per CU
    std::vector<float> placement_ms;  // forward y pass on each candidate placement of the spectrum arrays (init)
    int placement_kept = -1;
    // x launches that run beside a halo exchange (part 2 of a sharded step): compute units left free for the collective's
    // kernels, tiles handed out by a device counter instead of a fixed stride (mi_rl_set_overlap)
    int overlap_free_cus = 0;
    bool overlap_dynamic = true;
    bool x_dynamic = true;   // every other persistent x launch
    DevBuf ctr;
    bool z_dynamic = true;   // the paired z pass takes its tiles from a counter too
    int ctr_slot = 0;
    int persistent_grid(hipStream_t s, int ntiles, bool overlapped, unsigned* grid, int** ctr_out);
    VmmRange vmm;  // probe builds, MI_FFT_VMM: the spectrum arrays as a range mapped chunk by chunk
    ~NativeFft();

    static bool supported(const int F[3]);
    // smallest supported extent >= n of axis 0 (x), 1 (y), 2 (z); 0 when there is none
    static int good_size(int n, int axis);
    // buffers, tile sizes, twiddles; the OTF(s) are then built with build_otf
    int init(hipStream_t s, const int F[3], bool explicit_adjoint);
    // placed: the kernel on the circular grid (real, shape F; may alias scratch()); G (or G_adj) <- scale * FFT(placed)
    int build_otf(hipStream_t s, const float* placed, bool adjoint_slot, float scale);
    // the same transform into a caller buffer of otf_items() float4: spectrum of any real F volume, scaled (deconFFT_Wiener)
    int spectrum(hipStream_t s, const float* vol, float4* dst, float scale);
    size_t otf_items() const { return G.bytes / sizeof(float4); }
    float4* otf() { return G.as<float4>(); }
    // after build_otf: switch to the real OTF form when the PSF allows it (delta: centre offset from the grid origin)
    int try_real_otf(hipStream_t s, const int delta[3]);
    bool z_pipelined() const;
    float* scratch() { return reinterpret_cast<float*>(t_spec); }  // F floats, free between convolutions
    // after the OTFs are built: volumes handed to conv / iterate have extents n (x, y, z) and are padded on the fly
    void set_window(const int n[3], const int o[3], const int rep[3], const int k[3]);
    bool can_fuse() const;  // consecutive convolutions may share their x passes (every padded axis follows the zero rule)
    int conv(hipStream_t s, const float* in, bool conj_otf, float* out, int epi_kind, const ConvEpilogue& epi);
    // `conj_otf` selects the adjoint: conj(OTF), or the explicit adjoint OTF when one was given
    // n fused RL iterations on bl in place (lambda = 0, no regularisation step in between)
    int iterate(hipStream_t s, float* bl, int n_iters);
    int time_pass(hipStream_t s, int which, const float* bl, int reps, float* avg_ms);
    size_t spectrum_bytes() const { return spec_bytes; }   // one of the two spectrum arrays
    void settle_before_update();
    int settle_decide(hipStream_t s);
    // the spare buffer for S goes back to the driver (every consumer of S other than iterate() calls this first: the sharded steps of
    // the slab driver, single convolutions -- a C4-shaped rank carried 9.7 GB of it for the whole run)
    int release_spare();
    int alt_phase = 0;                 // 0 / 1: the next timed update launch writes the first / second S buffer; 2: decide; 3: settled
    hipEvent_t alt_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int time_between(hipStream_t s, int which, const float2* src, float2* dst, float* bl, int reps, float* avg_ms);
    size_t spec_bytes = 0;
    // rows [y0, y0 + rows) of S (the x-transformed input of the next convolution): dir 0 pack into buf, 1 unpack from buf, 2 zero
    // (z0, nzc: only the planes [z0, z0 + nzc), which keep their place in the packed buffer; nzc = 0: all)
    int spectrum_rows(hipStream_t s, int y0, int rows, float2* buf, int dir, int z0 = 0, int nzc = 0);
    // forward y pass of the planes [z0, z0 + nzc) only (both layouts); z0 and nzc multiples of y_z_granule()
    int y_forward_planes(hipStream_t s, int z0, int nzc);
    int y_z_granule() const { return dims.paired ? std::max(1, dims.tc / 2) : 1; }
    size_t spectrum_row_floats() const { return (size_t)2 * dims.nz * dims.hx; }
    int x_forward(hipStream_t s, const float* in);
    int middle(hipStream_t s, bool conj_otf);
    int y_pass(hipStream_t s, bool inverse, bool paired, const float2* src = nullptr, float2* dst = nullptr, int xk0 = 0, int xkn = -1);
    int z_conv(hipStream_t s, bool conj_otf, const float2* src = nullptr, float2* dst = nullptr, int xk0 = 0, int xkn = -1);
    int x_inverse(hipStream_t s, float* out, int epi_kind, const ConvEpilogue& epi, bool fuse_forward, const TileSelect* part = nullptr);
    bool pipe_ok() const;  // the fused x pass can run as the persistent pipelined kernel
    bool splits() const;   // ... and a subset of its tiles (unpadded grids)
    TileSelect edge_tiles(int mode, int a0, int a1, int b0, int b1) const;
    size_t device_bytes() const { return S.bytes + T2.bytes + S_alt.bytes + G.bytes + G_adj.bytes + Gr.bytes + Gr_adj.bytes + ph.bytes + tw.bytes; }
};

}  // namespace mi
