/* mi_lsdeconv.h -- C ABI of the Richardson-Lucy deconvolution hot path (LsDeconvolveMultiGPU).
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference
 * repository).  Conventions: include/mi_common.h.  The reference-side binding a maintainer would
 * add is shown in INTEGRATION.md.
 */
#ifndef MI_LSDECONV_H
#define MI_LSDECONV_H

#include "mi_common.h"

#ifdef __cplusplus
extern "C" {
#endif

/* boundary rule of a "same"-size 3-D convolution */
typedef enum {
    MI_BOUNDARY_ZERO = 0,      /* MATLAB convn(a,h,'same'): decon.m:61,64,70 */
    MI_BOUNDARY_REPLICATE = 1, /* conv3d_gpu clamp: conv3d_gpu.cu:82-91 */
    MI_BOUNDARY_CIRCULAR = 2   /* fftn/ifftn wrap on the given shape: decon.m:162-172 */
} mi_boundary;

/* convolution engine */
typedef enum {
    MI_ENGINE_AUTO = 0,   /* cost model on PSF taps / volume size (mi_engine_select) */
    MI_ENGINE_DIRECT = 1, /* LDS-tiled direct convolution (fp32 FMA); three 1-D passes for rank-1 PSFs (mi_rl_separable) */
    MI_ENGINE_FFT = 2     /* FFT convolution on the grid the boundary rule requires: hand-written pipeline (fft_native.hip), rocFFT
                             for extents it does not take */
    /* (a matrix-core direct engine was prototyped and measured -- banded-Toeplitz v_mfma_f32_16x16x4_f32, bit-identical results,
       70 ms against 58 ms of the fp32 FMA engine on BASELINE config 2: profiles/mfma_toeplitz_probe.hip,
       profiles/r02_mfma_toeplitz_probe.txt -- and is not part of the library) */
} mi_engine;

/* ---- single kernels -------------------------------------------------------------------------- */

/* out = conv3d_gpu(img, kernel)   [LsDeconvolveMultiGPU/conv3d_gpu.cu:101-148, kernel :68-99]
 * img/out: nx*ny*nz floats, ker: kx*ky*kz floats (device), all extents >= 1; out must not alias img. */
int mi_conv3d_replicate(int dev, void* stream, const float* img, const float* ker, float* out,
                        int nx, int ny, int nz, int kx, int ky, int kz);

/* out = convn(img, ker, 'same') with the chosen boundary rule and engine   [decon.m:61,64,70]
 * ker extents must be odd for MI_ENGINE_FFT (always true for LsMakePSF PSFs: LsMakePSF.m:32-38). */
int mi_conv3d(int dev, void* stream, const float* img, const float* ker, float* out,
              int nx, int ny, int nz, int kx, int ky, int kz, int boundary, int engine);

/* x = gauss3d_gpu(x, sigma[, ksize])   [LsDeconvolveMultiGPU/gauss3d_gpu.cu:209-311, :81-204]
 * Overwrites vol (destructive, like the reference); work = nx*ny*nz floats of scratch.
 * sigma [host] = {sx, sy, sz}; ksize [host] = {kx, ky, kz} or NULL for 2*ceil(3*sigma)+1 (:244-261);
 * ksize <= 51 (MAX_KERNEL_SIZE, :77).  Axes are filtered X, Y, Z with replicate boundary. */
int mi_gauss3d_inplace(int dev, void* stream, float* vol, float* work, int nx, int ny, int nz,
                       const float* sigma, const int* ksize);

/* bl = edgetaper_3d(bl, psf)   [LsDeconvolveMultiGPU/edgetaper_3d.m:13-44, make_taper.m:13-35]
 * In place. work = nx*ny*nz floats.  psf need not be normalised (it is divided by its sum, :14).
 * Only the border shell where the taper mask is < 1 is convolved. */
int mi_edgetaper3d(int dev, void* stream, float* bl, float* work, const float* psf,
                   int nx, int ny, int nz, int kx, int ky, int kz);

/* otf = fftn(ifftshift(zero-pad-centre(psf)))   [supplements/otf_gpu.cu:36-67,125-144; decon.m:131-133]
 * Written as the R2C half spectrum: complex interleaved float2 [fz][fy][fx/2+1], scaled by `scale`
 * (pass 1.0f for the plain OTF). work: mi_otf_workspace_bytes(). */
int mi_otf(int dev, void* stream, const float* psf, int kx, int ky, int kz, float* otf, int fx, int fy, int fz,
           float scale);

/* dst = single(src) * scale : im2single of uint16 blocks (LsDeconv.m:860,873; scale = 1/65535) */
int mi_u16_to_f32(int dev, void* stream, const uint16_t* src, float* dst, size_t n, float scale);

/* load_block (LsDeconv.m:817-904) on the device: dst [nz][ny][nx] (the padded block, float32) <- the box src [sz][sy][sx] that was
 * read from the volume where the padded block overlaps it, placed at offset (bx, by, bz); integer inputs are converted like
 * im2single (value / 255 or / 65535, a float32 division); where the block reaches beyond the volume it is extended like
 * padarray(..., 'symmetric') (edge-inclusive mirror of the box).  dtype: 1 = uint8, 2 = uint16, 4 = float32.  Only enqueues. */
int mi_load_block(int dev, void* stream, const void* src, int dtype, int sx, int sy, int sz,
                  float* dst, int nx, int ny, int nz, int bx, int by, int bz);

/* dst = max(src - dark, 0)   [LsDeconv.m:924-927], in place allowed */
int mi_subtract_dark(int dev, void* stream, const float* src, float* dst, size_t n, float dark);

/* *norm2 [host] = sqrt(sum(x.^2)) accumulated in double (norm(bl(:)), decon.m:47,109). Synchronises. */
int mi_norm2(int dev, void* stream, const float* x, size_t n, double* norm2);

/* [lb, ub] = deconvolved_stats(bl, clipval) = prctile(bl, [100-clipval clipval], "all")   [LsDeconv.m:1300-1307, called from
 * process_block :945].  Exact percentiles of the whole device volume (MATLAB's definition: the i-th sorted sample is the
 * 100(i-0.5)/n percentile, linear in between, clamped to min / max; NaN ignored), found by a three-level radix select on the
 * device: no sorted copy, no sub-sampling, only histograms cross PCIe.  n_pct = 1 or 2; out = n_pct host floats.  Synchronises. */
int mi_prctile(int dev, void* stream, const float* x, size_t n, const double* pct, int n_pct, float* out);

/* The rescale / round / clamp / convert that load_slab_lz4 applies to every float brick while it assembles the output slab
 * [load_slab_lz4.cpp:134-157; called from postprocess_save, LsDeconv.m:1091-1093]:
 *   val = (dmin > 0) ? (val - dmin) * (scal*ampl/(dmax-dmin)) : val * (scal*ampl/dmax);  val -= ampl;
 *   val = round-half-away-from-zero(val);  val = clamp(val, 0, scal);  dst = (uint8|uint16) val
 * in float arithmetic in that order.  out_bits = 8 or 16; dst is a device buffer of n elements of that type. */
int mi_rescale_block(int dev, void* stream, const float* src, void* dst, size_t n, int out_bits, float scal, float ampl,
                     float dmin, float dmax);

/* zero-pad-centre / crop   [decon.m:323-374: pre = floor(missing/2), post = ceil] */
int mi_pad_center(int dev, void* stream, const float* src, int nx, int ny, int nz, float* dst, int fx, int fy, int fz);
int mi_crop_center(int dev, void* stream, const float* src, int fx, int fy, int fz, float* dst, int nx, int ny, int nz);

/* ---- RL context: the two half-steps of one iteration (what the slab driver calls between halo
 *      exchanges) ------------------------------------------------------------------------------ */

typedef struct mi_rl_ctx mi_rl_ctx;

/* Prepares PSF-derived constants for arrays of shape (nz, ny, nx): flipped/padded PSF for the direct
 * engines, OTF + rocFFT plans + padded work buffers for the FFT engine (all owned by the context).
 * psf_inv may be NULL (= psf flipped in all axes, LsDeconv.m:163, convolved like decon.m:64 does: with convn 'same', so for even
 * extents under a non-circular rule it is NOT the transpose of the forward operator; the circular rule uses conj(otf),
 * decon.m:168).  Synchronises. */
int mi_rl_create(int dev, void* stream, int nx, int ny, int nz, const float* psf, const float* psf_inv,
                 int kx, int ky, int kz, int boundary, int engine, mi_rl_ctx** ctx);
/* Same with one boundary rule per axis {x, y, z} and an explicit PSF placement per axis: sample j of the
 * PSF acts at offset (j - shift) (forward) / (shift - j) (adjoint); shift_xyz == NULL or an entry < 0 selects
 * the default of that axis' rule.  The slab driver uses it to run an axis that is sharded across GPUs as
 * "circular on the local extent" (valid away from the halos) with the placement of the GLOBAL volume. */
int mi_rl_create_ex(int dev, void* stream, int nx, int ny, int nz, const float* psf, const float* psf_inv,
                    int kx, int ky, int kz, const int* boundary_xyz, const int* shift_xyz, int engine,
                    mi_rl_ctx** ctx);
/* > 0 when the context applies the PSF as three 1-D convolutions: the direct engine does so for PSFs (and explicit adjoint
 * kernels) that are an outer product of three lines to fp32 rounding (every sample within 4e-7 of ITS value of the product of
 * its line samples), e.g. the Gaussian PSF of BASELINE config 1 -- kx + ky + kz instead of kx * ky * kz taps per voxel [no
 * reference counterpart: conv3d_gpu.cu:68-99 always runs the dense loop].  2: all three in ONE pass over the volume (sep3d.hip:
 * 8 B/voxel; rows of whole float4, <= 51 taps per axis, a ring of kz xy-filtered planes in LDS); 1: three launches of the
 * dense kernel with 1-D tap tables (24 B/voxel; MI_NO_SEP_SINGLE=1 forces it).  MI_NO_SEPARABLE=1 disables the test. */
int mi_rl_separable(const mi_rl_ctx* ctx);
/* 1 when the FFT engine of the context keeps its spectra around the z pass in the pair-interleaved layout (every block of 8 lines
 * followed by its 8 mirror-partner lines; two 64-KB z tiles per CU): z extents 2^a (64..1024), 3 * 2^a (192..768) or 9 * 2^a (576, 1152) on the hand-written
 * pipeline.  MI_FFT_NO_PAIR=1 keeps the plain layout [no reference counterpart: cuFFT owns its layouts]. */
int mi_rl_pair_layout(const mi_rl_ctx* ctx);
int mi_rl_destroy(mi_rl_ctx* ctx);
/* engine actually chosen (mi_engine) and device bytes held by the context */
int mi_rl_engine(const mi_rl_ctx* ctx);
size_t mi_rl_device_bytes(const mi_rl_ctx* ctx);

/* ratio = bl ./ max(conv(bl, psf), eps('single'))          [decon.m:61-63 / :162-167] */
int mi_rl_forward_ratio(mi_rl_ctx* ctx, void* stream, const float* bl, float* ratio);
/* bl = abs(bl .* conv(ratio, psf_inv))                      [decon.m:64,76,79 / :169-186]
 * or, with lambda > 0 and reg != NULL, abs(bl.*conv.*(1-lambda) + reg.*lambda)   [decon.m:69-71] */
int mi_rl_adjoint_update(mi_rl_ctx* ctx, void* stream, const float* ratio, float* bl, float lambda, const float* reg);
/* n_iters plain RL iterations (lambda = 0, no regularisation step) on bl in place.  Engines that can fuse across
 * the two convolutions do (native FFT pipeline: 8 volume passes per iteration, the ratio never reaches HBM, `ratio`
 * may then be NULL); the others run forward_ratio / adjoint_update n_iters times using `ratio` as scratch. */
int mi_rl_iterate(mi_rl_ctx* ctx, void* stream, float* bl, float* ratio, int n_iters);
/* Sharded fused iteration (the multi-GPU slab driver; no reference counterpart -- the reference's blocks never talk to each
 * other, LsDeconv.m:647-654): mi_rl_iterate cut at the two places of an iteration where the y halo rows of a convolution's input
 * must be refreshed from the neighbouring slabs.  The input of every convolution lives in the pipeline's x-transformed buffer
 * ("spectrum rows": the x-FFT of the rows, still indexed by y), so the halo rows are exchanged THERE and the ratio still never
 * reaches HBM:
 *     mi_rl_sharded_begin(ctx, s, bl)         S <- x-forward(bl)                                     [then exchange rows of S]
 *     mi_rl_sharded_ratio(ctx, s, bl)         S <- x-forward(bl ./ max(conv(S), eps))                [then exchange rows of S]
 *     mi_rl_sharded_update(ctx, s, bl, more)  bl <- |bl .* conv_adj(S)|; more != 0: S <- x-forward(bl)  [then exchange rows of S]
 * The halo rows of `bl` itself are never read again (each row's x transform is independent) and hold meaningless values.
 * mi_rl_spectrum_rows packs (dir 0) / unpacks (dir 1) / zero-fills (dir 2) rows [y0, y0+rows) of S into / from a contiguous
 * device buffer of rows * mi_rl_spectrum_row_floats(ctx) floats.  mi_rl_fuses: 0 = the context does not run the fused native
 * pipeline (the calls return MI_ERR_UNSUPPORTED and the driver uses forward_ratio / adjoint_update on real halos), 1 = fused,
 * 2 = fused and the x pass can be split so that the exchange overlaps with it: part 0 = the whole step; part 1 = the y/z passes
 * and only the x tiles that hold rows of edge_rows = {a0, a1, b0, b1} ([a0,a1) and [b0,b1): the rows about to be sent); part 2 =
 * the remaining x tiles (launch it after the sends have been issued). */
int mi_rl_fuses(mi_rl_ctx* ctx);
/* 1 when the context keeps its OTF in the real form (PSF mirror-symmetric about its centre sample: 2 floats per spectrum pair
 * plus per-axis phase tables instead of 4 floats; the z pass then moves 10 instead of 12 bytes per voxel). */
int mi_rl_otf_is_real(mi_rl_ctx* ctx);
int mi_rl_sharded_begin(mi_rl_ctx* ctx, void* stream, const float* bl);
int mi_rl_sharded_ratio(mi_rl_ctx* ctx, void* stream, const float* bl, int part, const int* edge_rows);
int mi_rl_sharded_update(mi_rl_ctx* ctx, void* stream, float* bl, int more, int part, const int* edge_rows);
int mi_rl_spectrum_rows(mi_rl_ctx* ctx, void* stream, int y0, int rows, float* buf, int dir);
size_t mi_rl_spectrum_row_floats(mi_rl_ctx* ctx);
/* The same steps cut along z, for a halo exchange that travels in z chunks (slab.py: zchunks > 1).  The x transform of a row
 * depends on nothing else and the y transform of a column (z, px) only on the rows of its own plane, so a rank can (a) send the
 * edge rows of the planes [z0, z1) as soon as the x tiles of those planes have run and (b) start the next step's y-forward pass
 * on a chunk of planes as soon as THAT chunk's halo rows have landed, while later chunks still travel: the window of an exchange
 * grows from the rest of the x pass to almost the whole x pass plus the y pass.  One half-step (update = 0: ratio, 1: update):
 *     stage 0, per chunk   y-forward of the planes [z0, z1)             (their halo rows must have been unpacked)
 *     stage 1              z pass (OTF or its conjugate) + y-inverse     (all planes)
 *     stage 2, per chunk   x pass, the tiles that hold rows of edge_rows, planes [z0, z1)  [then pack + send that chunk]
 *     stage 3              x pass, all other tiles
 * Results are identical to part 0 / 1 / 2 of mi_rl_sharded_ratio / _update.  Chunk boundaries must be multiples of
 * mi_rl_z_granule(ctx) (0: the context cannot run chunked).  mi_rl_spectrum_rows_z is mi_rl_spectrum_rows restricted to the planes
 * [z0, z1); the chunk keeps its place in the packed buffer ([z * nx/2 + px][rows] complex), i.e. it is the contiguous range that
 * starts z0 * rows * (row_floats / nz) floats into it. */
int mi_rl_sharded_stage(mi_rl_ctx* ctx, void* stream, float* bl, int update, int stage, int z0, int z1, const int* edge_rows);
int mi_rl_spectrum_rows_z(mi_rl_ctx* ctx, void* stream, int y0, int rows, int z0, int z1, float* buf, int dir);
int mi_rl_z_granule(mi_rl_ctx* ctx);
/* How the "part 2" launches of a split step share the GPU with the halo exchange that is in flight beside them (the reference
 * has no counterpart: its blocks never exchange anything, LsDeconv.m:643-654).  The x pass is a persistent kernel of one
 * 16-wave work-group per compute unit; a collective's kernels need compute units too.  free_cus: work-groups NOT launched
 * (grid = CUs - free_cus), so that many units stay available to the collective; dynamic_tiles != 0: tiles are handed out by a
 * device counter (one atomicAdd per tile) instead of a fixed stride, so a work-group that starts late -- its unit was held by
 * the collective -- takes fewer tiles, or none, instead of running a fixed share as the tail of the pass.  Default: 0, 1 (the
 * single-GPU launch geometry; measured on one GPU with a stand-in collective: profiles/r03_overlap_probe.txt).  A copy-engine
 * transport (slab.py, transport="peer") occupies no compute unit at all. */
int mi_rl_set_overlap(mi_rl_ctx* ctx, int free_cus, int dynamic_tiles);
/* Measurement hook for the above on ONE GPU: launches `busy_wgs` stand-in work-groups (256 threads, holding their compute units
 * for busy_us microseconds) on a second stream, then part 2 of a ratio step (the tiles outside edge_rows) on `stream` with the
 * current overlap settings; out_ms[0] = HIP-event time of the x launch, out_ms[1] = from issuing the stand-in until both have
 * finished; averages over `reps` (plus one warm-up).  busy_wgs = 0: the x launch alone.  S / T keep whatever the last step
 * left in them.  Synchronises. */
int mi_rl_overlap_probe(mi_rl_ctx* ctx, void* stream, float* bl, const int* edge_rows, int busy_wgs, float busy_us, int reps,
                        float* out_ms);
/* Measurement hook: average duration in ms of `reps` back-to-back launches of ONE pass of the native FFT pipeline,
 * taken with HIP events on `stream` (which: 0 x-forward, 1 y-forward, 2 z-forward*OTF*z-inverse, 3 y-inverse,
 * 4 fused x-inverse+ratio+x-forward, 5 fused x-inverse+update+x-forward -- this one OVERWRITES bl with values that mean
 * nothing).  MI_ERR_UNSUPPORTED for other engines.  Synchronises. */
int mi_rl_time_pass(mi_rl_ctx* ctx, void* stream, int which, const float* bl, int reps, float* avg_ms);
/* Measurement hook: one pass of the native FFT pipeline between two caller-given spectrum buffers (device buffers of
 * mi_rl_fft_spectrum_bytes(ctx) bytes each, contents arbitrary) instead of the context's own arrays -- which 0: the forward
 * y pass reading `src`, writing `dst`; which 1: the update launch of the fused x pass reading `src` (in the role of T),
 * writing `dst` (in the role of S) and reading / writing the volume `bl` (overwritten with values that mean nothing);
 * which 2: the forward x pass reading the volume `bl`, writing `dst` (`src` unused); which 3: the z pass reading `src`, writing
 * `dst`; which 4: the z pass on the context's own arrays with `src` in the place of the real OTF (`dst` unused).
 * Average ms of `reps` launches (profiles/spectrum_halves_probe.py: a buffer's memory region decides how fast it is read
 * and how fast it is written, independently of its partner).  Synchronises. */
size_t mi_rl_fft_spectrum_bytes(mi_rl_ctx* ctx);
int mi_rl_time_between(mi_rl_ctx* ctx, void* stream, int which, const void* src, void* dst, float* bl, int reps, float* avg_ms);
/* Measurement hook: how the spectrum arrays of the native FFT pipeline were placed when the context was created (candidates
 * allocated side by side, "4 y passes + update launch" timed on each, the fastest kept: csrc/fft_native.hip, NativeFft::init).
 * Writes up to `cap` candidate costs in ms to cost_ms -- one per ordered pair (S, T) of the buffers tried, S slowest --, their
 * number to *n (0: a plain allocation) and the index of the kept pair to *kept.  No reference counterpart (the reference allocates inside MATLAB's gpuArray). */
int mi_rl_fft_placement(mi_rl_ctx* ctx, float* cost_ms, int cap, int* n, int* kept);
/* reg = convn(bl, R, 'same'), R = ones(3,3,3)/26 with centre 0   [decon.m:42,70] */
int mi_rl_reg_term(int dev, void* stream, const float* bl, float* reg, int nx, int ny, int nz);

/* ---- whole loops ------------------------------------------------------------------------------ */

typedef struct {
    int niter;               /* decon.m: niter */
    float lambda;            /* Tikhonov weight (decon.m:41,69) */
    float stop_criterion;    /* % change of ||bl||_2, 0 disables (decon.m:108-118) */
    int regularize_interval; /* decon.m:54-55 */
    int engine;              /* mi_engine */
    int skip_edgetaper;      /* 0: like decon.m:50/143; 1: caller tapered already (bench times the loop only) */
    int gauss_taps;          /* 0: gauss3d_gpu(bl,0.5) default 5 taps (GPU path); 3: imgaussfilt3 CPU flavour */
    int psf_grid[3];         /* deconFFT only, [x y z]; 0 = the FFT shape itself (the reference).  ifftshift(zero-pad-centre(psf))
                              * (decon.m:131-133, otf_gpu.cu:36-67,121-123) puts the centre sample of an odd PSF at index 0 of a grid
                              * of odd extent and at index -1 of a grid of even extent: where the PSF lands depends on the parity of
                              * fft_shape, and the result with it (the ratio bl ./ conv(bl) is taken one sample off).  A caller that
                              * runs a block on a larger grid than the reference's next_fast_len one (decwrap.py: an extent the
                              * hand-written transform takes) names the reference's extent here; the PSF is then placed where THAT
                              * grid would have put it (shift = g/2 - (g-k)/2 per axis) and the result differs from the
                              * reference grid's by the effect of the wider zero margin only.  Ignored by deconSpatial and
                              * deconFFT_Wiener. */
} mi_rl_options;

/* bl = deconSpatial(bl, psf, psf_inv, ...)   [decon.m:26-124]  in place; iters_done [host] may be NULL.
 * Allocates its scratch (2-3 volumes) with hipMalloc and frees it before returning.  Synchronises. */
int mi_rl_spatial(int dev, void* stream, float* bl, const float* psf, const float* psf_inv,
                  int nx, int ny, int nz, int kx, int ky, int kz, const mi_rl_options* opt, int* iters_done);

/* bl = deconFFT(bl, psf, fft_shape, ...)   [decon.m:127-204]  in place; (fx,fy,fz) >= (nx,ny,nz).
 * Synchronises. */
int mi_rl_fft(int dev, void* stream, float* bl, const float* psf, int nx, int ny, int nz, int kx, int ky, int kz,
              int fx, int fy, int fz, const mi_rl_options* opt, int* iters_done);

/* bl = deconFFT_Wiener(bl, psf, fft_shape, ...)   [decon.m:206-321]  RL with a Wiener re-estimate of the PSF after every
 * iteration but the last: otf_new = F{Y} conj(F{X}) / max(|F{X}|^2, eps), new psf = the centre box of real(ifftn(otf_new))
 * clamped at 0 and renormalised (decon.m:281-301).  bl in place; psf [kz][ky][kx] is READ AND OVERWRITTEN with the last
 * refined PSF (the reference keeps it local).  The quirks of the .m are kept: Gaussian smoothing whenever
 * regularize_interval > 0 && i %% interval == 0 (i > 1), Tikhonov blend only with lambda > 0 && i < niter, stop test without
 * the i > 1 guard.  fft_shape extents must satisfy mi_fft_good_size(f, axis) == f (the spectra live in the hand-written
 * pipeline's layout), else MI_ERR_UNSUPPORTED.  opt->engine is ignored.  Synchronises. */
int mi_rl_fft_wiener(int dev, void* stream, float* bl, float* psf, int nx, int ny, int nz, int kx, int ky, int kz,
                     int fx, int fy, int fz, const mi_rl_options* opt, int* iters_done);

/* bl = decon(bl, psf, niter, lambda, stop_criterion, regularize_interval, device_id, use_fft, fft_shape,
 *            adaptive_psf)   [decon.m:1-23]; adaptive_psf != 0 with use_fft runs mi_rl_fft_wiener on a private copy of psf. */
int mi_decon(int dev, void* stream, float* bl, const float* psf, const float* psf_inv,
             int nx, int ny, int nz, int kx, int ky, int kz, const mi_rl_options* opt,
             int use_fft, const int* fft_shape_xyz, int adaptive_psf, int* iters_done);

/* bl = filter_subband_3d_z(bl, sigma, levels, "db9")   [filter_subband_3d_z.m:1-123; called by process_block, LsDeconv.m:934-936]
 * log1p, db9 wavelet decomposition of every XZ slice ('sym' extension; odd extents zero-padded to even and cropped again),
 * Gaussian notch along z on the horizontal-detail sub-bands (g(k) = 1 - exp(-k^2 / (2 (sigma / n)^2)), n = coefficients along z),
 * reconstruction, expm1.  In place on bl [nz][ny][nx]; levels = 0 picks wmaxlev (mi_destripe_max_levels); with zero levels
 * (min(nx, nz) < 34) the block is returned unchanged.  Work memory (about 2.5 volumes) comes from the library's pool.
 * Synchronises. */
int mi_destripe_z(int dev, void* stream, float* bl, int nx, int ny, int nz, float sigma, int levels);
/* wmaxlev([nx nz] rounded up to even, 'db9') = fix(log2(min / 17)) */
int mi_destripe_max_levels(int nx, int nz);

/* Deconvolution plan: the blocks of a volume share shape and PSF (the parfeval workers of LsDeconv.m:620-668 call decon once
 * per block), so a worker keeps what mi_decon would rebuild every time -- the RL context (OTF, twiddles, scratch) and the FFT
 * engine of edgetaper_3d's blur -- in a plan.  mi_decon_plan_run has the semantics and arguments of mi_decon; it compares
 * shape, options and the PSF values with what the kept objects were built from and rebuilds them when they differ.  Results
 * are bit-identical to mi_decon.  One plan per worker thread (not re-entrant); destroy it on the thread's device. */
typedef struct mi_decon_plan mi_decon_plan;
int mi_decon_plan_create(int dev, mi_decon_plan** out);
int mi_decon_plan_run(mi_decon_plan* plan, void* stream, float* bl, const float* psf, const float* psf_inv,
                      int nx, int ny, int nz, int kx, int ky, int kz, const mi_rl_options* opt,
                      int use_fft, const int* fft_shape_xyz, int adaptive_psf, int* iters_done);
int mi_decon_plan_destroy(mi_decon_plan* plan);

/* cost model used by MI_ENGINE_AUTO: returns MI_ENGINE_DIRECT or MI_ENGINE_FFT */
int mi_engine_select(int nx, int ny, int nz, int kx, int ky, int kz, int boundary);

/* smallest extent >= n that the hand-written FFT pipeline handles natively on `axis` (0 = x, 1 = y, 2 = z):
 * 2^a, 3 * 2^a or 9 * 2^a (x: twice such a number; z: at most 2304; y also 5 * 2^a, a in 5..8).  Returns 0 when there is none.  Other (7-smooth)
 * shapes run through rocFFT; padded (zero / replicate boundary) convolutions pick such extents themselves. */
int mi_fft_good_size(int n, int axis);
/* next 7-smooth length >= n   [LsDeconv.m:405-419] */
int mi_next_fast_len(int n);

/* ---- slab halo helpers (multi-GPU sharding along Y) ------------------------------------------- */
/* copy rows [y0, y0+rows) of every z-plane of a (nz, ny, nx) volume into/out of a packed
 * (nz, rows, nx) buffer -- the send/recv staging of the RCCL halo exchange */
int mi_pack_rows(int dev, void* stream, const float* vol, int nx, int ny, int nz, int y0, int rows, float* packed);
int mi_unpack_rows(int dev, void* stream, const float* packed, int nx, int ny, int nz, int y0, int rows, float* vol);

/* ---- copy-engine transport of the halo exchange (one process per GPU; no reference counterpart: LsDeconv.m:643-654 farms
 * independent blocks out) ---------------------------------------------------------------------------------------------------
 * A LINK is a rank's side of the ring of slabs: a device allocation with its receive slots (2 directions x 2 alternating sets)
 * and a page of flag words in fine-grained device memory, both exported through HIP IPC memory handles (opaque
 * MI_IPC_HANDLE_BYTES-byte blobs [host] that travel once over any host channel), a copy stream, and the mapped memory of its
 * (at most two) neighbours.  A sender copies its packed rows into the neighbour's slot with hipMemcpyPeerAsync -- executed by the
 * SDMA engines, which take no compute unit away from the persistent x pass it overlaps with, unlike the kernels of a grouped
 * ncclSend/ncclRecv -- and then writes the sequence number (exchange - 1) * chunks + chunk + 1 into the neighbour's arrival word;
 * the receiver's launch stream waits in a one-lane kernel until that word has reached the number it needs.  A receiver
 * acknowledges exchange n when exchange n + 1 begins (its unpack kernels lie before that point of its launch stream) by writing
 * n into the sender's acknowledgement word, which the sender's copy stream waits for before it overwrites the set at exchange
 * n + 2.  No interprocess events, no host-side hand-shake: a wait names a value, so an early, late or repeated wait is the same
 * comparison (csrc/peer.hip has the reasons).  Every wait gives up after MI_PEER_TIMEOUT_S seconds (default 120) and counts
 * itself in the link's status word.
 * Directed edges of a rank: d = 0 "up" (its last interior rows -> slot 0 of the next rank), d = 1 "down" (its first interior rows
 * -> slot 1 of the previous rank).  Exchanges are numbered n = 1, 2, ... by the caller, the same on every rank. */
#define MI_IPC_HANDLE_BYTES 64
typedef struct mi_peer_link mi_peer_link;
/* slot_bytes: packed halo rows of one direction.  payload_handle, flag_handle [host, MI_IPC_HANDLE_BYTES each]: what the
 * neighbours need to connect. */
int mi_peer_link_create(int dev, size_t slot_bytes, mi_peer_link** link, unsigned char* payload_handle, unsigned char* flag_handle);
/* the rank that receives edge d (and fills slot 1 - d): its handles, and the ordinal under which THIS process sees its device
 * (-1: not visible under an ordinal -- the runtime resolves the mapped pointer).  An edge without a neighbour is not connected. */
int mi_peer_link_connect(mi_peer_link* link, int d, const unsigned char* payload_handle, const unsigned char* flag_handle, int peer_dev);
/* exchange n begins, BEFORE its rows are packed (launch stream): acknowledges exchange n - 1 to the senders of the slots in
 * src_mask (bit d: slot d has a sender), and makes the launch stream wait for the copies that read the staging buffers of this
 * set two exchanges ago. */
int mi_peer_link_begin(mi_peer_link* link, void* launch_stream, unsigned n, int src_mask);
/* chunk k of `chunks` of exchange n has been packed on the launch stream into src_up / src_dn [device; the caller's staging
 * buffers of this exchange's set, NULL for an edge without neighbour]: bytes [first_byte, first_byte + bytes) of both go to the
 * same place of the neighbours' slots on the copy stream, followed by the arrival number. */
int mi_peer_link_send(mi_peer_link* link, void* launch_stream, unsigned n, int k, int chunks, size_t first_byte, size_t bytes,
                      const void* src_up, const void* src_dn);
/* the launch stream waits for chunk k of slot d of exchange n; *slot [device]: the WHOLE slot */
int mi_peer_link_recv(mi_peer_link* link, void* launch_stream, unsigned n, int k, int chunks, int d, void** slot);
/* one exchange in one call (one chunk): send of both edges, then the waits for the slots in src_mask (*slot_lo = slot 0,
 * *slot_hi = slot 1; NULL where there is no sender).  The caller packs before it, unpacks after it. */
int mi_peer_exchange(mi_peer_link* link, void* launch_stream, unsigned n, int src_mask, const void* src_up, const void* src_dn,
                     void** slot_lo, void** slot_hi);
/* waits of this link that ended by their timeout so far (synchronises with the device) */
int mi_peer_link_status(mi_peer_link* link, int* timed_out);
/* teardown in two steps: every rank unmaps its neighbours (disconnect) before any rank frees what it exported (destroy) */
int mi_peer_link_disconnect(mi_peer_link* link);
int mi_peer_link_destroy(mi_peer_link* link);

#ifdef __cplusplus
}
#endif
#endif /* MI_LSDECONV_H */
