/* mi_crossmips.h -- C ABI of TeraStitcher's MIP-NCC pairwise tile registration (crossmips).
 *
 * Replaces  NCC_descr_t* norm_cross_corr_mips(real_t* A, real_t* B, int dimk, int dimi, int dimj,
 *           int nk, int ni, int nj, int delayk, int delayi, int delayj, int side, NCC_parms_t* p)
 * (TeraStitcher/src/crossmips/CrossMIPs.h:89-91, libcrossmips.cpp:101-515) as called by
 * PDAlgoMIPNCC::execute (stitcher/PDAlgoMIPNCC.cpp:96-97).  Differences from the reference ABI:
 * an error code instead of iom::exception / exit(); the result is written into a caller-owned
 * struct instead of a new-allocated one; no static device buffers (re-entrant per stream).
 * Conventions: include/mi_common.h.
 */
#ifndef MI_CROSSMIPS_H
#define MI_CROSSMIPS_H

#include "mi_common.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MI_NORTH_SOUTH 0 /* CrossMIPs.h:51 (== dir_vertical, Displacement.h:44) */
#define MI_WEST_EAST 1   /* CrossMIPs.h:52 (== dir_horizontal) */

/* NCC_descr_t (CrossMIPs.h:58-62): offset of B relative to A as (V, H, D), peak values, half widths */
typedef struct {
    int coord[3];
    float NCC_maxs[3];
    int NCC_widths[3];
} mi_ncc_descr;

/* NCC_parms_t (CrossMIPs.h:65-86) without the `enhance` transform tables (enhance must be 0:
 * PDAlgoMIPNCC.cpp:81 always passes false).  wRangeThr_* are IN-OUT: clamped to the clamped search
 * ranges exactly like libcrossmips.cpp:275-277, and read back by the caller (PDAlgoMIPNCC.cpp:104-106). */
typedef struct {
    int enhance;
    int maxIter;
    float maxThr;
    float widthThr;
    int wRangeThr_i, wRangeThr_j, wRangeThr_k;
    int minPoints;
    int minDim_NCCsrc;
    int minDim_NCCmap;
    float UNR_NCC;
    int INF_W;
    int INV_COORD;
} mi_ncc_params;

/* the fixed parameter set of PDAlgoMIPNCC::execute (PDAlgoMIPNCC.cpp:80-94) for search ranges
 * displ_max_{V,H,D} */
void mi_ncc_default_params(int displ_max_V, int displ_max_H, int displ_max_D, mi_ncc_params* p);

/* One pair; A and B are DEVICE pointers to dimk*dimi*dimj floats in [0,1] (read-only).
 * Semantics: Appendix B of SURVEY.md / libcrossmips.cpp:101-515.  Returns MI_ERR_INVALID where the
 * reference throws (wRangeThr > delay :212-219, bad side :316, nk != 0).  Synchronises `stream`. */
int mi_ncc_mips(int dev, void* stream, const float* A, const float* B, int dimk, int dimi, int dimj,
                int nk, int ni, int nj, int delayk, int delayi, int delayj, int side,
                mi_ncc_params* p, mi_ncc_descr* out);

/* Same with HOST pointers (drop-in for the reference signature: uploads both tiles first). */
int mi_ncc_mips_host(int dev, void* stream, const float* A, const float* B, int dimk, int dimi, int dimj,
                     int nk, int ni, int nj, int delayk, int delayi, int delayj, int side,
                     mi_ncc_params* p, mi_ncc_descr* out);

/* n_pairs independent pairs with device-resident tiles: pair q aligns tiles[a_idx[q]] and
 * tiles[b_idx[q]] (all tiles dimk*dimi*dimj) with side[q], nominal offsets ni[q]/nj[q] and its own
 * in-out params[q]; results in out[q].  All index/param arrays are [host].  Pairs of equal geometry go
 * through the device together (MIPs, tables, lag-transform cross terms, neighbourhood refinement: a fixed
 * number of launches and ONE synchronisation per group); only the final windows return to the host rules.
 * Synchronises. */
int mi_ncc_mips_batch(int dev, void* stream, int n_pairs, const float* const* tiles,
                      const int* a_idx, const int* b_idx, int dimk, int dimi, int dimj,
                      const int* ni, const int* nj, int delayk, int delayi, int delayj, const int* side,
                      mi_ncc_params* params, mi_ncc_descr* out);

/* The same batch on tiles kept as the integer samples they were loaded from (16-bit; mi_ncc_mips_batch_u8: 8-bit).  The reference turns the samples of its TIFF tiles into
 * iom::real_t in [0, 1] when it loads them (value / 255 or / 65535, iomanager tiff2D.cpp:606-610) and compute_3_MIPs
 * (compute_funcs.cu:502-521) reads those floats; the division is monotonic, so the MIPs of the floats are the divided MIPs of the
 * integers and everything downstream is unchanged: every field of every result equals mi_ncc_mips_batch on the converted tiles.
 * The MIP pass -- half of a batch's device time -- reads half the bytes.  tile value = sample / scale (65535; 255 for 8-bit samples
 * widened to 16 bits).  Needs an even dimj and dimk <= 32 (MI_ERR_UNSUPPORTED otherwise: convert and use the float entry). */
int mi_ncc_mips_batch_u16(int dev, void* stream, int n_pairs, const unsigned short* const* tiles, float scale,
                          const int* a_idx, const int* b_idx, int dimk, int dimi, int dimj,
                          const int* ni, const int* nj, int delayk, int delayi, int delayj, const int* side,
                          mi_ncc_params* params, mi_ncc_descr* out);
int mi_ncc_mips_batch_u8(int dev, void* stream, int n_pairs, const unsigned char* const* tiles, float scale,
                         const int* a_idx, const int* b_idx, int dimk, int dimi, int dimj,
                         const int* ni, const int* nj, int delayk, int delayi, int delayj, const int* side,
                         mi_ncc_params* params, mi_ncc_descr* out);

/* A batch in two halves, for callers that walk the z layers of a grid (StackStitcher.cpp:223-374 computes the same pairs layer
 * after layer): _begin copies the arguments, enqueues the device stage of every group and returns; _end waits for it, runs the
 * host rules (and the per-pair path of whatever the batched pipeline hands back), fills params [in-out blocks, may be NULL] and
 * out, and destroys the job.  Beginning layer l + 1 before ending layer l lets the first MIP pass of the next batch run beside
 * the last lag chain of this one -- the one chain of a batch that has nothing to hide behind.  sample_bytes: 4 (float tiles),
 * 2 or 1 (integer samples, tile value = sample / scale: see mi_ncc_mips_batch_u16).  The tiles stay alive and unchanged until
 * _end; jobs are ended in the order they were begun.  mi_ncc_mips_batch == _begin + _end. */
typedef struct mi_ncc_batch_job mi_ncc_batch_job;
int mi_ncc_mips_batch_begin(int dev, void* stream, int n_pairs, const void* const* tiles, int sample_bytes, float scale,
                            const int* a_idx, const int* b_idx, int dimk, int dimi, int dimj,
                            const int* ni, const int* nj, int delayk, int delayi, int delayj, const int* side,
                            const mi_ncc_params* params, mi_ncc_batch_job** job);
int mi_ncc_mips_batch_end(mi_ncc_batch_job* job, mi_ncc_params* params, mi_ncc_descr* out);

/* Cumulative counters of this process: out3[0] pairs finished by the batched pipeline, out3[1] pairs finished by the per-pair
 * path (geometries the lag transform does not take, MI_NCC_DIRECT=1, and pairs handed back because a decision was inside the
 * resolution of the map values), out3[2] map entries recomputed in the reference's two-pass fp64 form
 * (compute_funcs.cu:1163-1292) to take such decisions.  reset != 0 clears them after reading. */
void mi_ncc_stats(long long* out3, int reset);

/* Measurement hook (bench.py's roofline object): average duration in ms of ONE launch of the MIP kernel -- the streaming pass
 * over both overlap views, compute_3_MIPs (compute_funcs.cu:502-521), 2 * dimk * dimi_v * dimj_v * 4 bytes per pair -- over
 * n_pairs pairs of one geometry, HIP events on `stream` around `reps` launches.  Synchronises. */
int mi_ncc_time_mips(int dev, void* stream, int n_pairs, const float* const* tiles, const int* a_idx, const int* b_idx,
                     int dimk, int dimi, int dimj, int ni, int nj, int side, int reps, float* ms_per_launch);
/* ... of the integer MIP kernels (2 bytes / 1 byte per sample) */
int mi_ncc_time_mips_u16(int dev, void* stream, int n_pairs, const unsigned short* const* tiles, float scale, const int* a_idx,
                         const int* b_idx, int dimk, int dimi, int dimj, int ni, int nj, int side, int reps, float* ms_per_launch);
int mi_ncc_time_mips_u8(int dev, void* stream, int n_pairs, const unsigned char* const* tiles, float scale, const int* a_idx,
                        const int* b_idx, int dimk, int dimi, int dimj, int ni, int nj, int side, int reps, float* ms_per_launch);

/* ---- building blocks (exposed for parity tests against the reference's exported helpers) ------ */

/* compute_3_MIPs (compute_funcs.cu:502-521) on the overlap views selected by (side, ni, nj):
 * writes xy[dimi_v*dimj_v], xz[dimi_v*dimk], yz[dimj_v*dimk] for A then B (6 device arrays). */
int mi_ncc_compute_mips(int dev, void* stream, const float* A, const float* B, int dimk, int dimi, int dimj,
                        int ni, int nj, int side, float* xy1, float* xz1, float* yz1,
                        float* xy2, float* xz2, float* yz2);

/* compute_NCC_map (compute_funcs.cu:939-1160): map[(2*delayu+1)*(2*delayv+1)] (device) from two
 * dimu x dimv MIPs (device).  Window means / variances from fp64 summed-area tables (with the reference's
 * float tile sums), cross terms in fp64: shift by shift from LDS-staged MIP rows (mi_ncc_compute_map: blocks of
 * 4 x 8 shifts per wave), or through the lag transform along the long axis of the MIP that the batched pair
 * pipeline uses (mi_ncc_compute_map_lag: fp64 FFT along the long axis, direct correlation along the short one). */
int mi_ncc_compute_map(int dev, void* stream, const float* mip1, const float* mip2, int dimu, int dimv,
                       int delayu, int delayv, float* map);
int mi_ncc_compute_map_lag(int dev, void* stream, const float* mip1, const float* mip2, int dimu, int dimv,
                           int delayu, int delayv, float* map);

#ifdef __cplusplus
}
#endif
#endif /* MI_CROSSMIPS_H */
