// Batched lag-transform MIP-NCC pipeline (ncc_lag.hip), called by the C ABI entry points in ncc.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "mi_crossmips.h"

namespace mi {

#ifndef MI_TILE_FMT_DEFINED
#define MI_TILE_FMT_DEFINED
struct TileFmt {  // (see ncc_core.h)
    int bytes = 4;
    float scale = 65535.0f;
};
#endif

// resolution below which a decision is not taken on lag-transform / summed-area-table values (MI_NCC_MARGIN, default 4e-6)
float ncc_margin();
// whether every plane of this geometry fits the lag transform (FFT length <= 8192, LDS)
bool ncc_lag_supported(int dimk, int dimi, int dimj, int ni, int nj, int delayk, int delayi, int delayj, int side, const mi_ncc_params* p);
// n pairs of one geometry in two steps, so that several groups can be in flight: enqueue the device stage (on streams of the job's
// own, behind what `s` holds so far), then wait for it and run the host rules.  careful[q] != 0: pair q must be redone by the
// per-pair path, out[q] untouched.  A job that is not finished must be abandoned.
struct LagJob;
// groups_in_flight: how many groups the caller keeps enqueued at once -- they share the device-memory budget of a chunk
// (MI_NCC_CHUNK_MB); concurrent callers on one device share its three streams and serialise on them.
// defer_chains: only the MIP pass is enqueued; the caller enqueues the chains of all its jobs afterwards (ncc_lag_enqueue_chains)
// behind an event it records on the MIP stream after the last job's MIP pass
int ncc_lag_enqueue(int dev, hipStream_t s, int n, const float* const* a_ptrs, const float* const* b_ptrs, int dimk, int dimi, int dimj, int ni,
                    int nj, int delayk, int delayi, int delayj, int side, mi_ncc_params* params, LagJob** job, bool defer_chains = false,
                    TileFmt fmt = TileFmt(), int groups_in_flight = 1, bool chain_beside = false);
// (chain_beside: an earlier group's chain is still running when this group's MIP pass starts -- launch_mips)
int ncc_lag_enqueue_chains(LagJob* job, hipEvent_t gate);
hipStream_t ncc_lag_mip_stream(LagJob* job);
int ncc_lag_finish(LagJob* job, mi_ncc_params* params, mi_ncc_descr* out, unsigned char* careful);
void ncc_lag_abandon(LagJob* job);
// both steps at once
int ncc_lag_group(int dev, hipStream_t s, int n, const float* const* a_ptrs, const float* const* b_ptrs, int dimk, int dimi, int dimj, int ni,
                  int nj, int delayk, int delayi, int delayj, int side, mi_ncc_params* params, mi_ncc_descr* out, unsigned char* careful,
                  TileFmt fmt = TileFmt());
int ncc_lag_map(int dev, hipStream_t s, const float* mip1, const float* mip2, int dimu, int dimv, int delayu, int delayv, float* map);
int ncc_time_mips(int dev, hipStream_t s, int n, const float* const* a_ptrs, const float* const* b_ptrs, int dimk, int dimi, int dimj, int ni, int nj,
                  int side, int reps, float* ms, TileFmt fmt = TileFmt());
void ncc_lag_drop_cached(int dev);

}  // namespace mi
