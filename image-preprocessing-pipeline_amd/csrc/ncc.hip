// MIP-NCC pairwise tile registration on the device (replaces TeraStitcher/src/crossmips:
// libcrossmips.cpp:101-515 norm_cross_corr_mips and the CPU/CUDA branches of compute_funcs.cu).
//
// Device side
//   k_mips        one streaming pass over both overlap views -> 6 MIPs (compute_3_MIPs, :502-521).
//                 A lane owns one column j of a 16-row band and keeps the xy maxima in registers; the
//                 xz row maxima are reduced per wave with DPP shuffles, xz/yz partials are merged with
//                 integer atomicMax on the float bit pattern (all values >= 0 as the MIPs start at 0).
//   k_tile_sums   32x32 tile sums with the reference's FLOAT running sum in row/column order
//                 (seq_cpu_compute_partial_sums, :474-500) -- bit-identical, one lane per tile.
//   k_mip_mean / k_sat_rows / k_sat_cols   fp64 summed-area tables of (f - c0), (f - c0)^2 and of the float tile sums:
//                 every term of compute_NCC (:1163-1292) except the cross term sum f*t becomes O(1) per shift, with
//                 the reference's means (float tile sums + border pixels) reproduced exactly.
//   k_ncc_blk / k_ncc_finish   cross terms sum f*t in fp64 for groups of 8 blocks of 4 x 8 shifts from LDS-staged MIP rows, partial sums per
//                 row chunk added in a fixed order; replaces gpu_NCC_map/gpu_NCC_miss (:730-935).  Serves full maps and the
//                 "missing entries" of the neighbourhood refinement alike.
// Host side (this file, plain C++): argmax, neighbourhood refinement, peak widths and the final
// alignment rules, kept bit-identical in float/int arithmetic to compute_funcs.cu:160-342,1294-1609.
#include <atomic>
#include <map>

#include "ncc_core.h"
#include "ncc_lag.h"

namespace mi {
static std::atomic<long long> g_ncc_stats[3];
void ncc_count(int which, long long n) { g_ncc_stats[which] += n; }
}  // namespace mi

extern "C" void mi_ncc_stats(long long* out3, int reset) {
    for (int i = 0; i < 3; ++i) {
        if (out3) out3[i] = mi::g_ncc_stats[i].load();
        if (reset) mi::g_ncc_stats[i].store(0);
    }
}

namespace mi {
// the cached pair slots of a device (-1: all) are destroyed; their buffers return to the pool (mi_release_cached_memory)
void ncc_drop_cached_slots(int dev) {
    std::vector<std::unique_ptr<PairSlot>> drop;
    {
        std::lock_guard<std::mutex> g(g_slot_mu);
        for (size_t i = 0; i < g_slots.size();)
            if (dev < 0 || g_slots[i]->dev == dev) { drop.push_back(std::move(g_slots[i])); g_slots.erase(g_slots.begin() + i); }
            else ++i;
    }
    ncc_lag_drop_cached(dev);
}
}  // namespace mi

namespace {
// MI_NCC_DIRECT=1: every pair takes the per-pair path (shift-by-shift fp64 cross terms, host-driven refinement) -- the A/B
// reference of the batched lag-transform pipeline and its fallback
bool force_direct() {
    const char* e = std::getenv("MI_NCC_DIRECT");
    return e && *e && *e != '0';
}

// the per-pair path of one pair (also the careful path of pairs the batched pipeline hands back)
int pair_direct(hipStream_t s, const float* A, const float* B, int dimk, int dimi, int dimj, int ni, int nj, int delayk, int delayi, int delayj,
                int side, mi_ncc_params* p, mi_ncc_descr* out) {
    PairPlan pl;
    MI_TRY(plan_pair(dimk, dimi, dimj, 0, ni, nj, delayk, delayi, delayj, side, p, pl));
    Workspace ws;
    MI_TRY(pair_enqueue(s, A, B, dimi, dimj, pl, ws));
    ncc_count(1, 1);
    return pair_finish(s, ni, nj, side, p, pl, ws, out);
}
}  // namespace

extern "C" void mi_ncc_default_params(int displ_max_V, int displ_max_H, int displ_max_D, mi_ncc_params* p) {
    if (!p) return;
    const int width_max = 30;  // S_NCC_WIDTH_MAX, stitcher/S_config.h:86
    p->enhance = 0;
    p->maxIter = 2;
    p->maxThr = 0.10f;
    p->UNR_NCC = 0.0f;  // S_NCC_PEAK_MIN, S_config.h:83
    p->minPoints = 3;
    p->wRangeThr_i = imin(displ_max_V, width_max - 1);
    p->wRangeThr_j = imin(displ_max_H, width_max - 1);
    p->wRangeThr_k = imin(displ_max_D, width_max - 1);
    p->minDim_NCCsrc = 25;
    p->minDim_NCCmap = 3;
    p->INF_W = imax(p->wRangeThr_i, imax(p->wRangeThr_j, p->wRangeThr_k)) + 1;
    p->widthThr = 0.80f;
    p->INV_COORD = 0;
}

extern "C" int mi_ncc_mips(int dev, void* stream, const float* A, const float* B, int dimk, int dimi, int dimj, int nk, int ni, int nj,
                           int delayk, int delayi, int delayj, int side, mi_ncc_params* p, mi_ncc_descr* out) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(A && B && out, "CrossMIPs: null pointer");
    {   // parameter checks first (the messages of the reference), on a copy: the paths below clamp `p` themselves
        mi_ncc_params probe = p ? *p : mi_ncc_params{};
        PairPlan pl;
        MI_TRY(plan_pair(dimk, dimi, dimj, nk, ni, nj, delayk, delayi, delayj, side, p ? &probe : nullptr, pl));
    }
    hipStream_t s = as_stream(stream);
    if (!force_direct() && ncc_lag_supported(dimk, dimi, dimj, ni, nj, delayk, delayi, delayj, side, p)) {
        unsigned char careful = 1;
        MI_TRY(ncc_lag_group(dev, s, 1, &A, &B, dimk, dimi, dimj, ni, nj, delayk, delayi, delayj, side, p, out, &careful));
        if (!careful) return MI_OK;
    }
    return pair_direct(s, A, B, dimk, dimi, dimj, ni, nj, delayk, delayi, delayj, side, p, out);
}

extern "C" int mi_ncc_mips_host(int dev, void* stream, const float* A, const float* B, int dimk, int dimi, int dimj, int nk, int ni,
                                int nj, int delayk, int delayi, int delayj, int side, mi_ncc_params* p, mi_ncc_descr* out) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(A && B && out, "CrossMIPs: null pointer");
    MI_REQUIRE(dimk > 0 && dimi > 0 && dimj > 0, "CrossMIPs: empty stack");
    hipStream_t s = as_stream(stream);
    const size_t n = (size_t)dimk * dimi * dimj;
    DevBuf dA, dB;
    MI_TRY(dA.alloc(sizeof(float) * n));
    MI_TRY(dB.alloc(sizeof(float) * n));
    MI_HIP(hipMemcpyAsync(dA.p, A, sizeof(float) * n, hipMemcpyHostToDevice, s));
    MI_HIP(hipMemcpyAsync(dB.p, B, sizeof(float) * n, hipMemcpyHostToDevice, s));
    int rc = mi_ncc_mips(dev, stream, dA.as<float>(), dB.as<float>(), dimk, dimi, dimj, nk, ni, nj, delayk, delayi, delayj, side, p, out);
    hipError_t e = hipStreamSynchronize(s);
    if (rc == MI_OK && e != hipSuccess) rc = fail(MI_ERR_HIP, "mi_ncc_mips_host: %s", hipGetErrorString(e));
    return rc;
}

static int ncc_batch(int dev, void* stream, int n_pairs, const float* const* tiles, const int* a_idx, const int* b_idx, int dimk, int dimi,
                     int dimj, const int* ni, const int* nj, int delayk, int delayi, int delayj, const int* side, mi_ncc_params* params,
                     mi_ncc_descr* out, TileFmt fmt);

extern "C" int mi_ncc_mips_batch(int dev, void* stream, int n_pairs, const float* const* tiles, const int* a_idx, const int* b_idx,
                                 int dimk, int dimi, int dimj, const int* ni, const int* nj, int delayk, int delayi, int delayj,
                                 const int* side, mi_ncc_params* params, mi_ncc_descr* out) {
    return ncc_batch(dev, stream, n_pairs, tiles, a_idx, b_idx, dimk, dimi, dimj, ni, nj, delayk, delayi, delayj, side, params, out, TileFmt());
}

// The same batch on tiles kept as the integer samples they were loaded from: tile value = sample / scale (65535 for 16-bit, 255 for
// 8-bit samples: tiff2D.cpp:606-610); every result is identical to mi_ncc_mips_batch on the converted tiles.  Needs rows of whole
// 32-bit words (dimj even / a multiple of 4) and dimk <= 32 (MI_ERR_UNSUPPORTED otherwise: convert the tiles, use the float entry).
static int ncc_batch_int(const char* who, int bytes, int dev, void* stream, int n_pairs, const void* const* tiles, float scale, const int* a_idx,
                         const int* b_idx, int dimk, int dimi, int dimj, const int* ni, const int* nj, int delayk, int delayi, int delayj,
                         const int* side, mi_ncc_params* params, mi_ncc_descr* out) {
    if (!(scale > 0.0f)) return fail(MI_ERR_INVALID, "%s: scale must be positive", who);
    if (!mips_int_ok(bytes, dimk, dimj, (size_t)dimi * dimj))
        return fail(MI_ERR_UNSUPPORTED, "%s: integer tiles need rows of whole 32-bit words and at most %d slices", who, 4 * MIP_KPW);
    TileFmt fmt;
    fmt.bytes = bytes;
    fmt.scale = scale;
    return ncc_batch(dev, stream, n_pairs, reinterpret_cast<const float* const*>(tiles), a_idx, b_idx, dimk, dimi, dimj, ni, nj, delayk, delayi, delayj,
                     side, params, out, fmt);
}
extern "C" int mi_ncc_mips_batch_u16(int dev, void* stream, int n_pairs, const unsigned short* const* tiles, float scale, const int* a_idx,
                                     const int* b_idx, int dimk, int dimi, int dimj, const int* ni, const int* nj, int delayk, int delayi, int delayj,
                                     const int* side, mi_ncc_params* params, mi_ncc_descr* out) {
    return ncc_batch_int("mi_ncc_mips_batch_u16", 2, dev, stream, n_pairs, reinterpret_cast<const void* const*>(tiles), scale, a_idx, b_idx, dimk, dimi,
                         dimj, ni, nj, delayk, delayi, delayj, side, params, out);
}
extern "C" int mi_ncc_mips_batch_u8(int dev, void* stream, int n_pairs, const unsigned char* const* tiles, float scale, const int* a_idx,
                                    const int* b_idx, int dimk, int dimi, int dimj, const int* ni, const int* nj, int delayk, int delayi, int delayj,
                                    const int* side, mi_ncc_params* params, mi_ncc_descr* out) {
    return ncc_batch_int("mi_ncc_mips_batch_u8", 1, dev, stream, n_pairs, reinterpret_cast<const void* const*>(tiles), scale, a_idx, b_idx, dimk, dimi,
                         dimj, ni, nj, delayk, delayi, delayj, side, params, out);
}

// A batch in two halves: begin enqueues the device stage of every group (and returns), end waits for it, runs the host rules and
// the per-pair path of whatever the batched pipeline handed back.  A caller that walks the z layers of a grid (StackStitcher.cpp:
// 223-374) begins layer l + 1 before it ends layer l: the device's streams are per DEVICE (ncc_lag.hip), so the first MIP pass of the
// next batch runs beside the last chain of this one -- the only chain of a call that otherwise has nothing to hide behind.
struct mi_ncc_batch_job {
    int dev = 0;
    hipStream_t user = nullptr;
    TileFmt fmt;
    int n_pairs = 0, dimk = 0, dimi = 0, dimj = 0, delayk = 0, delayi = 0, delayj = 0;
    std::vector<const float*> ta, tb;                 // the tiles of pair q (the caller keeps them alive until end)
    std::vector<int> ni, nj, side;
    std::vector<mi_ncc_params> params;                // in-out parameter blocks, working copies
    struct InFlight {
        std::vector<int> idx;
        LagJob* job = nullptr;
        std::vector<const float*> pa, pb;
        std::vector<mi_ncc_params> pp;
    };
    std::vector<InFlight> flights;
    std::vector<int> todo;                            // pairs for the per-pair path
    bool counted = false;
    ~mi_ncc_batch_job() {
        for (InFlight& f : flights)
            if (f.job) ncc_lag_abandon(f.job);
    }
};
static std::atomic<int> g_batches_in_flight[16];     // per device: batches begun and not yet ended

static int ncc_batch_begin(mi_ncc_batch_job& J) {
    MI_TRY(use_device(J.dev));
    const int n_pairs = J.n_pairs, dimk = J.dimk, dimi = J.dimi, dimj = J.dimj;
    // Batched pipeline (ncc_lag.hip): pairs of equal geometry (side, nominal offsets, parameters) go through the device together,
    // one synchronisation per group.  Pairs it hands back (a decision inside the resolution of its values, a move outside the
    // transformed lags) and geometries it does not take continue on the per-pair path (ncc_batch_end).
    if (force_direct()) {
        for (int q = 0; q < n_pairs; ++q) J.todo.push_back(q);
        return MI_OK;
    }
    struct Key {
        int side, ni, nj;
        mi_ncc_params p;
        bool operator<(const Key& o) const { return std::memcmp(this, &o, sizeof(Key)) < 0; }
    };
    std::map<Key, std::vector<int>> groups;
    for (int q = 0; q < n_pairs; ++q) {
        Key k;
        std::memset(&k, 0, sizeof k);
        k.side = J.side[q]; k.ni = J.ni[q]; k.nj = J.nj[q]; k.p = J.params[q];
        groups[k].push_back(q);
    }
    // every group's device stage is enqueued before the first one is waited for: the host rules of a group run while the next
    // group's kernels do, and the tail of one group's lag chain overlaps the next group's MIP pass
    static const char* env_ser = MI_PROBE_ENV("MI_NCC_SERIAL_MIPS");
    const bool serial_mips = env_ser ? std::atoi(env_ser) != 0 : false;
    const bool earlier_batch = g_batches_in_flight[J.dev & 15].fetch_add(1) > 0;   // (its last chain is still under way)
    J.counted = true;
    J.flights.reserve(groups.size());
    for (auto& kv : groups) {
        const std::vector<int>& idx = kv.second;
        const int q0 = idx[0], n = (int)idx.size();
        {   // the reference's parameter checks, on a copy
            mi_ncc_params probe = J.params[q0];
            PairPlan pl;
            MI_TRY(plan_pair(dimk, dimi, dimj, 0, J.ni[q0], J.nj[q0], J.delayk, J.delayi, J.delayj, J.side[q0], &probe, pl));
        }
        if (!ncc_lag_supported(dimk, dimi, dimj, J.ni[q0], J.nj[q0], J.delayk, J.delayi, J.delayj, J.side[q0], &J.params[q0])) {
            J.todo.insert(J.todo.end(), idx.begin(), idx.end());
            continue;
        }
        J.flights.emplace_back();
        mi_ncc_batch_job::InFlight& f = J.flights.back();
        f.idx = idx;
        f.pa.resize(n); f.pb.resize(n); f.pp.resize(n);
        for (int i = 0; i < n; ++i) { f.pa[i] = J.ta[idx[i]]; f.pb[i] = J.tb[idx[i]]; f.pp[i] = J.params[idx[i]]; }
        MI_TRY(ncc_lag_enqueue(J.dev, J.user, n, f.pa.data(), f.pb.data(), dimk, dimi, dimj, J.ni[q0], J.nj[q0], J.delayk, J.delayi, J.delayj, J.side[q0],
                               f.pp.data(), &f.job, serial_mips, J.fmt, (int)groups.size(), (J.flights.size() > 1 || earlier_batch) && !serial_mips));
    }
    if (serial_mips && !J.flights.empty()) {
        // MI_NCC_SERIAL_MIPS=1 (probe builds): every MIP pass first (one HBM-bound stream after the other), then every chain
        hipEvent_t gate = nullptr;
        hipStream_t sm = ncc_lag_mip_stream(J.flights.back().job);
        int rc = MI_OK;
        if (hipEventCreateWithFlags(&gate, hipEventDisableTiming) != hipSuccess || hipEventRecord(gate, sm) != hipSuccess)
            rc = fail(MI_ERR_HIP, "mi_ncc_mips_batch: event setup failed");
        for (mi_ncc_batch_job::InFlight& f : J.flights)
            if (rc == MI_OK) rc = ncc_lag_enqueue_chains(f.job, gate);
        if (gate) (void)hipEventDestroy(gate);
        MI_TRY(rc);
    }
    return MI_OK;
}

static int ncc_batch_end(mi_ncc_batch_job& J, mi_ncc_params* params, mi_ncc_descr* out) {
    MI_TRY(use_device(J.dev));
    const int dev = J.dev, dimk = J.dimk, dimi = J.dimi, dimj = J.dimj, delayk = J.delayk, delayi = J.delayi, delayj = J.delayj;
    const TileFmt fmt = J.fmt;
    hipStream_t user = J.user;
    std::vector<int>& todo = J.todo;
    int rc = MI_OK;
    for (mi_ncc_batch_job::InFlight& f : J.flights) {
        if (rc != MI_OK) { ncc_lag_abandon(f.job); f.job = nullptr; continue; }
        const int n = (int)f.idx.size();
        std::vector<mi_ncc_descr> po(n);
        std::vector<unsigned char> careful(n, 1);
        rc = ncc_lag_finish(f.job, f.pp.data(), po.data(), careful.data());
        f.job = nullptr;
        if (rc != MI_OK) continue;
        for (int i = 0; i < n; ++i) {
            if (careful[i]) { todo.push_back(f.idx[i]); continue; }
            J.params[f.idx[i]] = f.pp[i];
            out[f.idx[i]] = po[i];
        }
    }
    if (rc != MI_OK) return rc;
    std::sort(todo.begin(), todo.end());
    const int* ni = J.ni.data();
    const int* nj = J.nj.data();
    const int* side = J.side.data();
    mi_ncc_params* wparams = J.params.data();
    auto tile_a = [&](int q) { return J.ta[q]; };
    auto tile_b = [&](int q) { return J.tb[q]; };
    const int n_todo = (int)todo.size();
    if (n_todo == 0) {
        if (params)
            for (int q = 0; q < J.n_pairs; ++q) params[q] = J.params[q];
        return MI_OK;
    }
    // Per-pair path.  The host side of a pair (argmax, neighbourhood refinement with its small launches and stream syncs, widths,
    // alignment) is a serial latency chain of about a millisecond, and few of a pair's kernels fill the device on their own: NT
    // host threads, each with its own stream and workspace, take every NT-th pair; within a thread the next pair's kernels are
    // enqueued before the current pair is refined.  All streams start after, and are joined back into, `stream`.
    int want = 3;
    if (const char* e = std::getenv("MI_NCC_THREADS")) want = std::max(1, std::min(32, std::atoi(e)));
    const int NT = std::min(n_todo, want);
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, user) != hipSuccess) {
        if (ev) (void)hipEventDestroy(ev);
        return fail(MI_ERR_HIP, "mi_ncc_mips_batch: event setup failed");
    }
    std::vector<int> rcs(NT, MI_OK);
    std::vector<std::string> msgs(NT);
    auto worker = [&](int t) {
        // workspaces (device tables, pinned staging) and streams are kept between calls: setting them up costs milliseconds
        struct Slot { std::unique_ptr<PairSlot> r; Workspace& ws; PairPlan pl; hipStream_t s; };
        int rc = use_device(dev);
        std::unique_ptr<PairSlot> res[2] = {take_pair_slot(dev), take_pair_slot(dev)};
        for (auto& r : res)
            if (rc == MI_OK && (!r || !r->s || hipStreamWaitEvent(r->s, ev, 0) != hipSuccess))
                rc = fail(MI_ERR_HIP, "mi_ncc_mips_batch: stream setup failed");
        if (rc != MI_OK) { msgs[t] = mi_last_error(); rcs[t] = rc; return; }
        Slot slot[2] = {{nullptr, res[0]->ws, PairPlan(), res[0]->s}, {nullptr, res[1]->ws, PairPlan(), res[1]->s}};
        int k = 0;  // index of the pair within this thread's sequence t, t + NT, ...
        for (int qi = t; rc == MI_OK; qi += NT, ++k) {
            if (qi < n_todo) {
                const int q = todo[qi];
                Slot& sl = slot[k & 1];
                rc = plan_pair(dimk, dimi, dimj, 0, ni[q], nj[q], delayk, delayi, delayj, side[q], &wparams[q], sl.pl);
                if (rc == MI_OK) rc = pair_enqueue(sl.s, tile_a(q), tile_b(q), dimi, dimj, sl.pl, sl.ws, fmt);
            }
            const int fi = qi - NT;
            if (fi >= 0 && fi < n_todo && rc == MI_OK) {
                const int f = todo[fi];
                Slot& pr = slot[(k - 1) & 1];
                rc = pair_finish(pr.s, ni[f], nj[f], side[f], &wparams[f], pr.pl, pr.ws, &out[f]);
                ncc_count(1, 1);
            }
            if (qi >= n_todo) break;
        }
        if (rc != MI_OK) msgs[t] = mi_last_error();
        for (auto& r : res) {
            (void)hipStreamSynchronize(r->s);
            give_pair_slot(std::move(r));
        }
        rcs[t] = rc;
    };
    std::vector<std::thread> pool;
    int started = 1;  // this thread is worker 0
    try {
        for (int t = 1; t < NT; ++t) { pool.emplace_back(worker, t); ++started; }
    } catch (const std::exception&) {  // no more threads: the pairs of the missing workers are taken below
    }
    worker(0);
    for (auto& th : pool) th.join();
    for (int t = started; t < NT; ++t) worker(t);
    (void)hipEventDestroy(ev);
    for (int t = 0; t < NT; ++t)
        if (rcs[t] != MI_OK) return fail(rcs[t], "%s", msgs[t].c_str());
    if (params)
        for (int q = 0; q < J.n_pairs; ++q) params[q] = J.params[q];
    return MI_OK;
}

static int ncc_batch_make(int dev, void* stream, int n_pairs, const float* const* tiles, const int* a_idx, const int* b_idx, int dimk, int dimi, int dimj,
                          const int* ni, const int* nj, int delayk, int delayi, int delayj, const int* side, const mi_ncc_params* params, TileFmt fmt,
                          std::unique_ptr<mi_ncc_batch_job>& out) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(n_pairs >= 0, "mi_ncc_mips_batch: negative pair count");
    MI_REQUIRE(n_pairs == 0 || (tiles && a_idx && b_idx && ni && nj && side && params), "mi_ncc_mips_batch: null pointer");
    out.reset(new (std::nothrow) mi_ncc_batch_job);
    if (!out) return fail(MI_ERR_NOMEM, "mi_ncc_mips_batch: out of host memory");
    mi_ncc_batch_job& J = *out;
    J.dev = dev; J.user = as_stream(stream); J.fmt = fmt; J.n_pairs = n_pairs;
    J.dimk = dimk; J.dimi = dimi; J.dimj = dimj; J.delayk = delayk; J.delayi = delayi; J.delayj = delayj;
    J.ta.resize(n_pairs); J.tb.resize(n_pairs);
    for (int q = 0; q < n_pairs; ++q) {
        MI_REQUIRE(tiles[a_idx[q]] && tiles[b_idx[q]], "mi_ncc_mips_batch: null tile for pair %d", q);
        J.ta[q] = tiles[a_idx[q]];
        J.tb[q] = tiles[b_idx[q]];
    }
    J.ni.assign(ni, ni + n_pairs); J.nj.assign(nj, nj + n_pairs); J.side.assign(side, side + n_pairs);
    J.params.assign(params, params + n_pairs);
    return MI_OK;
}

static int ncc_batch_finish(mi_ncc_batch_job* job, mi_ncc_params* params, mi_ncc_descr* out) {
    std::unique_ptr<mi_ncc_batch_job> J(job);
    int rc = MI_OK;
    if (J->n_pairs > 0) {
        if (!out) rc = fail(MI_ERR_INVALID, "mi_ncc_mips_batch_end: null pointer");
        else rc = ncc_batch_end(*J, params, out);
    }
    if (J->counted) g_batches_in_flight[J->dev & 15].fetch_sub(1);
    return rc;
}

static int ncc_batch(int dev, void* stream, int n_pairs, const float* const* tiles, const int* a_idx, const int* b_idx, int dimk, int dimi,
                     int dimj, const int* ni, const int* nj, int delayk, int delayi, int delayj, const int* side, mi_ncc_params* params,
                     mi_ncc_descr* out, TileFmt fmt) {
    MI_REQUIRE(n_pairs <= 0 || out, "mi_ncc_mips_batch: null pointer");
    std::unique_ptr<mi_ncc_batch_job> J;
    MI_TRY(ncc_batch_make(dev, stream, n_pairs, tiles, a_idx, b_idx, dimk, dimi, dimj, ni, nj, delayk, delayi, delayj, side, params, fmt, J));
    if (n_pairs == 0) return MI_OK;
    int rc = ncc_batch_begin(*J);
    if (rc != MI_OK) {
        if (J->counted) g_batches_in_flight[dev & 15].fetch_sub(1);
        return rc;
    }
    return ncc_batch_finish(J.release(), params, out);
}

extern "C" int mi_ncc_mips_batch_begin(int dev, void* stream, int n_pairs, const void* const* tiles, int sample_bytes, float scale, const int* a_idx,
                                       const int* b_idx, int dimk, int dimi, int dimj, const int* ni, const int* nj, int delayk, int delayi, int delayj,
                                       const int* side, const mi_ncc_params* params, mi_ncc_batch_job** job) {
    MI_REQUIRE(job, "mi_ncc_mips_batch_begin: null pointer");
    *job = nullptr;
    MI_REQUIRE(sample_bytes == 4 || sample_bytes == 2 || sample_bytes == 1, "mi_ncc_mips_batch_begin: samples of 4 (float), 2 or 1 bytes");
    TileFmt fmt;
    if (sample_bytes != 4) {
        if (!(scale > 0.0f)) return fail(MI_ERR_INVALID, "mi_ncc_mips_batch_begin: scale must be positive");
        if (!mips_int_ok(sample_bytes, dimk, dimj, (size_t)dimi * dimj))
            return fail(MI_ERR_UNSUPPORTED, "mi_ncc_mips_batch_begin: integer tiles need rows of whole 32-bit words and at most %d slices", 4 * MIP_KPW);
        fmt.bytes = sample_bytes;
        fmt.scale = scale;
    }
    std::unique_ptr<mi_ncc_batch_job> J;
    MI_TRY(ncc_batch_make(dev, stream, n_pairs, reinterpret_cast<const float* const*>(tiles), a_idx, b_idx, dimk, dimi, dimj, ni, nj, delayk, delayi,
                          delayj, side, params, fmt, J));
    if (n_pairs > 0) {
        int rc = ncc_batch_begin(*J);
        if (rc != MI_OK) {
            if (J->counted) g_batches_in_flight[dev & 15].fetch_sub(1);
            return rc;
        }
    }
    *job = J.release();
    return MI_OK;
}

extern "C" int mi_ncc_mips_batch_end(mi_ncc_batch_job* job, mi_ncc_params* params, mi_ncc_descr* out) {
    if (!job) return MI_OK;
    return ncc_batch_finish(job, params, out);
}

extern "C" int mi_ncc_compute_mips(int dev, void* stream, const float* A, const float* B, int dimk, int dimi, int dimj, int ni, int nj,
                                   int side, float* xy1, float* xz1, float* yz1, float* xy2, float* xz2, float* yz2) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(A && B && xy1 && xz1 && yz1 && xy2 && xz2 && yz2, "compute_3_MIPs: null pointer");
    MI_REQUIRE(side == MI_NORTH_SOUTH || side == MI_WEST_EAST, "CrossMIPs: unexpected alignment configuration");
    MI_REQUIRE(dimk > 0 && ni >= 0 && nj >= 0 && ni < dimi && nj < dimj, "compute_3_MIPs: invalid extents");
    hipStream_t s = as_stream(stream);
    const int dimi_v = side == MI_NORTH_SOUTH ? dimi - ni : dimi, dimj_v = side == MI_WEST_EAST ? dimj - nj : dimj;
    DevBuf tmp;
    MI_TRY(tmp.alloc(sizeof(float) * mips_tmp_floats(dimk, dimi_v, dimj_v)));
    MI_TRY(launch_mips(s, A, B, nullptr, 1, 0, dimk, dimi_v, dimj_v, (size_t)dimi * dimj, dimj, side == MI_NORTH_SOUTH ? ni : 0,
                       side == MI_WEST_EAST ? nj : 0, xy1, xz1, yz1, xy2, xz2, yz2, tmp.as<float>()));
    MI_HIP(hipStreamSynchronize(s));  // tmp dies at scope exit
    return MI_OK;
}

extern "C" int mi_ncc_time_mips(int dev, void* stream, int n_pairs, const float* const* tiles, const int* a_idx, const int* b_idx, int dimk, int dimi,
                                int dimj, int ni, int nj, int side, int reps, float* ms_per_launch) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(n_pairs > 0 && tiles && a_idx && b_idx, "mi_ncc_time_mips: null pointer");
    std::vector<const float*> pa(n_pairs), pb(n_pairs);
    for (int q = 0; q < n_pairs; ++q) { pa[q] = tiles[a_idx[q]]; pb[q] = tiles[b_idx[q]]; }
    return ncc_time_mips(dev, as_stream(stream), n_pairs, pa.data(), pb.data(), dimk, dimi, dimj, ni, nj, side, reps, ms_per_launch);
}

static int time_mips_int(int bytes, int dev, void* stream, int n_pairs, const void* const* tiles, float scale, const int* a_idx, const int* b_idx,
                         int dimk, int dimi, int dimj, int ni, int nj, int side, int reps, float* ms_per_launch) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(n_pairs > 0 && tiles && a_idx && b_idx && scale > 0.0f, "mi_ncc_time_mips_u%d: invalid arguments", 8 * bytes);
    std::vector<const float*> pa(n_pairs), pb(n_pairs);
    for (int q = 0; q < n_pairs; ++q) {
        pa[q] = reinterpret_cast<const float*>(tiles[a_idx[q]]);
        pb[q] = reinterpret_cast<const float*>(tiles[b_idx[q]]);
    }
    TileFmt fmt;
    fmt.bytes = bytes;
    fmt.scale = scale;
    return ncc_time_mips(dev, as_stream(stream), n_pairs, pa.data(), pb.data(), dimk, dimi, dimj, ni, nj, side, reps, ms_per_launch, fmt);
}
extern "C" int mi_ncc_time_mips_u16(int dev, void* stream, int n_pairs, const unsigned short* const* tiles, float scale, const int* a_idx,
                                    const int* b_idx, int dimk, int dimi, int dimj, int ni, int nj, int side, int reps, float* ms_per_launch) {
    return time_mips_int(2, dev, stream, n_pairs, reinterpret_cast<const void* const*>(tiles), scale, a_idx, b_idx, dimk, dimi, dimj, ni, nj, side, reps,
                         ms_per_launch);
}
extern "C" int mi_ncc_time_mips_u8(int dev, void* stream, int n_pairs, const unsigned char* const* tiles, float scale, const int* a_idx,
                                   const int* b_idx, int dimk, int dimi, int dimj, int ni, int nj, int side, int reps, float* ms_per_launch) {
    return time_mips_int(1, dev, stream, n_pairs, reinterpret_cast<const void* const*>(tiles), scale, a_idx, b_idx, dimk, dimi, dimj, ni, nj, side, reps,
                         ms_per_launch);
}

extern "C" int mi_ncc_compute_map_lag(int dev, void* stream, const float* mip1, const float* mip2, int dimu, int dimv, int delayu, int delayv,
                                      float* map) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(mip1 && mip2 && map, "compute_NCC_map: null pointer");
    MI_REQUIRE(dimu > 0 && dimv > 0 && delayu >= 0 && delayv >= 0, "compute_NCC_map: invalid extents");
    return ncc_lag_map(dev, as_stream(stream), mip1, mip2, dimu, dimv, delayu, delayv, map);
}

extern "C" int mi_ncc_compute_map(int dev, void* stream, const float* mip1, const float* mip2, int dimu, int dimv, int delayu, int delayv,
                                  float* map) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(mip1 && mip2 && map, "compute_NCC_map: null pointer");
    MI_REQUIRE(dimu > 0 && dimv > 0 && delayu >= 0 && delayv >= 0, "compute_NCC_map: invalid extents");
    hipStream_t s = as_stream(stream);
    const int nt = (dimu / TILE) * (dimv / TILE);
    DevBuf ps, sat;
    MI_TRY(ps.alloc(sizeof(float) * 2 * (size_t)(nt > 0 ? nt : 1)));
    MI_TRY(sat.alloc(sizeof(double) * SatLayout(dimu, dimv).total));
    SatView v1, v2;
    MI_TRY(prepare_plane(s, mip1, mip2, dimu, dimv, ps.as<float>(), ps.as<float>() + (nt > 0 ? nt : 0), sat.as<double>(), &v1, &v2));
    DevBuf partial;
    MI_TRY(ncc_launch(s, mip1, mip2, dimu, dimv, delayu, delayv, v1, v2, nullptr, 0, nullptr, 0, partial, map));
    MI_HIP(hipStreamSynchronize(s));  // ps / sat / partial die at scope exit
    return MI_OK;
}
