// Post-deconvolution statistics and output conversion of a block:
//   mi_prctile       [lb, ub] = deconvolved_stats(bl, clipval) = prctile(bl, [100-clip clip], "all")   (LsDeconv.m:1300-1307)
//   mi_rescale_block the rescale / round / clamp / convert of a float brick into the uint8 / uint16 slab
//                    (load_slab_lz4.cpp:134-157)
// Both are single-pass streaming kernels (HBM-bound); the percentile is an exact order statistic found by a three-level radix
// select over the monotone integer image of the floats (11 + 11 + 10 bits, one histogram pass over the volume per level), so no
// sorted copy, no sub-sampling and no D2H of the fp32 volume is needed.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "mi_internal.h"

namespace mi {
namespace {

constexpr int kStatThreads = 256;
inline unsigned stream_grid(size_t n_items) {
    size_t b = (n_items + kStatThreads - 1) / kStatThreads;
    const size_t cap = 256 * 16;  // 16 work-groups per CU, grid-stride beyond that
    return static_cast<unsigned>(b < 1 ? 1 : (b > cap ? cap : b));
}
constexpr int kMaxTargets = 4;  // two percentiles x the two neighbouring order statistics
constexpr int kBins = 2048;

// order-preserving map float -> uint32 (negative floats reversed, positive floats above them); NaN is handled by the caller
__device__ __forceinline__ unsigned float_key(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline float key_float(unsigned k) {
    const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

struct HistArgs {
    int n_targets;               // distinct prefixes at this level
    int shift;                   // bin = (key >> shift) & (bins - 1)
    int bins;
    int prefix_shift;            // an element belongs to target t iff (key >> prefix_shift) == prefix[t]  (32: everything)
    unsigned prefix[kMaxTargets];
};

// hist[t][bin] += 1 for every finite-or-infinite (non-NaN) element whose key has prefix t; nan_count += #NaN (level 0 only)
__global__ __launch_bounds__(kStatThreads) void k_radix_hist(const float* __restrict__ x, size_t n, HistArgs a,
                                                              unsigned long long* __restrict__ hist,
                                                              unsigned long long* __restrict__ nan_count) {
    extern __shared__ unsigned lh[];  // n_targets * bins
    const int total = a.n_targets * a.bins;
    for (int i = threadIdx.x; i < total; i += kStatThreads) lh[i] = 0u;
    __syncthreads();
    unsigned nans = 0;
    auto one = [&](float v) {
        if (v != v) { ++nans; return; }
        const unsigned k = float_key(v);
        const unsigned p = a.prefix_shift >= 32 ? 0u : (k >> a.prefix_shift);
        const unsigned b = (k >> a.shift) & (unsigned)(a.bins - 1);
#pragma unroll
        for (int t = 0; t < kMaxTargets; ++t)
            if (t < a.n_targets && p == a.prefix[t]) atomicAdd(&lh[t * a.bins + b], 1u);
    };
    const size_t n4 = n / 4;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const size_t tid = blockIdx.x * (size_t)kStatThreads + threadIdx.x, stride = (size_t)gridDim.x * kStatThreads;
    for (size_t i = tid; i < n4; i += stride) {
        const float4 v = x4[i];
        one(v.x); one(v.y); one(v.z); one(v.w);
    }
    for (size_t i = n4 * 4 + tid; i < n; i += stride) one(x[i]);
    __syncthreads();
    for (int i = threadIdx.x; i < total; i += kStatThreads)
        if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
    if (nan_count && nans) atomicAdd(nan_count, (unsigned long long)nans);
}

// val = (dmin > 0) ? (val - dmin) * scal*ampl/(dmax-dmin) : val * scal*ampl/dmax;  val -= ampl;
// round half away from zero; clamp to [0, scal]; convert            (load_slab_lz4.cpp:134-157, float arithmetic in this order)
template <typename OUT>
__global__ __launch_bounds__(kStatThreads) void k_rescale(const float* __restrict__ src, OUT* __restrict__ dst, size_t n, float scal,
                                                           float ampl, float dmin, float k_linear, float k_minmax, int use_minmax) {
    const size_t tid = blockIdx.x * (size_t)kStatThreads + threadIdx.x, stride = (size_t)gridDim.x * kStatThreads;
    auto conv = [&](float val) {
#pragma clang fp contract(off)
        if (use_minmax) val = (val - dmin) * k_minmax;
        else val = val * k_linear;
        val -= ampl;
        val = (val >= 0.f) ? floorf(val + 0.5f) : ceilf(val - 0.5f);
        val = fminf(fmaxf(val, 0.f), scal);  // std::clamp(val, 0, scal); a NaN stays NaN there and converts to 0 here
        return (OUT)val;
    };
    const size_t n4 = n / 4;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    for (size_t i = tid; i < n4; i += stride) {
        const float4 v = s4[i];
        dst[4 * i + 0] = conv(v.x);
        dst[4 * i + 1] = conv(v.y);
        dst[4 * i + 2] = conv(v.z);
        dst[4 * i + 3] = conv(v.w);
    }
    for (size_t i = n4 * 4 + tid; i < n; i += stride) dst[i] = conv(src[i]);
}

}  // namespace
}  // namespace mi

using namespace mi;

extern "C" int mi_prctile(int dev, void* stream, const float* x, size_t n, const double* pct, int n_pct, float* out) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(x && pct && out, "mi_prctile: null pointer");
    MI_REQUIRE(n > 0, "mi_prctile: empty input");
    MI_REQUIRE(n_pct >= 1 && n_pct <= 2, "mi_prctile: 1 or 2 percentiles per call");
    MI_REQUIRE(((uintptr_t)x % 16) == 0, "mi_prctile: pointer must be 16-byte aligned");
    for (int i = 0; i < n_pct; ++i) MI_REQUIRE(pct[i] >= 0.0 && pct[i] <= 100.0, "prctile: percentiles must be in [0, 100]");
    hipStream_t s = as_stream(stream);
    DevBuf d;
    const size_t hist_words = (size_t)kMaxTargets * kBins + 1;
    MI_TRY(d.alloc(sizeof(unsigned long long) * hist_words));
    std::vector<unsigned long long> h(hist_words);
    const unsigned grid = stream_grid(n / 4 + 1);

    // the order statistics still to resolve: key prefix found so far, rank inside that prefix
    struct Target { unsigned prefix; unsigned long long rank; };
    std::vector<Target> tg;
    unsigned long long n_valid = n;
    std::vector<unsigned long long> want_rank;  // 0-based ranks, two per percentile (lower, upper neighbour)
    const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
    for (int level = 0; level < 3; ++level) {
        HistArgs a{};
        a.shift = shifts[level];
        a.bins = 1 << widths[level];
        a.prefix_shift = level == 0 ? 32 : shifts[level - 1];
        std::vector<unsigned> prefixes;
        if (level == 0) {
            prefixes.push_back(0u);
        } else {
            for (const Target& t : tg)
                if (std::find(prefixes.begin(), prefixes.end(), t.prefix) == prefixes.end()) prefixes.push_back(t.prefix);
        }
        a.n_targets = (int)prefixes.size();
        for (int t = 0; t < a.n_targets; ++t) a.prefix[t] = prefixes[t];
        MI_HIP(hipMemsetAsync(d.p, 0, sizeof(unsigned long long) * hist_words, s));
        unsigned long long* dh = d.as<unsigned long long>();
        hipLaunchKernelGGL(k_radix_hist, dim3(grid), dim3(kStatThreads), sizeof(unsigned) * a.n_targets * a.bins, s, x, n, a, dh,
                           level == 0 ? dh + (size_t)kMaxTargets * kBins : nullptr);
        MI_TRY(launch_check("k_radix_hist"));
        MI_HIP(hipMemcpyAsync(h.data(), d.p, sizeof(unsigned long long) * hist_words, hipMemcpyDeviceToHost, s));
        MI_HIP(hipStreamSynchronize(s));
        if (level == 0) {
            // prctile ignores NaN (treated as missing); positions follow MATLAB's definition: the sorted sample i (1-based)
            // is the 100 (i - 0.5) / n percentile, linear in between, clamped to the extremes
            n_valid = n - h[(size_t)kMaxTargets * kBins];
            if (n_valid == 0) {
                for (int i = 0; i < n_pct; ++i) out[i] = NAN;
                return MI_OK;
            }
            for (int i = 0; i < n_pct; ++i) {
                double pos = pct[i] / 100.0 * (double)n_valid - 0.5;
                pos = std::min(std::max(pos, 0.0), (double)(n_valid - 1));
                const unsigned long long lo = (unsigned long long)std::floor(pos);
                want_rank.push_back(lo);
                want_rank.push_back(std::min(lo + 1, n_valid - 1));
            }
            for (unsigned long long r : want_rank) tg.push_back(Target{0u, r});
        }
        for (Target& t : tg) {
            const int slot = (int)(std::find(prefixes.begin(), prefixes.end(), t.prefix) - prefixes.begin());
            const unsigned long long* hh = h.data() + (size_t)slot * a.bins;
            unsigned long long acc = 0;
            int b = 0;
            for (; b < a.bins; ++b) {
                if (t.rank < acc + hh[b]) break;
                acc += hh[b];
            }
            MI_REQUIRE(b < a.bins, "mi_prctile: internal rank bookkeeping failed");
            t.rank -= acc;
            t.prefix = (t.prefix << widths[level]) | (unsigned)b;
        }
    }
    for (int i = 0; i < n_pct; ++i) {
        double pos = pct[i] / 100.0 * (double)n_valid - 0.5;
        pos = std::min(std::max(pos, 0.0), (double)(n_valid - 1));
        const double frac = pos - std::floor(pos);
        const double lo = (double)key_float(tg[2 * i].prefix), hi = (double)key_float(tg[2 * i + 1].prefix);
        out[i] = (float)(lo + frac * (hi - lo));
    }
    return MI_OK;
}

extern "C" int mi_rescale_block(int dev, void* stream, const float* src, void* dst, size_t n, int out_bits, float scal, float ampl,
                                float dmin, float dmax) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(src && dst, "mi_rescale_block: null pointer");
    MI_REQUIRE(out_bits == 8 || out_bits == 16, "mi_rescale_block: output must be uint8 or uint16");
    MI_REQUIRE(((uintptr_t)src % 16) == 0, "mi_rescale_block: source must be 16-byte aligned");
    MI_REQUIRE(scal > 0.f && scal <= (out_bits == 8 ? 255.f : 65535.f), "mi_rescale_block: scal outside the output type's range");
    if (n == 0) return MI_OK;
    // the two factors exactly as load_slab_lz4.cpp:134-136 forms them (float arithmetic, left to right)
    const float k_linear = scal * ampl / dmax;
    const float k_minmax = (dmin > 0.f) ? scal * ampl / (dmax - dmin) : k_linear;
    const int use_minmax = dmin > 0.f;
    hipStream_t s = as_stream(stream);
    const unsigned grid = stream_grid(n / 4 + 1);
    if (out_bits == 8)
        hipLaunchKernelGGL(k_rescale<unsigned char>, dim3(grid), dim3(kStatThreads), 0, s, src, (unsigned char*)dst, n, scal, ampl, dmin,
                           k_linear, k_minmax, use_minmax);
    else
        hipLaunchKernelGGL(k_rescale<unsigned short>, dim3(grid), dim3(kStatThreads), 0, s, src, (unsigned short*)dst, n, scal, ampl,
                           dmin, k_linear, k_minmax, use_minmax);
    return launch_check("k_rescale");
}
