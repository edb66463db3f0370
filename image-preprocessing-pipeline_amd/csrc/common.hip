// Error sink, device queries and the element-wise / data-movement kernels of the RL path.
// All of these are pure HBM streaming kernels: 16 B per lane, grid-stride, >= 4 waves per SIMD.
#include <algorithm>
#include <map>
#include <mutex>

#include <cstdlib>

#include "mi_internal.h"
#include "mi_lsdeconv.h"

namespace mi {
std::string& last_error_ref() {
    thread_local std::string err;
    return err;
}
}  // namespace mi

using namespace mi;

extern "C" const char* mi_last_error(void) { return last_error_ref().c_str(); }

extern "C" int mi_abi_version(void) { return 1; }

extern "C" int mi_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(MI_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

extern "C" int mi_stream_synchronize(int dev, void* stream) {
    MI_TRY(use_device(dev));
    MI_HIP(hipStreamSynchronize(as_stream(stream)));
    return MI_OK;
}

extern "C" int mi_next_fast_len(int n) {
    if (n < 1) n = 1;
    for (;; ++n) {
        int m = n;
        for (int p : {2, 3, 5, 7})
            while (m % p == 0) m /= p;
        if (m == 1) return n;
    }
}

// ------------------------------------------------------------------------------------------------ device memory pool
namespace mi {
namespace {
struct Pool {
    std::mutex mu;
    std::multimap<std::pair<int, size_t>, void*> free_blocks;  // (device, bytes) -> block
    size_t cached_bytes = 0;
    bool enabled = std::getenv("MI_NO_MEMORY_POOL") == nullptr;
    // every cached block of one device (or of all, dev < 0) goes back to the driver; returns the bytes released
    size_t trim(int dev) {
        size_t freed = 0;
        for (auto it = free_blocks.begin(); it != free_blocks.end();) {
            if (dev < 0 || it->first.first == dev) {
                (void)hipFree(it->second);
                freed += it->first.second;
                it = free_blocks.erase(it);
            } else {
                ++it;
            }
        }
        cached_bytes -= freed;
        return freed;
    }
};
Pool& pool() {
    static Pool* p = new Pool;  // never destroyed: blocks may be released during static destruction of other objects
    return *p;
}
}  // namespace

int pool_alloc(size_t n, void** out) {
    int dev = 0;
    MI_HIP(hipGetDevice(&dev));
    Pool& P = pool();
    if (P.enabled) {
        std::lock_guard<std::mutex> g(P.mu);
        auto it = P.free_blocks.find({dev, n});
        if (it != P.free_blocks.end()) {
            *out = it->second;
            P.cached_bytes -= n;
            P.free_blocks.erase(it);
            return MI_OK;
        }
    }
    hipError_t e = hipMalloc(out, n);
    if (e != hipSuccess && P.enabled) {  // give the cached blocks back and try once more
        (void)hipGetLastError();
        std::lock_guard<std::mutex> g(P.mu);
        if (P.trim(dev) > 0) e = hipMalloc(out, n);
    }
    if (e != hipSuccess) {
        *out = nullptr;
        (void)hipGetLastError();
        return fail(MI_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", n, hipGetErrorString(e));
    }
    return MI_OK;
}

void pool_free(void* p, size_t n) {
    if (!p) return;
    Pool& P = pool();
    if (!P.enabled || n < (size_t)(1 << 20)) {  // small blocks are cheap to allocate: straight back to the driver
        (void)hipFree(p);
        return;
    }
    {   // so do blocks of more than an eighth of the device: the pool exists for the working sets of decwrap's blocks (tens of GB in
        // pieces of a few GB), and a 69-GB array kept here -- the complex OTF of BASELINE config 4 on one device, released once its
        // real form exists -- is memory the caller's own allocator never gets to see (its failures do not trim this pool)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && n > total_b / 8) {
            (void)hipFree(p);
            return;
        }
    }
    // hipFree waits for the device before it releases memory; a cached block may be handed out to another stream, so the same
    // guarantee is kept here
    int cur = 0, dev = 0;
    (void)hipGetDevice(&cur);
    dev = cur;
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, p) == hipSuccess) dev = attr.device;
    if (dev != cur) (void)hipSetDevice(dev);
    MI_SPAN_BEGIN(spw, "pool_free: device-wide wait");
    (void)hipDeviceSynchronize();
    MI_SPAN_END(spw);
    if (dev != cur) (void)hipSetDevice(cur);
    std::lock_guard<std::mutex> g(P.mu);
    P.free_blocks.insert({{dev, n}, p});
    P.cached_bytes += n;
}

}  // namespace mi

extern "C" size_t mi_release_cached_memory(int dev) {
    mi::ncc_drop_cached_slots(dev);  // their buffers go back to the pool first
    mi::Pool& P = mi::pool();
    std::lock_guard<std::mutex> g(P.mu);
    return P.trim(dev);
}

extern "C" size_t mi_cached_memory_bytes(void) {
    mi::Pool& P = mi::pool();
    std::lock_guard<std::mutex> g(P.mu);
    return P.cached_bytes;
}

// ------------------------------------------------------------------------------------------------
namespace {

constexpr int kThreads = 256;
inline unsigned stream_grid(size_t n_items) {
    size_t b = (n_items + kThreads - 1) / kThreads;
    const size_t cap = 256 * 16;  // 16 work-groups per CU, grid-stride beyond that
    return static_cast<unsigned>(b < 1 ? 1 : (b > cap ? cap : b));
}

// uint16 -> float * scale (im2single). 8 elements (16 B in, 32 B out) per lane per step.
__global__ __launch_bounds__(kThreads) void k_u16_to_f32(const uint16_t* __restrict__ src, float* __restrict__ dst,
                                                          size_t n, float scale) {
    size_t n8 = n / 8;
    size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (size_t i = tid; i < n8; i += stride) {
        uint4 v = s4[i];
        float4 a, b;
        a.x = (float)(v.x & 0xffffu) * scale; a.y = (float)(v.x >> 16) * scale;
        a.z = (float)(v.y & 0xffffu) * scale; a.w = (float)(v.y >> 16) * scale;
        b.x = (float)(v.z & 0xffffu) * scale; b.y = (float)(v.z >> 16) * scale;
        b.z = (float)(v.w & 0xffffu) * scale; b.w = (float)(v.w >> 16) * scale;
        d4[2 * i] = a;
        d4[2 * i + 1] = b;
    }
    for (size_t i = n8 * 8 + tid; i < n; i += stride) dst[i] = (float)src[i] * scale;
}

__global__ __launch_bounds__(kThreads) void k_sub_dark(const float* __restrict__ src, float* __restrict__ dst, size_t n,
                                                        float dark) {
    size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < n; i += stride) dst[i] = fmaxf(src[i] - dark, 0.0f);
}

// sum of squares in double: per-thread double accumulators, wave shuffle tree, one atomicAdd(double)
// per work-group.  (fp64 atomics on distinct groups commute up to rounding; the stop test compares
// against a percentage threshold, decon.m:110-115.)
__global__ __launch_bounds__(kThreads) void k_sumsq(const float* __restrict__ x, size_t n, double* __restrict__ out) {
    size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    size_t n4 = n / 4;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (size_t i = tid; i < n4; i += stride) {
        float4 v = x4[i];
        acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    for (size_t i = n4 * 4 + tid; i < n; i += stride) acc += (double)x[i] * x[i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ double part[kThreads / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < kThreads / 64; ++w) s += part[w];
        atomicAdd(out, s);
    }
}

// dst (fx,fy,fz) = zero-pad-centre(src (nx,ny,nz)) or the crop back; one thread per 1 dst element
// (pad / crop: a work-group row per (y, z) row of the LARGER grid, four samples per lane; 16-byte accesses where both rows allow --
//  the first forms divided a 64-bit index twice per sample)
__global__ __launch_bounds__(kThreads) void k_pad_center(const float* __restrict__ src, int nx, int ny, int nz,
                                                          float* __restrict__ dst, int fx, int fy, int fz, int px, int py,
                                                          int pz) {
    const int y = blockIdx.y, z = blockIdx.z, sy = y - py, sz = z - pz;
    const bool live = sy >= 0 && sy < ny && sz >= 0 && sz < nz;   // (scalar: a row of zeros otherwise)
    const float* srow = src + ((size_t)(live ? sz : 0) * ny + (live ? sy : 0)) * nx;
    float* drow = dst + ((size_t)z * fy + y) * fx;
    const bool vec = (fx & 3) == 0 && ((uintptr_t)dst & 15) == 0;
    for (int x0 = 4 * (blockIdx.x * kThreads + threadIdx.x); x0 < fx; x0 += 4 * kThreads * gridDim.x) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int sx = x0 + e - px;
            const bool on = live && sx >= 0 && sx < nx;
            const float s = srow[on ? sx : 0];
            v[e] = on ? s : 0.0f;
        }
        if (vec) {
            *reinterpret_cast<float4*>(drow + x0) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (x0 + e < fx) drow[x0 + e] = v[e];
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_crop_center(const float* __restrict__ src, int fx, int fy, int fz,
                                                           float* __restrict__ dst, int nx, int ny, int nz, int px, int py,
                                                           int pz) {
    const int y = blockIdx.y, z = blockIdx.z;
    const float* srow = src + ((size_t)(z + pz) * fy + (y + py)) * fx + px;
    float* drow = dst + ((size_t)z * ny + y) * nx;
    const bool vec = (nx & 3) == 0 && ((uintptr_t)dst & 15) == 0;
    for (int x0 = 4 * (blockIdx.x * kThreads + threadIdx.x); x0 < nx; x0 += 4 * kThreads * gridDim.x) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = srow[min(x0 + e, nx - 1)];
        if (vec) {
            *reinterpret_cast<float4*>(drow + x0) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (x0 + e < nx) drow[x0 + e] = v[e];
        }
    }
}

// rows [y0, y0+rows) of every plane <-> packed (nz, rows, nx)
template <bool PACK>
__global__ __launch_bounds__(kThreads) void k_rows(float* __restrict__ vol, int nx, int ny, int nz, int y0, int rows,
                                                    float* __restrict__ packed) {
    size_t total = (size_t)nx * rows * nz;
    size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < total; i += stride) {
        int x = (int)(i % nx);
        size_t r = i / nx;
        int y = (int)(r % rows), z = (int)(r / rows);
        size_t v = ((size_t)z * ny + (y0 + y)) * nx + x;
        if (PACK) packed[i] = vol[v];
        else vol[v] = packed[i];
    }
}

// reg = convn(bl, R, 'same'), R = 1/26 on the 26 neighbours, 0 at the centre (decon.m:42,70).
// Accumulation order follows the kernel index order of a true convolution; zero boundary.
__global__ __launch_bounds__(kThreads) void k_reg_term(const float* __restrict__ bl, float* __restrict__ reg, int nx, int ny,
                                                        int nz) {
    size_t total = (size_t)nx * ny * nz;
    size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    const float w = 1.0f / 26.0f;
    for (size_t i = tid; i < total; i += stride) {
        int x = (int)(i % nx);
        size_t r = i / nx;
        int y = (int)(r % ny), z = (int)(r / ny);
        float acc = 0.0f;
        for (int dz = -1; dz <= 1; ++dz) {
            int zz = z + dz;
            if (zz < 0 || zz >= nz) continue;
            for (int dy = -1; dy <= 1; ++dy) {
                int yy = y + dy;
                if (yy < 0 || yy >= ny) continue;
                for (int dx = -1; dx <= 1; ++dx) {
                    int xx = x + dx;
                    if (xx < 0 || xx >= nx || (dx == 0 && dy == 0 && dz == 0)) continue;
                    acc += w * bl[((size_t)zz * ny + yy) * nx + xx];
                }
            }
        }
        reg[i] = acc;
    }
}

}  // namespace

extern "C" int mi_u16_to_f32(int dev, void* stream, const uint16_t* src, float* dst, size_t n, float scale) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(src && dst, "mi_u16_to_f32: null pointer");
    MI_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "mi_u16_to_f32: pointers must be 16-byte aligned");
    if (n == 0) return MI_OK;
    hipLaunchKernelGGL(k_u16_to_f32, dim3(stream_grid(n / 8 + 1)), dim3(kThreads), 0, as_stream(stream), src, dst, n, scale);
    return launch_check("k_u16_to_f32");
}

namespace {
// load_block (LsDeconv.m:817-904) on the device: dst (the padded block) <- the sub-box read from the volume, converted like
// im2single (integer types: value / max of the type, a float32 division) and extended by padarray(..., 'symmetric')
// (edge-inclusive mirror, repeated when the pad exceeds the box) where the padded block reaches beyond the volume
// One work-group row per (y, z) row of the padded block -- its source row is found once, by the scalar unit -- and four samples per
// lane along x (the first form divided a 64-bit index twice per sample: 1.04 ms for a 512 x 512 x 959 block, 1.5 GB of traffic).
template <typename T>
__global__ __launch_bounds__(kThreads) void k_load_block(const T* __restrict__ src, int sx, int sy, int sz, float* __restrict__ dst, int nx,
                                                          int ny, int nz, int bx, int by, int bz, float maxv) {
    auto mirror = [](int j, int n) {
        const int p = 2 * n;
        j %= p;
        if (j < 0) j += p;
        return j < n ? j : p - 1 - j;
    };
    const int y = blockIdx.y, z = blockIdx.z;
    const T* srow = src + ((size_t)mirror(z - bz, sz) * sy + mirror(y - by, sy)) * sx;
    float* drow = dst + ((size_t)z * ny + y) * nx;
    const bool vec = (nx & 3) == 0 && ((uintptr_t)dst & 15) == 0;   // (every row of the padded block then starts on a 16-byte boundary)
    for (int x0 = 4 * (blockIdx.x * kThreads + threadIdx.x); x0 < nx; x0 += 4 * kThreads * gridDim.x) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int x = min(x0 + e, nx - 1), j = x - bx;
            // (inside the box -- nearly every sample -- the mirror is the identity: no division)
            const float s = (float)srow[(unsigned)j < (unsigned)sx ? j : mirror(j, sx)];
            v[e] = maxv > 0.0f ? s / maxv : s;
        }
        if (vec) {
            *reinterpret_cast<float4*>(drow + x0) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (x0 + e < nx) drow[x0 + e] = v[e];
        }
    }
}
}  // namespace

extern "C" int mi_load_block(int dev, void* stream, const void* src, int dtype, int sx, int sy, int sz, float* dst, int nx, int ny, int nz,
                             int bx, int by, int bz) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(src && dst, "load_block: null pointer");
    MI_REQUIRE(sx > 0 && sy > 0 && sz > 0 && nx > 0 && ny > 0 && nz > 0, "load_block: empty box");
    MI_REQUIRE(bx >= 0 && by >= 0 && bz >= 0 && bx + sx <= nx && by + sy <= ny && bz + sz <= nz,
               "load_block: the box read from the volume must lie inside the padded block");
    MI_REQUIRE(ny <= 65535 && nz <= 65535, "load_block: block of %d x %d rows (at most 65535 per axis)", ny, nz);
    const dim3 grid((unsigned)std::min(64, (nx + 4 * kThreads - 1) / (4 * kThreads)), (unsigned)ny, (unsigned)nz), block(kThreads);
    hipStream_t s = as_stream(stream);
    switch (dtype) {
        case 1: hipLaunchKernelGGL(k_load_block<uint8_t>, grid, block, 0, s, (const uint8_t*)src, sx, sy, sz, dst, nx, ny, nz, bx, by, bz, 255.0f); break;
        case 2: hipLaunchKernelGGL(k_load_block<uint16_t>, grid, block, 0, s, (const uint16_t*)src, sx, sy, sz, dst, nx, ny, nz, bx, by, bz, 65535.0f); break;
        case 4: hipLaunchKernelGGL(k_load_block<float>, grid, block, 0, s, (const float*)src, sx, sy, sz, dst, nx, ny, nz, bx, by, bz, 0.0f); break;
        default: return fail(MI_ERR_INVALID, "load_block: dtype code %d (1 = uint8, 2 = uint16, 4 = float32)", dtype);
    }
    return launch_check("k_load_block");
}

extern "C" int mi_subtract_dark(int dev, void* stream, const float* src, float* dst, size_t n, float dark) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(src && dst, "mi_subtract_dark: null pointer");
    if (n == 0) return MI_OK;
    hipLaunchKernelGGL(k_sub_dark, dim3(stream_grid(n)), dim3(kThreads), 0, as_stream(stream), src, dst, n, dark);
    return launch_check("k_sub_dark");
}

namespace mi {
// enqueue sum(x^2) into *d_out (device double, zeroed here)
int sumsq_async(hipStream_t s, const float* x, size_t n, double* d_out) {
    MI_HIP(hipMemsetAsync(d_out, 0, sizeof(double), s));
    if (n) {
        hipLaunchKernelGGL(k_sumsq, dim3(stream_grid(n / 4 + 1)), dim3(kThreads), 0, s, x, n, d_out);
        MI_TRY(launch_check("k_sumsq"));
    }
    return MI_OK;
}
}  // namespace mi

extern "C" int mi_norm2(int dev, void* stream, const float* x, size_t n, double* norm2) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(x && norm2, "mi_norm2: null pointer");
    MI_REQUIRE(((uintptr_t)x % 16) == 0, "mi_norm2: pointer must be 16-byte aligned");
    DevBuf d;
    MI_TRY(d.alloc(sizeof(double)));
    MI_TRY(sumsq_async(as_stream(stream), x, n, d.as<double>()));
    double h = 0.0;
    MI_HIP(hipMemcpyAsync(&h, d.p, sizeof(double), hipMemcpyDeviceToHost, as_stream(stream)));
    MI_HIP(hipStreamSynchronize(as_stream(stream)));
    *norm2 = sqrt(h);
    return MI_OK;
}

extern "C" int mi_pad_center(int dev, void* stream, const float* src, int nx, int ny, int nz, float* dst, int fx, int fy,
                             int fz) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(src && dst, "mi_pad_center: null pointer");
    MI_REQUIRE(nx > 0 && ny > 0 && nz > 0 && fx >= nx && fy >= ny && fz >= nz,
               "pad_block_to_fft_shape: bl [%d %d %d] is larger than FFT shape [%d %d %d], cannot pad", nx, ny, nz, fx, fy, fz);
    MI_REQUIRE(fy <= 65535 && fz <= 65535, "pad_block_to_fft_shape: FFT shape of %d x %d rows (at most 65535 per axis)", fy, fz);
    hipLaunchKernelGGL(k_pad_center, dim3((unsigned)std::min(64, (fx + 4 * kThreads - 1) / (4 * kThreads)), (unsigned)fy, (unsigned)fz), dim3(kThreads), 0,
                       as_stream(stream), src, nx, ny, nz, dst, fx, fy, fz, (fx - nx) / 2, (fy - ny) / 2, (fz - nz) / 2);
    return launch_check("k_pad_center");
}

extern "C" int mi_crop_center(int dev, void* stream, const float* src, int fx, int fy, int fz, float* dst, int nx, int ny,
                              int nz) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(src && dst, "mi_crop_center: null pointer");
    MI_REQUIRE(nx > 0 && ny > 0 && nz > 0 && fx >= nx && fy >= ny && fz >= nz, "unpad_block: invalid sizes");
    MI_REQUIRE(ny <= 65535 && nz <= 65535, "unpad_block: block of %d x %d rows (at most 65535 per axis)", ny, nz);
    hipLaunchKernelGGL(k_crop_center, dim3((unsigned)std::min(64, (nx + 4 * kThreads - 1) / (4 * kThreads)), (unsigned)ny, (unsigned)nz), dim3(kThreads), 0,
                       as_stream(stream), src, fx, fy, fz, dst, nx, ny, nz, (fx - nx) / 2, (fy - ny) / 2, (fz - nz) / 2);
    return launch_check("k_crop_center");
}

extern "C" int mi_pack_rows(int dev, void* stream, const float* vol, int nx, int ny, int nz, int y0, int rows, float* packed) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(vol && packed, "mi_pack_rows: null pointer");
    MI_REQUIRE(nx > 0 && ny > 0 && nz > 0 && rows >= 0 && y0 >= 0 && y0 + rows <= ny, "mi_pack_rows: rows [%d,%d) outside [0,%d)", y0,
               y0 + rows, ny);
    if (rows == 0) return MI_OK;
    hipLaunchKernelGGL(k_rows<true>, dim3(stream_grid((size_t)nx * rows * nz)), dim3(kThreads), 0, as_stream(stream),
                       const_cast<float*>(vol), nx, ny, nz, y0, rows, packed);
    return launch_check("k_rows<pack>");
}

extern "C" int mi_unpack_rows(int dev, void* stream, const float* packed, int nx, int ny, int nz, int y0, int rows, float* vol) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(vol && packed, "mi_unpack_rows: null pointer");
    MI_REQUIRE(nx > 0 && ny > 0 && nz > 0 && rows >= 0 && y0 >= 0 && y0 + rows <= ny, "mi_unpack_rows: rows [%d,%d) outside [0,%d)",
               y0, y0 + rows, ny);
    if (rows == 0) return MI_OK;
    hipLaunchKernelGGL(k_rows<false>, dim3(stream_grid((size_t)nx * rows * nz)), dim3(kThreads), 0, as_stream(stream), vol, nx, ny, nz,
                       y0, rows, const_cast<float*>(packed));
    return launch_check("k_rows<unpack>");
}

extern "C" int mi_rl_reg_term(int dev, void* stream, const float* bl, float* reg, int nx, int ny, int nz) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(bl && reg && bl != reg, "mi_rl_reg_term: null or aliased pointers");
    MI_REQUIRE(nx > 0 && ny > 0 && nz > 0, "mi_rl_reg_term: empty volume");
    hipLaunchKernelGGL(k_reg_term, dim3(stream_grid((size_t)nx * ny * nz)), dim3(kThreads), 0, as_stream(stream), bl, reg, nx, ny, nz);
    return launch_check("k_reg_term");
}

// ------------------------------------------------------------------------------------------------ host-time spans (probe builds only)
#ifdef MI_PROBES
#include <chrono>
namespace mi {
namespace {
struct SpanTable {
    std::mutex m;
    std::map<std::string, std::pair<double, long>> t;
};
SpanTable& span_table() {
    static SpanTable* t = new SpanTable;  // (never destroyed: worker threads may still add at exit)
    return *t;
}
}  // namespace
double probe_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void probe_span_add(const char* label, double seconds) {
    SpanTable& T = span_table();
    std::lock_guard<std::mutex> g(T.m);
    auto& e = T.t[label];
    e.first += seconds;
    e.second += 1;
}
}  // namespace mi
// the table as text ("label calls seconds\n" per line), cleared by the call; returns the number of bytes written
extern "C" int mi_probe_host_spans(char* buf, int cap) {
    mi::SpanTable& T = mi::span_table();
    std::lock_guard<std::mutex> g(T.m);
    int n = 0;
    for (auto& kv : T.t) {
        if (n >= cap) break;
        n += std::snprintf(buf + n, (size_t)(cap - n), "%-40s %7ld %10.4f\n", kv.first.c_str(), kv.second.second, kv.second.first);
    }
    T.t.clear();
    return n < cap ? n : cap;
}
#endif
