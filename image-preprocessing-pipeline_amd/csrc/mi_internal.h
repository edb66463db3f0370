// Internal helpers shared by the libmi_ipp.so translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "mi_common.h"

namespace mi {

inline int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// thread-local error sink lives in common.hip
std::string& last_error_ref();

inline int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

#define MI_HIP(call)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return ::mi::fail(MI_ERR_HIP, "%s:%d: %s failed: %s", __FILE__, __LINE__, #call,             \
                              hipGetErrorString(e_));                                                    \
    } while (0)

#define MI_TRY(call)              \
    do {                          \
        int rc_ = (call);         \
        if (rc_ != MI_OK) return rc_; \
    } while (0)

#define MI_REQUIRE(cond, ...)                                          \
    do {                                                               \
        if (!(cond)) return ::mi::fail(MI_ERR_INVALID, __VA_ARGS__);   \
    } while (0)

// selects the device for the calling thread; every entry point starts with this
inline int use_device(int dev) {
    int n = 0;
    MI_HIP(hipGetDeviceCount(&n));
    MI_REQUIRE(dev >= 0 && dev < n, "device %d out of range (have %d)", dev, n);
    MI_HIP(hipSetDevice(dev));
    return MI_OK;
}

// Experiment switches (phase knock-outs, tile shapes, stream layouts: everything profiles/ varies) exist only in probe builds
// (`make EXTRA=-DMI_PROBES`): the product library does not contain their names and reads none of them.
#ifdef MI_PROBES
#define MI_PROBE_ENV(name) std::getenv(name)
#else
#define MI_PROBE_ENV(name) (static_cast<const char*>(nullptr))
#endif
// Host-time spans of a probe build: MI_SPAN_BEGIN(v, "label") ... MI_SPAN_END(v) adds the wall time between the two to the label's sum
// (all threads; common.hip keeps the table, mi_probe_host_spans prints it).  Nothing in the product build.
#ifdef MI_PROBES
double probe_now();
void probe_span_add(const char* label, double seconds);
#define MI_SPAN_BEGIN(v, label) const char* v##_label = (label); const double v##_t0 = ::mi::probe_now()
#define MI_SPAN_END(v) ::mi::probe_span_add(v##_label, ::mi::probe_now() - v##_t0)
#else
#define MI_SPAN_BEGIN(v, label) do { } while (0)
#define MI_SPAN_END(v) do { } while (0)
#endif

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int launch_check(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MI_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return MI_OK;
}

constexpr float kEpsSingle = 1.1920928955078125e-07f;  // eps('single') = 2^-23 (decon.m:62)

inline unsigned cdiv(size_t a, size_t b) { return static_cast<unsigned>((a + b - 1) / b); }

// Device memory of the library goes through a small caching pool (common.hip): blocks are kept per device and size when they are
// released and handed out again on the next request of that size -- a pipeline stage that builds its buffers per call (edge
// taper, RL context of a block) otherwise pays hipMalloc/hipFree of tens of GB each time (measured: 4 s per switch between two
// 30-GB working sets against 0.07 s of work).  Everything cached is returned to the driver when an allocation fails, and by
// mi_release_cached_memory().
int pool_alloc(size_t n, void** out);
void pool_free(void* p, size_t n);

// RAII device buffer for library-owned scratch
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    int alloc(size_t n) {
        release();
        if (n == 0) return MI_OK;
        MI_TRY(pool_alloc(n, &p));
        bytes = n;
        return MI_OK;
    }
    void release() {
        if (p) pool_free(p, bytes);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// ncc.hip: destroys the pair working sets kept between mi_ncc_mips_batch calls (dev < 0: of every device)
void ncc_drop_cached_slots(int dev);

}  // namespace mi
