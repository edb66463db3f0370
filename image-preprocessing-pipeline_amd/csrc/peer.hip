// Copy-engine transport for the halo exchange of the slab driver (one process per GPU): device buffers and events shared between
// processes through HIP IPC handles, hipMemcpyPeerAsync on a stream of its own.  No reference counterpart: the reference's blocks
// never exchange anything (LsDeconv.m:643-654).  The RCCL route (grouped ncclSend/ncclRecv) runs kernels that need compute units
// next to the persistent x pass; a peer copy is executed by the SDMA engines and needs none.
#include <cstring>

#include "mi_internal.h"
#include "mi_lsdeconv.h"

using namespace mi;

static_assert(sizeof(hipIpcMemHandle_t) <= MI_IPC_HANDLE_BYTES && sizeof(hipIpcEventHandle_t) <= MI_IPC_HANDLE_BYTES,
              "IPC handles must fit MI_IPC_HANDLE_BYTES");

extern "C" int mi_peer_alloc(int dev, size_t bytes, void** ptr, unsigned char* handle) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(ptr && handle && bytes > 0, "mi_peer_alloc: null pointer or zero size");
    *ptr = nullptr;
    // (a plain hipMalloc of its own: IPC handles name whole allocations, never blocks of the library's pool)
    hipError_t e = hipMalloc(ptr, bytes);
    if (e != hipSuccess) return fail(MI_ERR_NOMEM, "mi_peer_alloc: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, *ptr);
    if (e != hipSuccess) {
        (void)hipFree(*ptr);
        *ptr = nullptr;
        return fail(MI_ERR_HIP, "mi_peer_alloc: hipIpcGetMemHandle failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set on this pool)",
                    hipGetErrorString(e));
    }
    std::memset(handle, 0, MI_IPC_HANDLE_BYTES);
    std::memcpy(handle, &h, sizeof h);
    return MI_OK;
}

extern "C" int mi_peer_free(int dev, void* ptr) {
    MI_TRY(use_device(dev));
    if (ptr) MI_HIP(hipFree(ptr));
    return MI_OK;
}

extern "C" int mi_peer_open(int dev, const unsigned char* handle, void** ptr) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(ptr && handle, "mi_peer_open: null pointer");
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, sizeof h);
    MI_HIP(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
    return MI_OK;
}

extern "C" int mi_peer_close(int dev, void* ptr) {
    MI_TRY(use_device(dev));
    if (ptr) MI_HIP(hipIpcCloseMemHandle(ptr));
    return MI_OK;
}

extern "C" int mi_peer_event_create(int dev, void** event, unsigned char* handle) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(event && handle, "mi_peer_event_create: null pointer");
    hipEvent_t ev = nullptr;
    MI_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventInterprocess));
    hipIpcEventHandle_t h;
    hipError_t e = hipIpcGetEventHandle(&h, ev);
    if (e != hipSuccess) {
        (void)hipEventDestroy(ev);
        return fail(MI_ERR_HIP, "mi_peer_event_create: hipIpcGetEventHandle failed: %s", hipGetErrorString(e));
    }
    std::memset(handle, 0, MI_IPC_HANDLE_BYTES);
    std::memcpy(handle, &h, sizeof h);
    *event = ev;
    return MI_OK;
}

extern "C" int mi_peer_event_open(int dev, const unsigned char* handle, void** event) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(event && handle, "mi_peer_event_open: null pointer");
    hipIpcEventHandle_t h;
    std::memcpy(&h, handle, sizeof h);
    hipEvent_t ev = nullptr;
    MI_HIP(hipIpcOpenEventHandle(&ev, h));
    *event = ev;
    return MI_OK;
}

extern "C" int mi_peer_event_destroy(int dev, void* event) {
    MI_TRY(use_device(dev));
    if (event) MI_HIP(hipEventDestroy(static_cast<hipEvent_t>(event)));
    return MI_OK;
}

extern "C" int mi_peer_event_record(int dev, void* event, void* stream) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(event, "mi_peer_event_record: null event");
    MI_HIP(hipEventRecord(static_cast<hipEvent_t>(event), as_stream(stream)));
    return MI_OK;
}

extern "C" int mi_peer_stream_wait(int dev, void* stream, void* event) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(event, "mi_peer_stream_wait: null event");
    MI_HIP(hipStreamWaitEvent(as_stream(stream), static_cast<hipEvent_t>(event), 0));
    return MI_OK;
}

extern "C" int mi_peer_stream_create(int dev, void** stream) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(stream, "mi_peer_stream_create: null pointer");
    hipStream_t s = nullptr;
    MI_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return MI_OK;
}

extern "C" int mi_peer_stream_destroy(int dev, void* stream) {
    MI_TRY(use_device(dev));
    if (stream) MI_HIP(hipStreamDestroy(as_stream(stream)));
    return MI_OK;
}

extern "C" int mi_peer_copy(int dev, void* stream, void* dst, int dst_dev, const void* src, size_t bytes) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(dst && src, "mi_peer_copy: null pointer");
    if (bytes == 0) return MI_OK;
    int n = 0;
    MI_HIP(hipGetDeviceCount(&n));
    if (dst_dev >= 0 && dst_dev < n) {
        MI_HIP(hipMemcpyPeerAsync(dst, dst_dev, src, dev, bytes, as_stream(stream)));
    } else {  // the peer's device is not visible to this process under an ordinal: the runtime resolves the mapped pointer
        MI_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    }
    return MI_OK;
}
