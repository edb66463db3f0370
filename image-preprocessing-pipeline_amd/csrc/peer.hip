// Copy-engine transport for the halo exchange of the slab driver (one process per GPU): device buffers shared between processes
// through HIP IPC memory handles, hipMemcpyPeerAsync on a stream of its own, arrival and acknowledgement as FLAG WORDS in exported
// fine-grained device memory.  No reference counterpart: the reference's blocks never exchange anything (LsDeconv.m:643-654).  The
// RCCL route (grouped ncclSend/ncclRecv) runs kernels that need compute units next to the persistent x pass; a peer copy is
// executed by the SDMA engines and needs none.
//
// Why flag words and not interprocess events (rounds 3-4).  The round-4 probe left a hang on record: 200 consecutive
// hipStreamWaitEvent calls of one stream on the rank's own, long completed, hipEventInterprocess event did not return within four
// minutes (profiles/r04_slab_host_cost.txt).  The runtime's source is not in this image (the binary only names hip::IPCEvent and
// clr/hipamd/src/hip_event_ipc.cpp), so the cause cannot be proven here; what the open-source clr of this generation does, as far
// as we know it: an interprocess event is a small ring of signal words in host shared memory with a read and a write index, a
// record claims the next word, and a stream wait is served by a HOST callback on the waiting stream that polls the word the read
// index named when the wait was issued -- i.e. it is not a device-side primitive, every wait costs a marker, a callback-thread
// wake-up and a blocked stream, and waits that outnumber records are outside what the ring was built for.  The old link kept
// record and wait paired with host sequence numbers for exactly that reason.  Instead of leaning on that pairing, the link no
// longer uses interprocess events at all: a sender writes the sequence number of what it has delivered into a word of the
// receiver's exported memory, in stream order behind the payload; the receiver's stream waits for `word >= sequence` in a
// one-lane kernel.  A wait names a value, not "the latest record": waiting twice, early, or for something long delivered is the
// same comparison, nothing is consumed, nothing has to be paired on the host, and no process waits for another on the host at
// all.  Every wait ends by itself after MI_PEER_TIMEOUT_S (default 120 s) and raises the link's status word, so a rank that died
// leaves an error behind, not a hung device.
#include <cstring>
#include <ctime>
#include <new>

#include "mi_internal.h"
#include "mi_lsdeconv.h"

using namespace mi;

static_assert(sizeof(hipIpcMemHandle_t) <= MI_IPC_HANDLE_BYTES, "IPC handles must fit MI_IPC_HANDLE_BYTES");

namespace {

int export_handle(void* ptr, unsigned char* handle, const char* who) {
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, ptr);
    if (e != hipSuccess)
        return fail(MI_ERR_HIP, "%s: hipIpcGetMemHandle failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set on this pool)", who, hipGetErrorString(e));
    std::memset(handle, 0, MI_IPC_HANDLE_BYTES);
    std::memcpy(handle, &h, sizeof h);
    return MI_OK;
}

// flag words of a link's exported flag page (uint32 index)
constexpr int kArr = 0;     // [d]: chunks delivered into my slot d, (exchange - 1) * chunks + chunk + 1 -- written by the rank that fills it
constexpr int kAck = 2;     // [d]: last exchange of MY edge d its receiver has unpacked -- written by that receiver
constexpr int kStatus = 8;  // waits of mine that ran into the timeout -- written by my own wait kernels
constexpr size_t kFlagBytes = 4096;

// one lane polls a word another process writes: system-scope loads (the page is fine-grained memory: never served from a stale L2
// line), the wall clock bounds the wait
__global__ void k_flag_wait(const unsigned* flag, unsigned want, unsigned long long timeout_ticks, unsigned* status) {
    const unsigned long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - want) < 0) {
        __builtin_amdgcn_s_sleep(20);
        if (wall_clock64() - t0 > timeout_ticks) {
            __hip_atomic_fetch_add(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
}
// ... and one lane writes it, behind everything its stream held (the payload copy): release at system scope
__global__ void k_flag_write(unsigned* flag, unsigned value) { __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

}  // namespace

struct mi_peer_link {
    int dev = 0;
    size_t slot = 0, used = 0;        // bytes between two receive slots (a multiple of 256), bytes of a slot that carry rows
    char* payload = nullptr;          // mine, exported: [d][set] slots
    unsigned* flags = nullptr;        // mine, exported: the flag page
    hipStream_t copy = nullptr;
    hipEvent_t packed = nullptr;      // launch stream -> copy stream
    hipEvent_t copied[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [d][set]: the copy out of that staging buffer has run
    struct Peer {
        unsigned char hp[MI_IPC_HANDLE_BYTES], hf[MI_IPC_HANDLE_BYTES];
        char* payload = nullptr;
        unsigned* flags = nullptr;
        int dev = -1;
    } peers[2];
    int n_peers = 0;
    int dst[2] = {-1, -1};            // peer that receives my edge d (and fills my slot 1 - d)
    unsigned long long timeout_ticks = 0;
};

extern "C" int mi_peer_link_create(int dev, size_t slot_bytes, mi_peer_link** out, unsigned char* payload_handle, unsigned char* flag_handle) {
    MI_TRY(use_device(dev));
    MI_REQUIRE(out && payload_handle && flag_handle && slot_bytes > 0, "mi_peer_link_create: null pointer or empty slot");
    *out = nullptr;
    mi_peer_link* L = new (std::nothrow) mi_peer_link;
    if (!L) return fail(MI_ERR_NOMEM, "mi_peer_link_create: out of host memory");
    L->dev = dev;
    L->used = slot_bytes;
    L->slot = (slot_bytes + 255) & ~(size_t)255;
    double tmo = 120.0;
    if (const char* e = std::getenv("MI_PEER_TIMEOUT_S")) tmo = std::max(1.0, std::atof(e));
    L->timeout_ticks = (unsigned long long)(tmo * 1e8);  // wall_clock64: 100 MHz
    int rc = MI_OK;
    auto hip = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == MI_OK) rc = fail(e == hipErrorOutOfMemory ? MI_ERR_NOMEM : MI_ERR_HIP, "mi_peer_link_create: %s failed: %s", what, hipGetErrorString(e));
        return e == hipSuccess;
    };
    // (plain allocations of their own: IPC handles name whole allocations, never blocks of the library's pool)
    void* p = nullptr;
    if (hip(hipMalloc(&p, 4 * L->slot), "hipMalloc")) L->payload = static_cast<char*>(p);
    p = nullptr;
    if (rc == MI_OK && hip(hipExtMallocWithFlags(&p, kFlagBytes, hipDeviceMallocFinegrained), "hipExtMallocWithFlags(fine-grained)")) {
        L->flags = static_cast<unsigned*>(p);
        hip(hipMemset(p, 0, kFlagBytes), "hipMemset");
    }
    if (rc == MI_OK) rc = export_handle(L->payload, payload_handle, "mi_peer_link_create");
    if (rc == MI_OK) rc = export_handle(L->flags, flag_handle, "mi_peer_link_create (flag page)");
    if (rc == MI_OK) hip(hipStreamCreateWithFlags(&L->copy, hipStreamNonBlocking), "hipStreamCreateWithFlags");
    if (rc == MI_OK) hip(hipEventCreateWithFlags(&L->packed, hipEventDisableTiming), "hipEventCreateWithFlags");
    for (int d = 0; d < 2 && rc == MI_OK; ++d)
        for (int st = 0; st < 2 && rc == MI_OK; ++st) hip(hipEventCreateWithFlags(&L->copied[d][st], hipEventDisableTiming), "hipEventCreateWithFlags");
    if (rc == MI_OK) hip(hipDeviceSynchronize(), "hipDeviceSynchronize");  // (the zeroed flag page is in memory before any handle travels)
    if (rc != MI_OK) {
        (void)mi_peer_link_destroy(L);
        return rc;
    }
    *out = L;
    return MI_OK;
}

extern "C" int mi_peer_link_connect(mi_peer_link* L, int d, const unsigned char* payload_handle, const unsigned char* flag_handle, int peer_dev) {
    MI_REQUIRE(L && (d == 0 || d == 1) && payload_handle && flag_handle, "mi_peer_link_connect: invalid arguments");
    MI_TRY(use_device(L->dev));
    for (int i = 0; i < L->n_peers; ++i)  // (two ranks on a ring: both neighbours are the same process, mapped once)
        if (std::memcmp(L->peers[i].hp, payload_handle, MI_IPC_HANDLE_BYTES) == 0) {
            L->dst[d] = i;
            return MI_OK;
        }
    MI_REQUIRE(L->n_peers < 2, "mi_peer_link_connect: more than two neighbours");
    mi_peer_link::Peer& P = L->peers[L->n_peers];
    std::memcpy(P.hp, payload_handle, MI_IPC_HANDLE_BYTES);
    std::memcpy(P.hf, flag_handle, MI_IPC_HANDLE_BYTES);
    hipIpcMemHandle_t h;
    void* q = nullptr;
    std::memcpy(&h, payload_handle, sizeof h);
    MI_HIP(hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess));
    P.payload = static_cast<char*>(q);
    std::memcpy(&h, flag_handle, sizeof h);
    q = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
        (void)hipIpcCloseMemHandle(P.payload);
        P.payload = nullptr;
        return fail(MI_ERR_HIP, "mi_peer_link_connect: mapping the neighbour's flag page failed: %s", hipGetErrorString(e));
    }
    P.flags = static_cast<unsigned*>(q);
    P.dev = peer_dev;
    L->dst[d] = L->n_peers++;
    return MI_OK;
}

extern "C" int mi_peer_link_begin(mi_peer_link* L, void* launch_stream, unsigned n, int src_mask) {
    MI_REQUIRE(L && n >= 1, "mi_peer_link_begin: invalid arguments");
    MI_TRY(use_device(L->dev));
    hipStream_t ls = as_stream(launch_stream);
    const int st = (int)(n & 1u);
    for (int d = 0; d < 2; ++d) {
        // what exchange n - 1 left in my slot d has been unpacked (the unpack kernels lie before this point of the launch stream):
        // its sender -- the rank my edge 1 - d goes to -- may overwrite that set from exchange n + 1 on
        if (n > 1 && (src_mask >> d & 1) && L->dst[1 - d] >= 0) {
            hipLaunchKernelGGL(k_flag_write, dim3(1), dim3(1), 0, ls, L->peers[L->dst[1 - d]].flags + kAck + d, n - 1);
            MI_TRY(launch_check("k_flag_write"));
        }
        // the copy that read this set's staging buffer two exchanges ago has run before the pack kernels overwrite it
        if (n > 2 && L->dst[d] >= 0) MI_HIP(hipStreamWaitEvent(ls, L->copied[d][st], 0));
    }
    return MI_OK;
}

extern "C" int mi_peer_link_send(mi_peer_link* L, void* launch_stream, unsigned n, int k, int chunks, size_t first_byte, size_t bytes,
                                 const void* src_up, const void* src_dn) {
    MI_REQUIRE(L && n >= 1 && chunks >= 1 && k >= 0 && k < chunks && first_byte + bytes <= L->used, "mi_peer_link_send: invalid arguments");
    MI_TRY(use_device(L->dev));
    const int st = (int)(n & 1u);
    const void* src[2] = {src_up, src_dn};
    if (L->dst[0] < 0 && L->dst[1] < 0) return MI_OK;
    int ndev = 0;
    MI_HIP(hipGetDeviceCount(&ndev));
#ifdef MI_PROBES
    static const bool timing = std::getenv("MI_PEER_TIMING") != nullptr;
    static double acc[6] = {0, 0, 0, 0, 0, 0};
    static long calls = 0;
    auto now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; };
    double t_prev = timing ? now() : 0.0;
    auto lap = [&](int i) { if (timing) { const double t = now(); acc[i] += t - t_prev; t_prev = t; } };
#else
    auto lap = [](int) {};
#endif
    // the copy stream waits for what the launch stream holds so far (the pack kernels of this chunk)
    MI_HIP(hipEventRecord(L->packed, as_stream(launch_stream)));
    lap(0);
    MI_HIP(hipStreamWaitEvent(L->copy, L->packed, 0));
    lap(1);
    for (int d = 0; d < 2; ++d) {
        if (L->dst[d] < 0) continue;
        MI_REQUIRE(src[d], "mi_peer_link_send: no staging buffer for edge %d", d);
        mi_peer_link::Peer& P = L->peers[L->dst[d]];
        if (k == 0 && n > 2) {  // the receiver has unpacked what exchange n - 2 left in this set
            hipLaunchKernelGGL(k_flag_wait, dim3(1), dim3(1), 0, L->copy, L->flags + kAck + d, n - 2, L->timeout_ticks, L->flags + kStatus);
            MI_TRY(launch_check("k_flag_wait"));
        }
        lap(2);
        if (bytes) {
            char* dst = P.payload + (size_t)(2 * d + st) * L->slot + first_byte;
            const char* s = static_cast<const char*>(src[d]) + first_byte;
            if (P.dev >= 0 && P.dev < ndev) MI_HIP(hipMemcpyPeerAsync(dst, P.dev, s, L->dev, bytes, L->copy));
            else MI_HIP(hipMemcpyAsync(dst, s, bytes, hipMemcpyDeviceToDevice, L->copy));  // (the runtime resolves the mapped pointer)
        }
        lap(3);
        hipLaunchKernelGGL(k_flag_write, dim3(1), dim3(1), 0, L->copy, P.flags + kArr + d, (n - 1) * (unsigned)chunks + (unsigned)k + 1u);
        MI_TRY(launch_check("k_flag_write"));
        lap(4);
        if (k == chunks - 1) MI_HIP(hipEventRecord(L->copied[d][st], L->copy));
        lap(5);
    }
#ifdef MI_PROBES
    if (timing && ++calls % 16 == 0)
        std::fprintf(stderr, "mi_peer_link_send x %ld: record %.0f, stream wait %.0f, ack wait kernel %.0f, copy %.0f, flag write %.0f, copied record %.0f us per call\n",
                     calls, acc[0] / calls, acc[1] / calls, acc[2] / calls, acc[3] / calls, acc[4] / calls, acc[5] / calls);
#endif
    return MI_OK;
}

extern "C" int mi_peer_link_recv(mi_peer_link* L, void* launch_stream, unsigned n, int k, int chunks, int d, void** slot) {
    MI_REQUIRE(L && n >= 1 && chunks >= 1 && k >= 0 && k < chunks && (d == 0 || d == 1) && slot, "mi_peer_link_recv: invalid arguments");
    MI_TRY(use_device(L->dev));
    hipLaunchKernelGGL(k_flag_wait, dim3(1), dim3(1), 0, as_stream(launch_stream), L->flags + kArr + d, (n - 1) * (unsigned)chunks + (unsigned)k + 1u,
                       L->timeout_ticks, L->flags + kStatus);
    MI_TRY(launch_check("k_flag_wait"));
    *slot = L->payload + (size_t)(2 * d + (int)(n & 1u)) * L->slot;
    return MI_OK;
}

extern "C" int mi_peer_exchange(mi_peer_link* L, void* launch_stream, unsigned n, int src_mask, const void* src_up, const void* src_dn,
                                void** slot_lo, void** slot_hi) {
    MI_REQUIRE(L && slot_lo && slot_hi, "mi_peer_exchange: null pointer");
    *slot_lo = *slot_hi = nullptr;
    MI_TRY(mi_peer_link_send(L, launch_stream, n, 0, 1, 0, L->used, src_up, src_dn));
    if (src_mask & 1) MI_TRY(mi_peer_link_recv(L, launch_stream, n, 0, 1, 0, slot_lo));
    if (src_mask & 2) MI_TRY(mi_peer_link_recv(L, launch_stream, n, 0, 1, 1, slot_hi));
    return MI_OK;
}

extern "C" int mi_peer_link_status(mi_peer_link* L, int* timed_out) {
    MI_REQUIRE(L && timed_out, "mi_peer_link_status: null pointer");
    MI_TRY(use_device(L->dev));
    unsigned v = 0;
    MI_HIP(hipMemcpy(&v, L->flags + kStatus, sizeof v, hipMemcpyDeviceToHost));  // (synchronises with the device)
    *timed_out = (int)v;
    return MI_OK;
}

// unmaps the neighbours' memory; every rank does this before any rank frees what it exported (mi_peer_link_destroy)
extern "C" int mi_peer_link_disconnect(mi_peer_link* L) {
    if (!L) return MI_OK;
    MI_TRY(use_device(L->dev));
    (void)hipDeviceSynchronize();
    for (int i = 0; i < L->n_peers; ++i) {
        if (L->peers[i].payload) (void)hipIpcCloseMemHandle(L->peers[i].payload);
        if (L->peers[i].flags) (void)hipIpcCloseMemHandle(L->peers[i].flags);
        L->peers[i].payload = nullptr;
        L->peers[i].flags = nullptr;
    }
    L->n_peers = 0;
    L->dst[0] = L->dst[1] = -1;
    return MI_OK;
}

extern "C" int mi_peer_link_destroy(mi_peer_link* L) {
    if (!L) return MI_OK;
    (void)mi_peer_link_disconnect(L);
    if (L->copy) (void)hipStreamDestroy(L->copy);
    if (L->packed) (void)hipEventDestroy(L->packed);
    for (auto& row : L->copied)
        for (hipEvent_t e : row)
            if (e) (void)hipEventDestroy(e);
    if (L->payload) (void)hipFree(L->payload);
    if (L->flags) (void)hipFree(L->flags);
    delete L;
    return MI_OK;
}
