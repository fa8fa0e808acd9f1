// hipstub.cpp — a HOST-ONLY stand-in for the HIP runtime (TEST INFRASTRUCTURE; never shipped, never used by the product).
//
// Purpose (SURVEY 5 "race detection / sanitizers: the build must add its own"): the host half of csrc/lynse_hip.hip — reader / writer
// locks, search-context leasing, tickets, the IVF index guard, strikes, lazy builds, bounded waits — is compiled with
// `hipcc --cuda-host-only -fsanitize=thread` and linked against THIS file instead of libamdhip64, so that ThreadSanitizer can watch the
// state machine run on a box without a GPU.  The reference gets the same guarantee from Rust's `Send + Sync` (src/index/mod.rs:78,
// src/python/mod.rs:950); C++ gets no such help.
//
// Model: "device" memory is host memory (zero-filled); copies and fills run at enqueue time; kernel launches do nothing (counts and
// overflow flags therefore read 0: every search is "answered" with empty results on the first plan level — the arithmetic is the GPU
// suite's business).  A stream can become STUCK (an operation that never completes: the stub RCCL's collectives in hang mode — a dead
// peer): everything enqueued behind that point is dropped, events recorded there never become ready, hipStreamWaitEvent on such an event
// makes the waiting stream stuck too.  That is what the bounded-wait tests need: LYNSE_ERR_TIMEOUT instead of a hang.
//
// Deliberately lock-free (atomics only): a mutex in here would add happens-before edges between product threads and hide their races.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

struct ihipStream_t { std::atomic<int> stuck{0}; };
struct ihipEvent_t { std::atomic<int> ready{1}; };

static ihipStream_t g_null_stream;
static inline ihipStream_t* S(hipStream_t s) { return s ? reinterpret_cast<ihipStream_t*>(s) : &g_null_stream; }
static inline bool stuck(hipStream_t s) { return S(s)->stuck.load(std::memory_order_relaxed) != 0; }

extern "C" {

// (called by the stub RCCL in hang mode, and by tests)
void hipstub_make_stream_stuck(hipStream_t s) { S(s)->stuck.store(1, std::memory_order_relaxed); }

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof(*p));
    snprintf(p->name, sizeof(p->name), "hipstub (no device)");
    p->multiProcessorCount = 256;
    p->totalGlobalMem = (size_t)8 << 30;
    return hipSuccess;
}
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : (e == hipErrorNotReady ? "not ready" : "hipstub error"); }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned int) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }

hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) { if (n && !stuck(st)) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) { if (n && !stuck(st)) memset(d, v, n); return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int) { *s = reinterpret_cast<hipStream_t>(new ihipStream_t()); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { return hipStreamCreateWithFlags(s, 0); }
hipError_t hipStreamDestroy(hipStream_t s) { delete reinterpret_cast<ihipStream_t*>(s); return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t s) { return stuck(s) ? hipErrorNotReady : hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) {   // a stuck stream never drains: give up after a moment instead of hanging the test
    if (!stuck(s)) return hipSuccess;
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    return hipErrorUnknown;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned int) {
    if (!reinterpret_cast<ihipEvent_t*>(e)->ready.load(std::memory_order_relaxed)) S(s)->stuck.store(1, std::memory_order_relaxed);
    return hipSuccess;
}

hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned int) { *e = reinterpret_cast<hipEvent_t>(new ihipEvent_t()); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<ihipEvent_t*>(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { reinterpret_cast<ihipEvent_t*>(e)->ready.store(stuck(s) ? 0 : 1, std::memory_order_relaxed); return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t e) { return reinterpret_cast<ihipEvent_t*>(e)->ready.load(std::memory_order_relaxed) ? hipSuccess : hipErrorNotReady; }
hipError_t hipEventSynchronize(hipEvent_t e) {
    if (reinterpret_cast<ihipEvent_t*>(e)->ready.load(std::memory_order_relaxed)) return hipSuccess;
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    return hipErrorUnknown;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }

// ---- kernel launches: nothing runs
static thread_local dim3 t_grid, t_block;
static thread_local size_t t_shmem;
static thread_local hipStream_t t_stream;
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t shmem, hipStream_t st) { t_grid = g; t_block = b; t_shmem = shmem; t_stream = st; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* shmem, hipStream_t* st) { *g = t_grid; *b = t_block; *shmem = t_shmem; *st = t_stream; return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { return hipSuccess; }
void** __hipRegisterFatBinary(const void*) { static void* h = nullptr; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned int, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}

// clang's instrumentation calls these; gcc 11's libtsan predates them (mem* themselves are intercepted by the runtime)
void* __tsan_memcpy(void* d, const void* s, size_t n) { return memcpy(d, s, n); }
void* __tsan_memmove(void* d, const void* s, size_t n) { return memmove(d, s, n); }
void* __tsan_memset(void* d, int v, size_t n) { return memset(d, v, n); }

}  // extern "C"
