// rcclstub.cpp — a stand-in for librccl.so (TEST INFRASTRUCTURE), loaded by the library through LYNSE_HIP_RCCL_PATH in the TSAN /
// bounded-wait tests.  A communicator of ONE rank copies; a communicator of more than one rank behaves like a node whose peers are
// GONE: the collective never completes (the stream it was enqueued on becomes stuck, tests/hipstub/hipstub.cpp).
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstring>

extern "C" void hipstub_make_stream_stuck(hipStream_t s);

struct ncclComm { int world, rank; };

static size_t type_bytes(ncclDataType_t t) {
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 4;
    }
}

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { memset(id, 7, sizeof(*id)); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t* c, int world, ncclUniqueId, int rank) { *c = new ncclComm{world, rank}; return ncclSuccess; }
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t) { return "rcclstub"; }
ncclResult_t ncclAllGather(const void* s, void* r, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t st) {
    if (c->world > 1) { hipstub_make_stream_stuck(st); return ncclSuccess; }   // the peers never arrive
    if (hipStreamQuery(st) == hipSuccess) memmove(r, s, count * type_bytes(t));
    return ncclSuccess;
}
ncclResult_t ncclAllReduce(const void* s, void* r, size_t count, ncclDataType_t t, ncclRedOp_t, ncclComm_t c, hipStream_t st) {
    if (c->world > 1) { hipstub_make_stream_stuck(st); return ncclSuccess; }
    if (s != r && hipStreamQuery(st) == hipSuccess) memmove(r, s, count * type_bytes(t));
    return ncclSuccess;
}
}
