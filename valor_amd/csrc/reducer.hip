// Native gradient reducer over RCCL: the bucketed all-reduce of the data-parallel step behind the C ABI (SURVEY 8(b):
// valor_reducer_{create, launch_bucket, wait, destroy}). Replaces torch DDP's reducer for this path (train_utils.py:232; the hand-written
// model of the same thing in the reference tree is apex/apex/parallel/distributed.py:320-470: flat buckets, a side stream, events).
//
// What it owns: one RCCL communicator, one communication stream and one event per bucket. A bucket is a contiguous range of the flat gradient
// arena (element offset + count, fixed at creation), reduced IN PLACE:
//   launch_bucket(b, streams)  records an event on every compute stream that may still be writing gradients, makes the communication stream
//                              wait for them and enqueues the collective (all-reduce, or reduce-scatter + all-gather on the in-place shards:
//                              the two halves of a direct all-reduce on the fully connected xGMI mesh) -- no host synchronisation;
//   wait(stream)               makes `stream` (the one the optimizer runs on) wait for every bucket launched since the last wait -- again
//                              stream-side only.
// Nothing here touches the host clock, so the sequence can sit inside a captured graph (DESIGN 7 (ii) named the Python bookkeeping of
// dist.Reducer as what blocked that); the Python reducer keeps deciding WHEN a bucket is complete (autograd hooks), this file is what runs.
//
// RCCL is bound at run time (dlopen / dlsym): the process already holds the copy torch.distributed loaded, and a second copy of the library
// in one process is asking for trouble; resolving lazily also keeps libvalor_hip.so loadable (and its CPU-side tests running) on a machine
// without RCCL. Order: an already loaded librccl (RTLD_NOLOAD), $VALOR_RCCL_LIB, then the default search path.
// Sum in the arena's element type (bf16 or fp32). The 1 / world mean is folded into the optimizer's gradient scale by the caller.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>

#include "common.h"

extern "C" int valor_reducer_destroy(void* reducer);

// The handful of RCCL / NCCL types and constants this file needs, declared here instead of including <rccl/rccl.h>: the library is bound at
// run time (below), so nothing of RCCL is needed to BUILD libvalor_hip.so either. Values are those of the stable NCCL ABI (nccl.h /
// rccl.h: ncclSuccess 0, ncclSum 0, ncclFloat32 7, ncclBfloat16 9, a 128-byte opaque unique id, an opaque communicator pointer).
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;      // NCCL_UNIQUE_ID_BYTES
#define NCCL_UNIQUE_ID_BYTES 128
enum { ncclSuccess = 0, ncclSum = 0, ncclFloat32 = 7, ncclBfloat16 = 9 };

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    bool ok = false;
};

RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a;
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names)
            if (!a.lib) a.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (!a.lib) {
            const char* e = getenv("VALOR_RCCL_LIB");
            if (e) a.lib = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
        }
        for (const char* n : names)
            if (!a.lib) a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!a.lib) return a;
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.lib, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.lib, "ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.lib, "ncclCommDestroy");
        a.AllReduce = (decltype(a.AllReduce))dlsym(a.lib, "ncclAllReduce");
        a.ReduceScatter = (decltype(a.ReduceScatter))dlsym(a.lib, "ncclReduceScatter");
        a.AllGather = (decltype(a.AllGather))dlsym(a.lib, "ncclAllGather");
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.ReduceScatter && a.AllGather;
        return a;
    }();
    return api;
}

struct Reducer {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int rank = 0, world = 1, mode = 0;
    ncclDataType_t dt = ncclBfloat16;
    size_t esz = 2;
    char* base = nullptr;
    std::vector<int64_t> off, cnt;
    std::vector<hipEvent_t> done;
    std::vector<char> pending;          // launched since the last wait
    hipEvent_t ready = nullptr;         // scratch: "this compute stream is past its gradient writes"
};

}  // namespace

extern "C" int valor_reducer_unique_id(void* id128) {
    if (!id128) return VALOR_ERR_ARG;
    RcclApi& a = rccl();
    if (!a.ok) return VALOR_ERR_LAUNCH;
    ncclUniqueId id;
    if (a.GetUniqueId(&id) != ncclSuccess) return VALOR_ERR_LAUNCH;
    for (int i = 0; i < NCCL_UNIQUE_ID_BYTES; ++i) ((char*)id128)[i] = id.internal[i];
    return VALOR_OK;
}

extern "C" int valor_reducer_create(void** out, const void* id128, int rank, int world, int dtype, void* grad_base, const int64_t* offsets,
                                    const int64_t* counts, int nbuckets, int mode) {
    if (!out || !id128 || !grad_base || !offsets || !counts || nbuckets <= 0 || world <= 0 || rank < 0 || rank >= world) return VALOR_ERR_ARG;
    if ((dtype != VALOR_DT_BF16 && dtype != VALOR_DT_F32) || (mode != 0 && mode != 1)) return VALOR_ERR_ARG;
    for (int i = 0; i < nbuckets; ++i)
        if (offsets[i] < 0 || counts[i] <= 0) return VALOR_ERR_ARG;
    RcclApi& a = rccl();
    if (!a.ok) return VALOR_ERR_LAUNCH;
    Reducer* r = new Reducer;
    r->rank = rank; r->world = world; r->mode = mode;
    r->dt = dtype == VALOR_DT_BF16 ? ncclBfloat16 : ncclFloat32;
    r->esz = dtype == VALOR_DT_BF16 ? 2 : 4;
    r->base = (char*)grad_base;
    r->off.assign(offsets, offsets + nbuckets);
    r->cnt.assign(counts, counts + nbuckets);
    r->pending.assign(nbuckets, 0);
    ncclUniqueId id;
    for (int i = 0; i < NCCL_UNIQUE_ID_BYTES; ++i) id.internal[i] = ((const char*)id128)[i];
    bool ok = a.CommInitRank(&r->comm, world, id, rank) == ncclSuccess;
    ok = ok && hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&r->ready, hipEventDisableTiming) == hipSuccess;
    r->done.resize(nbuckets, nullptr);
    for (int i = 0; ok && i < nbuckets; ++i) ok = hipEventCreateWithFlags(&r->done[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        valor_reducer_destroy(r);
        return VALOR_ERR_LAUNCH;
    }
    *out = r;
    return VALOR_OK;
}

extern "C" int valor_reducer_launch_bucket(void* reducer, int bucket, void* const* compute_streams, int nstreams) {
    Reducer* r = (Reducer*)reducer;
    if (!r || bucket < 0 || bucket >= (int)r->off.size() || nstreams < 0 || (nstreams > 0 && !compute_streams)) return VALOR_ERR_ARG;
    RcclApi& a = rccl();
    for (int i = 0; i < nstreams; ++i) {          // the gradient writes of this bucket may be in flight on any of these
        if (hipEventRecord(r->ready, (hipStream_t)compute_streams[i]) != hipSuccess) return VALOR_ERR_LAUNCH;
        if (hipStreamWaitEvent(r->stream, r->ready, 0) != hipSuccess) return VALOR_ERR_LAUNCH;
    }
    char* buf = r->base + r->off[bucket] * (int64_t)r->esz;
    const size_t n = (size_t)r->cnt[bucket];
    ncclResult_t rc;
    if (r->mode == 1 && n % (size_t)r->world == 0) {
        const size_t shard = n / (size_t)r->world;
        char* mine = buf + (size_t)r->rank * shard * r->esz;        // in place: this rank's slice of the bucket
        rc = a.ReduceScatter(buf, mine, shard, r->dt, ncclSum, r->comm, r->stream);
        if (rc == ncclSuccess) rc = a.AllGather(mine, buf, shard, r->dt, r->comm, r->stream);
    } else {
        rc = a.AllReduce(buf, buf, n, r->dt, ncclSum, r->comm, r->stream);
    }
    if (rc != ncclSuccess) return VALOR_ERR_LAUNCH;
    if (hipEventRecord(r->done[bucket], r->stream) != hipSuccess) return VALOR_ERR_LAUNCH;
    r->pending[bucket] = 1;
    return VALOR_OK;
}

extern "C" int valor_reducer_wait(void* reducer, void* stream) {
    Reducer* r = (Reducer*)reducer;
    if (!r) return VALOR_ERR_ARG;
    for (size_t i = 0; i < r->pending.size(); ++i)
        if (r->pending[i]) {
            if (hipStreamWaitEvent((hipStream_t)stream, r->done[i], 0) != hipSuccess) return VALOR_ERR_LAUNCH;
            r->pending[i] = 0;
        }
    return VALOR_OK;
}

extern "C" int valor_reducer_destroy(void* reducer) {
    Reducer* r = (Reducer*)reducer;
    if (!r) return VALOR_OK;
    if (r->stream) hipStreamSynchronize(r->stream);
    if (r->comm && rccl().ok) rccl().CommDestroy(r->comm);
    for (hipEvent_t e : r->done)
        if (e) hipEventDestroy(e);
    if (r->ready) hipEventDestroy(r->ready);
    if (r->stream) hipStreamDestroy(r->stream);
    delete r;
    return VALOR_OK;
}
