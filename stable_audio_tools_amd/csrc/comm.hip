// Native gradient exchange behind the C-ABI (sat_allreduce_{init,bucket,finalize} and an opaque comm handle) — RCCL over xGMI,
// one communicator per process (one process per GPU), collectives enqueued on the caller's stream.  What it replaces: the all-reduce
// Lightning's `ddp` strategy performs for the reference (train.py:138, :148-164) — the same SUM over the flat gradient buffer that
// training.GradAllReduce asks torch.distributed for by default; `GradAllReduce(native=True)` routes its buckets here instead.
//
// RCCL is NOT a link-time dependency of libsat_amd.so: the library is found at first use with dlopen — first the copy PyTorch already
// mapped into the process (`librccl.so`, RTLD_NOLOAD: one RCCL, one HIP runtime), then the system one — so a box without RCCL loads the
// conv / attention / GEMM kernels as before and only these four entry points fail (with a message).  The enum values and the 128-byte
// unique id are RCCL's public ABI (rccl.h: ncclSum = 0, ncclFloat32 = 7, ncclBfloat16 = 9, NCCL_UNIQUE_ID_BYTES = 128).
// Status: the symbols, the id generation and the error paths are tested on the host; the collectives run on the MI355X on a 1-rank
// communicator (tests/test_train_step.py::test_native_exchange_gpu: bit-equal to the step without an exchange; round 5).
#include "sat_device.h"
#include <stdio.h>
#include <string.h>
#if !defined(SAT_HIPEMU)
#include <dlfcn.h>
#endif

namespace {
struct SatUid { char internal[128]; };
typedef int (*fn_get_uid)(SatUid*);
typedef int (*fn_init_rank)(void** comm, int nranks, SatUid id, int rank);
typedef int (*fn_all_reduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, void* stream);
typedef int (*fn_reduce_scatter)(const void* send, void* recv, size_t recvcount, int dtype, int op, void* comm, void* stream);
typedef int (*fn_all_gather)(const void* send, void* recv, size_t sendcount, int dtype, void* comm, void* stream);
typedef int (*fn_destroy)(void* comm);
typedef const char* (*fn_errstr)(int);

struct SatRccl {
    void* lib = nullptr;
    fn_get_uid get_uid = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_reduce_scatter reduce_scatter = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_destroy destroy = nullptr;
    fn_errstr errstr = nullptr;
    bool tried = false;
} g_rccl;

struct SatComm {
    void* comm;
    int world, rank;
};

bool sat_rccl_load() {
    if (g_rccl.tried) return g_rccl.lib != nullptr;
    g_rccl.tried = true;
#if defined(SAT_HIPEMU)
    sat_set_error("sat_allreduce: the host simulator has no RCCL");
    return false;
#else
    const char* names[] = {"librccl.so", "librccl.so.1"};
    void* h = nullptr;
    for (int pass = 0; pass < 2 && !h; ++pass)
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));      // pass 0: the copy already in the process
            if (h) break;
        }
    if (!h) {
        sat_set_error("sat_allreduce: librccl.so not found (dlopen)");
        return false;
    }
    g_rccl.get_uid = (fn_get_uid)dlsym(h, "ncclGetUniqueId");
    g_rccl.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
    g_rccl.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
    g_rccl.reduce_scatter = (fn_reduce_scatter)dlsym(h, "ncclReduceScatter");
    g_rccl.all_gather = (fn_all_gather)dlsym(h, "ncclAllGather");
    g_rccl.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    g_rccl.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
    if (!g_rccl.get_uid || !g_rccl.init_rank || !g_rccl.all_reduce || !g_rccl.reduce_scatter || !g_rccl.all_gather || !g_rccl.destroy) {
        sat_set_error("sat_allreduce: librccl.so lacks an expected entry point");
        dlclose(h);
        return false;
    }
    g_rccl.lib = h;
    return true;
#endif
}

int sat_rccl_fail(const char* who, int rc) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: RCCL error %d (%s)", who, rc, g_rccl.errstr ? g_rccl.errstr(rc) : "?");
    sat_set_error(buf);
    return 1;
}
}  // namespace

// 1 when an RCCL library can be loaded into this process (no communicator is created)
extern "C" int sat_allreduce_available(void) { return sat_rccl_load() ? 1 : 0; }

// rank 0: a fresh 128-byte unique id, to be handed to every rank by the caller (any side channel: the launcher's store, a file, MPI)
extern "C" int sat_allreduce_unique_id(void* id128) {
    if (!id128) { sat_set_error("sat_allreduce_unique_id: null buffer"); return 1; }
    if (!sat_rccl_load()) return 1;
    SatUid id;
    const int rc = g_rccl.get_uid(&id);
    if (rc != 0) return sat_rccl_fail("sat_allreduce_unique_id", rc);
    memcpy(id128, id.internal, 128);
    return 0;
}

// every rank: join the communicator of `world` ranks (the current HIP device is this rank's GPU); *comm is the opaque handle
extern "C" int sat_allreduce_init(const void* id128, int world, int rank, void** comm) {
    if (!id128 || !comm || world <= 0 || rank < 0 || rank >= world) { sat_set_error("sat_allreduce_init: bad arguments"); return 1; }
    if (!sat_rccl_load()) return 1;
    SatUid id;
    memcpy(id.internal, id128, 128);
    void* c = nullptr;
    const int rc = g_rccl.init_rank(&c, world, id, rank);
    if (rc != 0) return sat_rccl_fail("sat_allreduce_init", rc);
    *comm = new SatComm{c, world, rank};
    return 0;
}

// SUM of `count` elements (dtype 0 = fp32, 1 = bf16) over the ranks, in place in `buf`, enqueued on `stream`.
// mode 0: one all-reduce.  mode 1: reduce-scatter + all-gather in place (count % world == 0): both phases drive every xGMI link.
extern "C" int sat_allreduce_bucket(void* comm, void* buf, long long count, int dtype, int mode, void* stream) {
    SatComm* c = (SatComm*)comm;
    if (!c || !buf || count <= 0 || (dtype != 0 && dtype != 1) || (mode != 0 && mode != 1)) { sat_set_error("sat_allreduce_bucket: bad arguments"); return 1; }
    if (!sat_rccl_load()) return 1;
    const int dt = dtype == 0 ? 7 : 9;                      // ncclFloat32 / ncclBfloat16
    const size_t esz = dtype == 0 ? 4 : 2;
    if (mode == 1 && count % c->world == 0 && c->world > 0) {
        const size_t k = (size_t)(count / c->world);
        char* shard = (char*)buf + (size_t)c->rank * k * esz;
        int rc = g_rccl.reduce_scatter(buf, shard, k, dt, 0, c->comm, stream);
        if (rc != 0) return sat_rccl_fail("sat_allreduce_bucket (reduce_scatter)", rc);
        rc = g_rccl.all_gather(shard, buf, k, dt, c->comm, stream);
        if (rc != 0) return sat_rccl_fail("sat_allreduce_bucket (all_gather)", rc);
        return 0;
    }
    const int rc = g_rccl.all_reduce(buf, buf, (size_t)count, dt, 0, c->comm, stream);
    if (rc != 0) return sat_rccl_fail("sat_allreduce_bucket", rc);
    return 0;
}

extern "C" int sat_allreduce_finalize(void* comm) {
    SatComm* c = (SatComm*)comm;
    if (!c) return 0;
    int rc = 0;
    if (g_rccl.lib && c->comm) rc = g_rccl.destroy(c->comm);
    delete c;
    if (rc != 0) return sat_rccl_fail("sat_allreduce_finalize", rc);
    return 0;
}
