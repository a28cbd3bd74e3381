// The exchanges of the multi-GPU path behind the C ABI (SURVEY 8b family (10), 8e): a non-Python host reaches the row-sharded step --
// ids to the owner of their row, rows back, gradient rows home (emcdr.py:110-154 over row-sharded tables), the score all-gather of the
// sharded full-sort (emcdr.py:208-233), the loss / score all-reduce -- without torch.distributed.  Thin by design: one RCCL communicator
// per process (one process per GPU), collectives enqueued on the caller's HIP stream, nothing synchronises.  RCCL is bound at RUN TIME
// (dlopen / dlsym): inside a torch process that is the librccl torch already loaded (two copies of RCCL in one process would each
// bootstrap their own transport), in a plain C host the one on the loader path (/opt/rocm/lib).
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <rccl/rccl.h>
#include "cdr_common.h"

namespace {

struct rccl_api {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

rccl_api* rccl() {
    static rccl_api api;
    static int state = 0;                        // 0 untried, 1 bound, -1 unavailable  (first use is from one thread: communicator set-up)
    if (state == 0) {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (int pass = 0; pass < 2 && !api.handle; ++pass)
            for (const char* n : names) {
                api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));     // an already-loaded RCCL first
                if (api.handle) break;
            }
        state = -1;
        if (api.handle) {
#define CDR_BIND(f) api.f = (decltype(api.f))dlsym(api.handle, "nccl" #f)
            CDR_BIND(GetUniqueId); CDR_BIND(CommInitRank); CDR_BIND(CommDestroy); CDR_BIND(GroupStart); CDR_BIND(GroupEnd);
            CDR_BIND(Send); CDR_BIND(Recv); CDR_BIND(AllGather); CDR_BIND(AllReduce); CDR_BIND(GetErrorString);
#undef CDR_BIND
            if (api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.GroupStart && api.GroupEnd && api.Send && api.Recv &&
                api.AllGather && api.AllReduce && api.GetErrorString) state = 1;
        }
    }
    if (state != 1) { cdr_set_error("cdr_comm: librccl.so could not be loaded / bound (%s)", dlerror() ? dlerror() : "missing symbol"); return nullptr; }
    return &api;
}

}  // namespace

struct cdr_comm { ncclComm_t comm; int world, rank; };

#define CDR_NCCL(api, expr)                                                                        \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) { cdr_set_error("%s: %s -> %s", __func__, #expr, (api)->GetErrorString(r_)); return 1000 + (int)r_; } \
    } while (0)

static_assert(sizeof(ncclUniqueId) == CDR_COMM_ID_BYTES, "cdr_hip.h: CDR_COMM_ID_BYTES");

extern "C" int cdr_comm_unique_id(void* id_out) {
    CDR_CHECK_ARG(id_out);
    rccl_api* a = rccl();
    if (!a) return CDR_ENODEV;
    ncclUniqueId id;
    CDR_NCCL(a, a->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return CDR_OK;
}

extern "C" int cdr_comm_init(cdr_comm** comm, int rank, int world, const void* unique_id) {
    CDR_CHECK_ARG(comm && unique_id && world >= 1 && rank >= 0 && rank < world);
    rccl_api* a = rccl();
    if (!a) return CDR_ENODEV;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    cdr_comm* c = (cdr_comm*)calloc(1, sizeof(cdr_comm));
    if (!c) return CDR_ENOMEM;
    c->world = world; c->rank = rank;
    ncclResult_t r = a->CommInitRank(&c->comm, world, id, rank);             // on the calling thread's current HIP device
    if (r != ncclSuccess) { cdr_set_error("cdr_comm_init: ncclCommInitRank -> %s", a->GetErrorString(r)); free(c); return 1000 + (int)r; }
    *comm = c;
    return CDR_OK;
}

extern "C" int cdr_comm_destroy(cdr_comm* comm) {
    if (!comm) return CDR_OK;
    rccl_api* a = rccl();
    if (a && comm->comm) a->CommDestroy(comm->comm);
    free(comm);
    return CDR_OK;
}

extern "C" int cdr_comm_info(const cdr_comm* comm, int* rank, int* world) {
    CDR_CHECK_ARG(comm && rank && world);
    *rank = comm->rank; *world = comm->world;
    return CDR_OK;
}

// send: this rank's buffer, ordered by destination (rows for rank 0 first); counts are HOST arrays [world] in units of `unit` elements
static int a2a(cdr_comm* comm, void* stream, const void* send, const int64_t* send_counts, void* recv, const int64_t* recv_counts,
               int64_t unit, ncclDataType_t dt, size_t elem) {
    rccl_api* a = rccl();
    if (!a) return CDR_ENODEV;
    hipStream_t s = (hipStream_t)stream;
    CDR_NCCL(a, a->GroupStart());
    int64_t so = 0, ro = 0;
    for (int p = 0; p < comm->world; ++p) {
        if (send_counts[p] < 0 || recv_counts[p] < 0) { a->GroupEnd(); cdr_set_error("cdr_a2a: negative count for peer %d", p); return CDR_EINVAL; }
        if (send_counts[p]) CDR_NCCL(a, a->Send((const char*)send + so * unit * elem, (size_t)(send_counts[p] * unit), dt, p, comm->comm, s));
        if (recv_counts[p]) CDR_NCCL(a, a->Recv((char*)recv + ro * unit * elem, (size_t)(recv_counts[p] * unit), dt, p, comm->comm, s));
        so += send_counts[p]; ro += recv_counts[p];
    }
    CDR_NCCL(a, a->GroupEnd());
    return CDR_OK;
}

extern "C" int cdr_a2a_ids(cdr_comm* comm, void* stream, const int64_t* send, const int64_t* send_counts, int64_t* recv,
                           const int64_t* recv_counts) {
    CDR_CHECK_ARG(comm && send_counts && recv_counts);
    return a2a(comm, stream, send, send_counts, recv, recv_counts, 1, ncclInt64, sizeof(int64_t));
}

extern "C" int cdr_a2a_rows(cdr_comm* comm, void* stream, const float* send, const int64_t* send_rows, float* recv,
                            const int64_t* recv_rows, int D) {
    CDR_CHECK_ARG(comm && send_rows && recv_rows && D > 0);
    return a2a(comm, stream, send, send_rows, recv, recv_rows, D, ncclFloat32, sizeof(float));
}

extern "C" int cdr_allgather_scores(cdr_comm* comm, void* stream, const float* send, int64_t n, float* recv) {
    CDR_CHECK_ARG(comm && send && recv && n > 0);
    rccl_api* a = rccl();
    if (!a) return CDR_ENODEV;
    CDR_NCCL(a, a->AllGather(send, recv, (size_t)n, ncclFloat32, comm->comm, (hipStream_t)stream));
    return CDR_OK;
}

extern "C" int cdr_allreduce_sum_f32(cdr_comm* comm, void* stream, float* buf, int64_t n) {
    CDR_CHECK_ARG(comm && buf && n > 0);
    rccl_api* a = rccl();
    if (!a) return CDR_ENODEV;
    CDR_NCCL(a, a->AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, comm->comm, (hipStream_t)stream));
    return CDR_OK;
}
