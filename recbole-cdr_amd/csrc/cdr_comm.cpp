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
    if (state != 1) {
        const char* why = dlerror();             // (ONE call: dlerror() clears the message it returns)
        cdr_set_error("cdr_comm: librccl.so could not be loaded / bound (%s)", why ? why : "missing symbol");
        return nullptr;
    }
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

// The arithmetic of one all-to-all(v), host only (no RCCL, no device): peer p's slice of the send buffer starts where the slices of
// peers 0..p-1 end, likewise on the receive side; counts are in rows of `unit` elements of `elem` bytes.  Exported so that the
// offsets for peers != self are testable on a machine without a GPU (tests/test_abi.py) and a non-Python host can size its buffers.
extern "C" int cdr_a2a_plan(int world, const int64_t* send_counts, const int64_t* recv_counts, int64_t unit, int64_t elem,
                            int64_t* send_off_bytes, int64_t* recv_off_bytes, int64_t* send_elems, int64_t* recv_elems,
                            int64_t* send_total_rows, int64_t* recv_total_rows) {
    CDR_CHECK_ARG(world >= 1 && send_counts && recv_counts && unit > 0 && elem > 0);
    int64_t so = 0, ro = 0;
    for (int p = 0; p < world; ++p) {
        if (send_counts[p] < 0 || recv_counts[p] < 0) { cdr_set_error("cdr_a2a: negative count for peer %d", p); return CDR_EINVAL; }
        if (send_off_bytes) send_off_bytes[p] = so * unit * elem;
        if (recv_off_bytes) recv_off_bytes[p] = ro * unit * elem;
        if (send_elems) send_elems[p] = send_counts[p] * unit;
        if (recv_elems) recv_elems[p] = recv_counts[p] * unit;
        so += send_counts[p]; ro += recv_counts[p];
    }
    if (send_total_rows) *send_total_rows = so;
    if (recv_total_rows) *recv_total_rows = ro;
    return CDR_OK;
}

// send: this rank's buffer, ordered by destination (rows for rank 0 first); counts are HOST arrays [world] in units of `unit` elements
static int a2a(cdr_comm* comm, void* stream, const void* send, const int64_t* send_counts, void* recv, const int64_t* recv_counts,
               int64_t unit, ncclDataType_t dt, size_t elem) {
    rccl_api* a = rccl();
    if (!a) return CDR_ENODEV;
    hipStream_t s = (hipStream_t)stream;
    const int W = comm->world;
    int64_t* plan = (int64_t*)malloc(sizeof(int64_t) * 4 * (size_t)W);
    if (!plan) return CDR_ENOMEM;
    int64_t *soff = plan, *roff = plan + W, *sel = plan + 2 * W, *rel = plan + 3 * W;
    int rc = cdr_a2a_plan(W, send_counts, recv_counts, unit, (int64_t)elem, soff, roff, sel, rel, nullptr, nullptr);
    if (rc != CDR_OK) { free(plan); return rc; }
    // This rank's own slice never enters RCCL: a send/receive to self is a copy kernel on a few channels (measured at one rank, round 6:
    // the row-sharded step 9.9 ms with the self exchange inside RCCL against 5.9 ms with the buffer handed on) -- one device-to-device
    // copy on the same stream instead.  Peers != self go through one grouped send/receive as before.
    const int me = comm->rank;
    if (sel[me] || rel[me]) {
        if (sel[me] != rel[me]) { free(plan); cdr_set_error("cdr_a2a: self send count %lld != self receive count %lld", (long long)sel[me], (long long)rel[me]); return CDR_EINVAL; }
        hipError_t he = hipMemcpyAsync((char*)recv + roff[me], (const char*)send + soff[me], (size_t)sel[me] * elem, hipMemcpyDeviceToDevice, s);
        if (he != hipSuccess) { free(plan); cdr_set_error("cdr_a2a: self copy -> %s", hipGetErrorString(he)); return (int)he; }
    }
    if (W == 1) { free(plan); return CDR_OK; }
    ncclResult_t r = a->GroupStart();
    if (r != ncclSuccess) { free(plan); cdr_set_error("cdr_a2a: ncclGroupStart -> %s", a->GetErrorString(r)); return 1000 + (int)r; }
    const char* what = nullptr;
    for (int p = 0; p < W && r == ncclSuccess; ++p) {
        if (p == me) continue;
        if (sel[p]) { r = a->Send((const char*)send + soff[p], (size_t)sel[p], dt, p, comm->comm, s); what = "ncclSend"; }
        if (r == ncclSuccess && rel[p]) { r = a->Recv((char*)recv + roff[p], (size_t)rel[p], dt, p, comm->comm, s); what = "ncclRecv"; }
    }
    free(plan);
    ncclResult_t e = a->GroupEnd();              // ALWAYS closed: an error inside the group must not leave the communicator in an open group
    if (r != ncclSuccess) { cdr_set_error("cdr_a2a: %s -> %s", what, a->GetErrorString(r)); return 1000 + (int)r; }
    if (e != ncclSuccess) { cdr_set_error("cdr_a2a: ncclGroupEnd -> %s", a->GetErrorString(e)); return 1000 + (int)e; }
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
