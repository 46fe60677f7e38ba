// bfg_comm.hpp -- the one exchange step of the multi-GPU path: RCCL all-reduce (sum, float64) of the per-rank map
// (paint) or offset field (baryonify) over xGMI, on the context's stream.  Replaces the parent-side
// np.sum(outputs, axis=0) of the reference's joblib wrapper (utils/Parallelize.py:312-318).
//
// RCCL is bound at run time (dlopen of librccl.so.1, the SONAME PyTorch-ROCm's bundled copy carries too, so a process
// that has imported torch shares torch's instance): the library has no link-time dependency on RCCL and single-GPU
// users never load it.  One communicator per context = per GPU = per process.
#pragma once
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

constexpr int kCommTickets = 32;   // ring of completion events; a recycled slot only ever makes a waiter wait longer
struct bfg_comm_state {
    ncclComm_t comm;
    int rank, world;
    hipStream_t side;        // the communication stream of bfg_*_begin (created on first use)
    hipEvent_t ev_ready;     // context stream -> communication stream: the buffer has been produced
    hipEvent_t ev_done[kCommTickets];   // communication stream -> context stream: collective `ticket` has finished
    int64_t issued;          // tickets handed out so far (ticket k uses ev_done[k % kCommTickets])
};

namespace bfg_rccl {
struct Api {
    void *handle;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    const char *(*GetErrorString)(ncclResult_t);
};

static Api g_api;                 // handle == nullptr: not loaded
static std::string g_load_error;  // why (set once, under the once-flag; copied into the caller's g_last_error)
static std::once_flag g_once;

static void load_once()
{
    const char *names[] = {std::getenv("BFG_RCCL_SO"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    std::string why;
    for (const char *n : names) {
        if (!n || !*n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
        const char *e = dlerror();                    // one call: glibc clears the message when it is read
        if (why.empty()) why = e ? e : "not found";   // the first candidate's reason (BFG_RCCL_SO if set)
        if (n == names[0]) break;                     // an explicit BFG_RCCL_SO that does not load is an error, not a hint
    }
    if (!h) { g_load_error = "dlopen(librccl.so.1): " + (why.empty() ? std::string("not found") : why); return; }
    Api a;
    a.handle = h;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
    a.ReduceScatter = reinterpret_cast<decltype(a.ReduceScatter)>(dlsym(h, "ncclReduceScatter"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.ReduceScatter || !a.AllGather ||
        !a.GetErrorString) {
        g_load_error = "librccl.so.1 lacks one of the nccl* entry points";
        return;
    }
    g_api = a;
}

static const Api *api()
{
    std::call_once(g_once, load_once);
    if (g_api.handle) return &g_api;
    g_last_error = g_load_error;
    return nullptr;
}
}  // namespace bfg_rccl

#define RCCL_TRY(A, expr)                                                                       \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess) {                                                                \
            g_last_error = std::string(#expr) + ": " + (A)->GetErrorString(r_);                 \
            return BFG_ERR_COMM;                                                                \
        }                                                                                       \
    } while (0)

static void bfg_comm_release(bfg_ctx *c)
{
    if (!c->comm) return;
    if (c->comm->side) { (void)hipStreamSynchronize(c->comm->side); (void)hipStreamDestroy(c->comm->side); }
    if (c->comm->ev_ready) (void)hipEventDestroy(c->comm->ev_ready);
    for (hipEvent_t e : c->comm->ev_done) if (e) (void)hipEventDestroy(e);
    if (const bfg_rccl::Api *A = bfg_rccl::api()) (void)A->CommDestroy(c->comm->comm);
    delete c->comm;
    c->comm = nullptr;
}

extern "C" {

int bfg_comm_unique_id(char *id_out, size_t id_bytes)
{
    if (!id_out || id_bytes < (size_t)BFG_COMM_ID_BYTES) return BFG_ERR_INVALID;
    static_assert(BFG_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "bfg_comm id = ncclUniqueId");
    const bfg_rccl::Api *A = bfg_rccl::api();
    if (!A) return BFG_ERR_COMM;
    ncclUniqueId id;
    RCCL_TRY(A, A->GetUniqueId(&id));
    std::memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return BFG_OK;
}

int bfg_comm_init(bfg_ctx *c, const char *id, size_t id_bytes, int rank, int world)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (!id || id_bytes < (size_t)BFG_COMM_ID_BYTES || world < 1 || rank < 0 || rank >= world) return BFG_ERR_INVALID;
    const bfg_rccl::Api *A = bfg_rccl::api();
    if (!A) return BFG_ERR_COMM;
    bfg_comm_release(c);
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm;
    RCCL_TRY(A, A->CommInitRank(&comm, world, uid, rank));
    c->comm = new bfg_comm_state();
    c->comm->comm = comm; c->comm->rank = rank; c->comm->world = world;
    return BFG_OK;
}

int bfg_comm_destroy(bfg_ctx *c)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    (void)hipStreamSynchronize(c->stream);
    bfg_comm_release(c);
    return BFG_OK;
}

int bfg_comm_info(bfg_ctx *c, int *rank, int *world)
{
    if (!c) return BFG_ERR_INVALID;
    if (rank) *rank = c->comm ? c->comm->rank : 0;
    if (world) *world = c->comm ? c->comm->world : 1;
    return BFG_OK;
}

// in-place sum over the ranks of the communicator; asynchronous, on the context's stream (stream-ordered with the paint /
// offsets kernels before it and the regrid after it -- no host synchronisation).  Without a communicator: world size 1,
// nothing to do.
int bfg_allreduce_f64(bfg_ctx *c, double *d_buf, int64_t count)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (count < 0 || (count > 0 && !d_buf)) return BFG_ERR_INVALID;
    if (!c->comm || c->comm->world == 1 || count == 0) return BFG_OK;
    const bfg_rccl::Api *A = bfg_rccl::api();
    if (!A) return BFG_ERR_COMM;
    RCCL_TRY(A, A->AllReduce(d_buf, d_buf, (size_t)count, ncclDouble, ncclSum, c->comm->comm, c->stream));
    return BFG_OK;
}

// The collectives below run on the context's COMMUNICATION stream, so that they overlap whatever the caller enqueues next on
// the context's stream (the next slice of the same map, the next shell in another buffer): a collective is ordered after
// everything enqueued on the context's stream so far and gets a ticket; bfg_comm_wait(ticket) makes the context's stream wait
// for that collective only (and, the communication stream being in order, the ones begun before it).
static int comm_side_begin(bfg_ctx *c, const bfg_rccl::Api *A, int64_t *ticket, hipEvent_t *done)
{
    (void)A;
    bfg_comm_state *m = c->comm;
    if (!m->side) {
        // the highest priority the device offers: the persistent tile kernel fills every CU, and a collective enqueued behind it should
        // take the workgroup slots the next slice's launch frees before that launch's own workgroups do (what DDP does with its
        // NCCL streams); on a device without stream priorities the range is (0, 0) and this is a plain stream
        int prio_least = 0, prio_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        HIP_TRY(hipStreamCreateWithPriority(&m->side, hipStreamNonBlocking, prio_greatest));
        HIP_TRY(hipEventCreateWithFlags(&m->ev_ready, hipEventDisableTiming));
    }
    const int64_t tk = m->issued + 1;
    hipEvent_t &ev = m->ev_done[tk % kCommTickets];
    if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(m->ev_ready, c->stream));
    HIP_TRY(hipStreamWaitEvent(m->side, m->ev_ready, 0));
    *ticket = tk; *done = ev;
    return BFG_OK;
}

int bfg_allreduce_f64_begin(bfg_ctx *c, double *d_buf, int64_t count, int64_t *ticket)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (ticket) *ticket = 0;
    if (count < 0 || (count > 0 && !d_buf)) return BFG_ERR_INVALID;
    if (!c->comm || c->comm->world == 1 || count == 0) return BFG_OK;
    const bfg_rccl::Api *A = bfg_rccl::api();
    if (!A) return BFG_ERR_COMM;
    int64_t tk; hipEvent_t done;
    rc = comm_side_begin(c, A, &tk, &done);
    if (rc) return rc;
    RCCL_TRY(A, A->AllReduce(d_buf, d_buf, (size_t)count, ncclDouble, ncclSum, c->comm->comm, c->comm->side));
    HIP_TRY(hipEventRecord(done, c->comm->side));
    c->comm->issued = tk;
    if (ticket) *ticket = tk;
    return BFG_OK;
}

int bfg_reduce_scatter_f64_begin(bfg_ctx *c, double *d_buf, int64_t count, int64_t *ticket)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (ticket) *ticket = 0;
    if (count < 0 || (count > 0 && !d_buf)) return BFG_ERR_INVALID;
    if (!c->comm || c->comm->world == 1 || count == 0) return BFG_OK;
    if (count % c->comm->world) return BFG_ERR_INVALID;
    const bfg_rccl::Api *A = bfg_rccl::api();
    if (!A) return BFG_ERR_COMM;
    int64_t tk; hipEvent_t done;
    rc = comm_side_begin(c, A, &tk, &done);
    if (rc) return rc;
    const size_t chunk = (size_t)(count / c->comm->world);
    RCCL_TRY(A, A->ReduceScatter(d_buf, d_buf + chunk * (size_t)c->comm->rank, chunk, ncclDouble, ncclSum, c->comm->comm,
                                 c->comm->side));
    HIP_TRY(hipEventRecord(done, c->comm->side));
    c->comm->issued = tk;
    if (ticket) *ticket = tk;
    return BFG_OK;
}

int bfg_comm_wait(bfg_ctx *c, int64_t ticket)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (!c->comm || c->comm->issued == 0) return BFG_OK;
    if (ticket < 0 || ticket > c->comm->issued) return BFG_ERR_INVALID;
    if (ticket == 0) ticket = c->comm->issued;                   // everything begun so far
    // a slot recycled by a later ticket makes this wait for that later collective: still correct, the stream is in order
    HIP_TRY(hipStreamWaitEvent(c->stream, c->comm->ev_done[ticket % kCommTickets], 0));
    return BFG_OK;
}

// The two halves of the all-reduce, for callers that can work on their own slice in between (the distributed
// BaryonifyShell regrids the pixel range it owns): rank r ends up with the summed elements
// [r * chunk, (r + 1) * chunk) of d_buf (in place), chunk = count / world (count must be a multiple of world) ...
int bfg_reduce_scatter_f64(bfg_ctx *c, double *d_buf, int64_t count)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (count < 0 || (count > 0 && !d_buf)) return BFG_ERR_INVALID;
    if (!c->comm || c->comm->world == 1 || count == 0) return BFG_OK;
    if (count % c->comm->world) return BFG_ERR_INVALID;
    const bfg_rccl::Api *A = bfg_rccl::api();
    if (!A) return BFG_ERR_COMM;
    const size_t chunk = (size_t)(count / c->comm->world);
    RCCL_TRY(A, A->ReduceScatter(d_buf, d_buf + chunk * (size_t)c->comm->rank, chunk, ncclDouble, ncclSum, c->comm->comm,
                                 c->stream));
    return BFG_OK;
}

// ... and every rank's slice is sent to all the others (in place).
int bfg_allgather_f64(bfg_ctx *c, double *d_buf, int64_t count)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (count < 0 || (count > 0 && !d_buf)) return BFG_ERR_INVALID;
    if (!c->comm || c->comm->world == 1 || count == 0) return BFG_OK;
    if (count % c->comm->world) return BFG_ERR_INVALID;
    const bfg_rccl::Api *A = bfg_rccl::api();
    if (!A) return BFG_ERR_COMM;
    const size_t chunk = (size_t)(count / c->comm->world);
    RCCL_TRY(A, A->AllGather(d_buf + chunk * (size_t)c->comm->rank, d_buf, chunk, ncclDouble, c->comm->comm, c->stream));
    return BFG_OK;
}

}  // extern "C"
