// RCCL transport behind the C ABI (include/genomad_nn.h, "multi-GPU"): one process per GPU, one
// communicator per gnn_ctx, every collective enqueued on the ctx stream.  The path shards with no
// data-path collective (windows / contigs are independent, weights replicated); what crosses xGMI is the
// END-OF-RUN gather of 12 B/window class scores to rank 0 (the loop being sharded is
// nn_classification.py:316-320 of the reference) plus a few small control messages (counts, names).
//
// librccl.so is dlopen()ed on first use, so that loading libgenomad_nn_hip.so itself (CPU test-suite,
// single-GPU runs) neither needs nor pays for RCCL.  No torch anywhere.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "gnn_common.h"

namespace gnn {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

static Rccl g_rccl;

static int load_rccl() {
    if (g_rccl.handle) return GNN_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) {
        set_error(std::string("cannot dlopen librccl.so: ") + dlerror());
        return GNN_ERR_STATE;
    }
    Rccl r;
    r.handle = h;
#define GNN_SYM(field, name)                                                       \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                 \
    if (!r.field) {                                                                \
        set_error(std::string("librccl.so lacks ") + name);                        \
        return GNN_ERR_STATE;                                                      \
    }
    GNN_SYM(GetUniqueId, "ncclGetUniqueId")
    GNN_SYM(CommInitRank, "ncclCommInitRank")
    GNN_SYM(CommDestroy, "ncclCommDestroy")
    GNN_SYM(AllGather, "ncclAllGather")
    GNN_SYM(AllReduce, "ncclAllReduce")
    GNN_SYM(Gather, "ncclGather")
    GNN_SYM(GetErrorString, "ncclGetErrorString")
#undef GNN_SYM
    g_rccl = r;
    return GNN_OK;
}

#define GNN_NCCL(call)                                                                              \
    do {                                                                                            \
        ncclResult_t r_ = (call);                                                                   \
        if (r_ != ncclSuccess) {                                                                    \
            set_error(std::string(#call) + " failed: " + g_rccl.GetErrorString(r_));                \
            return GNN_ERR_HIP;                                                                     \
        }                                                                                           \
    } while (0)

static int need_comm(gnn_ctx* ctx) {
    if (!ctx) {
        set_error("ctx is NULL");
        return GNN_ERR_ARG;
    }
    if (!ctx->comm) {
        set_error("no communicator: call gnn_comm_init first");
        return GNN_ERR_STATE;
    }
    GNN_HIP(hipSetDevice(ctx->device));
    return finish_pending(ctx);     // scores of an asynchronous classification are ordered before the collective
}

// grow-only device staging buffer of the communicator
static int comm_scratch(gnn_ctx* ctx, size_t bytes, void** out) {
    if (ctx->comm_scratch_bytes < bytes) {
        GNN_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->comm_scratch) (void)hipFree(ctx->comm_scratch);
        ctx->comm_scratch = nullptr;
        ctx->comm_scratch_bytes = 0;
        GNN_HIP(hipMalloc(&ctx->comm_scratch, bytes));
        ctx->comm_scratch_bytes = bytes;
    }
    *out = ctx->comm_scratch;
    return GNN_OK;
}

}  // namespace gnn

using namespace gnn;

extern "C" {

int gnn_comm_unique_id(uint8_t* id128) {
    if (!id128) {
        set_error("id128 is NULL");
        return GNN_ERR_ARG;
    }
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    GNN_NCCL(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == GNN_COMM_ID_BYTES, "ncclUniqueId size");
    std::memcpy(id128, &id, sizeof(id));
    return GNN_OK;
}

int gnn_comm_init(gnn_ctx* ctx, int n_ranks, int rank, const uint8_t* id128) {
    if (!ctx || !id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) {
        set_error("bad argument to gnn_comm_init");
        return GNN_ERR_ARG;
    }
    if (ctx->comm) {
        set_error("communicator already initialised");
        return GNN_ERR_STATE;
    }
    int rc = load_rccl();
    if (rc) return rc;
    GNN_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    GNN_NCCL(g_rccl.CommInitRank(&comm, n_ranks, id, rank));
    ctx->comm = comm;
    ctx->comm_ranks = n_ranks;
    ctx->comm_rank = rank;
    return GNN_OK;
}

int gnn_comm_destroy(gnn_ctx* ctx) {
    if (!ctx) {
        set_error("ctx is NULL");
        return GNN_ERR_ARG;
    }
    if (ctx->comm) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        (void)g_rccl.CommDestroy(static_cast<ncclComm_t>(ctx->comm));
        ctx->comm = nullptr;
    }
    if (ctx->comm_scratch) (void)hipFree(ctx->comm_scratch);
    ctx->comm_scratch = nullptr;
    ctx->comm_scratch_bytes = 0;
    ctx->comm_ranks = 1;
    ctx->comm_rank = 0;
    return GNN_OK;
}

int gnn_comm_info(gnn_ctx* ctx, int* n_ranks, int* rank) {
    if (!ctx) {
        set_error("ctx is NULL");
        return GNN_ERR_ARG;
    }
    if (n_ranks) *n_ranks = ctx->comm ? ctx->comm_ranks : 1;
    if (rank) *rank = ctx->comm ? ctx->comm_rank : 0;
    return GNN_OK;
}

int gnn_comm_gather_dev(gnn_ctx* ctx, const void* send_dev, void* recv_dev, size_t bytes_per_rank, int root) {
    int rc = need_comm(ctx);
    if (rc) return rc;
    if (!send_dev || root < 0 || root >= ctx->comm_ranks || (ctx->comm_rank == root && !recv_dev && bytes_per_rank)) {
        set_error("bad argument to gnn_comm_gather_dev");
        return GNN_ERR_ARG;
    }
    if (!bytes_per_rank) return GNN_OK;
    // rccl.h:735: "recvbuff may be NULL on ranks other than root"
    GNN_NCCL(g_rccl.Gather(send_dev, recv_dev, bytes_per_rank, ncclUint8, root, static_cast<ncclComm_t>(ctx->comm), ctx->stream));
    return GNN_OK;
}

int gnn_comm_gather(gnn_ctx* ctx, const void* send_host, void* recv_host, size_t bytes_per_rank, int root) {
    int rc = need_comm(ctx);
    if (rc) return rc;
    if (!bytes_per_rank) return GNN_OK;
    if (!send_host || (ctx->comm_rank == root && !recv_host)) {
        set_error("bad argument to gnn_comm_gather");
        return GNN_ERR_ARG;
    }
    void* s = nullptr;
    const size_t total = bytes_per_rank * (size_t)(ctx->comm_ranks + 1);
    if ((rc = comm_scratch(ctx, total, &s))) return rc;
    uint8_t* send_dev = static_cast<uint8_t*>(s);
    uint8_t* recv_dev = send_dev + bytes_per_rank;
    GNN_HIP(hipMemcpyAsync(send_dev, send_host, bytes_per_rank, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = gnn_comm_gather_dev(ctx, send_dev, recv_dev, bytes_per_rank, root))) return rc;
    if (ctx->comm_rank == root)
        GNN_HIP(hipMemcpyAsync(recv_host, recv_dev, bytes_per_rank * (size_t)ctx->comm_ranks, hipMemcpyDeviceToHost, ctx->stream));
    GNN_HIP(hipStreamSynchronize(ctx->stream));
    return GNN_OK;
}

int gnn_comm_allgather(gnn_ctx* ctx, const void* send_host, void* recv_host, size_t bytes_per_rank) {
    int rc = need_comm(ctx);
    if (rc) return rc;
    if (!bytes_per_rank) return GNN_OK;
    if (!send_host || !recv_host) {
        set_error("bad argument to gnn_comm_allgather");
        return GNN_ERR_ARG;
    }
    void* s = nullptr;
    if ((rc = comm_scratch(ctx, bytes_per_rank * (size_t)(ctx->comm_ranks + 1), &s))) return rc;
    uint8_t* send_dev = static_cast<uint8_t*>(s);
    uint8_t* recv_dev = send_dev + bytes_per_rank;
    GNN_HIP(hipMemcpyAsync(send_dev, send_host, bytes_per_rank, hipMemcpyHostToDevice, ctx->stream));
    GNN_NCCL(g_rccl.AllGather(send_dev, recv_dev, bytes_per_rank, ncclUint8, static_cast<ncclComm_t>(ctx->comm), ctx->stream));
    GNN_HIP(hipMemcpyAsync(recv_host, recv_dev, bytes_per_rank * (size_t)ctx->comm_ranks, hipMemcpyDeviceToHost, ctx->stream));
    GNN_HIP(hipStreamSynchronize(ctx->stream));
    return GNN_OK;
}

int gnn_comm_allreduce_max(gnn_ctx* ctx, double* values_host, int n) {
    int rc = need_comm(ctx);
    if (rc) return rc;
    if (!values_host || n < 1) {
        set_error("bad argument to gnn_comm_allreduce_max");
        return GNN_ERR_ARG;
    }
    void* s = nullptr;
    if ((rc = comm_scratch(ctx, sizeof(double) * (size_t)n, &s))) return rc;
    GNN_HIP(hipMemcpyAsync(s, values_host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    GNN_NCCL(g_rccl.AllReduce(s, s, (size_t)n, ncclFloat64, ncclMax, static_cast<ncclComm_t>(ctx->comm), ctx->stream));
    GNN_HIP(hipMemcpyAsync(values_host, s, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    GNN_HIP(hipStreamSynchronize(ctx->stream));
    return GNN_OK;
}

int gnn_comm_barrier(gnn_ctx* ctx) {
    double one = 1.0;
    return gnn_comm_allreduce_max(ctx, &one, 1);
}

}  // extern "C"
