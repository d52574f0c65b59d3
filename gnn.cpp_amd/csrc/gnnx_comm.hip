// Multi-GPU exchange step of the sharded aggregation, natively on RCCL (one process per GPU, xGMI underneath):
//   gnnx_halo_exchange_f32  all-to-all-v of packed feature rows = ONE group of ncclSend/ncclRecv pairs, so every
//                           peer's slice travels on its own point-to-point xGMI link concurrently (never a ring);
//   gnnx_allreduce_sum_f32  dW / dbias reduction (KB..MB: latency-bound, RCCL's choice of algorithm is fine).
// The reference has no communication of any kind; this implements row (e) of section 8.
//
// librccl is bound lazily with dlopen at the first gnnx_comm_* call, so libgnnx_hip.so itself does not depend on it:
// inside a PyTorch process dlopen("librccl.so.1") resolves to the copy torch already loaded (one RCCL per process).
#include <dlfcn.h>

#include <cstring>

#include "gnnx_common.h"

using namespace gnnx;

namespace {

typedef void *ncclComm_t_;
struct NcclUniqueId { char internal[128]; };
constexpr int kNcclFloat32 = 7, kNcclSum = 0;

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(NcclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t_ *, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t_) = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

Rccl *rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return r.handle ? &r : nullptr;
    tried = true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
    }
    if (!r.handle) return nullptr;
    auto sym = [&](const char *n) { return dlsym(r.handle, n); };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd || !r.AllReduce) {
        dlclose(r.handle);
        r.handle = nullptr;
        return nullptr;
    }
    return &r;
}

}  // namespace

struct gnnx_comm {
    ncclComm_t_ comm = nullptr;
    int world = 1, rank = 0;
};

#define GNNX_RCCL_CHECK(R, expr)                                                                                         \
    do {                                                                                                                 \
        int _e = (expr);                                                                                                 \
        if (_e != 0)                                                                                                     \
            return gnnx::set_error(GNNX_ERR_HIP, "%s failed: %s", #expr, (R)->GetErrorString ? (R)->GetErrorString(_e) : "rccl error"); \
    } while (0)

GNNX_API int gnnx_comm_unique_id(void *id_out)
{
    GNNX_REQUIRE(id_out, GNNX_ERR_INVALID_ARG, "id_out is null");
    Rccl *R = rccl();
    GNNX_REQUIRE(R, GNNX_ERR_UNSUPPORTED, "librccl could not be loaded");
    NcclUniqueId id;
    GNNX_RCCL_CHECK(R, R->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return GNNX_OK;
}

GNNX_API int gnnx_comm_init(gnnx_comm **comm_out, int world, int rank, const void *id)
{
    GNNX_REQUIRE(comm_out && id && world >= 1 && rank >= 0 && rank < world, GNNX_ERR_INVALID_ARG, "bad arguments");
    Rccl *R = rccl();
    GNNX_REQUIRE(R, GNNX_ERR_UNSUPPORTED, "librccl could not be loaded");
    NcclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    auto *c = new gnnx_comm();
    c->world = world;
    c->rank = rank;
    int e = R->CommInitRank(&c->comm, world, uid, rank);
    if (e != 0) {
        delete c;
        return set_error(GNNX_ERR_HIP, "ncclCommInitRank failed: %s", R->GetErrorString ? R->GetErrorString(e) : "rccl error");
    }
    *comm_out = c;
    return GNNX_OK;
}

GNNX_API int gnnx_comm_destroy(gnnx_comm *comm)
{
    if (!comm) return GNNX_OK;
    Rccl *R = rccl();
    if (R && comm->comm) R->CommDestroy(comm->comm);
    delete comm;
    return GNNX_OK;
}

GNNX_API int gnnx_halo_exchange_f32(gnnx_comm *comm, const float *d_send, const int64_t *send_rows, float *d_recv,
                                    const int64_t *recv_rows, int32_t n_feat, void *stream)
{
    GNNX_REQUIRE(comm && send_rows && recv_rows && n_feat >= 0, GNNX_ERR_INVALID_ARG, "bad arguments");
    Rccl *R = rccl();
    GNNX_REQUIRE(R, GNNX_ERR_UNSUPPORTED, "librccl could not be loaded");
    hipStream_t st = as_stream(stream);
    int64_t soff = 0, roff = 0;
    for (int p = 0; p < comm->world; p++) {
        GNNX_REQUIRE(send_rows[p] >= 0 && recv_rows[p] >= 0, GNNX_ERR_INVALID_ARG, "negative row count");
        GNNX_REQUIRE((send_rows[p] == 0 || d_send) && (recv_rows[p] == 0 || d_recv), GNNX_ERR_INVALID_ARG, "null buffer");
    }
    GNNX_RCCL_CHECK(R, R->GroupStart());
    for (int p = 0; p < comm->world; p++) {
        if (send_rows[p] > 0)
            GNNX_RCCL_CHECK(R, R->Send(d_send + soff * n_feat, (size_t)send_rows[p] * n_feat, kNcclFloat32, p, comm->comm, st));
        if (recv_rows[p] > 0)
            GNNX_RCCL_CHECK(R, R->Recv(d_recv + roff * n_feat, (size_t)recv_rows[p] * n_feat, kNcclFloat32, p, comm->comm, st));
        soff += send_rows[p];
        roff += recv_rows[p];
    }
    GNNX_RCCL_CHECK(R, R->GroupEnd());
    return GNNX_OK;
}

GNNX_API int gnnx_allreduce_sum_f32(gnnx_comm *comm, float *d_buf, int64_t n, void *stream)
{
    GNNX_REQUIRE(comm && n >= 0, GNNX_ERR_INVALID_ARG, "bad arguments");
    if (n == 0) return GNNX_OK;
    GNNX_REQUIRE(d_buf, GNNX_ERR_INVALID_ARG, "null buffer");
    Rccl *R = rccl();
    GNNX_REQUIRE(R, GNNX_ERR_UNSUPPORTED, "librccl could not be loaded");
    GNNX_RCCL_CHECK(R, R->AllReduce(d_buf, d_buf, (size_t)n, kNcclFloat32, kNcclSum, comm->comm, as_stream(stream)));
    return GNNX_OK;
}
