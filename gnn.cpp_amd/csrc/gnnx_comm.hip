// Multi-GPU exchange step of the sharded aggregation; the reference has no communication of
// any kind (section 2a).  Two transports behind one handle:
//
//   RCCL   one process per GPU, xGMI underneath.  gnnx_halo_exchange_f32 = ONE group of ncclSend/ncclRecv pairs, so
//          every peer's slice travels on its own point-to-point xGMI link concurrently (never a ring);
//          gnnx_allreduce_sum_f32 = ncclAllReduce (KB..MB: latency-bound, RCCL's choice of algorithm is fine).
//          Types and enums come from <rccl/rccl.h>; the SYMBOLS are bound lazily with dlopen at the first gnnx_comm_*
//          call, so libgnnx_hip.so itself does not depend on librccl: inside a PyTorch process
//          dlopen("librccl.so.1") resolves to the copy torch already loaded (one RCCL per process).
//   local  ranks are threads of ONE process (gnnx_comm_init_local): a rendezvous on host memory plus device copies on
//          each rank's own stream.  Used where the ranks of a job live in one address space -- the C++ host's
//          thread-per-rank mode and every single-box test of the sharded path (P ranks on one GPU).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "gnnx_common.h"

using namespace gnnx;

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) return;
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
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd ||
            !r.AllReduce) {
            dlclose(r.handle);
            r.handle = nullptr;
        }
    });
    return r.handle ? &r : nullptr;
}

const char *rccl_error(Rccl *R, ncclResult_t e) { return R->GetErrorString ? R->GetErrorString(e) : "rccl error"; }

// ---- local transport: rendezvous of `world` threads -------------------------------------------------------------------
struct LocalGroup {
    int world = 1;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool broken = false;
    struct Slot {
        const char *send = nullptr;
        const int64_t *send_bytes = nullptr;
        float *reduce_buf = nullptr;
        int device = -1;  // the device the rank's thread had current when it entered the collective
    };
    std::vector<Slot> slots;

    // all ranks arrive -> all leave; false if a peer abandoned the group before this barrier completed (its handle was destroyed
    // mid-collective).  The reason for waking up is what counts: a peer that has passed the LAST barrier of a collective may destroy
    // its handle (setting `broken`) before a released rank re-acquires the mutex -- that rank's barrier did complete.
    bool barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return false;
        const uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            generation++;
            cv.notify_all();
            return true;
        }
        cv.wait(lk, [&] { return generation != gen || broken; });
        return generation != gen;
    }
    void abandon()
    {
        std::lock_guard<std::mutex> lk(mu);
        broken = true;
        cv.notify_all();
    }
};

// The local all-reduce's kernel reads every peer's buffer directly: when the ranks' threads drive different GPUs that needs peer
// access from this rank's device to each peer's (enabled on first use; GNNX_ERR_UNSUPPORTED if the topology does not allow it).
int ensure_peer_access(int my_dev, int peer_dev)
{
    if (my_dev == peer_dev) return GNNX_OK;
    int can = 0;
    GNNX_HIP_CHECK(hipDeviceCanAccessPeer(&can, my_dev, peer_dev));
    GNNX_REQUIRE(can, GNNX_ERR_UNSUPPORTED, "local communicator: device %d cannot access device %d (no peer access): use the RCCL transport",
                 my_dev, peer_dev);
    const hipError_t e = hipDeviceEnablePeerAccess(peer_dev, 0);
    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
        return set_error(GNNX_ERR_HIP, "hipDeviceEnablePeerAccess(%d) from device %d failed: %s", peer_dev, my_dev, hipGetErrorString(e));
    (void)hipGetLastError();
    return GNNX_OK;
}

__global__ void sum_buffers_kernel(const float *const *bufs, int world, int64_t n, float *out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = bufs[0][i];
    for (int q = 1; q < world; q++) acc += bufs[q][i];  // rank order: every rank computes the same bits
    out[i] = acc;
}

}  // namespace

struct gnnx_comm {
    ncclComm_t comm = nullptr;            // RCCL transport
    std::shared_ptr<LocalGroup> local;    // local transport
    int world = 1, rank = 0;
};

namespace gnnx {

// all-to-all-v of raw bytes: peer p gets send_bytes[p] bytes starting at offset sum(send_bytes[:p]) of d_send; what peer p
// sends lands at offset sum(recv_bytes[:p]) of d_recv.  send_bytes / recv_bytes: HOST arrays of `world` entries.
int comm_alltoallv_bytes(gnnx_comm *comm, const void *d_send, const int64_t *send_bytes, void *d_recv, const int64_t *recv_bytes,
                         void *stream)
{
    GNNX_REQUIRE(comm && send_bytes && recv_bytes, GNNX_ERR_INVALID_ARG, "bad arguments");
    hipStream_t st = as_stream(stream);
    for (int p = 0; p < comm->world; p++) {
        GNNX_REQUIRE(send_bytes[p] >= 0 && recv_bytes[p] >= 0, GNNX_ERR_INVALID_ARG, "negative count");
        GNNX_REQUIRE((send_bytes[p] == 0 || d_send) && (recv_bytes[p] == 0 || d_recv), GNNX_ERR_INVALID_ARG, "null buffer");
    }
    if (comm->local) {
        LocalGroup &g = *comm->local;
        GNNX_HIP_CHECK(hipStreamSynchronize(st));  // my send buffer is complete before any peer reads it
        g.slots[comm->rank].send = static_cast<const char *>(d_send);
        g.slots[comm->rank].send_bytes = send_bytes;
        GNNX_REQUIRE(g.barrier(), GNNX_ERR_HIP, "local communicator: a peer left the group");
        int64_t roff = 0;
        int status = GNNX_OK;
        for (int q = 0; q < g.world && status == GNNX_OK; q++) {
            const LocalGroup::Slot &s = g.slots[q];
            int64_t soff = 0;
            for (int p = 0; p < comm->rank; p++) soff += s.send_bytes[p];
            if (s.send_bytes[comm->rank] != recv_bytes[q])
                status = set_error(GNNX_ERR_SHAPE, "rank %d expects %lld bytes from rank %d, which sends %lld", comm->rank,
                                   (long long)recv_bytes[q], q, (long long)s.send_bytes[comm->rank]);
            else if (recv_bytes[q] > 0 &&
                     hipMemcpyAsync(static_cast<char *>(d_recv) + roff, s.send + soff, (size_t)recv_bytes[q], hipMemcpyDeviceToDevice,
                                    st) != hipSuccess)
                status = set_error(GNNX_ERR_HIP, "local communicator: device copy failed");
            roff += recv_bytes[q];
        }
        hipError_t e = hipStreamSynchronize(st);  // peers may reuse their send buffers after the closing barrier
        const bool ok = g.barrier();
        if (status != GNNX_OK) return status;
        GNNX_HIP_CHECK(e);
        GNNX_REQUIRE(ok, GNNX_ERR_HIP, "local communicator: a peer left the group");
        return GNNX_OK;
    }
    Rccl *R = rccl();
    GNNX_REQUIRE(R, GNNX_ERR_UNSUPPORTED, "librccl could not be loaded");
    ncclResult_t e = R->GroupStart();
    if (e != ncclSuccess) return set_error(GNNX_ERR_HIP, "ncclGroupStart failed: %s", rccl_error(R, e));
    // from here the group is open: record the first failure, ALWAYS close the group, then report
    ncclResult_t first = ncclSuccess;
    int64_t soff = 0, roff = 0;
    for (int p = 0; p < comm->world; p++) {
        if (send_bytes[p] > 0 && first == ncclSuccess)
            first = R->Send(static_cast<const char *>(d_send) + soff, (size_t)send_bytes[p], ncclInt8, p, comm->comm, st);
        if (recv_bytes[p] > 0 && first == ncclSuccess)
            first = R->Recv(static_cast<char *>(d_recv) + roff, (size_t)recv_bytes[p], ncclInt8, p, comm->comm, st);
        soff += send_bytes[p];
        roff += recv_bytes[p];
    }
    e = R->GroupEnd();
    if (first != ncclSuccess) return set_error(GNNX_ERR_HIP, "ncclSend/ncclRecv failed: %s", rccl_error(R, first));
    if (e != ncclSuccess) return set_error(GNNX_ERR_HIP, "ncclGroupEnd failed: %s", rccl_error(R, e));
    return GNNX_OK;
}

// one int64 to / from every peer (HOST arrays of `world` entries); synchronises `stream`
int comm_alltoall_i64(gnnx_comm *comm, const int64_t *h_send, int64_t *h_recv, void *stream)
{
    GNNX_REQUIRE(comm && h_send && h_recv, GNNX_ERR_INVALID_ARG, "bad arguments");
    const int P = comm->world;
    hipStream_t st = as_stream(stream);
    int64_t *d = nullptr;
    GNNX_HIP_CHECK(hipMalloc((void **)&d, sizeof(int64_t) * 2 * (size_t)P));
    std::vector<int64_t> eight((size_t)P, (int64_t)sizeof(int64_t));
    int status = GNNX_OK;
    if (hipMemcpyAsync(d, h_send, sizeof(int64_t) * P, hipMemcpyHostToDevice, st) != hipSuccess)
        status = set_error(GNNX_ERR_HIP, "hipMemcpyAsync failed");
    if (status == GNNX_OK) status = comm_alltoallv_bytes(comm, d, eight.data(), d + P, eight.data(), stream);
    if (status == GNNX_OK && hipMemcpyAsync(h_recv, d + P, sizeof(int64_t) * P, hipMemcpyDeviceToHost, st) != hipSuccess)
        status = set_error(GNNX_ERR_HIP, "hipMemcpyAsync failed");
    if (hipStreamSynchronize(st) != hipSuccess && status == GNNX_OK) status = set_error(GNNX_ERR_HIP, "hipStreamSynchronize failed");
    hipFree(d);
    return status;
}

}  // namespace gnnx

GNNX_API int gnnx_comm_unique_id(void *id_out)
{
    GNNX_REQUIRE(id_out, GNNX_ERR_INVALID_ARG, "id_out is null");
    Rccl *R = rccl();
    GNNX_REQUIRE(R, GNNX_ERR_UNSUPPORTED, "librccl could not be loaded");
    ncclUniqueId id;
    static_assert(sizeof(id) == 128, "include/gnnx.h promises a 128-byte id");
    ncclResult_t e = R->GetUniqueId(&id);
    if (e != ncclSuccess) return set_error(GNNX_ERR_HIP, "ncclGetUniqueId failed: %s", rccl_error(R, e));
    memcpy(id_out, &id, sizeof(id));
    return GNNX_OK;
}

GNNX_API int gnnx_comm_init(gnnx_comm **comm_out, int world, int rank, const void *id)
{
    GNNX_REQUIRE(comm_out && id && world >= 1 && rank >= 0 && rank < world, GNNX_ERR_INVALID_ARG, "bad arguments");
    Rccl *R = rccl();
    GNNX_REQUIRE(R, GNNX_ERR_UNSUPPORTED, "librccl could not be loaded");
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    auto *c = new gnnx_comm();
    c->world = world;
    c->rank = rank;
    ncclResult_t e = R->CommInitRank(&c->comm, world, uid, rank);
    if (e != ncclSuccess) {
        delete c;
        return set_error(GNNX_ERR_HIP, "ncclCommInitRank failed: %s", rccl_error(R, e));
    }
    *comm_out = c;
    return GNNX_OK;
}

GNNX_API int gnnx_comm_init_local(gnnx_comm **comms_out, int world)
{
    GNNX_REQUIRE(comms_out && world >= 1, GNNX_ERR_INVALID_ARG, "bad arguments");
    auto g = std::make_shared<LocalGroup>();
    g->world = world;
    g->slots.resize((size_t)world);
    for (int r = 0; r < world; r++) {
        auto *c = new gnnx_comm();
        c->world = world;
        c->rank = r;
        c->local = g;
        comms_out[r] = c;
    }
    return GNNX_OK;
}

GNNX_API int gnnx_comm_info(const gnnx_comm *comm, int *world, int *rank)
{
    GNNX_REQUIRE(comm, GNNX_ERR_INVALID_ARG, "comm is null");
    if (world) *world = comm->world;
    if (rank) *rank = comm->rank;
    return GNNX_OK;
}

GNNX_API int gnnx_comm_destroy(gnnx_comm *comm)
{
    if (!comm) return GNNX_OK;
    if (comm->local) {
        comm->local->abandon();  // wakes peers still waiting in a collective: they get an error, not a hang
    } else {
        Rccl *R = rccl();
        if (R && comm->comm) R->CommDestroy(comm->comm);
    }
    delete comm;
    return GNNX_OK;
}

GNNX_API int gnnx_halo_exchange_f32(gnnx_comm *comm, const float *d_send, const int64_t *send_rows, float *d_recv,
                                    const int64_t *recv_rows, int32_t n_feat, void *stream)
{
    GNNX_REQUIRE(comm && send_rows && recv_rows && n_feat >= 0, GNNX_ERR_INVALID_ARG, "bad arguments");
    std::vector<int64_t> sb((size_t)comm->world), rb((size_t)comm->world);
    for (int p = 0; p < comm->world; p++) {
        GNNX_REQUIRE(send_rows[p] >= 0 && recv_rows[p] >= 0, GNNX_ERR_INVALID_ARG, "negative row count");
        sb[p] = send_rows[p] * n_feat * (int64_t)sizeof(float);
        rb[p] = recv_rows[p] * n_feat * (int64_t)sizeof(float);
    }
    return comm_alltoallv_bytes(comm, d_send, sb.data(), d_recv, rb.data(), stream);
}

GNNX_API int gnnx_allreduce_sum_f32(gnnx_comm *comm, float *d_buf, int64_t n, void *stream)
{
    GNNX_REQUIRE(comm && n >= 0, GNNX_ERR_INVALID_ARG, "bad arguments");
    if (n == 0) return GNNX_OK;
    GNNX_REQUIRE(d_buf, GNNX_ERR_INVALID_ARG, "null buffer");
    hipStream_t st = as_stream(stream);
    if (comm->local) {
        LocalGroup &g = *comm->local;
        const int P = g.world;
        if (P == 1) return GNNX_OK;
        GNNX_HIP_CHECK(hipStreamSynchronize(st));
        int my_dev = 0;
        GNNX_HIP_CHECK(hipGetDevice(&my_dev));
        g.slots[comm->rank].reduce_buf = d_buf;
        g.slots[comm->rank].device = my_dev;
        GNNX_REQUIRE(g.barrier(), GNNX_ERR_HIP, "local communicator: a peer left the group");
        float *tmp = nullptr;
        const float **ptrs = nullptr;
        int status = GNNX_OK;
        for (int q = 0; q < P && status == GNNX_OK; q++) status = ensure_peer_access(my_dev, g.slots[q].device);
        if (status != GNNX_OK) {
            // no early return between the two barriers: the peers are waiting in the second one
        } else if (hipMalloc((void **)&tmp, sizeof(float) * (size_t)n) != hipSuccess ||
            hipMalloc((void **)&ptrs, sizeof(float *) * (size_t)P) != hipSuccess)
            status = set_error(GNNX_ERR_HIP, "local communicator: hipMalloc failed");
        if (status == GNNX_OK) {
            std::vector<const float *> h((size_t)P);
            for (int q = 0; q < P; q++) h[q] = g.slots[q].reduce_buf;
            if (hipMemcpyAsync(ptrs, h.data(), sizeof(float *) * P, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess)
                status = set_error(GNNX_ERR_HIP, "local communicator: copy failed");
        }
        if (status == GNNX_OK) {
            hipLaunchKernelGGL(sum_buffers_kernel, dim3((uint32_t)ceil_div(n, 256)), dim3(256), 0, st, ptrs, P, n, tmp);
            if (hipStreamSynchronize(st) != hipSuccess) status = set_error(GNNX_ERR_HIP, "local communicator: reduction failed");
        }
        const bool ok = g.barrier();  // every rank has read every buffer: now they may be overwritten
        if (status == GNNX_OK && ok && hipMemcpyAsync(d_buf, tmp, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice, st) != hipSuccess)
            status = set_error(GNNX_ERR_HIP, "local communicator: copy failed");
        hipStreamSynchronize(st);
        if (tmp) hipFree(tmp);
        if (ptrs) hipFree(ptrs);
        if (status != GNNX_OK) return status;
        GNNX_REQUIRE(ok, GNNX_ERR_HIP, "local communicator: a peer left the group");
        return GNNX_OK;
    }
    Rccl *R = rccl();
    GNNX_REQUIRE(R, GNNX_ERR_UNSUPPORTED, "librccl could not be loaded");
    ncclResult_t e = R->AllReduce(d_buf, d_buf, (size_t)n, ncclFloat32, ncclSum, comm->comm, st);
    if (e != ncclSuccess) return set_error(GNNX_ERR_HIP, "ncclAllReduce failed: %s", rccl_error(R, e));
    return GNNX_OK;
}
