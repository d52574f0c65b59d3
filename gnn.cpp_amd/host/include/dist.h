// dist.h -- the multi-GPU side of the graph layer: a communicator handle and the 1-D vertex partition of one graph
// (SURVEY.md section 8(e)).  The reference has no counterpart (one process, one thread, dense N x N: SURVEY section 2a);
// the types below are what `graph::GCNConv::shard()` consumes so that the reference's call sites -- `layer(data)`,
// `out->backward(G)` -- run unchanged on a rank's rows.  Everything lands on the C-ABI of include/gnnx.h
// (gnnx_comm_*, gnnx_partition_deal, gnnx_halo_plan_*, gnnx_halo_exchange_rows_f32).
//
// Execution model: one rank per GPU.  A rank is either a process (RCCL transport, `Comm::rccl`) or a thread of one
// process (`Comm::local_group`); in both cases a rank's tensors, stream, workspace and allocator cache belong to the
// thread that runs it: every thread gets a stream of its OWN on its first use of the backend (host.cpp: ThreadRuntime; no
// thread issues on the legacy NULL stream), and whatever crosses threads is ordered explicitly -- the in-process collectives
// synchronise the calling rank's stream on both sides of a host barrier (gnnx_comm.hip).
#ifndef GNNCPP_AMD_DIST_H
#define GNNCPP_AMD_DIST_H

#include <memory>
#include <vector>

#include "tensor.h"

namespace dist {

class Comm {
public:
    ~Comm();
    Comm(const Comm &) = delete;
    Comm &operator=(const Comm &) = delete;
    // `world` handles of an in-process group: hand handle r to the thread that runs rank r
    static std::vector<std::shared_ptr<Comm>> local_group(int world);
    // RCCL: rank 0 makes the id, ships the 128 bytes to the others by any channel, every rank calls rccl()
    static void unique_id(unsigned char id_out[128]);
    static std::shared_ptr<Comm> rccl(int world, int rank, const unsigned char id[128]);
    int world() const { return _world; }
    int rank() const { return _rank; }
    gnnx_comm *handle() const { return _h; }
    void allreduce_sum(float *d_buf, int64_t n);  // in place, on the calling thread's current stream

private:
    Comm(gnnx_comm *h, int world, int rank) : _h(h), _world(world), _rank(rank) {}
    gnnx_comm *_h;
    int _world, _rank;
};

}  // namespace dist

namespace graph {

// Device snapshots of the stages of ONE sharded layer step, taken on the rank's stream without any host synchronisation
// when Partition::trace is set (tests: a mismatch against the unsharded layer is localised to its first wrong stage).
struct ShardTrace {
    cyg::tptr<float> h_ext;   // [n_local + fwd.n_halo, F]: the transform's rows and the halo rows as received
    cyg::tptr<float> out;     // [n_local, F]
    cyg::tptr<float> g_ext;   // [n_local + bwd.n_halo, F]: the upstream gradient's rows and its halo rows as received
    cyg::tptr<float> dy;      // [n_local, F]: the backward aggregation's output
};

// The shard of one graph that one rank owns: vertices dealt by degree (every rank: n/P +- 1 rows, the same degree mix,
// the same send volume per peer), rows of A and of A^T as CSR over [local | halo] columns -- a row's entries in the
// reference's order, so a rank's output rows are the unsharded layer's rows bit for bit -- both halo plans, and the
// degree block (deg, deg^-1/2, norm; reference graph.cpp:177-185) with its one-off exchanges already done.
class Partition {
public:
    // edge_index: the WHOLE graph's [2,E] list (every rank passes the same one); collective over `comm`
    Partition(const cyg::tensor<int> &edge_index, size_t num_nodes, std::shared_ptr<dist::Comm> comm, int row_weight = 1);
    ~Partition();
    Partition(const Partition &) = delete;
    Partition &operator=(const Partition &) = delete;

    size_t num_nodes() const { return _n; }
    size_t num_local() const { return (size_t)_n_local; }
    int64_t nnz_local() const { return fwd.nnz; }
    const std::vector<int64_t> &cuts() const { return _cuts; }
    std::vector<int> local_vertices();                            // original id of local row k (rows are spread inside the rank's range)
    cyg::tptr<float> take_rows(const cyg::tptr<float> &full);     // my rows of a replicated [N,F] tensor (device gather)

    struct Side {
        void *rowptr = nullptr, *colidx = nullptr;  // int32, columns in [local | halo] numbering
        int64_t nnz = 0, n_halo = 0, n_send = 0;
        gnnx_halo_plan *plan = nullptr;
        gnnx_spmm_plan *spmm_plan = nullptr;
        int32_t spmm_plan_feat = 0;
    };
    Side fwd, bwd;
    std::shared_ptr<dist::Comm> comm;
    cyg::tptr<float> norm;          // [n_local, 1]
    cyg::tptr<float> s_ext;         // [n_local + fwd.n_halo, 1]: deg^-1/2 of my rows, then of the halo columns as received
    std::shared_ptr<ShardTrace> trace;   // optional: stage snapshots of the next layer step (see ShardTrace)
    std::vector<int> halo_new_ids(const Side &s) const;   // new (rank-contiguous) vertex id of every halo column, in halo order
    void *norm_nz_bwd = nullptr;    // float per non-zero of bwd: norm of the entry's column (the source vertex)
    void ensure_spmm_plans(int32_t n_feat);
    // exchange rows of a [n_local + n_halo, n_feat] device buffer: fills the halo tail from the owners
    void exchange(const Side &s, float *d_buf, int32_t n_feat);
    // ... of a send buffer its producer has already filled (the transform's epilogue: cyg::detail::SendSlotsRequest)
    void exchange_packed(const Side &s, float *d_buf, int32_t n_feat, const float *d_send);
    size_t packed_transforms = 0;   // how many transforms packed their send rows themselves (tests read it)

private:
    size_t _n = 0;
    int64_t _n_local = 0, _lo = 0;
    std::vector<int64_t> _cuts;
    void *_owner = nullptr, *_nid = nullptr, *_verts = nullptr;
    void build_side(Side &s, const int32_t *d_src, const int32_t *d_dst, int64_t n_edges, bool transpose);
};

}  // namespace graph

#endif
