// graph.h -- the graph layer of the reference API (reference include/graph.h:9-139, src/graph.cpp) on the
// MI355X backend: COO <-> adjacency helpers, Data, MessagePassing, GCNConv.
//
// The adjacency "matrix" is a cyg::tensor<float> of shape [N,N] backed by CSR on the device (never dense):
//   edge_to_adj_mat   COO -> CSR, duplicates collapse (reference graph.cpp:21-44)
//   fill_diagonal_(0) strips self loops (graph.cpp:72)
//   adj_to_edge_list  CSR -> COO in row-major order (graph.cpp:46-67)
// so GCNConv::forward / aggregate_and_update keep the reference's call sequence (graph.cpp:170-212) while each
// call lands on a hand-written HIP kernel.
#ifndef GNNCPP_AMD_GRAPH_H
#define GNNCPP_AMD_GRAPH_H

#include <memory>
#include <tuple>
#include <vector>

#include "nn.h"
#include "tensor.h"

namespace graph {

class Partition;  // dist.h

typedef enum DataType { TRAIN, VAL, TEST } DataType;

cyg::tptr<int> vec_to_edge_list(std::vector<int> source, std::vector<int> destination);
// (addition, not in the reference) new vertex id of v under the label scramble for synthetic data sets: (v * 2654435761) mod n, a
// bijection of [0, n).  Generators that draw the bits of an id independently (R-MAT) put the hubs on ids with few one-bits; with
// 1-KiB feature rows those pile onto a few memory channels.  Apply it once to the edge list AND to the row order of every [N, F]
// input before building graph::Data; results come back at row scrambled_label(v, n).  DESIGN.md section 5.
inline int scrambled_label(int v, size_t num_nodes)
{
    return (int)(((unsigned long long)v * 2654435761ull) % (unsigned long long)num_nodes);
}
inline void scramble_labels(std::vector<int> &source, std::vector<int> &destination, size_t num_nodes)
{
    for (int &v : source) v = scrambled_label(v, num_nodes);
    for (int &v : destination) v = scrambled_label(v, num_nodes);
}
// n_nodes == 0 uses max(edge_index)+1 (the reference uses max(edge_index), which overflows its own buffer:
// graph.cpp:25,40 -- SURVEY appendix A; pass n_nodes explicitly for identical behaviour)
cyg::tptr<float> edge_to_adj_mat(const cyg::tensor<int> &edge_index, cyg::tensor<float> *edge_attr = nullptr, size_t n_nodes = 0);
std::tuple<cyg::tptr<int>, cyg::tptr<float>> adj_to_edge_list(cyg::tensor<float> &adj_mat);
std::tuple<cyg::tptr<int>, cyg::tptr<float>> add_self_loops(const cyg::tensor<int> &edge_index, cyg::tensor<float> *edge_attr = nullptr,
                                                            const float &fillValue = 0, const int &num_nodes = 0);

class Data {
public:
    Data() {}
    Data(const cyg::tptr<float> &x, cyg::tensor<int> *edge_index = nullptr, cyg::tptr<float> edge_attr = nullptr,
         cyg::tensor<float> *y = nullptr);
    cyg::tensor<int> *edge_index();
    void set_edge_index(cyg::tensor<int> *edge_index, cyg::tptr<float> edge_attr = nullptr);
    cyg::tptr<float> to_adj();
    size_t num_nodes() const { return _num_nodes; }
    size_t num_node_features() const { return _num_node_features; }
    size_t num_edges() const { return _num_edges; }
    cyg::tptr<float> x() const { return _x; }
    cyg::tptr<float> edge_attr() const { return _edge_attr; }

protected:
    size_t _num_nodes = 0, _num_node_features = 0, _num_edges = 0, _num_edge_features = 0;
    cyg::tensor<int> *_edge_index = nullptr;
    cyg::tensor<float> *_y = nullptr;
    cyg::tptr<float> _x, _edge_attr;
};

class MessagePassing : public nn::Module {
public:
    MessagePassing() {}
    virtual cyg::tptr<float> message(const cyg::tptr<float> *, const cyg::tptr<float> *x_j, const cyg::tptr<float> * = nullptr) { return *x_j; }
    virtual cyg::tptr<float> aggregate_and_update(const cyg::tptr<float> &, const cyg::tensor<int> &, const cyg::tptr<float> *)
    {
        throw std::runtime_error("not yet implemented");
    }
    cyg::tptr<float> operator()(Data &input) { return forward(std::move(input)); }
    using nn::Module::operator();
    virtual cyg::tptr<float> forward(Data &&) { throw std::runtime_error("not yet implemented"); }
    using nn::Module::forward;
    virtual cyg::tptr<float> propagate(const cyg::tensor<int> &edge_index, const cyg::tptr<float> &x, const cyg::tptr<float> *norm = nullptr);
};

class GCNConv : public MessagePassing {
public:
    GCNConv(size_t in_channels, size_t out_channels, float dropout = 0.0);
    // transform -> BatchNorm -> ReLU -> normalised aggregation -> + bias   (reference graph.cpp:170-191)
    cyg::tptr<float> forward(Data &&input) override;
    using MessagePassing::forward;
    cyg::tptr<float> propagate(const cyg::tensor<int> &edge_index, const cyg::tptr<float> &x, const cyg::tptr<float> *others) override;
    cyg::tptr<float> aggregate_and_update(const cyg::tptr<float> &x, const cyg::tensor<int> &edge_index, const cyg::tptr<float> *other) override;

    // true (default): aggregation + norm scaling + bias run as ONE fused SpMM (row-scale + bias epilogue, load-balancing
    // plan) recorded as a single autograd op, and the CSR / degree / norm of a graph are cached across calls on the
    // same Data (a static graph is built once, not three times per forward).  false: op by op through the generic
    // tensor ops (MatMul, Mul, Add), exactly the reference's sequence; both give the same bits.
    bool fused = true;
    void invalidate_graph_cache() { _cache_adj.reset(); }
    // Multi-GPU (dist.h): after shard(), forward(Data) takes THIS RANK's rows of x ([Partition::num_local(), F_in], Data built
    // without an edge_index -- the partition holds the graph) and returns this rank's rows of the layer output; backward leaves
    // LOCAL partial sums in the parameter gradients until allreduce_gradients() sums them over the ranks.
    void shard(std::shared_ptr<Partition> part);
    void allreduce_gradients();
    // false (default): the reference's full layer, transform -> BatchNorm -> ReLU -> aggregation -> bias.
    // true: only the hot path of BASELINE.json (transform -> aggregation -> bias).
    bool hot_path_only = false;
    // true (default since round 5): the BatchNorm batch statistics of the full layer come out of the transform's own epilogue
    // (gnnx_gemm_bn_stats_f32: H is never re-read for them) -- a shifted single-pass variance finished in double.  Neither this nor
    // the two-pass reduction is the reference's own summation order (a sequential sum over the nodes, nn.cpp:303,312, which no
    // parallel sum reproduces); against float64 at 10 M x 256 the two are equally accurate (variance within 3.5e-7 relative vs
    // 3.9e-7, mean within 7e-7 sigma vs 3.5e-6 sigma: tests/test_gpu_parity.py::test_bn_stats_from_the_transform_vs_float64), and the
    // layer output differs from the two-pass path's at the 1e-6 level.  Shapes the fused kernel does not cover (small graphs, the
    // golden cases) take the exact pair of calls either way.  false: always the two-pass statistics.
    bool fuse_bn_stats = true;
    // how many times forward() (re)built the adjacency / norm / plans from the edge list (the static-graph cache's misses)
    size_t graph_cache_builds = 0;
    // the row pitch (floats) the last forward asked the transform's output onto: out_channels, or out_channels + 64 when the graph's hub
    // ids call for the padded gather pitch (gnnx_gather_row_stride; INTEGRATION.md "Vertex order of synthetic graphs")
    size_t gathered_row_pitch = 0;
    size_t _in_channels, _out_channels;
    float _dropout;

private:
    // key of the static-graph cache: (storage id, content version) of the edge_index tensor's device copy + the node count
    size_t _cache_nodes = 0;
    uint64_t _cache_ei_id = 0, _cache_ei_version = 0;
    cyg::tptr<float> _cache_adj, _cache_norm;
    std::shared_ptr<Partition> _part;
    cyg::tptr<float> forward_sharded(const cyg::tptr<float> &x_local);
};

}  // namespace graph

#endif
