// TEST INFRASTRUCTURE ONLY -- never linked, imported or called by the product path.
//
// Driver for the real walexi/gnn.cpp reference CPU path.  Compiled by oracle/build_ref.sh against the
// reference's own headers/sources (patched view in a temp dir, see that script) into
// oracle/_ref/ref_driver.  It feeds deterministic inputs through the reference's public API and dumps
// op-level outputs of the GCN hot path so that (a) tests/golden/*.npz can be generated from the
// reference itself and (b) oracle/gcn_oracle.c (the CPU restatement) can be pinned against it.
//
// Everything below is calls into the reference API; the call sequence mirrors
//   graph.cpp:170-191 (GCNConv::forward), graph.cpp:204-212 (aggregate_and_update),
//   nn.cpp:205-211 (Linear::forward), tensor.h:260-276 (backward).
//
// usage: ref_driver <case.bin> <outdir> [flags]
//   case.bin : int32 N,E,Fin,Fout | int32 src[E] | int32 dst[E] | f32 X[N*Fin] | f32 W[Fout*Fin]
//              | f32 bias[Fout] | f32 G[N*Fout]
//              [| f32 w[E]  with the "weighted" flag]
//   flags    : "full"  also run the whole GCNConv layer (with BatchNorm+ReLU) forward
//              "nobwd" skip backward
//              "weighted" also exercise the edge_attr forms of the graph functions (graph.cpp:21-75) on weights w
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "graph.h"
#include "nn.h"
#include "tensor.h"

using namespace cyg;
using namespace std;

static double now_s()
{
    return chrono::duration<double>(chrono::steady_clock::now().time_since_epoch()).count();
}

template <class T>
static void dump(const string &dir, const string &name, const T *p, size_t n)
{
    string path = dir + "/" + name;
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(2); }
    if (n) fwrite(p, sizeof(T), n, f);
    fclose(f);
}

template <class T>
static void dump_va(const string &dir, const string &name, const valarray<T> &v)
{
    vector<T> tmp(begin(v), end(v));
    dump(dir, name, tmp.data(), tmp.size());
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: ref_driver case.bin outdir [full] [nobwd]\n"); return 2; }
    bool full = false, nobwd = false, weighted = false;
    for (int i = 3; i < argc; i++) {
        if (!strcmp(argv[i], "full")) full = true;
        if (!strcmp(argv[i], "nobwd")) nobwd = true;
        if (!strcmp(argv[i], "weighted")) weighted = true;
    }
    string outdir = argv[2];
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int hdr[4];
    if (fread(hdr, sizeof(int), 4, f) != 4) return 2;
    size_t N = hdr[0], E = hdr[1], Fin = hdr[2], Fout = hdr[3];
    vector<int> src(E), dst(E);
    vector<float> X(N * Fin), W(Fout * Fin), B(Fout), G(N * Fout);
    bool ok = fread(src.data(), 4, E, f) == E && fread(dst.data(), 4, E, f) == E &&
              fread(X.data(), 4, X.size(), f) == X.size() && fread(W.data(), 4, W.size(), f) == W.size() &&
              fread(B.data(), 4, B.size(), f) == B.size() && fread(G.data(), 4, G.size(), f) == G.size();
    vector<float> Wt(weighted ? E : 0);
    if (ok && weighted) ok = fread(Wt.data(), 4, E, f) == E;
    fclose(f);
    if (!ok) { fprintf(stderr, "short case file\n"); return 2; }

    // ---- graph: COO -> (dense) -> strip diagonal -> COO         graph.cpp:172
    auto ei = graph::vec_to_edge_list(src, dst);
    double t0 = now_s();
    auto [ei2, ea2_] = graph::add_self_loops(*ei, nullptr, 0, (int)N);
    double t_selfloops = now_s() - t0;
    dump_va(outdir, "ei2.i32", *ei2->data()); // [2, nnz] row-major: sources then destinations

    // ---- degree / norm block                                   graph.cpp:177-185
    t0 = now_s();
    auto adj_mat = graph::edge_to_adj_mat(*ei2, nullptr, N);
    auto deg = adj_mat->sum(-1, true) + 1;
    deg = deg->pow(-0.5);
    auto norm = adj_mat->mm(deg);
    norm *= deg;
    double t_norm = now_s() - t0;
    dump_va(outdir, "s.f32", *deg->data());
    dump_va(outdir, "norm.f32", *norm->data());

    // ---- transform: H = lin(x) = x.mm(W.t())                   graph.cpp:173 -> nn.cpp:205-211
    vector<size_t> xd = {N, Fin};
    auto x = make_shared<tensor<float>>(xd, new valarray<float>(X.data(), X.size()), true);
    graph::GCNConv layer(Fin, Fout);
    valarray<float> wv(W.data(), W.size()), bv(B.data(), B.size());
    layer.get_parameter("weight")->set_data(&wv);
    layer.get_parameter("bias")->set_data(&bv);
    auto lin = layer.get_module("lin");
    t0 = now_s();
    auto H = (*lin)(x);
    double t_lin = now_s() - t0;
    dump_va(outdir, "H.f32", *H->data());

    // ---- aggregate: norm (.) (A.H), then + bias                graph.cpp:204-212, :188
    t0 = now_s();
    auto agg = layer.aggregate_and_update(H, *ei2, &norm);
    double t_agg = now_s() - t0;
    auto out = agg + layer.get_parameter("bias");
    dump_va(outdir, "agg.f32", *agg->data());
    dump_va(outdir, "out.f32", *out->data());

    // ---- backward through Add -> Mul -> MatMul(A,H) -> MatMul(x,W^T) -> Transpose
    double t_bwd = 0;
    if (!nobwd) {
        vector<size_t> gd = {N, Fout};
        auto g = make_shared<tensor<float>>(gd, new valarray<float>(G.data(), G.size()), false);
        t0 = now_s();
        out->backward(g);
        t_bwd = now_s() - t0;
        dump_va(outdir, "dX.f32", *x->grad());
        dump_va(outdir, "dW.f32", *layer.get_parameter("weight")->grad());
        dump_va(outdir, "dbias.f32", *layer.get_parameter("bias")->grad());
    }

    // ---- whole layer (BatchNorm + ReLU in between; "next" row of SURVEY 8(f))  graph.cpp:170-191
    double t_full = 0;
    if (full) {
        graph::GCNConv layer2(Fin, Fout);
        layer2.get_parameter("weight")->set_data(&wv);
        layer2.get_parameter("bias")->set_data(&bv);
        auto x2 = make_shared<tensor<float>>(xd, new valarray<float>(X.data(), X.size()), false);
        auto ei_copy = new tensor<int>(ei->shape(), new valarray<int>(*ei->data()), false);
        graph::Data data(x2, ei_copy);
        t0 = now_s();
        auto out_full = layer2(data);
        t_full = now_s() - t0;
        dump_va(outdir, "out_full.f32", *out_full->data());
        // the two modules between transform and aggregation, on their own (graph.cpp:174-175)
        auto Hbn = (*layer2.get_module("bnorm"))(H);
        auto Hrelu = (*layer2.get_module("relu"))(Hbn);
        dump_va(outdir, "Hbn.f32", *Hbn->data());
        dump_va(outdir, "Hrelu.f32", *Hrelu->data());
        // softmax cross-entropy of the layer output against synthetic targets t_i = (7 i + 3) mod F_out (nn.cpp:442-453)
        vector<int> tgt(N);
        for (size_t i = 0; i < N; i++) tgt[i] = (int)((7 * i + 3) % Fout);
        vector<size_t> td = {N};
        auto target = make_shared<tensor<int>>(td, new valarray<int>(tgt.data(), N), false);
        auto loss = nn::cross_entropy_loss(out_full, target);
        dump_va(outdir, "loss.f32", *loss->data());
        // gradients THROUGH the whole layer: transform <- BatchNorm <- ReLU <- aggregation <- + bias.  In the reference an op
        // that has finished its backward drops every later arrival (operation.h:80-88), and BatchNorm feeds its input to three
        // consumers (x - mean, mean, var): what arrives first wins.  These dumps pin that behaviour (SURVEY.md 8(f) rank 1,
        // "reference-quirk mode"); a backend run with GNNCPP_REFERENCE_QUIRKS=1 must reproduce them.
        if (!nobwd) {
            graph::GCNConv layer3(Fin, Fout);
            layer3.get_parameter("weight")->set_data(&wv);
            layer3.get_parameter("bias")->set_data(&bv);
            auto x3 = make_shared<tensor<float>>(xd, new valarray<float>(X.data(), X.size()), true);
            auto ei_copy3 = new tensor<int>(ei->shape(), new valarray<int>(*ei->data()), false);
            graph::Data data3(x3, ei_copy3);
            auto out3 = layer3(data3);
            vector<size_t> gd3 = {N, Fout};
            auto g3 = make_shared<tensor<float>>(gd3, new valarray<float>(G.data(), G.size()), false);
            out3->backward(g3);
            dump_va(outdir, "full_dX.f32", *x3->grad());
            dump_va(outdir, "full_dW.f32", *layer3.get_parameter("weight")->grad());
            dump_va(outdir, "full_dbias.f32", *layer3.get_parameter("bias")->grad());
            dump_va(outdir, "full_dgamma.f32", *layer3.get_parameter("gammas")->grad());
            dump_va(outdir, "full_dbeta.f32", *layer3.get_parameter("betas")->grad());
        }
    }

    // ---- weighted adjacency: edge_attr through edge_to_adj_mat / sum / mm / add_self_loops   graph.cpp:21-75
    if (weighted) {
        vector<size_t> ed = {E};
        auto ea = make_shared<tensor<float>>(ed, new valarray<float>(Wt.data(), E), false);
        auto adjw = graph::edge_to_adj_mat(*ei, ea.get(), N);          // A[r][c] = w, last duplicate wins, diagonal kept
        auto degw = adjw->sum(-1, true);
        vector<size_t> hd = {N, Fout};
        auto hc = make_shared<tensor<float>>(hd, new valarray<float>(*H->data()), false);
        auto mmw = adjw->mm(hc);
        dump_va(outdir, "w_deg.f32", *degw->data());
        dump_va(outdir, "w_mm.f32", *mmw->data());
        auto [ei_f, ea_f] = graph::add_self_loops(*ei, ea.get(), 2.5f, (int)N);  // diagonal := 2.5, then int(w) != 0 filter
        dump_va(outdir, "w_fill_ei.i32", *ei_f->data());
        dump_va(outdir, "w_fill_ea.f32", *ea_f->data());
        auto [ei_z, ea_z] = graph::add_self_loops(*ei, ea.get(), 0.0f, (int)N);  // diagonal := 0 (dropped by the filter)
        dump_va(outdir, "w_strip_ei.i32", *ei_z->data());
        dump_va(outdir, "w_strip_ea.f32", *ea_z->data());
    }

    printf("{\"N\": %zu, \"E\": %zu, \"nnz\": %zu, \"Fin\": %zu, \"Fout\": %zu, \"t_selfloops\": %.6f, "
           "\"t_norm\": %.6f, \"t_linear\": %.6f, \"t_aggregate\": %.6f, \"t_backward\": %.6f, \"t_full_fwd\": %.6f}\n",
           N, E, ei2->shape()[1], Fin, Fout, t_selfloops, t_norm, t_lin, t_agg, t_bwd, t_full);
    fflush(stdout);
    _Exit(0); // the reference's teardown has mismatched deletes (nn.h:56); skip it
}
