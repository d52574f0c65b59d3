// API-level benchmark: the GCN layer through the C++ mirror of the reference API (graph::GCNConv on graph::Data),
// forward + backward, to show what a user of the reference's call sites gets on an MI355X.
//   bench_host_api [n_nodes=1000000] [n_edges=10000000] [features=128] [steps=5] [hot_path_only=1|0|2=both] [scramble_labels=0|1|2=both]
//                  [rmat_seed=1] [also_fuse_bn_stats=0]   one JSON line per configuration (bench.py starts this as a child and adds them to its line)
// scramble_labels: the data set's vertex ids are multiplied by 2654435761 mod n (an isomorphic graph) before the API sees them --
// what a user would do once at load time: R-MAT's hubs are the ids with few one-bits, and 1-KiB feature rows at such ids pile onto
// a few memory channels; the API itself keeps the caller's vertex order.
// Edges: R-MAT from the same SplitMix64 stream as gnn.cpp_amd/synth.py (seed 1, a,b,c = 0.57,0.19,0.19).
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <valarray>
#include <vector>

#include "graph.h"
#include "nn.h"
#include "tensor.h"

using namespace cyg;
using namespace std;

static inline uint64_t splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static double now_s() { return chrono::duration<double>(chrono::steady_clock::now().time_since_epoch()).count(); }

// R-MAT edge list in the given vertex order (OpenMP over edges; the same SplitMix64 stream as gnn.cpp_amd/synth.py).
static double gen_edges(long n, long e, bool scramble, uint64_t seed, vector<int> &src, vector<int> &dst)
{
    int scale = 1;
    while ((1l << scale) < n) scale++;
    const uint64_t key = splitmix64(seed), G = 0x9E3779B97F4A7C15ull;
    const uint32_t ta = (uint32_t)(0.57 * 4294967296.0), tb = (uint32_t)(0.76 * 4294967296.0), tc = (uint32_t)(0.95 * 4294967296.0);
    src.resize(e);
    dst.resize(e);
    double t0 = now_s();
#pragma omp parallel for
    for (long i = 0; i < e; i++) {
        uint64_t s = 0, d = 0;
        for (int l = 0; l < scale; l++) {
            uint32_t r = (uint32_t)(splitmix64(key ^ (((uint64_t)i * 64 + l) * G)) >> 32);
            s = (s << 1) | (r >= tb);
            d = (d << 1) | (((r >= ta) & (r < tb)) | (r >= tc));
        }
        s %= (uint64_t)n;
        d %= (uint64_t)n;
        src[i] = scramble ? graph::scrambled_label((int)s, (size_t)n) : (int)s;
        dst[i] = scramble ? graph::scrambled_label((int)d, (size_t)n) : (int)d;
    }
    return now_s() - t0;
}

// One configuration: one layer over the given edge list; first call + timed steps; prints one JSON line.
static void run_config(long n, long e, size_t F, int steps, bool hot, bool scramble, uint64_t seed, double t_gen, const vector<int> &src,
                       const vector<int> &dst, tptr<float> x, tptr<float> g, bool fuse_bn_stats = true)
{
    auto ei = graph::vec_to_edge_list(src, dst);
    graph::Data data(x, ei.get());
    graph::GCNConv layer(F, F);
    layer.hot_path_only = hot;
    layer.fuse_bn_stats = fuse_bn_stats;   // true (the default): BatchNorm statistics from the transform's epilogue; false: the two-pass reduction

    double t0 = now_s();
    auto out = layer(data);  // first call: uploads, CSR build, norm, plans
    out->backward(g);
    gnnx_device_sync();
    double t_first = now_s() - t0;

    for (int s = 0; s < 3; s++) {  // warm-up: clocks, allocator pool, plans (the first call above idles the GPU behind uploads)
        layer.zero_grad();
        x->zero_grad();
        auto o = layer(data);
        o->backward(g);
    }
    gnnx_device_sync();
    t0 = now_s();
    for (int s = 0; s < steps; s++) {
        layer.zero_grad();
        x->zero_grad();  // the input's gradient too, like the parameters': every step starts from zeroed grads
        auto o = layer(data);
        o->backward(g);
    }
    gnnx_device_sync();
    double ms = (now_s() - t0) / steps * 1e3;
    // probes of the results (every 997th element, summed in double, as hex so that two runs can be compared bit for bit): the first
    // call's output, and the input / weight gradients the last step left
    auto probe = [](const std::valarray<float> &v) {
        double acc = 0.0;
        for (size_t i = 0; i < v.size(); i += 997) acc += (double)v[i];
        uint64_t bits;
        memcpy(&bits, &acc, sizeof(bits));
        return bits;
    };
    const uint64_t p_out = probe(*out->data()), p_dx = probe(*x->grad()), p_dw = probe(*layer.get_parameter("weight")->grad());
    printf("{\"bench\": \"host_api GCNConv fwd+bwd\", \"n_nodes\": %ld, \"n_edges\": %ld, \"features\": %zu, \"hot_path_only\": %d, "
           "\"scrambled_labels\": %d, \"fuse_bn_stats\": %d, \"rmat_seed\": %llu, \"ms_per_step\": %.3f, \"first_call_s\": %.3f, \"edge_gen_s\": %.3f, "
           "\"out_checksum\": %.6e, \"gathered_row_pitch\": %zu, \"probe_out\": \"%016llx\", \"probe_dx\": \"%016llx\", \"probe_dw\": \"%016llx\"}\n",
           n, e, F, (int)hot, (int)scramble, (int)fuse_bn_stats, (unsigned long long)seed, ms, t_first, t_gen, (double)(*out->data())[12345 % (n * F)],
           layer.gathered_row_pitch, (unsigned long long)p_out, (unsigned long long)p_dx, (unsigned long long)p_dw);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 1000000, e = argc > 2 ? atol(argv[2]) : 10000000;
    const size_t F = argc > 3 ? atol(argv[3]) : 128;
    const int steps = argc > 4 ? atoi(argv[4]) : 5;
    const int hot = argc > 5 ? atoi(argv[5]) : 1;            // 1 hot path, 0 full BatchNorm + ReLU layer, 2 both
    const int scramble = argc > 6 ? atoi(argv[6]) : 0;       // 0 as generated, 1 scrambled labels, 2 both
    const uint64_t seed = argc > 7 ? (uint64_t)atoll(argv[7]) : 1;
    const int optin = argc > 8 ? atoi(argv[8]) : 0;          // 1: also time the full layer with the OTHER statistics path (GCNConv::fuse_bn_stats = false: two-pass)
    // features and upstream gradient: drawn once, shared by every configuration (their values do not depend on the vertex order; the
    // first configuration's first_call_s includes their upload, the later ones find them resident)
    manual_seed(7);   // the layers' parameter initialisation
    auto uniform_pm1 = [&](uint64_t stream, bool requires_grad) {   // U[-1, 1) from the counter-based generator, filled by every host core
        auto *v = new std::valarray<float>((size_t)n * F);
        float *pv = &(*v)[0];
        const uint64_t key = splitmix64(stream);
#pragma omp parallel for
        for (long i = 0; i < n * (long)F; i++)
            pv[i] = (float)(splitmix64(key ^ ((uint64_t)i * 0x9E3779B97F4A7C15ull)) >> 40) * (1.0f / 8388608.0f) - 1.0f;
        return std::make_shared<tensor<float>>(std::vector<size_t>{(size_t)n, F}, v, requires_grad);   // adopts the valarray
    };
    auto x = uniform_pm1(101, true);
    auto g = uniform_pm1(102, false);
    vector<int> src, dst;
    for (int sc = 0; sc < 2; sc++) {
        if (scramble != 2 && sc != scramble) continue;
        const double t_gen = gen_edges(n, e, sc != 0, seed, src, dst);
        for (int h = 1; h >= 0; h--) {
            if (hot != 2 && h != hot) continue;
            run_config(n, e, F, steps, h != 0, sc != 0, seed, t_gen, src, dst, x, g, true);   // the layer's default: statistics from the transform
            if (h == 0 && optin) run_config(n, e, F, steps, false, sc != 0, seed, t_gen, src, dst, x, g, false);
        }
    }
    return 0;
}
