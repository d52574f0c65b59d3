// GPU test of the API's elementwise operator classes over the HIP backend.  Modelled on the reference's
// tests/operation.test.cpp:32-89 ("testing Add", "testing Mul", "testing Div": forward against the valarray closed form,
// backward against the closed-form gradients, exact equality) -- which cannot be built here (doctest.h is not vendored) --
// written against the same API, plus the broadcast shapes of the GCN path ([N,F] op [N,1], [N,F] op [F], scalars).
// Every element is rounded once on the device, so forward results must equal the host's IEEE float results bit for bit.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <valarray>

#include "graph.h"
#include "nn.h"
#include "tensor.h"

using namespace cyg;
using namespace std;

static int failures = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);          \
            failures++;                                                      \
        }                                                                    \
    } while (0)

static bool all_zero(const valarray<float> &d)
{
    for (float v : d)
        if (!(v == 0.0f)) return false;
    return true;
}
static bool all_close(const valarray<float> &a, const valarray<float> &b, float rtol)
{
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); i++)
        if (!(fabsf(a[i] - b[i]) <= rtol * fmaxf(1.0f, fabsf(b[i])))) return false;
    return true;
}

template <class F>
static void section(const char *name, F fn)
{
    try {
        fn();
    } catch (const std::exception &e) {
        printf("FAIL section '%s' threw: %s\n", name, e.what());
        failures++;
    }
}

int main()
{
    manual_seed(1234);
    section("Add", [] {  // ---- "testing Add" (operation.test.cpp:32-46)
        auto add = make_shared<Add<tensor<float>>>();
        vector<size_t> dims = {2, 5};
        auto lhs = randn(dims, -1, 1, true), rhs = randn(dims, -1, 1, true);
        valarray<float> res_vec = *lhs->data() + *rhs->data();
        auto res = add->forward(lhs, rhs);
        CHECK(all_zero(*res->data() - res_vec));
        auto g = randn(res->shape());
        add->backward(g);
        CHECK(all_zero(*g->data() - *lhs->grad()));
        CHECK(all_zero(*g->data() - *rhs->grad()));
    });
    section("Mul", [] {  // ---- "testing Mul" (operation.test.cpp:48-66), rank 3
        auto mul = make_shared<Mul<tensor<float>>>();
        vector<size_t> dims = {2, 3, 4};
        auto lhs = randn(dims, -1, 1, true), rhs = randn(dims, -1, 1, true);
        valarray<float> res_vec = *lhs->data() * *rhs->data();
        auto res = mul->forward(lhs, rhs);
        CHECK(all_zero(*res->data() - res_vec));
        auto g = make_shared<tensor<float>>(dims, 2.0f, false);
        mul->backward(g);
        CHECK(all_zero(valarray<float>(*g->data() * *lhs->data()) - *rhs->grad()));
        CHECK(all_zero(valarray<float>(*g->data() * *rhs->data()) - *lhs->grad()));
    });
    section("Div", [] {  // ---- "testing Div" (operation.test.cpp:68-89)
        auto div = make_shared<Div<tensor<float>>>();
        vector<size_t> dims = {2, 3, 4};
        auto num = randn(dims, -1, 1, true), den = randn(dims, 1, 3, true);
        valarray<float> res_vec = *num->data() / *den->data();
        auto res = div->forward(num, den);
        CHECK(all_zero(*res->data() - res_vec));
        auto g = make_shared<tensor<float>>(dims, 2.0f, false);
        div->backward(g);
        CHECK(all_zero(valarray<float>(*g->data() / *den->data()) - *num->grad()));
        valarray<float> dden = (*g->data() * -1.0f) * (*num->data() / (*den->data() * *den->data()));
        CHECK(all_zero(dden - *den->grad()));
    });
    section("broadcast operators", [] {  // ---- operators with the path's broadcasts: [N,F] op [N,1], [N,F] op [F], scalars, unary minus, in-place forms
        const size_t N = 37, F = 19;
        vector<size_t> nf = {N, F}, n1 = {N, 1}, f = {F};
        auto X = randn(nf, -1, 1, true);
        auto c = randn(n1, 1, 2, true);   // column operand (like norm)
        auto r = randn(f, 1, 2, true);    // row operand (like bias)
        valarray<float> x = *X->data(), cv = *c->data(), rv = *r->data();
        auto expect = [&](auto fn) {
            valarray<float> e(N * F);
            for (size_t i = 0; i < N; i++)
                for (size_t j = 0; j < F; j++) e[i * F + j] = fn(x[i * F + j], cv[i], rv[j]);
            return e;
        };
        CHECK(all_zero(*(X * c)->data() - expect([](float a, float cc, float) { return a * cc; })));
        CHECK(all_zero(*(X / c)->data() - expect([](float a, float cc, float) { return a / cc; })));
        CHECK(all_zero(*(X + r)->data() - expect([](float a, float, float rr) { return a + rr; })));
        CHECK(all_zero(*(X - r)->data() - expect([](float a, float, float rr) { return a + (-rr); })));
        CHECK(all_zero(*(X / r)->data() - expect([](float a, float, float rr) { return a / rr; })));
        CHECK(all_zero(*(c + r)->data() - expect([](float, float cc, float rr) { return cc + rr; })));  // [N,1]+[F] -> [N,F]
        CHECK(all_zero(*(X * 3.0f)->data() - expect([](float a, float, float) { return a * 3.0f; })));
        CHECK(all_zero(*(X / 3.0f)->data() - expect([](float a, float, float) { return a / 3.0f; })));
        CHECK(all_zero(*(X - 0.25f)->data() - expect([](float a, float, float) { return a + (-0.25f); })));
        CHECK(all_zero(*(-X)->data() - expect([](float a, float, float) { return -a; })));
        auto Y = X->clone(false);
        Y -= r;
        Y /= c;
        Y *= 0.5f;
        CHECK(all_zero(*Y->data() - expect([](float a, float cc, float rr) { return ((a + (-rr)) / cc) * 0.5f; })));
        // backward through the broadcasts: out = ((X * c) + r) / c ; G = ones
        auto out = ((X * c) + r) / c;
        CHECK((out->shape() == nf));
        out->backward(make_shared<tensor<float>>(nf, 1.0f, false));
        // d/dX = 1 ;  d/dr_j = sum_i 1/c_i ;  d/dc_i = sum_j ( x/c - (x c + r)/c^2 ) = - sum_j r_j / c_i^2
        valarray<float> ones(1.0f, N * F);
        CHECK(all_close(*X->grad(), valarray<float>(expect([](float, float cc, float) { return (1.0f / cc) * cc; })), 1e-6f));
        valarray<float> dr(F), dc(N);
        double inv = 0;
        for (size_t i = 0; i < N; i++) inv += 1.0 / cv[i];
        double rs = 0;
        for (size_t j = 0; j < F; j++) rs += rv[j];
        for (size_t j = 0; j < F; j++) dr[j] = (float)inv;
        for (size_t i = 0; i < N; i++) dc[i] = (float)(-rs / ((double)cv[i] * cv[i]));
        CHECK(all_close(*r->grad(), dr, 1e-5f));
        CHECK(all_close(*c->grad(), dc, 2e-4f));  // cancellation between the two paths into c
        // shape errors keep the reference's message
        bool threw = false;
        try {
            auto bad = X + randn(vector<size_t>{N + 1, F});
        } catch (const runtime_error &e) {
            threw = true;
        }
        CHECK(threw);
    });
    section("row sums", [] {  // ---- dense sum(-1, keepdim) = row sums, ascending order
        const size_t N = 9, F = 50;
        auto X = randn(vector<size_t>{N, F});
        auto s = X->sum(-1, true);
        CHECK((s->shape() == vector<size_t>{N, 1}));
        valarray<float> e(N);
        for (size_t i = 0; i < N; i++) {
            float acc = 0.f;
            for (size_t j = 0; j < F; j++) acc += (*X->data())[i * F + j];
            e[i] = acc;
        }
        CHECK(all_zero(*s->data() - e));
    });
    section("row pitch", [] {  // ---- a product asked onto a row pitch (detail::PitchRequest): pitch-aware consumers see the pitch,
                               // everything else sees the same values on contiguous rows (tensor.h, detail::Store)
        const size_t N = 300, K = 64, F = 32, LD = 48;
        auto X = randn(vector<size_t>{N, K}, -1, 1, true), W = randn(vector<size_t>{F, K}, -1, 1, true);
        auto plain = X->mm(W->t());
        const valarray<float> want = *plain->data();
        tptr<float> h;
        {
            detail::PitchRequest on_pitch(N, F, LD);
            h = X->mm(W->t());
        }
        int64_t ld = 0;
        const float *hp = h->device_pitched(ld);
        CHECK(ld == (int64_t)LD && hp != nullptr);
        {   // a request that does not match the product's shape is left alone
            detail::PitchRequest other(N + 1, F, LD);
            auto h2 = X->mm(W->t());
            int64_t ld2 = 0;
            h2->device_pitched(ld2);
            CHECK(ld2 == (int64_t)F);
        }
        // a pitch-aware consumer: the aggregation gathers the pitched rows; same bits as from contiguous rows
        vector<int> src, dst;
        for (int i = 0; i < (int)N; i++)
            for (int d = 1; d <= 3; d++) {
                src.push_back(i);
                dst.push_back((i * 7 + d * 13) % (int)N);
            }
        auto ei = graph::vec_to_edge_list(src, dst);
        auto adj = graph::edge_to_adj_mat(*ei, nullptr, N);
        adj->fill_diagonal_(0);
        auto deg = adj->sum(-1, true) + 1;
        deg = deg->pow(-0.5);
        auto norm = adj->mm(deg);
        norm *= deg;
        graph::GCNConv layer(K, F);
        auto a_pitched = layer.aggregate_and_update(h, *ei, &norm), a_plain = layer.aggregate_and_update(plain, *ei, &norm);
        CHECK(all_zero(*a_pitched->data() - *a_plain->data()));
        // ... and its backward reaches X through the product whose output was pitched
        auto g = randn(vector<size_t>{N, F});
        a_pitched->backward(g);
        const valarray<float> dx1 = *X->grad();
        X->zero_grad();
        W->zero_grad();
        a_plain->backward(g);
        CHECK(all_zero(dx1 - *X->grad()));
        // generic code (data(), elementwise operators) sees contiguous rows with the same values
        auto twice = h + h;
        CHECK(all_zero(*twice->data() - (want + want)));
        CHECK(all_zero(*h->data() - want));
        int64_t ld3 = 0;
        h->device_pitched(ld3);
        CHECK(ld3 == (int64_t)F);   // the generic accessors made the storage contiguous
    });
    section("threads and streams", [] {
        // a tensor made by another thread outlives it: its device block was allocated on that thread's stream; read here, released here
        // (host.cpp: owner-tagged blocks -- a block released by a thread other than its owner goes back to the driver, not into a cache)
        tptr<float> from_thread;
        valarray<float> want;
        std::thread t([&] {
            auto a = randn(vector<size_t>{300, 64}), b = randn(vector<size_t>{300, 64});
            from_thread = a + b;                         // computed on the worker's own stream
            want = *a->data() + *b->data();
            (void)from_thread->device_data();            // device copy alive, host copy too
        });
        t.join();
        CHECK(all_zero(*from_thread->data() - want));
        auto again = from_thread + from_thread;          // used on THIS thread's stream
        CHECK(all_zero(*again->data() - (want + want)));
        from_thread.reset();                             // released by a thread that does not own the block
        // an application-supplied stream: same results, and the old (owned) stream is drained and destroyed
        auto x = randn(vector<size_t>{257, 32}), y = randn(vector<size_t>{257, 32});
        const valarray<float> before = *(x * y)->data();
        void *mine = nullptr;
        CHECK(gnnx_stream_create(&mine) == 0);
        detail::set_current_stream(mine);
        CHECK(detail::current_stream() == mine);
        CHECK(all_zero(*(x * y)->data() - before));
        detail::set_current_stream(nullptr);             // back to the NULL stream explicitly (an application's choice, never the default)
        CHECK(all_zero(*(x * y)->data() - before));
        CHECK(gnnx_stream_sync(mine) == 0 && gnnx_stream_destroy(mine) == 0);
    });
    if (failures == 0) printf("OK\n");
    return failures ? 1 : 0;
}
