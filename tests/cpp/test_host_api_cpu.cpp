// CPU-side conventions of the C++ API mirror (no device call is made): constructor errors with the reference's
// message texts, ownership, grad bookkeeping, shape helpers, module registry.  Modelled on the assertions of the
// reference's tests/tensor.test.cpp:28-69 and tests/nn.test.cpp:19-31 (which cannot be built here: doctest.h is
// not vendored), written against the same API.
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>

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

static bool throws_with(const function<void()> &f, const char *msg)
{
    try {
        f();
    } catch (const runtime_error &e) {
        return msg == nullptr || strcmp(e.what(), msg) == 0;
    }
    return false;
}

int main()
{
    vector<size_t> dims = {3, 6, 9};
    // --- tensor construction (tests/tensor.test.cpp:22-38)
    CHECK(throws_with([&] { tensor<int> t(dims, initialize<int>(dims, 3), true); }, ERROR_GRAD_DTYPE));
    auto arr1 = initialize<float>(dims, 3);
    auto t1 = make_shared<tensor<float>>(dims, arr1, true);
    CHECK(t1->data() == arr1);  // the valarray is adopted, not copied
    CHECK((*t1->data())[7] == 3.0f && t1->numel() == 162 && t1->rank() == 3);
    vector<size_t> dims2 = {10, 20};
    CHECK(throws_with([&] { tensor<float> t(dims2, initialize<float>(dims, 3), false); }, ERROR_SIZE_MISMATCH));
    CHECK(throws_with([&] { tensor<float> t(vector<size_t>{}, 1.0f, false); }, ERROR_INVALID_DIMS));
    CHECK(throws_with([&] { tensor<float> t(vector<size_t>{3, 0}, 1.0f, false); }, ERROR_INVALID_DIMS));
    // --- grad bookkeeping (tests/tensor.test.cpp:40-45)
    t1->requires_grad_(true);
    CHECK(t1->grad()->sum() == 0.0f && t1->grad()->size() == 162);
    t1->requires_grad_(false);
    CHECK(throws_with([&] { t1->grad(); }, nullptr));
    // --- squeeze / unsqueeze (tests/tensor.test.cpp:46-60)
    auto t2 = make_shared<tensor<float>>(vector<size_t>{1, 1, 1, 2, 3, 4}, 1.0f);
    t2->squeeze();
    CHECK((t2->shape() == vector<size_t>{2, 3, 4}));
    t2->unsqueeze(2);
    CHECK((t2->shape() == vector<size_t>{2, 3, 1, 4}));
    t2->squeeze();
    t2->unsqueeze(-2);
    CHECK((t2->shape() == vector<size_t>{2, 3, 1, 4}));
    CHECK(throws_with([&] { t2->unsqueeze(5); }, ERROR_OUT_OF_BOUND_DIM));
    // --- in-place op on a leaf that requires grad (tests/tensor.test.cpp:67-69)
    t1->requires_grad_(true);
    CHECK(throws_with([&] { t1 += 3; }, ERROR_IN_PLACE_OP_LEAF));
    // --- backward argument checks (tensor.h:262-267)
    CHECK(throws_with([&] { t1->backward(); }, ERROR_NON_SCALAR_BACKPROP));
    CHECK(throws_with([&] { t1->backward(make_shared<tensor<float>>(dims2, 1.0f)); }, ERROR_GRAD_MISMATCH));
    // --- shape checks of mm / t (utils.cpp:8-78)
    auto a = make_shared<tensor<float>>(vector<size_t>{4, 5}, 1.0f), b = make_shared<tensor<float>>(vector<size_t>{6, 7}, 1.0f);
    CHECK(throws_with([&] { a->mm(b); }, ERROR_MM_COMPATIBLE));
    CHECK(throws_with([&] { a->t(0, 0); }, ERROR_TRANSPOSE));
    CHECK(throws_with([&] { a->add(b); }, ERROR_SIZE_MISMATCH));
    auto at = a->t(-1, -2);  // a view: no device work
    CHECK((at->shape() == vector<size_t>{5, 4}) && at->is_transposed_view());

    // --- module registry (tests/nn.test.cpp:19-31)
    auto module = make_shared<nn::Module>();
    module->register_parameter("m1", make_shared<tensor<float>>(vector<size_t>{2, 3, 4}, 1.0f, true));
    module->eval();
    CHECK(module->training == false);
    module->train();
    CHECK(module->training == true);
    CHECK(module->parameters().size() == 1);
    // --- Linear: weight [out,in], U(-1/sqrt(in), 1/sqrt(in)) (nn.cpp:187-204)
    nn::Linear lin(16, 4, true);
    CHECK((lin.get_parameter("weight")->shape() == vector<size_t>{4, 16}) && (lin.get_parameter("bias")->shape() == vector<size_t>{4}));
    CHECK(lin.get_parameter("weight")->requires_grad());
    float wmax = abs(*lin.get_parameter("weight")->data()).max();
    CHECK(wmax <= 0.25f && wmax > 0.0f);
    // --- GCNConv registers lin (no bias) / bnorm / drop / relu + its own bias = 0 (graph.cpp:160-168)
    graph::GCNConv conv(10, 20);
    CHECK(conv.get_module("lin") && conv.get_module("bnorm") && conv.get_module("drop") && conv.get_module("relu"));
    CHECK((conv.get_parameter("weight")->shape() == vector<size_t>{20, 10}));
    CHECK((conv.get_parameter("bias")->shape() == vector<size_t>{20}) && conv.get_parameter("bias")->data()->sum() == 0.0f);
    CHECK(throws_with([&] { conv.get_parameter("nope"); }, nullptr));
    conv.eval();
    CHECK(conv.get_module("lin")->training == false);

    // --- graph containers (tests/graph.test.cpp:19-36, graph.cpp:10-19,77-100)
    auto ei = graph::vec_to_edge_list({1, 2, 3, 0, 4, 1, 2, 3}, {1, 2, 0, 1, 2, 2, 1, 1});
    CHECK((ei->shape() == vector<size_t>{2, 8}) && (*ei->data())[8] == 1 && (*ei->data())[3] == 0);
    CHECK(throws_with([&] { graph::vec_to_edge_list({1, 2}, {1}); }, "input vectors must be of same length"));
    {   // (addition) label scramble for synthetic data sets: a bijection, applied to both ends of every edge
        vector<int> s = {0, 1, 2, 3, 14}, d = {14, 0, 7, 7, 1}, seen(15, 0);
        for (int v = 0; v < 15; v++) seen[graph::scrambled_label(v, 15)]++;
        bool bij = true;
        for (int c : seen) bij = bij && c == 1;
        graph::scramble_labels(s, d, 15);
        CHECK(bij && s[0] == 0 && s[1] == graph::scrambled_label(1, 15) && d[0] == s[4] && d[2] == d[3]);
    }
    auto x = make_shared<tensor<float>>(vector<size_t>{15, 10}, 0.5f);
    graph::Data data(x, ei.get());
    CHECK(data.num_nodes() == 15 && data.num_node_features() == 10 && data.num_edges() == 8);
    auto x4 = make_shared<tensor<float>>(vector<size_t>{4, 10}, 0.5f);
    CHECK(throws_with([&] { graph::Data d(x4, ei.get()); },
                      "invalid input, max value in edge_index should be less than the number of nodes from x"));
    CHECK(throws_with([&] { graph::Data d(make_shared<tensor<float>>(vector<size_t>{4}, 0.5f), ei.get()); },
                      "invalid input for x, must be 2D"));

    printf(failures ? "HOST_API_CPU_FAILED %d\n" : "HOST_API_CPU_OK\n", failures);
    return failures ? 1 : 0;
}
