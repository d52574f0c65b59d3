// nn.h -- nn::Module registry, nn::Linear, nn::BatchNorm and nn::ReLU of the reference API (reference
// include/nn.h:28-123, src/nn.cpp:12-330) over the MI355X tensor backend.  Only what GCNConv touches is implemented:
// the module / parameter registry (`get_module("lin")`, `get_parameter("bias")`, reference graph.cpp:173,188),
// Linear (the dense X.W^T step) and the BatchNorm + ReLU pair between transform and aggregation (graph.cpp:174-175).
#ifndef GNNCPP_AMD_NN_H
#define GNNCPP_AMD_NN_H

#include <memory>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "tensor.h"

namespace nn {

class Module : public std::enable_shared_from_this<Module> {
public:
    explicit Module(std::string n = "Module") : name(std::move(n)) {}
    virtual ~Module() = default;  // the reference's is non-virtual (nn.h:56); modules are deleted through the base here
    bool training = true;
    std::string name;

    // takes ownership of `module` (reference nn.h:34-37)
    void register_module(std::string name, Module *module);
    void register_parameter(std::string name, cyg::tptr<float> p);
    void register_buffer(std::string name, cyg::tptr<float> p);
    void zero_grad();
    void eval() { train(false); }
    void train(const bool &isTrain = true);
    cyg::tptr<float> get_parameter(std::string name);
    cyg::tptr<float> get_buffer(std::string name);
    std::shared_ptr<Module> get_module(std::string name);
    // (*module)(x) -> forward(x); with a label tensor -> forward(x, y)   (reference nn.h:46-48, nn.cpp:71-76)
    cyg::tptr<float> operator()(const cyg::tptr<float> &input_tensor, cyg::tensor<int> *y = nullptr)
    {
        return y == nullptr ? forward(input_tensor) : forward(input_tensor, y);
    }
    virtual cyg::tptr<float> forward(const cyg::tptr<float> &) { throw std::runtime_error("not implemented"); }
    virtual cyg::tptr<float> forward(const cyg::tptr<float> &, cyg::tensor<int> *) { throw std::runtime_error("not implemented"); }
    std::vector<std::shared_ptr<Module>> modules(const bool &recurse = true);
    // flat name -> module over the descendants; a module without children (or recurse == false) maps its own name to
    // itself; a colliding key is stored as "<child>_<key>" (reference nn.cpp:87-102)
    std::unordered_map<std::string, std::shared_ptr<Module>> named_modules(const bool &recurse = true);
    std::vector<cyg::tptr<float>> parameters(const bool &recurse = true);
    // flat name -> tensor over this module and its descendants; a child's key that already exists is stored
    // as "<child>_<key>" (reference nn.cpp:110-125)
    std::unordered_map<std::string, cyg::tptr<float>> named_parameters(const bool &recurse = true);
    std::unordered_map<std::string, cyg::tptr<float>> named_buffers(const bool &recurse = true);

protected:
    std::vector<std::pair<std::string, std::shared_ptr<Module>>> _modules;
    std::unordered_map<std::string, cyg::tptr<float>> _parameters;
    std::unordered_map<std::string, cyg::tptr<float>> _buffers;
};

// y = x . W^T (+ b); W is [out, in], initialised U(-1/sqrt(in), 1/sqrt(in)) (reference nn.cpp:187-211)
class Linear : public Module {
public:
    Linear(const size_t &in_features, const size_t &out_features, const bool &bias = true, const std::string &n = "Linear");
    void reset_parameters();
    cyg::tptr<float> forward(const cyg::tptr<float> &input_tensor) override;

    bool _bias;
    size_t _in_features, _out_features;
};

// BatchNorm over the node dimension with batch statistics, y = ((x - mean) / (var + eps)^0.5) * gammas + betas
// (reference nn.cpp:285-330).  Like the reference, the running statistics are registered but never updated by
// forward (the reference assigns them to temporaries, nn.cpp:323-324); in eval mode they are what normalises.
// Backward is the mathematically correct one (the reference's own drops fan-in gradients, operation.h:82-86).
class BatchNorm : public Module {
public:
    explicit BatchNorm(const size_t &num_features, const float &eps = 1e-05, const float &momentum = 0.1, const bool &affine = true,
                       const bool &track_running_stats = true, const std::string &n = "BatchNorm");
    cyg::tptr<float> forward(const cyg::tptr<float> &x) override;
    bool uses_batch_stats() const { return training || !_tracking_running_stats; }

    size_t _num_features;
    float _eps, _momentum;
    bool _affine, _tracking_running_stats;
};

// gnnx_bn_relu_bwd_f32, or its reference-quirk form when GNNCPP_REFERENCE_QUIRKS is set (see host.cpp)
decltype(&gnnx_bn_relu_bwd_f32) bn_backward_fn();

// max(0, x) as a select (reference nn.cpp:229-237 -> functional.h:443-470)
class ReLU : public Module {
public:
    explicit ReLU(const std::string &n = "ReLU") : Module(n) {}
    cyg::tptr<float> forward(const cyg::tptr<float> &input_tensor) override;
};

// Registered by GCNConv like in the reference (graph.cpp:164) and, like there, never applied (graph.cpp:170-191).
class Dropout : public Module {
public:
    explicit Dropout(float p = 0.2f, const std::string &n = "Dropout") : Module(n), _p(p) {}
    cyg::tptr<float> forward(const cyg::tptr<float> &) override
    {
        throw std::runtime_error("Dropout: not implemented by the MI355X backend (never applied on the GCN path)");
    }
    float _p;
};

// mean softmax cross-entropy of [N,C] logits against [N] class ids; forward arithmetic of the reference
// (nn.cpp:442-453: -log(exp(x_t) / (sum_c exp(x_c) + 1e-20)), no max-subtraction); backward (softmax - onehot)/N
// (the reference's own backward throws).
cyg::tptr<float> cross_entropy_loss(const cyg::tptr<float> logits, const cyg::tptr<int> target);
// (addition) a shard's share of the loss over a batch of n_total rows (1-D vertex partition): the value is this rank's term of the
// mean -- sum the ranks' values for the loss -- and the gradient carries 1 / n_total, as the unsharded call on all rows would
cyg::tptr<float> cross_entropy_loss(const cyg::tptr<float> logits, const cyg::tptr<int> target, size_t n_total);

// Optimisers over device-resident parameters (reference nn.h:156-191).  SGD is the textbook update
// p -= lr * (g + weight_decay * p) (+ momentum buffers); the reference's step() reads an empty velocity vector
// (nn.cpp:414) and cannot run, so there is nothing to be bit-compatible with.
class Optimizer {
public:
    explicit Optimizer(std::vector<cyg::tptr<float>> parameters) : _parameters(std::move(parameters)) {}
    void zero_grad();
    std::vector<cyg::tptr<float>> _parameters;
};

class SGD : public Optimizer {
public:
    SGD(std::vector<cyg::tptr<float>> parameters, float lr, float momentum = 0, float dampening = 0, float weight_decay = 0,
        bool nestorov = false)
        : Optimizer(std::move(parameters)), _lr(lr), _dampening(dampening), _momentum(momentum), _weight_decay(weight_decay),
          _nestorov(nestorov)
    {
    }
    void step();
    float _lr, _dampening, _momentum, _weight_decay;
    bool _nestorov;
};

}  // namespace nn

#endif
