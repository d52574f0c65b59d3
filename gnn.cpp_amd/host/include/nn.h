// nn.h -- nn::Module registry and nn::Linear of the reference API (reference include/nn.h:28-73,
// src/nn.cpp:12-211) over the MI355X tensor backend.  Only what the GCN hot path touches is implemented:
// the module / parameter registry GCNConv relies on (`get_module("lin")`, `get_parameter("bias")`,
// reference graph.cpp:173,188) and Linear (the dense X.W^T step).  BatchNorm / ReLU / Dropout are registered by
// GCNConv like in the reference but are "next" rows of SURVEY.md section 8(f): their forward throws here.
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
    cyg::tptr<float> operator()(const cyg::tptr<float> &input_tensor) { return forward(input_tensor); }
    virtual cyg::tptr<float> forward(const cyg::tptr<float> &) { throw std::runtime_error("not implemented"); }
    std::vector<std::shared_ptr<Module>> modules(const bool &recurse = true);
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

// Registered by GCNConv exactly like the reference does (graph.cpp:163-165); not on the hot path.
class NotOnHotPath : public Module {
public:
    explicit NotOnHotPath(const std::string &n) : Module(n) {}
    cyg::tptr<float> forward(const cyg::tptr<float> &) override
    {
        throw std::runtime_error(name + ": not implemented by the MI355X backend yet (SURVEY.md section 8(f) 'next' row)");
    }
};
class BatchNorm : public NotOnHotPath {
public:
    explicit BatchNorm(size_t num_features, float = 1e-5f, float = 0.1f) : NotOnHotPath("BatchNorm"), _num_features(num_features) {}
    size_t _num_features;
};
class ReLU : public NotOnHotPath {
public:
    ReLU() : NotOnHotPath("ReLU") {}
};
class Dropout : public NotOnHotPath {
public:
    explicit Dropout(float p = 0.2f) : NotOnHotPath("Dropout"), _p(p) {}
    float _p;
};

}  // namespace nn

#endif
