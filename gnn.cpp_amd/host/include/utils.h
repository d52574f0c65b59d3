// utils.h -- error conventions and shape checks of the cyg / nn / graph API, MI355X backend.
//
// Mirrors the observable behaviour of the reference's include/utils.h + src/utils.cpp for the hot path:
// every failure is a std::runtime_error carrying one of the reference's message texts (callers and the
// reference's tests match on them: reference tests/tensor.test.cpp:28,38,69; texts utils.h:19-30), and the
// CHECK_* helpers reject the same inputs (utils.cpp:8-78).  Written from scratch; nothing here computes.
#ifndef GNNCPP_AMD_UTILS_H
#define GNNCPP_AMD_UTILS_H

#include <algorithm>
#include <climits>
#include <cstddef>
#include <cstdlib>
#include <functional>
#include <numeric>
#include <stdexcept>
#include <string>
#include <valarray>
#include <vector>

// message texts: part of the API contract (compared verbatim by callers)
inline constexpr char ERROR_GRAD_DTYPE[] = "Only Tensors of floating point dtype can require gradients";
inline constexpr char WARNING_GRAD_NOT_LEAF[] =
    "UserWarning: The .grad attribute of a Tensor that is not a leaf Tensor is being accessed. Its .grad attribute won't be "
    "populated during autograd.backward()";
inline constexpr char ERROR_IN_PLACE_OP_LEAF[] =
    "RuntimeError: a leaf Variable that requires grad is being used in an in-place operation.";
inline constexpr char ERROR_SIZE_MISMATCH[] =
    "tensors must be of same shape/size - mismatch between number of elements and dimension of tensor";
inline constexpr char ERROR_RANK_MISMATCH[] = "tensors are of different ranks";
inline constexpr char ERROR_OUT_OF_RANGE[] = "out of bound range";
inline constexpr char ERROR_INVALID_DIMS[] = "dims cannot be empty or zero";
inline constexpr char ERROR_NON_SCALAR_BACKPROP[] = "pass in tensor to backprop on non-scalar tensor";
inline constexpr char ERROR_MM_COMPATIBLE[] = "tensors are not compatible, tensors should of shape [...,A,B] and [...,B,A]";
inline constexpr char ERROR_OUT_OF_BOUND_DIM[] = "dim is out of range";
inline constexpr char ERROR_GRAD_MISMATCH[] = "size mismatch, incoming gradient must be same dimension with tensor";
inline constexpr char ERROR_TRANSPOSE[] = "invalid inp";
// backend-specific (no counterpart in the reference, which is CPU only)
inline constexpr char ERROR_BACKEND_UNSUPPORTED[] =
    "operation/shape is outside the GCN hot path implemented by the MI355X backend (no CPU fallback)";

inline size_t numel_of(const std::vector<size_t> &dims)
{
    return std::accumulate(dims.begin(), dims.end(), (size_t)1, std::multiplies<size_t>());
}

inline void CHECK_VALID_DIMS(const std::vector<size_t> &dims)
{
    if (dims.empty() || *std::min_element(dims.begin(), dims.end()) < 1) throw std::runtime_error(ERROR_INVALID_DIMS);
}

inline void CHECK_RANK(const std::vector<size_t> &a, const std::vector<size_t> &b)
{
    if (a.size() != b.size()) throw std::runtime_error(ERROR_RANK_MISMATCH);
}

inline void CHECK_SIZE(const std::vector<size_t> &dims, size_t n_elements)
{
    if (numel_of(dims) != n_elements) throw std::runtime_error(ERROR_SIZE_MISMATCH);
}

inline bool is_broadcastable(const std::vector<size_t> &a, const std::vector<size_t> &b)
{
    for (size_t i = 1; i <= std::min(a.size(), b.size()); i++) {
        size_t x = a[a.size() - i], y = b[b.size() - i];
        if (std::min(x, y) != 1 && x != y) return false;
    }
    return true;
}

inline void CHECK_ARGS_OPS_BROADCAST(const std::vector<size_t> &a, const std::vector<size_t> &b)
{
    if (!is_broadcastable(a, b)) throw std::runtime_error(ERROR_SIZE_MISMATCH);
}

inline void CHECK_MM_DIMS(const std::vector<size_t> &l, const std::vector<size_t> &r)
{
    // [..., a, b] . [..., b, c]: inner dimensions must agree
    if (l.size() < 2 || r.size() < 2 || l[l.size() - 1] != r[r.size() - 2]) throw std::runtime_error(ERROR_MM_COMPATIBLE);
}

inline void CHECK_VALID_RANGE(int dim, int rank, int low = 0)
{
    if (dim != INT_MAX && (dim >= rank || dim < low)) throw std::runtime_error(ERROR_OUT_OF_BOUND_DIM);
}

inline void CHECK_EQUAL_SIZES(const std::vector<size_t> &a, const std::vector<size_t> &b)
{
    if (a != b) throw std::runtime_error("invalid op, tensors sizes must be the same");
}

inline void CHECK_TRANSPOSE(const std::vector<size_t> &s, int a, int b)
{
    const int n = (int)s.size();
    if (a < -n || a >= n || b < -n || b >= n || std::abs(a - b) != 1) throw std::runtime_error(ERROR_TRANSPOSE);
}

// new std::valarray<T>(value, prod(dims)) -- the buffer a tensor constructor adopts (reference utils.h:150-158)
template <class T>
std::valarray<T> *initialize(const std::vector<size_t> &dims, T value = 0)
{
    CHECK_VALID_DIMS(dims);
    return new std::valarray<T>(value, numel_of(dims));
}

namespace cyg {
namespace detail {
// C-ABI status -> exception (message from gnnx_last_error, or the reference text for the known cases)
void gx(int status, const char *where);
// the stream every tensor op of this thread is enqueued on (NULL = default stream)
void *current_stream();
void set_current_stream(void *stream);
// grow-only device scratch for workspaces (csr build, split-K slabs, colsum partials)
void *workspace(size_t bytes);
// Size-bucketed caching allocator for tensor storage: every op result is a fresh tensor (reference semantics), and a
// hipMalloc/hipFree pair per op costs milliseconds plus a device-wide sync at these sizes.  Blocks go back to a
// free list keyed by their (256-B rounded) size and are reused by the next tensor of that size; all work is on one
// in-order stream, so reuse needs no event.  cyg::empty_cache() returns the cached blocks to the driver.
void *dev_alloc(size_t bytes);
void dev_free(void *ptr, size_t bytes);
}  // namespace detail
void empty_cache();
}  // namespace cyg

#endif
