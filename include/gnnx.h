/*
 * gnnx.h -- C-ABI of the MI355X (gfx950) backend for the walexi/gnn.cpp GCN hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI: its device hook is
 * the empty include/device.cuh + the commented call `device::add(out_data, lhs_data, rhs_data)`
 * (reference include/functional.h:174,180) -- free functions on raw buffers called from
 * functional::<op>.  These entry points sit exactly there: plain pointers and sizes, int status
 * returns, an explicit stream, no C++/torch types.  The C++ mirror of the reference API
 * (gnn.cpp_amd/host/) and the ctypes binding (gnn.cpp_amd/capi.py) are the two callers.
 *
 * Conventions
 *   - every pointer named d_* is DEVICE memory (hipMalloc'd, e.g. torch.Tensor.data_ptr() or gnnx_malloc);
 *   - dense matrices are row-major fp32 with an explicit leading dimension in ELEMENTS (ld >= cols),
 *     exactly the reference's contiguous std::valarray<float> layout (reference include/tensor.h:825)
 *     when ld == cols;
 *   - CSR indices are int32 (rowptr has n_rows+1 entries; nnz < 2^31), columns ASCENDING in a row
 *     (the row-major scan order of reference src/graph.cpp:52-60);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are asynchronous on
 *     it unless stated; nothing is allocated or synchronised inside a compute call;
 *   - return 0 (GNNX_OK) or a negative gnnx_status; gnnx_last_error() gives the message of the
 *     calling thread's last failure.  The C++ wrapper maps them to std::runtime_error with the
 *     reference's ERROR_* texts (reference include/utils.h:19-30).
 *   - threads: every entry point may be called from any host thread; a call works on the stream it is given and keeps no state of
 *     its own between calls beyond what its handles (plans, communicators) own -- one thread drives one handle at a time.  Host
 *     threads that run ranks of an in-process group each use a stream of their OWN (gnnx_stream_create; the C++ layer does this
 *     by itself), never the NULL stream; what crosses threads is ordered by the collectives (gnnx_comm.hip) and nothing else
 *     needs to be.  Process-wide state is immutable once built (the libm table of the degree block, kernel attribute opt-ins).
 */
#ifndef GNNX_H
#define GNNX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNNX_VERSION 100

typedef enum gnnx_status {
    GNNX_OK = 0,
    GNNX_ERR_INVALID_ARG = -1,  /* null pointer / negative size / ld < cols / misaligned */
    GNNX_ERR_SHAPE = -2,        /* operand shapes do not agree (reference CHECK_MM_DIMS, utils.cpp:8-78) */
    GNNX_ERR_INDEX_RANGE = -3,  /* an edge endpoint is outside [0, n_nodes) (reference graph.cpp:89) */
    GNNX_ERR_WORKSPACE = -4,    /* caller-provided workspace too small */
    GNNX_ERR_HIP = -5,          /* a HIP runtime call failed (message in gnnx_last_error) */
    GNNX_ERR_NO_DEVICE = -6,    /* no gfx950 device visible */
    GNNX_ERR_UNSUPPORTED = -7
} gnnx_status;

int gnnx_version(void);
const char *gnnx_status_string(int status);
const char *gnnx_last_error(void);

/* ------------------------------------------------------------------ runtime plumbing ------------- */
/* Thin HIP wrappers so that host code above this ABI needs no HIP headers. */
int gnnx_device_count(int *count);
int gnnx_set_device(int device);
int gnnx_device_name(int device, char *buf, size_t buflen);
int gnnx_malloc(void **d_ptr, size_t bytes);
int gnnx_free(void *d_ptr);
int gnnx_memset(void *d_ptr, int value, size_t bytes, void *stream);
int gnnx_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream);
int gnnx_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream); /* synchronises `stream` */
int gnnx_memcpy_d2d(void *d_dst, const void *d_src, size_t bytes, void *stream);
/* rows x width_bytes between two pitched device buffers (stream-ordered) */
int gnnx_memcpy2d_d2d(void *d_dst, size_t dst_pitch_bytes, const void *d_src, size_t src_pitch_bytes, size_t width_bytes, size_t rows,
                      void *stream);
int gnnx_stream_create(void **stream);
int gnnx_stream_destroy(void *stream);
int gnnx_stream_sync(void *stream);
int gnnx_device_sync(void);
int gnnx_event_create(void **event);
int gnnx_event_destroy(void *event);
int gnnx_event_record(void *event, void *stream);
int gnnx_event_sync(void *event);
int gnnx_event_elapsed_ms(void *start, void *stop, float *ms);
int gnnx_stream_wait_event(void *stream, void *event); /* work queued on `stream` after this call waits for `event` */

/* ------------------------------------------------------------------ graph build ------------------ */
/*
 * COO [2,E] -> CSR with the reference's adjacency semantics:
 *   A[src][dst] = 1 by assignment  => duplicate edges collapse   (reference graph.cpp:21-44, line 40)
 *   diagonal zeroed                => self loops are REMOVED     (graph.cpp:68-75 with fillValue 0, called
 *                                                                 from GCNConv::forward graph.cpp:172)
 *   row-major scan                 => entries ordered by (src,dst) (graph.cpp:46-67)
 * Replaces the dense round trip edge_to_adj_mat -> fill_diagonal_ -> adj_to_edge_list.
 * Pass (d_dst, d_src) to get the CSR of A^T (what MatMul::_backward's dense transpose,
 * reference operation.h:524-527, becomes).
 *
 * d_rowptr: n_nodes+1 int32.  d_colidx: capacity n_edges int32.  *nnz_out: host int64, valid on return
 * (this call synchronises `stream`).  Workspace: gnnx_csr_from_coo_workspace() bytes of device memory.
 * flags: bit0 keep self loops, bit1 keep duplicates (both 0 = reference semantics).
 * Errors: GNNX_ERR_INDEX_RANGE if any endpoint is outside [0,n_nodes) (the reference would write out
 * of bounds at graph.cpp:40; its Data ctor throws at graph.cpp:89).
 */
#define GNNX_CSR_KEEP_SELF_LOOPS 1u
#define GNNX_CSR_KEEP_DUPLICATES 2u
int gnnx_csr_from_coo_workspace(int64_t n_edges, int32_t n_nodes, size_t *bytes);
int gnnx_csr_from_coo(const int32_t *d_src, const int32_t *d_dst, int64_t n_edges, int32_t n_nodes, uint32_t flags,
                      int32_t *d_rowptr, int32_t *d_colidx, int64_t *nnz_out, void *d_workspace,
                      size_t workspace_bytes, void *stream);

/* The SpMM entry points TRUST the CSR they are given (column ids index X without a bounds check: a check per gathered row would
 * sit in the hottest loop).  A CSR made by gnnx_csr_from_coo / gnnx_halo_plan_create is valid by construction; validate one from
 * anywhere else once, before its first use: rowptr[0] == 0, monotone, every column id in [0, n_cols).  Synchronises `stream`;
 * GNNX_ERR_INDEX_RANGE otherwise. */
int gnnx_csr_validate(const int32_t *d_rowptr, const int32_t *d_colidx, int32_t n_rows, int32_t n_cols, void *stream);

/* *equal_out = 1 iff the two device arrays hold the same n int32 values (synchronises `stream`).  The host layer compares an
 * incoming edge_index with the copy its cached adjacency was built from, so an edge list edited in place -- or a new one that
 * happens to be allocated at the old address -- never hits a stale CSR / norm. */
int gnnx_equal_i32(const int32_t *d_a, const int32_t *d_b, int64_t n, int *equal_out, void *stream);

/* Weighted adjacency (edge_attr, reference graph.h:35): A[r][c] = w by assignment, so of duplicate (r, c) pairs the LAST one
 * in the list wins (edge_to_adj_mat, graph.cpp:38-40).  diag_mode:
 *   GNNX_DIAG_KEEP   self loops stay as given                         (edge_to_adj_mat alone)
 *   GNNX_DIAG_STRIP  self loops are removed                           (add_self_loops(..., fillValue 0), graph.cpp:72)
 *   GNNX_DIAG_FILL   every (i, i) becomes diag_value, given or not    (add_self_loops(..., fillValue v))
 * flags: GNNX_CSR_KEEP_DUPLICATES as above; GNNX_CSR_DROP_TRUNCATED_ZERO drops entries whose value truncates to the integer
 * 0 -- adj_to_edge_list's `int(data[i]) != 0` test (graph.cpp:54), through which |w| < 1 vanishes on the
 * add_self_loops round trip; without it explicit zeros stay as entries (they add +-0 in a product, like the dense matrix).
 * Output: rowptr[N+1], colidx[nnz], vals[nnz] sorted by (row, column); capacity n_edges (+ n_nodes with GNNX_DIAG_FILL). */
enum { GNNX_DIAG_KEEP = 0, GNNX_DIAG_STRIP = 1, GNNX_DIAG_FILL = 2 };
#define GNNX_CSR_DROP_TRUNCATED_ZERO 4u
int gnnx_csr_from_coo_weighted_workspace(int64_t n_edges, int32_t n_nodes, size_t *bytes);
int gnnx_csr_from_coo_weighted(const int32_t *d_src, const int32_t *d_dst, const float *d_weights, int64_t n_edges, int32_t n_nodes,
                               uint32_t flags, int diag_mode, float diag_value, int32_t *d_rowptr, int32_t *d_colidx, float *d_vals,
                               int64_t *nnz_out, void *d_workspace, size_t workspace_bytes, void *stream);

/*
 * Degree / symmetric-normalisation block of GCNConv::forward (reference graph.cpp:177-185):
 *   deg_i = 1 + sum_j A_ij        (adj_mat->sum(-1,true) + 1;  the "+1" stays although self loops were removed)
 *   s_i   = deg_i ^ (-1/2)        (deg->pow(-0.5))
 *   norm_i = s_i * sum_j A_ij s_j (adj_mat->mm(deg); norm *= deg), inner sum in the reference's matmul
 *            order (descending j, see DESIGN.md "summation order").
 * d_s and d_norm: n_rows fp32 each (the reference's [N,1] tensors).  Either may be NULL.
 * For a row block of a sharded graph pass d_s_cols (the s values indexed by COLUMN id, i.e. [local|halo])
 * and d_s is written for the block's own rows only; with d_s_cols == NULL columns index d_s itself.
 * s written here is LOOKED UP in a table of the host libm's powf(k, -0.5f), k = 1 .. 1 + max degree -- functional.h:253 is that call,
 * and glibc's powf is 1 ulp away from the correctly rounded value for 9 685 of the 2^24 degrees (the smallest 1058) -- so s, norm and
 * both aggregations carry the reference's bits at every size (test_headline_config_whole_graph_vs_oracle).  Writing d_s costs one
 * host synchronisation (the maximum degree); a caller may still pass its own s through d_s_cols with d_s == NULL.
 */
int gnnx_degree_norm_f32(const int32_t *d_rowptr, const int32_t *d_colidx, int32_t n_rows, float *d_s,
                         const float *d_s_cols, float *d_norm, void *stream);

/* ------------------------------------------------------------------ hot path: aggregation -------- */
/*
 * CSR SpMM with fused prologue/epilogue -- replaces functional::matmul on the dense N x N adjacency
 * (reference functional.h:399-441 called from graph.cpp:208), the broadcast multiply by norm
 * (graph.cpp:209 -> functional.h:190-213) and the bias add (graph.cpp:188 -> functional.h:163-187):
 *
 *   Y[i,:] = beta * Y[i,:] + rowscale[i] * ( sum_{p in row i, DESCENDING column} vals[p] * colscale[c_p] * X[c_p,:] ) + bias
 *
 *   forward  (graph.cpp:204-212,188):  vals=NULL colscale=NULL rowscale=norm bias=bias|NULL  on CSR(A)
 *   backward (operation.h:144-167 then :524-531):  dH = A^T . (norm (.) G):
 *                                      vals=NULL colscale=norm rowscale=NULL bias=NULL      on CSR(A^T)
 *   Mode SYM (textbook D^-1/2 A D^-1/2): colscale=s rowscale=s.
 * Every product/add is separately rounded fp32 in the reference's order -- ONE accumulator per output element, whatever the
 * row's degree -- so the result is bit-identical to the reference CPU path (modulo the sign of 0), with or without a plan.
 * vals, colscale, rowscale, bias may each be NULL.  beta is 0 or 1 (1 = the reference's `_grad +=`,
 * tensor.h:268-271).  X: [n_cols, F] ld ldx.  Y: [n_rows, F] ld ldy.  X and Y must not alias.
 *
 * `plan` (optional, from gnnx_spmm_plan_create) load-balances power-law rows without touching the arithmetic: rows longer
 * than `chunk` (the hub rows) are summed by wavefronts of their own -- one per (row, 64-feature slab), the neighbour rows in
 * flight in an LDS ring, still one accumulator per feature in descending column order (functional.h:433-439) -- and the other
 * rows are cut into non-zero-balanced blocks.  Planned and unplanned results are the same bits; the plan only changes the
 * time (RMAT 10M / 100M, F = 256: 18.0 -> 13.6 ms).  max_feat is ignored (kept for source compatibility).
 * gnnx_spmm_plan_info: number of hub rows and their non-zeros.
 */
typedef struct gnnx_spmm_plan gnnx_spmm_plan; /* opaque, device-resident work list */
int gnnx_spmm_plan_create(const int32_t *d_rowptr, int32_t n_rows, int32_t chunk, int32_t max_feat,
                          gnnx_spmm_plan **plan, void *stream);
int gnnx_spmm_plan_destroy(gnnx_spmm_plan *plan);
/* GNNX_OK, or GNNX_ERR_HIP when a completed launch of the plan's producer / consumer kernel gave up one of its bounded waits (the
 * rows it owned were then NOT written: nothing is ever summed from a slot that did not land).  One read of pinned host memory, no
 * synchronisation; every planned aggregation call makes the same check on entry and returns the error instead of launching. */
int gnnx_spmm_plan_status(const gnnx_spmm_plan *plan);
int gnnx_spmm_plan_info(const gnnx_spmm_plan *plan, int64_t *n_hub_rows, int64_t *n_hub_nnz);
/* The longest hub rows are summed by the producer / consumer hub kernel -- a CU per (row, 64-feature slab): one wavefront adds in
 * the reference's order, three keep the row's slices coming through a 128 KiB LDS ring -- because their time is their own chain of
 * dependent adds, not their bytes.  By default the library picks them per call (rows whose chain in the plain hub kernel would
 * exceed half of what all hub rows' bytes take: a handful on a whole 10 M / 100 M graph, the rows beyond ~18 k non-zeros on an
 * eighth of it).  threshold >= 0 fixes the cut (rows longer than it; tests pass 0: every hub row), < 0 restores the default.  Same
 * bits for every choice. */
int gnnx_spmm_plan_set_big_row_threshold(gnnx_spmm_plan *plan, int32_t threshold);
/* *structured = 1 when the plan's hub rows sit on vertex ids with few one-bits (non-zero-weighted mean popcount well below half the id
 * width) and are not one dense block of consecutive ids -- what R-MAT and other generators that draw the bits of an id independently
 * produce; ids spread at random, or hubs sorted to the front, do not.  The hub rows of CSR(A) are the rows the aggregation over
 * CSR(A^T) gathers most, and vice versa: with such ids the gathered matrix wants the padded row pitch of gnnx_gather_row_stride. */
int gnnx_spmm_plan_hub_ids_structured(const gnnx_spmm_plan *plan, int *structured);

int gnnx_spmm_csr_f32(int32_t n_rows, int32_t n_cols, int32_t n_feat, const int32_t *d_rowptr,
                      const int32_t *d_colidx, const float *d_vals, const float *d_colscale,
                      const float *d_rowscale, const float *d_bias, const float *d_X, int64_t ldx, float beta,
                      float *d_Y, int64_t ldy, const gnnx_spmm_plan *plan, void *stream);

/* The same SpMM with the modules that GCNConv::forward runs either side of the aggregation folded in, so the
 * normalised / rectified activations never make their own trip through HBM:
 *   prologue, applied to every gathered row of X before it is added (forward mode only: vals = colscale = NULL):
 *     bn_mean/bn_var != NULL : x <- ((x - mean) / (var + eps)^0.5) * gamma + beta   (nn.cpp:285-330 BatchNorm::forward with
 *                              the batch statistics of gnnx_bn_stats_f32; gamma / beta NULL = 1 / 0)
 *     relu_in  != 0          : x <- where(x > 0, x, 0)                              (nn.cpp:229-237, graph.cpp:174-175)
 *   epilogue: relu_out != 0  : Y <- where(Y > 0, Y, 0) after rowscale / bias / beta (the ReLU between two stacked layers).
 * Separately rounded ops in the order of gnnx_bn_relu_fwd_f32: fused and unfused results are the same bits.
 * fusion == NULL is gnnx_spmm_csr_f32. */
typedef struct gnnx_spmm_fusion {
    const float *bn_mean, *bn_var, *bn_gamma, *bn_beta; /* [F] device pointers, or NULL */
    float bn_eps;
    int relu_in;
    int relu_out;
} gnnx_spmm_fusion;
int gnnx_spmm_csr_fused_f32(int32_t n_rows, int32_t n_cols, int32_t n_feat, const int32_t *d_rowptr,
                            const int32_t *d_colidx, const float *d_vals, const float *d_colscale,
                            const float *d_rowscale, const float *d_bias, const float *d_X, int64_t ldx, float beta,
                            float *d_Y, int64_t ldy, const gnnx_spmm_fusion *fusion, const gnnx_spmm_plan *plan,
                            void *stream);

/* Backward aggregation of a layer whose forward fused BatchNorm + ReLU into the gather (gnnx_spmm_csr_fused_f32): dY = A^T . (vals (.) G)
 * on CSR(A^T) as gnnx_spmm_csr_f32 computes it (same bits), PLUS the two column sums BatchNorm's backward needs before it can
 * produce dX -- dbeta = sum_i g_i and dgamma = sum_i g_i xhat_i with g = dY where relu(BN(h)) > 0 (the mask recomputed from H
 * with the forward's arithmetic; relu = 0: g = dY), xhat = (h - mean) * (var + eps)^-1/2 -- accumulated by the wavefronts that
 * store the rows of dY (the row of dY comes back from L2, the row of H is the one extra read) and reduced in a fixed order: the
 * separate sums pass over dY and H (gnnx_bn_relu_bwd_sums_f32: 8 F bytes per node) disappears.  Follow with
 * gnnx_bn_relu_bwd_apply_f32.  Same summands as gnnx_bn_relu_bwd_sums_f32 in another (fixed) order: rounding-level agreement.
 * Shapes: n_feat % 4 == 0, n_feat > 64, 16-byte aligned rows; GNNX_ERR_UNSUPPORTED otherwise (use the two separate calls). */
int gnnx_spmm_csr_bn_sums_workspace(int32_t n_rows, int32_t n_feat, const gnnx_spmm_plan *plan, size_t *bytes);
int gnnx_spmm_csr_bn_sums_f32(int32_t n_rows, int32_t n_cols, int32_t n_feat, const int32_t *d_rowptr, const int32_t *d_colidx,
                              const float *d_vals, const float *d_G, int64_t ldg, float *d_dY, int64_t ldy, const float *d_H, int64_t ldh,
                              const float *d_mean, const float *d_var, float eps, const float *d_gamma, const float *d_beta, int relu,
                              float *d_dgamma, float *d_dbeta, void *d_workspace, size_t workspace_bytes, const gnnx_spmm_plan *plan,
                              void *stream);

/* Opt-in bf16 FEATURE STORAGE: the same SpMM gathering rows of X stored as bf16 -- 2 bytes per
 * feature instead of 4, i.e. about half the algorithmic bytes of the aggregation -- widened exactly to f32 in registers and
 * accumulated in f32 in the same order.  NOT the parity path: rounding X to bf16 (gnnx_f32_to_bf16, round to nearest even)
 * costs up to 2^-8 relative per element, far outside the 1e-5 bar; if X is exactly representable in bf16 the result is the f32
 * path's, bit for bit.  No fusion with this entry.  ldx in bf16 elements. */
int gnnx_f32_to_bf16(const float *d_X, int64_t ldx, int64_t n_rows, int32_t n_cols, uint16_t *d_Y_bf16, int64_t ldy, void *stream);
int gnnx_spmm_csr_bf16_f32(int32_t n_rows, int32_t n_cols, int32_t n_feat, const int32_t *d_rowptr, const int32_t *d_colidx,
                           const float *d_vals, const float *d_colscale, const float *d_rowscale, const float *d_bias,
                           const uint16_t *d_X_bf16, int64_t ldx, float beta, float *d_Y, int64_t ldy,
                           const gnnx_spmm_plan *plan, void *stream);
/* H = X . W^T (X [M,K], W [N,K]: the layer's transform, nn.cpp:205-211) written as bf16 by the product's own epilogue -- the same
 * round-to-nearest-even as gnnx_f32_to_bf16, so the result equals gnnx_gemm_f32 followed by gnnx_f32_to_bf16 bit for bit, without
 * the 4 M N-byte f32 round trip.  LDS-DMA kernel shapes only (K % 64 == 0, N % 4 == 0, N >= 64, M >= 2048, 16-byte aligned rows):
 * GNNX_ERR_SHAPE otherwise.  ldh in bf16 elements. */
int gnnx_gemm_nt_bf16out_workspace(int64_t M, int64_t N, int64_t K, size_t *bytes);
int gnnx_gemm_nt_bf16out_f32(int64_t M, int64_t N, int64_t K, const float *d_X, int64_t ldx, const float *d_W, int64_t ldw,
                             uint16_t *d_H_bf16, int64_t ldh, void *d_workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ hot path: transform ---------- */
/*
 * fp32 GEMM on the MFMA units (v_mfma_f32_32x32x2_f32, exact f32) -- replaces functional::matmul for the
 * dense products of the path (reference functional.h:399-441):
 *   C[M,N] = alpha * op(A)[M,K] . op(B)[K,N] + beta * C         row-major, ld in elements
 *   forward   H  = X . W^T      (nn.cpp:205-211)        transA=0 transB=1  A=X[N,Fin]   B=W[Fout,Fin]
 *   backward  dX = dH . W       (operation.h:516-523)   transA=0 transB=0  A=dH[N,Fout] B=W[Fout,Fin]
 *             dW = dH^T . X     (operation.h:524-531 + Transpose::_backward :416-433)
 *                                                       transA=1 transB=0  A=dH[N,Fout] B=X[N,Fin]
 * The big operands (X, dH, the outputs) are never transposed in memory (the reference materialises W^T, nn.cpp:207, and a
 * clone of each operand's transpose in backward).  transA=1 (reduction over the node dimension) runs split-K over workgroups
 * into the caller's workspace and reduces slabs in a fixed order (deterministic).  Tall products with whole 256-row tiles,
 * N % 128 == 0 and K % 64 == 0 take the LDS-DMA kernels; for X . W^T those want the small W k-major, so
 * gnnx_gemm_workspace() asks for K * N floats and the call transposes W into them first (same products, same order).  Every
 * kernel computes an output element as the same k-ascending fmaf chain: which kernel ran never changes a bit.
 * Workspace: gnnx_gemm_workspace() bytes (split-K slabs for transA=1; W^T for the tall transB=1 case; else 0).  A call with
 * less workspace than that still works, on the generic kernels.
 */
int gnnx_gemm_workspace(int transA, int transB, int64_t M, int64_t N, int64_t K, size_t *bytes);
int gnnx_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const float *d_A,
                  int64_t lda, const float *d_B, int64_t ldb, float beta, float *d_C, int64_t ldc,
                  void *d_workspace, size_t workspace_bytes, void *stream);

/* The backward GEMM of a stacked layer with its two followers fused into the epilogue:
 *   C[M,N] = (A[M,K] . B[K,N]) masked by the ReLU of the layer below: C[m][n] = 0 where Ymask[m][n] <= 0   (Mask::_backward,
 *            reference operation.h:557-562; Ymask = that layer's stored forward output)
 *   colsum[n] = sum_m C[m][n]                                                 (its bias gradient: Add::_backward -> sum_to_size,
 *            operation.h:114-128, tensor.h:618-638)
 * i.e. G_{l-1} = (dH_l . W_l) (.) (Y_{l-1} > 0) and db_{l-1} in ONE pass over the output: the unmasked gradient is never written,
 * the mask pass (read Y, read G, write G) and the column-sum pass (read G) disappear.  C holds the same bits as gnnx_gemm_f32
 * followed by the mask; colsum is summed per workgroup then over workgroups in a fixed order (deterministic; a different order
 * than gnnx_colsum_f32, so equal within rounding).  Any shape: what the fused kernel does not cover runs as three plain passes. */
int gnnx_gemm_relu_colsum_workspace(int64_t M, int64_t N, int64_t K, size_t *bytes);
int gnnx_gemm_relu_colsum_f32(int64_t M, int64_t N, int64_t K, const float *d_A, int64_t lda, const float *d_B, int64_t ldb,
                              const float *d_Ymask, int64_t ldy, float *d_C, int64_t ldc, float *d_colsum, void *d_workspace,
                              size_t workspace_bytes, void *stream);

/* OPT-IN (never the default; the parity path uses gnnx_gemm_f32 + gnnx_bn_stats_f32): H = X[M,K] . W[N,K]^T together with the
 * BatchNorm batch statistics of H's columns in ONE pass over H -- SURVEY.md 2b "BN statistics as the GEMM epilogue".  H holds the
 * same bits as gnnx_gemm_f32.  mean / var (biased) come from per-lane sums of d = h - shift[n] and d^2, shift = row 0 of H (a
 * sample value, so Q/M - (S/M)^2 cancels mildly), finished in double: a single-pass variance, within rounding of -- not bit-equal
 * to -- the exact two-pass statistics (x->mean(-2), x->var(-2, 0); reference nn.cpp:303,312).  Saves the two reads of H that
 * gnnx_bn_stats_f32 makes.  Shapes the fused kernel does not cover fall back to the exact pair of calls. */
int gnnx_gemm_bn_stats_workspace(int64_t M, int64_t N, int64_t K, size_t *bytes);
int gnnx_gemm_bn_stats_f32(int64_t M, int64_t N, int64_t K, const float *d_X, int64_t ldx, const float *d_W, int64_t ldw, float *d_H,
                           int64_t ldh, float *d_mean, float *d_var, void *d_workspace, size_t workspace_bytes, void *stream);

/* OPT-IN split-precision GEMM -- never the default, not used by any parity-graded call.  C[M,N] = A[M,K] . op(B)
 * (transB: B is [N,K], else [K,N]) on the bf16 matrix cores: every f32 operand is split exactly into three bf16 pieces
 * (8 + 8 + 8 significand bits) and the six piece products with i + j <= 2 are accumulated in f32, smallest first -- f32-level
 * accuracy (error within a small factor of the f32 FMA chain's, far inside the 1e-5 bar; tests compare both with float64) at
 * several times the f32 MFMA rate, but not the reference's arithmetic.  Needs K % 16 == 0, N % 128 == 0, A 16-byte aligned
 * with lda % 4 == 0; workspace gnnx_gemm_split_workspace() bytes (the split copy of B). */
int gnnx_gemm_split_workspace(int64_t M, int64_t N, int64_t K, size_t *bytes);
int gnnx_gemm_split_bf16_f32(int transB, int64_t M, int64_t N, int64_t K, const float *d_A, int64_t lda, const float *d_B,
                             int64_t ldb, float *d_C, int64_t ldc, void *d_workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ small ops on the path -------- */
/* dbias: out[f] = beta*out[f] + sum_i G[i,f]  (Add::_backward -> sum_to_size, reference operation.h:114-128,
 * tensor.h:618-638).  Two-stage deterministic tree (fixed grid); workspace gnnx_colsum_workspace() bytes. */
int gnnx_colsum_workspace(int64_t n_rows, int32_t n_feat, size_t *bytes);
/* gnnx_colsum_copy_f32: the same sums (same bits) and, from the same pass, a copy of G's rows on another row stride: d_copy[i * ldc + f]
 * = G[i * ldg + f] -- the upstream gradient laid out for the backward aggregation's gather (gnnx_gather_row_stride).
 * gnnx_gather_row_stride: the row pitch (in floats) a matrix whose rows are GATHERED by the aggregation should be stored on:
 * n_feat, or n_feat + 64 when a row is a multiple of 512 bytes and the matrix is large -- with a power-of-two pitch the hub rows of
 * a synthetic power-law graph (vertex ids with few one-bits) pile onto a few memory channels (forward aggregation of RMAT 10 M /
 * 100 M, F = 256, vertices as generated: 17.8 ms on pitch 256, 13.9 ms on pitch 320; 13.8 ms after relabelling the vertices). */
/* gnnx_rows_to_slots_f32: the halo pack driven from the PRODUCER's side (the sharded step): d_slots[row][8] lists the
 * positions of `row` in the send buffer (packed to the front, -1 behind; a row goes to at most world - 1 <= 7 peers); every row is
 * read once and written to each of its slots: d_send[slot * ld_send + f] = d_X[row * ldx + f] -- the same send buffer the gather
 * pack (gnnx_gather_rows_f32 by the send list) fills.  d_colsum != NULL: the column sums of ALL rows from the same pass (out[f] =
 * beta * out[f] + sum_i X[i, f], the bits of gnnx_colsum_f32; workspace gnnx_colsum_workspace() bytes) -- the layer's dbias and the
 * pack of the upstream gradient in one read.  Rows of 16-byte pieces (n_feat % 4 == 0, n_feat / 4 a divisor of 256). */
int gnnx_rows_to_slots_f32(const float *d_X, int64_t ldx, int64_t n_rows, int32_t n_feat, const int32_t *d_slots, float *d_send,
                           int64_t ld_send, float *d_colsum, float beta, void *d_workspace, size_t workspace_bytes, void *stream);

/* gnnx_gemm_nt_rows_to_slots_f32: the sharded step's transform with the halo pack in the product's epilogue (the product
 * replaces nn::Linear::forward, /root/reference/src/nn.cpp:205-211, as gnnx_gemm_f32(0, 1, ...) does).  H[M][N] = X[M][K] . W[N][K]^T,
 * and every row of H listed in d_slots ([M][8], gnnx_rows_to_slots_f32's table) is ALSO stored to its send-buffer rows from the
 * registers the epilogue stores H from: the same bits in H and in d_send as gnnx_gemm_f32 followed by gnnx_rows_to_slots_f32, without
 * the pass that reads H back.  Shapes off the LDS-DMA kernel's grid (and the ragged last rows) run exactly those two calls.  Rows of
 * 16-byte pieces, as gnnx_rows_to_slots_f32.  Workspace: gnnx_gemm_nt_rows_to_slots_workspace() bytes (W^T).  As there, every listed
 * slot must be a row of d_send (the table is the caller's: gnnx_halo_plan_slot_table, or built like it) -- the kernels do not check. */
int gnnx_gemm_nt_rows_to_slots_workspace(int64_t M, int64_t N, int64_t K, size_t *bytes);
int gnnx_gemm_nt_rows_to_slots_f32(int64_t M, int64_t N, int64_t K, const float *d_X, int64_t ldx, const float *d_W, int64_t ldw,
                                   float *d_H, int64_t ldh, const int32_t *d_slots, float *d_send, int64_t ld_send,
                                   void *d_workspace, size_t workspace_bytes, void *stream);
int gnnx_gather_row_stride(int64_t n_rows, int32_t n_feat, int64_t *ld_out);
int gnnx_colsum_copy_f32(const float *d_G, int64_t ldg, int64_t n_rows, int32_t n_feat, float beta, float *d_out, float *d_copy,
                         int64_t ldc, void *d_workspace, size_t workspace_bytes, void *stream);
int gnnx_colsum_f32(const float *d_G, int64_t ldg, int64_t n_rows, int32_t n_feat, float beta, float *d_out,
                    void *d_workspace, size_t workspace_bytes, void *stream);

/* Unfused forms of the epilogues, for callers that go op by op through the tensor API:
 *   rowscale: Y[i,:] = X[i,:] * v[i]      ([N,F] (.) [N,1], reference functional.h:190-213 + utils.h:181-228)
 *   bias    : Y[i,:] = X[i,:] + b[:]      ([N,F] + [F],     reference functional.h:163-187)
 *   axpy    : y += a * x                  (tensor.h:268-271 `_grad +=`, a = 1)
 * X and Y may alias. */
int gnnx_rowscale_f32(const float *d_X, int64_t ldx, const float *d_v, int64_t n_rows, int32_t n_feat, float *d_Y,
                      int64_t ldy, void *stream);
int gnnx_bias_add_f32(const float *d_X, int64_t ldx, const float *d_b, int64_t n_rows, int32_t n_feat, float *d_Y,
                      int64_t ldy, void *stream);
int gnnx_axpy_f32(int64_t n, float a, const float *d_x, float *d_y, void *stream);

/* General 2-D broadcast of the API's elementwise operators (operator+ - * / on tensors: reference tensor.h:30-77 ->
 * functional.h:163-239, broadcast rule utils.h:181-228):  Y[r][c] = A[r*a_row_stride + c*a_col_stride] (op)
 * B[r*b_row_stride + c*b_col_stride], a stride of 0 broadcasts that dimension ([N,F] op [N,1], [N,F] op [F],
 * [N,F] op scalar, either side).  Each element is rounded once (IEEE add / sub / mul / div).  Y may alias A or B when
 * the aliased operand is not broadcast.  rowsum is the reduction that undoes a column broadcast in backward
 * (sum_to_size to [N,1], tensor.h:618-638): out[r] = sum_c X[r][c]. */
enum { GNNX_OP_ADD = 0, GNNX_OP_SUB = 1, GNNX_OP_MUL = 2, GNNX_OP_DIV = 3 };
int gnnx_binary_bcast_f32(int op, int64_t n_rows, int64_t n_cols, const float *d_A, int64_t a_row_stride, int64_t a_col_stride,
                          const float *d_B, int64_t b_row_stride, int64_t b_col_stride, float *d_Y, int64_t ldy, void *stream);
int gnnx_rowsum_f32(const float *d_X, int64_t ldx, int64_t n_rows, int32_t n_cols, float *d_out, void *stream);

/* Elementwise helpers behind the tensor API mirror (gnn.cpp_amd/host/):
 *   fill      : x[i] = value                                   (tensor(dims, value) ctor, reference tensor.h:106)
 *   pow       : y[i] = pow(x[i], e).  The reference's pow is the HOST libm's powf (functional.h:253) and its one use on the path is
 *               deg->pow(-0.5) on the degrees (graph.cpp:183): when every x[i] is an integer in [0, 2^24] the result is looked up in a
 *               table of that libm call (table[k] = powf(k, e), built on the host: one synchronisation, a graph-build call) -- the
 *               reference's bits; any other argument vector is evaluated on the device (e == -0.5: correctly rounded 1/sqrt)
 *   csr_rowsum: out[i] = sum_j A_ij = rowptr[i+1]-rowptr[i] (vals == NULL) -- adj_mat->sum(-1,true), reference
 *               graph.cpp:178 -> functional.h:267-296, without the dense N x N matrix
 *   transpose : Y[c,r] = X[r,c]  (materialises a 2-D transpose only when a caller insists on the data;
 *               the GEMM never needs it).  X and Y must not alias. */
int gnnx_fill_f32(float *d_x, int64_t n, float value, void *stream);
int gnnx_pow_f32(const float *d_x, int64_t n, float exponent, float *d_y, void *stream);
int gnnx_csr_rowsum_f32(const int32_t *d_rowptr, const float *d_vals, int32_t n_rows, float *d_out, void *stream);
int gnnx_transpose_f32(const float *d_X, int64_t ldx, int64_t n_rows, int64_t n_cols, float *d_Y, int64_t ldy, void *stream);

/* ------------------------------------------------------------------ next row: BatchNorm + ReLU ---- */
/*
 * The pair that sits between transform and aggregation inside GCNConv::forward (reference graph.cpp:174-175):
 *   BatchNorm::forward (nn.cpp:301-330, training mode): mean = x->mean(-2,true); var = x->var(-2, 0, true);
 *       y = ((x - mean) / (var + eps)->pow(0.5)) * gammas + betas     (running stats are never really updated in the
 *       reference: they are assigned to temporaries, nn.cpp:323-324)
 *   ReLU::forward (nn.cpp:229-237): where(x > 0, x, 0)
 * stats : d_mean[F], d_var[F] (biased variance).   fwd: d_mean/d_var may both be NULL (ReLU only); gamma / beta may be NULL.
 * bwd   : mathematically correct gradients (the reference's own drop fan-in contributions, operation.h:82-86):
 *         g = dY (.) (Y > 0);  dbeta = colsum g;  dgamma = colsum g (.) xhat;  dX = gamma/sigma (.) (g - dbeta/N - xhat (.) dgamma/N).
 *         d_Y (the forward output) is only read when relu != 0; d_Y == NULL with relu != 0 redoes the forward arithmetic on
 *         x (with d_beta) to find the sign -- the backward of a fused forward (gnnx_spmm_csr_fused_f32), where the rectified
 *         activations were never stored.  d_beta is read for that only.  Workspace: gnnx_bn_workspace() bytes.
 */
int gnnx_bn_workspace(int64_t n_rows, int32_t n_feat, size_t *bytes);
int gnnx_bn_stats_f32(const float *d_X, int64_t ldx, int64_t n_rows, int32_t n_feat, float *d_mean, float *d_var,
                      void *d_workspace, size_t workspace_bytes, void *stream);
int gnnx_bn_relu_fwd_f32(const float *d_X, int64_t ldx, int64_t n_rows, int32_t n_feat, const float *d_mean, const float *d_var,
                         float eps, const float *d_gamma, const float *d_beta, int relu, float *d_Y, int64_t ldy, void *stream);
int gnnx_bn_relu_bwd_f32(const float *d_X, int64_t ldx, const float *d_Y, int64_t ldy, const float *d_dY, int64_t ldd, int64_t n_rows,
                         int32_t n_feat, const float *d_mean, const float *d_var, float eps, const float *d_gamma, const float *d_beta,
                         int relu, float *d_dX, int64_t ldo, float *d_dgamma, float *d_dbeta, void *d_workspace,
                         size_t workspace_bytes, void *stream);
/* Cross-shard BatchNorm: the batch is the whole graph, a rank holds n_rows of its n_total rows.
 *   partial : d_out[f] = scale * sum_i x_if (d_mean == NULL), or scale * sum_i (x_if - d_mean[f])^2.  With scale = 1 / n_total an
 *             all-reduce of the first gives the global mean, then an all-reduce of the second (against the GLOBAL mean) the global
 *             biased variance: the exact two-pass arithmetic of gnnx_bn_stats_f32, two [F] vectors on the wire per layer.
 *   bwd_sums / bwd_apply : the two halves of gnnx_bn_relu_bwd_f32 -- local dgamma / dbeta sums, then (after the caller's
 *             all-reduce of those two [F] vectors) dX with the global sums and n_total. */
/* OPT-IN reference-quirk backward: dgamma / dbeta as above, dX = (g * gamma) / (var + eps)^0.5 with the batch statistics treated
 * as constants.  That is what the REFERENCE's own backward delivers through BatchNorm: an op that has completed its backward drops
 * every later arrival (operation.h:80-88) and BatchNorm's input has three consumers (x - mean, mean, var; nn.cpp:301-316), so only
 * the first, direct path (Mul::_backward operation.h:159-164 -> Div::_backward :192-198) reaches the transform.  Pinned against
 * the reference's through-layer gradients (tests/golden ref_full_*); exists only to compare with them, never the default. */
int gnnx_bn_relu_bwd_quirk_f32(const float *d_X, int64_t ldx, const float *d_Y, int64_t ldy, const float *d_dY, int64_t ldd,
                               int64_t n_rows, int32_t n_feat, const float *d_mean, const float *d_var, float eps, const float *d_gamma,
                               const float *d_beta, int relu, float *d_dX, int64_t ldo, float *d_dgamma, float *d_dbeta,
                               void *d_workspace, size_t workspace_bytes, void *stream);
int gnnx_bn_partial_f32(const float *d_X, int64_t ldx, int64_t n_rows, int32_t n_feat, const float *d_mean, float scale, float *d_out,
                        void *d_workspace, size_t workspace_bytes, void *stream);
int gnnx_bn_relu_bwd_sums_f32(const float *d_X, int64_t ldx, const float *d_Y, int64_t ldy, const float *d_dY, int64_t ldd,
                              int64_t n_rows, int32_t n_feat, const float *d_mean, const float *d_var, float eps, const float *d_gamma,
                              const float *d_beta, int relu, float *d_dgamma, float *d_dbeta, void *d_workspace, size_t workspace_bytes,
                              void *stream);
int gnnx_bn_relu_bwd_apply_f32(const float *d_X, int64_t ldx, const float *d_Y, int64_t ldy, const float *d_dY, int64_t ldd,
                               int64_t n_rows, int32_t n_feat, const float *d_mean, const float *d_var, float eps, const float *d_gamma,
                               const float *d_beta, int relu, const float *d_dgamma, const float *d_dbeta, int64_t n_total, float *d_dX,
                               int64_t ldo, void *d_workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ next row: loss + optimiser ---- */
/*
 * Softmax cross-entropy on the last layer's logits and the SGD update, so a multi-layer GCN runs as a whole training
 * step on the device.
 *   loss    = mean_i -log( exp(x_i[t_i]) / (sum_c exp(x_ic) + 1e-20) )     the reference's forward, nn.cpp:442-453
 *             (no max-subtraction, like the reference); d_loss: one float on the device (may be NULL)
 *   dlogits = (softmax(x_i) - onehot(t_i)) / N                             textbook (the reference's backward throws);
 *             may be NULL.  Synchronises `stream` (validates the targets: GNNX_ERR_INDEX_RANGE).
 *   sgd     : p -= lr * (g + weight_decay * p)                             textbook (nn.cpp:395-421 indexes an empty vector)
 */
int gnnx_softmax_ce_workspace(int64_t n_rows, size_t *bytes);
int gnnx_softmax_ce_f32(const float *d_logits, int64_t ldx, const int32_t *d_target, int64_t n_rows, int32_t n_classes, float *d_loss,
                        float *d_dlogits, int64_t ldd, void *d_workspace, size_t workspace_bytes, void *stream);
/* The same with the column sums of dlogits -- the last layer's bias gradient (operation.h:114-128 sum_to_size on the Add node's
 * bias operand, reached with dlogits as the incoming gradient) -- accumulated by the kernel that writes dlogits instead of one more
 * pass over them.  d_colsum [n_classes] may be NULL (then exactly gnnx_softmax_ce_f32).  Deterministic (fixed summation order);
 * the order differs from gnnx_colsum_f32's, so the two agree to rounding. */
int gnnx_softmax_ce_colsum_workspace(int64_t n_rows, int32_t n_classes, size_t *bytes);
int gnnx_softmax_ce_colsum_f32(const float *d_logits, int64_t ldx, const int32_t *d_target, int64_t n_rows, int32_t n_classes,
                               float *d_loss, float *d_dlogits, int64_t ldd, float *d_colsum, void *d_workspace,
                               size_t workspace_bytes, void *stream);
/* A shard's share (1-D vertex partition): n_rows local rows of a batch of n_total; d_loss is this rank's term of the mean (the caller
 * sums the ranks' terms), dlogits and the column sums carry 1 / n_total.  n_total == n_rows is gnnx_softmax_ce_colsum_f32. */
int gnnx_softmax_ce_partial_f32(const float *d_logits, int64_t ldx, const int32_t *d_target, int64_t n_rows, int32_t n_classes,
                                int64_t n_total, float *d_loss, float *d_dlogits, int64_t ldd, float *d_colsum, void *d_workspace,
                                size_t workspace_bytes, void *stream);
int gnnx_sgd_step_f32(float *d_param, const float *d_grad, int64_t n, float lr, float weight_decay, void *stream);

/* ------------------------------------------------------------------ halo (multi-GPU) ------------- */
/* Pack rows for the all-to-all-v send buffer: out[k,:] = X[idx[k],:]; and the reverse for backward:
 * Y[idx[k],:] += in[k,:] (idx may repeat across calls but NOT within one call => no atomics, deterministic). */
int gnnx_gather_rows_f32(const float *d_X, int64_t ldx, const int32_t *d_idx, int64_t n_idx, int32_t n_feat,
                         float *d_out, int64_t ldo, void *stream);
int gnnx_scatter_add_rows_f32(const float *d_in, int64_t ldi, const int32_t *d_idx, int64_t n_idx, int32_t n_feat,
                              float *d_Y, int64_t ldy, void *stream);

/* The exchange itself.  Two transports behind one handle:
 *   RCCL  (gnnx_comm_init): one process per GPU; librccl is bound with dlopen at first use.  Every rank creates the
 *         communicator from the same 128-byte id (rank 0 calls gnnx_comm_unique_id and ships it to the others by whatever
 *         channel the host program has).
 *   local (gnnx_comm_init_local): the `world` ranks are threads of ONE process; handles for all ranks are created by one call
 *         and handed to the threads.  Collectives rendezvous on host memory and move data with device copies on each rank's
 *         stream (they synchronise that stream).  Rank threads may drive different GPUs: the all-to-all-v uses device-to-device
 *         copies, the all-reduce reads every peer's buffer from a kernel and enables peer access between the ranks' devices on
 *         first use (GNNX_ERR_UNSUPPORTED where the topology has none: use the RCCL transport there).  Destroying a handle while
 *         peers wait in a collective fails their call instead of hanging it; a collective that completed is never failed by a
 *         peer's later destroy.
 * gnnx_halo_exchange_f32 is the all-to-all-v of the halo step: rows for peer p are send_rows[p] consecutive rows of d_send
 * (peer-major, as gnnx_gather_rows_f32 packs them with the peer-major send list), rows from peer p land as recv_rows[p]
 * consecutive rows of d_recv (the [halo] tail of the feature buffer, halo ids being grouped by owner).  On RCCL: one group of
 * ncclSend/ncclRecv pairs, each pair of GPUs uses its own xGMI link; a failed Send/Recv still closes the group before the
 * error is returned.  send_rows / recv_rows are HOST arrays of `world` entries. */
typedef struct gnnx_comm gnnx_comm;
int gnnx_comm_unique_id(void *id_out_128_bytes);
int gnnx_comm_init(gnnx_comm **comm, int world, int rank, const void *id_128_bytes);
int gnnx_comm_init_local(gnnx_comm **comms_out /* [world] */, int world);
int gnnx_comm_info(const gnnx_comm *comm, int *world, int *rank);
int gnnx_comm_destroy(gnnx_comm *comm);
int gnnx_halo_exchange_f32(gnnx_comm *comm, const float *d_send, const int64_t *send_rows, float *d_recv, const int64_t *recv_rows,
                           int32_t n_feat, void *stream);
int gnnx_allreduce_sum_f32(gnnx_comm *comm, float *d_buf, int64_t n, void *stream);

/* ------------------------------------------------------------------ partition + halo plan -------- */
/*
 * 1-D vertex partition and the halo plan of a shard, built on the device (SURVEY.md 8(b) `gnnx_halo_plan`, 8(e)).  The
 * reference is a single process on a dense N x N matrix and has no counterpart; what these calls must preserve is its
 * SUMMATION ORDER: a row's entries stay sorted by ORIGINAL column id (the CSR build sees original ids) and are renumbered
 * value by value, so a shard's SpMM adds the same terms in the same order as the unsharded one (bit-identical rows).
 *
 *   gnnx_vertex_weights     w[v] = out-degree + in-degree + row_weight (the cost model: one unit per incident edge for the two
 *                           aggregations, row_weight ~ 0.08 F for the three GEMM passes over the row).  Synchronises.
 *   gnnx_partition_deal     vertices in stable descending-weight order (ties: ascending id) are dealt to the ranks in snake
 *                           order (0..P-1, P-1..0, ...): every rank gets n/P +- 1 rows with the same degree mix => equal GEMM
 *                           rows, non-zeros and per-link halo volume.  d_owner[v] = rank; d_nid[v] = new id, rank p owning the
 *                           contiguous new-id range [cuts[p], cuts[p+1]) in ascending original id.  cuts: HOST, world+1
 *                           entries.  world == 1 is the identity.  Synchronises.
 *   gnnx_partition_scramble a second relabelling INSIDE every rank's range: position k of rank p's n_p vertices moves to
 *                           (k * 2654435761) mod n_p (a bijection: the multiplier is a prime above every n_p).  Synthetic power-law
 *                           generators (R-MAT) put the hubs on the ids with few one-bits, i.e. feature rows whose addresses have few
 *                           one-bits: the address bits that select the memory channel / cache slice are mostly zero and the hottest rows pile onto a few
 *                           (10 M / 100 M, F = 256: aggregation 19.0 -> 13.7 ms once the labels are scrambled).  world == 1 with
 *                           d_nid = 0..n-1 gives the single-GPU relabelling.  Row contents and summation order are untouched
 *                           (see above): every vertex's result has the same bits, stored at row nid[v].  Synchronises.
 *   gnnx_shard_select_edges the edges rank `rank` owns (owner[src] == rank; transpose != 0: owner[dst] == rank, roles swapped),
 *                           self loops dropped on ORIGINAL ids: d_rows = local row id (nid - lo), d_cols = ORIGINAL column id;
 *                           capacity n_edges each.  Feed them to gnnx_csr_from_coo (n_nodes = max(n_local, n_nodes),
 *                           GNNX_CSR_KEEP_SELF_LOOPS: a local row id may equal an unrelated original column id).  Synchronises.
 *   gnnx_halo_plan_create   from the shard's CSR columns (ORIGINAL ids): halo = sorted new ids of the remote columns (grouped by
 *                           owner), d_colidx_local[p] = column in [local rows | halo rows] numbering (may alias
 *                           d_colidx_orig), recv_rows[q] = halo rows owned by q.  Synchronises.
 *   gnnx_halo_plan_exchange_requests  tells every owner which rows this rank reads (all-to-all of counts, then of the halo
 *                           ids) and stores the peer-major send list; gnnx_halo_plan_set_send_list does the same from ids the
 *                           host program moved itself (d_want_new_ids: peer-major new ids requested by the peers).
 *   gnnx_halo_plan_info     any out pointer may be NULL; recv_rows / send_rows: HOST arrays of `world` entries; d_halo_ids /
 *                           d_send_idx: device pointers owned by the plan.
 *   gnnx_halo_exchange_rows_f32  the per-aggregation step: pack rows [send list] of d_buf[:n_local] into d_send_buf
 *                           ([n_send, n_feat]) and exchange into d_buf[n_local:] (d_buf: [n_local + n_halo, n_feat]).
 */
typedef struct gnnx_halo_plan gnnx_halo_plan;
int gnnx_vertex_weights(const int32_t *d_src, const int32_t *d_dst, int64_t n_edges, int32_t n_nodes, int32_t row_weight,
                        int32_t *d_weight, void *stream);
int gnnx_partition_deal(const int32_t *d_weight, int32_t n_nodes, int world, int32_t *d_owner, int32_t *d_nid, int64_t *cuts_out,
                        void *stream);
int gnnx_partition_scramble(const int32_t *d_owner, int32_t n_nodes, int world, const int64_t *cuts, int32_t *d_nid, void *stream);
int gnnx_shard_select_edges(const int32_t *d_src, const int32_t *d_dst, int64_t n_edges, const int32_t *d_owner, const int32_t *d_nid,
                            int rank, int64_t lo, int transpose, int32_t *d_rows, int32_t *d_cols, int64_t *n_selected, void *stream);
int gnnx_halo_plan_create(const int32_t *d_colidx_orig, int64_t nnz, const int32_t *d_nid, int32_t n_nodes, int world, int rank,
                          const int64_t *cuts, int32_t *d_colidx_local, gnnx_halo_plan **plan, void *stream);
int gnnx_halo_plan_destroy(gnnx_halo_plan *plan);
int gnnx_halo_plan_info(const gnnx_halo_plan *plan, int64_t *n_local, int64_t *n_halo, int64_t *n_send, int64_t *recv_rows,
                        int64_t *send_rows, const int32_t **d_halo_ids, const int32_t **d_send_idx);
int gnnx_halo_plan_set_send_list(gnnx_halo_plan *plan, const int32_t *d_want_new_ids, const int64_t *send_rows, void *stream);
int gnnx_halo_plan_exchange_requests(gnnx_halo_plan *plan, gnnx_comm *comm, void *stream);
/* (the pack inside gnnx_halo_exchange_rows_f32 runs from the producer's side -- gnnx_rows_to_slots_f32 on the plan's own slot table,
 * built by gnnx_halo_plan_set_send_list -- whenever the rows are 16-byte pieces; the gather by the send list otherwise: same buffer) */
int gnnx_halo_exchange_rows_f32(const gnnx_halo_plan *plan, gnnx_comm *comm, float *d_buf, int64_t ldb, int32_t n_feat,
                                float *d_send_buf, void *stream);
/* A producer that packs while it produces (gnnx_gemm_nt_rows_to_slots_f32: the transform's epilogue; gnnx_rows_to_slots_f32 with the
 * column sums: the upstream gradient's dbias pass) takes the plan's slot table ([n_local][8]; NULL when the plan has none: more than
 * 8 ranks, or no rows to send) and hands the filled send buffer ([n_send][n_feat], dense) to gnnx_halo_exchange_packed_f32 -- the
 * exchange step of gnnx_halo_exchange_rows_f32 without its pack. */
int gnnx_halo_plan_slot_table(const gnnx_halo_plan *plan, const int32_t **d_slots);
int gnnx_halo_exchange_packed_f32(const gnnx_halo_plan *plan, gnnx_comm *comm, float *d_buf, int64_t ldb, int32_t n_feat,
                                  const float *d_send_buf, void *stream);

/* ------------------------------------------------------------------ synthetic inputs ------------- */
/* Counter-based SplitMix64 generators, bit-identical to gnn.cpp_amd/synth.py. */
int gnnx_rmat_edges(uint64_t seed, int32_t n_nodes, int64_t n_edges, int64_t first_edge, double a, double b,
                    double c, int32_t *d_src, int32_t *d_dst, void *stream);
int gnnx_uniform_pm1_f32(uint64_t seed, int64_t n, float scale, float *d_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GNNX_H */
