/*
 * TEST INFRASTRUCTURE ONLY.  CPU restatement ("oracle") of the walexi/gnn.cpp GCN hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object, and only as the checker / the reported CPU baseline.  The product path
 * (gnn.cpp_amd/, include/gnnx.h, libgnnx_hip.so) never links, imports or falls back to it.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit (modulo the sign of zero)
 * against outputs of the real reference compiled by oracle/build_ref.sh (oracle/_ref/ref_driver);
 * those outputs are committed as tests/golden/<case>.npz by tests/golden/make_golden.py.
 * The reference's own tests pin nothing numerically on this path.
 *
 * What is restated (reference file:line):
 *   adjacency semantics   graph.cpp:21-44 (A[r][c] = 1, duplicates collapse), graph.cpp:68-75 with
 *                         fillValue 0 (diagonal zeroed => self loops REMOVED), graph.cpp:46-67
 *                         (row-major scan => edges sorted by (src,dst))
 *   degree / norm         graph.cpp:177-185: deg = rowsum(A) + 1; s = pow(deg,-0.5); norm = (A.s) * s
 *   transform             nn.cpp:205-211: H = x.mm(W.t(-1,-2))     (GCNConv's lin has no bias, graph.cpp:162)
 *   aggregate             graph.cpp:204-212: agg = A.mm(H); agg = agg * norm ; graph.cpp:188: + bias
 *   backward              operation.h:114-128 (Add), :144-167 (Mul), :504-534 (MatMul), :416-433
 *                         (Transpose), tensor.h:618-638 (sum_to_size), tensor.h:260-276 (_grad +=)
 *
 * Arithmetic order (this is what makes bit-exact parity possible):
 *   functional::matmul (functional.h:433-439) computes every output element as
 *   `(r_slice * l_slice).sum()`.  That is libstdc++'s expression-template _Expr::sum()
 *   (bits/valarray_after.h:293-305) which starts from the LAST element and walks DOWN:
 *       s = p[n-1]; s += p[n-2]; ... ; s += p[0];      p[k] = fl32(l[k] * r[k])
 *   i.e. sequential fp32, DESCENDING k, product rounded before the add (no FMA).
 *   For the adjacency products p[k] is either H[k] exactly (A=1) or +-0 (A=0, an exact no-op),
 *   so a CSR walk over the row's neighbours in DESCENDING column order reproduces the dense
 *   result exactly (up to the sign of a zero result).
 *   functional::sum (functional.h:267-296) materialises a valarray first, and valarray::sum()
 *   (bits/valarray_array.h:348-354) walks UP (ascending): used for deg and for dbias.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).  -ffp-contract=off keeps
 * mul and add separately rounded exactly like the reference's x86-64 baseline build.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define API __attribute__((visibility("default")))

static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y);
}

/*
 * COO [2,E] -> CSR with the reference's adjacency semantics (graph.cpp:21-75):
 * duplicates collapse (assignment at graph.cpp:40), self loops dropped (fill_diagonal_(0) at
 * graph.cpp:72), entries ordered row-major (scan at graph.cpp:52-60).
 * rowptr has N+1 entries, colidx capacity E.  Returns nnz, or -1 if an index is out of [0,N)
 * (the reference would write out of bounds, graph.cpp:40; Data's ctor rejects it, graph.cpp:89).
 */
API int64_t gcn_oracle_coo_to_csr(const int32_t *src, const int32_t *dst, int64_t E, int32_t N,
                                  int64_t *rowptr, int32_t *colidx)
{
    uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(E > 0 ? E : 1));
    int64_t m = 0;
    for (int64_t e = 0; e < E; e++) {
        int32_t r = src[e], c = dst[e];
        if (r < 0 || c < 0 || r >= N || c >= N) { free(keys); return -1; }
        if (r == c) continue;
        keys[m++] = ((uint64_t)(uint32_t)r << 32) | (uint32_t)c;
    }
    qsort(keys, (size_t)m, sizeof(uint64_t), cmp_u64);
    memset(rowptr, 0, sizeof(int64_t) * ((size_t)N + 1));
    int64_t nnz = 0;
    for (int64_t i = 0; i < m; i++) {
        if (i && keys[i] == keys[i - 1]) continue;
        colidx[nnz++] = (int32_t)(keys[i] & 0xffffffffu);
        rowptr[(keys[i] >> 32) + 1]++;
    }
    for (int32_t i = 0; i < N; i++) rowptr[i + 1] += rowptr[i];
    free(keys);
    return nnz;
}

/* Weighted adjacency (edge_attr): graph.cpp:21-44 edge_to_adj_mat writes A[r][c] = w edge by edge, so the LAST duplicate
 * wins; add_self_loops (graph.cpp:68-75) then overwrites the diagonal with fillValue and adj_to_edge_list (graph.cpp:46-67)
 * scans row-major keeping entries with int(value) != 0.
 *   diag_mode 0: diagonal as given; 1: diagonal removed; 2: every (i,i) = diag_value.
 *   drop_truncated_zero: apply adj_to_edge_list's int() filter (otherwise explicit zeros are kept as entries).
 * Capacity of colidx / vals: E + N.  Returns nnz or -1 on an out-of-range index. */
typedef struct { uint64_t key; int64_t pos; } key_pos_t;
static int cmp_key_pos(const void *a, const void *b)
{
    const key_pos_t *x = (const key_pos_t *)a, *y = (const key_pos_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos ? 1 : 0);
}
API int64_t gcn_oracle_coo_to_csr_weighted(const int32_t *src, const int32_t *dst, const float *w, int64_t E, int32_t N,
                                           int diag_mode, float diag_value, int drop_truncated_zero, int64_t *rowptr,
                                           int32_t *colidx, float *vals)
{
    int64_t cap = E + N + 1;
    key_pos_t *kp = (key_pos_t *)malloc(sizeof(key_pos_t) * (size_t)cap);
    int64_t m = 0;
    for (int64_t e = 0; e < E; e++) {
        int32_t r = src[e], c = dst[e];
        if (r < 0 || c < 0 || r >= N || c >= N) { free(kp); return -1; }
        if (r == c && diag_mode != 0) continue;
        kp[m].key = ((uint64_t)(uint32_t)r << 32) | (uint32_t)c;
        kp[m].pos = e;
        m++;
    }
    if (diag_mode == 2)
        for (int32_t i = 0; i < N; i++) {
            kp[m].key = ((uint64_t)(uint32_t)i << 32) | (uint32_t)i;
            kp[m].pos = E + i;
            m++;
        }
    qsort(kp, (size_t)m, sizeof(key_pos_t), cmp_key_pos);
    memset(rowptr, 0, sizeof(int64_t) * ((size_t)N + 1));
    int64_t nnz = 0;
    for (int64_t i = 0; i < m; i++) {
        if (i + 1 < m && kp[i + 1].key == kp[i].key) continue; /* a later assignment overwrites this one */
        float v = kp[i].pos < E ? w[kp[i].pos] : diag_value;
        if (drop_truncated_zero && (int)v == 0) continue;
        colidx[nnz] = (int32_t)(kp[i].key & 0xffffffffu);
        vals[nnz] = v;
        nnz++;
        rowptr[(kp[i].key >> 32) + 1]++;
    }
    for (int32_t i = 0; i < N; i++) rowptr[i + 1] += rowptr[i];
    free(kp);
    return nnz;
}

/* adj->mm(x) on a weighted adjacency (functional.h:433-439): out[i][f] = sum over DESCENDING column c of fl(A[i][c] * x[c][f]),
 * products and sums rounded separately; absent entries contribute +-0. */
API void gcn_oracle_spmm_vals(const int64_t *rowptr, const int32_t *colidx, const float *vals, int32_t N, int32_t F, const float *X,
                              float *out)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int32_t i = 0; i < N; i++) {
        float *o = out + (int64_t)i * F;
        for (int32_t f = 0; f < F; f++) o[f] = 0.0f;
        for (int64_t p = rowptr[i + 1] - 1; p >= rowptr[i]; p--) {
            const float *x = X + (int64_t)colidx[p] * F;
            const float v = vals[p];
            for (int32_t f = 0; f < F; f++) {
                float t = v * x[f];
                o[f] = o[f] + t;
            }
        }
    }
}

/* adj->sum(-1) on a weighted adjacency: functional::sum walks UP (valarray::sum), zeros included */
API void gcn_oracle_rowsum_vals(const int64_t *rowptr, const float *vals, int32_t N, float *out)
{
    for (int32_t i = 0; i < N; i++) {
        float acc = 0.0f;
        for (int64_t p = rowptr[i]; p < rowptr[i + 1]; p++) acc = acc + vals[p];
        out[i] = acc;
    }
}

/* CSR of A^T (columns ascending inside each row): what MatMul::_backward's dense transpose of A
 * (operation.h:524-527) turns into when A is never materialised. */
API void gcn_oracle_csr_transpose(const int64_t *rowptr, const int32_t *colidx, int32_t N,
                                  int64_t *rowptrT, int32_t *colidxT)
{
    int64_t nnz = rowptr[N];
    memset(rowptrT, 0, sizeof(int64_t) * ((size_t)N + 1));
    for (int64_t e = 0; e < nnz; e++) rowptrT[colidx[e] + 1]++;
    for (int32_t i = 0; i < N; i++) rowptrT[i + 1] += rowptrT[i];
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * ((size_t)N + 1));
    memcpy(cur, rowptrT, sizeof(int64_t) * ((size_t)N + 1));
    for (int32_t i = 0; i < N; i++)
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; e++) colidxT[cur[colidx[e]]++] = i;
    free(cur);
}

/* graph.cpp:177-185.  s_i = powf(1 + outdeg_i, -0.5f);  norm_i = fl(fl(sum_j A_ij s_j) * s_i),
 * inner sum sequential over DESCENDING j (matmul order, see header). */
API void gcn_oracle_degree_norm(const int64_t *rowptr, const int32_t *colidx, int32_t N, float *s, float *norm)
{
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < N; i++) {
        float deg = (float)(rowptr[i + 1] - rowptr[i]) + 1.0f; /* ascending sum of 1.0f's is exact below 2^24 */
        s[i] = powf(deg, -0.5f);
    }
#pragma omp parallel for schedule(dynamic, 1024)
    for (int32_t i = 0; i < N; i++) {
        float acc = 0.0f;
        for (int64_t e = rowptr[i + 1] - 1; e >= rowptr[i]; e--) acc += s[colidx[e]];
        norm[i] = acc * s[i];
    }
}

/* Hot inner loops, compiled for several x86 vector widths and picked at load time (GCC function multi-versioning): the
 * arithmetic per element is the same one rounded multiply and one rounded add at any width (-ffp-contract=off: no FMA),
 * so the result does not depend on which clone runs; the host baseline just should not be an SSE2-only number. */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define CLONES __attribute__((target_clones("avx512f", "avx2", "default"), noinline))
#else
#define CLONES __attribute__((noinline))
#endif

/* h[0..n) = sum over k DESCENDING of fl(x[k] * Wt[k][0..n))  (one output row of a GEMM, reference order per element) */
CLONES static void row_times_matrix(const float *x, const float *Wt, int32_t K, int32_t n, float *h)
{
    enum { S = 64 };  /* outputs per strip: the strip's accumulators live in vector registers across the whole k loop */
    int32_t o0 = 0;
    for (; o0 + S <= n; o0 += S) {
        float acc[S];
        const float *w = Wt + (int64_t)(K - 1) * n + o0;
        float xv = x[K - 1];
        for (int32_t o = 0; o < S; o++) acc[o] = xv * w[o];
        for (int32_t k = K - 2; k >= 0; k--) {
            w = Wt + (int64_t)k * n + o0;
            xv = x[k];
            for (int32_t o = 0; o < S; o++) acc[o] += xv * w[o];
        }
        for (int32_t o = 0; o < S; o++) h[o0 + o] = acc[o];
    }
    if (o0 < n) {  /* ragged tail, same order */
        const int32_t m = n - o0;
        const float *w = Wt + (int64_t)(K - 1) * n + o0;
        float xv = x[K - 1];
        for (int32_t o = 0; o < m; o++) h[o0 + o] = xv * w[o];
        for (int32_t k = K - 2; k >= 0; k--) {
            w = Wt + (int64_t)k * n + o0;
            xv = x[k];
            for (int32_t o = 0; o < m; o++) h[o0 + o] += xv * w[o];
        }
    }
}

/* dW[o0..o1)[k0..k1) = sum over i DESCENDING of fl(X[i][k] * dH[i][o]); accumulators of the block stay in L1 */
enum { DW_OB = 8, DW_KB = 64 };
CLONES static void dw_block(const float *X, const float *dH, int64_t N, int32_t Fin, int32_t Fout, int32_t o0, int32_t o1, int32_t k0,
                            int32_t k1, float *dW)
{
    float acc[DW_OB][DW_KB];
    const int32_t kn = k1 - k0;
    {
        const float *x = X + (N - 1) * Fin + k0;
        const float *g = dH + (N - 1) * Fout;
        for (int32_t o = o0; o < o1; o++) {
            const float gv = g[o];
            float *a = acc[o - o0];
            for (int32_t k = 0; k < kn; k++) a[k] = x[k] * gv;
        }
    }
    for (int64_t i = N - 2; i >= 0; i--) {
        const float *x = X + i * Fin + k0;
        const float *g = dH + i * Fout;
        for (int32_t o = o0; o < o1; o++) {
            const float gv = g[o];
            float *a = acc[o - o0];
            for (int32_t k = 0; k < kn; k++) a[k] += x[k] * gv;
        }
    }
    for (int32_t o = o0; o < o1; o++)
        for (int32_t k = 0; k < kn; k++) dW[(int64_t)o * Fin + k0 + k] = acc[o - o0][k];
}

/* nn.cpp:205-211 via functional.h:433-439:  H[i,o] = sum_{k desc} fl(X[i,k] * W[o,k]).
 * Loop nest: k outermost-descending per row with W pre-transposed, so the inner loop runs over o contiguously
 * (vectorisable) while every output element still sees its products in the reference's order. */
API void gcn_oracle_linear_fwd(const float *X, const float *W, int64_t N, int32_t Fin, int32_t Fout, float *H)
{
    float *Wt = (float *)malloc(sizeof(float) * (size_t)(Fin > 0 ? Fin : 1) * (size_t)(Fout > 0 ? Fout : 1));
    for (int32_t o = 0; o < Fout; o++)
        for (int32_t k = 0; k < Fin; k++) Wt[(int64_t)k * Fout + o] = W[(int64_t)o * Fin + k];
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; i++) {
        const float *x = X + i * Fin;
        float *h = H + i * Fout;
        if (Fin == 0) {
            for (int32_t o = 0; o < Fout; o++) h[o] = 0.0f;
            continue;
        }
        row_times_matrix(x, Wt, Fin, Fout, h);
    }
    free(Wt);
}

/* graph.cpp:208-209 (+ graph.cpp:188 when bias != NULL):
 *   out[i,:] = fl(fl(sum_{j in N(i), DESCENDING j} H[j,:]) * norm[i]) (+ bias).  norm may be NULL (plain A.H). */
API void gcn_oracle_aggregate_fwd(const int64_t *rowptr, const int32_t *colidx, int32_t N, int32_t F,
                                  const float *H, const float *norm, const float *bias, float *out)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int32_t i = 0; i < N; i++) {
        float *o = out + (int64_t)i * F;
        int64_t b = rowptr[i], e = rowptr[i + 1];
        if (e == b) {
            for (int32_t f = 0; f < F; f++) o[f] = 0.0f;
        } else {
            const float *h = H + (int64_t)colidx[e - 1] * F;
            for (int32_t f = 0; f < F; f++) o[f] = h[f];
            for (int64_t p = e - 2; p >= b; p--) {
                h = H + (int64_t)colidx[p] * F;
                for (int32_t f = 0; f < F; f++) o[f] += h[f];
            }
        }
        if (norm) {
            float nv = norm[i];
            for (int32_t f = 0; f < F; f++) o[f] = o[f] * nv;
        }
        if (bias)
            for (int32_t f = 0; f < F; f++) o[f] = o[f] + bias[f];
    }
}

/* Backward of the aggregate (operation.h:144-167 then :524-531):
 *   G' = fl(G (.) norm);  dH[j,:] = sum_{i : A_ij = 1, DESCENDING i} G'[i,:]  (rows of A^T). */
API void gcn_oracle_aggregate_bwd(const int64_t *rowptrT, const int32_t *colidxT, int32_t N, int32_t F,
                                  const float *G, const float *norm, float *dH)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int32_t j = 0; j < N; j++) {
        float *o = dH + (int64_t)j * F;
        int64_t b = rowptrT[j], e = rowptrT[j + 1];
        for (int32_t f = 0; f < F; f++) o[f] = 0.0f;
        for (int64_t p = e - 1; p >= b; p--) {
            int32_t i = colidxT[p];
            const float *g = G + (int64_t)i * F;
            float nv = norm ? norm[i] : 1.0f;
            if (p == e - 1)
                for (int32_t f = 0; f < F; f++) o[f] = g[f] * nv;
            else
                for (int32_t f = 0; f < F; f++) o[f] += g[f] * nv;
        }
    }
}

/* dbias (operation.h:114-128 -> tensor.h:618-638 -> functional.h:285-288): column sums, ASCENDING row order. */
API void gcn_oracle_colsum(const float *G, int64_t N, int32_t F, float *out)
{
    for (int32_t f = 0; f < F; f++) out[f] = 0.0f;
    if (N == 0) return;
#pragma omp parallel for schedule(static)
    for (int32_t f0 = 0; f0 < F; f0 += 16) {
        int32_t f1 = f0 + 16 < F ? f0 + 16 : F;
        for (int32_t f = f0; f < f1; f++) out[f] = G[f];
        for (int64_t i = 1; i < N; i++)
            for (int32_t f = f0; f < f1; f++) out[f] += G[i * F + f];
    }
}

/* Backward of the transform (operation.h:516-531, :416-433):
 *   dX[i,k] = sum_{o desc} fl(dH[i,o] * W[o,k])          (G . (W^T)^T)
 *   dW[o,k] = sum_{i desc} fl(X[i,k] * dH[i,o])          (x^T . G, then transposed back)
 * Loop nests keep the reduction index outermost (descending) and a contiguous inner loop; the per-element
 * order of additions is the reference's. */
API void gcn_oracle_linear_bwd(const float *dH, const float *X, const float *W, int64_t N, int32_t Fin,
                               int32_t Fout, float *dX, float *dW)
{
    if (dX) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < N; i++) {
            float *d = dX + i * Fin;
            const float *g = dH + i * Fout;
            if (Fout == 0) {
                for (int32_t k = 0; k < Fin; k++) d[k] = 0.0f;
                continue;
            }
            row_times_matrix(g, W, Fout, Fin, d);  /* dX[i][k] = sum_{o desc} fl(dH[i][o] * W[o][k]) */
        }
    }
    if (dW) {
        /* dW[o][k] = sum_{i DESCENDING} fl(X[i][k] * dH[i][o]).  Every output element keeps that order; the loop nest is
         * blocked (OB outputs x KB inputs per task, i innermost over the block) so that X is streamed Fout/OB times instead
         * of Fout times and the accumulators stay in L1 -- same bits, a host baseline that is not bound by re-reading X. */
        const int32_t n_ob = (Fout + DW_OB - 1) / DW_OB, n_kb = (Fin + DW_KB - 1) / DW_KB;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
        for (int32_t ob = 0; ob < n_ob; ob++)
            for (int32_t kb = 0; kb < n_kb; kb++) {
                const int32_t o0 = ob * DW_OB, o1 = o0 + DW_OB < Fout ? o0 + DW_OB : Fout;
                const int32_t k0 = kb * DW_KB, k1 = k0 + DW_KB < Fin ? k0 + DW_KB : Fin;
                if (N == 0) {
                    for (int32_t o = o0; o < o1; o++)
                        for (int32_t k = k0; k < k1; k++) dW[(int64_t)o * Fin + k] = 0.0f;
                    continue;
                }
                dw_block(X, dH, N, Fin, Fout, o0, o1, k0, k1, dW);
            }
    }
}

/*
 * Literal dense restatement of graph.cpp:204-212 for tiny N (O(N^2 F)); used by the tests to show
 * that the CSR walk above equals the dense matmul the reference actually runs.
 * A is built as graph.cpp:21-44 does (assignment), the diagonal is NOT touched here (callers pass
 * the already-stripped edge list, as GCNConv::forward does after add_self_loops).
 */
API void gcn_oracle_dense_aggregate(const int32_t *src, const int32_t *dst, int64_t E, int32_t N, int32_t F,
                                    const float *H, const float *norm, float *out)
{
    float *A = (float *)calloc((size_t)N * N, sizeof(float));
    for (int64_t e = 0; e < E; e++) A[(int64_t)src[e] * N + dst[e]] = 1.0f;
    for (int32_t i = 0; i < N; i++)
        for (int32_t f = 0; f < F; f++) {
            float acc = A[(int64_t)i * N + N - 1] * H[(int64_t)(N - 1) * F + f];
            for (int32_t k = N - 2; k >= 0; k--) acc += A[(int64_t)i * N + k] * H[(int64_t)k * F + f];
            out[(int64_t)i * F + f] = norm ? acc * norm[i] : acc;
        }
    free(A);
}

/*
 * BatchNorm (training mode) + ReLU as GCNConv::forward applies them between transform and aggregation
 * (graph.cpp:174-175).  Restates nn.cpp:301-330 and nn.cpp:229-237 with their arithmetic order:
 *   mean_f = (sum_{i ASC} x_if) / (float)N                         functional.h:299-307 (sum materialises => ascending)
 *   var_f  = (sum_{i DESC} powf(x_if - fl(sum_asc/N), 2)) / max(0,N) functional.h:380-388: std::pow(valarray,2) is an
 *            expression template, its .sum() walks DOWN; correction = 0 (nn.cpp:312)
 *   y = ((x - mean) / powf(var + eps, 0.5f)) * gamma + beta          nn.cpp:313-316, each op rounded
 *   relu: x > 0 ? x : 0                                              functional.h:459-460 (a select)
 * mean_out / var_out may be NULL.  do_bn = 0 applies only the ReLU.
 */
API void gcn_oracle_bn_relu_fwd(const float *X, int64_t N, int32_t F, const float *gamma, const float *beta, float eps, int do_bn,
                                int do_relu, float *Y, float *mean_out, float *var_out)
{
    float *mean = (float *)malloc(sizeof(float) * (size_t)(F > 0 ? F : 1));
    float *sd = (float *)malloc(sizeof(float) * (size_t)(F > 0 ? F : 1));
    if (do_bn) {
        for (int32_t f = 0; f < F; f++) {
            float s = X[f];
            for (int64_t i = 1; i < N; i++) s += X[i * F + f];
            float m = s / (float)(int)N;
            float d = X[(N - 1) * F + f] - m;
            float v = powf(d, 2.0f);
            for (int64_t i = N - 2; i >= 0; i--) {
                d = X[i * F + f] - m;
                v += powf(d, 2.0f);
            }
            v = v / (float)(int)N;
            mean[f] = m;
            sd[f] = powf(v + eps, 0.5f);
            if (mean_out) mean_out[f] = m;
            if (var_out) var_out[f] = v;
        }
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; i++)
        for (int32_t f = 0; f < F; f++) {
            float v = X[i * F + f];
            if (do_bn) {
                v = (v - mean[f]) / sd[f];
                v = v * gamma[f];
                v = v + beta[f];
            }
            if (do_relu) v = v > 0.0f ? v : 0.0f;
            Y[i * F + f] = v;
        }
    free(mean);
    free(sd);
}

/* Backward THROUGH BatchNorm + ReLU as the reference's traversal delivers it ("reference-quirk mode").
 * An op that has completed its backward drops every later arrival (operation.h:80-88); BatchNorm's input feeds three consumers
 * (x - mean, mean, var: nn.cpp:301-316) and the direct one is visited first, so the statistics act as constants:
 *   mask   : g = (bn_out > 0) ? dY : 0                              Mask::_backward         operation.h:557-562
 *   dbeta  : column sums of g, ascending rows                       Add::_backward          operation.h:114-128 -> tensor.h:618-638
 *   dgamma : column sums of fl(g * scaled_x)                        Mul::_backward (rhs)    operation.h:151-157
 *   dX     : fl(fl(g * gamma) / sd)                                 Mul::_backward (lhs) :158-164, Div::_backward :192-198
 * with scaled_x, sd, bn_out exactly the forward's values (gcn_oracle_bn_relu_fwd).  Pinned to the reference's through-layer
 * gradients (tests/golden ref_full_*). */
API void gcn_oracle_bn_relu_bwd_quirk(const float *X, int64_t N, int32_t F, const float *gamma, const float *beta, float eps,
                                      const float *dY, float *dX, float *dgamma, float *dbeta)
{
    float *mean = (float *)malloc(sizeof(float) * (size_t)(F > 0 ? F : 1));
    float *sd = (float *)malloc(sizeof(float) * (size_t)(F > 0 ? F : 1));
    float *var = (float *)malloc(sizeof(float) * (size_t)(F > 0 ? F : 1));
    float *Y = (float *)malloc(sizeof(float) * (size_t)(N * F > 0 ? N * F : 1));
    gcn_oracle_bn_relu_fwd(X, N, F, gamma, beta, eps, 1, 0, Y, mean, var);  /* bn_out, no ReLU */
    for (int32_t f = 0; f < F; f++) {
        sd[f] = powf(var[f] + eps, 0.5f);
        dgamma[f] = 0.0f;
        dbeta[f] = 0.0f;
    }
    for (int64_t i = 0; i < N; i++)
        for (int32_t f = 0; f < F; f++) {
            const float g = Y[i * F + f] > 0.0f ? dY[i * F + f] : 0.0f;
            const float scaled = (X[i * F + f] - mean[f]) / sd[f];
            const float t = g * scaled;
            if (i == 0) {
                dgamma[f] = t;
                dbeta[f] = g;
            } else {
                dgamma[f] += t;
                dbeta[f] += g;
            }
            const float gg = g * gamma[f];
            dX[i * F + f] = gg / sd[f];
        }
    free(mean);
    free(sd);
    free(var);
    free(Y);
}

/* Softmax cross-entropy, forward value only (the reference's backward throws): nn.cpp:442-453.
 *   x_n = logits->at(target); out = exp(x_n) / (exp(logits)->sum(-1) + 1e-20); out = -(log(out)); out->sum() / numel
 * row sums and the final sum walk UP (functional::sum materialises), every op separately rounded. */
API float gcn_oracle_cross_entropy(const float *logits, const int32_t *target, int64_t N, int32_t Cn)
{
    float total = 0.0f;
    for (int64_t i = 0; i < N; i++) {
        const float *x = logits + i * Cn;
        float s = expf(x[0]);
        for (int32_t c = 1; c < Cn; c++) s += expf(x[c]);
        float denom = s + (float)1e-20;
        float o = expf(x[target[i]]) / denom;
        float l = logf(o) * -1.0f;
        total = i == 0 ? l : total + l;
    }
    return total / (float)(int)N;
}

/* out[d] = powf((float)d, -0.5f) for d in [0,n): the libm call behind functional.h:253 (std::pow on valarrays),
 * exposed so the tests can measure where a device-side (float)(1/sqrt((double)d)) differs from it. */
API void gcn_oracle_powf_table(int32_t n, float *out)
{
    for (int32_t d = 0; d < n; d++) out[d] = powf((float)d, -0.5f);
}

API void gcn_oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

API int gcn_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
