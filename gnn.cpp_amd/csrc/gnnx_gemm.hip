// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, exact f32;
// 157 TFLOP/s dense peak) -- the dense feature transform of the GCN hot path.
//
// Replaces reference functional::matmul (functional.h:399-441) for X.W^T (nn.cpp:205-211) and, in
// backward, dH.W and dH^T.X (operation.h:504-534); W^T / X^T / dH^T are never materialised (the reference
// does: nn.cpp:207, operation.h:518-519,526-527).
//
// Tiling: 128 x 128 output tile per 256-thread workgroup (4 wavefronts as 2 x 2, each 64 x 64 = 2 x 2
// MFMA tiles of 32 x 32, 64 accumulator VGPRs), K step 32.  Both operand tiles live in LDS k-major
// ([k][m] and [k][n], row stride 130 words) so that an MFMA fragment read is 32 consecutive words per
// half-wave (conflict-free ds_read_b32); K-contiguous operands (X, W) are transposed on the way in
// (register staging, 16-byte global loads, b32 LDS writes at worst 2-way conflicted), k-major operands
// (W in dH.W, both operands of dH^T.X) go in as they are.  Register-staged double buffering: tile t+1 is in
// flight from HBM while tile t is multiplied.
// transA = 1 reduces over the node dimension (K = millions, M x N = F_out x F_in tiny): split-K over
// workgroups into fp32 slabs + an in-order slab reduction (deterministic; no float atomics).
#include "gnnx_common.h"

using namespace gnnx;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDS_LD = 130;  // words; 4*130 % 32 == 8 keeps the transposing writes at <= 2-way conflicts

struct GemmArgs {
    int64_t M, N, K;
    const float *A;
    int64_t lda;
    const float *B;
    int64_t ldb;
    float *C;
    int64_t ldc;
    float alpha, beta;
    int64_t k_per_split;  // multiple of BK
    float *slab;          // split-K partials [splits][M][N] (nullptr => write C directly)
};

// Global -> registers for one operand tile [128 rows(m or n)] x [32 k].
// KC (K-contiguous): element (r, k) at base[r*ld + k]  (X, W^T-as-W).   !KC (k-major): base[k*ld + r].
// VEC: 16-byte loads are legal (ld % 4 == 0, base 16-B aligned, and for KC also K % 4 == 0).
// Out-of-range elements are zero.  Loads are never predicated (see gnnx_spmm.hip): addresses are clamped
// into the matrix and the value is zeroed by a select.
template <bool KC, bool VEC>
__device__ __forceinline__ void load_tile(float (&reg)[16], const float *base, int64_t ld, int64_t r0, int64_t rmax,
                                          int64_t k0, int64_t kmax, int tid)
{
    if constexpr (KC) {
        // thread -> (row = tid/8 + 32*i, kq = tid%8): 8 lanes cover one 128-B row segment
        const int kq = tid & 7;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int64_t r = r0 + (tid >> 3) + 32 * i;
            int64_t k = k0 + kq * 4;
            bool rok = r < rmax;
            int64_t rc = rok ? r : rmax - 1;
            if constexpr (VEC) {
                bool ok = rok && k < kmax;  // K % 4 == 0 => whole float4 in or out
                int64_t kc = k < kmax ? k : 0;
                float4 v = *reinterpret_cast<const float4 *>(base + rc * ld + kc);
                reg[4 * i + 0] = ok ? v.x : 0.f;
                reg[4 * i + 1] = ok ? v.y : 0.f;
                reg[4 * i + 2] = ok ? v.z : 0.f;
                reg[4 * i + 3] = ok ? v.w : 0.f;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    bool ok = rok && (k + j) < kmax;
                    int64_t kc = (k + j) < kmax ? (k + j) : 0;
                    float v = base[rc * ld + kc];
                    reg[4 * i + j] = ok ? v : 0.f;
                }
            }
        }
    } else {
        // thread -> (k = tid/32 + 8*i, rq = tid%32): 32 lanes cover 512 B of one k-row
        const int rq = tid & 31;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int64_t k = k0 + (tid >> 5) + 8 * i;
            int64_t r = r0 + rq * 4;
            bool kok = k < kmax;
            int64_t kc = kok ? k : kmax - 1;
            if constexpr (VEC) {
                bool ok = kok && r < rmax;  // rmax % 4 == 0 on this path
                int64_t rc = r < rmax ? r : 0;
                float4 v = *reinterpret_cast<const float4 *>(base + kc * ld + rc);
                reg[4 * i + 0] = ok ? v.x : 0.f;
                reg[4 * i + 1] = ok ? v.y : 0.f;
                reg[4 * i + 2] = ok ? v.z : 0.f;
                reg[4 * i + 3] = ok ? v.w : 0.f;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    bool ok = kok && (r + j) < rmax;
                    int64_t rc = (r + j) < rmax ? (r + j) : 0;
                    float v = base[kc * ld + rc];
                    reg[4 * i + j] = ok ? v : 0.f;
                }
            }
        }
    }
}

// registers -> LDS tile [32 k][LDS_LD]
template <bool KC>
__device__ __forceinline__ void store_tile(float *lds, const float (&reg)[16], int tid)
{
    if constexpr (KC) {
        const int kq = tid & 7;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int r = (tid >> 3) + 32 * i;
#pragma unroll
            for (int j = 0; j < 4; j++) lds[(kq * 4 + j) * LDS_LD + r] = reg[4 * i + j];
        }
    } else {
        const int rq = tid & 31;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int k = (tid >> 5) + 8 * i;
#pragma unroll
            for (int j = 0; j < 4; j++) lds[k * LDS_LD + rq * 4 + j] = reg[4 * i + j];
        }
    }
}

template <bool A_KC, bool B_KC, bool VEC_A, bool VEC_B>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs g)
{
    __shared__ float lds[2][2][BK * LDS_LD];  // [buffer][A|B]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;  // 2 x 2 wavefronts
    const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
    const int64_t kbeg = (int64_t)blockIdx.z * g.k_per_split;
    const int64_t kend = kbeg + g.k_per_split < g.K ? kbeg + g.k_per_split : g.K;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    float ra[16], rb[16];
    int buf = 0;
    if (kbeg < kend) {
        load_tile<A_KC, VEC_A>(ra, g.A, g.lda, m0, g.M, kbeg, kend, tid);
        load_tile<B_KC, VEC_B>(rb, g.B, g.ldb, n0, g.N, kbeg, kend, tid);
        store_tile<A_KC>(lds[0][0], ra, tid);
        store_tile<B_KC>(lds[0][1], rb, tid);
    }
    __syncthreads();
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = k0 + BK < kend;
        if (more) {  // next tile: HBM -> registers, in flight during the MFMAs below
            load_tile<A_KC, VEC_A>(ra, g.A, g.lda, m0, g.M, k0 + BK, kend, tid);
            load_tile<B_KC, VEC_B>(rb, g.B, g.ldb, n0, g.N, k0 + BK, kend, tid);
        }
        const float *As = lds[buf][0] + wm * 64 + (lane & 31);
        const float *Bs = lds[buf][1] + wn * 64 + (lane & 31);
        const int kh = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a0 = As[(kk + kh) * LDS_LD], a1 = As[(kk + kh) * LDS_LD + 32];
            float b0 = Bs[(kk + kh) * LDS_LD], b1 = Bs[(kk + kh) * LDS_LD + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) {
            store_tile<A_KC>(lds[buf ^ 1][0], ra, tid);
            store_tile<B_KC>(lds[buf ^ 1][1], rb, tid);
        }
        __syncthreads();
        buf ^= 1;
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float *out = g.slab ? g.slab + (int64_t)blockIdx.z * g.M * g.N : g.C;
    const int64_t ldo = g.slab ? g.N : g.ldc;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            int64_t col = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < g.M && col < g.N) {
                    float v = acc[i][j][r];
                    if (g.slab) {
                        out[row * ldo + col] = v;
                    } else {
                        v *= g.alpha;
                        if (g.beta != 0.f) v += g.beta * out[row * ldo + col];
                        out[row * ldo + col] = v;
                    }
                }
            }
        }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *slab, int splits, int64_t M, int64_t N, float alpha,
                                                            float beta, float *C, int64_t ldc)
{
    int64_t total = M * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        float acc = 0.f;
        for (int s = 0; s < splits; s++) acc += slab[(int64_t)s * total + i];
        int64_t r = i / N, c = i - r * N;
        float v = alpha * acc;
        if (beta != 0.f) v += beta * C[r * ldc + c];
        C[r * ldc + c] = v;
    }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int choose_splits(int64_t M, int64_t N, int64_t K)
{
    int64_t tiles = ceil_div(M, BM) * ceil_div(N, BN);
    int64_t ksteps = ceil_div(K, BK);
    int64_t want = ceil_div(4 * kNumCU, tiles);  // ~4 workgroups per CU
    if (want > ksteps / 8) want = ksteps / 8;    // keep >= 8 K-steps per split
    if (want < 1) want = 1;
    if (want > 1024) want = 1024;
    return (int)want;
}

template <bool A_KC, bool B_KC>
int launch(const GemmArgs &g, int splits, bool va, bool vb, hipStream_t st)
{
    dim3 grid((uint32_t)ceil_div(g.N, BN), (uint32_t)ceil_div(g.M, BM), (uint32_t)splits);
    if (va && vb) hipLaunchKernelGGL((gemm_kernel<A_KC, B_KC, true, true>), grid, dim3(256), 0, st, g);
    else if (va) hipLaunchKernelGGL((gemm_kernel<A_KC, B_KC, true, false>), grid, dim3(256), 0, st, g);
    else if (vb) hipLaunchKernelGGL((gemm_kernel<A_KC, B_KC, false, true>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_kernel<A_KC, B_KC, false, false>), grid, dim3(256), 0, st, g);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}

}  // namespace

GNNX_API int gnnx_gemm_workspace(int transA, int transB, int64_t M, int64_t N, int64_t K, size_t *bytes)
{
    (void)transB;
    GNNX_REQUIRE(bytes && M >= 0 && N >= 0 && K >= 0, GNNX_ERR_INVALID_ARG, "bad arguments");
    *bytes = 0;
    if (transA && M > 0 && N > 0 && K > 0) {
        int splits = choose_splits(M, N, K);
        if (splits > 1) *bytes = sizeof(float) * (size_t)splits * (size_t)M * (size_t)N;
    }
    return GNNX_OK;
}

GNNX_API int gnnx_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const float *d_A,
                           int64_t lda, const float *d_B, int64_t ldb, float beta, float *d_C, int64_t ldc,
                           void *d_workspace, size_t workspace_bytes, void *stream)
{
    GNNX_REQUIRE(M >= 0 && N >= 0 && K >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    if (M == 0 || N == 0) return GNNX_OK;
    GNNX_REQUIRE(d_C && ldc >= N, GNNX_ERR_INVALID_ARG, "C null or ldc < N");
    GNNX_REQUIRE(K == 0 || (d_A && d_B), GNNX_ERR_INVALID_ARG, "null operand");
    GNNX_REQUIRE(K == 0 || lda >= (transA ? M : K), GNNX_ERR_SHAPE, "lda too small");
    GNNX_REQUIRE(K == 0 || ldb >= (transB ? K : N), GNNX_ERR_SHAPE, "ldb too small");
    hipStream_t st = as_stream(stream);
    GemmArgs g{};
    g.M = M; g.N = N; g.K = K;
    g.A = d_A; g.lda = lda; g.B = d_B; g.ldb = ldb; g.C = d_C; g.ldc = ldc;
    g.alpha = alpha; g.beta = beta;
    const bool a_kc = !transA;  // A[M,K] row-major: K contiguous
    const bool b_kc = transB;   // B given as [N,K] row-major: K contiguous
    const bool va = aligned16(d_A) && lda % 4 == 0 && (a_kc ? K % 4 == 0 : M % 4 == 0);
    const bool vb = aligned16(d_B) && ldb % 4 == 0 && (b_kc ? K % 4 == 0 : N % 4 == 0);
    int splits = 1;
    if (transA && K > 0) splits = choose_splits(M, N, K);
    int64_t ksteps = ceil_div(K > 0 ? K : 1, BK);
    g.k_per_split = ceil_div(ksteps, splits) * BK;
    if (splits > 1) {
        size_t need = sizeof(float) * (size_t)splits * (size_t)M * (size_t)N;
        GNNX_REQUIRE(d_workspace && workspace_bytes >= need, GNNX_ERR_WORKSPACE, "workspace %zu < required %zu",
                     workspace_bytes, need);
        g.slab = static_cast<float *>(d_workspace);
    }
    int rc;
    if (a_kc && b_kc) rc = launch<true, true>(g, splits, va, vb, st);
    else if (a_kc && !b_kc) rc = launch<true, false>(g, splits, va, vb, st);
    else if (!a_kc && b_kc) rc = launch<false, true>(g, splits, va, vb, st);
    else rc = launch<false, false>(g, splits, va, vb, st);
    if (rc != GNNX_OK) return rc;
    if (splits > 1) {
        int64_t blocks = ceil_div(M * N, 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, g.slab, splits, M, N, alpha, beta,
                           d_C, ldc);
        GNNX_LAUNCH_CHECK();
    }
    return GNNX_OK;
}
