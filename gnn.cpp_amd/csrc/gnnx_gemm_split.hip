// OPT-IN split-precision GEMM (NOT the default path, NOT used by any parity-graded call): the dense transform on the bf16
// matrix cores with f32-level accuracy.
//
// fp32 MFMA is 1/16 of the bf16 rate on gfx950 and the three GEMMs are ~31 of the layer step's 70 ms.  Every f32 operand is
// split EXACTLY into three bf16 pieces, x = x0 + x1 + x2 (8 + 8 + 8 significand bits: x0 = bf16(x), x1 = bf16(x - x0),
// x2 = bf16(x - x0 - x1); the two subtractions are exact in f32), and the product is accumulated in f32 from the six piece
// products with i + j <= 2, smallest first:  a.b ~= a2.b0 + a1.b1 + a0.b2 + a1.b0 + a0.b1 + a0.b0  (each piece product is
// exact in f32: 8 x 8 bits).  The three dropped terms are <= 3 * 2^-24 |a||b| per product, the size of one f32 rounding.
// Measured against float64 (tests/test_gpu_parity.py::test_split_gemm_*) the error is within 2-3x of the f32 FMA chain's and
// two orders of magnitude inside the 1e-5 parity bar -- but it is not the same arithmetic as the reference's f32 products,
// so it stays behind its own entry point (gnnx_gemm_split_bf16_f32) and bench.py --split-gemm.
//
// Kernel: C[M,N] = A[M,K] . B  with A row-major f32 (K contiguous), B pre-split once into bf16 pieces [3][N][K]
// (split_b_kernel; W is small).  128 x (64 NB) output tile per 256-thread workgroup (4 wavefronts as 2 x 2, each
// 64 x 32 NB = 2 x NB MFMA tiles of v_mfma_f32_32x32x16_bf16), K step 16, LDS double-buffered:
//   operand fragment of that MFMA = 8 consecutive k for one row = 16 bytes; LDS holds [piece][k-group][row][16 B], so a
//   fragment read is one conflict-free ds_read_b128 (a half-wave reads 512 contiguous bytes);
//   A: each thread loads 8 consecutive k of one row (two 16-byte loads), splits them in registers (v_cvt_pk_bf16_f32 +
//   shifts + two exact subtractions) and writes three ds_write_b128; B pieces are copied as they are.
#include <cstdlib>
#include <cstring>

#include "gnnx_common.h"

using namespace gnnx;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float bf16_to_f32(__bf16 h)
{
    union { __bf16 h; uint16_t u; } v;
    v.h = h;
    return __uint_as_float((uint32_t)v.u << 16);
}

// x -> (x0, x1, x2) with x0 + x1 + x2 == x exactly (barring underflow of the last piece)
__device__ __forceinline__ void split3(float x, __bf16 &p0, __bf16 &p1, __bf16 &p2)
{
    p0 = (__bf16)x;
    const float r1 = x - bf16_to_f32(p0);
    p1 = (__bf16)r1;
    const float r2 = r1 - bf16_to_f32(p1);
    p2 = (__bf16)r2;
}

// B (f32, [N][K] row-major if b_kc else [K][N]) -> pieces [3][N][K] bf16
__global__ __launch_bounds__(256) void split_b_kernel(const float *B, int64_t ldb, int b_kc, int32_t N, int32_t K, uint16_t *out)
{
    const int64_t total = (int64_t)N * K;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int32_t n = (int32_t)(i / K), k = (int32_t)(i - (int64_t)n * K);
        const float x = b_kc ? B[(int64_t)n * ldb + k] : B[(int64_t)k * ldb + n];
        __bf16 p0, p1, p2;
        split3(x, p0, p1, p2);
        union { __bf16 h; uint16_t u; } c;
        c.h = p0; out[i] = c.u;
        c.h = p1; out[total + i] = c.u;
        c.h = p2; out[2 * total + i] = c.u;
    }
}

struct SplitArgs {
    int64_t M;
    int32_t N, K;
    const float *A;
    int64_t lda;
    const uint16_t *Bp;  // [3][N][K]
    float *C;
    int64_t ldc;
};

// NB: 32-column MFMA tiles per wavefront (BN = 64 * NB columns per workgroup)
template <int NB>
__global__ __launch_bounds__(256) void gemm_split_kernel(SplitArgs g)
{
    constexpr int BM = 128, BN = 64 * NB, BK = 16;
    constexpr int A_STAGE = 3 * 2 * BM * 16;   // bytes: [piece][k-group][row][16]
    constexpr int B_STAGE = 3 * 2 * BN * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // [2][A_STAGE] then [2][B_STAGE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * BM;
    const int32_t n0 = blockIdx.x * BN;

    // loader roles: A: row = tid / 2, k-group = tid % 2 (8 consecutive k = 32 bytes of a row)
    const int arow = tid >> 1, akg = tid & 1;
    int64_t ar = m0 + arow;
    ar = ar < g.M ? ar : g.M - 1;  // clamped: rows past M are computed on a valid row and never stored
    const float *ap = g.A + ar * g.lda + akg * 8;
    // B pieces: NB * 64 columns x 2 k-groups = 128 NB 16-byte granules per piece; 256 threads => NB / 2 granules each per piece
    // (NB = 2: one, NB = 4: two)
    constexpr bool TWO = NB == 4;  // second granule per thread and piece for the 256-wide tile
    const int col0 = tid >> 1, kg0 = tid & 1;
    const uint16_t *bpa = g.Bp + (int64_t)(n0 + col0) * g.K + kg0 * 8;
    const int boffa = (kg0 * BN + col0) * 16;
    const int col1 = (tid + 256) >> 1;  // same k-group
    const uint16_t *bpb = g.Bp + (int64_t)(n0 + (TWO ? col1 : col0)) * g.K + kg0 * 8;
    const int boffb = (kg0 * BN + (TWO ? col1 : col0)) * 16;
    const int64_t piece_stride = (int64_t)g.N * g.K;

    // Register staging of the next K-tile.  (A three-tile-deep register ring for the A stream -- the only HBM stream, the B
    // pieces sit in L2 -- was tried: 176 instead of 148 registers costs a resident workgroup and is slower, 10.1 vs 9.6 ms at
    // 10M x 256 x 256.)  Plain variables and macros: arrays captured by a lambda ended up in scratch.
    float4 ra0, ra1;
    uint4 rb0a, rb1a, rb2a, rb0b, rb1b, rb2b;
#define SPLIT_LOAD(k0_)                                                                         \
    do {                                                                                        \
        ra0 = *reinterpret_cast<const float4 *>(ap + (k0_));                                    \
        ra1 = *reinterpret_cast<const float4 *>(ap + (k0_) + 4);                                \
        rb0a = *reinterpret_cast<const uint4 *>(bpa + (k0_));                                   \
        rb1a = *reinterpret_cast<const uint4 *>(bpa + piece_stride + (k0_));                    \
        rb2a = *reinterpret_cast<const uint4 *>(bpa + 2 * piece_stride + (k0_));                \
        if constexpr (TWO) {                                                                    \
            rb0b = *reinterpret_cast<const uint4 *>(bpb + (k0_));                               \
            rb1b = *reinterpret_cast<const uint4 *>(bpb + piece_stride + (k0_));                \
            rb2b = *reinterpret_cast<const uint4 *>(bpb + 2 * piece_stride + (k0_));            \
        }                                                                                       \
    } while (0)
#define SPLIT_STASH(buf_)                                                                       \
    do {                                                                                        \
        bf16x8 s0, s1, s2;                                                                      \
        __bf16 t0, t1, t2;                                                                      \
        split3(ra0.x, t0, t1, t2); s0[0] = t0; s1[0] = t1; s2[0] = t2;                          \
        split3(ra0.y, t0, t1, t2); s0[1] = t0; s1[1] = t1; s2[1] = t2;                          \
        split3(ra0.z, t0, t1, t2); s0[2] = t0; s1[2] = t1; s2[2] = t2;                          \
        split3(ra0.w, t0, t1, t2); s0[3] = t0; s1[3] = t1; s2[3] = t2;                          \
        split3(ra1.x, t0, t1, t2); s0[4] = t0; s1[4] = t1; s2[4] = t2;                          \
        split3(ra1.y, t0, t1, t2); s0[5] = t0; s1[5] = t1; s2[5] = t2;                          \
        split3(ra1.z, t0, t1, t2); s0[6] = t0; s1[6] = t1; s2[6] = t2;                          \
        split3(ra1.w, t0, t1, t2); s0[7] = t0; s1[7] = t1; s2[7] = t2;                          \
        unsigned char *a_ = lds + (buf_) * A_STAGE + (akg * BM + arow) * 16;                    \
        *reinterpret_cast<bf16x8 *>(a_) = s0;                                                   \
        *reinterpret_cast<bf16x8 *>(a_ + 2 * BM * 16) = s1;                                     \
        *reinterpret_cast<bf16x8 *>(a_ + 4 * BM * 16) = s2;                                     \
        unsigned char *b_ = lds + 2 * A_STAGE + (buf_) * B_STAGE;                               \
        *reinterpret_cast<uint4 *>(b_ + boffa) = rb0a;                                          \
        *reinterpret_cast<uint4 *>(b_ + 2 * BN * 16 + boffa) = rb1a;                            \
        *reinterpret_cast<uint4 *>(b_ + 4 * BN * 16 + boffa) = rb2a;                            \
        if constexpr (TWO) {                                                                    \
            *reinterpret_cast<uint4 *>(b_ + boffb) = rb0b;                                      \
            *reinterpret_cast<uint4 *>(b_ + 2 * BN * 16 + boffb) = rb1b;                        \
            *reinterpret_cast<uint4 *>(b_ + 4 * BN * 16 + boffb) = rb2b;                        \
        }                                                                                       \
    } while (0)

    f32x16 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < NB; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    SPLIT_LOAD(0);
    SPLIT_STASH(0);
    __syncthreads();
    int buf = 0;
    for (int32_t k0 = 0; k0 < g.K; k0 += BK) {
        const bool more = k0 + BK < g.K;
        if (more) SPLIT_LOAD(k0 + BK);
        // fragments: lane -> row (lane & 31) of its 32-row tile, k-group (lane >> 5)
        const unsigned char *as = lds + buf * A_STAGE + ((lane >> 5) * BM + wm * 64 + (lane & 31)) * 16;
        const unsigned char *bs = lds + 2 * A_STAGE + buf * B_STAGE + ((lane >> 5) * BN + wn * 32 * NB + (lane & 31)) * 16;
        bf16x8 a[3][2];
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int i = 0; i < 2; i++) a[p][i] = *reinterpret_cast<const bf16x8 *>(as + p * 2 * BM * 16 + i * 32 * 16);
        // column tiles two at a time: only six B fragments are live at once (the 256-wide tile spilled with all twelve)
#pragma unroll
        for (int jh = 0; jh < NB; jh += 2) {
            bf16x8 b[3][2];
#pragma unroll
            for (int p = 0; p < 3; p++)
#pragma unroll
                for (int j = 0; j < 2; j++) b[p][j] = *reinterpret_cast<const bf16x8 *>(bs + p * 2 * BN * 16 + (jh + j) * 32 * 16);
            // six piece products per MFMA tile, smallest first
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    f32x16 c = acc[i][jh + j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], c, 0, 0, 0);
                    acc[i][jh + j] = c;
                }
        }
        if (more) SPLIT_STASH(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

#undef SPLIT_LOAD
#undef SPLIT_STASH
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const int32_t col = n0 + wn * 32 * NB + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < g.M) g.C[row * g.ldc + col] = acc[i][j][r];
            }
        }
}

template <int NB>
int launch_split(const SplitArgs &g, hipStream_t st)
{
    constexpr size_t lds = 2 * (3 * 2 * 128 * 16 + 3 * 2 * 64 * NB * 16);
    static std::atomic<uint64_t> attr_done{0};
    const int rc_lds = lds_opt_in(&gemm_split_kernel<NB>, lds, attr_done, "gemm_split_kernel");
    if (rc_lds != GNNX_OK) return rc_lds;
    dim3 grid((uint32_t)(g.N / (64 * NB)), (uint32_t)ceil_div(g.M, 128));
    hipLaunchKernelGGL((gemm_split_kernel<NB>), grid, dim3(256), lds, st, g);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}

}  // namespace

GNNX_API int gnnx_gemm_split_workspace(int64_t M, int64_t N, int64_t K, size_t *bytes)
{
    GNNX_REQUIRE(bytes && M >= 0 && N >= 0 && K >= 0, GNNX_ERR_INVALID_ARG, "bad arguments");
    *bytes = sizeof(uint16_t) * 3 * (size_t)N * (size_t)K;
    return GNNX_OK;
}

GNNX_API int gnnx_gemm_split_bf16_f32(int transB, int64_t M, int64_t N, int64_t K, const float *d_A, int64_t lda, const float *d_B,
                                      int64_t ldb, float *d_C, int64_t ldc, void *d_workspace, size_t workspace_bytes, void *stream)
{
    GNNX_REQUIRE(M >= 0 && N >= 0 && K >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    if (M == 0 || N == 0) return GNNX_OK;
    GNNX_REQUIRE(d_A && d_B && d_C, GNNX_ERR_INVALID_ARG, "null operand");
    GNNX_REQUIRE(K > 0 && K % 16 == 0 && N % 128 == 0, GNNX_ERR_UNSUPPORTED, "split GEMM needs K %% 16 == 0 and N %% 128 == 0 (got N=%lld K=%lld)",
                 (long long)N, (long long)K);
    GNNX_REQUIRE(N < (1ll << 31) && K < (1ll << 31), GNNX_ERR_UNSUPPORTED, "N, K must fit 32 bits");
    GNNX_REQUIRE(lda >= K && lda % 4 == 0 && (reinterpret_cast<uintptr_t>(d_A) & 15u) == 0, GNNX_ERR_UNSUPPORTED,
                 "A must be 16-byte aligned with lda %% 4 == 0");
    GNNX_REQUIRE(ldb >= (transB ? K : N) && ldc >= N, GNNX_ERR_SHAPE, "leading dimension too small");
    size_t need = 0;
    gnnx_gemm_split_workspace(M, N, K, &need);
    GNNX_REQUIRE(d_workspace && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(d_workspace) & 15u) == 0, GNNX_ERR_WORKSPACE,
                 "workspace %zu < required %zu (or not 16-byte aligned)", workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    uint16_t *bp = static_cast<uint16_t *>(d_workspace);
    int64_t blocks = ceil_div(N * K, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split_b_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, d_B, ldb, transB ? 1 : 0, (int32_t)N, (int32_t)K, bp);
    GNNX_LAUNCH_CHECK();
    SplitArgs g{M, (int32_t)N, (int32_t)K, d_A, lda, bp, d_C, ldc};
    static const int nb_env = [] { const char *e = experiment_env("GNNX_SPLIT_NB"); return e ? atoi(e) : 0; }();  // A/B: force the tile width
    if (N % 256 == 0 && nb_env != 2) return launch_split<4>(g, st);
    return launch_split<2>(g, st);
}
