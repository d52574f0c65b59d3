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
#include <cstdlib>
#include <cstring>

#include "gnnx_common.h"

using namespace gnnx;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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

// Tile configuration: BM x BN output tile, BK K-step, WM x WN wavefronts (each (BM/WM) x (BN/WN), made of 32x32 MFMA tiles).
template <int BM_, int BN_, int BK_, int WM_, int WN_>
struct Cfg {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_;
    static constexpr int NT = 64 * WM * WN;                     // threads
    static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;  // MFMA tiles per wavefront
    // LDS row strides (words).  A K-contiguous operand is transposed on the way in (b32 writes): stride % 32 == 2 keeps those
    // at <= 2-way conflicts.  A k-major operand goes in as it is: stride % 4 == 0 makes its 4 consecutive words one 16-byte
    // ds_write_b128.  Fragment reads are 32 consecutive words per half-wave either way.
    static constexpr int ld_a(bool kc) { return kc ? BM + 2 : BM + 4; }
    static constexpr int ld_b(bool kc) { return kc ? BN + 2 : BN + 4; }
    static constexpr size_t lds_bytes(bool a_kc, bool b_kc) { return sizeof(float) * 2 * BK * (ld_a(a_kc) + ld_b(b_kc)); }
    static constexpr int A_VECS = BM * BK / 4 / NT;   // float4 per thread per tile
    static constexpr int B_VECS = BN * BK / 4 / NT;
    static_assert(A_VECS >= 1 && B_VECS >= 1, "tile too small for the thread count");
};

// Global -> registers for one operand tile [ROWS (m or n)] x [BK k].
// KC (K-contiguous): element (r, k) at base[r*ld + k]  (X, W^T-as-W).   !KC (k-major): base[k*ld + r].
// VEC: 16-byte loads are legal (ld % 4 == 0, base 16-B aligned, and the contiguous extent % 4 == 0).
// Loads are never predicated (see gnnx_spmm.hip) and their results are NOT touched here: addresses are clamped
// into the matrix and out-of-range elements are zeroed later, in store_tile -- a select right after the load
// would make the compiler wait for HBM before the MFMA loop instead of after it.
template <bool KC, bool VEC, int ROWS, int BK, int NT, int NV>
__device__ __forceinline__ void load_tile(float (&reg)[NV * 4], const float *base, int64_t ld, int64_t r0, int64_t rmax,
                                          int64_t k0, int64_t kmax, int tid)
{
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int idx = tid + i * NT;
        int64_t r, k;  // first of 4 consecutive elements along the contiguous axis
        if constexpr (KC) {
            r = r0 + idx / (BK / 4);
            k = k0 + (idx % (BK / 4)) * 4;
            const int64_t rc = r < rmax ? r : rmax - 1;
            if constexpr (VEC) {
                const float4 v = *reinterpret_cast<const float4 *>(base + rc * ld + (k < kmax ? k : 0));
                reg[4 * i + 0] = v.x;
                reg[4 * i + 1] = v.y;
                reg[4 * i + 2] = v.z;
                reg[4 * i + 3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) reg[4 * i + j] = base[rc * ld + ((k + j) < kmax ? (k + j) : 0)];
            }
        } else {
            k = k0 + idx / (ROWS / 4);
            r = r0 + (idx % (ROWS / 4)) * 4;
            const int64_t kc = k < kmax ? k : kmax - 1;
            if constexpr (VEC) {
                const float4 v = *reinterpret_cast<const float4 *>(base + kc * ld + (r < rmax ? r : 0));
                reg[4 * i + 0] = v.x;
                reg[4 * i + 1] = v.y;
                reg[4 * i + 2] = v.z;
                reg[4 * i + 3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) reg[4 * i + j] = base[kc * ld + ((r + j) < rmax ? (r + j) : 0)];
            }
        }
    }
}

// registers -> LDS tile [BK k][LD] (k-major).  MASK = false: the whole tile is inside the matrix (the hot case, no
// selects); MASK = true: out-of-range elements are written as zero.
template <bool KC, bool MASK, int ROWS, int BK, int NT, int NV, int LD>
__device__ __forceinline__ void store_tile_impl(float *lds, const float (&reg)[NV * 4], int tid, int64_t r0, int64_t rmax,
                                                int64_t k0, int64_t kmax)
{
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int idx = tid + i * NT;
        if constexpr (KC) {
            const int r = idx / (BK / 4), k = (idx % (BK / 4)) * 4;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v = reg[4 * i + j];
                if constexpr (MASK) v = (r0 + r < rmax && k0 + k + j < kmax) ? v : 0.f;
                lds[(k + j) * LD + r] = v;
            }
        } else {
            const int k = idx / (ROWS / 4), r = (idx % (ROWS / 4)) * 4;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                v[j] = reg[4 * i + j];
                if constexpr (MASK) v[j] = (k0 + k < kmax && r0 + r + j < rmax) ? v[j] : 0.f;
            }
            if constexpr (LD % 4 == 0) {  // one ds_write_b128
                *reinterpret_cast<float4 *>(lds + k * LD + r) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) lds[k * LD + r + j] = v[j];
            }
        }
    }
}

template <bool KC, int ROWS, int BK, int NT, int NV, int LD>
__device__ __forceinline__ void store_tile(float *lds, const float (&reg)[NV * 4], int tid, bool full, int64_t r0, int64_t rmax,
                                           int64_t k0, int64_t kmax)
{
    if (full) store_tile_impl<KC, false, ROWS, BK, NT, NV, LD>(lds, reg, tid, r0, rmax, k0, kmax);
    else store_tile_impl<KC, true, ROWS, BK, NT, NV, LD>(lds, reg, tid, r0, rmax, k0, kmax);
}

template <class C, bool A_KC, bool B_KC, bool VEC_A, bool VEC_B>
__global__ __launch_bounds__(C::NT) void gemm_kernel(GemmArgs g)
{
    constexpr int BM = C::BM, BN = C::BN, BK = C::BK, NT = C::NT, TM = C::TM, TN = C::TN;
    constexpr int LDA = C::ld_a(A_KC), LDB = C::ld_b(B_KC);
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];  // [2][BK*LDA] A tiles, then [2][BK*LDB] B tiles
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
    const int64_t kbeg = (int64_t)blockIdx.z * g.k_per_split;
    const int64_t kend = kbeg + g.k_per_split < g.K ? kbeg + g.k_per_split : g.K;
    const bool full_mn = m0 + BM <= g.M && n0 + BN <= g.N;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    constexpr int STORE_AT = (BK * 3 / 4) & ~1;  // k-step after which the next tile is written to LDS
    float ra[C::A_VECS * 4], rb[C::B_VECS * 4];
    int buf = 0;
    if (kbeg < kend) {
        const bool full = full_mn && kbeg + BK <= kend;
        load_tile<A_KC, VEC_A, BM, BK, NT, C::A_VECS>(ra, g.A, g.lda, m0, g.M, kbeg, kend, tid);
        load_tile<B_KC, VEC_B, BN, BK, NT, C::B_VECS>(rb, g.B, g.ldb, n0, g.N, kbeg, kend, tid);
        store_tile<A_KC, BM, BK, NT, C::A_VECS, LDA>(lds_raw, ra, tid, full, m0, g.M, kbeg, kend);
        store_tile<B_KC, BN, BK, NT, C::B_VECS, LDB>(lds_raw + 2 * BK * LDA, rb, tid, full, n0, g.N, kbeg, kend);
    }
    __syncthreads();
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = k0 + BK < kend;
        if (more) {  // next tile: HBM -> registers, in flight during the MFMAs below
            load_tile<A_KC, VEC_A, BM, BK, NT, C::A_VECS>(ra, g.A, g.lda, m0, g.M, k0 + BK, kend, tid);
            load_tile<B_KC, VEC_B, BN, BK, NT, C::B_VECS>(rb, g.B, g.ldb, n0, g.N, k0 + BK, kend, tid);
        }
        const float *As = lds_raw + buf * BK * LDA + wm * (BM / C::WM) + (lane & 31) + (lane >> 5) * LDA;
        const float *Bs = lds_raw + 2 * BK * LDA + buf * BK * LDB + wn * (BN / C::WN) + (lane & 31) + (lane >> 5) * LDB;
        // MFMA operand fragments are read one k-step ahead of their use (two register sets, static indices)
        float a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; i++) a[0][i] = As[32 * i];
#pragma unroll
        for (int j = 0; j < TN; j++) b[0][j] = Bs[32 * j];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
            if (kk + 2 < BK) {
#pragma unroll
                for (int i = 0; i < TM; i++) a[nxt][i] = As[(kk + 2) * LDA + 32 * i];
#pragma unroll
                for (int j = 0; j < TN; j++) b[nxt][j] = Bs[(kk + 2) * LDB + 32 * j];
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the fragment reads of step kk+2 AHEAD of the MFMAs of step kk
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk == STORE_AT && more) {
                // tile t+1: registers -> the OTHER LDS buffer, three quarters into the MFMAs of tile t: the HBM loads
                // issued at the top have landed by now and the ds_writes hide under the remaining MFMAs, so only the
                // barrier itself sits between two tiles
                const bool full = full_mn && k0 + 2 * BK <= kend;
                store_tile<A_KC, BM, BK, NT, C::A_VECS, LDA>(lds_raw + (buf ^ 1) * BK * LDA, ra, tid, full, m0, g.M, k0 + BK, kend);
                store_tile<B_KC, BN, BK, NT, C::B_VECS, LDB>(lds_raw + 2 * BK * LDA + (buf ^ 1) * BK * LDB, rb, tid, full, n0, g.N,
                                                             k0 + BK, kend);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        buf ^= 1;
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float *out = g.slab ? g.slab + (int64_t)blockIdx.z * g.M * g.N : g.C;
    const int64_t ldo = g.slab ? g.N : g.ldc;
    const float alpha = g.slab ? 1.f : g.alpha;
    const float beta = g.slab ? 0.f : g.beta;
    float *obase = out + (m0 + wm * (BM / C::WM) + 4 * (lane >> 5)) * ldo + n0 + wn * (BN / C::WN) + (lane & 31);
    if (full_mn && beta == 0.f) {  // hot case: no bounds checks, no read-modify-write, no branches between stores
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    obase[(int64_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldo + j * 32] = alpha * acc[i][j][r];
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int64_t col = n0 + wn * (BN / C::WN) + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int64_t row = m0 + wm * (BM / C::WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < g.M && col < g.N) {
                    float v = alpha * acc[i][j][r];
                    if (beta != 0.f) v += beta * out[row * ldo + col];
                    out[row * ldo + col] = v;
                }
            }
        }
}

// ---- streaming variant for tall products (X.W^T, dH.W on millions of rows) -------------------------------------------
// A 10M x 256 x 256 product is 39k output tiles of only K / BK = 8 K-tiles each: with one output tile per workgroup the
// exposed first loads and the C store cost ~10 % (ablation build).  Here a RESIDENT workgroup walks the
// M-tiles blockIdx.y, + gridDim.y, ... of its column and the K-tile stream runs ACROSS them: while the last K-tile of output
// tile t is multiplied, the first K-tile of the next one is already on its way HBM -> registers -> LDS, and the C store of
// tile t (fire-and-forget) drains under the MFMAs of tile t + 1.  Only whole tiles: M-tiles inside [0, m_tiles), K % BK == 0,
// N % BN == 0, 16-byte aligned operands, beta == 0 (the host sends the ragged remainder to gemm_kernel).  Addresses are a
// wave-uniform 64-bit base + a per-lane 32-bit offset fixed for the whole launch, so the walk costs no vector registers.
template <class C, bool B_KC>
__global__ __launch_bounds__(C::NT) void gemm_stream_kernel(GemmArgs g, int64_t m_tiles)
{
    constexpr int BM = C::BM, BN = C::BN, BK = C::BK, NT = C::NT, TM = C::TM, TN = C::TN;
    constexpr int LDA = C::ld_a(true), LDB = C::ld_b(B_KC);
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];  // [2][BK*LDA] A tiles, then [2][BK*LDB] B tiles
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int64_t n0 = (int64_t)blockIdx.x * BN;

    // per-lane element offsets of this thread's float4 slots inside an operand tile (A is K-contiguous: [BM rows][BK k])
    uint32_t offa[C::A_VECS], offb[C::B_VECS];
#pragma unroll
    for (int i = 0; i < C::A_VECS; i++) {
        const int idx = tid + i * NT;
        offa[i] = (uint32_t)((idx / (BK / 4)) * g.lda + (idx % (BK / 4)) * 4);
    }
#pragma unroll
    for (int i = 0; i < C::B_VECS; i++) {
        const int idx = tid + i * NT;
        if constexpr (B_KC) offb[i] = (uint32_t)((idx / (BK / 4)) * g.ldb + (idx % (BK / 4)) * 4);  // [BN rows][BK k]
        else offb[i] = (uint32_t)((idx / (BN / 4)) * g.ldb + (idx % (BN / 4)) * 4);               // [BK k][BN cols]
    }
    const float *bcol = g.B + (B_KC ? n0 * g.ldb : n0);
    const int64_t bstep = B_KC ? 1 : g.ldb;  // elements per unit of k
    const uint32_t offc = (uint32_t)((wm * (BM / C::WM) + 4 * (lane >> 5)) * g.ldc + wn * (BN / C::WN) + (lane & 31));

    float ra[C::A_VECS * 4], rb[C::B_VECS * 4];
    auto load = [&](int64_t mt, int64_t k0) {
        const float *abase = g.A + mt * BM * g.lda + k0;
        const float *bbase = bcol + k0 * bstep;
#pragma unroll
        for (int i = 0; i < C::A_VECS; i++) {
            const float4 v = *reinterpret_cast<const float4 *>(abase + offa[i]);
            ra[4 * i + 0] = v.x; ra[4 * i + 1] = v.y; ra[4 * i + 2] = v.z; ra[4 * i + 3] = v.w;
        }
#pragma unroll
        for (int i = 0; i < C::B_VECS; i++) {
            const float4 v = *reinterpret_cast<const float4 *>(bbase + offb[i]);
            rb[4 * i + 0] = v.x; rb[4 * i + 1] = v.y; rb[4 * i + 2] = v.z; rb[4 * i + 3] = v.w;
        }
    };
    auto stash = [&](int b) {  // registers -> LDS buffer b (whole tiles: no masks)
        store_tile_impl<true, false, BM, BK, NT, C::A_VECS, LDA>(lds_raw + b * BK * LDA, ra, tid, 0, 0, 0, 0);
        store_tile_impl<B_KC, false, BN, BK, NT, C::B_VECS, LDB>(lds_raw + 2 * BK * LDA + b * BK * LDB, rb, tid, 0, 0, 0, 0);
    };

    f32x16 acc[TM][TN];
    constexpr int STORE_AT = (BK * 3 / 4) & ~1;
    int buf = 0;
    int64_t mt = blockIdx.y;
    if (mt < m_tiles) {
        load(mt, 0);
        stash(0);
    }
    __syncthreads();
    for (; mt < m_tiles; mt += gridDim.y) {
        const int64_t mt_next = mt + gridDim.y;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
        for (int64_t k0 = 0; k0 < g.K; k0 += BK) {
            const bool last_k = k0 + BK >= g.K;
            const bool more = !last_k || mt_next < m_tiles;
            if (more) load(last_k ? mt_next : mt, last_k ? 0 : k0 + BK);  // in flight during the MFMAs below
            const float *As = lds_raw + buf * BK * LDA + wm * (BM / C::WM) + (lane & 31) + (lane >> 5) * LDA;
            const float *Bs = lds_raw + 2 * BK * LDA + buf * BK * LDB + wn * (BN / C::WN) + (lane & 31) + (lane >> 5) * LDB;
            float a[2][TM], b[2][TN];
#pragma unroll
            for (int i = 0; i < TM; i++) a[0][i] = As[32 * i];
#pragma unroll
            for (int j = 0; j < TN; j++) b[0][j] = Bs[32 * j];
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
                if (kk + 2 < BK) {
#pragma unroll
                    for (int i = 0; i < TM; i++) a[nxt][i] = As[(kk + 2) * LDA + 32 * i];
#pragma unroll
                    for (int j = 0; j < TN; j++) b[nxt][j] = Bs[(kk + 2) * LDB + 32 * j];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk == STORE_AT && more) {
                    stash(buf ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
            buf ^= 1;
        }
        // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
        float *cbase = g.C + mt * BM * g.ldc + n0;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    (cbase + (int64_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * g.ldc + j * 32)[offc] = g.alpha * acc[i][j][r];
    }
}

// Split-K partials -> C in a FIXED order (deterministic, no float atomics): 8 lane groups each sum a contiguous eighth of the
// slabs for 32 consecutive elements (128-byte coalesced reads), then the eight partial sums are combined left to right.
// (One thread walking all slabs of its element was a chain of up to 1024 dependent loads: 0.31 ms next to a 0.30 ms GEMM
// at 1M x 128 x 128.)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *slab, int splits, int64_t M, int64_t N, float alpha,
                                                            float beta, float *C, int64_t ldc)
{
    __shared__ float red[8][32];
    const int e = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int64_t total = M * N;
    const int64_t i = (int64_t)blockIdx.x * 32 + e;
    const int per = (splits + 7) / 8;
    const int s0 = q * per, s1 = s0 + per < splits ? s0 + per : splits;
    float acc = 0.f;
    if (i < total)
        for (int s = s0; s < s1; s++) acc += slab[(int64_t)s * total + i];
    red[q][e] = acc;
    __syncthreads();
    if (q == 0 && i < total) {
        float sum = red[0][e];
#pragma unroll
        for (int k = 1; k < 8; k++) sum += red[k][e];
        const int64_t r = i / N, c = i - r * N;
        float v = alpha * sum;
        if (beta != 0.f) v += beta * C[r * ldc + c];
        C[r * ldc + c] = v;
    }
}

// Calibration: register-only MFMA loop (4 independent accumulators per wavefront, no memory traffic) -- what the
// fp32 matrix pipe sustains on THIS chip at the clock it holds under load; the GEMM's fraction of that is the honest
// utilisation figure next to the 157.3 TFLOP/s datasheet peak.
__global__ __launch_bounds__(256) void mfma_peak_kernel(int iters, float *sink)
{
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = (float)(threadIdx.x + r);
    float a = 1.0f + threadIdx.x * 1e-3f, b = 1.0f - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][7];
    if (s == 12345.678f) sink[0] = s;  // keep the loop alive
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

using CfgDefault = Cfg<128, 128, 32, 2, 2>;  // 256 threads, 66 KB LDS, 2 workgroups per CU
using CfgWide = Cfg<128, 256, 16, 2, 4>;     // 512 threads, 49 KB LDS: one pass over A for 256-wide outputs
using CfgK16 = Cfg<128, 128, 16, 2, 2>;      // 33 KB LDS: 3 workgroups per CU
using CfgWide32 = Cfg<128, 256, 32, 2, 4>;   // 512 threads, 97 KB LDS: one workgroup per CU, half the barriers
using CfgTall = Cfg<256, 256, 16, 4, 4>;     // 1024 threads, 66 KB LDS
using CfgTall32 = Cfg<256, 256, 32, 4, 4>;   // 1024 threads, 132 KB LDS: half the barriers

// Tile choice.  Default ("auto"): 256 x 256 x 32 / 1024 threads (132 KB LDS) when the output is at least that big (A and B are each
// streamed once for a 256-wide output: 118-121 TFLOP/s at the bench shape vs 111 for 128 x 128), 128 x 256 for wide
// but short outputs, 128 x 128 otherwise.  GNNX_GEMM_TILE=square|wide|k16|wide32|tall forces one (A/B experiments).
enum TileId { kSquare = 0, kWide = 1, kK16 = 2, kWide32 = 3, kTall = 4, kTall32 = 5 };

int forced_tile()
{
    static const int v = [] {
        const char *e = getenv("GNNX_GEMM_TILE");
        if (!e) return -1;
        if (!strcmp(e, "square")) return (int)kSquare;
        if (!strcmp(e, "wide")) return (int)kWide;
        if (!strcmp(e, "k16")) return (int)kK16;
        if (!strcmp(e, "wide32")) return (int)kWide32;
        if (!strcmp(e, "tall")) return (int)kTall;
        if (!strcmp(e, "tall32")) return (int)kTall32;
        return -1;
    }();
    return v;
}

int pick_tile(int64_t M, int64_t N)
{
    int f = forced_tile();
    if (f == kK16 || f == kSquare) return f;
    if (N <= 128) return kSquare;
    if (f == kWide || f == kWide32) return f;
    if (f == kTall || f == kTall32) return f;
    return M >= 256 ? kTall32 : kWide;
}

struct TileDims { int bm, bn, bk; };
TileDims tile_dims(int64_t M, int64_t N)
{
    switch (pick_tile(M, N)) {
    case kWide: return {CfgWide::BM, CfgWide::BN, CfgWide::BK};
    case kK16: return {CfgK16::BM, CfgK16::BN, CfgK16::BK};
    case kWide32: return {CfgWide32::BM, CfgWide32::BN, CfgWide32::BK};
    case kTall: return {CfgTall::BM, CfgTall::BN, CfgTall::BK};
    case kTall32: return {CfgTall32::BM, CfgTall32::BN, CfgTall32::BK};
    default: return {CfgDefault::BM, CfgDefault::BN, CfgDefault::BK};
    }
}

int choose_splits(int64_t M, int64_t N, int64_t K)
{
    TileDims t = tile_dims(M, N);
    int64_t tiles = ceil_div(M, t.bm) * ceil_div(N, t.bn);
    int64_t ksteps = ceil_div(K, t.bk);
    int64_t per_cu = t.bm * t.bn >= 256 * 256 ? 1 : 2;  // workgroups per CU wanted (= resident: LDS fits 1 resp. 2)
    int64_t want = ceil_div(per_cu * kNumCU, tiles);
    if (want > ksteps / 8) want = ksteps / 8;  // keep >= 8 K-steps per split
    if (want < 1) want = 1;
    if (want > 1024) want = 1024;
    return (int)want;
}

template <class C, bool A_KC, bool B_KC, bool VA, bool VB>
int launch_one(const GemmArgs &g, int splits, hipStream_t st)
{
    dim3 grid((uint32_t)ceil_div(g.N, C::BN), (uint32_t)ceil_div(g.M, C::BM), (uint32_t)splits);
    constexpr size_t lds = C::lds_bytes(A_KC, B_KC);
    static bool attr_set = false;  // > 64 KB of dynamic LDS needs the opt-in, once per kernel instantiation
    if (!attr_set) {
        GNNX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_kernel<C, A_KC, B_KC, VA, VB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<C, A_KC, B_KC, VA, VB>), grid, dim3(C::NT), lds, st, g);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}

template <class C, bool A_KC, bool B_KC>
int launch_cfg(const GemmArgs &g, int splits, bool va, bool vb, hipStream_t st)
{
    if (va && vb) return launch_one<C, A_KC, B_KC, true, true>(g, splits, st);
    if (va) return launch_one<C, A_KC, B_KC, true, false>(g, splits, st);
    if (vb) return launch_one<C, A_KC, B_KC, false, true>(g, splits, st);
    return launch_one<C, A_KC, B_KC, false, false>(g, splits, st);
}

template <bool A_KC, bool B_KC>
int launch(const GemmArgs &g, int splits, bool va, bool vb, hipStream_t st)
{
    switch (pick_tile(g.M, g.N)) {
    case kWide: return launch_cfg<CfgWide, A_KC, B_KC>(g, splits, va, vb, st);
    case kK16: return launch_cfg<CfgK16, A_KC, B_KC>(g, splits, va, vb, st);
    case kWide32: return launch_cfg<CfgWide32, A_KC, B_KC>(g, splits, va, vb, st);
    case kTall: return launch_cfg<CfgTall, A_KC, B_KC>(g, splits, va, vb, st);
    case kTall32: return launch_cfg<CfgTall32, A_KC, B_KC>(g, splits, va, vb, st);
    default: return launch_cfg<CfgDefault, A_KC, B_KC>(g, splits, va, vb, st);
    }
}


template <class C, bool B_KC>
int launch_stream_cfg(const GemmArgs &g, int64_t m_tiles, int64_t gy, hipStream_t st)
{
    constexpr size_t lds = C::lds_bytes(true, B_KC);
    static bool attr_set = false;
    if (!attr_set) {
        GNNX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_stream_kernel<C, B_KC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    dim3 grid((uint32_t)(g.N / C::BN), (uint32_t)gy, 1);
    hipLaunchKernelGGL((gemm_stream_kernel<C, B_KC>), grid, dim3(C::NT), lds, st, g, m_tiles);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}

// *rows_done = number of leading rows of C written by the streaming kernel (0: shape not eligible).
int launch_stream(const GemmArgs &g, bool b_kc, int waves_per_slot, hipStream_t st, int64_t *rows_done)
{
    *rows_done = 0;
    const int tile = pick_tile(g.M, g.N);
    if (tile != kTall32 && tile != kSquare) return GNNX_OK;
    // measured at 10M rows (scripts/exp_gemm.py, same box, GNNX_GEMM_STREAM=0/1): dH.W 256-wide 10.69 -> 10.22 ms, 128-wide
    // 3.28 -> 3.15 ms; X.W^T 128-wide 3.29 -> 3.17 ms; X.W^T 256-wide 10.51 -> 10.65 ms (no gain: stays on the generic kernel)
    if (b_kc && tile == kTall32 && waves_per_slot < 3) return GNNX_OK;
    const TileDims t = tile_dims(g.M, g.N);
    if (g.K % t.bk != 0 || g.N % t.bn != 0) return GNNX_OK;
    const int64_t m_tiles = g.M / t.bm, cols = g.N / t.bn;
    const int64_t slots = (int64_t)kNumCU * (tile == kTall32 ? 1 : 2);
    if (m_tiles * cols < 4 * slots) return GNNX_OK;  // too few tiles for residency to matter
    // per-lane 32-bit offsets must cover one operand / output tile
    if ((int64_t)t.bm * g.lda >= (1ll << 30) || (int64_t)(b_kc ? t.bn : t.bk) * g.ldb >= (1ll << 30) || (int64_t)t.bm * g.ldc >= (1ll << 30))
        return GNNX_OK;
    int64_t gy = ceil_div(slots * waves_per_slot, cols);
    if (gy > m_tiles) gy = m_tiles;
    int rc;
    if (tile == kTall32) rc = b_kc ? launch_stream_cfg<CfgTall32, true>(g, m_tiles, gy, st) : launch_stream_cfg<CfgTall32, false>(g, m_tiles, gy, st);
    else rc = b_kc ? launch_stream_cfg<CfgDefault, true>(g, m_tiles, gy, st) : launch_stream_cfg<CfgDefault, false>(g, m_tiles, gy, st);
    if (rc != GNNX_OK) return rc;
    *rows_done = m_tiles * t.bm;
    return GNNX_OK;
}

}  // namespace

GNNX_API int gnnx_mfma_peak_f32(int32_t iters, int32_t n_workgroups, float *d_sink, double *flops_out, void *stream)
{
    GNNX_REQUIRE(iters > 0 && n_workgroups > 0 && d_sink, GNNX_ERR_INVALID_ARG, "bad arguments");
    hipLaunchKernelGGL(mfma_peak_kernel, dim3((uint32_t)n_workgroups), dim3(256), 0, as_stream(stream), iters, d_sink);
    GNNX_LAUNCH_CHECK();
    if (flops_out) *flops_out = (double)n_workgroups * 4 /*waves*/ * 4 /*acc*/ * (double)iters * (2.0 * 32 * 32 * 2);
    return GNNX_OK;
}

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
    const int bk = tile_dims(M, N).bk;
    int64_t ksteps = ceil_div(K > 0 ? K : 1, bk);
    g.k_per_split = ceil_div(ksteps, splits) * bk;
    if (splits > 1) {
        size_t need = sizeof(float) * (size_t)splits * (size_t)M * (size_t)N;
        GNNX_REQUIRE(d_workspace && workspace_bytes >= need, GNNX_ERR_WORKSPACE, "workspace %zu < required %zu",
                     workspace_bytes, need);
        g.slab = static_cast<float *>(d_workspace);
    }
    // tall products with whole tiles: the resident streaming kernel takes the whole M-tiles, the generic kernel the rest
    static const int stream_env = [] { const char *e = getenv("GNNX_GEMM_STREAM"); return e ? atoi(e) : 1; }();
    if (stream_env > 0 && a_kc && splits == 1 && beta == 0.f && va && vb && K > 0) {
        int64_t rows = 0;  // leading rows of C written by the streaming kernel
        const int src = launch_stream(g, b_kc, stream_env, st, &rows);
        if (src != GNNX_OK) return src;
        if (rows == M) return GNNX_OK;
        g.A += rows * lda;
        g.C += rows * ldc;
        g.M = M - rows;
    }
    int rc;
    if (a_kc && b_kc) rc = launch<true, true>(g, splits, va, vb, st);
    else if (a_kc && !b_kc) rc = launch<true, false>(g, splits, va, vb, st);
    else if (!a_kc && b_kc) rc = launch<false, true>(g, splits, va, vb, st);
    else rc = launch<false, false>(g, splits, va, vb, st);
    if (rc != GNNX_OK) return rc;
    if (splits > 1) {
        const int64_t blocks = ceil_div(M * N, 32);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, g.slab, splits, M, N, alpha, beta,
                           d_C, ldc);
        GNNX_LAUNCH_CHECK();
    }
    return GNNX_OK;
}
