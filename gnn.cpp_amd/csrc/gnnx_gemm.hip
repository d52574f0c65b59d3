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
#include <atomic>
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
    int64_t k_pairs;      // gemm_dma_tn_kernel: K / 64, dealt to the gridDim.z splits as evenly as whole K-tile pairs allow
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

// ---- LDS-DMA kernel for tall products (the bench shape: millions of rows x 256 x 256) ---------------------------------
// Resident workgroup, 256 x 256 output tile, 16 wavefronts of 64 x 64, BK = 32, the K-tile stream running across output
// tiles -- the decomposition of gemm_stream_kernel -- but the operands never pass through VGPRs: `global_load_lds_dwordx4`
// (LDS-DMA) moves 1 KiB per wave-instruction HBM/L2 -> LDS, 4 instructions per wavefront and K-tile.  LDS images (the DMA
// destination is wave-uniform base + lane * 16 B, so an image is whatever order the 64 lanes' SOURCE addresses are given in):
//   A, K-contiguous (X, dH): [256 rows][8 slots of 16 B], slot s of row r holding k-group s ^ ((r >> 1) & 7); one
//     wave-instruction = 8 rows x 128 B, every source line used whole;
//   B, k-major (W as [K][N]; for X.W^T the host transposes the small W once into the workspace): [32 k][256 n] as it lies in
//     memory, one wave-instruction per k row, rows padded to 272 words so that the two k rows a half-wave reads fall on
//     different banks.
typedef __attribute__((address_space(3))) void gemm_lds_void_t;
typedef float gemm_f32x4 __attribute__((ext_vector_type(4)));
typedef float gemm_f32x4acc __attribute__((ext_vector_type(4)));

// The loop carries (almost) NO vector-ALU instruction.  Measured on this chip (mfma_peak_kernel<mode>, 4 wavefronts per SIMD): beside the f32 MFMA every VALU instruction per MFMA
// costs 4-7 % of the matrix rate (1 / 2 / 4 per MFMA: 143 / 136 / 125 of 155 TFLOP/s), an LDS read 2-3 %, an s_nop nothing.
// The register-staged kernels carry 2-3 VALU per MFMA (staging, fragment addresses), and so did two first LDS-DMA versions
// (ds_read_b128 + element selects on 32x32x2, per-read address arithmetic on 16x16x4: loop-only 84-85 % of peak with loads
// and stores switched off) -- that, not bank conflicts or the barrier, is their 78-85 %.  Here:
//   * v_mfma_f32_16x16x4_f32, wavefront tile 64 x 64 = 4 x 4 blocks, operands SWAPPED (D = B^T-block . A^T-block), so that a
//     lane ends up with 4 consecutive output COLUMNS of one row: 16 global_store_dwordx4 per lane and tile instead of 64 dword
//     stores (a * b is commutative in IEEE arithmetic: the same fmaf chain, the same bits);
//   * a lane needs exactly ONE word per operand block and 4-k step: ds_read_b32 at a precomputed per-lane address + an
//     IMMEDIATE offset.  The k-group XOR of the image costs 8 address registers per operand (block i of 16 rows is +2048 B,
//     the stage +32768 B: immediates), computed once per launch -- no select, no address arithmetic in the loop;
//   * the K-tile loop is unrolled by two so that the LDS stage is a compile-time constant.
// Needs K % 64 == 0 (two K-tiles per trip).
template <int OFF>
__device__ __forceinline__ void lds_read_b32(float &dst, uint32_t addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset is a 16-bit unsigned immediate");
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF) : "memory");
}

// Geometry of an LDS-DMA GEMM workgroup: WM x WN wavefronts of 64 x 64 -> BM x BN output tile.  4 x 4 (1024 threads, 256 x 256)
// is the bench shape; 4 x 2 (512 threads, 256 x 128) takes 128-wide outputs (BASELINE configs[2]).  A k-major operand row of
// 256 floats is one wave-instruction; one of 128 floats is half of one, so an instruction carries two rows (k and k + 16): see the
// operand image below -- conflict-free fragment reads at both widths.
template <int WM_, int WN_, int BK_ = 32>
struct DmaGeo {
    static constexpr int WM = WM_, WN = WN_, NW = WM * WN, NT = 64 * NW;
    static constexpr int BM = 64 * WM, BN = 64 * WN, BK = BK_;
    static_assert(BK == 32 || BK == 16, "K-tile of 32 or 16");
    // K-contiguous image: [BM rows][SLOTS slots of 16 B], slot s of row r holding k-group s ^ a_xor(r).  A 16-lane fragment read takes
    // one word of 16 consecutive rows x 4 q: rows 64 / BK apart share their banks, so the XOR runs over row / (64 / BK).
    static constexpr int SLOTS = BK / 4;                              // k-groups (16 B) per row of the image
    static constexpr int A_ROW_BYTES = BK * 4, A_BLOCK_BYTES = 16 * A_ROW_BYTES;   // one row; one MFMA block of 16 rows
    static constexpr int A_ROWS_PER_INSTR = 1024 / A_ROW_BYTES;       // rows one 1-KiB wave-instruction brings (8 / 16)
    static constexpr int a_xor(int row) { return (row / (64 / BK)) % SLOTS; }
    static constexpr int A_STAGE_BYTES = BM * BK * 4;
    static constexpr int A_PER_WAVE = (BM / A_ROWS_PER_INSTR) / NW;   // 1-KiB wave-instructions per wavefront and K-tile
    // k-major operand image (B here, both operands of the TN kernel): one DMA wave-instruction = 1 KiB = one k row of 256 floats or
    // TWO k rows of 128, landing in a piece of 272 words (256 of data + 16 of padding: every instruction has its own LDS base, so
    // pieces can be padded although the bytes of one instruction cannot).  A fragment read takes word (k = 4 KG + q, column c) for
    // the four q of a wavefront at once, so consecutive q must sit 16 banks apart: with one row per piece they are consecutive
    // pieces (272 = 16 mod 64); with two rows per piece the instruction carries rows I and I + BK / 2 -- NOT two neighbours, which
    // would put q and q + 1 a multiple of 64 words apart (the 2-way conflict of the first version: 0.69 of the matrix peak at
    // N = 128) -- so that rows 4 KG .. 4 KG + 3 are again four consecutive pieces.
    static constexpr int rows_per_instr(int width) { return 256 / width; }
    static constexpr int KPIECE_BYTES = 272 * 4;
    static constexpr int kstage_bytes(int width) { return (BK / rows_per_instr(width)) * KPIECE_BYTES; }
    // byte offset of k-group KG (rows 4 KG .. 4 KG + 3; the row inside the group is q * KPIECE_BYTES in the lane's address register)
    static constexpr int kgroup_off(int width, int KG)
    {
        return rows_per_instr(width) == 1 ? 4 * KG * KPIECE_BYTES : (4 * (KG % (BK / 8))) * KPIECE_BYTES + (KG / (BK / 8)) * 512;
    }
    // k row (inside the K-tile) that lane `lane` of wave-instruction I brings
    static constexpr int krow_of(int width, int I, int lane) { return rows_per_instr(width) == 1 ? I : I + (BK / 2) * (lane / 32); }
    static constexpr int B_STAGE_BYTES = kstage_bytes(BN);
    static constexpr int B_PIECE_BYTES = KPIECE_BYTES;                 // LDS bytes one B wave-instruction covers (>= 1024)
    static constexpr int B_PER_WAVE = (BK / rows_per_instr(BN)) / NW;
    static constexpr int STAGES_BYTES = 2 * A_STAGE_BYTES + 2 * B_STAGE_BYTES;
    // the epilogue stages a block row of C (16 rows x 64 columns) in FOUR 1-KiB pieces private to the wavefront: its own DMA pieces
    // of the idle stage and, where a K-tile gives a wavefront fewer than four, pieces of its own behind the stages
    static constexpr int EP_EXTRA = A_PER_WAVE + B_PER_WAVE >= 4 ? 0 : 4 - A_PER_WAVE - B_PER_WAVE;
    static constexpr int LDS_BYTES = STAGES_BYTES + NW * EP_EXTRA * 1024;
    static_assert(A_PER_WAVE >= 1 && B_PER_WAVE >= 1, "too many wavefronts for the operand tile");
};

template <class GEO, int STAGE_ID, int KG>
__device__ __forceinline__ void dma2_read_group(float (&a)[4], float (&b)[4], const uint32_t (&ak)[GEO::SLOTS], uint32_t bk)
{
    constexpr int SA = STAGE_ID * GEO::A_STAGE_BYTES, BLK = GEO::A_BLOCK_BYTES;
    lds_read_b32<SA + 0 * BLK>(a[0], ak[KG]);
    lds_read_b32<SA + 1 * BLK>(a[1], ak[KG]);
    lds_read_b32<SA + 2 * BLK>(a[2], ak[KG]);
    lds_read_b32<SA + 3 * BLK>(a[3], ak[KG]);
    constexpr int SB = STAGE_ID * GEO::B_STAGE_BYTES + GEO::kgroup_off(GEO::BN, KG);  // k row 4 KG + q (q is in the address register)
    lds_read_b32<SB + 0 * 64>(b[0], bk);
    lds_read_b32<SB + 1 * 64>(b[1], bk);
    lds_read_b32<SB + 2 * 64>(b[2], bk);
    lds_read_b32<SB + 3 * 64>(b[3], bk);
}

#define GNNX_DMA2_WAIT(N, set)                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(a[set][0]), "+v"(a[set][1]), "+v"(a[set][2]), "+v"(a[set][3]), "+v"(b[set][0]), \
                 "+v"(b[set][1]), "+v"(b[set][2]), "+v"(b[set][3])::"memory")
#define GNNX_DMA2_MFMA(set)                                                                                \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) _Pragma("unroll") for (int j_ = 0; j_ < 4; j_++)      \
        acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[set][j_], a[set][i_], acc[i_][j_], 0, 0, 0)
// one k-group: fragments of group KG + 1 on their way while group KG is multiplied
#define GNNX_DMA2_STEP(ST, KG, cur, nxt)                                  \
    dma2_read_group<GEO, ST, KG + 1>(a[nxt], b[nxt], ak, bk);            \
    GNNX_DMA2_WAIT(8, cur);                                               \
    GNNX_DMA2_MFMA(cur)
#define GNNX_DMA2_KTILE(ST)                                               \
    dma2_read_group<GEO, ST, 0>(a[0], b[0], ak, bk);                     \
    GNNX_DMA2_STEP(ST, 0, 0, 1);                                          \
    GNNX_DMA2_STEP(ST, 1, 1, 0);                                          \
    GNNX_DMA2_STEP(ST, 2, 0, 1);                                          \
    GNNX_DMA2_STEP(ST, 3, 1, 0);                                          \
    GNNX_DMA2_STEP(ST, 4, 0, 1);                                          \
    GNNX_DMA2_STEP(ST, 5, 1, 0);                                          \
    GNNX_DMA2_STEP(ST, 6, 0, 1);                                          \
    GNNX_DMA2_WAIT(0, 1);                                                 \
    GNNX_DMA2_MFMA(1)
#define GNNX_DMA2_KTILE16(ST)                                             \
    dma2_read_group<GEO, ST, 0>(a[0], b[0], ak, bk);                     \
    GNNX_DMA2_STEP(ST, 0, 0, 1);                                          \
    GNNX_DMA2_STEP(ST, 1, 1, 0);                                          \
    GNNX_DMA2_STEP(ST, 2, 0, 1);                                          \
    GNNX_DMA2_WAIT(0, 1);                                                 \
    GNNX_DMA2_MFMA(1)

// LDS-DMA with a wave-uniform 64-bit base in SGPRs + a per-lane 32-bit BYTE offset (the saddr form: no 64-bit vector address
// arithmetic, no zero-extended offset pairs to keep alive).  M0 = LDS destination of lane 0; it is compiler-reserved, so it is
// saved and restored inside the statement.  The leading s_nop covers a base freshly written by the scalar ALU.
__device__ __forceinline__ void dma_16B(uint32_t voff_bytes, const float *sbase, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff_bytes), "s"(sbase), "s"(lds_dst) : "memory");
}

#ifdef GNNX_EXPERIMENTS
#define GNNX_ABLATE(bit) ((ablate & (bit)) != 0)  // timing only -- 1: no epilogue, 2: no operand loads after the first, 4: the epilogue's LDS round trip without its stores, 8: stores straight from the accumulators (16 rows x 64 B per instruction), 16: FUSE 4's list walk without its send stores, 32: non-temporal C stores, 64: non-temporal send stores
#else
#define GNNX_ABLATE(bit) false
#endif
// FUSE: the epilogue of the stacked layers' backward GEMM (SURVEY.md 2b "fused ops"): C = (A . B) masked
// by the ReLU of the layer below (C[m][n] = 0 where Ymask[m][n] <= 0: Mask::_backward, reference operation.h:557-562) and
// colsum_partial[blockIdx.y][n] = sum over this workgroup's rows of the masked C -- the bias gradient of the layer below
// (Add::_backward -> sum_to_size, operation.h:114-128) -- so neither the unmasked gradient nor a separate column-sum pass
// touches HBM.  The mask values are loaded with the same full-line pattern the stores use, under the LDS round trip.
// FUSE == 2 (opt-in, gnnx_gemm_bn_stats_f32): the batch statistics of BatchNorm over the columns of C = X . W^T in the same
// pass -- per-lane sums of d = c - shift[n] and d^2 (shift = row 0 of C: a sample value, so the single-pass variance
// Q/M - (S/M)^2 cancels mildly), reduced per workgroup into [gridDim.y][2][N] partials and finished in double.  C itself is
// stored unchanged.  Not the reference's two-pass arithmetic (nn.cpp:303,312): tolerance-level, next to the exact gnnx_bn_stats_f32.
// FUSE == 4 (gnnx_gemm_nt_rows_to_slots_f32, the sharded step's transform): the halo PACK rides in the epilogue.  slots[row][8] lists
// the send-buffer rows that row `row` of C goes to (-1 padded, packed at the front: gnnx_rows_to_slots_f32's table); every store
// instruction of the epilogue (4 rows x 256 B, a quarter-wave per row) is repeated for each listed slot of its rows, from the same
// registers -- the rows never come back from HBM to be packed.  A lane loads ONE dword of its row's eight (lane & 7) in front of the
// LDS round trip and the quarter-wave shares them through ds_bpermute.  C itself is stored unchanged.
struct GemmFuse {
    const float *ymask;      // FUSE 1: [M][N] forward output of the layer below (ld ldy), mask = ymask > 0; FUSE 2: shift[N]
    int64_t ldy;             // FUSE 4: ld of `send` (elements)
    float *colsum_partial;   // FUSE 1: [gridDim.y][N]; FUSE 2: [gridDim.y][2][N]; FUSE 4: the send buffer
    const int32_t *slots;    // FUSE 4: [M][8]
};

// NG (N guard): the last column tile is narrower than BN (N % 4 == 0): lanes whose 16 bytes lie past column N neither load B
// (their LDS slots keep whatever they held: the products of those columns are garbage and stay in registers) nor store C nor
// contribute column sums.
// BK_ = 16: half the K-tile, so that TWO workgroups are resident per CU (4 wavefronts per SIMD as before): the C-store epilogue and
// the K-tile barriers of one run under the MFMAs of the other.
// DEFER (EXPERIMENTS build only; plain products on the 256 x 128 tile, K >= 128): the C tile does not leave through an epilogue.  Eight
// wavefronts per CU leave every wavefront 256 VGPRs, twice what the tile needs, so the finished accumulators are parked in a second
// set of 64 registers and stored STRAIGHT from there -- 16 rows x 64 bytes per instruction, one block row in front of each of the next
// tile's first four K-tiles -- while the matrix pipe works on that tile.  Same bits; measured slower than the LDS-staged epilogue
// (launch_dma_geo), kept as the record of that experiment.
template <int WM_, int WN_, int FUSE, bool NG, int BK_ = 32, bool DEFER = false>
__global__ __launch_bounds__(64 * WM_ * WN_, BK_ == 16 ? 4 : 1) void gemm_dma_kernel(GemmArgs g, int64_t m_tiles, int ablate, GemmFuse fu)
{
    static_assert(!DEFER || (FUSE == 0 && WM_ * WN_ <= 8 && BK_ == 32), "deferred stores: plain product, at most two wavefronts per SIMD");
    (void)ablate;
    using GEO = DmaGeo<WM_, WN_, BK_>;
    constexpr int BM = GEO::BM, BN = GEO::BN, BK = GEO::BK, NW = GEO::NW;
    constexpr int APW = GEO::A_PER_WAVE, BPW = GEO::B_PER_WAVE;
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];  // [2][A stage], then [2][B stage]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / GEO::WN, wn = wave % GEO::WN;
    const int q = lane >> 4, r16 = lane & 15;
    const int64_t n0 = (int64_t)blockIdx.x * BN;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(gemm_lds_void_t *)lds_raw;
    constexpr uint32_t B0 = 2 * GEO::A_STAGE_BYTES;   // byte offset of the B stages

    // ---- DMA: per-lane BYTE offsets inside an operand tile; wave-instruction (wave + NW u) of the K-tile
    uint32_t offa[APW], offb[BPW], pa[APW], pb[BPW];
#pragma unroll
    for (int u = 0; u < APW; u++) {
        const int row = GEO::A_ROWS_PER_INSTR * (wave + NW * u) + lane / GEO::SLOTS;
        const int kg = (lane % GEO::SLOTS) ^ GEO::a_xor(row);
        offa[u] = (uint32_t)(row * g.lda + 4 * kg) * 4u;
        pa[u] = lds0 + (uint32_t)((wave + NW * u) * 1024);
    }
#pragma unroll
    for (int u = 0; u < BPW; u++) {
        constexpr int RPI = GEO::rows_per_instr(BN), LPR = 64 / RPI;   // k rows per instruction, lanes per row
        const int krow = GEO::krow_of(BN, wave + NW * u, lane);
        offb[u] = (uint32_t)(krow * g.ldb + 4 * (lane % LPR)) * 4u;
        pb[u] = lds0 + B0 + (uint32_t)((wave + NW * u) * GEO::B_PIECE_BYTES);
    }
    const float *bcol = g.B + n0;
    const int64_t n_left = g.N - n0;   // columns of this tile that exist
    const bool okb = !NG || 4 * (lane % (64 / GEO::rows_per_instr(BN))) + 4 <= n_left;   // this lane's 4 B columns
    const bool okc = !NG || wn * 64 + 4 * (lane & 15) + 4 <= n_left;                      // this lane's 4 C columns
    // first row of M-tile mt.  The host may ask for ceil(M / BM) tiles of a plain product: the last one then starts at M - BM and
    // overlaps its neighbour -- the shared rows are computed twice, to the same bits, and stored twice -- so that no ragged rows are
    // left for another kernel (M >= BM).  With floor(M / BM) tiles (the column-sum epilogues) the clamp never acts.
    const int64_t last_row0 = g.M - BM;
    auto row0_of = [&](int64_t mt) { return mt * BM <= last_row0 ? mt * BM : last_row0; };
    auto issue = [&](int stage, int64_t mt, int64_t k0) {
        const float *abase = g.A + row0_of(mt) * g.lda + k0;
        const float *bbase = bcol + k0 * g.ldb;
#pragma unroll
        for (int u = 0; u < (APW > BPW ? APW : BPW); u++) {
            if (u < APW) dma_16B(offa[u < APW ? u : 0], abase, pa[u < APW ? u : 0] + stage * GEO::A_STAGE_BYTES);
            if (u < BPW && okb) dma_16B(offb[u < BPW ? u : 0], bbase, pb[u < BPW ? u : 0] + stage * GEO::B_STAGE_BYTES);
        }
    };
    // ---- fragment address registers (bytes): the k-groups of this lane's row of block 0 (the row's slot XOR is the same in
    // every 16-row block: 16 rows are a whole number of XOR periods)
    uint32_t ak[GEO::SLOTS];
    {
        const int row = wm * 64 + r16;
        const uint32_t xa = (uint32_t)(GEO::a_xor(row) << 4);
#pragma unroll
        for (int kg = 0; kg < GEO::SLOTS; kg++) ak[kg] = lds0 + (uint32_t)(row * GEO::A_ROW_BYTES + 4 * q) + (((uint32_t)kg << 4) ^ xa);
    }
    const uint32_t bk = lds0 + B0 + (uint32_t)(q * GEO::KPIECE_BYTES + (wn * 64 + r16) * 4);
    // ---- epilogue through LDS.  D = mfma(b, a): a lane holds C[16 i + r16][16 j + 4 q .. + 3] of its wavefront's 64 x 64
    // block.  Block row i (16 rows x 64 columns = 4 KB) is written to four of the wavefront's OWN DMA pieces of stage 1 (free
    // after the tile's last K-tile; only this wavefront's next DMA overwrites them, and that is issued behind its reads), piece
    // p = rows 4p..4p+3, and read back one piece per instruction: 64 lanes x 16 B = 4 rows x 256 B = whole 128-byte lines per
    // store.  16-byte columns are XORed with (row & 3) to spread the writers over the banks.
    uint32_t ep[4];   // stage-1 byte address of private piece p
#pragma unroll
    for (int p = 0; p < 4; p++) {
        if (p < APW) ep[p] = pa[p < APW ? p : 0] + GEO::A_STAGE_BYTES;
        else if (p - APW < BPW) ep[p] = pb[p >= APW && p - APW < BPW ? p - APW : 0] + GEO::B_STAGE_BYTES;
        else ep[p] = lds0 + (uint32_t)(GEO::STAGES_BYTES + (wave * GEO::EP_EXTRA + (p - APW - BPW)) * 1024);
    }
    const int pw = r16 >> 2;
    const uint32_t ep_wr = (pw == 0 ? ep[0] : pw == 1 ? ep[1] : pw == 2 ? ep[2] : ep[3]) + (uint32_t)((r16 & 3) * 256) +
                           (uint32_t)((q ^ (r16 & 3)) << 4);   // + 64 j
    // reader: lane L reads 16 B at byte L*16 of a piece = row L/16, 16-B column L%16 (stored at column ^ (row & 3))
    const uint32_t ep_rd = (uint32_t)((lane >> 4) * 256) + (uint32_t)(((lane & 15) ^ (lane >> 4)) << 4);
    constexpr int ES = FUSE == 3 ? 2 : 4;   // bytes per stored element (FUSE 3: C is bf16, ldc in bf16 elements)
    const uint32_t offc = (uint32_t)((wm * 64 + (lane >> 4)) * g.ldc + wn * 64 + 4 * (lane & 15)) * (uint32_t)ES;  // bytes: row L/16, col 4 (L%16)

    gemm_f32x4acc acc[4][4];
    gemm_f32x4acc pend[DEFER ? 4 : 1][DEFER ? 4 : 1];   // DEFER: the previous tile's accumulators (x alpha), on their way out
    char *pend_tile = nullptr;                          // ... and where they go (nullptr: nothing pending)
    const uint32_t offd = (uint32_t)((wm * 64 + r16) * g.ldc + wn * 64 + 4 * q) * 4u;   // direct store: row r16, columns 4 q .. of block (i, j)
    auto flush_row = [&](int i) {   // block row i of the pending tile: 4 stores of 16 rows x 64 bytes
        if constexpr (DEFER) {
            if (pend_tile) {
                char *crow = pend_tile + (int64_t)(16 * i) * g.ldc * 4;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (!NG || wn * 64 + 16 * j + 4 * q + 4 <= n_left) *reinterpret_cast<gemm_f32x4acc *>(crow + offd + 64 * j) = pend[i][j];
            }
        }
    };
    float a[2][4], b[2][4];
    gemm_f32x4acc csum = {0.f, 0.f, 0.f, 0.f};   // FUSE: column sums of this lane's 4 columns over every row it stores
    gemm_f32x4acc csq = {0.f, 0.f, 0.f, 0.f};    // FUSE 2: sums of squares (of the shifted values)
    gemm_f32x4acc shift4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (FUSE == 2)
        if (okc) shift4 = *reinterpret_cast<const gemm_f32x4acc *>(fu.ymask + n0 + wn * 64 + 4 * (lane & 15));
    const uint32_t offy = FUSE == 1 ? (uint32_t)((wm * 64 + (lane >> 4)) * fu.ldy + wn * 64 + 4 * (lane & 15)) * 4u : 0u;
    // FUSE 4: the tile's slot lists (BM rows x 32 B) come into LDS by DMA with the tile's first K-tile -- two buffers behind the
    // operand stages, alternating per tile: a wavefront that is already in the next tile must not overwrite lists its neighbours
    // still read (they are at most one barrier apart).  The epilogue must not LOAD from global memory: vmcnt counts loads and stores
    // in order, so waiting for a load issued behind the stores of the rows before it is waiting for those stores to be acknowledged
    // (the first version did: + 0.26 ms on a 1.24 ms product).
    constexpr uint32_t SLOT_BUF = (uint32_t)BM * 32u;
    const uint32_t slot_lds = lds0 + (uint32_t)GEO::LDS_BYTES;
    const uint32_t offs = (uint32_t)((wm * 64 + (lane >> 4)) * 8 + (lane & 7)) * 4u;   // bytes into the tile's slot lists
    char *send_col = FUSE == 4 ? reinterpret_cast<char *>(fu.colsum_partial + n0 + wn * 64 + 4 * (lane & 15)) : nullptr;
    uint32_t sbuf = 0;
    int64_t mt = blockIdx.y;
    if (mt < m_tiles) issue(0, mt, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (; mt < m_tiles; mt += gridDim.y) {
        const int64_t mt_next = mt + gridDim.y;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[i][j][r] = 0.f;
        if constexpr (FUSE == 4) {   // one 1-KiB wave-instruction per 32 rows; lands before the barrier of the first K-tile
            sbuf ^= SLOT_BUF;
            if (wave < BM / 32)
                dma_16B((uint32_t)lane * 16u, reinterpret_cast<const float *>(fu.slots + row0_of(mt) * 8 + wave * 256), slot_lds + sbuf + (uint32_t)wave * 1024u);
        }
        for (int64_t k0 = 0; k0 < g.K; k0 += 2 * BK) {
            const bool last = k0 + 2 * BK >= g.K;
            if (!GNNX_ABLATE(2)) issue(1, mt, k0 + BK);   // K % 64 == 0: the odd K-tile of this trip always exists
            if constexpr (DEFER) {   // the pending tile's block rows 0 / 2 (first two trips: K >= 128) leave under this K-tile's MFMAs
                if (k0 == 0) flush_row(0);
                else if (k0 == 2 * BK) flush_row(2);
            }
            if constexpr (BK == 32) { GNNX_DMA2_KTILE(0); } else { GNNX_DMA2_KTILE16(0); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if ((!last || mt_next < m_tiles) && !GNNX_ABLATE(2)) issue(0, last ? mt_next : mt, last ? 0 : k0 + 2 * BK);
            if constexpr (DEFER) {
                if (k0 == 0) flush_row(1);
                else if (k0 == 2 * BK) flush_row(3);
            }
            if constexpr (BK == 32) { GNNX_DMA2_KTILE(1); } else { GNNX_DMA2_KTILE16(1); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (GNNX_ABLATE(1)) {
            float s_ = 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) s_ += acc[i][j][r];
            if (s_ == 1.2345e30f) g.C[0] = s_;
            continue;
        }
        const float alpha = g.alpha;
        char *ctile = reinterpret_cast<char *>(g.C) + (row0_of(mt) * g.ldc + n0) * ES;
        if constexpr (DEFER) {   // park the tile (every block row of the tile before it left during this tile's first four K-tiles)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) pend[i][j] = alpha != 1.0f ? acc[i][j] * alpha : acc[i][j];
            pend_tile = ctile;
            continue;
        }
        if constexpr (FUSE == 0 && !NG) {
            if (GNNX_ABLATE(8)) {   // A/B: straight from the accumulators, 16 rows x 64 B per store instruction, no LDS
                const uint32_t offd = (uint32_t)((wm * 64 + r16) * g.ldc + wn * 64 + 4 * q) * 4u;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    char *crow = ctile + (int64_t)(16 * i) * g.ldc * 4;
#pragma unroll
                    for (int j = 0; j < 4; j++) *reinterpret_cast<gemm_f32x4acc *>(crow + offd + 64 * j) = acc[i][j] * alpha;
                }
                continue;
            }
        }
        const char *ytile = FUSE == 1 ? reinterpret_cast<const char *>(fu.ymask + row0_of(mt) * fu.ldy + n0) : nullptr;
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                gemm_f32x4acc v = acc[i][j];
                if (alpha != 1.0f) v = v * alpha;
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(ep_wr), "v"(v), "i"(64 * j) : "memory");
            }
            gemm_f32x4acc ym[4] = {};
            if constexpr (FUSE == 1) {   // issued behind the LDS writes (block row i's accumulators are dead by now: their registers
                                    // take the mask values) and in flight during the LDS round trip below
                const char *yrow = ytile + (int64_t)(16 * i) * fu.ldy * 4;
#pragma unroll
                for (int p = 0; p < 4; p++)
                    if (okc) ym[p] = *reinterpret_cast<const gemm_f32x4acc *>(yrow + (int64_t)(4 * p) * fu.ldy * 4 + offy);
            }
            int32_t sl[4] = {-1, -1, -1, -1};
            if constexpr (FUSE == 4) {   // dword (lane & 7) of the slot list of row 16 i + 4 p + lane / 16, from the tile's lists in LDS
#pragma unroll
                for (int p = 0; p < 4; p++)
                    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(sl[p]) : "v"(slot_lds + sbuf + offs), "i"((16 * i + 4 * p) * 32) : "memory");
            }
            if constexpr (FUSE == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sl[0]), "+v"(sl[1]), "+v"(sl[2]), "+v"(sl[3])::"memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            gemm_f32x4acc o[4];
#pragma unroll
            for (int p = 0; p < 4; p++) asm volatile("ds_read_b128 %0, %1" : "=v"(o[p]) : "v"(ep_rd + ep[p]) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3])::"memory");
            char *crow = ctile + (int64_t)(16 * i) * g.ldc * ES;   // wave-uniform
#pragma unroll
            for (int p = 0; p < 4; p++) {
                if constexpr (FUSE == 1) {
#pragma unroll
                    for (int c = 0; c < 4; c++) o[p][c] = ym[p][c] > 0.f ? o[p][c] : 0.f;
                    csum += o[p];
                }
                if constexpr (FUSE == 2) {
                    const gemm_f32x4acc d = o[p] - shift4;
                    csum += d;
                    csq += d * d;
                }
                if constexpr (FUSE == 3) {   // round to nearest even, as gnnx_f32_to_bf16: 8 bytes per lane, 128-byte row segments
                    union { __bf16 h[4]; uint2 u; } ob;
                    ob.h[0] = (__bf16)o[p][0];
                    ob.h[1] = (__bf16)o[p][1];
                    ob.h[2] = (__bf16)o[p][2];
                    ob.h[3] = (__bf16)o[p][3];
                    if (okc) *reinterpret_cast<uint2 *>(crow + (int64_t)(4 * p) * g.ldc * ES + offc) = ob.u;
                } else {
                    if (GNNX_ABLATE(4)) {   // A/B: the LDS round trip without the stores
                        asm volatile("" ::"v"(o[p]));
                    } else if (GNNX_ABLATE(32)) {   // A/B: C leaves with the non-temporal policy
                        if (okc) __builtin_nontemporal_store(o[p], reinterpret_cast<gemm_f32x4acc *>(crow + (int64_t)(4 * p) * g.ldc * ES + offc));
                    } else if (okc) *reinterpret_cast<gemm_f32x4acc *>(crow + (int64_t)(4 * p) * g.ldc * ES + offc) = o[p];
                }
                if constexpr (FUSE == 4) {
                    // the same 4 rows x 256 B once more per listed slot.  Lists are packed at the front, so the number of rounds the
                    // wavefront needs is the longest list of its four rows: one ballot, no data-dependent exit
                    const uint64_t has = __ballot(sl[p] >= 0);
                    const int rounds = __builtin_popcountll((has | (has >> 16) | (has >> 32) | (has >> 48)) & 0xffull);
#pragma unroll 1
                    for (int k = 0; k < rounds; k += 2) {   // two lists entries per trip: the second bpermute is in flight under the first store
                        const int32_t s0 = __builtin_amdgcn_ds_bpermute(((lane & 48) + k) << 2, sl[p]);
                        const int32_t s1 = __builtin_amdgcn_ds_bpermute(((lane & 48) + k + 1) << 2, sl[p]);   // k + 1 <= 7
                        if (GNNX_ABLATE(16)) {   // A/B: the list walk without its stores
                            asm volatile("" ::"v"(s0), "v"(s1));
                            continue;
                        }
                        if (GNNX_ABLATE(64)) {   // A/B: the send rows leave with the non-temporal policy (nothing on this GPU reads them again)
                            if (s0 >= 0 && okc) __builtin_nontemporal_store(o[p], reinterpret_cast<gemm_f32x4acc *>(send_col + (int64_t)s0 * fu.ldy * 4));
                            if (s1 >= 0 && okc) __builtin_nontemporal_store(o[p], reinterpret_cast<gemm_f32x4acc *>(send_col + (int64_t)s1 * fu.ldy * 4));
                            continue;
                        }
                        if (s0 >= 0 && okc) *reinterpret_cast<gemm_f32x4acc *>(send_col + (int64_t)s0 * fu.ldy * 4) = o[p];
                        if (s1 >= 0 && okc) *reinterpret_cast<gemm_f32x4acc *>(send_col + (int64_t)s1 * fu.ldy * 4) = o[p];
                    }
                }
            }
        }
    }
    if constexpr (DEFER) {   // the last tile of this workgroup
        flush_row(0);
        flush_row(1);
        flush_row(2);
        flush_row(3);
    }
    if constexpr (FUSE == 1 || FUSE == 2) {
        // column sums of the workgroup: lanes with equal (lane & 15) in the WM wavefronts of a column group wn hold the same 4
        // columns.  Everything is parked in LDS (the operand stages are dead: every DMA was waited for, the last K-tile
        // ended with a barrier) and added in a fixed order: wm 0..WM-1, lane group 0..3.
        gemm_f32x4acc *red = reinterpret_cast<gemm_f32x4acc *>(lds_raw);   // [wave][lane]
        __syncthreads();
        red[wave * 64 + lane] = csum;
        __syncthreads();
        if (tid < BN / 4 && (!NG || 4 * tid + 4 <= n_left)) {   // thread t: columns 4t..4t+3 of the tile = column group t / 16, 16-byte column t % 16
            const int cg = tid >> 4, c16 = tid & 15;
            gemm_f32x4acc sum = {0.f, 0.f, 0.f, 0.f};
            for (int w = 0; w < GEO::WM; w++)
                for (int lg = 0; lg < 4; lg++) sum += red[(w * GEO::WN + cg) * 64 + lg * 16 + c16];
            float *prow = fu.colsum_partial + (int64_t)blockIdx.y * (FUSE == 2 ? 2 : 1) * g.N;
            *reinterpret_cast<gemm_f32x4acc *>(prow + n0 + 4 * tid) = sum;
        }
        if constexpr (FUSE == 2) {   // the same for the squares
            __syncthreads();
            red[wave * 64 + lane] = csq;
            __syncthreads();
            if (tid < BN / 4 && (!NG || 4 * tid + 4 <= n_left)) {
                const int cg = tid >> 4, c16 = tid & 15;
                gemm_f32x4acc sum = {0.f, 0.f, 0.f, 0.f};
                for (int w = 0; w < GEO::WM; w++)
                    for (int lg = 0; lg < 4; lg++) sum += red[(w * GEO::WN + cg) * 64 + lg * 16 + c16];
                *reinterpret_cast<gemm_f32x4acc *>(fu.colsum_partial + ((int64_t)blockIdx.y * 2 + 1) * g.N + n0 + 4 * tid) = sum;
            }
        }
    }
}

#ifdef GNNX_EXPERIMENTS
// (EXPERIMENTS build only, GNNX_GEMM_W128=1: a record of what does NOT lift the 128-wide products -- DESIGN.md section 4.2.  Same bits
// as gemm_dma_kernel<4, 2>; 10 M x 128 x 128: 2.83 ms against 2.82.  With its stores aimed at L2-resident rows 2.54, with its loads
// 2.52, with both 2.45 = 0.85 of the matrix peak: the loop's nine LDS reads per eight MFMAs cost what the barriers cost the
// 256 x 128 tile.  A first version -- strips of 32 rows, ten reads per sixteen MFMAs, two stages -- ran 2.70 against 2.76.)
// ---- 128-wide products without a barrier: C[M][128] = A[M][128] . B[128][128] (K = N = 128: BASELINE configs[2]) ----------------------
// At this width bytes and flops are nearly balanced and a K-tile of gemm_dma_kernel<4, 2> lasts only 3.4 us: its one-K-tile prefetch
// distance no longer covers an HBM round trip once the C stores share the memory pipeline (loop alone 0.90 of the matrix peak, with
// loads 0.89, with loads AND stores 0.74), and `s_waitcnt vmcnt(0)` behind a tile's first K-tile waits for the previous tile's stores
// to be acknowledged, because vmcnt counts loads and stores in one order.  Here
//   * W (64 KB) is loaded into LDS once and stays (k-major operand image, 68 KB); a wavefront owns a STRIP of 16 rows x all 128
//     columns and brings its own operand rows into its own ring of four 2-KB stages: after the barrier behind the load of W no
//     wavefront waits for another, and the two wavefronts of a SIMD drift apart (one's epilogue under the other's MFMAs);
//   * the ring is THREE K-tiles ahead of the multiply (stage = K-tile index: a strip is four K-tiles), and every wait is counted:
//     `vmcnt(12)` / `vmcnt(4)` leave the two younger K-tiles (2 DMA instructions each) AND the last strip's 8 stores in flight -- no
//     wait ever waits for a store (first strip of a wavefront: no stores yet, vmcnt(4));
//   * same MFMA chain per output element as gemm_dma_kernel (D = mfma(b, a) over k = 0 .. 127 in steps of 4): same bits;
//   * per k-group 1 a-fragment + 8 b-fragments (one ds_read_b32 each, precomputed addresses + immediates) feed 8 MFMAs;
//   * C leaves through a 2-KB staging area of the wavefront's own, 16 rows x 32 columns at a time: a store instruction covers
//     8 rows x 128 B (whole lines);
//   * loads are never predicated: past the last strip the ring re-reads the last strip (so that the counts above stay exact), and
//     the last strip starts at M - 16, overlapping its neighbour (same values stored twice).
struct W128 {
    static constexpr int NW = 8, NT = 64 * NW, ROWS = 16;                    // wavefronts per workgroup, rows per strip
    using G = DmaGeo<4, 2, 32>;                                              // (the operand images of the 256 x 128 tile: BK = 32, BN = 128)
    static constexpr int B_TILE_BYTES = G::B_STAGE_BYTES;                    // one K-tile of W: 16 pieces of 272 words
    static constexpr int B_BYTES = 4 * B_TILE_BYTES;                         // K = 128
    static constexpr int A_STAGE_BYTES = ROWS * 32 * 4;                      // 2 KB
    static constexpr int WAVE_BYTES = 4 * A_STAGE_BYTES + 2048;              // ring of four stages + C staging
    static constexpr int LDS_BYTES = B_BYTES + NW * WAVE_BYTES;              // 68 KB + 80 KB
};

template <int KT, int KG>
__device__ __forceinline__ void w128_read_group(float &a, float (&b)[8], const uint32_t (&ak)[8], uint32_t bk_lo, uint32_t bk_hi)
{
    lds_read_b32<KT * W128::A_STAGE_BYTES>(a, ak[KG]);
    // K-tiles 0, 1 through bk_lo, 2, 3 through bk_hi (= bk_lo + 2 K-tiles): the ds_read offset is a 16-bit immediate
    constexpr int SB = (KT & 1) * W128::B_TILE_BYTES + W128::G::kgroup_off(128, KG);
    const uint32_t bk = KT < 2 ? bk_lo : bk_hi;
    lds_read_b32<SB + 0 * 64>(b[0], bk);
    lds_read_b32<SB + 1 * 64>(b[1], bk);
    lds_read_b32<SB + 2 * 64>(b[2], bk);
    lds_read_b32<SB + 3 * 64>(b[3], bk);
    lds_read_b32<SB + 4 * 64>(b[4], bk);
    lds_read_b32<SB + 5 * 64>(b[5], bk);
    lds_read_b32<SB + 6 * 64>(b[6], bk);
    lds_read_b32<SB + 7 * 64>(b[7], bk);
}

#define GNNX_W128_WAIT(N, set)                                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(a[set]), "+v"(b[set][0]), "+v"(b[set][1]), "+v"(b[set][2]), "+v"(b[set][3]),      \
                 "+v"(b[set][4]), "+v"(b[set][5]), "+v"(b[set][6]), "+v"(b[set][7])::"memory")
#define GNNX_W128_MFMA(set) \
    _Pragma("unroll") for (int j_ = 0; j_ < 8; j_++) acc[j_] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[set][j_], a[set], acc[j_], 0, 0, 0)
#define GNNX_W128_STEP(KT, KG, cur, nxt)                                           \
    w128_read_group<KT, KG + 1>(a[nxt], b[nxt], ak, bk_lo, bk_hi);                 \
    GNNX_W128_WAIT(9, cur);                                                        \
    GNNX_W128_MFMA(cur)
#define GNNX_W128_KTILE(KT)                                                        \
    w128_read_group<KT, 0>(a[0], b[0], ak, bk_lo, bk_hi);                          \
    GNNX_W128_STEP(KT, 0, 0, 1);                                                   \
    GNNX_W128_STEP(KT, 1, 1, 0);                                                   \
    GNNX_W128_STEP(KT, 2, 0, 1);                                                   \
    GNNX_W128_STEP(KT, 3, 1, 0);                                                   \
    GNNX_W128_STEP(KT, 4, 0, 1);                                                   \
    GNNX_W128_STEP(KT, 5, 1, 0);                                                   \
    GNNX_W128_STEP(KT, 6, 0, 1);                                                   \
    GNNX_W128_WAIT(0, 1);                                                          \
    GNNX_W128_MFMA(1)

__global__ __launch_bounds__(W128::NT, 1) void gemm_w128_kernel(GemmArgs g, int64_t n_strips, int ablate)
{
    (void)ablate;
    using G = W128::G;
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];   // [W: 4 K-tiles][per wavefront: ring of 4 stages, C staging]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r16 = lane & 15;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(gemm_lds_void_t *)lds_raw;
    const uint32_t a0 = lds0 + (uint32_t)W128::B_BYTES + (uint32_t)wave * (uint32_t)W128::WAVE_BYTES;   // this wavefront's stage 0

    // ---- W into LDS, once: K-tile t = 16 wave-instructions (two k rows of 128 floats each: rows I and I + 16 of the K-tile)
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int inst = wave + W128::NW * u;   // 0 .. 63: K-tile inst / 16, instruction inst % 16
        const int t = inst >> 4, I = inst & 15;
        const int krow = 32 * t + G::krow_of(128, I, lane);
        dma_16B((uint32_t)(krow * g.ldb + 4 * (lane % 32)) * 4u, g.B, lds0 + (uint32_t)(t * W128::B_TILE_BYTES + I * G::B_PIECE_BYTES));
    }
    // ---- this wavefront's operand DMA: instruction u of a K-tile brings rows 8 u .. 8 u + 7 (128 bytes each) of the strip
    uint32_t offa[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int row = 8 * u + lane / 8;
        const int kg = (lane % 8) ^ G::a_xor(row);
        offa[u] = (uint32_t)(row * g.lda + 4 * kg) * 4u;
    }
    const int64_t last_row0 = g.M - W128::ROWS;
    const int64_t last_strip = n_strips - 1;
    auto row0_of = [&](int64_t s) { return s * W128::ROWS <= last_row0 ? s * W128::ROWS : last_row0; };
    auto issue = [&](int kt, int64_t strip) {   // K-tile kt of `strip` (clamped to the last strip: never predicated) into stage kt
        const float *abase = g.A + row0_of(GNNX_ABLATE(2) ? 0 : (strip < last_strip ? strip : last_strip)) * g.lda + 32 * kt;   // A/B 2: every strip re-reads strip 0 (L2 hits)
        if (GNNX_ABLATE(4)) {   // A/B 4 (timing only, wrong products): the same bytes as ONE contiguous 2-KB run per K-tile instead of 16 row slices of 128 B
            const float *cbase = g.A + row0_of(strip < last_strip ? strip : last_strip) * g.lda + 512 * kt;
            dma_16B((uint32_t)lane * 16u, cbase, a0 + (uint32_t)(kt * W128::A_STAGE_BYTES));
            dma_16B((uint32_t)lane * 16u + 1024u, cbase, a0 + (uint32_t)(kt * W128::A_STAGE_BYTES + 1024));
            return;
        }
        dma_16B(offa[0], abase, a0 + (uint32_t)(kt * W128::A_STAGE_BYTES));
        dma_16B(offa[1], abase, a0 + (uint32_t)(kt * W128::A_STAGE_BYTES + 1024));
    };
    // ---- fragment addresses
    uint32_t ak[8];
    {
        const uint32_t xa = (uint32_t)(G::a_xor(r16) << 4);
#pragma unroll
        for (int kg = 0; kg < 8; kg++) ak[kg] = a0 + (uint32_t)(r16 * G::A_ROW_BYTES + 4 * q) + (((uint32_t)kg << 4) ^ xa);
    }
    const uint32_t bk_lo = lds0 + (uint32_t)(q * G::KPIECE_BYTES + r16 * 4);
    const uint32_t bk_hi = bk_lo + 2u * W128::B_TILE_BYTES;
    // ---- C staging: 16 rows x 32 columns (128 B per row); 16-byte column c of row r is stored at c ^ ((r >> 1) & 7): conflict-free
    // for the writers (16 rows of one 16-byte column) and the readers (8 columns of two rows)
    const uint32_t ep = a0 + 4u * W128::A_STAGE_BYTES;
    const uint32_t ep_wr0 = ep + (uint32_t)(r16 * 128) + (uint32_t)(((uint32_t)q ^ (((uint32_t)r16 >> 1) & 7u)) << 4);         // block j even
    const uint32_t ep_wr1 = ep + (uint32_t)(r16 * 128) + (uint32_t)(((uint32_t)(4 + q) ^ (((uint32_t)r16 >> 1) & 7u)) << 4);   // block j odd
    const int rr = lane >> 3, rc = lane & 7;   // reader: row rr (+ 8 p), 16-byte column rc
    const uint32_t ep_rd0 = ep + (uint32_t)(rr * 128) + (uint32_t)(((uint32_t)rc ^ (((uint32_t)rr >> 1) & 7u)) << 4);
    const uint32_t ep_rd1 = ep + (uint32_t)((rr + 8) * 128) + (uint32_t)(((uint32_t)rc ^ (((uint32_t)(rr + 8) >> 1) & 7u)) << 4);
    const uint32_t offc = (uint32_t)(rr * g.ldc + 4 * rc) * 4u;   // row rr, columns 4 rc .. of a 32-column group

    const int64_t stride = (int64_t)gridDim.x * W128::NW;
    int64_t strip = (int64_t)blockIdx.x * W128::NW + wave;
    issue(0, strip);
    issue(1, strip);
    issue(2, strip);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // W and K-tile 0 have landed (K-tiles 1, 2 may be on their way)
    __syncthreads();   // W is complete in LDS (the only barrier of the kernel)
    gemm_f32x4acc acc[8];
    float a[2], b[2][8];
    const float alpha = g.alpha;
    // one strip: K-tile kt is multiplied while K-tiles kt + 1 .. kt + 3 (the last ones: the NEXT strip's first) are in flight or landed.
    // STORES = the number of store instructions of the previous strip's epilogue that may still be in flight (0 on a wavefront's
    // first strip, else 8): the waits count them so that they are never waited for.
    auto strip_body = [&](auto stores_tag, int64_t cur, int64_t next) {
        constexpr int STORES = decltype(stores_tag)::value;
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[j][r] = 0.f;
        issue(3, cur);
        GNNX_W128_KTILE(0);
        // K-tile 1 must have landed; younger: K-tiles 2, 3 (4 instructions) and, issued between K-tile 1's and 2's DMA, the last strip's stores
        if constexpr (STORES == 8) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        issue(0, next);
        GNNX_W128_KTILE(1);
        if constexpr (STORES == 8) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // K-tile 2; younger: 3, next 0, and the stores (behind K-tile 2's DMA)
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        issue(1, next);
        GNNX_W128_KTILE(2);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // K-tile 3; younger: next 0, next 1 (the last strip's stores are older than K-tile 3's DMA)
        issue(2, next);
        GNNX_W128_KTILE(3);
        // ---- epilogue: four groups of 32 columns through the staging area
        char *ctile = reinterpret_cast<char *>(g.C + row0_of(cur) * g.ldc);
#pragma unroll
        for (int c4 = 0; c4 < 4; c4++) {
            gemm_f32x4acc v0 = acc[2 * c4], v1 = acc[2 * c4 + 1];
            if (alpha != 1.0f) {
                v0 = v0 * alpha;
                v1 = v1 * alpha;
            }
            asm volatile("ds_write_b128 %0, %1" ::"v"(ep_wr0), "v"(v0) : "memory");
            asm volatile("ds_write_b128 %0, %1" ::"v"(ep_wr1), "v"(v1) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            gemm_f32x4acc o0, o1;
            asm volatile("ds_read_b128 %0, %1" : "=v"(o0) : "v"(ep_rd0) : "memory");
            asm volatile("ds_read_b128 %0, %1" : "=v"(o1) : "v"(ep_rd1) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o0), "+v"(o1)::"memory");
            char *cg = (GNNX_ABLATE(1) ? reinterpret_cast<char *>(g.C + (int64_t)wave * 16 * g.ldc) : ctile) + 128 * c4;   // wave-uniform: columns 32 c4 .. (A/B 1: every strip stores over the same rows: L2 absorbs them)
            *reinterpret_cast<gemm_f32x4acc *>(cg + offc) = o0;
            *reinterpret_cast<gemm_f32x4acc *>(cg + (int64_t)8 * g.ldc * 4 + offc) = o1;
        }
        // the next strip's K-tile 0 must have landed; younger: its K-tiles 1, 2 (4 instructions) and the 8 stores above
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    };
    if (strip < n_strips) {
        strip_body(std::integral_constant<int, 0>{}, strip, strip + stride);
        for (strip += stride; strip < n_strips; strip += stride) strip_body(std::integral_constant<int, 8>{}, strip, strip + stride);
    }
}
#undef GNNX_W128_KTILE
#undef GNNX_W128_STEP
#undef GNNX_W128_MFMA
#undef GNNX_W128_WAIT
#endif  // GNNX_EXPERIMENTS

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

#ifdef GNNX_EXPERIMENTS
// Calibration: register-only MFMA loop (4 independent accumulators per wavefront, no memory traffic) -- what the
// fp32 matrix pipe sustains on THIS chip at the clock it holds under load; the GEMM's fraction of that is the honest
// utilisation figure next to the 157.3 TFLOP/s datasheet peak.
template <int mode>
__global__ __launch_bounds__(256) void mfma_peak_kernel(int iters, float *sink)
{
    __shared__ float lds[4096];
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = (float)(threadIdx.x + r);
    float a = 1.0f + threadIdx.x * 1e-3f, b = 1.0f - threadIdx.x * 1e-3f;
    lds[threadIdx.x] = a;
    __syncthreads();
    const uint32_t la = (uint32_t)(uintptr_t)(gemm_lds_void_t *)lds + (threadIdx.x & 63) * 16;
    float d0 = a, d1 = b, d2 = a, d3 = b;
    gemm_f32x4 w0 = {a, b, a, b}, w1 = w0;
    uint32_t sc = (uint32_t)iters;
    // mode (measurement only): what one extra instruction per MFMA costs the matrix pipe
    //   0 none | 1: 1 VALU | 2: 2 VALU | 4: 4 VALU | 10: 1 ds_read_b32 | 11: 1 ds_read_b128 per 2 MFMA | 12: 1 ds_read_b128 per MFMA
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            if (mode == 1 || mode == 2 || mode == 4) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(d0) : "v"(d1));
            if (mode == 2 || mode == 4) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(d1) : "v"(d0));
            if (mode == 4) {
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(d2) : "v"(d3));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(d3) : "v"(d2));
            }
            if (mode == 20) asm volatile("s_nop 0");
            if (mode == 21) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc));
            if (mode == 10) asm volatile("ds_read_b32 %0, %1" : "=v"(d2) : "v"(la) : "memory");
            if (mode == 11 && (i & 1)) asm volatile("ds_read_b128 %0, %1" : "=v"(w0) : "v"(la) : "memory");
            if (mode == 12) asm volatile("ds_read_b128 %0, %1" : "=v"(w1) : "v"(la) : "memory");
        }
        if (mode >= 10) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d2), "+v"(w0), "+v"(w1)::"memory");
    }
    float s = d0 + d1 + d2 + d3 + w0.x + w1.y + (float)sc;
#pragma unroll
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][7];
    if (s == 12345.678f) sink[0] = s;  // keep the loop alive
}
#endif  // GNNX_EXPERIMENTS

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
        const char *e = experiment_env("GNNX_GEMM_TILE");
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
    // > 64 KB of dynamic LDS needs the opt-in: once per kernel instantiation AND device, checked against what the runtime grants
    static std::atomic<uint64_t> attr_done{0};
    const int rc_lds = lds_opt_in(&gemm_kernel<C, A_KC, B_KC, VA, VB>, lds, attr_done, "gemm_kernel");
    if (rc_lds != GNNX_OK) return rc_lds;
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
#ifdef GNNX_EXPERIMENTS
    case kK16: return launch_cfg<CfgK16, A_KC, B_KC>(g, splits, va, vb, st);
    case kWide32: return launch_cfg<CfgWide32, A_KC, B_KC>(g, splits, va, vb, st);
    case kTall: return launch_cfg<CfgTall, A_KC, B_KC>(g, splits, va, vb, st);
#endif
    case kTall32: return launch_cfg<CfgTall32, A_KC, B_KC>(g, splits, va, vb, st);
    default: return launch_cfg<CfgDefault, A_KC, B_KC>(g, splits, va, vb, st);
    }
}


template <class C, bool B_KC>
int launch_stream_cfg(const GemmArgs &g, int64_t m_tiles, int64_t gy, hipStream_t st)
{
    constexpr size_t lds = C::lds_bytes(true, B_KC);
    static std::atomic<uint64_t> attr_done{0};
    const int rc_lds = lds_opt_in(&gemm_stream_kernel<C, B_KC>, lds, attr_done, "gemm_stream_kernel");
    if (rc_lds != GNNX_OK) return rc_lds;
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

// ---- the same loop for dW = dH^T . X (reduction over the node dimension, split-K) --------------------------------------
// Both operands are k-major as they lie in memory (dH [K][M], X [K][N]): each K-tile is 32 rows of 256 floats per operand, one
// wave-instruction per row, rows padded to 272 words in LDS; a lane's fragment word is base + immediate for both operands
// (two address registers in all).  A workgroup owns one 256 x 256 output tile and a K range (blockIdx.z); its partial tile
// goes to the split-K slab once, at the end, and the slabs are summed in a fixed order by splitk_reduce_kernel.
template <class GEO, int STAGE_ID, int KG>
__device__ __forceinline__ void dma_tn_read_group(float (&a)[4], float (&b)[4], uint32_t ak, uint32_t bk)
{
    constexpr int SA = STAGE_ID * GEO::AT_STAGE_BYTES + GEO::kgroup_off(GEO::BM, KG);
    constexpr int SB = STAGE_ID * GEO::B_STAGE_BYTES + GEO::kgroup_off(GEO::BN, KG);
    lds_read_b32<SA + 0 * 64>(a[0], ak);
    lds_read_b32<SA + 1 * 64>(a[1], ak);
    lds_read_b32<SA + 2 * 64>(a[2], ak);
    lds_read_b32<SA + 3 * 64>(a[3], ak);
    lds_read_b32<SB + 0 * 64>(b[0], bk);
    lds_read_b32<SB + 1 * 64>(b[1], bk);
    lds_read_b32<SB + 2 * 64>(b[2], bk);
    lds_read_b32<SB + 3 * 64>(b[3], bk);
}
#define GNNX_DMATN_STEP(ST, KG, cur, nxt)                                 \
    dma_tn_read_group<GEO, ST, KG + 1>(a[nxt], b[nxt], ak, bk);           \
    GNNX_DMA2_WAIT(8, cur);                                               \
    GNNX_DMA2_MFMA(cur)
#define GNNX_DMATN_KTILE(ST)                                              \
    dma_tn_read_group<GEO, ST, 0>(a[0], b[0], ak, bk);                    \
    GNNX_DMATN_STEP(ST, 0, 0, 1);                                         \
    GNNX_DMATN_STEP(ST, 1, 1, 0);                                         \
    GNNX_DMATN_STEP(ST, 2, 0, 1);                                         \
    GNNX_DMATN_STEP(ST, 3, 1, 0);                                         \
    GNNX_DMATN_STEP(ST, 4, 0, 1);                                         \
    GNNX_DMATN_STEP(ST, 5, 1, 0);                                         \
    GNNX_DMATN_STEP(ST, 6, 0, 1);                                         \
    GNNX_DMA2_WAIT(0, 1);                                                 \
    GNNX_DMA2_MFMA(1)

// both operands k-major: A^T tile [32 k][BM m], B tile [32 k][BN n]
template <int WM_, int WN_>
struct DmaGeoTN : DmaGeo<WM_, WN_> {
    using Base = DmaGeo<WM_, WN_>;
    static constexpr int AT_STAGE_BYTES = Base::kstage_bytes(Base::BM);
    static constexpr int AT_PIECE_BYTES = Base::KPIECE_BYTES;
    static constexpr int AT_PER_WAVE = (Base::BK / Base::rows_per_instr(Base::BM)) / Base::NW;
    static constexpr int LDS_BYTES_TN = 2 * AT_STAGE_BYTES + 2 * Base::B_STAGE_BYTES;
    static_assert(AT_PER_WAVE >= 1, "too many wavefronts for the operand tile");
};

template <int WM_, int WN_>
__global__ __launch_bounds__(64 * WM_ * WN_) void gemm_dma_tn_kernel(GemmArgs g)
{
    using GEO = DmaGeoTN<WM_, WN_>;
    constexpr int BM = GEO::BM, BN = GEO::BN, BK = GEO::BK, NW = GEO::NW;
    constexpr int APW = GEO::AT_PER_WAVE, BPW = GEO::B_PER_WAVE;
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];  // [2][A^T stage] then [2][B stage]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / GEO::WN, wn = wave % GEO::WN;
    const int q = lane >> 4, r16 = lane & 15;
    const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
    // split z takes K-tile pairs [z T / S, (z + 1) T / S): whole pairs, sizes differing by at most one (19 531 pairs on 256 splits
    // are 76 or 77 each; ceil-sized ranges would leave the last two splits empty)
    const int64_t kbeg = (int64_t)blockIdx.z * g.k_pairs / gridDim.z * 64;
    const int64_t kend = ((int64_t)blockIdx.z + 1) * g.k_pairs / gridDim.z * 64;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(gemm_lds_void_t *)lds_raw;
    constexpr uint32_t B0 = 2 * GEO::AT_STAGE_BYTES;
    // DMA: wave-instruction (wave + NW u) covers 1 (256-wide) or 2 (128-wide) k rows; a lane moves 4 consecutive m (n)
    uint32_t offa[APW], offb[BPW], pa[APW], pb[BPW];
#pragma unroll
    for (int u = 0; u < APW; u++) {
        constexpr int RPI = GEO::rows_per_instr(BM), LPR = 64 / RPI;
        offa[u] = (uint32_t)(GEO::krow_of(BM, wave + NW * u, lane) * g.lda + 4 * (lane % LPR)) * 4u;
        pa[u] = lds0 + (uint32_t)((wave + NW * u) * GEO::AT_PIECE_BYTES);
    }
#pragma unroll
    for (int u = 0; u < BPW; u++) {
        constexpr int RPI = GEO::rows_per_instr(BN), LPR = 64 / RPI;
        offb[u] = (uint32_t)(GEO::krow_of(BN, wave + NW * u, lane) * g.ldb + 4 * (lane % LPR)) * 4u;
        pb[u] = lds0 + B0 + (uint32_t)((wave + NW * u) * GEO::B_PIECE_BYTES);
    }
    const float *acol = g.A + m0, *bcol = g.B + n0;
    auto issue = [&](int stage, int64_t k0) {
        const float *abase = acol + k0 * g.lda;
        const float *bbase = bcol + k0 * g.ldb;
#pragma unroll
        for (int u = 0; u < (APW > BPW ? APW : BPW); u++) {
            if (u < APW) dma_16B(offa[u < APW ? u : 0], abase, pa[u < APW ? u : 0] + stage * GEO::AT_STAGE_BYTES);
            if (u < BPW) dma_16B(offb[u < BPW ? u : 0], bbase, pb[u < BPW ? u : 0] + stage * GEO::B_STAGE_BYTES);
        }
    };
    const uint32_t ak = lds0 + (uint32_t)(q * GEO::KPIECE_BYTES + (wm * 64 + r16) * 4);
    const uint32_t bk = lds0 + B0 + (uint32_t)(q * GEO::KPIECE_BYTES + (wn * 64 + r16) * 4);
    gemm_f32x4acc acc[4][4];
    float a[2][4], b[2][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[i][j][r] = 0.f;
    if (kbeg < kend) issue(0, kbeg);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int64_t k0 = kbeg; k0 < kend; k0 += 2 * BK) {
        issue(1, k0 + BK);
        GNNX_DMATN_KTILE(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (k0 + 2 * BK < kend) issue(0, k0 + 2 * BK);
        GNNX_DMATN_KTILE(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // D = mfma(b, a): lane holds partial C[m0 + wm*64 + 16 i + r16][n0 + wn*64 + 16 j + 4 q .. + 3] -> slab of this split
    float *out = g.slab + (int64_t)blockIdx.z * g.M * g.N + (m0 + wm * 64 + r16) * g.N + n0 + wn * 64 + 4 * q;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) *reinterpret_cast<gemm_f32x4acc *>(out + (int64_t)(16 * i) * g.N + 16 * j) = acc[i][j];
}

bool dma_shape_ok(int64_t M, int64_t N, int64_t K)
{
    return K % 64 == 0 && N % 4 == 0 && N >= 64 && M >= 8 * 256;   // N off the 128 grid: guarded last column tile (NG)
}

template <int WM, int WN, bool NG, int BKT = 32>
int launch_dma_geo(const GemmArgs &g, hipStream_t st, int64_t *rows_done, const GemmFuse *fuse, int64_t *partial_rows, int fuse_mode)
{
    using GEO = DmaGeo<WM, WN, BKT>;
    constexpr int BM = GEO::BM, BN = GEO::BN, BK = GEO::BK;
    if ((int64_t)BM * g.lda >= (1ll << 28) || (int64_t)BK * g.ldb >= (1ll << 28) || (int64_t)BM * g.ldc >= (1ll << 28))
        return GNNX_OK;  // per-lane BYTE offsets are 32-bit
    // plain products (and the send-slot epilogue) cover ragged last rows with one more, overlapping tile (gemm_dma_kernel: row0_of)
    const bool cover = (!fuse && fuse_mode != 3) || (fuse && fuse_mode == 4);
    const int64_t m_tiles = cover ? ceil_div(g.M, (int64_t)BM) : g.M / BM, cols = ceil_div(g.N, (int64_t)BN);
    if (g.M < BM) return GNNX_OK;
    constexpr size_t lds = GEO::LDS_BYTES;   // 4 x 4: 2 x (32 KB + 34 KB) = 132 KB; 4 x 2: 2 x (32 KB + 17 KB) = 98 KB; 2 x 2: 66 KB
    constexpr int wg_per_cu = 160 * 1024 / lds >= 2 ? 2 : 1;   // resident workgroups: as many as the LDS of a CU holds
    static const int wgpcu_env = [] { const char *e = experiment_env("GNNX_GEMM_WGPCU"); return e ? atoi(e) : 0; }();   // A/B: resident workgroups per CU
    int64_t gy = ceil_div((int64_t)kNumCU * (wgpcu_env > 0 ? wgpcu_env : wg_per_cu), cols);
    if (gy > m_tiles) gy = m_tiles;
    static std::atomic<uint64_t> done_plain{0}, done_fuse{0}, done_stats{0};
    static const int ablate = [] { const char *e = experiment_env("GNNX_GEMM_ABLATE"); return e ? atoi(e) : 0; }();
    const dim3 grid((uint32_t)cols, (uint32_t)gy, 1);
    if (fuse && fuse_mode == 4) {
        if constexpr (NG || BKT != 32) return GNNX_OK;
        else {
            if (fuse->ldy % 4 || !aligned16(fuse->colsum_partial)) return GNNX_OK;
            static std::atomic<uint64_t> done_send{0};
            constexpr size_t lds_send = lds + 2 * (size_t)BM * 32;   // + the two slot-list buffers
            int rc = lds_opt_in(&gemm_dma_kernel<WM, WN, 4, false, 32>, lds_send, done_send, "gemm_dma_kernel");
            if (rc) return rc;
            hipLaunchKernelGGL((gemm_dma_kernel<WM, WN, 4, false, 32>), grid, dim3(GEO::NT), lds_send, st, g, m_tiles, ablate, *fuse);
        }
    } else if (fuse_mode == 3) {
        static std::atomic<uint64_t> done_bf16{0};
        int rc = lds_opt_in(&gemm_dma_kernel<WM, WN, 3, NG, BKT>, lds, done_bf16, "gemm_dma_kernel");
        if (rc) return rc;
        hipLaunchKernelGGL((gemm_dma_kernel<WM, WN, 3, NG, BKT>), grid, dim3(GEO::NT), lds, st, g, m_tiles, ablate, GemmFuse{});
    } else if (fuse && fuse_mode == 2) {
        if (!aligned16(fuse->ymask)) return GNNX_OK;
        int rc = lds_opt_in(&gemm_dma_kernel<WM, WN, 2, NG, BKT>, lds, done_stats, "gemm_dma_kernel");
        if (rc) return rc;
        hipLaunchKernelGGL((gemm_dma_kernel<WM, WN, 2, NG, BKT>), grid, dim3(GEO::NT), lds, st, g, m_tiles, ablate, *fuse);
        if (partial_rows) *partial_rows = gy;
    } else if (fuse) {
        if (fuse->ldy % 4 || !aligned16(fuse->ymask) || (int64_t)BM * fuse->ldy >= (1ll << 28)) return GNNX_OK;
        int rc = lds_opt_in(&gemm_dma_kernel<WM, WN, 1, NG, BKT>, lds, done_fuse, "gemm_dma_kernel");
        if (rc) return rc;
        hipLaunchKernelGGL((gemm_dma_kernel<WM, WN, 1, NG, BKT>), grid, dim3(GEO::NT), lds, st, g, m_tiles, ablate, *fuse);
        if (partial_rows) *partial_rows = gy;
    } else {
        bool defer = false;
#ifdef GNNX_EXPERIMENTS
        // A/B only (GNNX_GEMM_DEFER=1): the C tile parked in a second accumulator set and stored straight from registers under the
        // next tile's K-tiles (gemm_dma_kernel: DEFER).  Measured SLOWER than the LDS-staged epilogue -- 10 M x 128 x 128: X.W^T 3.20
        // against 3.07 ms, dH.W 2.98 against 2.87 -- 64-byte row segments cost more on the store path than the LDS round trip they
        // save, even spread over the tile: the product library does not instantiate it.
        if constexpr (WM * WN <= 8 && BKT == 32) {
            static const int defer_env = [] { const char *e = experiment_env("GNNX_GEMM_DEFER"); return e ? atoi(e) : 0; }();
            defer = defer_env != 0 && g.K >= 4 * BK && ablate == 0;
            if (defer) {
                static std::atomic<uint64_t> done_defer{0};
                int rc = lds_opt_in(&gemm_dma_kernel<WM, WN, 0, NG, BKT, true>, lds, done_defer, "gemm_dma_kernel");
                if (rc) return rc;
                hipLaunchKernelGGL((gemm_dma_kernel<WM, WN, 0, NG, BKT, true>), grid, dim3(GEO::NT), lds, st, g, m_tiles, ablate, GemmFuse{});
            }
        }
#endif
        if (!defer) {
            int rc = lds_opt_in(&gemm_dma_kernel<WM, WN, 0, NG, BKT>, lds, done_plain, "gemm_dma_kernel");
            if (rc) return rc;
            hipLaunchKernelGGL((gemm_dma_kernel<WM, WN, 0, NG, BKT>), grid, dim3(GEO::NT), lds, st, g, m_tiles, ablate, GemmFuse{});
        }
    }
    GNNX_LAUNCH_CHECK();
    *rows_done = m_tiles * BM < g.M ? m_tiles * BM : g.M;
    return GNNX_OK;
}

// *rows_done = number of leading rows of C written by the LDS-DMA kernel (0: shape not eligible).  B must be k-major ([K][N]).
int launch_dma(const GemmArgs &g, hipStream_t st, int64_t *rows_done, const GemmFuse *fuse = nullptr, int64_t *partial_rows = nullptr,
               int fuse_mode = 1)
{
    *rows_done = 0;
    if (!dma_shape_ok(g.M, g.N, g.K)) return GNNX_OK;
    if (g.lda % 4 || g.ldb % 4 || g.ldc % 4 || !aligned16(g.A) || !aligned16(g.B) || !aligned16(g.C)) return GNNX_OK;
    if (fuse_mode == 3 && g.alpha != 1.0f) return GNNX_OK;
#ifdef GNNX_EXPERIMENTS
    static const int geo256 = [] { const char *e = experiment_env("GNNX_GEMM_GEO256"); return e ? atoi(e) : 0; }();
    if (g.N % 256 == 0 && geo256 == 2416) return launch_dma_geo<2, 4, false, 16>(g, st, rows_done, fuse, partial_rows, fuse_mode);
    // A/B: the 256 x 128 tile (8 wavefronts, 128 VGPRs, 98 KB of LDS per CU) for 256-wide outputs too: half of every SIMD's registers
    // stay free for wavefronts of ANOTHER kernel (an aggregation on a second stream: scripts/exp_concurrent.py)
    if (g.N % 256 == 0 && geo256 == 42) return launch_dma_geo<4, 2, false>(g, st, rows_done, fuse, partial_rows, fuse_mode);
#endif
    // The last, partly filled round of tiles (1.25 M rows x 256: 4 882 tiles of 256 x 256 on 256 CUs = 19 rounds and 18 tiles that cost
    // a 20th; 1 M rows x 128: 15 rounds and 66 tiles) goes to a launch of its own on SMALLER tiles, so that it occupies many CUs for a
    // fraction of a round: 128 x 128 (4 wavefronts) or 256 x 128 (8).  Same MFMA chain per output element in every geometry: same
    // bits.  Plain products and the send-slot epilogue only (the column-sum epilogues index their partials by workgroup).
    static const int tail_env = [] { const char *e = experiment_env("GNNX_GEMM_TAIL"); return e ? atoi(e) : 1; }();
    const bool plain = (!fuse && fuse_mode != 3) || (fuse && fuse_mode == 4);
    const int64_t mt256 = g.M / 256, r = mt256 % kNumCU;
    auto with_tail = [&](auto main_geo, auto tail_geo) -> int {   // whole rounds on the main geometry, the rest on the tail's
        GemmArgs gm = g;
        gm.M = (mt256 - r) * 256;
        int64_t rows_main = 0;
        int rc = main_geo(gm, fuse, &rows_main);
        *rows_done = rows_main;
        if (rc || rows_main != gm.M) return rc;
        GemmArgs gt = g;
        gt.A += rows_main * g.lda;
        gt.C += rows_main * g.ldc;
        gt.M = g.M - rows_main;
        GemmFuse ft{};
        if (fuse) {
            ft = *fuse;
            ft.slots += rows_main * 8;
        }
        int64_t rows_tail = 0;
        rc = tail_geo(gt, fuse ? &ft : nullptr, &rows_tail);
        *rows_done = rows_main + rows_tail;
        return rc;
    };
    if (g.N % 256 == 0) {
        auto g44 = [&](const GemmArgs &a, const GemmFuse *f, int64_t *rows) { return launch_dma_geo<4, 4, false>(a, st, rows, f, partial_rows, fuse_mode); };
        auto g42 = [&](const GemmArgs &a, const GemmFuse *f, int64_t *rows) { return launch_dma_geo<4, 2, false>(a, st, rows, f, partial_rows, fuse_mode); };
        auto g22 = [&](const GemmArgs &a, const GemmFuse *f, int64_t *rows) { return launch_dma_geo<2, 2, false>(a, st, rows, f, partial_rows, fuse_mode); };
        if (tail_env > 0 && plain && g.N == 256 && mt256 > kNumCU && r > 0 && r <= kNumCU / 4) return with_tail(g44, g22);   // a quarter of a round
        if (tail_env > 0 && plain && g.N == 256 && mt256 > kNumCU && r > 0 && r <= kNumCU / 2) return with_tail(g44, g42);   // half
        return launch_dma_geo<4, 4, false>(g, st, rows_done, fuse, partial_rows, fuse_mode);
    }
    static const int geo128 = [] { const char *e = experiment_env("GNNX_GEMM_GEO128"); return e ? atoi(e) : 0; }();
    if (geo128 == 22) {   // A/B: 128 x 128 tiles, two resident workgroups of 4 wavefronts per CU
        if (g.N % 128 == 0) return launch_dma_geo<2, 2, false>(g, st, rows_done, fuse, partial_rows, fuse_mode);
        return launch_dma_geo<2, 2, true>(g, st, rows_done, fuse, partial_rows, fuse_mode);
    }
#ifdef GNNX_EXPERIMENTS
    if (geo128 == 4216) {   // A/B: the 256 x 128 tile on K-tiles of 16, two resident workgroups per CU
        if (g.N % 128 == 0) return launch_dma_geo<4, 2, false, 16>(g, st, rows_done, fuse, partial_rows, fuse_mode);
        return launch_dma_geo<4, 2, true, 16>(g, st, rows_done, fuse, partial_rows, fuse_mode);
    }
#endif
#ifdef GNNX_EXPERIMENTS
    static const int w128_env = [] { const char *e = experiment_env("GNNX_GEMM_W128"); return e ? atoi(e) : 0; }();
    if (w128_env > 0 && geo128 == 0 && !fuse && fuse_mode != 3 && g.N == 128 && g.K == 128 && (int64_t)W128::ROWS * g.lda < (1ll << 28) &&
        (int64_t)128 * g.ldb < (1ll << 28) && g.M >= W128::ROWS) {
        // A/B: K = N = 128 with W resident in LDS, a strip of 16 rows per wavefront, no barrier in the loop (gemm_w128_kernel)
        static std::atomic<uint64_t> done_w128{0};
        int rc = lds_opt_in(&gemm_w128_kernel, (size_t)W128::LDS_BYTES, done_w128, "gemm_w128_kernel");
        if (rc) return rc;
        const int64_t n_strips = ceil_div(g.M, (int64_t)W128::ROWS);
        int64_t gx = ceil_div(n_strips, (int64_t)W128::NW);
        if (gx > kNumCU) gx = kNumCU;
        static const int ablate = [] { const char *e = experiment_env("GNNX_GEMM_ABLATE"); return e ? atoi(e) : 0; }();
        hipLaunchKernelGGL(gemm_w128_kernel, dim3((uint32_t)gx), dim3(W128::NT), (size_t)W128::LDS_BYTES, st, g, n_strips, ablate);
        GNNX_LAUNCH_CHECK();
        *rows_done = g.M;
        return GNNX_OK;
    }
#endif
    if (g.N % 128 == 0) {
        if (tail_env > 0 && geo128 == 0 && plain && g.N == 128 && mt256 > kNumCU && r > 0 && r <= kNumCU / 2) {   // 128 x 128: half a round
            auto g42 = [&](const GemmArgs &a, const GemmFuse *f, int64_t *rows) { return launch_dma_geo<4, 2, false>(a, st, rows, f, partial_rows, fuse_mode); };
            auto g22 = [&](const GemmArgs &a, const GemmFuse *f, int64_t *rows) { return launch_dma_geo<2, 2, false>(a, st, rows, f, partial_rows, fuse_mode); };
            return with_tail(g42, g22);
        }
        return launch_dma_geo<4, 2, false>(g, st, rows_done, fuse, partial_rows, fuse_mode);
    }
    return launch_dma_geo<4, 2, true>(g, st, rows_done, fuse, partial_rows, fuse_mode);
}

bool dma_tn_shape_ok(int64_t M, int64_t N, int64_t K) { return M % 128 == 0 && N % 128 == 0 && K % 64 == 0 && K >= 64 * 1024; }

template <int WM, int WN>
int launch_dma_tn_geo(const GemmArgs &g, int splits, hipStream_t st)
{
    using GEO = DmaGeoTN<WM, WN>;
    constexpr size_t lds = GEO::LDS_BYTES_TN;   // 4 x 4: 136 KB; 2 x 2: 64 KB (two workgroups per CU)
    static std::atomic<uint64_t> done{0};
    int rc = lds_opt_in(&gemm_dma_tn_kernel<WM, WN>, lds, done, "gemm_dma_tn_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL((gemm_dma_tn_kernel<WM, WN>), dim3((uint32_t)(g.N / GEO::BN), (uint32_t)(g.M / GEO::BM), (uint32_t)splits),
                       dim3(GEO::NT), lds, st, g);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}

int launch_dma_tn(const GemmArgs &g, int splits, hipStream_t st)
{
    if (g.M % 256 == 0 && g.N % 256 == 0) return launch_dma_tn_geo<4, 4>(g, splits, st);
    return launch_dma_tn_geo<2, 2>(g, splits, st);
}

// Y[c][r] = X[r][c] for the small weight matrix (tile through LDS; the general entry point is gnnx_transpose_f32)
__global__ __launch_bounds__(256) void gemm_transpose_w_kernel(const float *X, int64_t ldx, int64_t n_rows, int64_t n_cols, float *Y,
                                                                int64_t ldy)
{
    __shared__ float t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < n_rows && c0 + tx < n_cols) t[k][tx] = X[(r0 + k) * ldx + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < n_cols && r0 + tx < n_rows) Y[(c0 + k) * ldy + r0 + tx] = t[tx][k];
}

}  // namespace

#ifdef GNNX_EXPERIMENTS
// Measurement entry of the EXPERIMENTS build only (scripts/exp_gemm.py; not declared in include/gnnx.h): a register-only f32 MFMA
// loop, optionally with extra instructions per MFMA (GNNX_PEAK_MODE), to calibrate what bounds such a loop on this chip.
GNNX_API int gnnx_mfma_peak_f32(int32_t iters, int32_t n_workgroups, float *d_sink, double *flops_out, void *stream)
{
    GNNX_REQUIRE(iters > 0 && n_workgroups > 0 && d_sink, GNNX_ERR_INVALID_ARG, "bad arguments");
    static const int peak_mode = [] { const char *e = experiment_env("GNNX_PEAK_MODE"); return e ? atoi(e) : 0; }();
    const dim3 pg((uint32_t)n_workgroups), pb(256);
    hipStream_t pst = as_stream(stream);
    switch (peak_mode) {
    case 1: hipLaunchKernelGGL(mfma_peak_kernel<1>, pg, pb, 0, pst, iters, d_sink); break;
    case 2: hipLaunchKernelGGL(mfma_peak_kernel<2>, pg, pb, 0, pst, iters, d_sink); break;
    case 4: hipLaunchKernelGGL(mfma_peak_kernel<4>, pg, pb, 0, pst, iters, d_sink); break;
    case 10: hipLaunchKernelGGL(mfma_peak_kernel<10>, pg, pb, 0, pst, iters, d_sink); break;
    case 11: hipLaunchKernelGGL(mfma_peak_kernel<11>, pg, pb, 0, pst, iters, d_sink); break;
    case 12: hipLaunchKernelGGL(mfma_peak_kernel<12>, pg, pb, 0, pst, iters, d_sink); break;
    case 20: hipLaunchKernelGGL(mfma_peak_kernel<20>, pg, pb, 0, pst, iters, d_sink); break;
    case 21: hipLaunchKernelGGL(mfma_peak_kernel<21>, pg, pb, 0, pst, iters, d_sink); break;
    default: hipLaunchKernelGGL(mfma_peak_kernel<0>, pg, pb, 0, pst, iters, d_sink); break;
    }
    GNNX_LAUNCH_CHECK();
    if (flops_out) *flops_out = (double)n_workgroups * 4 /*waves*/ * 4 /*acc*/ * (double)iters * (2.0 * 32 * 32 * 2);
    return GNNX_OK;
}
#endif  // GNNX_EXPERIMENTS

GNNX_API int gnnx_gemm_workspace(int transA, int transB, int64_t M, int64_t N, int64_t K, size_t *bytes)
{
    (void)transB;
    GNNX_REQUIRE(bytes && M >= 0 && N >= 0 && K >= 0, GNNX_ERR_INVALID_ARG, "bad arguments");
    *bytes = 0;
    if (transA && M > 0 && N > 0 && K > 0) {
        int splits = choose_splits(M, N, K);
        if (splits > 1) *bytes = sizeof(float) * (size_t)(splits + 1) * (size_t)M * (size_t)N;   // + the slab of the last K % 64 rows
    } else if (!transA && transB && dma_shape_ok(M, N, K)) {
        *bytes = sizeof(float) * (size_t)K * (size_t)N;  // W^T as [K][N] for the LDS-DMA kernel
    }
    return GNNX_OK;
}

// C = (A . B) with the ReLU mask of the layer below applied, colsum[n] = sum_m C[m][n]: the stacked layers' backward step
// G_{l-1} = (dH_l . W_l) (.) (Y_{l-1} > 0), db_{l-1} = colsum(G_{l-1}) in ONE pass over the output (LDS-DMA kernel, FUSE epilogue);
// rows outside whole 256-row tiles and shapes the kernel does not cover go through gemm -> mask -> colsum.
GNNX_API int gnnx_gemm_relu_colsum_workspace(int64_t M, int64_t N, int64_t K, size_t *bytes)
{
    (void)K;
    GNNX_REQUIRE(bytes && M >= 0 && N >= 0, GNNX_ERR_INVALID_ARG, "bad arguments");
    size_t cs = 0;
    int rc = gnnx_colsum_workspace(M > 256 ? M : 256, (int32_t)N, &cs);
    if (rc) return rc;
    *bytes = sizeof(float) * 256 * (size_t)N + cs;
    return GNNX_OK;
}

GNNX_API int gnnx_gemm_relu_colsum_f32(int64_t M, int64_t N, int64_t K, const float *d_A, int64_t lda, const float *d_B, int64_t ldb,
                                       const float *d_Ymask, int64_t ldy, float *d_C, int64_t ldc, float *d_colsum, void *d_workspace,
                                       size_t workspace_bytes, void *stream)
{
    GNNX_REQUIRE(M >= 0 && N > 0 && K > 0 && N < (1ll << 31), GNNX_ERR_INVALID_ARG, "bad sizes");
    GNNX_REQUIRE(d_colsum, GNNX_ERR_INVALID_ARG, "colsum is null");
    size_t need = 0;
    gnnx_gemm_relu_colsum_workspace(M, N, K, &need);
    GNNX_REQUIRE(d_workspace && workspace_bytes >= need, GNNX_ERR_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    if (M == 0) {
        GNNX_HIP_CHECK(hipMemsetAsync(d_colsum, 0, sizeof(float) * (size_t)N, st));
        return GNNX_OK;
    }
    GNNX_REQUIRE(d_A && d_B && d_Ymask && d_C && lda >= K && ldb >= N && ldy >= N && ldc >= N, GNNX_ERR_INVALID_ARG, "null pointer or ld");
    float *partial = static_cast<float *>(d_workspace);
    void *cs_ws = static_cast<char *>(d_workspace) + sizeof(float) * 256 * (size_t)N;
    const size_t cs_bytes = workspace_bytes - sizeof(float) * 256 * (size_t)N;
    int64_t rows = 0, prow = 0;
    GemmArgs g{};
    g.M = M; g.N = N; g.K = K; g.A = d_A; g.lda = lda; g.B = d_B; g.ldb = ldb; g.C = d_C; g.ldc = ldc; g.alpha = 1.f; g.beta = 0.f;
    g.k_per_split = K;
    GemmFuse fu{d_Ymask, ldy, partial};
    int rc = launch_dma(g, st, &rows, &fu, &prow);
    if (rc) return rc;
    float beta = 0.f;
    if (rows > 0) {   // the workgroups' partial column sums -> colsum, fixed order
        rc = gnnx_colsum_f32(partial, N, prow, (int32_t)N, 0.f, d_colsum, cs_ws, cs_bytes, stream);
        if (rc) return rc;
        beta = 1.f;
    }
    if (rows < M) {   // ragged tail (or a shape the fused kernel does not take): three plain passes over those rows
        const int64_t mr = M - rows;
        float *cr = d_C + rows * ldc;
        const float *yr = d_Ymask + rows * ldy;
        rc = gnnx_gemm_f32(0, 0, mr, N, K, 1.f, d_A + rows * lda, lda, d_B, ldb, 0.f, cr, ldc, nullptr, 0, stream);
        if (rc) return rc;
        rc = gnnx_bn_relu_bwd_f32(yr, ldy, yr, ldy, cr, ldc, mr, (int32_t)N, nullptr, nullptr, 0.f, nullptr, nullptr, 1, cr, ldc, nullptr, nullptr,
                                  nullptr, 0, stream);
        if (rc) return rc;
        rc = gnnx_colsum_f32(cr, ldc, mr, (int32_t)N, beta, d_colsum, cs_ws, cs_bytes, stream);
        if (rc) return rc;
    }
    return GNNX_OK;
}

// ---- opt-in: H = X . W^T with the BatchNorm batch statistics of H in the same pass (single-pass shifted variance) --------------
__global__ void gemm_row0_kernel(const float *A, const float *B, int64_t ldb, int64_t N, int64_t K, int transB, float *shift)
{
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float acc = 0.f;
    for (int64_t k = 0; k < K; k++) acc = fmaf(A[k], transB ? B[n * ldb + k] : B[k * ldb + n], acc);
    shift[n] = acc;   // row 0 of C (any rounding will do: it is only the shift of the one-pass variance)
}

__global__ void shifted_sums_rows_kernel(const float *C, int64_t ldc, int64_t n_rows, int64_t N, const float *shift, float *sum, float *sq)
{
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f, q = 0.f;
    const float sh = shift[n];
    for (int64_t r = 0; r < n_rows; r++) {
        const float d = C[r * ldc + n] - sh;
        s += d;
        q += d * d;
    }
    sum[n] = s;
    sq[n] = q;
}

__global__ void bn_stats_finalize_kernel(const float *partial, int64_t n_part, int64_t N, int64_t M, const float *shift, float *mean, float *var)
{
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double S = 0.0, Q = 0.0;
    for (int64_t p = 0; p < n_part; p++) {   // fixed order
        S += (double)partial[(2 * p) * N + n];
        Q += (double)partial[(2 * p + 1) * N + n];
    }
    const double m = S / (double)M;
    mean[n] = (float)((double)shift[n] + m);
    const double v = Q / (double)M - m * m;
    var[n] = (float)(v > 0.0 ? v : 0.0);
}

GNNX_API int gnnx_gemm_bn_stats_workspace(int64_t M, int64_t N, int64_t K, size_t *bytes)
{
    GNNX_REQUIRE(bytes && M >= 0 && N >= 0 && K >= 0, GNNX_ERR_INVALID_ARG, "bad arguments");
    size_t bn = 0;
    int rc = gnnx_bn_workspace(M, (int32_t)N, &bn);
    if (rc) return rc;
    // [257][2][N] partial sums + shift[N] + W^T [K][N] + the exact path's scratch (fallback)
    *bytes = sizeof(float) * ((size_t)257 * 2 * N + (size_t)N + (size_t)K * N) + bn + 64;
    return GNNX_OK;
}

GNNX_API int gnnx_gemm_bn_stats_f32(int64_t M, int64_t N, int64_t K, const float *d_X, int64_t ldx, const float *d_W, int64_t ldw,
                                    float *d_H, int64_t ldh, float *d_mean, float *d_var, void *d_workspace, size_t workspace_bytes,
                                    void *stream)
{
    GNNX_REQUIRE(M > 0 && N > 0 && K > 0 && N < (1ll << 31), GNNX_ERR_INVALID_ARG, "bad sizes");
    GNNX_REQUIRE(d_X && d_W && d_H && d_mean && d_var && ldx >= K && ldw >= K && ldh >= N, GNNX_ERR_INVALID_ARG, "null pointer or ld");
    size_t need = 0;
    gnnx_gemm_bn_stats_workspace(M, N, K, &need);
    GNNX_REQUIRE(d_workspace && workspace_bytes >= need, GNNX_ERR_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    float *partial = static_cast<float *>(d_workspace);
    float *shift = partial + (size_t)257 * 2 * N;
    float *wt = shift + (((size_t)N + 3) & ~(size_t)3);
    char *bn_ws = reinterpret_cast<char *>(wt + (size_t)K * N);
    bn_ws += (16 - (reinterpret_cast<uintptr_t>(bn_ws) & 15)) & 15;
    const size_t bn_bytes = workspace_bytes - (size_t)(bn_ws - static_cast<char *>(d_workspace));
    const bool eligible = dma_shape_ok(M, N, K) && ldx % 4 == 0 && ldh % 4 == 0 && aligned16(d_X) && aligned16(d_H) && aligned16(d_workspace);
    if (!eligible) {   // exact two-pass statistics on the plain product
        int rc = gnnx_gemm_f32(0, 1, M, N, K, 1.f, d_X, ldx, d_W, ldw, 0.f, d_H, ldh, wt, sizeof(float) * (size_t)K * N, stream);
        if (rc) return rc;
        return gnnx_bn_stats_f32(d_H, ldh, M, (int32_t)N, d_mean, d_var, bn_ws, bn_bytes, stream);
    }
    hipLaunchKernelGGL(gemm_row0_kernel, dim3((uint32_t)ceil_div(N, 256)), dim3(256), 0, st, d_X, d_W, ldw, N, K, 1, shift);
    GNNX_LAUNCH_CHECK();
    hipLaunchKernelGGL(gemm_transpose_w_kernel, dim3((uint32_t)ceil_div(K, 32), (uint32_t)ceil_div(N, 32)), dim3(256), 0, st, d_W, ldw, N, K, wt, N);
    GNNX_LAUNCH_CHECK();
    GemmArgs g{};
    g.M = M; g.N = N; g.K = K; g.A = d_X; g.lda = ldx; g.B = wt; g.ldb = N; g.C = d_H; g.ldc = ldh; g.alpha = 1.f; g.beta = 0.f;
    g.k_per_split = K;
    GemmFuse fu{shift, 0, partial, nullptr};
    int64_t rows = 0, prow = 0;
    int rc = launch_dma(g, st, &rows, &fu, &prow, 2);
    if (rc) return rc;
    if (rows == 0) {   // the kernel declined (alignment of a sub-buffer): exact path
        rc = gnnx_gemm_f32(0, 1, M, N, K, 1.f, d_X, ldx, d_W, ldw, 0.f, d_H, ldh, wt, sizeof(float) * (size_t)K * N, stream);
        if (rc) return rc;
        return gnnx_bn_stats_f32(d_H, ldh, M, (int32_t)N, d_mean, d_var, bn_ws, bn_bytes, stream);
    }
    if (rows < M) {    // ragged tail: plain product, then its shifted sums as one more partial row
        const int64_t mr = M - rows;
        rc = gnnx_gemm_f32(0, 0, mr, N, K, 1.f, d_X + rows * ldx, ldx, wt, N, 0.f, d_H + rows * ldh, ldh, nullptr, 0, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(shifted_sums_rows_kernel, dim3((uint32_t)ceil_div(N, 256)), dim3(256), 0, st, d_H + rows * ldh, ldh, mr, N, shift,
                           partial + (size_t)(2 * prow) * N, partial + (size_t)(2 * prow + 1) * N);
        GNNX_LAUNCH_CHECK();
        prow++;
    }
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((uint32_t)ceil_div(N, 256)), dim3(256), 0, st, partial, prow, N, M, shift, d_mean, d_var);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}

// ---- opt-in: H = X . W^T stored as bf16 (the feature storage of gnnx_spmm_csr_bf16_f32) straight from the product's epilogue ------
GNNX_API int gnnx_gemm_nt_bf16out_workspace(int64_t M, int64_t N, int64_t K, size_t *bytes)
{
    GNNX_REQUIRE(bytes && M >= 0 && N >= 0 && K >= 0, GNNX_ERR_INVALID_ARG, "bad arguments");
    *bytes = sizeof(float) * ((size_t)K * N + (size_t)256 * N) + 64;   // W^T [K][N] + f32 rows of the ragged tail
    return GNNX_OK;
}

GNNX_API int gnnx_gemm_nt_bf16out_f32(int64_t M, int64_t N, int64_t K, const float *d_X, int64_t ldx, const float *d_W, int64_t ldw,
                                      uint16_t *d_H_bf16, int64_t ldh, void *d_workspace, size_t workspace_bytes, void *stream)
{
    GNNX_REQUIRE(M > 0 && N > 0 && K > 0 && N < (1ll << 31), GNNX_ERR_INVALID_ARG, "bad sizes");
    GNNX_REQUIRE(d_X && d_W && d_H_bf16 && ldx >= K && ldw >= K && ldh >= N, GNNX_ERR_INVALID_ARG, "null pointer or ld");
    GNNX_REQUIRE(dma_shape_ok(M, N, K) && ldx % 4 == 0 && ldh % 4 == 0 && aligned16(d_X) && aligned16(d_H_bf16),
                 GNNX_ERR_SHAPE, "bf16-output product: needs K %% 64 == 0, N %% 4 == 0, N >= 64, M >= 2048, 16-byte aligned rows "
                 "(use gnnx_gemm_f32 + gnnx_f32_to_bf16 for other shapes)");
    size_t need = 0;
    gnnx_gemm_nt_bf16out_workspace(M, N, K, &need);
    GNNX_REQUIRE(d_workspace && workspace_bytes >= need && aligned16(d_workspace), GNNX_ERR_WORKSPACE, "workspace %zu < required %zu",
                 workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    float *wt = static_cast<float *>(d_workspace);
    float *tail = wt + (size_t)K * N;
    hipLaunchKernelGGL(gemm_transpose_w_kernel, dim3((uint32_t)ceil_div(K, 32), (uint32_t)ceil_div(N, 32)), dim3(256), 0, st, d_W, ldw, N, K, wt, N);
    GNNX_LAUNCH_CHECK();
    GemmArgs g{};
    g.M = M; g.N = N; g.K = K; g.A = d_X; g.lda = ldx; g.B = wt; g.ldb = N; g.C = reinterpret_cast<float *>(d_H_bf16); g.ldc = ldh;
    g.alpha = 1.f; g.beta = 0.f; g.k_per_split = K;
    int64_t rows = 0;
    int rc = launch_dma(g, st, &rows, nullptr, nullptr, 3);
    if (rc) return rc;
    GNNX_REQUIRE(rows > 0, GNNX_ERR_SHAPE, "bf16-output product: shape not taken by the LDS-DMA kernel");
    if (rows < M) {   // ragged tail (< 256 rows): f32 product, then the same rounding
        const int64_t mr = M - rows;
        rc = gnnx_gemm_f32(0, 0, mr, N, K, 1.f, d_X + rows * ldx, ldx, wt, N, 0.f, tail, N, nullptr, 0, stream);
        if (rc) return rc;
        rc = gnnx_f32_to_bf16(tail, N, mr, (int32_t)N, d_H_bf16 + rows * ldh, ldh, stream);
        if (rc) return rc;
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
    // dW = dH^T . X over a row count that is no multiple of 64 (a shard's 1.25 M rows): the LDS-DMA kernel takes the whole K-tile
    // pairs, the generic kernel the last K % 64 rows into one more slab, added last by the in-order reduction
    const int64_t k_dma = transA && !transB && splits > 1 && dma_tn_shape_ok(M, N, K - K % 64) ? K - K % 64 : 0;
    if (k_dma > 0) g.k_per_split = ceil_div(k_dma / 64, splits) * 64;  // the LDS-DMA loop takes K-tiles in pairs
    const int slabs = splits + (k_dma > 0 && k_dma < K ? 1 : 0);
    if (splits > 1) {
        size_t need = sizeof(float) * (size_t)slabs * (size_t)M * (size_t)N;
        GNNX_REQUIRE(d_workspace && workspace_bytes >= need, GNNX_ERR_WORKSPACE, "workspace %zu < required %zu",
                     workspace_bytes, need);
        g.slab = static_cast<float *>(d_workspace);
    }
    // tall products with whole tiles: the resident streaming kernel takes the whole M-tiles, the generic kernel the rest
    // tall products with whole tiles: the LDS-DMA kernel takes the whole 256-row tiles (B k-major; for X.W^T the small W is
    // transposed once into the workspace, same products in the same order), the generic kernel the ragged rest
    static const int dma_env = [] { const char *e = experiment_env("GNNX_GEMM_DMA"); return e ? atoi(e) : 1; }();
    if (dma_env > 0 && a_kc && splits == 1 && beta == 0.f && K > 0 && dma_shape_ok(M, N, K) &&
        (!b_kc || (d_workspace && workspace_bytes >= sizeof(float) * (size_t)K * (size_t)N && aligned16(d_workspace)))) {
        GemmArgs gd = g;
        if (b_kc) {
            float *wt = static_cast<float *>(d_workspace);
            hipLaunchKernelGGL(gemm_transpose_w_kernel, dim3((uint32_t)ceil_div(K, 32), (uint32_t)ceil_div(N, 32)), dim3(256), 0, st, d_B, ldb,
                               N, K, wt, N);
            GNNX_LAUNCH_CHECK();
            gd.B = wt;
            gd.ldb = N;
        }
        int64_t rows = 0;
        const int src = launch_dma(gd, st, &rows);
        if (src != GNNX_OK) return src;
        if (rows == g.M) return GNNX_OK;
        g.A += rows * lda;
        g.C += rows * ldc;
        g.M -= rows;
    }
    static const int stream_env = [] { const char *e = experiment_env("GNNX_GEMM_STREAM"); return e ? atoi(e) : 1; }();
    if (stream_env > 0 && a_kc && splits == 1 && beta == 0.f && va && vb && K > 0) {
        int64_t rows = 0;  // leading rows of C written by the streaming kernel
        const int src = launch_stream(g, b_kc, stream_env, st, &rows);
        if (src != GNNX_OK) return src;
        if (rows == g.M) return GNNX_OK;
        g.A += rows * lda;
        g.C += rows * ldc;
        g.M -= rows;  // relative to what the LDS-DMA kernel above left over
    }
    int rc;
    // dW = dH^T . X on the LDS-DMA loop: both operands k-major, whole 256 x 256 tiles, K in whole pairs of K-tiles per split
    const bool dma_tn = dma_env > 0 && !a_kc && !b_kc && splits > 1 && g.slab && k_dma > 0 && g.k_per_split % 64 == 0 &&
                        lda % 4 == 0 && ldb % 4 == 0 && aligned16(d_A) && aligned16(d_B) && aligned16(g.slab) &&
                        32 * lda < (1ll << 28) && 32 * ldb < (1ll << 28);
    int reduce_slabs = splits;
    if (dma_tn) {
        GemmArgs gk = g;
        gk.K = k_dma;
        gk.k_pairs = k_dma / 64;
        rc = launch_dma_tn(gk, splits, st);
        if (rc == GNNX_OK && k_dma < K) {   // the last K % 64 rows: one workgroup per output tile, into slab `splits`
            GemmArgs gr = g;
            gr.A = d_A + k_dma * lda;
            gr.B = d_B + k_dma * ldb;
            gr.K = K - k_dma;
            gr.k_per_split = ceil_div(gr.K, (int64_t)bk) * bk;
            gr.slab = g.slab + (size_t)splits * (size_t)M * (size_t)N;
            rc = launch<false, false>(gr, 1, va, vb, st);
            reduce_slabs = splits + 1;
        }
    } else if (k_dma > 0) {   // the DMA kernel declined (alignment): the generic kernel's own split of all K rows
        g.k_per_split = ceil_div(ksteps, splits) * bk;
        rc = launch<false, false>(g, splits, va, vb, st);
    } else if (a_kc && b_kc) rc = launch<true, true>(g, splits, va, vb, st);
    else if (a_kc && !b_kc) rc = launch<true, false>(g, splits, va, vb, st);
    else if (!a_kc && b_kc) rc = launch<false, true>(g, splits, va, vb, st);
    else rc = launch<false, false>(g, splits, va, vb, st);
    if (rc != GNNX_OK) return rc;
    if (splits > 1) {
        const int64_t blocks = ceil_div(M * N, 32);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, g.slab, reduce_slabs, M, N, alpha, beta,
                           d_C, ldc);
        GNNX_LAUNCH_CHECK();
    }
    return GNNX_OK;
}

// ---- H = X . W^T with the halo pack in the product's epilogue (the sharded step's transform) ---------------------------
// Every row of H listed in d_slots (gnnx_rows_to_slots_f32's table) is ALSO stored to its send-buffer rows, from the registers the
// epilogue stores H from: same bits as gnnx_gemm_f32 followed by gnnx_rows_to_slots_f32, without the pass that reads H back.  Rows the
// LDS-DMA kernel does not take (a ragged tail of < 256 rows, shapes off its grid) go through exactly those two calls.
GNNX_API int gnnx_gemm_nt_rows_to_slots_workspace(int64_t M, int64_t N, int64_t K, size_t *bytes)
{
    GNNX_REQUIRE(bytes && M >= 0 && N >= 0 && K >= 0, GNNX_ERR_INVALID_ARG, "bad arguments");
    *bytes = sizeof(float) * (size_t)K * (size_t)N + 64;   // W^T [K][N]
    return GNNX_OK;
}

GNNX_API int gnnx_gemm_nt_rows_to_slots_f32(int64_t M, int64_t N, int64_t K, const float *d_X, int64_t ldx, const float *d_W, int64_t ldw,
                                            float *d_H, int64_t ldh, const int32_t *d_slots, float *d_send, int64_t ld_send,
                                            void *d_workspace, size_t workspace_bytes, void *stream)
{
    GNNX_REQUIRE(M >= 0 && N > 0 && K > 0 && N < (1ll << 31), GNNX_ERR_INVALID_ARG, "bad sizes");
    if (M == 0) return GNNX_OK;
    GNNX_REQUIRE(d_X && d_W && d_H && d_slots && d_send && ldx >= K && ldw >= K && ldh >= N && ld_send >= N, GNNX_ERR_INVALID_ARG,
                 "null pointer or ld");
    GNNX_REQUIRE(N % 4 == 0 && N / 4 <= 256 && 256 % (N / 4) == 0 && ldh % 4 == 0 && ld_send % 4 == 0 && aligned16(d_H) && aligned16(d_send) &&
                     aligned16(d_slots),
                 GNNX_ERR_UNSUPPORTED, "rows of 16-byte pieces only, as gnnx_rows_to_slots_f32 (N %% 4 == 0, N / 4 a divisor of 256)");
    size_t need = 0;
    gnnx_gemm_nt_rows_to_slots_workspace(M, N, K, &need);
    GNNX_REQUIRE(d_workspace && workspace_bytes >= need, GNNX_ERR_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    int64_t rows = 0;
    if (dma_shape_ok(M, N, K) && N % 128 == 0 && aligned16(d_workspace)) {
        float *wt = static_cast<float *>(d_workspace);
        hipLaunchKernelGGL(gemm_transpose_w_kernel, dim3((uint32_t)ceil_div(K, 32), (uint32_t)ceil_div(N, 32)), dim3(256), 0, st, d_W, ldw, N, K, wt, N);
        GNNX_LAUNCH_CHECK();
        GemmArgs g{};
        g.M = M; g.N = N; g.K = K; g.A = d_X; g.lda = ldx; g.B = wt; g.ldb = N; g.C = d_H; g.ldc = ldh; g.alpha = 1.f; g.beta = 0.f;
        g.k_per_split = K;
        GemmFuse fu{nullptr, ld_send, d_send, d_slots};
        int rc = launch_dma(g, st, &rows, &fu, nullptr, 4);
        if (rc) return rc;
    }
    if (rows < M) {   // what the kernel left: the plain product, then the pack as a pass of its own
        const int64_t mr = M - rows;
        int rc = gnnx_gemm_f32(0, 1, mr, N, K, 1.f, d_X + rows * ldx, ldx, d_W, ldw, 0.f, d_H + rows * ldh, ldh, d_workspace, workspace_bytes, stream);
        if (rc) return rc;
        rc = gnnx_rows_to_slots_f32(d_H + rows * ldh, ldh, mr, (int32_t)N, d_slots + rows * 8, d_send, ld_send, nullptr, 0.f, nullptr, 0, stream);
        if (rc) return rc;
    }
    return GNNX_OK;
}
