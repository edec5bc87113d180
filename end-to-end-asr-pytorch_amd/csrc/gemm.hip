// MFMA GEMM with fused bias / tanh / accumulate epilogue (gfx950).
//
//   C[M,N] = alpha * opA(A) * opB(B) + beta * C + bias[N]   ;  optional tanh
//
// Replaces every nn.Linear on the hot path (reference src/asr.py:307,316 proj+tanh; :46,69 ctc_layer;
// :384,419 psi; :41,92 char_trans) and the x*W_ih^T halves of nn.LSTM (:473-481) and their backward
// (dX = dY*W, dW += dY^T*X).  fp32 in HBM; operands are converted while being staged into LDS:
// LAS_PREC_BF16 -> bf16 tiles, v_mfma_f32_16x16x32_bf16;  LAS_PREC_F32 -> f32 tiles, v_mfma_f32_16x16x4_f32.
//
// Tile 128x128x32, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA tiles of 16x16.
// LDS tiles are always [row][k] (k contiguous, row stride padded by 16 B so the 16-lane groups of a
// ds_read_b128 hit distinct banks); k-strided ("transposed") sources are read coalesced along rows and
// scattered into that layout.  Register prefetch of the next k-tile overlaps the MFMAs of the current one.
#include "las_common.h"
#include "gemm_tile.h"

namespace {

using namespace las_tile;

// A16 / B16: that operand is a bf16 matrix in memory (same layout, element strides) -- activation twins written by
// their producers and the bf16 shadow of the weights; C16 (optional): a bf16 copy of the result for the next consumer.
template <int PREC, bool A_KCONT, bool B_KCONT, bool VEC, bool A16 = false, bool B16 = false>
__global__ __launch_bounds__(NT) void gemm_kernel(int M, int N, int K, float alpha, const float* __restrict__ A,
                                                  long lda, long sA, const float* __restrict__ B, long ldb, long sB,
                                                  float beta, float* __restrict__ C, long ldc, long sC,
                                                  const float* __restrict__ bias, int act, int ksplit, int swz,
                                                  bf16_t* __restrict__ C16 = nullptr, long ldc16 = 0) {
    static_assert(!(A16 || B16) || (PREC == LAS_PREC_BF16 && VEC), "bf16 sources: bf16 mode, aligned operands");
    const bf16_t* A16p = (const bf16_t*)A;
    const bf16_t* B16p = (const bf16_t*)B;
    typedef typename Elem<PREC>::T T;
    constexpr int LD = BK + Elem<PREC>::PAD;
    __shared__ __attribute__((aligned(16))) T As2[2][BM * LD];      // double buffered: one barrier per k-tile
    __shared__ __attribute__((aligned(16))) T Bs2[2][BN * LD];

    // blockIdx.z = batch index, or (ksplit > 1, batch == 1) the K-slice whose partial product is added atomically
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so in launch order the
    // tiles that share an A panel land in 8 different L2s and every panel is fetched 8 times.  Give the blocks that share
    // an XCD (id % 8) one contiguous run of tiles instead (bijective for any grid size), n fastest inside the run.
    int bx = blockIdx.x, by = blockIdx.y, bzz = blockIdx.z;
    if (swz) {
        const int nx = gridDim.x, nxy = nx * gridDim.y, nwg = nxy * gridDim.z;
        const int orig = (bzz * (int)gridDim.y + by) * nx + bx;
        const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        bzz = t / nxy; t -= bzz * nxy;
        by = t / nx; bx = t - by * nx;
    }
    const int bz = ksplit > 1 ? 0 : bzz;
    if (A16) A16p += (long)bz * sA; else A += (long)bz * sA;
    if (B16) B16p += (long)bz * sB; else B += (long)bz * sB;
    C += (long)bz * sC;
    if (C16) C16 += (long)bz * sC;
    const int m0 = by * BM, n0 = bx * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int fr = lane & 15, fq = lane >> 4;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    Frag16 ra, rb;
    Frag8h ha, hb;
    const bool a_full = VEC && (m0 + BM <= M), b_full = VEC && (n0 + BN <= N);
    const float* a_base = A_KCONT ? A + (long)(m0 + (threadIdx.x >> 3)) * lda + (threadIdx.x & 7) * 4
                                  : A + (long)((threadIdx.x >> 5) * 4) * lda + m0 + (threadIdx.x & 31) * 4;
    const float* b_base = B_KCONT ? B + (long)(n0 + (threadIdx.x >> 3)) * ldb + (threadIdx.x & 7) * 4
                                  : B + (long)((threadIdx.x >> 5) * 4) * ldb + n0 + (threadIdx.x & 31) * 4;
    const int nk_all = (K + BK - 1) / BK;
    const int per = (nk_all + ksplit - 1) / ksplit;
    const int kt0 = ksplit > 1 ? bzz * per : 0;
    const int nk = min(nk_all, kt0 + per);
    if (kt0 >= nk) return;
    {
        const bool k_full = (kt0 + 1) * BK <= K;
        if constexpr (A16) g_load_bf16<A_KCONT>(ha, A16p, lda, m0, kt0 * BK, M, K);
        else if (a_full && k_full) g_load_fast<A_KCONT>(ra, a_base, lda, kt0 * BK); else g_load<A_KCONT, VEC>(ra, A, lda, m0, kt0 * BK, M, K);
        if constexpr (B16) g_load_bf16<B_KCONT>(hb, B16p, ldb, n0, kt0 * BK, N, K);
        else if (b_full && k_full) g_load_fast<B_KCONT>(rb, b_base, ldb, kt0 * BK); else g_load<B_KCONT, VEC>(rb, B, ldb, n0, kt0 * BK, N, K);
    }
    if constexpr (PREC == LAS_PREC_BF16) {
        if constexpr (A16) tile_store_bf16_src<A_KCONT>(ha, (bf16_t*)As2[0]); else tile_store_bf16<A_KCONT>(ra, As2[0]);
        if constexpr (B16) tile_store_bf16_src<B_KCONT>(hb, (bf16_t*)Bs2[0]); else tile_store_bf16<B_KCONT>(rb, Bs2[0]);
    }
    else { s_store<A_KCONT, T, LD>(ra, As2[0]); s_store<B_KCONT, T, LD>(rb, Bs2[0]); }
    __syncthreads();
    for (int kt = kt0; kt < nk; ++kt) {
        const int cur = (kt - kt0) & 1;
        const T* As = As2[cur];
        const T* Bs = Bs2[cur];
        if (kt + 1 < nk) {                     // next k-tile: global -> registers, in flight across the MFMAs below
            const bool k_full = (kt + 2) * BK <= K;
            if constexpr (A16) g_load_bf16<A_KCONT>(ha, A16p, lda, m0, (kt + 1) * BK, M, K);
            else if (a_full && k_full) g_load_fast<A_KCONT>(ra, a_base, lda, (kt + 1) * BK); else g_load<A_KCONT, VEC>(ra, A, lda, m0, (kt + 1) * BK, M, K);
            if constexpr (B16) g_load_bf16<B_KCONT>(hb, B16p, ldb, n0, (kt + 1) * BK, N, K);
            else if (b_full && k_full) g_load_fast<B_KCONT>(rb, b_base, ldb, (kt + 1) * BK); else g_load<B_KCONT, VEC>(rb, B, ldb, n0, (kt + 1) * BK, N, K);
        }
        if constexpr (PREC == LAS_PREC_BF16) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = tile_frag_bf16<A_KCONT>(As, wm + i * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = tile_frag_bf16<B_KCONT>(Bs, wn + j * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int kb = 0; kb < BK; kb += 16) {
                float4 af[4], bfr[4];          // lane holds k = kb + 4*fq + {0..3}; MFMA step j uses element j
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *(const float4*)(As + (wm + i * 16 + fr) * LD + kb + fq * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) bfr[j] = *(const float4*)(Bs + (wn + j * 16 + fr) * LD + kb + fq * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bfr[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bfr[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].z, bfr[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].w, bfr[j].w, acc[i][j], 0, 0, 0);
                    }
            }
        }
        if (kt + 1 < nk) {                     // registers -> the other LDS buffer (nobody reads it this iteration)
            if constexpr (PREC == LAS_PREC_BF16) {
                if constexpr (A16) tile_store_bf16_src<A_KCONT>(ha, (bf16_t*)As2[cur ^ 1]); else tile_store_bf16<A_KCONT>(ra, As2[cur ^ 1]);
                if constexpr (B16) tile_store_bf16_src<B_KCONT>(hb, (bf16_t*)Bs2[cur ^ 1]); else tile_store_bf16<B_KCONT>(rb, Bs2[cur ^ 1]);
            }
            else { s_store<A_KCONT, T, LD>(ra, As2[cur ^ 1]); s_store<B_KCONT, T, LD>(rb, Bs2[cur ^ 1]); }
        }
        __syncthreads();
    }
    // epilogue: C/D layout col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn + j * 16 + fr;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + i * 16 + fq * 4 + r;
                if (m >= M) continue;
                float* c = C + (long)m * ldc + n;
                if (ksplit > 1) {                       // C was pre-scaled by beta; bias/act are not allowed here
                    atomicAdd(c, alpha * acc[i][j][r]);
                    continue;
                }
                float v = alpha * acc[i][j][r] + bv;
                if (beta != 0.f) v += beta * (*c);
                if (act == LAS_ACT_TANH) v = tanhf(v);
                else if (act == LAS_ACT_RELU) v = v < 0.f ? 0.f : v;
                *c = v;
                if (C16) C16[(long)m * ldc16 + n] = f2bf(v);
            }
        }
    }
}

template <int PREC, bool VEC>
int launch(int ta, int tb, dim3 grid, hipStream_t st, int M, int N, int K, float alpha, const float* A, long lda,
           long sA, const float* B, long ldb, long sB, float beta, float* C, long ldc, long sC, const float* bias,
           int act, int ksplit, int swz) {
#define LAS_GEMM_GO(AK, BK_)                                                                                          \
    hipLaunchKernelGGL((gemm_kernel<PREC, AK, BK_, VEC>), grid, dim3(NT), 0, st, M, N, K, alpha, A, lda, sA, B, ldb, \
                       sB, beta, C, ldc, sC, bias, act, ksplit, swz)
    if (!ta && tb) LAS_GEMM_GO(true, true);
    else if (!ta && !tb) LAS_GEMM_GO(true, false);
    else if (ta && !tb) LAS_GEMM_GO(false, false);
    else LAS_GEMM_GO(false, true);
#undef LAS_GEMM_GO
    LAS_LAUNCH_OK();
    return LAS_OK;
}

// bf16-source variants (bf16 mode, aligned operands only)
template <bool A16, bool B16>
int launch16(int ta, int tb, dim3 grid, hipStream_t st, int M, int N, int K, float alpha, const void* A, long lda, long sA,
             const void* B, long ldb, long sB, float beta, float* C, long ldc, long sC, const float* bias, int act, int ksplit,
             int swz, bf16_t* C16, long ldc16) {
#define LAS_GEMM_GO(AK, BK_)                                                                                              \
    hipLaunchKernelGGL((gemm_kernel<LAS_PREC_BF16, AK, BK_, true, A16, B16>), grid, dim3(NT), 0, st, M, N, K, alpha,        \
                       (const float*)A, lda, sA, (const float*)B, ldb, sB, beta, C, ldc, sC, bias, act, ksplit, swz, C16, ldc16)
    if (!ta && tb) LAS_GEMM_GO(true, true);
    else if (!ta && !tb) LAS_GEMM_GO(true, false);
    else if (ta && !tb) LAS_GEMM_GO(false, false);
    else LAS_GEMM_GO(false, true);
#undef LAS_GEMM_GO
    LAS_LAUNCH_OK();
    return LAS_OK;
}

// C[m][:] = beta * C[m][:]  (or 0)
__global__ __launch_bounds__(256) void scale2d_kernel(float beta, int N, float* __restrict__ C, long ldc) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < N) { float* p = C + (long)blockIdx.y * ldc + c; *p = beta != 0.f ? beta * (*p) : 0.f; }
}
__global__ __launch_bounds__(256) void colsum_scale_kernel(float beta, int N, float* __restrict__ out, float* __restrict__ out2) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < N) {
        out[c] = beta != 0.f ? beta * out[c] : 0.f;
        if (out2) out2[c] = beta != 0.f ? beta * out2[c] : 0.f;
    }
}
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, long ld, int M, int N, int rows_per,
                                                     float* __restrict__ out, float* __restrict__ out2) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int m0 = blockIdx.y * rows_per, m1 = min(M, m0 + rows_per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < N) {
        int m = m0 + g;
        for (; m + 12 < m1; m += 16) {
            s0 += X[(long)m * ld + c]; s1 += X[(long)(m + 4) * ld + c];
            s2 += X[(long)(m + 8) * ld + c]; s3 += X[(long)(m + 12) * ld + c];
        }
        for (; m < m1; m += 4) s0 += X[(long)m * ld + c];
    }
    part[g][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && c < N) {
        const float v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        atomicAdd(&out[c], v);
        if (out2) atomicAdd(&out2[c], v);
    }
}

}  // namespace

int las_gemm_big(int transA, int transB, int M, int N, int K, float alpha, const void* A, int64_t lda, const void* B, int64_t ldb,
                 float beta, float* C, int64_t ldc, const float* bias, int act, void* C16, int64_t ldc16, int cfg, hipStream_t st);

static int gemm_common(int prec, int transA, int transB, int M, int N, int K, float alpha, const void* A, int a16,
                       int64_t lda, int64_t strideA, const void* B, int b16, int64_t ldb, int64_t strideB, float beta,
                       float* C, int64_t ldc, int64_t strideC, const float* bias, int act, int batch, void* C16,
                       int64_t ldc16, void* stream) {
    LAS_CHECK_ARG(A && B && C && M >= 0 && N >= 0 && K >= 0 && batch >= 1);
    LAS_CHECK_ARG(prec == LAS_PREC_BF16 || prec == LAS_PREC_F32);
    LAS_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N);
    LAS_CHECK_ARG(!C16 || ldc16 >= N);
    if (M == 0 || N == 0) return LAS_OK;
    if (prec == LAS_PREC_BF16 && a16 && b16 && batch == 1) {
        // both operands are bf16 twins: the large-tile LDS-DMA kernel (gemm_big.hip) where the shape fills the chip with it
        static const char* knob = LAS_AB_KNOB("LAS_GEMM_BIG");          // (diagnostic build: 1 | 2 force a tile, -1 = never)
        const int cfg = knob ? atoi(knob) : 0;
        const int rc = cfg < 0 ? LAS_E_UNSUPPORTED
                               : las_gemm_big(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, act, C16, ldc16, cfg, (hipStream_t)stream);
        if (rc != LAS_E_UNSUPPORTED) return rc;
    }
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, batch);
    if (grid.y > 65535 || grid.z > 65535) return LAS_E_UNSUPPORTED;
    // split-K for the weight-gradient shapes (few output tiles, K = T*B in the tens of thousands)
    int ksplit = 1;
    const long tiles = (long)grid.x * grid.y;
    if (batch == 1 && !bias && act == LAS_ACT_NONE && !C16 && tiles < 256 && K >= 2048) {
        ksplit = (int)((512 + tiles - 1) / tiles);
        const int nk = (K + BK - 1) / BK;
        if (ksplit > nk / 8) ksplit = nk / 8;
        if (ksplit > 512) ksplit = 512;          // (one-tile outputs with K in the 10^5..10^6: the first conv layer's dW)
        if (ksplit < 1) ksplit = 1;
    }
    // fast path: every operand vector is a whole, 16-byte aligned vector inside the matrix (4 floats / 8 bf16)
    const int va = a16 ? 8 : 4, vb = b16 ? 8 : 4;
    const bool vecA = (lda % va == 0) && (strideA % va == 0) && (((uintptr_t)A & 15) == 0) && (transA ? M % va == 0 : K % va == 0);
    const bool vecB = (ldb % vb == 0) && (strideB % vb == 0) && (((uintptr_t)B & 15) == 0) && (transB ? K % vb == 0 : N % vb == 0);
    const bool vec = vecA && vecB && K >= 8 && M >= 8 && N >= 8;
    if ((a16 || b16) && (!vec || prec != LAS_PREC_BF16)) return LAS_E_UNSUPPORTED;       // (the caller falls back to its fp32 copy)
    hipStream_t st = (hipStream_t)stream;
    static const int swz_env = LAS_AB_KNOB("LAS_GEMM_NOSWZ") ? 0 : 1;
    if (ksplit > 1) grid.z = ksplit;
    const int swz = swz_env && (long)grid.x * grid.y * grid.z >= 16;
    if (ksplit > 1 && beta != 1.f) {               // (beta = 1: the slices are added to what is there)
        hipLaunchKernelGGL(scale2d_kernel, dim3((N + 255) / 256, M), dim3(256), 0, st, beta, N, C, (long)ldc);
        LAS_LAUNCH_OK();
    }
    if (a16 || b16) {
#define LAS_G16_ARGS transA, transB, grid, st, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc, strideC, bias, act, ksplit, swz, (bf16_t*)C16, ldc16
        if (a16 && b16) return launch16<true, true>(LAS_G16_ARGS);
        if (a16) return launch16<true, false>(LAS_G16_ARGS);
        return launch16<false, true>(LAS_G16_ARGS);
#undef LAS_G16_ARGS
    }
    if (C16) {                                   // fp32 sources with a bf16 copy of the result: the bf16-mode kernel family
        if (prec != LAS_PREC_BF16 || !vec) return LAS_E_UNSUPPORTED;
        return launch16<false, false>(transA, transB, grid, st, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc,
                                      strideC, bias, act, ksplit, swz, (bf16_t*)C16, ldc16);
    }
    const float* Af = (const float*)A;
    const float* Bf = (const float*)B;
#define LAS_GEMM_ARGS transA, transB, grid, st, M, N, K, alpha, Af, lda, strideA, Bf, ldb, strideB, beta, C, ldc, strideC, bias, act, ksplit, swz
    if (prec == LAS_PREC_BF16) return vec ? launch<LAS_PREC_BF16, true>(LAS_GEMM_ARGS) : launch<LAS_PREC_BF16, false>(LAS_GEMM_ARGS);
    return vec ? launch<LAS_PREC_F32, true>(LAS_GEMM_ARGS) : launch<LAS_PREC_F32, false>(LAS_GEMM_ARGS);
#undef LAS_GEMM_ARGS
}

extern "C" int las_gemm(int prec, int transA, int transB, int M, int N, int K, float alpha, const float* A,
                        int64_t lda, int64_t strideA, const float* B, int64_t ldb, int64_t strideB, float beta,
                        float* C, int64_t ldc, int64_t strideC, const float* bias, int act, int batch,
                        void* stream) {
    return gemm_common(prec, transA, transB, M, N, K, alpha, A, 0, lda, strideA, B, 0, ldb, strideB, beta, C, ldc, strideC, bias, act,
                       batch, nullptr, 0, stream);
}

extern "C" int las_gemm_ex(int prec, int transA, int transB, int M, int N, int K, float alpha, const void* A, int a_bf16,
                           int64_t lda, int64_t strideA, const void* B, int b_bf16, int64_t ldb, int64_t strideB, float beta,
                           float* C, int64_t ldc, int64_t strideC, const float* bias, int act, int batch, void* C16,
                           int64_t ldc16, void* stream) {
    return gemm_common(prec, transA, transB, M, N, K, alpha, A, a_bf16, lda, strideA, B, b_bf16, ldb, strideB, beta, C, ldc, strideC,
                       bias, act, batch, C16, ldc16, stream);
}

extern "C" int las_colsum2(const float* X, int64_t ld, int M, int N, float beta, float* out, float* out2, void* stream) {
    LAS_CHECK_ARG(X && out && out != out2 && M >= 0 && N > 0 && ld >= N);
    hipStream_t st = (hipStream_t)stream;
    if (beta != 1.f) {                                       // (beta = 1: the sums are added to what is there)
        hipLaunchKernelGGL(colsum_scale_kernel, dim3((N + 255) / 256), dim3(256), 0, st, beta, N, out, out2);
        LAS_LAUNCH_OK();
    }
    if (M == 0) return LAS_OK;
    const int nx = (N + 63) / 64;
    int slices = (2048 + nx - 1) / nx;                       // ~2048 workgroups in total
    if (slices > (M + 63) / 64) slices = (M + 63) / 64;
    if (slices < 1) slices = 1;
    const int rows_per = ((M + slices - 1) / slices + 3) / 4 * 4;
    hipLaunchKernelGGL(colsum_kernel, dim3(nx, (M + rows_per - 1) / rows_per), dim3(256), 0, st, X, ld, M, N, rows_per, out, out2);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

extern "C" int las_colsum(const float* X, int64_t ld, int M, int N, float beta, float* out, void* stream) {
    return las_colsum2(X, ld, M, N, beta, out, nullptr, stream);
}
