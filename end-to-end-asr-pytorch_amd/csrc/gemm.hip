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

template <int PREC, bool A_KCONT, bool B_KCONT, bool VEC>
__global__ __launch_bounds__(NT) void gemm_kernel(int M, int N, int K, float alpha, const float* __restrict__ A,
                                                  long lda, long sA, const float* __restrict__ B, long ldb, long sB,
                                                  float beta, float* __restrict__ C, long ldc, long sC,
                                                  const float* __restrict__ bias, int act, int ksplit, int swz) {
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
    A += (long)bz * sA; B += (long)bz * sB; C += (long)bz * sC;
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
        if (a_full && k_full) g_load_fast<A_KCONT>(ra, a_base, lda, kt0 * BK); else g_load<A_KCONT, VEC>(ra, A, lda, m0, kt0 * BK, M, K);
        if (b_full && k_full) g_load_fast<B_KCONT>(rb, b_base, ldb, kt0 * BK); else g_load<B_KCONT, VEC>(rb, B, ldb, n0, kt0 * BK, N, K);
    }
    if constexpr (PREC == LAS_PREC_BF16) { tile_store_bf16<A_KCONT>(ra, As2[0]); tile_store_bf16<B_KCONT>(rb, Bs2[0]); }
    else { s_store<A_KCONT, T, LD>(ra, As2[0]); s_store<B_KCONT, T, LD>(rb, Bs2[0]); }
    __syncthreads();
    for (int kt = kt0; kt < nk; ++kt) {
        const int cur = (kt - kt0) & 1;
        const T* As = As2[cur];
        const T* Bs = Bs2[cur];
        if (kt + 1 < nk) {                     // next k-tile: global -> registers, in flight across the MFMAs below
            const bool k_full = (kt + 2) * BK <= K;
            if (a_full && k_full) g_load_fast<A_KCONT>(ra, a_base, lda, (kt + 1) * BK); else g_load<A_KCONT, VEC>(ra, A, lda, m0, (kt + 1) * BK, M, K);
            if (b_full && k_full) g_load_fast<B_KCONT>(rb, b_base, ldb, (kt + 1) * BK); else g_load<B_KCONT, VEC>(rb, B, ldb, n0, (kt + 1) * BK, N, K);
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
            if constexpr (PREC == LAS_PREC_BF16) { tile_store_bf16<A_KCONT>(ra, As2[cur ^ 1]); tile_store_bf16<B_KCONT>(rb, Bs2[cur ^ 1]); }
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

// C[m][:] = beta * C[m][:]  (or 0)
__global__ __launch_bounds__(256) void scale2d_kernel(float beta, int N, float* __restrict__ C, long ldc) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < N) { float* p = C + (long)blockIdx.y * ldc + c; *p = beta != 0.f ? beta * (*p) : 0.f; }
}
__global__ __launch_bounds__(256) void colsum_scale_kernel(float beta, int N, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < N) out[c] = beta != 0.f ? beta * out[c] : 0.f;
}
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, long ld, int M, int N, int rows_per,
                                                     float* __restrict__ out) {
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
    if (g == 0 && c < N)
        atomicAdd(&out[c], part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

}  // namespace

extern "C" int las_gemm(int prec, int transA, int transB, int M, int N, int K, float alpha, const float* A,
                        int64_t lda, int64_t strideA, const float* B, int64_t ldb, int64_t strideB, float beta,
                        float* C, int64_t ldc, int64_t strideC, const float* bias, int act, int batch,
                        void* stream) {
    LAS_CHECK_ARG(A && B && C && M >= 0 && N >= 0 && K >= 0 && batch >= 1);
    LAS_CHECK_ARG(prec == LAS_PREC_BF16 || prec == LAS_PREC_F32);
    LAS_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N);
    if (M == 0 || N == 0) return LAS_OK;
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, batch);
    if (grid.y > 65535 || grid.z > 65535) return LAS_E_UNSUPPORTED;
    // split-K for the weight-gradient shapes (few output tiles, K = T*B in the tens of thousands)
    int ksplit = 1;
    const long tiles = (long)grid.x * grid.y;
    if (batch == 1 && !bias && act == LAS_ACT_NONE && tiles < 256 && K >= 2048) {
        ksplit = (int)((512 + tiles - 1) / tiles);
        const int nk = (K + BK - 1) / BK;
        if (ksplit > nk / 8) ksplit = nk / 8;
        if (ksplit > 512) ksplit = 512;          // (one-tile outputs with K in the 10^5..10^6: the first conv layer's dW)
        if (ksplit < 1) ksplit = 1;
    }
    // fast path: every operand vector is a whole, 16-byte aligned float4 inside the matrix
    const bool vecA = (lda % 4 == 0) && (strideA % 4 == 0) && (((uintptr_t)A & 15) == 0) && (transA ? M % 4 == 0 : K % 4 == 0);
    const bool vecB = (ldb % 4 == 0) && (strideB % 4 == 0) && (((uintptr_t)B & 15) == 0) && (transB ? K % 4 == 0 : N % 4 == 0);
    const bool vec = vecA && vecB && K >= 4 && M >= 4 && N >= 4;
    hipStream_t st = (hipStream_t)stream;
    static const int swz_env = getenv("LAS_GEMM_NOSWZ") ? 0 : 1;
    if (ksplit > 1) grid.z = ksplit;
    const int swz = swz_env && (long)grid.x * grid.y * grid.z >= 16;
    if (ksplit > 1) {
        hipLaunchKernelGGL(scale2d_kernel, dim3((N + 255) / 256, M), dim3(256), 0, st, beta, N, C, (long)ldc);
        LAS_LAUNCH_OK();
    }
#define LAS_GEMM_ARGS transA, transB, grid, st, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc, strideC, bias, act, ksplit, swz
    if (prec == LAS_PREC_BF16) return vec ? launch<LAS_PREC_BF16, true>(LAS_GEMM_ARGS) : launch<LAS_PREC_BF16, false>(LAS_GEMM_ARGS);
    return vec ? launch<LAS_PREC_F32, true>(LAS_GEMM_ARGS) : launch<LAS_PREC_F32, false>(LAS_GEMM_ARGS);
#undef LAS_GEMM_ARGS
}

extern "C" int las_colsum(const float* X, int64_t ld, int M, int N, float beta, float* out, void* stream) {
    LAS_CHECK_ARG(X && out && M >= 0 && N > 0 && ld >= N);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_scale_kernel, dim3((N + 255) / 256), dim3(256), 0, st, beta, N, out);
    LAS_LAUNCH_OK();
    if (M == 0) return LAS_OK;
    const int nx = (N + 63) / 64;
    int slices = (2048 + nx - 1) / nx;                       // ~2048 workgroups in total
    if (slices > (M + 63) / 64) slices = (M + 63) / 64;
    if (slices < 1) slices = 1;
    const int rows_per = ((M + slices - 1) / slices + 3) / 4 * 4;
    hipLaunchKernelGGL(colsum_kernel, dim3(nx, (M + rows_per - 1) / rows_per), dim3(256), 0, st, X, ld, M, N, rows_per, out);
    LAS_LAUNCH_OK();
    return LAS_OK;
}
