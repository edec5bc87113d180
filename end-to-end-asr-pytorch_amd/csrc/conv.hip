// Implicit-GEMM 3x3 convolution over channels-last activations (VGG front-end, reference src/asr.py:515-520,
// 546-553), on the 128x?x32 MFMA tile of gemm.hip.  No patch matrix is materialised: the operand loader gathers
// the shifted pixels while staging into LDS, so HBM sees each activation about once (the 9 taps of a tile re-read
// the same ~210 pixels from L2) instead of the 9x-inflated im2col write + read.
//
//   conv_fwd_kernel   out[p][n] = epi( sum_{tap,c} in[p + off(tap)][c] * w[n][tap*C + c] )      p = (b,t,f) pixel
//                     epi 0: + bias[n], ReLU (forward);  epi 1: zero where mask[p][n] <= 0 (data gradient fused
//                     with the ReLU of the layer below; the data gradient is the same convolution of dY with the
//                     tap-flipped, transposed weights)
//   conv_wgrad_kernel dw[co][tap*C + c] += sum_p dy[p][co] * in[p + off(tap)][c]               split over pixel
//                     ranges, partial tiles added with atomics; workgroups sharing a pixel range are placed on
//                     the same XCD so the gathered activations are fetched into one L2 only.
// C (input channels of the gathered operand) is a multiple of 32, so a 32-wide k-slab never straddles two taps.
#include "las_common.h"
#include "gemm_tile.h"
#include "conv.h"

namespace {

using namespace las_tile;

struct ConvGeo { int T, F, C; long P; };      // image [.,T,F,C], P = B*T*F pixels

template <int PREC, int MI>
__device__ __forceinline__ void mma_tile(const typename Elem<PREC>::T* __restrict__ As, const typename Elem<PREC>::T* __restrict__ Bs,
                                         f32x4 (&acc)[MI][4], int wm, int wn, int fr, int fq) {
    constexpr int LD = BK + Elem<PREC>::PAD;
    if constexpr (PREC == LAS_PREC_BF16) {
        bf16x8 af[MI], bfr[4];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(As + (wm + i * 16 + fr) * LD + fq * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(Bs + (wn + j * 16 + fr) * LD + fq * 8);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
        for (int kb = 0; kb < BK; kb += 16) {
            float4 af[MI], bfr[4];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *(const float4*)(As + (wm + i * 16 + fr) * LD + kb + fq * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *(const float4*)(Bs + (wn + j * 16 + fr) * LD + kb + fq * 4);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bfr[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bfr[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].z, bfr[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].w, bfr[j].w, acc[i][j], 0, 0, 0);
                }
        }
    }
}

// k-contiguous tile rows tid/8 + 32p (p < NP) -> LDS [row][k]
template <typename T, int LD, int NP>
__device__ __forceinline__ void s_store_rows(const Frag16& reg, T* __restrict__ tile) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < NP; ++p)
        store4_ct(tile + ((tid >> 3) + 32 * p) * LD + (tid & 7) * 4, reg.v[4 * p], reg.v[4 * p + 1], reg.v[4 * p + 2],
                  reg.v[4 * p + 3]);
}

// WN = waves along N: 1 -> 128 x 64 tile (each wave 32 x 64), 2 -> 128 x 128 tile (each wave 64 x 64).  N == 64*WN.
template <int PREC, int WN, int EPI>
__global__ __launch_bounds__(NT) void conv_fwd_kernel(const float* __restrict__ in, ConvGeo g, const float* __restrict__ w,
                                                      const float* __restrict__ bias, const float* __restrict__ mask,
                                                      float* __restrict__ out) {
    typedef typename Elem<PREC>::T T;
    constexpr int LD = BK + Elem<PREC>::PAD;
    constexpr int BNc = 64 * WN, MI = 2 * WN, NPB = 2 * WN;
    __shared__ __attribute__((aligned(16))) T As2[2][BM * LD];
    __shared__ __attribute__((aligned(16))) T Bs2[2][BNc * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = WN == 2 ? (wave >> 1) * 64 : wave * 32, wn = WN == 2 ? (wave & 1) * 64 : 0;
    const int fr = lane & 15, fq = lane >> 4;
    const long m0 = (long)blockIdx.x * BM;
    const int C = g.C, K9 = 9 * C, spt = C / BK, nk = 9 * spt;

    // per-thread gather state: 4 pixels (tile rows tid/8 + 32p), their 9-bit neighbour-validity masks
    const float* ab[4];
    int vm[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const long r = m0 + (tid >> 3) + 32 * p;
        const long pc = r < g.P ? r : g.P - 1;
        const int f = (int)(pc % g.F), t = (int)((pc / g.F) % g.T);
        int m = 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int tt = t + tap / 3 - 1, ff = f + tap % 3 - 1;
            if (tt >= 0 && tt < g.T && ff >= 0 && ff < g.F) m |= 1 << tap;
        }
        vm[p] = r < g.P ? m : 0;
        ab[p] = in + pc * C + (tid & 7) * 4;
    }
    const float* wb = w + (long)(tid >> 3) * K9 + (tid & 7) * 4;

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    Frag16 ra, rb;
    auto load = [&](int s) {
        const int tap = s / spt, c0 = (s - tap * spt) * BK;
        const long off = ((long)(tap / 3 - 1) * g.F + (tap % 3 - 1)) * C + c0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const bool ok = (vm[p] >> tap) & 1;
            const float4 v = ldg4(ok ? ab[p] + off : ab[p]);
            ra.v[4 * p + 0] = ok ? v.x : 0.f; ra.v[4 * p + 1] = ok ? v.y : 0.f;
            ra.v[4 * p + 2] = ok ? v.z : 0.f; ra.v[4 * p + 3] = ok ? v.w : 0.f;
        }
#pragma unroll
        for (int p = 0; p < NPB; ++p) {
            const float4 v = ldg4(wb + (long)(32 * p) * K9 + s * BK);
            rb.v[4 * p + 0] = v.x; rb.v[4 * p + 1] = v.y; rb.v[4 * p + 2] = v.z; rb.v[4 * p + 3] = v.w;
        }
    };
    load(0);
    s_store_rows<T, LD, 4>(ra, As2[0]);
    s_store_rows<T, LD, NPB>(rb, Bs2[0]);
    __syncthreads();
    for (int s = 0; s < nk; ++s) {
        const int cur = s & 1;
        if (s + 1 < nk) load(s + 1);
        mma_tile<PREC, MI>(As2[cur], Bs2[cur], acc, wm, wn, fr, fq);
        if (s + 1 < nk) {
            s_store_rows<T, LD, 4>(ra, As2[cur ^ 1]);
            s_store_rows<T, LD, NPB>(rb, Bs2[cur ^ 1]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = wn + j * 16 + fr;
        const float bv = (EPI == 0 && bias) ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long m = m0 + wm + i * 16 + fq * 4 + r;
                if (m >= g.P) continue;
                float v = acc[i][j][r];
                if (EPI == 0) { v += bv; v = v < 0.f ? 0.f : v; }
                else if (mask) { if (!(mask[m * BNc + n] > 0.f)) v = 0.f; }
                out[m * BNc + n] = v;
            }
        }
    }
}

template <int PREC>
__global__ __launch_bounds__(NT) void conv_wgrad_kernel(const float* __restrict__ dy, int Co, const float* __restrict__ in,
                                                        ConvGeo g, float* __restrict__ dwr, int ntn, int ks, int per) {
    typedef typename Elem<PREC>::T T;
    constexpr int LD = BK + Elem<PREC>::PAD;
    __shared__ __attribute__((aligned(16))) T As2[2][BM * LD];
    __shared__ __attribute__((aligned(16))) T Bs2[2][BN * LD];
    // XCD-aware placement: consecutive workgroup ids round-robin over the 8 XCDs, so the n-tiles of one pixel range
    // take ids with the same id % 8 and sit next to each other in that XCD's dispatch order
    const int id = blockIdx.x, xcd = id & 7, q = id >> 3;
    const int kslice = (q / ntn) * 8 + xcd, n0 = (q % ntn) * BN;
    if (kslice >= ks) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int fr = lane & 15, fq = lane >> 4;
    const int C = g.C, K9 = 9 * C;
    const long nk_all = (g.P + BK - 1) / BK;
    const long kt0 = (long)kslice * per;
    const long nk = nk_all < kt0 + per ? nk_all : kt0 + per;
    if (kt0 >= nk) return;

    // gathered operand: this thread's 4 output columns (tap, c..c+3) are fixed; its 4 pixels advance by 32 per k-tile
    const int n = n0 + (tid & 31) * 4;
    const bool n_ok = n < K9;
    const int tap = n_ok ? n / C : 0, c = n_ok ? n - tap * C : 0;
    const int dt = tap / 3 - 1, df = tap % 3 - 1;
    const long shift = ((long)dt * g.F + df) * C;
    int pt[4], pf[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const long qx = kt0 * BK + (tid >> 5) * 4 + p;
        pf[p] = (int)(qx % g.F);
        pt[p] = (int)((qx / g.F) % g.T);
    }
    const int adv_f = BK % g.F, adv_t = BK / g.F;
    const bool a_fast = Co == BM;
    const float* a_base = dy + (long)((tid >> 5) * 4) * Co + (tid & 31) * 4;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    Frag16 ra, rb;
    auto load = [&](long kt) {
        const long k0 = kt * BK;
        if (a_fast && k0 + BK <= g.P) g_load_fast<false>(ra, a_base, Co, (int)k0);
        else g_load<false, true>(ra, dy, Co, 0, (int)k0, Co, (int)g.P);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const long qx = k0 + (tid >> 5) * 4 + p;
            const int tt = pt[p] + dt, ff = pf[p] + df;
            const bool ok = n_ok && qx < g.P && tt >= 0 && tt < g.T && ff >= 0 && ff < g.F;
            const long qc = qx < g.P ? qx : g.P - 1;
            const float4 v = ldg4(in + qc * C + c + (ok ? shift : 0));
            rb.v[4 * p + 0] = ok ? v.x : 0.f; rb.v[4 * p + 1] = ok ? v.y : 0.f;
            rb.v[4 * p + 2] = ok ? v.z : 0.f; rb.v[4 * p + 3] = ok ? v.w : 0.f;
            pf[p] += adv_f; pt[p] += adv_t;
            if (pf[p] >= g.F) { pf[p] -= g.F; ++pt[p]; }
            if (pt[p] >= g.T) pt[p] -= g.T;
            if (pt[p] >= g.T) pt[p] -= g.T;
        }
    };
    // bf16: both operands are k-strided (k = pixel) -> k-major LDS images + transposing reads (gemm_tile.h)
    auto store = [&](int buf) {
        if constexpr (PREC == LAS_PREC_BF16) { tile_store_bf16<false>(ra, As2[buf]); tile_store_bf16<false>(rb, Bs2[buf]); }
        else { s_store<false, T, LD>(ra, As2[buf]); s_store<false, T, LD>(rb, Bs2[buf]); }
    };
    load(kt0);
    store(0);
    __syncthreads();
    for (long kt = kt0; kt < nk; ++kt) {
        const int cur = (int)(kt - kt0) & 1;
        if (kt + 1 < nk) load(kt + 1);
        if constexpr (PREC == LAS_PREC_BF16) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = tile_frag_bf16<false>(As2[cur], wm + i * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = tile_frag_bf16<false>(Bs2[cur], wn + j * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        } else mma_tile<PREC, 4>(As2[cur], Bs2[cur], acc, wm, wn, fr, fq);
        if (kt + 1 < nk) store(cur ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nn = n0 + wn + j * 16 + fr;
        if (nn >= K9) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = wm + i * 16 + fq * 4 + r;
                if (m < Co) atomicAdd(dwr + (long)m * K9 + nn, acc[i][j][r]);
            }
    }
}

template <int WN, int EPI>
int launch_fwd(int prec, hipStream_t st, const float* in, ConvGeo g, const float* w, const float* bias, const float* mask,
               float* out) {
    const unsigned nb = (unsigned)((g.P + BM - 1) / BM);
    if (prec == LAS_PREC_BF16)
        hipLaunchKernelGGL((conv_fwd_kernel<LAS_PREC_BF16, WN, EPI>), dim3(nb), dim3(NT), 0, st, in, g, w, bias, mask, out);
    else
        hipLaunchKernelGGL((conv_fwd_kernel<LAS_PREC_F32, WN, EPI>), dim3(nb), dim3(NT), 0, st, in, g, w, bias, mask, out);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

}  // namespace

int las_conv3x3_fwd(int prec, const float* in, int T, int F, int C, long P, const float* w, int N, const float* bias,
                    const float* mask, int epi, float* out, hipStream_t st) {
    if (!(C % BK == 0 && (N == 64 || N == 128) && P > 0 && (((uintptr_t)in | (uintptr_t)w) & 15) == 0)) return LAS_E_UNSUPPORTED;
    const ConvGeo g{T, F, C, P};
    if (N == 64) return epi == 0 ? launch_fwd<1, 0>(prec, st, in, g, w, bias, mask, out) : launch_fwd<1, 1>(prec, st, in, g, w, bias, mask, out);
    return epi == 0 ? launch_fwd<2, 0>(prec, st, in, g, w, bias, mask, out) : launch_fwd<2, 1>(prec, st, in, g, w, bias, mask, out);
}

int las_conv3x3_wgrad(int prec, const float* dy, int Co, const float* in, int T, int F, int C, long P, float* dwr,
                      hipStream_t st) {
    if (!(C % BK == 0 && (Co == 64 || Co == 128) && P > 0 && (((uintptr_t)in | (uintptr_t)dy) & 15) == 0)) return LAS_E_UNSUPPORTED;
    const ConvGeo g{T, F, C, P};
    const int K9 = 9 * C, ntn = (K9 + BN - 1) / BN;
    const long nk_all = (P + BK - 1) / BK;
    long ks = (768 + ntn - 1) / ntn;                      // ~3 workgroups per CU in total
    if (ks > nk_all / 8) ks = nk_all / 8;                 // at least 8 k-tiles (256 pixels) per slice
    if (ks < 1) ks = 1;
    const int per = (int)((nk_all + ks - 1) / ks);
    ks = (nk_all + per - 1) / per;
    LAS_HIP(hipMemsetAsync(dwr, 0, sizeof(float) * (size_t)Co * K9, st));
    const unsigned nb = (unsigned)(8 * ntn * ((ks + 7) / 8));
    if (prec == LAS_PREC_BF16)
        hipLaunchKernelGGL((conv_wgrad_kernel<LAS_PREC_BF16>), dim3(nb), dim3(NT), 0, st, dy, Co, in, g, dwr, ntn, (int)ks, per);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<LAS_PREC_F32>), dim3(nb), dim3(NT), 0, st, dy, Co, in, g, dwr, ntn, (int)ks, per);
    LAS_LAUNCH_OK();
    return LAS_OK;
}
