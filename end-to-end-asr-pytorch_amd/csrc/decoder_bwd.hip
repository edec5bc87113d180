// Attend-and-spell decoder loop, backward through time (one C-ABI call for all L steps).
//
// Hand-written counterpart of what autograd derives for reference src/asr.py:84-107 (Attention.forward
// :410-457, Speller.forward :352-357).  Per step, newest first:
//   cell pointwise bwd (per layer)           dgates from dh (top-layer grad + recurrent carry) and dc carry
//   dgates * [W_ih | W_hh]                   skinny MFMA products on transposed weight copies
//   d a = enc . d ctx (+ loc carry), softmax bwd, energy bwd     att_bwd_step  grid (T'-chunks, B)  HBM-bound on enc / s
//   dh0_{t-1} += dq_pre * W_phi              skinny MFMA product (accumulate)
// Sums over the L steps that are plain contractions (dW of every Linear/LSTMCell, d enc, d psi in dot mode)
// are left to ONE las_gemm each after the loop, on the buffers this call fills.
#include "las_mma.h"
#include "decoder_pk.h"
#include <stdlib.h>

int las_skinny_launch(int prec, const float* x0, long ldx0, const float* w0, long ldw0, int K0, const float* x1,
                      long ldx1, const float* w1, long ldw1, int K1, const float* x2, long ldx2, const float* w2,
                      long ldw2, int K2, int B, int N, const float* bias0, const float* bias1, int mode, float* out,
                      long ldo, int accumulate, int C, const float* c_prev, float* h_out, float* c_out,
                      float* gates_out, hipStream_t st);

namespace {

constexpr int LOC_C = 10, LOC_K = 100, LOC_W = 2 * LOC_K + 1;
constexpr float ATT_SCALE = 2.0f;

// dgates[b][g*C+u] from dh = dh_ext + dh_carry and the running dc; updates dc_carry in place.
__global__ __launch_bounds__(256) void cell_pw_bwd(int B, int C, const float* __restrict__ dh_ext, long ld_ext,
                                                   const float* __restrict__ dh_carry, float* __restrict__ dc_carry,
                                                   const float* __restrict__ gates, const float* __restrict__ c_t,
                                                   const float* __restrict__ c_prev, float* __restrict__ dgates) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, u = i % C;
    const float* g = gates + (long)b * 4 * C;
    const float ig = g[u], fg = g[C + u], gg = g[2 * C + u], og = g[3 * C + u];
    const float dh = (dh_ext ? dh_ext[(long)b * ld_ext + u] : 0.f) + dh_carry[i];
    const float tc = tanhf(c_t[i]);
    const float dc = dh * og * (1.f - tc * tc) + dc_carry[i];
    float* d = dgates + (long)b * 4 * C;
    d[u] = dc * gg * ig * (1.f - ig);
    d[C + u] = dc * c_prev[i] * fg * (1.f - fg);
    d[2 * C + u] = dc * ig * (1.f - gg * gg);
    d[3 * C + u] = dh * tc * og * (1.f - og);
    dc_carry[i] = dc * fg;
}

struct AttBwdArgs {
    int B, Tp, E, A, TC, NCH;
    const float* enc;         // [B][Tp][E]
    const bf16_t* enc16;      // the same as bf16 (EB kernels), or null
    const float* psi; const int32_t* lens;
    const float* att;         // [B][Tp] this step's attention
    const float* dctx; long ld_dctx;   // [B][E] d loss / d context of this step
    const float* ctx; long ld_ctx;     // [B][E] this step's context (saved)
    const float* q;           // [B][A]
    float* dq_pre;            // [B][A]  (+=, atomics; already multiplied by 1-q^2)
    float* de;                // [B][Tp] d loss / d energy of this step (saved for the post-loop contractions)
    // loc
    const float* f;           // [B][10][Tp] location features of this step
    const void* s; int s16;   // [B][Tp][A] tanh(psi + q + u): fp32, or the 16-bit code of las_common.h (s16, bf16 mode)
    const float* w_lp; const float* w_e; const float* conv_w;
    float* df;                // [B][10][Tp] d loss / d f of this step (caller-zeroed; frames < len written)
    const float* df_next;     // [B][10][Tp] d f of step t+1 (NULL at the last step): the location conv of step t+1
    const float* f_next;      //             read THIS step's attention, so its d f carries gradient back to it
};

// grid (NCH, B): attention backward of one step for a chunk of <= 20 frames of one utterance.
//   d a[t] = enc[b,t,:] . dctx  +  sum_c sum_k w[c][k] * df_next[c][t + K - k]        (context + location-conv paths)
//   d e[t] = 2 a[t] (d a[t] - dot),  dot = sum_t a[t] d a[t]
// The softmax dot needs no other chunk:  sum_t a[t] enc[t] . dctx = ctx . dctx  with the saved context, and
// sum_t a[t] * (conv-path term) = <df_next, f_next>  because f_next = conv(a).  Only what the NEXT (earlier) step
// needs stays on the sequential chain: d e, d q (sum over frames of d z) and d f (sum over the attention dim of
// d u * W_lp).  Everything that is a plain sum over the L steps -- d psi, d w_e, d b_e, d W_lp, d conv_w -- is left to
// att_loc_post / att_conv_wgrad after the loop, from the saved s / f / d e / d f.
// 8 waves per workgroup: a chunk's <= 20 frames are <= 3 per wave, which keeps the two per-frame phases short.
constexpr int ATT_NW = 8, ATT_NT = 64 * ATT_NW;
template <bool LOC, int AI, int EV = 4, bool EB = false>  // EV: float4 pieces of an enc row per lane (4: E <= 1024, 8: E <= 2048);
                                                          // EB: the rows come from the bf16 copy of enc (EV / 2 sixteen-byte pieces of
                                                          // 8 values: half the bytes and half the row registers -- with E = 2 048 the fp32
                                                          // form is 168 registers, one workgroup per CU, and the 360 workgroups of
                                                          // BASELINE configs[4] run as two rounds)
__global__ __launch_bounds__(ATT_NT) void att_bwd_step(AttBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float red[32];
    const int b = blockIdx.y, ch = blockIdx.x, t0 = ch * a.TC, t1 = min(t0 + a.TC, a.Tp), len = a.lens[b];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (t0 >= len) {                                 // nothing flows through frames beyond the utterance
        for (int t = t0 + threadIdx.x; t < t1; t += ATT_NT) a.de[(long)b * a.Tp + t] = 0.f;
        return;
    }
    const int tcv = min(t1, len) - t0;               // valid frames in this chunk (> 0 here)
    const int Ep = (a.E + 3) & ~3, W = a.TC + 2 * LOC_K;
    float* dctx_l = sm;                              // [Ep]
    float* de_l = dctx_l + Ep;                       // [TC]
    float* we_l = de_l + a.TC;                       // [A]
    float* wlp_l = we_l + a.A;                       // [10][A]
    float* f_l = wlp_l + LOC_C * a.A;                // [10][TC]
    float* dq_l = f_l + LOC_C * a.TC;                // [ATT_NW][A] per-wave partials
    float* cw_l = dq_l + ATT_NW * a.A;                    // [10][201]
    float* dfh_l = cw_l + LOC_C * LOC_W;             // [10][TC + 200]: df_next[c][t0 - K + i]
    // all s rows this wave touches (<= ATT_ROWS, chunks are <= 20 frames) are requested up front: one round trip
    constexpr int ATT_ROWS = (20 + ATT_NW - 1) / ATT_NW;
    float svr[ATT_ROWS][AI];
    if (LOC) {
#pragma unroll
        for (int r = 0; r < ATT_ROWS; ++r) {
            const int t = min(t0 + wave + ATT_NW * r, a.Tp - 1);
#pragma unroll
            for (int k = 0; k < AI; ++k) {           // (only 1 - s^2 is needed here)
                float s_;
                las_s_load(a.s, ((long)b * a.Tp + t) * a.A + min(lane + 64 * k, a.A - 1), a.s16, s_, svr[r][k]);
            }
        }
    }
    // ... and so are this wave's enc rows (<= EV float4 per lane and row: E <= 256 EV on this path; the 6 x 1024 pBLSTM of
    // BASELINE configs[4] has E = 2048, and without this path its step was 48.7 us of dependent scalar loads)
    const bool vec = EB ? ((a.E & 7) == 0 && a.E <= 256 * EV) : ((a.E & 3) == 0 && a.E <= 256 * EV && ((((uintptr_t)a.enc) & 15) == 0));
    float4 er[EB ? 1 : ATT_ROWS][EB ? 1 : EV];
    u32x4 eb[EB ? ATT_ROWS : 1][EB ? EV / 2 : 1];
    if (vec) {
#pragma unroll
        for (int r = 0; r < ATT_ROWS; ++r) {
            const int t = min(t0 + wave + ATT_NW * r, t0 + tcv - 1);
            if constexpr (EB) {
                const u32x4* __restrict__ p = (const u32x4*)(a.enc16 + ((long)b * a.Tp + t) * a.E);
#pragma unroll
                for (int k = 0; k < EV / 2; ++k) eb[r][k] = p[min(lane + 64 * k, a.E / 8 - 1)];
            } else {
                const float4* __restrict__ p = (const float4*)(a.enc + ((long)b * a.Tp + t) * a.E);
#pragma unroll
                for (int k = 0; k < EV; ++k) er[r][k] = p[min(lane + 64 * k, a.E / 4 - 1)];
            }
        }
    }
    float att_r[ATT_ROWS];
#pragma unroll
    for (int r = 0; r < ATT_ROWS; ++r) att_r[r] = a.att[(long)b * a.Tp + min(t0 + wave + ATT_NW * r, t0 + tcv - 1)];
    // ---- staging + the softmax dot.  EVERY global load of this section is issued before the first LDS store: section
    // by section (load, wait, store) it was seven dependent memory round trips queued behind the row prefetch above --
    // 11-15 k of the kernel's ~42 k cycles (cycle stamps).  Loads are unconditional from clamped indices.
    float part = 0.f;
    const bool carry = LOC && a.df_next != nullptr;
    constexpr int NE = 2;                                            // E <= 1024
    constexpr int NWLP = (LOC_C * 64 * AI + ATT_NT - 1) / ATT_NT;    // A <= 64 AI
    constexpr int NCW = (LOC_C * LOC_W + ATT_NT - 1) / ATT_NT;
    constexpr int HU = (LOC_C * (20 + 2 * LOC_K) + ATT_NT - 1) / ATT_NT;       // chunks are <= 20 frames
    constexpr int NDOT = 6;                                          // first 6 * 512 elements of <df_next, f_next>
    float d_r[NE], c_r[NE], we_r = 0.f, wlp_r[LOC ? NWLP : 1], f_r = 0.f, cw_r[LOC ? NCW : 1], hv[LOC ? HU : 1];
    float dn_r[LOC ? NDOT : 1], fn_r[LOC ? NDOT : 1];
    const int tid = threadIdx.x;
    const float* __restrict__ dn = LOC && carry ? a.df_next + (long)b * LOC_C * a.Tp : nullptr;
    const float* __restrict__ fn = LOC && carry ? a.f_next + (long)b * LOC_C * a.Tp : nullptr;
    const int ndot = LOC_C * a.Tp;
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int i = min(tid + ATT_NT * u, a.E - 1);
        d_r[u] = a.dctx[(long)b * a.ld_dctx + i];
        c_r[u] = a.ctx[(long)b * a.ld_ctx + i];
    }
    if (LOC) {
        we_r = a.w_e[min(tid, a.A - 1)];
#pragma unroll
        for (int u = 0; u < NWLP; ++u) wlp_r[u] = a.w_lp[min(tid + ATT_NT * u, LOC_C * a.A - 1)];
        {
            const int i = min(tid, LOC_C * a.TC - 1), c = i / a.TC, tt = i - c * a.TC;
            f_r = a.f[(long)b * LOC_C * a.Tp + (long)c * a.Tp + min(t0 + tt, a.Tp - 1)];
        }
        if (carry) {
#pragma unroll
            for (int u = 0; u < NCW; ++u) cw_r[u] = a.conv_w[min(tid + ATT_NT * u, LOC_C * LOC_W - 1)];
#pragma unroll
            for (int u = 0; u < HU; ++u) {
                const int i = min(tid + ATT_NT * u, LOC_C * W - 1);
                const int c = i / W, j = i - c * W, t = t0 - LOC_K + j;
                const float v = dn[(long)c * a.Tp + min(max(t, 0), a.Tp - 1)];
                hv[u] = (t >= 0 && t < a.Tp) ? v : 0.f;
            }
#pragma unroll
            for (int u = 0; u < NDOT; ++u) {
                const int i = min(tid + ATT_NT * u, ndot - 1);
                dn_r[u] = dn[i]; fn_r[u] = fn[i];
            }
        }
    }
    // ---- now the LDS stores and the dot partials
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int i = tid + ATT_NT * u;
        if (i < a.E) { dctx_l[i] = d_r[u]; part += d_r[u] * c_r[u]; }
    }
    for (int i = tid + ATT_NT * NE; i < a.E; i += ATT_NT) {          // (E > 1024)
        const float d = a.dctx[(long)b * a.ld_dctx + i];
        dctx_l[i] = d;
        part += d * a.ctx[(long)b * a.ld_ctx + i];
    }
    if (LOC) {
        if (tid < a.A) we_l[tid] = we_r;
#pragma unroll
        for (int u = 0; u < NWLP; ++u) {
            const int i = tid + ATT_NT * u;
            if (i < LOC_C * a.A) { const int aa = i / LOC_C, c = i - aa * LOC_C; wlp_l[c * a.A + aa] = wlp_r[u]; }
        }
        if (tid < LOC_C * a.TC) { const int tt = tid % a.TC; f_l[tid] = (tt < tcv) ? f_r : 0.f; }
        if (carry) {
#pragma unroll
            for (int u = 0; u < NCW; ++u) {
                const int i = tid + ATT_NT * u;
                if (i < LOC_C * LOC_W) cw_l[i] = cw_r[u];
            }
#pragma unroll
            for (int u = 0; u < HU; ++u) {
                const int i = tid + ATT_NT * u;
                if (i < LOC_C * W) dfh_l[i] = hv[u];
            }
#pragma unroll
            for (int u = 0; u < NDOT; ++u)
                if (tid + ATT_NT * u < ndot) part += dn_r[u] * fn_r[u];
            for (int i = tid + ATT_NT * NDOT; i < ndot; i += ATT_NT) part += dn[i] * fn[i];     // (T' > 307)
        }
    }
    const float dot = block_sum(part, red);          // (its barriers also publish the LDS staging)
    // ---- d a, d e for this wave's frames
    float de_r[ATT_ROWS];
#pragma unroll
    for (int r = 0; r < ATT_ROWS; ++r) {
        const int tt = wave + ATT_NW * r, t = t0 + tt;
        de_r[r] = 0.f;
        if (tt >= a.TC || t >= t1) break;
        if (tt < tcv) {
            float acc = 0.f;
            const float* __restrict__ p = a.enc + ((long)b * a.Tp + t) * a.E;
            if (vec && EB) {
#pragma unroll
                for (int k = 0; k < EV / 2; ++k) {
                    const int i = lane + 64 * k;
                    if (i < a.E / 8) {
                        const u32x4 v = eb[r][k];
                        const float4 w0 = ((const float4*)dctx_l)[2 * i], w1 = ((const float4*)dctx_l)[2 * i + 1];
                        acc += __uint_as_float(v[0] << 16) * w0.x + __uint_as_float(v[0] & 0xffff0000u) * w0.y +
                               __uint_as_float(v[1] << 16) * w0.z + __uint_as_float(v[1] & 0xffff0000u) * w0.w +
                               __uint_as_float(v[2] << 16) * w1.x + __uint_as_float(v[2] & 0xffff0000u) * w1.y +
                               __uint_as_float(v[3] << 16) * w1.z + __uint_as_float(v[3] & 0xffff0000u) * w1.w;
                    }
                }
            } else if (vec) {
#pragma unroll
                for (int k = 0; k < EV; ++k) {
                    const int i = lane + 64 * k;
                    if (i < a.E / 4) {
                        const float4 v = er[EB ? 0 : r][EB ? 0 : k], w = ((const float4*)dctx_l)[i];
                        acc += v.x * w.x + v.y * w.y + v.z * w.z + v.w * w.w;
                    }
                }
            } else {
                for (int i = lane; i < a.E; i += 64) acc += p[i] * dctx_l[i];
            }
            if (carry) {
                // local index of df_next[c][t + K - k] is tt + 2K - k
                const float* __restrict__ dl = dfh_l + tt + 2 * LOC_K;
                float g0 = 0.f, g1 = 0.f;
#pragma unroll
                for (int c = 0; c < LOC_C; ++c) {
                    const float* __restrict__ cw = cw_l + c * LOC_W;
                    const float* __restrict__ dc = dl + c * W;
                    g0 += cw[lane] * dc[-lane] + cw[lane + 64] * dc[-(lane + 64)];
                    g1 += cw[lane + 128] * dc[-(lane + 128)];
                    if (lane < LOC_W - 192) g1 += cw[lane + 192] * dc[-(lane + 192)];
                }
                acc += g0 + g1;
            }
            acc = wave_sum(acc);
            de_r[r] = ATT_SCALE * att_r[r] * (acc - dot);
        }
        if (lane == 0) {
            a.de[(long)b * a.Tp + t] = de_r[r];
            if (!LOC) de_l[tt] = de_r[r];
        }
    }
    if (!LOC) {
        __syncthreads();
        // dq[a] = sum_t de[t] * psi[b,t,a]
        const int tv = t0 + tcv;
        for (int i = threadIdx.x; i < a.A; i += ATT_NT) {
            float acc0 = 0.f, acc1 = 0.f;
            const float* __restrict__ p = a.psi + ((long)b * a.Tp) * a.A + i;
            int t = t0;
            for (; t + 1 < tv; t += 2) {
                acc0 += de_l[t - t0] * p[(long)t * a.A];
                acc1 += de_l[t + 1 - t0] * p[(long)(t + 1) * a.A];
            }
            if (t < tv) acc0 += de_l[t - t0] * p[(long)t * a.A];
            const float qv = a.q[(long)b * a.A + i];
            atomicAdd(&a.dq_pre[(long)b * a.A + i], (acc0 + acc1) * (1.f - qv * qv));
        }
        return;
    }
    float dq_r[AI];
#pragma unroll
    for (int k = 0; k < AI; ++k) dq_r[k] = 0.f;
    float* __restrict__ dfg = a.df + (long)b * LOC_C * a.Tp;
#pragma unroll
    for (int r = 0; r < ATT_ROWS; ++r) {
        const int tt = wave + ATT_NW * r;
        if (tt >= tcv) break;
        const float de = de_r[r];
        float fc[LOC_C], dfc[LOC_C];
#pragma unroll
        for (int c = 0; c < LOC_C; ++c) { fc[c] = f_l[c * a.TC + tt]; dfc[c] = 0.f; }
#pragma unroll
        for (int k = 0; k < AI; ++k) {
            const int i = lane + 64 * k;
            if (i < a.A) {
                float u = 0.f;
#pragma unroll
                for (int c = 0; c < LOC_C; ++c) u += wlp_l[c * a.A + i] * fc[c];
                u = fast_tanh(u);
                const float dz = de * we_l[i] * svr[r][k];
                dq_r[k] += dz;
                const float du = dz * (1.f - u * u);
#pragma unroll
                for (int c = 0; c < LOC_C; ++c) dfc[c] += du * wlp_l[c * a.A + i];
            }
        }
#pragma unroll
        for (int c = 0; c < LOC_C; ++c) {
            const float v = wave_sum(dfc[c]);
            if (lane == 0) dfg[(long)c * a.Tp + t0 + tt] = v;
        }
    }
#pragma unroll
    for (int k = 0; k < AI; ++k) {
        const int i = lane + 64 * k;
        if (i < a.A) dq_l[wave * a.A + i] = dq_r[k];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.A; i += ATT_NT) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < ATT_NW; ++w) v += dq_l[w * a.A + i];
        const float qv = a.q[(long)b * a.A + i];
        atomicAdd(&a.dq_pre[(long)b * a.A + i], v * (1.f - qv * qv));
    }
}

struct LocPostArgs {
    int B, Tp, A, TC, L;
    const int32_t* lens;
    const float* de;          // [L][B][Tp]
    const float* f;           // [L][B][10][Tp]
    const void* s; int s16;   // [L][B][Tp][A]: fp32, or the 16-bit code of las_common.h (s16, bf16 mode)
    const float* w_lp; const float* w_e;
    float* dpsi;              // [B][Tp][A]   (caller-zeroed; rows < len written); null: the BPTT loop has summed d psi itself
    float* acc;               // [B][acc_stride]: d w_lp^T [10][A] | d w_e [A] | d b_e [1] | pad | d conv [10*201]  (+=)
    long acc_stride;
};

// grid (ceil(Tp / POST_TC), B), one wave per 64 attention dims (blockDim = 64 * ceil(A/64)): the sums over the L
// steps that are off the sequential chain, in one pass over the saved s:
//   d psi[b,t,:] = sum_l dz_l,  d w_e = sum de_l * s_l,  d b_e = sum de_l,  d W_lp = sum du_l (x) f_l
// with dz_l = de_l * w_e * (1 - s_l^2), du_l = dz_l * (1 - u_l^2), u_l = tanh(W_lp f_l) recomputed.
// A lane owns one attention dim for POST_TC frames, so its running sums are ~20 registers, >= 8 waves per SIMD hide
// the HBM latency of the s rows, and no cross-wave reduction is needed (lanes add their sums with coalesced atomics).
constexpr int POST_TC = 4;                // frames per lane (8: 907 vs 754 us -- the kernel is VALU-bound: ~55 instructions
                                          // per frame and step for the u = tanh(W_lp f) recompute, the two 10-term products and the broadcasts)
constexpr int POST_NAUX = ((LOC_C + 1) * POST_TC + 63) / 64;
__global__ __launch_bounds__(512) void att_loc_post(LocPostArgs a) {
    const int b = blockIdx.y, t0 = blockIdx.x * POST_TC, t1 = min(t0 + POST_TC, a.Tp), len = a.lens[b];
    const int tcv = min(t1, len) - t0;
    if (tcv <= 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = wave * 64 + lane, ic = min(i, a.A - 1);
    float wlp_r[LOC_C];
#pragma unroll
    for (int c = 0; c < LOC_C; ++c) wlp_r[c] = a.w_lp[ic * LOC_C + c];
    const float we_r = a.w_e[ic];
    float dps[POST_TC], dwlp_r[LOC_C], dwe_r = 0.f, dbe = 0.f;
#pragma unroll
    for (int r = 0; r < POST_TC; ++r) dps[r] = 0.f;
#pragma unroll
    for (int c = 0; c < LOC_C; ++c) dwlp_r[c] = 0.f;
    const long step_s = (long)a.B * a.Tp * a.A, step_f = (long)a.B * LOC_C * a.Tp, step_e = (long)a.B * a.Tp;
    // per-step side data {de, f[0..9]} of the POST_TC frames: ONE vector load per step, lane POST_TC * j + r holding
    // item j (0: de, 1..10: f[j-1]) of frame r, broadcast with v_readlane when used
    // (item index li = POST_TC * j + r lives in lane li % 64 of load li / 64)
    const float* auxp[POST_NAUX];
    long aux_step[POST_NAUX];
#pragma unroll
    for (int q = 0; q < POST_NAUX; ++q) {
        const int li = lane + 64 * q, j = min(li / POST_TC, LOC_C), ra = li % POST_TC;
        auxp[q] = (j == 0 ? a.de + (long)b * a.Tp : a.f + ((long)b * LOC_C + (j - 1)) * a.Tp) + min(t0 + ra, t0 + tcv - 1);
        aux_step[q] = j == 0 ? step_e : step_f;
    }
    auto load = [&](float (&sv)[POST_TC], float (&aux)[POST_NAUX], int l) {
#pragma unroll
        for (int r = 0; r < POST_TC; ++r) {
            const int t = min(t0 + r, t0 + tcv - 1);
            const long si = l * step_s + ((long)b * a.Tp + t) * a.A + ic;
            sv[r] = a.s16 ? __uint_as_float((unsigned)((const bf16_t*)a.s)[si]) : ((const float*)a.s)[si];      // (raw: decoded where used)
        }
#pragma unroll
        for (int q = 0; q < POST_NAUX; ++q) aux[q] = auxp[q][l * aux_step[q]];
    };
    auto item = [&](const float (&aux)[POST_NAUX], int li) {       // li is a compile-time constant at every call site
        return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, aux[li / 64]), li % 64));
    };
    auto compute = [&](const float (&sv)[POST_TC], const float (&aux)[POST_NAUX]) {
#pragma unroll
        for (int r = 0; r < POST_TC; ++r) {
            if (r >= tcv) break;
            const float de = item(aux, r);
            dbe += de;
            float fc[LOC_C], u = 0.f;
#pragma unroll
            for (int c = 0; c < LOC_C; ++c) {
                fc[c] = item(aux, (c + 1) * POST_TC + r);
                u += wlp_r[c] * fc[c];
            }
            u = fast_tanh(u);
            float s_ = sv[r], ds_ = 1.f - s_ * s_;
            if (a.s16) { const unsigned x = __float_as_uint(sv[r]); s_ = las_s16_s(x); ds_ = las_s16_ds(x); }
            const float dz = de * we_r * ds_;
            dps[r] += dz;
            dwe_r += de * s_;
            const float du = dz * (1.f - u * u);
#pragma unroll
            for (int c = 0; c < LOC_C; ++c) dwlp_r[c] += du * fc[c];
        }
    };
    float svA[POST_TC], svB[POST_TC], auxA[POST_NAUX], auxB[POST_NAUX];
    load(svA, auxA, 0);
    for (int l = 0; l < a.L; l += 2) {
        if (l + 1 < a.L) load(svB, auxB, l + 1);
        compute(svA, auxA);
        if (l + 1 < a.L) {
            if (l + 2 < a.L) load(svA, auxA, l + 2);
            compute(svB, auxB);
        }
    }
    if (i < a.A) {
#pragma unroll
        for (int r = 0; r < POST_TC; ++r)
            if (r < tcv && a.dpsi) a.dpsi[((long)b * a.Tp + t0 + r) * a.A + i] = dps[r];
        float* accg = a.acc + (long)b * a.acc_stride;
        atomicAdd(&accg[a.A * LOC_C + i], dwe_r);
#pragma unroll
        for (int c = 0; c < LOC_C; ++c) atomicAdd(&accg[c * a.A + i], dwlp_r[c]);      // [c][a]: contiguous per wave
        if (i == 0) atomicAdd(&accg[a.A * LOC_C + a.A], dbe);
    }
}

// The same sums on the matrix cores (bf16 mode).  Workgroup = (utterance b, 16 frames), wave w owns the attention-dim tiles
// j = w + NW*i (16 dims each): per step and tile
//   u   = F [16 frames x 10 channels] * W_lp^T      one 16x16x16 MFMA; a lane gets rows (frames) 4g..4g+3 of column a
//   dz  = de[t] * w_e[a] * (1 - s^2),  d psi += dz,  d w_e += de[t] * s,  du = dz * (1 - tanh(u)^2)     (~12 VALU per element
//         instead of ~46: no broadcasts, no 10-term products)
//   d W_lp^T [10 x 16] += F^T [10 x 16 frames] * du   one MFMA: du in its OUTPUT layout (column a, four consecutive frames)
//         is exactly the B operand of a product that contracts over the frames -- no trip through LDS.
// The saved s is read once, as 64-byte row segments, two steps in flight per wave; nothing is shared between waves (no LDS,
// no barrier).  Frames beyond the utterance are masked by SELECTS (s, f there are undefined).
typedef __attribute__((ext_vector_type(4))) short bf16x4;
template <int NT>
__global__ __launch_bounds__(512) void att_loc_post_mma(LocPostArgs a, int NW) {
    const int b = blockIdx.y, t0 = blockIdx.x * 16, len = a.lens[b];
    if (t0 >= len) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), fr = lane & 15, g = lane >> 4;
    const int A = a.A, Tp = a.Tp, ntiles = (A + 15) / 16;
    bool okc[NT];
    int corr[NT];                        // byte correction that keeps a column beyond A (last tile) inside its row
    float we_r[NT];
    bf16x4 wlpB[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int acol = (wave + NW * i) * 16 + fr;
        okc[i] = acol < A;
        const int ac = min(acol, A - 1);
        corr[i] = (min(acol & ~1, A - 2) - (acol & ~1)) * 2;      // (of my column PAIR: A is even on this path)
        we_r[i] = okc[i] ? a.w_e[ac] : 0.f;
        float w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * g + r;
            w[r] = (okc[i] && c < LOC_C) ? a.w_lp[ac * LOC_C + min(c, LOC_C - 1)] : 0.f;
        }
        const unsigned lo = pack_bf16x2(w[0], w[1]), hi = pack_bf16x2(w[2], w[3]);
        wlpB[i] = bf16x4{(short)(lo & 0xffff), (short)(lo >> 16), (short)(hi & 0xffff), (short)(hi >> 16)};
    }
    bool tv[4];
    int voff_s[2], off_fu[4];                           // byte offsets inside the step's slab of utterance b
    // s (the 16-bit code) is read as 32-bit words = column pairs: an even lane takes the pair (fr, fr + 1) of rows 0 and 2, its odd
    // neighbour the same pair of rows 1 and 3, and the two swap over DPP -- 2 loads per tile instead of 4 two-byte ones
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int tc = min(t0 + 4 * g + 2 * k + (fr & 1), len - 1);
        voff_s[k] = (tc * A + wave * 16 + (fr & ~1)) * 2;      // tile i: + i * NW * 32 bytes (wave-uniform: the scalar offset)
    }
    bool v_fu[4], v_fw[4];
    // de and the d W_lp operand are four CONSECUTIVE frames: one 16-byte load each (frames beyond the utterance are
    // masked, beyond the slab the buffer returns zero)
    const int off_e = (t0 + 4 * g) * 4, off_fw = (min(fr, LOC_C - 1) * Tp + t0 + 4 * g) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = t0 + 4 * g + r;
        tv[r] = t < len;
        // u operand: lane (m = frame t0 + fr, k = channel 4g + r)
        const int c = 4 * g + r, tf = t0 + fr;
        v_fu[r] = c < LOC_C && tf < len;
        off_fu[r] = (min(c, LOC_C - 1) * Tp + min(tf, len - 1)) * 4;
        // d W_lp operand: lane (m = channel fr, k = frame t0 + 4g + r)
        v_fw[r] = fr < LOC_C && tv[r];
    }
    const long step_s = (long)a.B * Tp * A, step_f = (long)a.B * LOC_C * Tp, step_e = (long)a.B * Tp;
    const bf16_t* __restrict__ sb = (const bf16_t*)a.s + (long)b * Tp * A;
    const float* __restrict__ fb = a.f + (long)b * LOC_C * Tp;
    const float* __restrict__ eb = a.de + (long)b * Tp;
    const int nt_w = (ntiles - wave + NW - 1) / NW;           // tiles this wave really has (wave-uniform)

    // One buffer resource per step and tensor (scalar work), 32-bit lane offsets: the 4*NT loads of a step share four
    // offset registers.  Every s element a lane reads is a DEFINED one (rows clamped to the utterance's last frame, columns
    // to A - 1), so s needs no select: the masked de / w_e make its terms exact zeros.
    struct Set { float s[NT][2]; float de[4], fu[4], fw[4]; };
    auto load = [&](Set& q, int l) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(sb + l * step_s), 0, Tp * A * 2, 0x00020000);
        __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void*)(fb + l * step_f), 0, LOC_C * Tp * 4, 0x00020000);
        __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void*)(eb + l * step_e), 0, Tp * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < NT; ++i)
            if (i < nt_w) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    q.s[i][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff_s[k] + corr[i], i * NW * 32, 0));      // (raw codes of a column pair)
            }
        const u32x4 ve = __builtin_amdgcn_raw_buffer_load_b128(re, off_e, 0, 0);
        const u32x4 vw = __builtin_amdgcn_raw_buffer_load_b128(rf, off_fw, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            q.de[r] = __uint_as_float(ve[r]);
            q.fw[r] = __uint_as_float(vw[r]);
            q.fu[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rf, off_fu[r], 0, 0));
        }
    };
    float dps[NT][4], dwe[NT], dbe = 0.f;
    f32x4 dW[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        dwe[i] = 0.f;
        dW[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) dps[i][r] = 0.f;
    }
    auto pack4 = [](float x0, float x1, float x2, float x3) {
        const unsigned lo = pack_bf16x2(x0, x1), hi = pack_bf16x2(x2, x3);
        return bf16x4{(short)(lo & 0xffff), (short)(lo >> 16), (short)(hi & 0xffff), (short)(hi >> 16)};
    };
    auto compute = [&](const Set& q) {
        float de[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { de[r] = tv[r] ? q.de[r] : 0.f; dbe += de[r]; }
        const bf16x4 fu = pack4(v_fu[0] ? q.fu[0] : 0.f, v_fu[1] ? q.fu[1] : 0.f, v_fu[2] ? q.fu[2] : 0.f, v_fu[3] ? q.fu[3] : 0.f);
        const bf16x4 fw = pack4(v_fw[0] ? q.fw[0] : 0.f, v_fw[1] ? q.fw[1] : 0.f, v_fw[2] ? q.fw[2] : 0.f, v_fw[3] ? q.fw[3] : 0.f);
#pragma unroll
        for (int i = 0; i < NT; ++i)
            if (i < nt_w) {
                const f32x4 u = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fu, wlpB[i], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                float du[4];
                // my words (rows 0, 2 | 1, 3) and my neighbour's (rows 1, 3 | 0, 2); my column is the low (even lane) or high half
                const unsigned w0 = __float_as_uint(q.s[i][0]), w1 = __float_as_uint(q.s[i][1]);
                const unsigned p0 = __float_as_uint(las_dpp<0xB1, 0xf>(0.f, q.s[i][0])), p1 = __float_as_uint(las_dpp<0xB1, 0xf>(0.f, q.s[i][1]));
                const bool odd = fr & 1;
                const unsigned rw[4] = {odd ? p0 : w0, odd ? w0 : p0, odd ? p1 : w1, odd ? w1 : p1};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned sx = odd ? rw[r] >> 16 : rw[r] & 0xffffu;
                    const float s_ = las_s16_s(sx);
                    const float th = fast_tanh(u[r]);          // (rows beyond the utterance: F = 0 there, u = 0)
                    const float dz = de[r] * we_r[i] * las_s16_ds(sx);
                    dps[i][r] += dz;
                    dwe[i] = fmaf(de[r], s_, dwe[i]);
                    du[r] = dz * (1.f - th * th);
                }
                dW[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fw, pack4(du[0], du[1], du[2], du[3]), dW[i], 0, 0, 0);
            }
    };
    // three register sets: a step's rows are requested two steps before they are used (one step ahead left every step
    // waiting out most of the HBM latency: 497 us; a step's arithmetic is ~0.6 us)
    // Steps behind the utterance's last label carry no gradient (d e = 0 exactly: the loss ignores them and nothing flows
    // back from later steps), and labels are ragged: find the last step with a nonzero d e in this workgroup's frames
    // (lane l checks steps l, l + 64, ...) and stop there.
    int lmax = -1;
    for (int l = lane; l < a.L; l += 64) {
        const float* __restrict__ el = eb + l * step_e + t0;
        bool nz = false;
#pragma unroll
        for (int k = 0; k < 16; ++k) nz |= (t0 + k < len) && (el[min(k, len - 1 - t0)] != 0.f);
        if (nz) lmax = l;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) lmax = max(lmax, __shfl_xor(lmax, m));
    // gridDim.z workgroups share the steps of one (utterance, frame tile) in contiguous runs (d psi is ADDED to the
    // caller-zeroed buffer): the ragged step counts would otherwise leave the kernel waiting for its longest workgroups
    const int Lall = __builtin_amdgcn_readfirstlane(lmax + 1);
    const int per = (Lall + (int)gridDim.z - 1) / (int)gridDim.z, lb = (int)blockIdx.z * per, L = min(Lall, lb + per);
    if (lb >= L) return;
    Set q0, q1, q2;
    load(q0, lb);
    if (lb + 1 < L) load(q1, lb + 1);
    for (int l = lb; l < L; l += 3) {
        if (l + 2 < L) load(q2, l + 2);
        compute(q0);
        if (l + 1 < L) {
            if (l + 3 < L) load(q0, l + 3);
            compute(q1);
        }
        if (l + 2 < L) {
            if (l + 4 < L) load(q1, l + 4);
            compute(q2);
        }
    }
    float* accg = a.acc + (long)b * a.acc_stride;
#pragma unroll
    for (int i = 0; i < NT; ++i)
        if (i < nt_w) {
            const int acol = (wave + NW * i) * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (okc[i] && tv[r] && a.dpsi) atomicAdd(&a.dpsi[((long)b * Tp + t0 + 4 * g + r) * A + acol], dps[i][r]);
            float v = dwe[i];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (g == 0 && okc[i]) atomicAdd(&accg[A * LOC_C + acol], v);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * g + r;
                if (okc[i] && c < LOC_C) atomicAdd(&accg[c * A + acol], dW[i][r]);
            }
        }
    if (wave == 0) {                       // every lane of a 16-lane row holds the same four frames
        float v = dbe;
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane == 0) atomicAdd(&accg[A * LOC_C + A], v);
    }
}

// d psi alone, bf16 mode behind the persistent loops (coded s, even A): the one post-loop sum the encoder's backward waits for
// needs only s, d e and w_e -- d psi[b,t,a] = w_e[a] * sum_l d e_l[b,t] (1 - s_l[b,t,a]^2) -- so it is a pure stream over the saved
// s (2 bytes per element) on the caller's main stream, and att_loc_post_mma's MFMA work (d W_lp, d w_e, d b_e: parameter
// gradients) moves to the parameter sums beside it.  Workgroup = (utterance, DPS_FT frames): per step its s values are ONE
// contiguous run of DPS_FT * A codes, read as 32-bit words (column pairs), K words per thread; plain stores (a thread owns its
// elements), steps behind the block's last nonzero d e are skipped (ragged label lengths).
constexpr int DPS_FT = 8, DPS_NT = 256;
template <int K>
__global__ __launch_bounds__(DPS_NT) void att_dpsi_kernel(LocPostArgs a) {
    __shared__ int lmax_s;
    const int b = blockIdx.y, t0 = blockIdx.x * DPS_FT, len = a.lens[b];
    const int A = a.A, Tp = a.Tp, nf = max(0, min(DPS_FT, len - t0)), nw = nf * A / 2, tid = threadIdx.x;
    {                                                    // rows beyond the utterance: zeros (the buffer is not pre-filled)
        float* __restrict__ z = a.dpsi + ((long)b * Tp + t0 + nf) * A;
        const int nz = (min(DPS_FT, Tp - t0) - nf) * A;
        for (int i = tid; i < nz; i += DPS_NT) z[i] = 0.f;
    }
    if (nf == 0) return;
    const long step_e = (long)a.B * Tp, step_w = (long)a.B * Tp * A / 2;
    const float* __restrict__ dep = a.de + (long)b * Tp + t0;
    if (tid == 0) lmax_s = -1;
    __syncthreads();
    {
        int lm = -1;
        for (int l = tid; l < a.L; l += DPS_NT) {
            bool nz = false;
            for (int f = 0; f < nf; ++f) nz |= dep[l * step_e + f] != 0.f;
            if (nz) lm = l;
        }
        if (lm >= 0) atomicMax(&lmax_s, lm);
    }
    __syncthreads();
    const int L = lmax_s + 1;
    const unsigned* __restrict__ sp = (const unsigned*)((const bf16_t*)a.s + ((long)b * Tp + t0) * A);
    float we0[K], we1[K], acc0[K], acc1[K];
    int fo[K], wo[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int w = tid + DPS_NT * k, wc = min(w, nw - 1), e = 2 * wc, f = e / A, c = e - f * A;      // (A even: a pair never straddles two rows)
        const bool ok = w < nw;
        wo[k] = wc; fo[k] = f;
        we0[k] = ok ? a.w_e[c] : 0.f; we1[k] = ok ? a.w_e[c + 1] : 0.f;
        acc0[k] = 0.f; acc1[k] = 0.f;
    }
#pragma unroll 2
    for (int l = 0; l < L; ++l) {
        unsigned x[K];
        float de[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { x[k] = __builtin_nontemporal_load(sp + l * step_w + wo[k]); de[k] = dep[l * step_e + fo[k]]; }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            acc0[k] = fmaf(de[k], las_s16_ds(x[k] & 0xffffu), acc0[k]);
            acc1[k] = fmaf(de[k], las_s16_ds(x[k] >> 16), acc1[k]);
        }
    }
    float* __restrict__ out = a.dpsi + ((long)b * Tp + t0) * A;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int w = tid + DPS_NT * k;
        if (w < nw) *(float2*)(out + 2 * w) = make_float2(acc0[k] * we0[k], acc1[k] * we1[k]);
    }
}

// d conv_w[c][k] += sum over this block's (step, utterance) pairs of  sum_t df[c][t] * prev[t + k - K].
// Thread = (c, 10 consecutive k): a sliding register window over prev gives 10 FMAs per two LDS reads.
constexpr int CW_KPT = 10, CW_GROUPS = (LOC_W + CW_KPT - 1) / CW_KPT;       // 21 groups x 10 channels = 210 threads
__global__ __launch_bounds__(256) void att_conv_wgrad(int B, int Tp, int L, int pairs_per_block,
                                                      const int32_t* __restrict__ lens, const float* __restrict__ att,
                                                      const float* __restrict__ df, float* __restrict__ acc,
                                                      long acc_stride, long conv_off) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* pl = sm;                                  // [Tp + 2K + CW_KPT] prev[i - K], zero outside
    float* dl = pl + Tp + 2 * LOC_K + CW_KPT;        // [10][Tp]
    const int c = threadIdx.x / CW_GROUPS, k0 = (threadIdx.x % CW_GROUPS) * CW_KPT;
    const bool active = threadIdx.x < LOC_C * CW_GROUPS;
    float accr[CW_KPT];
#pragma unroll
    for (int j = 0; j < CW_KPT; ++j) accr[j] = 0.f;
    const long total = (long)L * B;
    const long p0 = (long)blockIdx.x * pairs_per_block;
    int last_b = -1;
    for (long p = p0; p < min(total, p0 + pairs_per_block); ++p) {
        const int b = (int)(p / L), l = (int)(p % L), len = lens[b];          // utterance-major: one flush per utterance
        __syncthreads();
        const float* __restrict__ pr = att + ((long)l * B + b) * Tp;            // attention fed to step l's conv
        for (int i = threadIdx.x; i < Tp + 2 * LOC_K + CW_KPT; i += 256) {
            const int t = i - LOC_K;
            const float v = pr[min(max(t, 0), Tp - 1)];
            pl[i] = (t >= 0 && t < Tp) ? v : 0.f;
        }
        const float* __restrict__ dp = df + ((long)l * B + b) * LOC_C * Tp;
        for (int i = threadIdx.x; i < LOC_C * Tp; i += 256) dl[i] = dp[i];
        __syncthreads();
        if (active) {
            if (last_b >= 0 && last_b != b) {        // flush the finished utterance's partial sums
                float* o = acc + (long)last_b * acc_stride + conv_off + c * LOC_W + k0;
#pragma unroll
                for (int j = 0; j < CW_KPT; ++j)
                    if (k0 + j < LOC_W) { atomicAdd(o + j, accr[j]); accr[j] = 0.f; }
            }
            float win[CW_KPT];
#pragma unroll
            for (int j = 0; j < CW_KPT; ++j) win[j] = pl[k0 + j];
            const float* __restrict__ dr = dl + c * Tp;
            for (int t = 0; t < len; ++t) {
                const float d = dr[t];
#pragma unroll
                for (int j = 0; j < CW_KPT; ++j) accr[j] += d * win[j];
#pragma unroll
                for (int j = 0; j + 1 < CW_KPT; ++j) win[j] = win[j + 1];
                win[CW_KPT - 1] = pl[t + 1 + k0 + CW_KPT - 1];
            }
        }
        last_b = b;
    }
    if (active && last_b >= 0) {
        float* o = acc + (long)last_b * acc_stride + conv_off + c * LOC_W + k0;
#pragma unroll
        for (int j = 0; j < CW_KPT; ++j)
            if (k0 + j < LOC_W) atomicAdd(o + j, accr[j]);
    }
}

// demb[tok[r]][:] += dx[r][0:C]
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int32_t* __restrict__ tok, const float* __restrict__ dx,
                                                        long ldx, int C, float* __restrict__ demb) {
    const int r = blockIdx.x, tk = tok[r];
    for (int i = threadIdx.x; i < C; i += 256) atomicAdd(&demb[(long)tk * C + i], dx[(long)r * ldx + i]);
}

// T' is cut into chunks of <= 20 frames: a wave of the energy kernels then owns <= 5 rows (ATT_ROWS) and can request
// all of them up front.  Must be identical in decoder.hip and decoder_bwd.hip.
int att_chunks(int Tp) { const int n = (Tp + 19) / 20; return n < 1 ? 1 : n; }

}  // namespace

extern "C" int64_t las_decoder_loc_acc_floats(int A) { return ((A * LOC_C + A + 1 + 3) / 4) * 4 + LOC_C * LOC_W; }
extern "C" int las_decoder_att_chunks(int Tp) { return att_chunks(Tp); }
extern "C" size_t las_decoder_s_elem_bytes(int prec) { return prec == LAS_PREC_BF16 ? 2 : 4; }
extern "C" size_t las_decoder_pk_bwd_workspace_bytes(const las_dec_dims* d) { return las_dec_pk_bwd_ws_bytes(d); }

static int decoder_bwd_run(const las_dec_dims* d, const las_dec_params* p, const float* enc, const float* psi,
                           const int32_t* enc_len, const las_dec_state* st_, const float* g_htop,
                           las_dec_bwd_state* bw_, int parts, void* stream);

extern "C" int las_decoder_bwd(const las_dec_dims* d, const las_dec_params* p, const float* enc, const float* psi,
                               const int32_t* enc_len, const las_dec_state* st_, const float* g_htop,
                               las_dec_bwd_state* bw_, void* stream) {
    LAS_CHECK_ARG(d && p && enc && psi && enc_len && st_ && g_htop && bw_);
    return decoder_bwd_run(d, p, enc, psi, enc_len, st_, g_htop, bw_, LAS_DEC_BWD_CHAIN | LAS_DEC_BWD_PARAM_SUMS, stream);
}

extern "C" int las_decoder_bwd_parts(const las_dec_dims* d, const las_dec_params* p, const float* enc, const float* psi,
                                     const int32_t* enc_len, const las_dec_state* st_, const float* g_htop,
                                     las_dec_bwd_state* bw_, int parts, void* stream) {
    LAS_CHECK_ARG(d && p && enc && psi && enc_len && st_ && g_htop && bw_);
    LAS_CHECK_ARG(parts > 0 && !(parts & ~(LAS_DEC_BWD_CHAIN | LAS_DEC_BWD_PARAM_SUMS)));
    return decoder_bwd_run(d, p, enc, psi, enc_len, st_, g_htop, bw_, parts, stream);
}

static int decoder_bwd_run(const las_dec_dims* d, const las_dec_params* p, const float* enc, const float* psi,
                           const int32_t* enc_len, const las_dec_state* st_, const float* g_htop,
                           las_dec_bwd_state* bw_, int parts, void* stream) {
    LAS_CHECK_ARG(d && p && enc && psi && enc_len && st_ && g_htop && bw_);
    const int B = d->B, Tp = d->Tp, E = d->E, A = d->A, C = d->C, NL = d->NL, L = d->L, loc = d->loc, prec = d->prec;
    LAS_CHECK_ARG(B > 0 && Tp > 0 && E > 0 && A > 0 && C > 0 && NL >= 1 && NL <= 4 && L >= 0);
    if (L == 0) return LAS_OK;
    const las_dec_state& s = *st_;
    las_dec_bwd_state& w = *bw_;
    hipStream_t st = (hipStream_t)stream;
    const long XI = C + E, BC = (long)B * C;
    const int NCH = att_chunks(Tp), TC = (Tp + NCH - 1) / NCH;
    const long acc_stride = las_decoder_loc_acc_floats(A);
    const int AI = (A + 63) / 64;
    if (loc && AI > 8) return LAS_E_UNSUPPORTED;
    const bool chain = parts & LAS_DEC_BWD_CHAIN, sums = parts & LAS_DEC_BWD_PARAM_SUMS;
    if (loc) LAS_CHECK_ARG(w.df && w.de && w.dpsi && w.acc);
    // one persistent launch for the whole sequential chain when the shape / mode allows it and the caller gave the workspace
    const bool pk = w.pk_ws && w.pk_status && las_dec_pk_bwd_ws_bytes(d) > 0;
    // behind the persistent loops in bf16 mode d psi is a stream of its own and the MFMA post-loop pass is left with parameter gradients (below)
    const bool split_post = loc && pk && prec == LAS_PREC_BF16 && (A & 1) == 0 && A <= 512 && !las_fallback("LAS_LOC_POST_VALU");
    if (chain) {
        if (!pk) {       // (the persistent loop keeps the carries in registers and writes every element of d q_pre itself)
            LAS_HIP(hipMemsetAsync(w.dh_carry, 0, sizeof(float) * NL * BC, st));
            LAS_HIP(hipMemsetAsync(w.dc_carry, 0, sizeof(float) * NL * BC, st));
            LAS_HIP(hipMemsetAsync(w.dq_pre, 0, sizeof(float) * (size_t)L * B * A, st));
        }
        if (loc) {
            // d f: the per-step chain reads the next step's rows over all T' frames (zeros beyond the utterance); the persistent loop
            // exchanges its window through a buffer of its own and att_conv_wgrad stops at the utterance's length: no 57 MB fill at c3.
            // d psi: att_dpsi_kernel writes every row (zeros beyond the utterance); the other post-loop kernels ADD to it.
            if (!pk) LAS_HIP(hipMemsetAsync(w.df, 0, sizeof(float) * (size_t)L * B * LOC_C * Tp, st));
            if (!split_post) LAS_HIP(hipMemsetAsync(w.dpsi, 0, sizeof(float) * (size_t)B * Tp * A, st));
            LAS_HIP(hipMemsetAsync(w.acc, 0, sizeof(float) * (size_t)B * acc_stride, st));
        }
    }
    // [E] dctx | [TC] de | loc: [A] w_e | [10][A] w_lp | [10][TC] f | [8][A] dq partials | [10][201] conv_w | [10][TC+200] df halo
    size_t lds_e = sizeof(float) * (((size_t)E + 3) / 4 * 4 + TC);
    if (loc) lds_e += sizeof(float) * (A + LOC_C * (size_t)A + LOC_C * TC + ATT_NW * (size_t)A + LOC_C * LOC_W + LOC_C * (TC + 2 * LOC_K));
    const size_t lds_cw = sizeof(float) * ((size_t)Tp + 2 * LOC_K + CW_KPT + LOC_C * (size_t)Tp);
    if (lds_e > 160 * 1024 || (loc && lds_cw > 160 * 1024)) return LAS_E_UNSUPPORTED;
    const bool fuse_pw = NL == 1;
    const bool use_e16 = w.enc_bf16 && prec == LAS_PREC_BF16 && (E & 7) == 0 && E <= 2048 && ((((uintptr_t)w.enc_bf16) & 15) == 0);
    const bool drop = d->dropout > 0.f;
    if (pk && chain) {
        int rc = las_dec_pk_bwd(d, p, enc, psi, enc_len, st_, g_htop, bw_, st);
        if (rc) return rc;
    }
    for (int t = L - 1; t >= 0 && !pk && chain; --t) {
        // ---- LSTM cells, top layer first
        for (int l = NL - 1; l >= 0; --l) {
            const float* dh_ext = (l == NL - 1) ? g_htop + (long)t * BC : w.d_below;
            float* dg = w.dgates + ((long)l * L + t) * B * 4 * C;
            // single-layer decoders: the cell backward of step t < L-1 already ran in the epilogue of step t+1's
            // recurrent product (below), which is where its dh comes from
            if (!(fuse_pw && t < L - 1)) {
                hipLaunchKernelGGL(cell_pw_bwd, dim3((B * C + 255) / 256), dim3(256), 0, st, B, C, dh_ext, (long)C,
                                   w.dh_carry + (long)l * BC, w.dc_carry + (long)l * BC,
                                   s.gates + ((long)l * L + t) * B * 4 * C, s.cs + ((long)l * (L + 1) + t + 1) * BC,
                                   s.cs + ((long)l * (L + 1) + t) * BC, dg);
                LAS_LAUNCH_OK();
            }
            const int Kx = l == 0 ? (int)XI : C;
            float* dx = l == 0 ? w.dxin + (long)t * B * XI : w.d_below;
            int rc = las_skinny_launch_pk(prec, dg, 4 * C, p->w_ihT[l], 4 * C, 4 * C, nullptr, 0, nullptr, 0, 0, nullptr, 0, nullptr,
                                          0, 0, B, Kx, nullptr, nullptr, 0, dx, Kx, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr,
                                          p->pk_dx[l], st);
            if (rc) return rc;
            if (drop && l == 0) {                      // d xin = d(xdrop) * mask of the forward pass
                rc = las_dropout_rows(dx, Kx, dx, Kx, B, Kx, d->dropout, las_decoder_drop_seed(d->drop_seed, t, 0), stream);
                if (rc) return rc;
            }
            // recurrent carry dh_{l,t-1} = dgates * W_hh; layer 0 gets the attention-query path in the same product
            // (second k-segment dq_pre_t * W_phi) and is therefore launched after the attention backward below
            if (l > 0 && t > 0) {
                rc = las_skinny_launch_pk(prec, dg, 4 * C, p->w_hhT[l], 4 * C, 4 * C, nullptr, 0, nullptr, 0, 0, nullptr, 0, nullptr, 0,
                                          0, B, C, nullptr, nullptr, 0, w.dh_carry + (long)l * BC, C, 0, 0, nullptr, nullptr, nullptr,
                                          nullptr, nullptr, p->pk_dh[l], st);
                if (rc) return rc;
                if (drop) {                            // layer l >= 1 saw dropout(h_{l,t-1}) as its recurrent state
                    rc = las_dropout_rows(w.dh_carry + (long)l * BC, C, w.dh_carry + (long)l * BC, C, B, C, d->dropout,
                                          las_decoder_drop_seed(d->drop_seed, t, l), stream);
                    if (rc) return rc;
                }
            }
        }
        // ---- attention of step t
        AttBwdArgs a{};
        a.B = B; a.Tp = Tp; a.E = E; a.A = A; a.TC = TC; a.NCH = NCH;
        a.enc = enc; a.psi = psi; a.lens = enc_len;
        a.enc16 = use_e16 ? (const bf16_t*)w.enc_bf16 : nullptr;
        a.att = s.att + (long)(t + 1) * B * Tp;
        a.dctx = w.dxin + (long)t * B * XI + C; a.ld_dctx = XI;
        a.ctx = s.xin + (long)t * B * XI + C; a.ld_ctx = XI;
        a.q = s.q + (long)t * B * A;
        a.dq_pre = w.dq_pre + (long)t * B * A;
        a.de = w.de + (long)t * B * Tp;
        if (loc) {
            a.f = s.f + (long)t * B * LOC_C * Tp;
            a.s16 = prec == LAS_PREC_BF16;
            a.s = (const char*)s.s + (size_t)t * B * Tp * A * (a.s16 ? 2 : 4);
            a.w_lp = p->w_lp; a.w_e = p->w_e; a.conv_w = p->conv_w;
            a.df = w.df + (long)t * B * LOC_C * Tp;
            if (t + 1 < L) {
                a.df_next = w.df + (long)(t + 1) * B * LOC_C * Tp;
                a.f_next = s.f + (long)(t + 1) * B * LOC_C * Tp;
            }
        }
        if (!loc) {
            if (use_e16 && E > 1024) hipLaunchKernelGGL((att_bwd_step<false, 1, 8, true>), dim3(NCH, B), dim3(ATT_NT), lds_e, st, a);
            else if (use_e16) hipLaunchKernelGGL((att_bwd_step<false, 1, 4, true>), dim3(NCH, B), dim3(ATT_NT), lds_e, st, a);
            else if (E > 1024) hipLaunchKernelGGL((att_bwd_step<false, 1, 8>), dim3(NCH, B), dim3(ATT_NT), lds_e, st, a);
            else hipLaunchKernelGGL((att_bwd_step<false, 1, 4>), dim3(NCH, B), dim3(ATT_NT), lds_e, st, a);
        } else {
#define LAS_ATT_GO(AIV)                                                                                           \
    {                                                                                                             \
        auto k = use_e16 ? (E > 1024 ? att_bwd_step<true, AIV, 8, true> : att_bwd_step<true, AIV, 4, true>)       \
                         : (E > 1024 ? att_bwd_step<true, AIV, 8> : att_bwd_step<true, AIV, 4>);                  \
        if (lds_e > 64 * 1024 && t == L - 1) LAS_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_e)); \
        hipLaunchKernelGGL(k, dim3(NCH, B), dim3(ATT_NT), lds_e, st, a);                                          \
    }
            if (AI <= 1) LAS_ATT_GO(1) else if (AI <= 2) LAS_ATT_GO(2) else if (AI <= 4) LAS_ATT_GO(4)
            else if (AI <= 5) LAS_ATT_GO(5) else LAS_ATT_GO(8)
#undef LAS_ATT_GO
        }
        LAS_LAUNCH_OK();
        // ---- dh0_{t-1} = dgates_{0,t} * W_hh + dq_pre_t * W_phi   (one two-segment product)
        if (t > 0) {
            const float* dg0 = w.dgates + (long)t * B * 4 * C;
            las_skinny_pw pw{};
            if (fuse_pw) {                                // cell backward of step t-1 rides on this product
                pw.dh_ext = g_htop + (long)(t - 1) * BC; pw.ld_ext = C; pw.dc_carry = w.dc_carry;
                pw.gates = s.gates + (long)(t - 1) * B * 4 * C; pw.c_t = s.cs + (long)t * BC; pw.c_prev = s.cs + (long)(t - 1) * BC;
                pw.dgates = w.dgates + (long)(t - 1) * B * 4 * C;
            }
            int rc = las_skinny_launch_pk(prec, dg0, 4 * C, p->w_hhT[0], 4 * C, 4 * C, w.dq_pre + (long)t * B * A, A, p->w_phiT, A, A,
                                          nullptr, 0, nullptr, 0, 0, B, C, nullptr, nullptr, 0, w.dh_carry, C, 0, 0, nullptr, nullptr,
                                          nullptr, nullptr, fuse_pw ? &pw : nullptr, p->pk_dh[0], st);
            if (rc) return rc;
        }
    }
    // ---- sums over the L steps that are off the sequential chain.  d psi is the one the encoder's backward waits for; behind the
    // persistent loops in bf16 mode (coded s, even A) it is a stream of its own on the caller's main stream (att_dpsi_kernel) and
    // the MFMA pass over s / f / d e is left with parameter gradients, which run with the other parameter sums; otherwise one pass
    // makes both.  (Summing d psi inside the persistent loop instead was measured at c3: as float atomics at the L2 the loop grew by
    // 5 us a step, as a read-modify-write pass at the end of every step by 9 us -- the attention role has no slack.)
    if (split_post && chain) {
        LocPostArgs q{};
        q.B = B; q.Tp = Tp; q.A = A; q.TC = TC; q.L = L; q.lens = enc_len;
        q.de = w.de; q.s = s.s; q.s16 = 1; q.w_e = p->w_e; q.dpsi = w.dpsi;
        const dim3 grid((Tp + DPS_FT - 1) / DPS_FT, B);
        const int K = (DPS_FT * A / 2 + DPS_NT - 1) / DPS_NT;
        if (K <= 2) hipLaunchKernelGGL(att_dpsi_kernel<2>, grid, dim3(DPS_NT), 0, st, q);
        else if (K <= 4) hipLaunchKernelGGL(att_dpsi_kernel<4>, grid, dim3(DPS_NT), 0, st, q);
        else if (K <= 6) hipLaunchKernelGGL(att_dpsi_kernel<6>, grid, dim3(DPS_NT), 0, st, q);
        else hipLaunchKernelGGL(att_dpsi_kernel<8>, grid, dim3(DPS_NT), 0, st, q);
        LAS_LAUNCH_OK();
    }
    if (loc && (split_post ? sums : chain)) {
        LocPostArgs q{};
        q.B = B; q.Tp = Tp; q.A = A; q.TC = TC; q.L = L; q.lens = enc_len;
        q.de = w.de; q.f = s.f; q.s = s.s; q.w_lp = p->w_lp; q.w_e = p->w_e;
        q.dpsi = split_post ? nullptr : w.dpsi; q.acc = w.acc; q.acc_stride = acc_stride; q.s16 = prec == LAS_PREC_BF16;
        const int no_mma = las_fallback("LAS_LOC_POST_VALU") ? 1 : 0;         // (tests compare the two kernels; read per call)
        if (prec == LAS_PREC_BF16 && !no_mma && (A & 1) == 0) {       // (the MFMA kernel reads s as column pairs)
            const int ntiles = (A + 15) / 16, NW = ntiles <= 20 ? 4 : 8, NT = (ntiles + NW - 1) / NW;
            const dim3 grid((Tp + 15) / 16, B, L >= 48 ? 3 : 1), blk(64 * NW);
            switch (NT) {
                case 1: hipLaunchKernelGGL(att_loc_post_mma<1>, grid, blk, 0, st, q, NW); break;
                case 2: hipLaunchKernelGGL(att_loc_post_mma<2>, grid, blk, 0, st, q, NW); break;
                case 3: hipLaunchKernelGGL(att_loc_post_mma<3>, grid, blk, 0, st, q, NW); break;
                case 4: hipLaunchKernelGGL(att_loc_post_mma<4>, grid, blk, 0, st, q, NW); break;
                default: hipLaunchKernelGGL(att_loc_post_mma<5>, grid, blk, 0, st, q, NW); break;
            }
        } else {
            hipLaunchKernelGGL(att_loc_post, dim3((Tp + POST_TC - 1) / POST_TC, B), dim3(64 * AI), 0, st, q);
        }
        LAS_LAUNCH_OK();
    }
    if (!sums) return LAS_OK;
    if (loc) {
        const long conv_off = ((A * LOC_C + A + 1 + 3) / 4) * 4;
        const int ppb = L < 8 ? L : 8;                                   // (step, utterance) pairs per workgroup
        const long nblk = ((long)L * B + ppb - 1) / ppb;
        if (lds_cw > 64 * 1024) LAS_HIP(hipFuncSetAttribute((const void*)att_conv_wgrad, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cw));
        hipLaunchKernelGGL(att_conv_wgrad, dim3((unsigned)nblk), dim3(256), lds_cw, st, B, Tp, L, ppb, enc_len, s.att, w.df,
                           w.acc, acc_stride, conv_off);
        LAS_LAUNCH_OK();
    }
    // embedding rows
    if (pk) {
        int rc = las_dec_pk_bwd_emb(d, p, bw_, st);
        if (rc) return rc;
    }
    LAS_HIP(hipMemsetAsync(w.demb, 0, sizeof(float) * (size_t)d->V * C, st));
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(L * B), dim3(256), 0, st, s.tok, w.dxin, XI, C, w.demb);
    LAS_LAUNCH_OK();
    return LAS_OK;
}
