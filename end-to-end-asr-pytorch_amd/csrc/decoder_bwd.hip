// Attend-and-spell decoder loop, backward through time (one C-ABI call for all L steps).
//
// Hand-written counterpart of what autograd derives for reference src/asr.py:84-107 (Attention.forward
// :410-457, Speller.forward :352-357).  Per step, newest first:
//   cell pointwise bwd (per layer)           dgates from dh (top-layer grad + recurrent carry) and dc carry
//   dgates * [W_ih | W_hh]                   skinny MFMA products on transposed weight copies
//   d a = enc . d ctx (+ loc carry)          att_bwd_da       grid (T'-chunks, B)  HBM-bound on enc
//   softmax bwd, energy bwd                  att_bwd_energy   grid (T'-chunks, B)  HBM-bound on psi / s
//   dh0_{t-1} += dq_pre * W_phi              skinny MFMA product (accumulate)
// Sums over the L steps that are plain contractions (dW of every Linear/LSTMCell, d enc, d psi in dot mode)
// are left to ONE las_gemm each after the loop, on the buffers this call fills.
#include "las_mma.h"
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

// grid (NCH, B): d a[t'] = enc[b,t',:] . dctx[b,:]  (+ in loc mode the gradient that reaches this step's attention
// through the NEXT step's location convolution, gathered from that step's saved d f:
//   d prev[tau] = sum_c sum_k w[c][k] * df[c][tau + K - k] ).
__global__ __launch_bounds__(256) void att_bwd_da(int Tp, int E, int TC, const float* __restrict__ enc,
                                                  const int32_t* __restrict__ lens, const float* __restrict__ dctx,
                                                  long ld_dctx, const float* __restrict__ df_next,
                                                  const float* __restrict__ conv_w, float* __restrict__ da) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.y, t0 = blockIdx.x * TC, t1 = min(t0 + TC, Tp), len = lens[b];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Ep = (E + 3) & ~3, W = TC + 2 * LOC_K;
    float* cw_l = sm + Ep;                           // [10][201]
    float* df_l = cw_l + LOC_C * LOC_W;              // [10][TC + 200]: df[c][t0 - K + i]
    if (t0 >= len) {                                 // whole chunk beyond the utterance
        for (int t = t0 + threadIdx.x; t < t1; t += 256) da[(long)b * Tp + t] = 0.f;
        return;
    }
    for (int i = threadIdx.x; i < E; i += 256) sm[i] = dctx[(long)b * ld_dctx + i];
    if (df_next) {
        fill_batched<8>(conv_w, LOC_C * LOC_W, [&](int i, float v) { cw_l[i] = v; });
        const float* __restrict__ fp = df_next + (long)b * LOC_C * Tp;
        for (int i = threadIdx.x; i < LOC_C * W; i += 256) {
            const int c = i / W, j = i - c * W, t = t0 - LOC_K + j;
            const float v = fp[(long)c * Tp + min(max(t, 0), Tp - 1)];
            df_l[i] = (t >= 0 && t < Tp) ? v : 0.f;
        }
    }
    __syncthreads();
    const bool vec = (E & 3) == 0 && ((((uintptr_t)enc) & 15) == 0);
    for (int t = t0 + wave; t < t1; t += 4) {
        float acc = 0.f;
        if (t < len) {
            const float* __restrict__ p = enc + ((long)b * Tp + t) * E;
            if (vec) {
                float a0 = 0.f, a1 = 0.f;
                int i = lane;
                for (; i + 64 < E / 4; i += 128) {
                    const float4 v0 = ((const float4*)p)[i], v1 = ((const float4*)p)[i + 64];
                    const float4 w0 = ((const float4*)sm)[i], w1 = ((const float4*)sm)[i + 64];
                    a0 += v0.x * w0.x + v0.y * w0.y + v0.z * w0.z + v0.w * w0.w;
                    a1 += v1.x * w1.x + v1.y * w1.y + v1.z * w1.z + v1.w * w1.w;
                }
                for (; i < E / 4; i += 64) {
                    const float4 v0 = ((const float4*)p)[i], w0 = ((const float4*)sm)[i];
                    a0 += v0.x * w0.x + v0.y * w0.y + v0.z * w0.z + v0.w * w0.w;
                }
                acc = a0 + a1;
            } else {
                for (int i = lane; i < E; i += 64) acc += p[i] * sm[i];
            }
            if (df_next) {
                // local index of df[c][tau + K - k] is (tau - t0) + 2K - k
                const float* __restrict__ dl = df_l + (t - t0) + 2 * LOC_K;
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
        }
        if (lane == 0) da[(long)b * Tp + t] = acc;
    }
}

struct AttBwdArgs {
    int B, Tp, A, TC, NCH;
    const float* psi; const int32_t* lens;
    const float* att;         // [B][Tp] this step's attention
    const float* da;          // [B][Tp]
    const float* q;           // [B][A]
    float* dq_pre;            // [B][A]  (+=, atomics; already multiplied by 1-q^2)
    float* de;                // [B][Tp] d loss / d energy of this step (saved for the post-loop contractions)
    // loc
    const float* f;           // [B][10][Tp] location features of this step
    const float* s;           // [B][Tp][A] tanh(psi + q + u)
    const float* w_lp; const float* w_e;
    float* df;                // [B][10][Tp] d loss / d f of this step (caller-zeroed; frames < len written)
};

// grid (NCH, B).  Only what the NEXT (earlier) step needs stays on the sequential chain: d e (softmax backward),
// d q (sum over frames of d z) and d f (sum over the attention dim of d u * W_lp).  Everything that is a plain sum
// over the L steps -- d psi, d w_e, d b_e, d W_lp, d conv_w -- is left to att_loc_post / att_conv_wgrad after the
// loop, from the saved s / f / d e / d f.
template <bool LOC, int AI>
__global__ __launch_bounds__(256) void att_bwd_energy(AttBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float red[32];
    const int b = blockIdx.y, ch = blockIdx.x, t0 = ch * a.TC, t1 = min(t0 + a.TC, a.Tp), len = a.lens[b];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* de_l = sm;                                // [TC]
    if (t0 >= len) {                                 // nothing flows through frames beyond the utterance
        for (int t = t0 + threadIdx.x; t < t1; t += 256) a.de[(long)b * a.Tp + t] = 0.f;
        return;
    }
    // softmax backward needs the full-row dot  sum_t a[t] * da[t]
    float dot = 0.f;
    for (int i = threadIdx.x; i < len; i += 256) dot += a.att[(long)b * a.Tp + i] * a.da[(long)b * a.Tp + i];
    dot = block_sum(dot, red);
    for (int i = threadIdx.x; i < a.TC; i += 256) {
        const int t = t0 + i;
        float v = 0.f;
        if (t < t1 && t < len) v = ATT_SCALE * a.att[(long)b * a.Tp + t] * (a.da[(long)b * a.Tp + t] - dot);
        de_l[i] = v;
        if (t < t1) a.de[(long)b * a.Tp + t] = v;
    }
    __syncthreads();
    if (!LOC) {
        // dq[a] = sum_t de[t] * psi[b,t,a]
        const int tv = min(t1, len);
        for (int i = threadIdx.x; i < a.A; i += 256) {
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
    const int tcv = min(t1, len) - t0;               // valid frames in this chunk (> 0 here)
    float* we_l = de_l + a.TC;                       // [A]
    float* wlp_l = we_l + a.A;                       // [10][A]
    float* f_l = wlp_l + LOC_C * a.A;                // [10][TC]
    float* dq_l = f_l + LOC_C * a.TC;                // [4][A] per-wave partials
    // all rows this wave touches (<= ATT_ROWS, chunks are <= 20 frames) are requested up front: one round trip
    constexpr int ATT_ROWS = 5;
    float svr[ATT_ROWS][AI];
#pragma unroll
    for (int r = 0; r < ATT_ROWS; ++r) {
        const int t = min(t0 + wave + 4 * r, a.Tp - 1);
        const float* __restrict__ sp = a.s + ((long)b * a.Tp + t) * a.A;
#pragma unroll
        for (int k = 0; k < AI; ++k) svr[r][k] = sp[min(lane + 64 * k, a.A - 1)];
    }
    fill_batched<2>(a.w_e, a.A, [&](int i, float v) { we_l[i] = v; });
    fill_batched<8>(a.w_lp, LOC_C * a.A, [&](int i, float v) { const int aa = i / LOC_C, c = i - aa * LOC_C; wlp_l[c * a.A + aa] = v; });
    {
        const float* __restrict__ fp = a.f + (long)b * LOC_C * a.Tp;
        for (int i = threadIdx.x; i < LOC_C * a.TC; i += 256) {
            const int c = i / a.TC, tt = i - c * a.TC;
            const float v = fp[(long)c * a.Tp + min(t0 + tt, a.Tp - 1)];
            f_l[i] = (tt < tcv) ? v : 0.f;
        }
    }
    __syncthreads();
    float dq_r[AI];
#pragma unroll
    for (int k = 0; k < AI; ++k) dq_r[k] = 0.f;
    float* __restrict__ dfg = a.df + (long)b * LOC_C * a.Tp;
#pragma unroll
    for (int r = 0; r < ATT_ROWS; ++r) {
        const int tt = wave + 4 * r;
        if (tt >= tcv) break;
        const float de = de_l[tt];
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
                const float sv = svr[r][k];
                const float dz = de * we_l[i] * (1.f - sv * sv);
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
    for (int i = threadIdx.x; i < a.A; i += 256) {
        const float v = dq_l[i] + dq_l[a.A + i] + dq_l[2 * a.A + i] + dq_l[3 * a.A + i];
        const float qv = a.q[(long)b * a.A + i];
        atomicAdd(&a.dq_pre[(long)b * a.A + i], v * (1.f - qv * qv));
    }
}

struct LocPostArgs {
    int B, Tp, A, TC, L;
    const int32_t* lens;
    const float* de;          // [L][B][Tp]
    const float* f;           // [L][B][10][Tp]
    const float* s;           // [L][B][Tp][A]
    const float* w_lp; const float* w_e;
    float* dpsi;              // [B][Tp][A]   (caller-zeroed; rows < len written)
    float* acc;               // [B][acc_stride]: d w_lp^T [10][A] | d w_e [A] | d b_e [1] | pad | d conv [10*201]  (+=)
    long acc_stride;
};

// grid (NCH, B): the sums over the L steps that are off the sequential chain, in one pass over the saved s:
//   d psi[b,t,:] = sum_l dz_l,  d w_e = sum de_l * s_l,  d b_e = sum de_l,  d W_lp = sum du_l (x) f_l
// with dz_l = de_l * w_e * (1 - s_l^2), du_l = dz_l * (1 - u_l^2), u_l = tanh(W_lp f_l) recomputed.
template <int AI>
__global__ __launch_bounds__(256) void att_loc_post(LocPostArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float red[4];
    const int b = blockIdx.y, t0 = blockIdx.x * a.TC, t1 = min(t0 + a.TC, a.Tp), len = a.lens[b];
    const int tcv = min(t1, len) - t0;
    if (tcv <= 0) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* we_l = sm;                                // [A]
    float* wlp_l = we_l + a.A;                       // [10][A], later the [4][12][A] per-wave partials
    fill_batched<2>(a.w_e, a.A, [&](int i, float v) { we_l[i] = v; });
    fill_batched<8>(a.w_lp, LOC_C * a.A, [&](int i, float v) { const int aa = i / LOC_C, c = i - aa * LOC_C; wlp_l[c * a.A + aa] = v; });
    __syncthreads();
    constexpr int ROWS = 5;
    float wlp_r[AI][LOC_C], we_r[AI];
#pragma unroll
    for (int k = 0; k < AI; ++k) {
        const int i = min(lane + 64 * k, a.A - 1);
        we_r[k] = we_l[i];
#pragma unroll
        for (int c = 0; c < LOC_C; ++c) wlp_r[k][c] = wlp_l[c * a.A + i];
    }
    float dps[ROWS][AI], dwe_r[AI], dwlp_r[AI][LOC_C];
#pragma unroll
    for (int k = 0; k < AI; ++k) {
        dwe_r[k] = 0.f;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) dps[r][k] = 0.f;
#pragma unroll
        for (int c = 0; c < LOC_C; ++c) dwlp_r[k][c] = 0.f;
    }
    float dbe = 0.f;
    const long step_s = (long)a.B * a.Tp * a.A, step_f = (long)a.B * LOC_C * a.Tp, step_e = (long)a.B * a.Tp;
    auto load = [&](float (&sv)[ROWS][AI], int l) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int t = min(t0 + wave + 4 * r, t0 + tcv - 1);
            const float* __restrict__ sp = a.s + l * step_s + ((long)b * a.Tp + t) * a.A;
#pragma unroll
            for (int k = 0; k < AI; ++k) sv[r][k] = sp[min(lane + 64 * k, a.A - 1)];
        }
    };
    auto compute = [&](const float (&sv)[ROWS][AI], int l) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int tt = wave + 4 * r;
            if (tt >= tcv) break;
            const float de = a.de[l * step_e + (long)b * a.Tp + t0 + tt];     // wave-uniform: scalar loads
            dbe += de;
            float fc[LOC_C];
#pragma unroll
            for (int c = 0; c < LOC_C; ++c) fc[c] = a.f[l * step_f + ((long)b * LOC_C + c) * a.Tp + t0 + tt];
#pragma unroll
            for (int k = 0; k < AI; ++k) {
                float u = 0.f;
#pragma unroll
                for (int c = 0; c < LOC_C; ++c) u += wlp_r[k][c] * fc[c];
                u = fast_tanh(u);
                const float s_ = sv[r][k];
                const float dz = de * we_r[k] * (1.f - s_ * s_);
                dps[r][k] += dz;
                dwe_r[k] += de * s_;
                const float du = dz * (1.f - u * u);
#pragma unroll
                for (int c = 0; c < LOC_C; ++c) dwlp_r[k][c] += du * fc[c];
            }
        }
    };
    float svA[ROWS][AI], svB[ROWS][AI];
    load(svA, 0);
    for (int l = 0; l < a.L; l += 2) {
        if (l + 1 < a.L) load(svB, l + 1);
        compute(svA, l);
        if (l + 1 < a.L) {
            if (l + 2 < a.L) load(svA, l + 2);
            compute(svB, l + 1);
        }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int tt = wave + 4 * r;
        if (tt >= tcv) break;
        float* __restrict__ dp = a.dpsi + ((long)b * a.Tp + t0 + tt) * a.A;
#pragma unroll
        for (int k = 0; k < AI; ++k) {
            const int i = lane + 64 * k;
            if (i < a.A) dp[i] = dps[r][k];
        }
    }
    // ---- block reduction of the per-lane accumulators: per-wave partials [wave][11][A], then a summing pass
    __syncthreads();                                 // (wlp_l is dead: the weights live in registers)
    float* acc_l = wlp_l;
#pragma unroll
    for (int k = 0; k < AI; ++k) {
        const int i = lane + 64 * k;
        if (i < a.A) {
            float* o = acc_l + (long)wave * 11 * a.A + i;
            o[0] = dwe_r[k];
#pragma unroll
            for (int c = 0; c < LOC_C; ++c) o[(1 + c) * a.A] = dwlp_r[k][c];
        }
    }
    if (lane == 0) red[wave] = dbe;                  // dbe is wave-uniform
    __syncthreads();
    float* accg = a.acc + (long)b * a.acc_stride;
    for (int i = threadIdx.x; i < a.A * 11; i += 256) {
        const int j = i / a.A, aa = i - j * a.A;
        const float v = acc_l[i] + acc_l[11 * a.A + i] + acc_l[22 * a.A + i] + acc_l[33 * a.A + i];
        if (j == 0) atomicAdd(&accg[a.A * LOC_C + aa], v);
        else atomicAdd(&accg[(j - 1) * a.A + aa], v);        // [c][a]: contiguous atomics per wave
    }
    if (threadIdx.x == 0) atomicAdd(&accg[a.A * LOC_C + a.A], red[0] + red[1] + red[2] + red[3]);
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

extern "C" int las_decoder_bwd(const las_dec_dims* d, const las_dec_params* p, const float* enc, const float* psi,
                               const int32_t* enc_len, const las_dec_state* st_, const float* g_htop,
                               las_dec_bwd_state* bw_, void* stream) {
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
    LAS_HIP(hipMemsetAsync(w.dh_carry, 0, sizeof(float) * NL * BC, st));
    LAS_HIP(hipMemsetAsync(w.dc_carry, 0, sizeof(float) * NL * BC, st));
    LAS_HIP(hipMemsetAsync(w.dq_pre, 0, sizeof(float) * (size_t)L * B * A, st));
    if (loc) {
        LAS_CHECK_ARG(w.df && w.de && w.dpsi && w.acc);
        LAS_HIP(hipMemsetAsync(w.df, 0, sizeof(float) * (size_t)L * B * LOC_C * Tp, st));
        LAS_HIP(hipMemsetAsync(w.dpsi, 0, sizeof(float) * (size_t)B * Tp * A, st));
        LAS_HIP(hipMemsetAsync(w.acc, 0, sizeof(float) * (size_t)B * acc_stride, st));
    }
    size_t lds_e = sizeof(float) * (size_t)TC;
    // [TC] de | [A] w_e | [10][A] w_lp | [10][TC] f | [4][A] dq partials
    if (loc) lds_e = sizeof(float) * ((size_t)TC + A + LOC_C * (size_t)A + LOC_C * TC + 4 * (size_t)A);
    const size_t lds_da = sizeof(float) * (((size_t)E + 3) / 4 * 4 + (loc ? LOC_C * LOC_W + LOC_C * (TC + 2 * LOC_K) : 0));
    const size_t lds_post = sizeof(float) * ((size_t)A + 44 * (size_t)A);
    const size_t lds_cw = sizeof(float) * ((size_t)Tp + 2 * LOC_K + CW_KPT + LOC_C * (size_t)Tp);
    if (lds_e > 160 * 1024 || lds_da > 64 * 1024 || (loc && (lds_post > 160 * 1024 || lds_cw > 160 * 1024))) return LAS_E_UNSUPPORTED;
    for (int t = L - 1; t >= 0; --t) {
        // ---- LSTM cells, top layer first
        for (int l = NL - 1; l >= 0; --l) {
            const float* dh_ext = (l == NL - 1) ? g_htop + (long)t * BC : w.d_below;
            float* dg = w.dgates + ((long)l * L + t) * B * 4 * C;
            hipLaunchKernelGGL(cell_pw_bwd, dim3((B * C + 255) / 256), dim3(256), 0, st, B, C, dh_ext, (long)C,
                               w.dh_carry + (long)l * BC, w.dc_carry + (long)l * BC,
                               s.gates + ((long)l * L + t) * B * 4 * C, s.cs + ((long)l * (L + 1) + t + 1) * BC,
                               s.cs + ((long)l * (L + 1) + t) * BC, dg);
            LAS_LAUNCH_OK();
            const int Kx = l == 0 ? (int)XI : C;
            float* dx = l == 0 ? w.dxin + (long)t * B * XI : w.d_below;
            int rc = las_skinny_launch(prec, dg, 4 * C, p->w_ihT[l], 4 * C, 4 * C, nullptr, 0, nullptr, 0, 0, nullptr, 0, nullptr,
                                       0, 0, B, Kx, nullptr, nullptr, 0, dx, Kx, 0, 0, nullptr, nullptr, nullptr, nullptr, st);
            if (rc) return rc;
            // recurrent carry dh_{l,t-1} = dgates * W_hh; layer 0 gets the attention-query path in the same product
            // (second k-segment dq_pre_t * W_phi) and is therefore launched after the attention backward below
            if (l > 0 && t > 0) {
                rc = las_skinny_launch(prec, dg, 4 * C, p->w_hhT[l], 4 * C, 4 * C, nullptr, 0, nullptr, 0, 0, nullptr, 0, nullptr, 0,
                                       0, B, C, nullptr, nullptr, 0, w.dh_carry + (long)l * BC, C, 0, 0, nullptr, nullptr, nullptr,
                                       nullptr, st);
                if (rc) return rc;
            }
        }
        // ---- attention of step t
        // the location conv of step t+1 read this step's attention: its saved d f carries that gradient back
        const float* df_next = (loc && t + 1 < L) ? w.df + (long)(t + 1) * B * LOC_C * Tp : nullptr;
        hipLaunchKernelGGL(att_bwd_da, dim3(NCH, B), dim3(256), lds_da, st, Tp, E, TC, enc, enc_len,
                           w.dxin + (long)t * B * XI + C, XI, df_next, p->conv_w, w.da);
        LAS_LAUNCH_OK();
        AttBwdArgs a{};
        a.B = B; a.Tp = Tp; a.A = A; a.TC = TC; a.NCH = NCH;
        a.psi = psi; a.lens = enc_len;
        a.att = s.att + (long)(t + 1) * B * Tp;
        a.da = w.da;
        a.q = s.q + (long)t * B * A;
        a.dq_pre = w.dq_pre + (long)t * B * A;
        a.de = w.de + (long)t * B * Tp;
        a.f = loc ? s.f + (long)t * B * LOC_C * Tp : nullptr;
        a.s = loc ? s.s + (long)t * B * Tp * A : nullptr;
        a.w_lp = p->w_lp; a.w_e = p->w_e;
        a.df = loc ? w.df + (long)t * B * LOC_C * Tp : nullptr;
        if (!loc) {
            hipLaunchKernelGGL((att_bwd_energy<false, 1>), dim3(NCH, B), dim3(256), lds_e, st, a);
        } else {
#define LAS_ATT_GO(AIV)                                                                                           \
    {                                                                                                             \
        auto k = att_bwd_energy<true, AIV>;                                                                       \
        if (lds_e > 64 * 1024 && t == L - 1) LAS_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_e)); \
        hipLaunchKernelGGL(k, dim3(NCH, B), dim3(256), lds_e, st, a);                                             \
    }
            if (AI <= 1) LAS_ATT_GO(1) else if (AI <= 2) LAS_ATT_GO(2) else if (AI <= 4) LAS_ATT_GO(4)
            else if (AI <= 5) LAS_ATT_GO(5) else LAS_ATT_GO(8)
#undef LAS_ATT_GO
        }
        LAS_LAUNCH_OK();
        // ---- dh0_{t-1} = dgates_{0,t} * W_hh + dq_pre_t * W_phi   (one two-segment product)
        if (t > 0) {
            const float* dg0 = w.dgates + (long)t * B * 4 * C;
            int rc = las_skinny_launch(prec, dg0, 4 * C, p->w_hhT[0], 4 * C, 4 * C, w.dq_pre + (long)t * B * A, A, p->w_phiT, A, A,
                                       nullptr, 0, nullptr, 0, 0, B, C, nullptr, nullptr, 0, w.dh_carry, C, 0, 0, nullptr, nullptr,
                                       nullptr, nullptr, st);
            if (rc) return rc;
        }
    }
    if (loc) {
        // ---- sums over the L steps that are off the sequential chain
        LocPostArgs q{};
        q.B = B; q.Tp = Tp; q.A = A; q.TC = TC; q.L = L; q.lens = enc_len;
        q.de = w.de; q.f = s.f; q.s = s.s; q.w_lp = p->w_lp; q.w_e = p->w_e;
        q.dpsi = w.dpsi; q.acc = w.acc; q.acc_stride = acc_stride;
#define LAS_POST_GO(AIV)                                                                                          \
    {                                                                                                             \
        auto k = att_loc_post<AIV>;                                                                               \
        if (lds_post > 64 * 1024) LAS_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_post)); \
        hipLaunchKernelGGL(k, dim3(NCH, B), dim3(256), lds_post, st, q);                                          \
    }
        if (AI <= 1) LAS_POST_GO(1) else if (AI <= 2) LAS_POST_GO(2) else if (AI <= 4) LAS_POST_GO(4)
        else if (AI <= 5) LAS_POST_GO(5) else LAS_POST_GO(8)
#undef LAS_POST_GO
        LAS_LAUNCH_OK();
        const long conv_off = ((A * LOC_C + A + 1 + 3) / 4) * 4;
        const int ppb = L < 8 ? L : 8;                                   // (step, utterance) pairs per workgroup
        const long nblk = ((long)L * B + ppb - 1) / ppb;
        if (lds_cw > 64 * 1024) LAS_HIP(hipFuncSetAttribute((const void*)att_conv_wgrad, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cw));
        hipLaunchKernelGGL(att_conv_wgrad, dim3((unsigned)nblk), dim3(256), lds_cw, st, B, Tp, L, ppb, enc_len, s.att, w.df,
                           w.acc, acc_stride, conv_off);
        LAS_LAUNCH_OK();
    }
    // embedding rows
    LAS_HIP(hipMemsetAsync(w.demb, 0, sizeof(float) * (size_t)d->V * C, st));
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(L * B), dim3(256), 0, st, s.tok, w.dxin, XI, C, w.demb);
    LAS_LAUNCH_OK();
    return LAS_OK;
}
