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

// grid (NCH, B): d a[t'] = enc[b,t',:] . dctx[b,:] (+ carry from the later step's location conv); the consumed
// carry row is zeroed for reuse two steps later.
__global__ __launch_bounds__(256) void att_bwd_da(int Tp, int E, int TC, const float* __restrict__ enc,
                                                  const int32_t* __restrict__ lens, const float* __restrict__ dctx,
                                                  long ld_dctx, float* __restrict__ extra, float* __restrict__ da) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.y, t0 = blockIdx.x * TC, t1 = min(t0 + TC, Tp), len = lens[b];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < E; i += 256) sm[i] = dctx[(long)b * ld_dctx + i];
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
            acc = wave_sum(acc);
            if (extra) acc += extra[(long)b * Tp + t];
        }
        if (lane == 0) {
            da[(long)b * Tp + t] = acc;
            if (extra) extra[(long)b * Tp + t] = 0.f;
        }
    }
}

struct AttBwdArgs {
    int B, Tp, A, TC, NCH;
    const float* psi; const int32_t* lens;
    const float* att;         // [B][Tp] this step's attention
    const float* da;          // [B][Tp]
    const float* q;           // [B][A]
    float* dq_pre;            // [B][A]  (+=, atomics; already multiplied by 1-q^2)
    float* de;                // [B][Tp] (dot: saved for the d psi contraction)
    // loc
    const float* prev;        // [B][Tp] attention of the previous step (input of the conv)
    const float* f;           // [B][10][Tp]
    const float* s;           // [B][Tp][A]
    const float* conv_w; const float* w_lp; const float* w_e;
    float* dpsi;              // [B][Tp][A] (+=, block-owned rows)
    float* extra_out;         // [B][Tp] (+=, atomics): d loss / d prev
    float* acc;               // [B][acc_stride]: d w_lp^T [10][A] | d w_e [A] | d b_e [1] | pad | d conv [10*201]
    long acc_stride;
    int dbg;                  // timing experiments only (LAS_DBG_ATT): bit0 skip main loop, bit1 skip acc flush,
                              // bit2 skip conv backward, bit3 skip the LDS fills
};

// grid (NCH, B)
template <bool LOC, int AI>
__global__ __launch_bounds__(256) void att_bwd_energy(AttBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float red[32];
    const int b = blockIdx.y, ch = blockIdx.x, t0 = ch * a.TC, t1 = min(t0 + a.TC, a.Tp), len = a.lens[b];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (a.dbg & 16) return;
    float* de_l = sm;                                // [TC]
    // softmax backward needs the full-row dot  sum_t a[t] * da[t]
    float dot = 0.f;
    for (int i = threadIdx.x; i < len; i += 256) dot += a.att[(long)b * a.Tp + i] * a.da[(long)b * a.Tp + i];
    dot = block_sum(dot, red);
    for (int i = threadIdx.x; i < a.TC; i += 256) {
        const int t = t0 + i;
        float v = 0.f;
        if (t < t1 && t < len) v = ATT_SCALE * a.att[(long)b * a.Tp + t] * (a.da[(long)b * a.Tp + t] - dot);
        de_l[i] = v;
        if (!LOC && t < t1) a.de[(long)b * a.Tp + t] = v;
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
    const int tcv = min(t1, len) - t0;               // valid frames in this chunk
    if (tcv <= 0 || (a.dbg & 32)) return;                            // whole chunk beyond the utterance: nothing flows
    float* we_l = de_l + a.TC;                       // [A]
    float* wlp_l = we_l + a.A;                       // [10][A]
    float* cw_l = wlp_l + 48 * a.A;                  // [10][201]   (wlp_l region is 48A long: reused for the partials)
    float* prev_l = cw_l + LOC_C * LOC_W;            // [TC + 200]
    float* f_l = prev_l + a.TC + 2 * LOC_K;          // [10][TC]  saved location features of this chunk
    float* df_l = f_l + LOC_C * a.TC;                // [10][TC]
    // per-wave partial sums {dq, dwe, dwlp[10]} laid out [wave][12][A] (lane-contiguous: conflict-free plain stores);
    // the region starts on the w_lp tile, which is dead once the main loop is over
    float* acc_l = wlp_l;
    if (!(a.dbg & 8)) {
    fill_batched<2>(a.w_e, a.A, [&](int i, float v) { we_l[i] = v; });
    fill_batched<8>(a.w_lp, LOC_C * a.A, [&](int i, float v) { const int aa = i / LOC_C, c = i - aa * LOC_C; wlp_l[c * a.A + aa] = v; });
    fill_batched<8>(a.conv_w, LOC_C * LOC_W, [&](int i, float v) { cw_l[i] = v; });
    }
    {
        const float* __restrict__ pr = a.prev + (long)b * a.Tp;
        for (int i = threadIdx.x; i < a.TC + 2 * LOC_K; i += 256) {
            const int t = t0 - LOC_K + i;
            const float v = pr[min(max(t, 0), a.Tp - 1)];
            prev_l[i] = (t >= 0 && t < a.Tp) ? v : 0.f;
        }
        const float* __restrict__ fp = a.f + (long)b * LOC_C * a.Tp;
        for (int i = threadIdx.x; i < LOC_C * a.TC; i += 256) {
            const int c = i / a.TC, tt = i - c * a.TC;
            const float v = fp[(long)c * a.Tp + min(t0 + tt, a.Tp - 1)];
            f_l[i] = (tt < tcv) ? v : 0.f;
            df_l[i] = 0.f;
        }
    }
    __syncthreads();
    if (a.dbg & 64) return;
    float dq_r[AI], dwe_r[AI], dwlp_r[AI][LOC_C];
#pragma unroll
    for (int k = 0; k < AI; ++k) {
        dq_r[k] = 0.f; dwe_r[k] = 0.f;
#pragma unroll
        for (int c = 0; c < LOC_C; ++c) dwlp_r[k][c] = 0.f;
    }
    float dbe = 0.f;
    // all rows this wave touches (<= ATT_ROWS, chunks are <= 20 frames) are requested up front: one round trip
    constexpr int ATT_ROWS = 5;
    float svr[ATT_ROWS][AI], dpr[ATT_ROWS][AI];
#pragma unroll
    for (int r = 0; r < ATT_ROWS; ++r) {
        const int t = min(t0 + wave + 4 * r, a.Tp - 1);
        const float* __restrict__ sp = a.s + ((long)b * a.Tp + t) * a.A;
        const float* __restrict__ dpp = a.dpsi + ((long)b * a.Tp + t) * a.A;
#pragma unroll
        for (int k = 0; k < AI; ++k) {
            const int i = min(lane + 64 * k, a.A - 1);
            svr[r][k] = sp[i];
            dpr[r][k] = dpp[i];
        }
    }
#pragma unroll
    for (int r = 0; r < ATT_ROWS; ++r) {
        const int tt = wave + 4 * r;
        if (tt >= tcv || (a.dbg & 1)) break;
        const int t = t0 + tt;
        const float de = de_l[tt];
        dbe += de;
        float fc[LOC_C], dfc[LOC_C];
#pragma unroll
        for (int c = 0; c < LOC_C; ++c) { fc[c] = f_l[c * a.TC + tt]; dfc[c] = 0.f; }
        float* __restrict__ dp = a.dpsi + ((long)b * a.Tp + t) * a.A;
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
                dp[i] = dpr[r][k] + dz;
                dwe_r[k] += de * sv;
                const float du = dz * (1.f - u * u);
#pragma unroll
                for (int c = 0; c < LOC_C; ++c) { dwlp_r[k][c] += du * fc[c]; dfc[c] += du * wlp_l[c * a.A + i]; }
            }
        }
#pragma unroll
        for (int c = 0; c < LOC_C; ++c) {
            const float v = wave_sum(dfc[c]);
            if (lane == 0) df_l[c * a.TC + tt] = v;
        }
    }
    // ---- block reduction of the per-lane accumulators: per-wave partials, then a summing pass
    __syncthreads();                                 // every wave is done reading wlp_l
#pragma unroll
    for (int k = 0; k < AI; ++k) {
        const int i = lane + 64 * k;
        if (i < a.A) {
            float* o = acc_l + (long)wave * 12 * a.A + i;
            o[0] = dq_r[k]; o[a.A] = dwe_r[k];
#pragma unroll
            for (int c = 0; c < LOC_C; ++c) o[(2 + c) * a.A] = dwlp_r[k][c];
        }
    }
    dbe = wave_sum(dbe);
    if (lane == 0) red[wave] = dbe;
    __syncthreads();
    float* accg = a.acc + (long)b * a.acc_stride;      // one accumulator row per utterance (float atomics)
    for (int i = threadIdx.x; i < ((a.dbg & 2) ? 0 : a.A * 12); i += 256) {
        const int j = i / a.A, aa = i - j * a.A;         // j-major: coalesced LDS reads and global atomics
        const float v = acc_l[i] + acc_l[12 * a.A + i] + acc_l[24 * a.A + i] + acc_l[36 * a.A + i];
        if (j == 0) {
            const float qv = a.q[(long)b * a.A + aa];
            atomicAdd(&a.dq_pre[(long)b * a.A + aa], v * (1.f - qv * qv));
        } else if (j == 1) {
            atomicAdd(&accg[a.A * LOC_C + aa], v);
        } else {
            atomicAdd(&accg[(j - 2) * a.A + aa], v);     // [c][a]: contiguous atomics per wave
        }
    }
    if (threadIdx.x == 0) atomicAdd(&accg[a.A * LOC_C + a.A], red[0] + red[1] + red[2] + red[3]);
    // ---- location conv backward
    if (a.dbg & 4) return;
    // d prev[tau] += sum_c sum_{t in chunk} w[c][tau - t + K] * df[c][t]
    for (int i = threadIdx.x; i < a.TC + 2 * LOC_K; i += 256) {
        const int tau = t0 - LOC_K + i;
        if (tau < 0 || tau >= a.Tp) continue;
        float acc = 0.f;
        const int lo = max(0, tau - LOC_K - t0), hi = min(tcv, tau + LOC_K - t0 + 1);
        for (int c = 0; c < LOC_C; ++c)
            for (int tt = lo; tt < hi; ++tt) acc += cw_l[c * LOC_W + (tau - (t0 + tt) + LOC_K)] * df_l[c * a.TC + tt];
        if (acc != 0.f) atomicAdd(&a.extra_out[(long)b * a.Tp + tau], acc);
    }
    // d w[c][k] += sum_{t in chunk} df[c][t] * prev[t + k - K]
    const long conv_off = ((a.A * LOC_C + a.A + 1 + 3) / 4) * 4;
    for (int i = threadIdx.x; i < LOC_C * LOC_W; i += 256) {
        const int c = i / LOC_W, k = i % LOC_W;
        float acc = 0.f;
        for (int tt = 0; tt < tcv; ++tt) acc += df_l[c * a.TC + tt] * prev_l[tt + k];
        atomicAdd(&accg[conv_off + i], acc);
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
        LAS_HIP(hipMemsetAsync(w.extra, 0, sizeof(float) * 2 * (size_t)B * Tp, st));
        LAS_HIP(hipMemsetAsync(w.dpsi, 0, sizeof(float) * (size_t)B * Tp * A, st));
        LAS_HIP(hipMemsetAsync(w.acc, 0, sizeof(float) * (size_t)B * acc_stride, st));
    }
    size_t lds_e = sizeof(float) * (size_t)TC;
    // [TC] de | [A] w_e | max([10A] w_lp, then [4][12][A] partials overlaid from here) ... the overlay may run over
    // cw/prev/f/df, which must stay live for the conv backward, so it gets its own tail instead: size = 48A after w_e
    if (loc) lds_e = sizeof(float) * ((size_t)TC + A + 48 * (size_t)A + LOC_C * LOC_W + TC + 2 * LOC_K + 2 * LOC_C * TC);
    if (lds_e > 160 * 1024) return LAS_E_UNSUPPORTED;
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
        float* extra_in = loc ? w.extra + (long)(t & 1) * B * Tp : nullptr;
        float* extra_out = loc ? w.extra + (long)((t + 1) & 1) * B * Tp : nullptr;
        hipLaunchKernelGGL(att_bwd_da, dim3(NCH, B), dim3(256), sizeof(float) * E, st, Tp, E, TC, enc, enc_len,
                           w.dxin + (long)t * B * XI + C, XI, extra_in, w.da);
        LAS_LAUNCH_OK();
        AttBwdArgs a{};
        a.B = B; a.Tp = Tp; a.A = A; a.TC = TC; a.NCH = NCH;
        a.psi = psi; a.lens = enc_len;
        a.att = s.att + (long)(t + 1) * B * Tp;
        a.da = w.da;
        a.q = s.q + (long)t * B * A;
        a.dq_pre = w.dq_pre + (long)t * B * A;
        a.de = loc ? nullptr : w.de + (long)t * B * Tp;
        a.prev = s.att + (long)t * B * Tp;
        a.f = loc ? s.f + (long)t * B * LOC_C * Tp : nullptr;
        a.s = loc ? s.s + (long)t * B * Tp * A : nullptr;
        a.conv_w = p->conv_w; a.w_lp = p->w_lp; a.w_e = p->w_e;
        a.dpsi = w.dpsi; a.extra_out = extra_out; a.acc = w.acc; a.acc_stride = acc_stride;
        { static const char* e = getenv("LAS_DBG_ATT"); a.dbg = e ? atoi(e) : 0; }
        if (!loc) {
            hipLaunchKernelGGL((att_bwd_energy<false, 1>), dim3(NCH, B), dim3(256), lds_e, st, a);
        } else {
#define LAS_ATT_GO(AIV)                                                                                           \
    {                                                                                                             \
        auto k = att_bwd_energy<true, AIV>;                                                                       \
        if (lds_e > 64 * 1024) LAS_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_e)); \
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
    // embedding rows
    LAS_HIP(hipMemsetAsync(w.demb, 0, sizeof(float) * (size_t)d->V * C, st));
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(L * B), dim3(256), 0, st, s.tok, w.dxin, XI, C, w.demb);
    LAS_LAUNCH_OK();
    return LAS_OK;
}
