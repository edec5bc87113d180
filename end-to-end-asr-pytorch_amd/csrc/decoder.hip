// Attend-and-spell decoder loop (forward), one C-ABI call for all L steps.
//
// Replaces the Python loop at reference src/asr.py:84-107: Attention.forward (:410-457, modes 'dot' and
// 'loc', softmax scale 2.0, -inf mask beyond enc_len, context over the RAW encoder features), the Speller
// stack of nn.LSTMCell (:352-357), the embedding look-ups (:74,79,100,102) and the scheduled-sampling /
// greedy feedback (:95-102).  No per-step device->host traffic (the reference does a.cpu() per step, :107).
//
// Per step: q = tanh(phi h0_{t-1})            skinny MFMA product, tanh epilogue
//           energies e[b,t']                   att_energy_fwd   grid (T'-chunks, B)   HBM-bound on psi(enc)
//           softmax(2e), context               att_softmax_ctx  grid (E-chunks, B)    HBM-bound on enc
//           LSTM cells                         skinny MFMA product, cell epilogue
// Everything needed by the backward pass is kept in caller-owned buffers (las_dec_state).
#include "las_mma.h"
#include "decoder_pk.h"
#include <stdlib.h>

int las_skinny_launch(int prec, const float* x0, long ldx0, const float* w0, long ldw0, int K0, const float* x1,
                      long ldx1, const float* w1, long ldw1, int K1, const float* x2, long ldx2, const float* w2,
                      long ldw2, int K2, int B, int N, const float* bias0, const float* bias1, int mode, float* out,
                      long ldo, int accumulate, int C, const float* c_prev, float* h_out, float* c_out,
                      float* gates_out, hipStream_t st);

namespace {

constexpr int LOC_C = 10, LOC_K = 100, LOC_W = 2 * LOC_K + 1;     // reference asr.py:395-398
constexpr float ATT_SCALE = 2.0f;                                 // reference asr.py:410

// rows[r][0:C] = emb[tok(r)]; tok from int64 teacher labels y[b][t] (r = t*B + b) or from int32 tok[r]
__global__ __launch_bounds__(256) void embed_rows_kernel(const float* __restrict__ emb, const long long* __restrict__ y,
                                                         int Ly, const int32_t* __restrict__ tok_in, int B, int C,
                                                         int V, float* __restrict__ out, long ldo,
                                                         int32_t* __restrict__ tok_out) {
    const int r = blockIdx.x, t = r / B, b = r % B;
    int tk = tok_in ? tok_in[r] : (int)y[(long)b * Ly + t];
    tk = min(max(tk, 0), V - 1);
    if (tok_out && threadIdx.x == 0) tok_out[r] = tk;
    const float* s = emb + (long)tk * C;
    float* o = out + (long)r * ldo;
    for (int i = threadIdx.x; i < C; i += 256) o[i] = s[i];
}

// prev[b][t'] = 1/len for t' < len else 0   (reference asr.py:444-449)
__global__ void uniform_att_kernel(const int32_t* __restrict__ lens, int Tp, float* __restrict__ prev) {
    const int b = blockIdx.x, l = lens[b];
    for (int i = threadIdx.x; i < Tp; i += blockDim.x) prev[(long)b * Tp + i] = i < l ? 1.f / (float)l : 0.f;
}

struct AttArgs {
    int B, Tp, E, A, loc, TC;
    const float* psi; const float* enc; const int32_t* lens;
    const float* q;           // [B][A] this step
    const float* prev;        // [B][Tp] previous attention (loc)
    const float* conv_w; const float* w_lp; const float* w_e; const float* b_e;
    float* e;                 // [B][Tp]
    float* f;                 // [B][10][Tp]  (loc, saved)
    void* s; int s16;         // [B][Tp][A]   (loc, saved): fp32, or the 16-bit code of las_common.h (s16, bf16 mode)
};

// grid (NCH, B): energies of T'-chunk [t0, t0+TC) of utterance b.  8 waves per workgroup: a chunk's <= 20 frames
// (att_chunks) are <= 3 per wave, and the location convolution's 10 x TC outputs are split over two half-ranges of
// taps per output so that all 512 threads work on it.
constexpr int ATTF_NW = 8, ATTF_NT = 64 * ATTF_NW;
constexpr int ATT_ROWS = (20 + ATTF_NW - 1) / ATTF_NW;

template <bool LOC, int AI>     // AI = ceil(A / 64) rounded up to {1,2,4,5,8}
__global__ __launch_bounds__(ATTF_NT) void att_energy_fwd(AttArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.y, t0 = blockIdx.x * a.TC, len = a.lens[b];
    const int t1 = min(t0 + a.TC, a.Tp);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* q_l = sm;                                 // [A]
    // every psi row this wave will touch is requested before anything else (one memory round trip, overlapped
    // with the staging and the convolution below)
    float pv[ATT_ROWS][AI];
    if (LOC) {
#pragma unroll
        for (int r = 0; r < ATT_ROWS; ++r) {
            const int t = min(t0 + wave + ATTF_NW * r, a.Tp - 1);
            const float* __restrict__ p = a.psi + ((long)b * a.Tp + t) * a.A;
#pragma unroll
            for (int k = 0; k < AI; ++k) pv[r][k] = p[min(lane + 64 * k, a.A - 1)];
        }
    }
    fill_batched<2>(a.q + (long)b * a.A, a.A, [&](int i, float v) { q_l[i] = v; });
    if (!LOC) {
        __syncthreads();
        for (int t = t0 + wave; t < t1; t += ATTF_NW) {
            float acc = 0.f;
            if (t < len) {
                const float* p = a.psi + ((long)b * a.Tp + t) * a.A;
                for (int i = lane; i < a.A; i += 64) acc += p[i] * q_l[i];
                acc = wave_sum(acc);
            }
            if (lane == 0) a.e[(long)b * a.Tp + t] = acc;
        }
        return;
    }
    float* we_l = q_l + a.A;                         // [A]
    float* wlp_l = we_l + a.A;                       // [10][A] (transposed: conflict-free over a)
    float* cw_l = wlp_l + LOC_C * a.A;               // [10][201]
    float* prev_l = cw_l + LOC_C * LOC_W;            // [TC + 200]
    float* f_l = prev_l + a.TC + 2 * LOC_K;          // [10][TC]
    float* fh_l = f_l + LOC_C * a.TC;                // [10][TC] second half-range of taps
    fill_batched<2>(a.w_e, a.A, [&](int i, float v) { we_l[i] = v; });
    fill_batched<8>(a.w_lp, LOC_C * a.A, [&](int i, float v) { const int aa = i / LOC_C, c = i - aa * LOC_C; wlp_l[c * a.A + aa] = v; });
    fill_batched<4>(a.conv_w, LOC_C * LOC_W, [&](int i, float v) { cw_l[i] = v; });
    {
        const float* __restrict__ pr = a.prev + (long)b * a.Tp;
        for (int i = threadIdx.x; i < a.TC + 2 * LOC_K; i += ATTF_NT) {
            const int t = t0 - LOC_K + i;
            const float v = pr[min(max(t, 0), a.Tp - 1)];
            prev_l[i] = (t >= 0 && t < a.Tp) ? v : 0.f;
        }
    }
    __syncthreads();
    // location features f[c][t] = sum_k w[c][k] * prev[t + k - K]: thread = (output, half of the taps)
    {
        const int half = threadIdx.x >= ATTF_NT / 2, i = threadIdx.x - half * (ATTF_NT / 2);
        if (i < LOC_C * a.TC) {
            const int c = i / a.TC, tt = i % a.TC;
            float acc = 0.f;
            if (t0 + tt < t1) {
                constexpr int H0 = 102;              // taps [0,102) and [102,201): 3 x 34 and 3 x 33
                const int k0 = half ? H0 : 0, k1 = half ? LOC_W : H0;
                const float* w = cw_l + c * LOC_W;
                const float* p = prev_l + tt;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f;         // 3 chains x unroll: LDS latency, not issue, bounds this loop
#pragma unroll 8
                for (int k = k0; k + 2 < k1; k += 3) { a0 += w[k] * p[k]; a1 += w[k + 1] * p[k + 1]; a2 += w[k + 2] * p[k + 2]; }
                acc = (a0 + a1) + a2;
            }
            (half ? fh_l : f_l)[c * a.TC + tt] = acc;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LOC_C * a.TC; i += ATTF_NT) {
        const int c = i / a.TC, tt = i % a.TC;
        const float v = f_l[i] + fh_l[i];
        f_l[i] = v;
        if (t0 + tt < t1) a.f[((long)b * LOC_C + c) * a.Tp + t0 + tt] = v;
    }
    __syncthreads();
    const float be = a.b_e[0];
#pragma unroll
    for (int r = 0; r < ATT_ROWS; ++r) {
        const int t = t0 + wave + ATTF_NW * r;
        if (t >= t1) break;
        float acc = 0.f;
        const int tt = t - t0;
        if (t < len) {
            const long so = ((long)b * a.Tp + t) * a.A;
            float fc[LOC_C];
#pragma unroll
            for (int c = 0; c < LOC_C; ++c) fc[c] = f_l[c * a.TC + tt];
#pragma unroll
            for (int k = 0; k < AI; ++k) {
                const int i = lane + 64 * k;
                if (i < a.A) {
                    float u = 0.f;
#pragma unroll
                    for (int c = 0; c < LOC_C; ++c) u += wlp_l[c * a.A + i] * fc[c];
                    u = fast_tanh(u);
                    const float sv = fast_tanh(pv[r][k] + q_l[i] + u);
                    las_s_store(a.s, so + i, a.s16, sv);
                    acc += we_l[i] * sv;
                }
            }
            acc = wave_sum(acc) + be;
        }
        if (lane == 0) a.e[(long)b * a.Tp + t] = acc;
    }
}

// grid (ECH, B): masked softmax(2e) over T' (recomputed per block: T' floats), context for 64 columns of E
__global__ __launch_bounds__(256) void att_softmax_ctx(int Tp, int E, const float* __restrict__ e,
                                                       const float* __restrict__ enc, const int32_t* __restrict__ lens,
                                                       float* __restrict__ att_out, float* __restrict__ ctx_out,
                                                       long ld_ctx) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* a_l = sm;                 // [Tp] (padded to 4)
    float* red = sm + ((Tp + 3) & ~3);   // [32]
    float* part = red + 32;          // [16][64]
    const int b = blockIdx.y, len = lens[b];
    float m = -INFINITY;
    for (int i = threadIdx.x; i < len; i += 256) { const float v = ATT_SCALE * e[(long)b * Tp + i]; a_l[i] = v; m = fmaxf(m, v); }
    m = block_max(m, red);
    float s = 0.f;
    for (int i = threadIdx.x; i < len; i += 256) { const float v = expf(a_l[i] - m); a_l[i] = v; s += v; }
    s = block_sum(s, red);
    const float inv = 1.f / s;
    for (int i = threadIdx.x; i < Tp; i += 256) {
        const float v = i < len ? a_l[i] * inv : 0.f;
        a_l[i] = v;
        if (blockIdx.x == 0) att_out[(long)b * Tp + i] = v;
    }
    __syncthreads();
    // context for my 64 columns: thread = (column quad, one of 16 time groups); 8 independent 16-byte loads in flight
    // per thread (the 4-deep, one-column-per-thread form spent ~19 dependent round trips on T' = 300)
    const bool vec = (E & 3) == 0 && ((((uintptr_t)enc) & 15) == 0);
    if (vec) {
        const int cq = threadIdx.x & 15, tg = threadIdx.x >> 4, col = blockIdx.x * 64 + cq * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < E) {
            const float* __restrict__ p = enc + (long)b * Tp * E + col;
            int t = tg;
            for (; t + 7 * 16 < len; t += 8 * 16) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *(const float4*)(p + (long)(t + 16 * u) * E);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float w = a_l[t + 16 * u];
                    acc.x += w * v[u].x; acc.y += w * v[u].y; acc.z += w * v[u].z; acc.w += w * v[u].w;
                }
            }
            for (; t < len; t += 16) {
                const float4 v = *(const float4*)(p + (long)t * E);
                const float w = a_l[t];
                acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
            }
        }
        float* part4 = part;                             // [16][64] (the caller sizes the region for it)
        *(float4*)(part4 + tg * 64 + cq * 4) = acc;
        __syncthreads();
        const int c = threadIdx.x;
        if (c < 64 && blockIdx.x * 64 + c < E) {
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) sum += part4[g * 64 + c];
            ctx_out[(long)b * ld_ctx + blockIdx.x * 64 + c] = sum;
        }
        return;
    }
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), tg = threadIdx.x >> 6;
    float acc = 0.f;
    if (col < E) {
        const float* __restrict__ p = enc + (long)b * Tp * E + col;
        float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
        int t = tg;
        for (; t + 12 < len; t += 16) {
            c0 += a_l[t] * p[(long)t * E];
            c1 += a_l[t + 4] * p[(long)(t + 4) * E];
            c2 += a_l[t + 8] * p[(long)(t + 8) * E];
            c3 += a_l[t + 12] * p[(long)(t + 12) * E];
        }
        for (; t < len; t += 4) c0 += a_l[t] * p[(long)t * E];
        acc = (c0 + c1) + (c2 + c3);
    }
    part[tg * 64 + (threadIdx.x & 63)] = acc;
    __syncthreads();
    if (tg == 0 && col < E)
        ctx_out[(long)b * ld_ctx + col] = part[threadIdx.x] + part[64 + threadIdx.x] + part[128 + threadIdx.x] + part[192 + threadIdx.x];
}

// next token: argmax (greedy, reference asr.py:102) or a draw from softmax(logits) (asr.py:99)
__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__global__ __launch_bounds__(256) void pick_token_kernel(const float* __restrict__ logits, int V, int greedy,
                                                         unsigned seed, int32_t* __restrict__ tok) {
    __shared__ float red[32];
    __shared__ float bestv[256];
    __shared__ int besti[256];
    const int b = blockIdx.x;
    const float* x = logits + (long)b * V;
    float bv = -INFINITY; int bi = 0;
    if (greedy) {
        for (int i = threadIdx.x; i < V; i += 256) if (x[i] > bv) { bv = x[i]; bi = i; }
    } else {
        // Gumbel-max: argmax_i (x_i + g_i), g_i = -log(-log(u_i))
        for (int i = threadIdx.x; i < V; i += 256) {
            const unsigned h = hash32(seed ^ hash32((unsigned)(b * 0x9e3779b9u + i)));
            const float u = ((h >> 8) + 0.5f) * (1.f / 16777216.f);
            const float v = x[i] - logf(-logf(u));
            if (v > bv) { bv = v; bi = i; }
        }
    }
    bestv[threadIdx.x] = bv; besti[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float ov = bestv[threadIdx.x + o]; const int oi = besti[threadIdx.x + o];
            if (ov > bestv[threadIdx.x] || (ov == bestv[threadIdx.x] && oi < besti[threadIdx.x])) { bestv[threadIdx.x] = ov; besti[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) tok[b] = besti[0];
    (void)red;
}

// T' is cut into chunks of <= 20 frames: a wave of the energy kernels then owns <= 5 rows (ATT_ROWS) and can request
// all of them up front.  Must be identical in decoder.hip and decoder_bwd.hip.
int att_chunks(int Tp) { const int n = (Tp + 19) / 20; return n < 1 ? 1 : n; }

}  // namespace

// resume: one step (L = 1) from caller-provided states (slot 0 of hs/cs, att[0], tok[0..B)), see las_decoder_step
static int decoder_run(const las_dec_dims* d, const las_dec_params* p, const float* enc, const float* psi,
                       const int32_t* enc_len, const int64_t* y, int Ly, const uint8_t* step_mode, unsigned seed,
                       las_dec_state* st_, void* stream, bool resume) {
    LAS_CHECK_ARG(d && p && enc && psi && enc_len && st_);
    const int B = d->B, Tp = d->Tp, E = d->E, A = d->A, C = d->C, NL = d->NL, V = d->V, L = d->L, loc = d->loc, prec = d->prec;
    LAS_CHECK_ARG(B > 0 && Tp > 0 && E > 0 && A > 0 && C > 0 && NL >= 1 && NL <= 4 && L >= 0 && V > 1);
    if (L == 0) return LAS_OK;
    las_dec_state& s = *st_;
    hipStream_t st = (hipStream_t)stream;
    bool all_teacher_dummy_; (void)all_teacher_dummy_;
    bool all_teacher = true;
    if (step_mode) for (int t = 0; t < L; ++t) if (step_mode[t] != 1) all_teacher = false;
    LAS_CHECK_ARG(!all_teacher || (y && Ly >= L));
    LAS_CHECK_ARG(!y || Ly >= L);
    LAS_CHECK_ARG(all_teacher || (p->w_char && p->b_char && s.logits_step));
    const long XI = C + E, BC = (long)B * C;
    // zero initial states (reference asr.py:338-340)
    for (int l = 0; l < NL && !resume; ++l) {
        LAS_HIP(hipMemsetAsync(s.hs + (long)l * (L + 1) * BC, 0, sizeof(float) * BC, st));
        LAS_HIP(hipMemsetAsync(s.cs + (long)l * (L + 1) * BC, 0, sizeof(float) * BC, st));
    }
    if (loc && !resume) { hipLaunchKernelGGL(uniform_att_kernel, dim3(B), dim3(256), 0, st, enc_len, Tp, s.att); LAS_LAUNCH_OK(); }
    // embeddings of the teacher tokens for every step (step 0 feeds y[:,0] = <sos>)
    if (y) {
        hipLaunchKernelGGL(embed_rows_kernel, dim3(L * B), dim3(256), 0, st, p->emb, (const long long*)y, Ly, nullptr, B, C, V,
                           s.xin, XI, s.tok);
        LAS_LAUNCH_OK();
    }
    // all steps teacher-forced, one layer, loc attention, no dropout, and the caller provided the workspace: one
    // persistent launch for the whole loop (decoder_pk.hip); everything else: four launches per step below
    if (all_teacher && y && !resume && s.pk_ws && s.pk_status && las_dec_pk_fwd_ws_bytes(d) > 0)
        return las_dec_pk_fwd(d, p, enc, psi, enc_len, st_, st);
    const int NCH = att_chunks(Tp), TC = (Tp + NCH - 1) / NCH, ECH = (E + 63) / 64;
    if (TC > 20) return LAS_E_UNSUPPORTED;
    size_t lds_e = sizeof(float) * (size_t)A;
    if (loc) lds_e = sizeof(float) * ((size_t)2 * A + LOC_C * A + LOC_C * LOC_W + TC + 2 * LOC_K + 2 * LOC_C * TC);
    if (lds_e > 64 * 1024) return LAS_E_UNSUPPORTED;
    const size_t lds_s = sizeof(float) * (((size_t)Tp + 3) / 4 * 4 + 32 + 1024);
    if (lds_s > 64 * 1024) return LAS_E_UNSUPPORTED;
    for (int t = 0; t < L; ++t) {
        const float* h0_prev = s.hs + (long)t * BC;                           // layer 0, slot t = h_{t-1}
        float* q_t = s.q + (long)t * B * A;
        int rc = las_skinny_launch_pk(prec, h0_prev, C, p->w_phi, C, C, nullptr, 0, nullptr, 0, 0, nullptr, 0, nullptr, 0, 0,
                                      B, A, nullptr, nullptr, 1, q_t, A, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr,
                                      p->pk_phi, st);
        if (rc) return rc;
        AttArgs a{};
        a.B = B; a.Tp = Tp; a.E = E; a.A = A; a.loc = loc; a.TC = TC;
        a.psi = psi; a.enc = enc; a.lens = enc_len; a.q = q_t;
        a.prev = s.att + (long)t * B * Tp;
        a.conv_w = p->conv_w; a.w_lp = p->w_lp; a.w_e = p->w_e; a.b_e = p->b_e;
        a.e = s.ebuf;
        a.f = loc ? s.f + (long)t * B * LOC_C * Tp : nullptr;
        a.s16 = prec == LAS_PREC_BF16;
        a.s = loc ? (void*)((char*)s.s + (size_t)t * B * Tp * A * (a.s16 ? 2 : 4)) : nullptr;
        if (!loc) hipLaunchKernelGGL((att_energy_fwd<false, 1>), dim3(NCH, B), dim3(ATTF_NT), lds_e, st, a);
        else {
            const int AI = (A + 63) / 64;
            if (AI <= 1) hipLaunchKernelGGL((att_energy_fwd<true, 1>), dim3(NCH, B), dim3(ATTF_NT), lds_e, st, a);
            else if (AI <= 2) hipLaunchKernelGGL((att_energy_fwd<true, 2>), dim3(NCH, B), dim3(ATTF_NT), lds_e, st, a);
            else if (AI <= 4) hipLaunchKernelGGL((att_energy_fwd<true, 4>), dim3(NCH, B), dim3(ATTF_NT), lds_e, st, a);
            else if (AI <= 5) hipLaunchKernelGGL((att_energy_fwd<true, 5>), dim3(NCH, B), dim3(ATTF_NT), lds_e, st, a);
            else if (AI <= 8) hipLaunchKernelGGL((att_energy_fwd<true, 8>), dim3(NCH, B), dim3(ATTF_NT), lds_e, st, a);
            else return LAS_E_UNSUPPORTED;
        }
        LAS_LAUNCH_OK();
        float* xin_t = s.xin + (long)t * B * XI;
        hipLaunchKernelGGL(att_softmax_ctx, dim3(ECH, B), dim3(256), lds_s, st, Tp, E, s.ebuf, enc, enc_len,
                           s.att + (long)(t + 1) * B * Tp, xin_t + C, XI);
        LAS_LAUNCH_OK();
        // feedback token for this step's input if it is not the teacher's (decided after step t-1's logits)
        if (t > 0 && step_mode && step_mode[t] != 1) {
            const float* htop = s.hs + ((long)(NL - 1) * (L + 1) + t) * BC;   // h_top of step t-1
            rc = las_skinny_launch(prec, htop, C, p->w_char, C, C, nullptr, 0, nullptr, 0, 0, nullptr, 0, nullptr, 0, 0, B, V,
                                   p->b_char, nullptr, 0, s.logits_step, V, 0, 0, nullptr, nullptr, nullptr, nullptr, st);
            if (rc) return rc;
            hipLaunchKernelGGL(pick_token_kernel, dim3(B), dim3(256), 0, st, s.logits_step, V, step_mode[t] == 2 ? 1 : 0,
                               seed + 0x9e3779b9u * (unsigned)t, s.tok + (long)t * B);
            LAS_LAUNCH_OK();
            hipLaunchKernelGGL(embed_rows_kernel, dim3(B), dim3(256), 0, st, p->emb, nullptr, 0, s.tok + (long)t * B, B, C, V,
                               xin_t, XI, nullptr);
            LAS_LAUNCH_OK();
        } else if (t == 0 && !y) {
            // no teacher at all: <sos> = 0 (or, resuming, the caller's tokens)
            if (!resume) LAS_HIP(hipMemsetAsync(s.tok, 0, sizeof(int32_t) * B, st));
            hipLaunchKernelGGL(embed_rows_kernel, dim3(B), dim3(256), 0, st, p->emb, nullptr, 0, s.tok, B, C, V, xin_t, XI, nullptr);
            LAS_LAUNCH_OK();
        }
        // LSTM cells (reference asr.py:352-357): dropout on the cell-0 input, and on the recurrent state of layers >= 1
        const bool drop = d->dropout > 0.f;
        if (drop) {
            LAS_CHECK_ARG(s.xdrop && (NL == 1 || s.hdrop));
            rc = las_dropout_rows(xin_t, XI, s.xdrop + (long)t * B * XI, XI, B, (int)XI, d->dropout,
                                  las_decoder_drop_seed(d->drop_seed, t, 0), stream);
            if (rc) return rc;
        }
        for (int l = 0; l < NL; ++l) {
            const float* x = l == 0 ? (drop ? s.xdrop + (long)t * B * XI : xin_t) : s.hs + ((long)(l - 1) * (L + 1) + t + 1) * BC;
            const int Kx = l == 0 ? (int)XI : C;
            const float* hp = s.hs + ((long)l * (L + 1) + t) * BC;
            if (drop && l > 0) {
                float* hd = s.hdrop + ((long)l * L + t) * BC;
                rc = las_dropout_rows(hp, C, hd, C, B, C, d->dropout, las_decoder_drop_seed(d->drop_seed, t, l), stream);
                if (rc) return rc;
                hp = hd;
            }
            const float* cp = s.cs + ((long)l * (L + 1) + t) * BC;
            rc = las_skinny_launch_pk(prec, x, Kx, p->w_ih[l], Kx, Kx, hp, C, p->w_hh[l], C, C, nullptr, 0, nullptr, 0, 0, B, 4 * C,
                                      p->b_ih[l], p->b_hh[l], 2, nullptr, 0, 0, C, cp, s.hs + ((long)l * (L + 1) + t + 1) * BC,
                                      s.cs + ((long)l * (L + 1) + t + 1) * BC, s.gates + ((long)l * L + t) * B * 4 * C, nullptr,
                                      p->pk_cell[l], st);
            if (rc) return rc;
        }
    }
    return LAS_OK;
}

// The token draw of a scheduled-sampling / greedy step on its own (reference asr.py:99 Categorical(softmax).sample(), :102 argmax):
// the kernel las_decoder_fwd launches per sampled step, exposed so that its distribution can be tested.
extern "C" int las_sample_rows(const float* logits, int rows, int V, int greedy, unsigned seed, int32_t* tok, void* stream) {
    LAS_CHECK_ARG(logits && tok && rows >= 0 && V > 0);
    if (rows == 0) return LAS_OK;
    hipLaunchKernelGGL(pick_token_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, V, greedy ? 1 : 0, seed, tok);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

extern "C" int las_decoder_fwd(const las_dec_dims* d, const las_dec_params* p, const float* enc, const float* psi,
                               const int32_t* enc_len, const int64_t* y, int Ly, const uint8_t* step_mode,
                               unsigned seed, las_dec_state* st_, void* stream) {
    return decoder_run(d, p, enc, psi, enc_len, y, Ly, step_mode, seed, st_, stream, false);
}

extern "C" size_t las_decoder_pk_workspace_bytes(const las_dec_dims* d) { return las_dec_pk_fwd_ws_bytes(d); }

extern "C" int las_decoder_step(const las_dec_dims* d, const las_dec_params* p, const float* enc, const float* psi,
                                const int32_t* enc_len, las_dec_state* st_, float* logits, void* stream) {
    LAS_CHECK_ARG(d && p && st_ && logits && d->L == 1 && p->w_char && p->b_char && st_->tok);
    static const uint8_t teacher_none = 2;                  // no teacher tensor: step 0 feeds state->tok
    int rc = decoder_run(d, p, enc, psi, enc_len, nullptr, 0, &teacher_none, 0, st_, stream, true);
    if (rc) return rc;
    const long BC = (long)d->B * d->C;
    const float* htop = st_->hs + ((long)(d->NL - 1) * 2 + 1) * BC;
    return las_skinny_launch(d->prec, htop, d->C, p->w_char, d->C, d->C, nullptr, 0, nullptr, 0, 0, nullptr, 0, nullptr, 0, 0,
                             d->B, d->V, p->b_char, nullptr, 0, logits, d->V, 0, 0, nullptr, nullptr, nullptr, nullptr,
                             (hipStream_t)stream);
}
