// CTC loss with fused log-softmax: row pass -> LDS-staged alpha/beta wavefront scan -> gradient row pass.
//
// Replaces F.log_softmax + torch.nn.CTCLoss(blank=0) at reference src/solver.py:93,160.
//
// HBM traffic per (b,t) row of V logits: read once in ctc_rowpass_fwd (log-sum-exp + gather of the
// <= L+1 label columns into a compact lattice-input tensor lpc[B,T,L+1]), read once more and written
// once in ctc_rowpass_bwd.  The sequential scans touch only lpc / alpha / beta (coalesced rows of
// L+1 resp. 2L+1 floats per time step), never the V-wide tensor.
//
// Workspace: lse[B*T] | lpc[B*T*(L+1)] | ahat,bhat[B*T*(2L+1)] (normalised rows) | Ca,Cb[B*T] fp64 offsets |
// nlld[B] fp64.
#include "las_common.h"

namespace {

constexpr int ROW_THREADS = 256;

struct CtcWs {
    float* lse;
    float* lpc;
    float* ahat;
    float* bhat;
    double* Ca;
    double* Cb;
    double* nlld;
    size_t bytes;
};
inline CtcWs carve(void* ws, int B, int T, int L) {
    CtcWs w;
    size_t o = 0;
    char* p = (char*)ws;
    w.lse = (float*)(p + o);  o += las_align(sizeof(float) * (size_t)B * T);
    w.lpc = (float*)(p + o);  o += las_align(sizeof(float) * (size_t)B * T * (L + 1));
    w.ahat = (float*)(p + o); o += las_align(sizeof(float) * (size_t)B * T * (2 * L + 1));
    w.bhat = (float*)(p + o); o += las_align(sizeof(float) * (size_t)B * T * (2 * L + 1));
    w.Ca = (double*)(p + o);  o += las_align(sizeof(double) * (size_t)B * T);
    w.Cb = (double*)(p + o);  o += las_align(sizeof(double) * (size_t)B * T);
    w.nlld = (double*)(p + o); o += las_align(sizeof(double) * (size_t)B);
    w.bytes = o;
    return w;
}

// One block per (b,t) row: lse over V (coalesced, float4 when possible), then gather label columns.
__global__ __launch_bounds__(ROW_THREADS) void ctc_rowpass_fwd(const float* __restrict__ logits,
                                                                const int32_t* __restrict__ label,
                                                                const int32_t* __restrict__ enc_len,
                                                                const int32_t* __restrict__ tgt_len, int T, int V, int L,
                                                                int blank, float* __restrict__ lse_out,
                                                                float* __restrict__ lpc) {
    __shared__ float red[32];
    const int row = blockIdx.x, b = row / T, t = row % T;
    if (t >= enc_len[b]) return;
    const float* x = logits + (size_t)row * V;
    float m = -INFINITY;
    const bool vec = ((V & 3) == 0);
    if (vec) {
        const float4* x4 = (const float4*)x;
        for (int i = threadIdx.x; i < V / 4; i += ROW_THREADS) {
            float4 v = x4[i];
            m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
        }
    } else {
        for (int i = threadIdx.x; i < V; i += ROW_THREADS) m = fmaxf(m, x[i]);
    }
    m = block_max(m, red);
    float s = 0.f;
    if (vec) {
        const float4* x4 = (const float4*)x;
        for (int i = threadIdx.x; i < V / 4; i += ROW_THREADS) {
            float4 v = x4[i];
            s += expf(v.x - m) + expf(v.y - m) + expf(v.z - m) + expf(v.w - m);
        }
    } else {
        for (int i = threadIdx.x; i < V; i += ROW_THREADS) s += expf(x[i] - m);
    }
    s = block_sum(s, red);
    const float lse = m + logf(s);
    if (threadIdx.x == 0) lse_out[row] = lse;
    const int n = tgt_len[b];
    float* o = lpc + (size_t)row * (L + 1);
    for (int i = threadIdx.x; i <= n; i += ROW_THREADS) {
        const int c = (i == 0) ? blank : label[(size_t)b * L + i - 1];
        o[i] = x[c] - lse;
    }
}

__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c));
    if (m == -INFINITY) return -INFINITY;
    return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

// float max into an LDS word holding a float (works for mixed signs and -inf).
__device__ __forceinline__ void lds_fmax(float* addr, float v) {
    if (v >= 0.f) atomicMax((int*)addr, __float_as_int(v));
    else atomicMin((unsigned*)addr, __float_as_uint(v));
}

// One block per (utterance, direction).  dir 0: alpha forward in time; dir 1: beta backward in time.
// The lattice row lives in LDS (double buffered); one barrier per time step; the next step's lattice
// inputs are prefetched before the barrier.
// Numerics: the row is kept normalised, a^_t(s) = a_t(s) - C_t, where C_t (fp64, identical in every
// thread) accumulates the block maximum of the previous row (computed one step late through an LDS
// atomic, so it costs no extra barrier).  |a^| stays O(100) instead of O(T * |lp|), so fp32 rounding
// does not random-walk with T; the emitted log_alpha/log_beta = a^ + C_t are rounded once.
template <int SPT>   // states per thread
__global__ void ctc_scan(const float* __restrict__ lpc, const int32_t* __restrict__ label,
                         const int32_t* __restrict__ enc_len, const int32_t* __restrict__ tgt_len, int T, int L,
                         int blank, int do_alpha, int do_beta, float* __restrict__ nll,
                         float* __restrict__ log_alpha, float* __restrict__ ahat, float* __restrict__ bhat,
                         double* __restrict__ Ca, double* __restrict__ Cb, double* __restrict__ nlld) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int S = 2 * L + 1;
    const int b = blockIdx.x;
    const int dir = do_alpha ? (do_beta ? (int)blockIdx.y : 0) : 1;
    const int Tb = enc_len[b], n = tgt_len[b], Sn = 2 * n + 1;
    float* out = (dir == 0) ? log_alpha + (size_t)b * T * S : nullptr;   // public lattice (alpha only)
    float* hat = (dir == 0 ? ahat : bhat) + (size_t)b * T * S;           // normalised rows for the gradient
    double* Cst = (dir == 0 ? Ca : Cb) + (size_t)b * T;
    // rows outside the utterance: -inf
    if (out) for (int i = threadIdx.x; i < (T - Tb) * S; i += blockDim.x) out[(size_t)Tb * S + i] = -INFINITY;
    if (Tb <= 0) {
        if (dir == 0 && threadIdx.x == 0) { nll[b] = INFINITY; nlld[b] = (double)INFINITY; }
        return;
    }
    float* buf0 = lds;
    float* buf1 = lds + S + 2;          // +2: room for the two out-of-range neighbours
    float* mx = lds + 2 * (S + 2);      // 3 rotating slots for the delayed row maximum
    // per-state constants
    int idx[SPT];       // column of lpc for state s
    bool skip[SPT];     // may take the s-2 (alpha) / s+2 (beta) transition
    bool live[SPT];
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        const int s = threadIdx.x + k * blockDim.x;
        live[k] = s < Sn;
        idx[k] = (s & 1) ? (s >> 1) + 1 : 0;
        skip[k] = false;
        if (live[k] && (s & 1)) {
            const int li = s >> 1;
            if (dir == 0) {
                if (li >= 1) skip[k] = label[(size_t)b * L + li] != label[(size_t)b * L + li - 1];
            } else {
                if (li + 1 < n) skip[k] = label[(size_t)b * L + li] != label[(size_t)b * L + li + 1];
            }
        }
    }
    const float* lp_b = lpc + (size_t)b * T * (L + 1);
    const int t0 = dir == 0 ? 0 : Tb - 1, dt = dir == 0 ? 1 : -1;
    float lp[SPT], lpn[SPT];
#pragma unroll
    for (int k = 0; k < SPT; ++k) lp[k] = live[k] ? lp_b[(size_t)t0 * (L + 1) + idx[k]] : 0.f;
    if (threadIdx.x == 0) {
        buf0[0] = -INFINITY; buf1[0] = -INFINITY; buf0[S + 1] = -INFINITY; buf1[S + 1] = -INFINITY;
        mx[0] = -INFINITY; mx[1] = -INFINITY; mx[2] = -INFINITY;
    }
    __syncthreads();
    // init row (step 0): C_0 = 0
    float* prev = buf0;
    float* cur = buf1;
    float vmax = -INFINITY;
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        const int s = threadIdx.x + k * blockDim.x;
        if (s < S) {
            float v = -INFINITY;
            if (live[k]) {
                if (dir == 0) { if (s <= 1) v = lp[k]; }
                else          { if (s >= Sn - 2) v = lp[k]; }
            }
            prev[s + 1] = v;            // stored with +1 offset; prev[0] and prev[S+1] are guards
            if (out) out[(size_t)t0 * S + s] = v;
            hat[(size_t)t0 * S + s] = v;
            vmax = fmaxf(vmax, v);
        }
    }
    vmax = wave_max(vmax);
    if ((threadIdx.x & 63) == 0 && vmax > -INFINITY) lds_fmax(&mx[0], vmax);
    if (threadIdx.x == 0) Cst[t0] = 0.0;
    __syncthreads();
    double C = 0.0;
    for (int step = 1; step < Tb; ++step) {
        const int t = t0 + dt * step;
#pragma unroll
        for (int k = 0; k < SPT; ++k) lpn[k] = live[k] ? lp_b[(size_t)t * (L + 1) + idx[k]] : 0.f;
        float m = mx[(step - 1) % 3];                      // max of the previous (normalised) row
        if (!(m > -INFINITY)) m = 0.f;                     // dead lattice: nothing to normalise
        C += (double)m;
        if (threadIdx.x == 0) { mx[(step + 1) % 3] = -INFINITY; Cst[t] = C; }
        vmax = -INFINITY;
#pragma unroll
        for (int k = 0; k < SPT; ++k) {
            const int s = threadIdx.x + k * blockDim.x;
            if (s < S) {
                float v = -INFINITY;
                if (live[k]) {
                    float a0 = prev[s + 1], a1, a2 = -INFINITY;
                    if (dir == 0) {
                        a1 = prev[s];                                   // s-1 (guard at s=0)
                        if (skip[k]) a2 = prev[s - 1];                  // s-2
                    } else {
                        a1 = (s + 1 < Sn) ? prev[s + 2] : -INFINITY;    // s+1
                        // s -> s+2 allowed iff ext[s+2] is a label different from ext[s] (odd s only)
                        if (s + 2 < Sn && skip[k]) a2 = prev[s + 3];
                    }
                    v = lse3(a0 - m, a1 - m, a2 - m) + lpn[k];
                }
                cur[s + 1] = v;
                if (out) out[(size_t)t * S + s] = (float)((double)v + C);
                hat[(size_t)t * S + s] = v;
                vmax = fmaxf(vmax, v);
            }
        }
        vmax = wave_max(vmax);
        if ((threadIdx.x & 63) == 0 && vmax > -INFINITY) lds_fmax(&mx[step % 3], vmax);
        __syncthreads();
        float* tmp = prev; prev = cur; cur = tmp;
    }
    if (dir == 0 && threadIdx.x == 0) {
        const float a = prev[Sn - 1 + 1];
        const float c = (Sn > 1) ? prev[Sn - 2 + 1] : -INFINITY;
        const double nd = -((double)lse3(a, c, -INFINITY) + C);
        nll[b] = (float)nd;
        nlld[b] = nd;
    }
}

// One block per (b,t) row: occupancies gamma_t(s) scattered into an LDS row of V floats, then
// grad = gscale * (softmax - occ), written coalesced.
__global__ __launch_bounds__(ROW_THREADS) void ctc_rowpass_bwd(
    const float* __restrict__ logits, const int32_t* __restrict__ label, const int32_t* __restrict__ enc_len,
    const int32_t* __restrict__ tgt_len, int T, int V, int L, int blank, const float* __restrict__ lse_in,
    const float* __restrict__ lpc, const float* __restrict__ nll, const float* __restrict__ ahat,
    const float* __restrict__ bhat, const double* __restrict__ Ca, const double* __restrict__ Cb,
    const double* __restrict__ nlld, const float* __restrict__ gscale, float* __restrict__ grad) {
    extern __shared__ __attribute__((aligned(16))) float occ[];
    const int row = blockIdx.x, b = row / T, t = row % T;
    float* g = grad + (size_t)row * V;
    const bool vec = ((V & 3) == 0);
    if (t >= enc_len[b]) {
        if (vec) for (int i = threadIdx.x; i < V / 4; i += ROW_THREADS) ((float4*)g)[i] = make_float4(0, 0, 0, 0);
        else for (int i = threadIdx.x; i < V; i += ROW_THREADS) g[i] = 0.f;
        return;
    }
    const float nl = nll[b];
    if (!(nl < INFINITY)) {               // infeasible alignment: ATen yields NaN (zero_infinity=False)
        const float q = __builtin_nanf("");
        for (int i = threadIdx.x; i < V; i += ROW_THREADS) g[i] = q;
        return;
    }
    for (int i = threadIdx.x; i < V; i += ROW_THREADS) occ[i] = 0.f;
    __syncthreads();
    const int S = 2 * L + 1, n = tgt_len[b], Sn = 2 * n + 1;
    const float* al = ahat + (size_t)row * S;
    const float* be = bhat + (size_t)row * S;
    const float off = (float)(Ca[row] + Cb[row] + nlld[b]);   // fp64 sum of the big offsets; result is O(1)
    const float* lp = lpc + (size_t)row * (L + 1);
    for (int s = threadIdx.x; s < Sn; s += ROW_THREADS) {
        const float ab = al[s] + be[s];
        if (ab > -INFINITY) {
            const int li = s >> 1;
            const int c = (s & 1) ? label[(size_t)b * L + li] : blank;
            const float l = (s & 1) ? lp[li + 1] : lp[0];
            atomicAdd(&occ[c], expf(ab - l + off));
        }
    }
    __syncthreads();
    const float lse = lse_in[row], gs = gscale[b];
    const float* x = logits + (size_t)row * V;
    if (vec) {
        for (int i = threadIdx.x; i < V / 4; i += ROW_THREADS) {
            const float4 v = ((const float4*)x)[i];
            const float4 o = ((const float4*)occ)[i];
            float4 r;
            r.x = gs * (expf(v.x - lse) - o.x);
            r.y = gs * (expf(v.y - lse) - o.y);
            r.z = gs * (expf(v.z - lse) - o.z);
            r.w = gs * (expf(v.w - lse) - o.w);
            ((float4*)g)[i] = r;
        }
    } else {
        for (int i = threadIdx.x; i < V; i += ROW_THREADS) g[i] = gs * (expf(x[i] - lse) - occ[i]);
    }
}

template <int SPT>
int launch_scan(const CtcWs& w, const int32_t* label, const int32_t* enc_len, const int32_t* tgt_len, int B, int T,
                int L, int blank, int do_alpha, int do_beta, float* nll, float* log_alpha, int threads,
                hipStream_t st) {
    const int S = 2 * L + 1;
    const size_t lds = sizeof(float) * (2 * (S + 2) + 4);
    dim3 grid(B, (do_alpha && do_beta) ? 2 : 1);
    hipLaunchKernelGGL(ctc_scan<SPT>, grid, dim3(threads), lds, st, w.lpc, label, enc_len, tgt_len, T, L, blank,
                       do_alpha, do_beta, nll, log_alpha, w.ahat, w.bhat, w.Ca, w.Cb, w.nlld);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

int scan_dispatch(const CtcWs& w, const int32_t* label, const int32_t* enc_len, const int32_t* tgt_len, int B, int T,
                  int L, int blank, int do_alpha, int do_beta, float* nll, float* log_alpha, hipStream_t st) {
    const int S = 2 * L + 1;
    int threads = (S + 63) / 64 * 64;
    if (threads <= 1024) return launch_scan<1>(w, label, enc_len, tgt_len, B, T, L, blank, do_alpha, do_beta, nll, log_alpha, threads, st);
    if (S <= 2048) return launch_scan<2>(w, label, enc_len, tgt_len, B, T, L, blank, do_alpha, do_beta, nll, log_alpha, 1024, st);
    if (S <= 4096) return launch_scan<4>(w, label, enc_len, tgt_len, B, T, L, blank, do_alpha, do_beta, nll, log_alpha, 1024, st);
    return LAS_E_UNSUPPORTED;
}

}  // namespace

extern "C" size_t las_ctc_workspace_bytes(int B, int T, int V, int L) {
    (void)V;
    return carve(nullptr, B, T, L).bytes;
}

extern "C" int las_ctc_loss_fwd(const float* logits, const int32_t* label, const int32_t* enc_len,
                                const int32_t* tgt_len, int B, int T, int V, int L, int blank, float* nll,
                                float* log_alpha, void* workspace, size_t ws_bytes, void* stream) {
    LAS_CHECK_ARG(logits && label && enc_len && tgt_len && nll && log_alpha && workspace);
    LAS_CHECK_ARG(B > 0 && T > 0 && V > 1 && L > 0 && blank >= 0 && blank < V);
    CtcWs w = carve(workspace, B, T, L);
    if (ws_bytes < w.bytes) return LAS_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ctc_rowpass_fwd, dim3(B * T), dim3(ROW_THREADS), 0, st, logits, label, enc_len, tgt_len, T, V,
                       L, blank, w.lse, w.lpc);
    LAS_LAUNCH_OK();
    return scan_dispatch(w, label, enc_len, tgt_len, B, T, L, blank, 1, 0, nll, log_alpha, st);
}

extern "C" int las_ctc_loss_bwd(const float* logits, const int32_t* label, const int32_t* enc_len,
                                const int32_t* tgt_len, int B, int T, int V, int L, int blank, const float* nll,
                                const float* log_alpha, const float* gscale, float* grad_logits, void* workspace,
                                size_t ws_bytes, void* stream) {
    LAS_CHECK_ARG(logits && label && enc_len && tgt_len && nll && log_alpha && gscale && grad_logits && workspace);
    LAS_CHECK_ARG(B > 0 && T > 0 && V > 1 && L > 0 && blank >= 0 && blank < V);
    if ((size_t)V * sizeof(float) > 150 * 1024) return LAS_E_UNSUPPORTED;
    CtcWs w = carve(workspace, B, T, L);
    if (ws_bytes < w.bytes) return LAS_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int rc = scan_dispatch(w, label, enc_len, tgt_len, B, T, L, blank, 0, 1, nullptr, nullptr, st);
    if (rc) return rc;
    hipLaunchKernelGGL(ctc_rowpass_bwd, dim3(B * T), dim3(ROW_THREADS), sizeof(float) * (size_t)((V + 3) & ~3), st,
                       logits, label, enc_len, tgt_len, T, V, L, blank, w.lse, w.lpc, nll, w.ahat, w.bhat, w.Ca, w.Cb, w.nlld,
                       gscale, grad_logits);
    LAS_LAUNCH_OK();
    return LAS_OK;
}
