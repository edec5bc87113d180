// Device side of the joint CTC/attention beam search (reference src/asr.py:155-258, src/ctc.py): per-row
// log-softmax and top-k over the vocabulary, the CTC prefix scorer batched over (hypothesis, candidate) pairs, and the
// score combination.  The reference walks T' frames in numpy once per hypothesis and candidate on the host; here one
// thread owns one (hypothesis, candidate) lattice column pair and all beam x candidates pairs advance together.
#include "las_common.h"

namespace {

constexpr float LOGZERO = -100000000.0f;            // ctc.py:11

// float32 logaddexp as numpy's npy_logaddexpf (max + log1p(exp(-|d|))), on the hardware exp2/log2 units: the T'-long
// chain of a (hypothesis, candidate) pair is three of these per frame, strictly sequential, so their latency IS the
// kernel time.  log(1 + e) instead of log1p(e) differs by < 6e-8 absolute (e < 1), far below the float32 resolution of
// the log-probabilities being summed.
__device__ __forceinline__ float logaddexp_np(float x, float y) {
    const float m = fmaxf(x, y), d = -fabsf(x - y);
    return m + __logf(1.f + __expf(d));
}

__global__ __launch_bounds__(256) void log_softmax_rows_kernel(const float* __restrict__ x, int V, float* __restrict__ out) {
    __shared__ float red[32];
    const float* p = x + (long)blockIdx.x * V;
    float* o = out + (long)blockIdx.x * V;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < V; i += 256) m = fmaxf(m, p[i]);
    m = block_max(m, red);
    float s = 0.f;
    for (int i = threadIdx.x; i < V; i += 256) s += expf(p[i] - m);
    s = block_sum(s, red);
    const float lse = m + logf(s);
    for (int i = threadIdx.x; i < V; i += 256) o[i] = p[i] - lse;
}

// k largest of each row, descending, ties to the lower index; one workgroup per row, the row staged in LDS
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ x, int V, int k,
                                                        float* __restrict__ vals, int32_t* __restrict__ idx) {
    extern __shared__ float row[];
    __shared__ float bv[256];
    __shared__ int bi[256];
    const float* p = x + (long)blockIdx.x * V;
    for (int i = threadIdx.x; i < V; i += 256) row[i] = p[i];
    __syncthreads();
    for (int j = 0; j < k; ++j) {
        float v = -INFINITY;
        int ix = V;
        for (int i = threadIdx.x; i < V; i += 256) {
            const float r = row[i];
            if (r > v || (r == v && i < ix) || (ix == V && r != r)) { v = r; ix = i; }
        }
        bv[threadIdx.x] = v; bi[threadIdx.x] = ix;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) {
                const float ov = bv[threadIdx.x + o]; const int oi = bi[threadIdx.x + o];
                if (ov > bv[threadIdx.x] || (ov == bv[threadIdx.x] && oi < bi[threadIdx.x])) { bv[threadIdx.x] = ov; bi[threadIdx.x] = oi; }
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const int w = min(bi[0], V - 1);
            vals[(long)blockIdx.x * k + j] = bv[0];
            idx[(long)blockIdx.x * k + j] = w;
            row[w] = -INFINITY;
        }
        __syncthreads();
    }
}

// r0[t] = {LOGZERO, sum_{u<=t} lp[u][blank]}   (CTCPrefixScore.init_state, ctc.py:19-27)
__global__ void ctc_prefix_init_kernel(const float* __restrict__ lp, int T, int V, float* __restrict__ r0) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float acc = 0.f;
    for (int t = 0; t < T; ++t) {
        acc = t == 0 ? lp[0] : acc + lp[(long)t * V];
        r0[2 * t] = LOGZERO;
        r0[2 * t + 1] = acc;
    }
}

// CTCPrefixScore.cheap_compute (ctc.py:65-101) for every (hypothesis n, candidate j): thread = pair.
__global__ __launch_bounds__(64) void ctc_prefix_score_kernel(const float* __restrict__ lp, int T, int V,
                                                              const float* __restrict__ r_prev, const int32_t* __restrict__ last,
                                                              const int32_t* __restrict__ plen, const int32_t* __restrict__ cand,
                                                              int N, int K, float* __restrict__ psi, float* __restrict__ r_out) {
    const int pair = blockIdx.x * 64 + threadIdx.x;
    if (pair >= N * K) return;
    const int n = pair / K;
    const int c = min(max(cand[pair], 0), V - 1);
    const int len = plen[n], lc = len > 0 ? last[n] : 0, start = max(1, len);
    const float* rp = r_prev + (long)n * T * 2;
    float* ro = r_out + (long)pair * T * 2;
    for (int t = 0; t < min(start, T); ++t) { ro[2 * t] = LOGZERO; ro[2 * t + 1] = LOGZERO; }
    float rn = LOGZERO, rb = LOGZERO;                // r[start-1][0], r[start-1][1]
    if (len == 0) { rn = lp[c]; ro[0] = rn; }        // empty prefix: r[0][0] = x[0][c]  (start = 1)
    float p = rn;                                    // psi = r[start-1][0]
    // the operands of the recurrence (r_prev, lp) do not depend on it: 8 frames are requested together, then consumed
    constexpr int U = 8;
    for (int t0 = start; t0 < T; t0 += U) {
        float2 rv[U];
        float xc[U], xb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = min(t0 + u, T - 1);
            rv[u] = *(const float2*)(rp + 2 * (t - 1));
            xc[u] = lp[(long)t * V + c];
            xb[u] = lp[(long)t * V];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + u;
            if (t < T) {
                const float pb = c == lc ? LOGZERO : rv[u].y;
                const float phi = logaddexp_np(rv[u].x, pb);
                const float nn = logaddexp_np(rn, phi) + xc[u];
                const float nb = logaddexp_np(rb, rn) + xb[u];
                p = logaddexp_np(p, phi + xc[u]);
                rn = nn; rb = nb;
                *(float2*)(ro + 2 * t) = make_float2(nn, nb);
            }
        }
    }
    psi[pair] = p;
}

// cur[n][:] = (1-lam)*cur + lam*hack, hack = -1e6 except hack[cand[j]] = psi[n][j] - prev_ctc[n]; then cur[n][0] = -1e7
// (asr.py:218-229)
__global__ __launch_bounds__(256) void beam_combine_kernel(float* __restrict__ cur, int V, const int32_t* __restrict__ cand,
                                                           const float* __restrict__ psi, const float* __restrict__ prev_ctc,
                                                           int K, float lam) {
    extern __shared__ float co[];                    // [K] attention log-probs of the candidates
    const int n = blockIdx.x;
    float* row = cur + (long)n * V;
    for (int j = threadIdx.x; j < K; j += 256) co[j] = row[min(max(cand[(long)n * K + j], 0), V - 1)];
    __syncthreads();
    for (int i = threadIdx.x; i < V; i += 256) row[i] = (1.f - lam) * row[i] + lam * -1000000.0f;
    __syncthreads();
    for (int j = threadIdx.x; j < K; j += 256) {
        const float ctc_char = psi[(long)n * K + j] - prev_ctc[n];
        row[min(max(cand[(long)n * K + j], 0), V - 1)] = (1.f - lam) * co[j] + lam * ctc_char;
    }
    __syncthreads();
    if (threadIdx.x == 0) row[0] = -10000000.0f;
}

}  // namespace

extern "C" int las_log_softmax_rows(const float* x, int R, int V, float* out, void* stream) {
    LAS_CHECK_ARG(x && out && R >= 0 && V > 0);
    if (R == 0) return LAS_OK;
    hipLaunchKernelGGL(log_softmax_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, x, V, out);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

extern "C" int las_topk_rows(const float* x, int R, int V, int k, float* vals, int32_t* idx, void* stream) {
    LAS_CHECK_ARG(x && vals && idx && R >= 0 && V > 0 && k > 0 && k <= V);
    if ((size_t)V * sizeof(float) > 60 * 1024) return LAS_E_UNSUPPORTED;
    if (R == 0) return LAS_OK;
    hipLaunchKernelGGL(topk_rows_kernel, dim3(R), dim3(256), sizeof(float) * V, (hipStream_t)stream, x, V, k, vals, idx);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

extern "C" int las_ctc_prefix_init(const float* lp, int T, int V, float* r0, void* stream) {
    LAS_CHECK_ARG(lp && r0 && T > 0 && V > 1);
    hipLaunchKernelGGL(ctc_prefix_init_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, lp, T, V, r0);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

extern "C" int las_ctc_prefix_score(const float* lp, int T, int V, const float* r_prev, const int32_t* last_tok,
                                    const int32_t* prefix_len, const int32_t* cand, int N, int K, float* psi, float* r_out,
                                    void* stream) {
    LAS_CHECK_ARG(lp && r_prev && last_tok && prefix_len && cand && psi && r_out && T > 0 && V > 1 && N >= 0 && K > 0);
    if (N == 0) return LAS_OK;
    hipLaunchKernelGGL(ctc_prefix_score_kernel, dim3((N * K + 63) / 64), dim3(64), 0, (hipStream_t)stream, lp, T, V, r_prev,
                       last_tok, prefix_len, cand, N, K, psi, r_out);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

extern "C" int las_beam_combine(float* cur, int N, int V, const int32_t* cand, const float* psi, const float* prev_ctc, int K,
                                float ctc_weight, void* stream) {
    LAS_CHECK_ARG(cur && cand && psi && prev_ctc && N >= 0 && V > 1 && K > 0 && K <= V);
    if (N == 0) return LAS_OK;
    hipLaunchKernelGGL(beam_combine_kernel, dim3(N), dim3(256), sizeof(float) * K, (hipStream_t)stream, cur, V, cand, psi,
                       prev_ctc, K, ctc_weight);
    LAS_LAUNCH_OK();
    return LAS_OK;
}
