// Attention cross-entropy with per-utterance normalisation, and the scalar reductions of the joint loss.
//
// Replaces CrossEntropyLoss(ignore_index=0, reduction='none') + sum_t / sum(y!=0) + batch mean at reference
// src/solver.py:90,149-155, and the 'mean' reduction of CTCLoss (nll_b / clamp(target_len_b, 1), batch mean,
// solver.py:93,160).  One pass over the logits computes the row log-sum-exp, the row loss and writes the
// gradient row; the reductions are deterministic (no float atomics).
#include "las_common.h"

namespace {

// one block per (b,t) row of logits [B][L][V]; label = y[b][t+1]
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, const long long* __restrict__ y,
                                                      int Ly, const int32_t* __restrict__ ntok, int B, int L, int V,
                                                      float gscale, float* __restrict__ rowloss,
                                                      float* __restrict__ dlogits) {
    __shared__ float red[32];
    const int row = blockIdx.x, b = row / L, t = row % L;
    const int label = (int)y[(long)b * Ly + t + 1];
    const float* x = logits + (long)row * V;
    float* g = dlogits ? dlogits + (long)row * V : nullptr;
    if (label == 0 || label >= V || label < 0) {            // ignore_index = 0
        if (threadIdx.x == 0) rowloss[row] = 0.f;
        if (g) for (int i = threadIdx.x; i < V; i += 256) g[i] = 0.f;
        return;
    }
    float m = -INFINITY;
    for (int i = threadIdx.x; i < V; i += 256) m = fmaxf(m, x[i]);
    m = block_max(m, red);
    float s = 0.f;
    for (int i = threadIdx.x; i < V; i += 256) s += expf(x[i] - m);
    s = block_sum(s, red);
    const float lse = m + logf(s);
    if (threadIdx.x == 0) rowloss[row] = lse - x[label];
    if (g) {
        const float w = gscale / ((float)B * (float)max(ntok[b], 1));
        for (int i = threadIdx.x; i < V; i += 256) g[i] = w * (expf(x[i] - lse) - (i == label ? 1.f : 0.f));
    }
}

// loss = mean_b( sum_t rowloss[b,t] / max(ntok_b,1) );  one block
__global__ __launch_bounds__(256) void ce_finish_kernel(const float* __restrict__ rowloss,
                                                        const int32_t* __restrict__ ntok, int B, int L,
                                                        float* __restrict__ loss) {
    __shared__ float red[32];
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
        float s = 0.f;
        for (int t = threadIdx.x; t < L; t += 256) s += rowloss[(long)b * L + t];
        s = block_sum(s, red);
        acc += s / (float)max(ntok[b], 1);
    }
    if (threadIdx.x == 0) loss[0] = acc / (float)B;
}

// out = mean_b x[b] / max(n[b],1)
__global__ void norm_mean_fwd_kernel(const float* __restrict__ x, const int32_t* __restrict__ n, int B,
                                     float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += x[b] / (float)max(n[b], 1);
        out[0] = acc / (float)B;
    }
}
// gx[b] = g * scale / (B * max(n[b],1))
__global__ void norm_mean_bwd_kernel(const float* __restrict__ g, float scale, const int32_t* __restrict__ n, int B,
                                     float* __restrict__ gx) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) gx[b] = g[0] * scale / ((float)B * (float)max(n[b], 1));
}

// x *= alpha[0] (device scalar)
__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, long n, const float* __restrict__ alpha) {
    const float a = alpha[0];
    if (a == 1.f) return;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] *= a;
}

// out[0] = wa*a[0] + wb*b[0]   (either pointer may be NULL)
__global__ void combine_kernel(const float* a, float wa, const float* b, float wb, float* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (a ? wa * a[0] : 0.f) + (b ? wb * b[0] : 0.f);
}


// pred[row] = argmax_v logits[row][v] (first maximum)
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ logits, int V, int32_t* __restrict__ pred) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    const float* x = logits + (long)blockIdx.x * V;
    float v = -INFINITY; int idx = 0;
    for (int i = threadIdx.x; i < V; i += 256) if (x[i] > v) { v = x[i]; idx = i; }
    bv[threadIdx.x] = v; bi[threadIdx.x] = idx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float ov = bv[threadIdx.x + o]; const int oi = bi[threadIdx.x + o];
            if (ov > bv[threadIdx.x] || (ov == bv[threadIdx.x] && oi < bi[threadIdx.x])) { bv[threadIdx.x] = ov; bi[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) pred[blockIdx.x] = bi[0];
}

// mean over utterances of token accuracy up to the first 0 label (reference postprocess.py:121-133)
__global__ void token_acc_kernel(const int32_t* __restrict__ pred, const long long* __restrict__ y, int Ly, int B, int L,
                                 float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
        int correct = 0, total = 0;
        for (int t = 0; t < L; ++t) {
            const long long l = y[(long)b * Ly + t + 1];
            if (l == 0) break;
            correct += (pred[(long)b * L + t] == (int)l);
            ++total;
        }
        acc += (float)correct / (float)max(total, 1);
    }
    out[0] = acc / (float)B;
}

}  // namespace

extern "C" int las_ce_loss(const float* logits, const int64_t* y, int Ly, const int32_t* ntok, int B, int L, int V,
                           float gscale, float* rowloss, float* loss, float* dlogits, void* stream) {
    LAS_CHECK_ARG(logits && y && ntok && rowloss && loss && B > 0 && L > 0 && V > 1 && Ly >= L + 1);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ce_rows_kernel, dim3(B * L), dim3(256), 0, st, logits, (const long long*)y, Ly, ntok, B, L, V, gscale,
                       rowloss, dlogits);
    LAS_LAUNCH_OK();
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, st, rowloss, ntok, B, L, loss);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

extern "C" int las_norm_mean_fwd(const float* x, const int32_t* n, int B, float* out, void* stream) {
    LAS_CHECK_ARG(x && n && out && B > 0);
    hipLaunchKernelGGL(norm_mean_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, x, n, B, out);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

extern "C" int las_norm_mean_bwd(const float* g, float scale, const int32_t* n, int B, float* gx, void* stream) {
    LAS_CHECK_ARG(g && n && gx && B > 0);
    hipLaunchKernelGGL(norm_mean_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, g, scale, n, B, gx);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

extern "C" int las_scale_dev(float* x, int64_t n, const float* alpha, void* stream) {
    LAS_CHECK_ARG(x && alpha && n >= 0);
    if (n == 0) return LAS_OK;
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long)n, alpha);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

extern "C" int las_combine2(const float* a, float wa, const float* b, float wb, float* out, void* stream) {
    LAS_CHECK_ARG(out && (a || b));
    hipLaunchKernelGGL(combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, wa, b, wb, out);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

extern "C" int las_argmax_rows(const float* logits, int rows, int V, int32_t* pred, void* stream) {
    LAS_CHECK_ARG(logits && pred && rows > 0 && V > 0);
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, V, pred);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

extern "C" int las_token_acc(const int32_t* pred, const int64_t* y, int Ly, int B, int L, float* out, void* stream) {
    LAS_CHECK_ARG(pred && y && out && B > 0 && L > 0 && Ly >= L + 1);
    hipLaunchKernelGGL(token_acc_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, pred, (const long long*)y, Ly, B, L, out);
    LAS_LAUNCH_OK();
    return LAS_OK;
}
