// Global-norm gradient clipping + fused Adam / Adadelta over ONE flat parameter vector.
//
// Replaces torch.nn.utils.clip_grad_norm_(params, 5) + the NaN guard + optimizer.step() at reference
// src/solver.py:178-182 (torch.optim.Adam / Adadelta(lr, eps=1e-8) or apex FusedAdam, solver.py:101-106).
// All parameters / gradients / moments live in flat fp32 buffers, so this is 2 + 1 launches per step with no
// host synchronisation: the NaN-skip decision and the step counter stay on the device.
// HBM-bound: Adam reads p,g,m,v and writes p,m,v (+ zeroed g) = 32 B per parameter.
#include "las_common.h"

namespace {

constexpr int NPART = 1024;

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ part) {
    __shared__ float red[32];
    float s = 0.f;
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * 256 * 4;
    for (; i + 3 < n; i += stride) {
        const float4 v = *(const float4*)(g + i);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (i < n) for (long k = i; k < n && k < i + 4; ++k) s += g[k] * g[k];
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// out[0] = ||gscale*g||, out[1] = clip coefficient (includes gscale), out[2] = 1 if the step must be skipped (NaN)
__global__ __launch_bounds__(256) void norm_finish_kernel(const float* __restrict__ part, int np, float gscale,
                                                          float max_norm, float* __restrict__ out,
                                                          int32_t* __restrict__ step) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) s += (double)part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) {
        const float total = (float)(sqrt(red[0]) * (double)gscale);
        const bool bad = isnan(total);
        float coef = max_norm / (total + 1e-6f);           // torch.nn.utils.clip_grad_norm_
        if (coef > 1.f) coef = 1.f;
        out[0] = total;
        out[1] = bad ? 0.f : coef * gscale;
        out[2] = bad ? 1.f : 0.f;
        if (!bad) step[0] += 1;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                   const float* __restrict__ nrm, const int32_t* __restrict__ step,
                                                   int zero_grad) {
    const bool skip = nrm[2] != 0.f;
    const float coef = nrm[1];
    const double t = (double)step[0];
    const float step_size = (float)((double)lr / (1.0 - pow((double)b1, t)));
    const float bc2s = (float)sqrt(1.0 - pow((double)b2, t));
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        if (!skip) {
            const float gi = g[i] * coef;
            const float mi = b1 * m[i] + (1.f - b1) * gi;
            const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
            m[i] = mi; v[i] = vi;
            p[i] -= step_size * (mi / (sqrtf(vi) / bc2s + eps));
        }
        if (zero_grad) g[i] = 0.f;
    }
}

__global__ __launch_bounds__(256) void adadelta_kernel(float* __restrict__ p, float* __restrict__ g,
                                                       float* __restrict__ sq, float* __restrict__ acc, long n, float lr,
                                                       float rho, float eps, const float* __restrict__ nrm,
                                                       int zero_grad) {
    const bool skip = nrm[2] != 0.f;
    const float coef = nrm[1];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        if (!skip) {
            const float gi = g[i] * coef;
            const float s = rho * sq[i] + (1.f - rho) * gi * gi;
            const float delta = sqrtf(acc[i] + eps) / sqrtf(s + eps) * gi;
            sq[i] = s;
            acc[i] = rho * acc[i] + (1.f - rho) * delta * delta;
            p[i] -= lr * delta;
        }
        if (zero_grad) g[i] = 0.f;
    }
}

unsigned grid_for(long n) { long b = (n + 255) / 256; return (unsigned)(b > 2048 ? 2048 : (b < 1 ? 1 : b)); }

}  // namespace

extern "C" size_t las_grad_norm_workspace_bytes(void) { return sizeof(float) * NPART; }

extern "C" int las_grad_norm(const float* g, int64_t n, float gscale, float max_norm, void* workspace, float* out3,
                             int32_t* step_dev, void* stream) {
    LAS_CHECK_ARG(g && workspace && out3 && step_dev && n > 0);
    LAS_CHECK_ARG((((uintptr_t)g) & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    long blocks = (n / 4 + 255) / 256;
    const int np = (int)(blocks > NPART ? NPART : (blocks < 1 ? 1 : blocks));
    hipLaunchKernelGGL(sumsq_kernel, dim3(np), dim3(256), 0, st, g, (long)n, (float*)workspace);
    LAS_LAUNCH_OK();
    hipLaunchKernelGGL(norm_finish_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, np, gscale, max_norm, out3,
                       step_dev);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

extern "C" int las_adam_step(float* p, float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                             const float* norm3, const int32_t* step_dev, int zero_grad, void* stream) {
    LAS_CHECK_ARG(p && g && m && v && norm3 && step_dev && n > 0);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, lr, b1, b2, eps,
                       norm3, step_dev, zero_grad);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

extern "C" int las_adadelta_step(float* p, float* g, float* sq, float* acc, int64_t n, float lr, float rho, float eps,
                                 const float* norm3, int zero_grad, void* stream) {
    LAS_CHECK_ARG(p && g && sq && acc && norm3 && n > 0);
    hipLaunchKernelGGL(adadelta_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, sq, acc, (long)n, lr, rho,
                       eps, norm3, zero_grad);
    LAS_LAUNCH_OK();
    return LAS_OK;
}
