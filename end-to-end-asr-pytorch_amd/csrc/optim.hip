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

// One element of the update; shared by the float4 body and the scalar tail of both kernels.
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float coef, float b1, float b2, float eps,
                                         float step_size, float bc2s) {
    const float gi = g * coef;
    const float mi = b1 * m + (1.f - b1) * gi;
    const float vi = b2 * v + (1.f - b2) * gi * gi;
    m = mi; v = vi;
    p -= step_size * (mi / (sqrtf(vi) / bc2s + eps));
}
__device__ __forceinline__ void adadelta_one(float& p, float g, float& sq, float& acc, float coef, float lr, float rho, float eps) {
    const float gi = g * coef;
    const float s = rho * sq + (1.f - rho) * gi * gi;
    const float delta = sqrtf(acc + eps) / sqrtf(s + eps) * gi;
    sq = s;
    acc = rho * acc + (1.f - rho) * delta * delta;
    p -= lr * delta;
}

// 16 bytes per lane per stream (HBM-bound: 4 streams read, 3-4 written); the flat vectors are 256-byte aligned.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                   const float* __restrict__ nrm, const int32_t* __restrict__ step,
                                                   int zero_grad, bf16_t* __restrict__ p16) {
    const bool skip = nrm[2] != 0.f;
    const float coef = nrm[1];
    const double t = (double)step[0];
    const float step_size = (float)((double)lr / (1.0 - pow((double)b1, t)));
    const float bc2s = (float)sqrt(1.0 - pow((double)b2, t));
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        if (!skip) {
            float4 pp = ((float4*)p)[i], mm = ((float4*)m)[i], vv = ((float4*)v)[i];
            const float4 gg = ((const float4*)g)[i];
            adam_one(pp.x, gg.x, mm.x, vv.x, coef, b1, b2, eps, step_size, bc2s);
            adam_one(pp.y, gg.y, mm.y, vv.y, coef, b1, b2, eps, step_size, bc2s);
            adam_one(pp.z, gg.z, mm.z, vv.z, coef, b1, b2, eps, step_size, bc2s);
            adam_one(pp.w, gg.w, mm.w, vv.w, coef, b1, b2, eps, step_size, bc2s);
            ((float4*)p)[i] = pp; ((float4*)m)[i] = mm; ((float4*)v)[i] = vv;
            if (p16) store4_ct(p16 + 4 * i, pp.x, pp.y, pp.z, pp.w);      // bf16 shadow of the weights (GEMM operands)
        }
        if (zero_grad) ((float4*)g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        if (!skip) { adam_one(p[i], g[i], m[i], v[i], coef, b1, b2, eps, step_size, bc2s); if (p16) p16[i] = f2bf(p[i]); }
        if (zero_grad) g[i] = 0.f;
    }
}

__global__ __launch_bounds__(256) void adadelta_kernel(float* __restrict__ p, float* __restrict__ g,
                                                       float* __restrict__ sq, float* __restrict__ acc, long n, float lr,
                                                       float rho, float eps, const float* __restrict__ nrm,
                                                       int zero_grad, bf16_t* __restrict__ p16) {
    const bool skip = nrm[2] != 0.f;
    const float coef = nrm[1];
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        if (!skip) {
            float4 pp = ((float4*)p)[i], ss = ((float4*)sq)[i], aa = ((float4*)acc)[i];
            const float4 gg = ((const float4*)g)[i];
            adadelta_one(pp.x, gg.x, ss.x, aa.x, coef, lr, rho, eps);
            adadelta_one(pp.y, gg.y, ss.y, aa.y, coef, lr, rho, eps);
            adadelta_one(pp.z, gg.z, ss.z, aa.z, coef, lr, rho, eps);
            adadelta_one(pp.w, gg.w, ss.w, aa.w, coef, lr, rho, eps);
            ((float4*)p)[i] = pp; ((float4*)sq)[i] = ss; ((float4*)acc)[i] = aa;
            if (p16) store4_ct(p16 + 4 * i, pp.x, pp.y, pp.z, pp.w);
        }
        if (zero_grad) ((float4*)g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        if (!skip) { adadelta_one(p[i], g[i], sq[i], acc[i], coef, lr, rho, eps); if (p16) p16[i] = f2bf(p[i]); }
        if (zero_grad) g[i] = 0.f;
    }
}

unsigned grid_for(long n) { long b = (n / 4 + 255) / 256; return (unsigned)(b > 2048 ? 2048 : (b < 1 ? 1 : b)); }

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
                             const float* norm3, const int32_t* step_dev, int zero_grad, void* p_bf16, void* stream) {
    LAS_CHECK_ARG(p && g && m && v && norm3 && step_dev && n > 0);
    LAS_CHECK_ARG(((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, lr, b1, b2, eps,
                       norm3, step_dev, zero_grad, (bf16_t*)p_bf16);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

extern "C" int las_adadelta_step(float* p, float* g, float* sq, float* acc, int64_t n, float lr, float rho, float eps,
                                 const float* norm3, int zero_grad, void* p_bf16, void* stream) {
    LAS_CHECK_ARG(p && g && sq && acc && norm3 && n > 0);
    LAS_CHECK_ARG(((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)sq) | ((uintptr_t)acc)) & 15) == 0);
    hipLaunchKernelGGL(adadelta_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, sq, acc, (long)n, lr, rho,
                       eps, norm3, zero_grad, (bf16_t*)p_bf16);
    LAS_LAUNCH_OK();
    return LAS_OK;
}
