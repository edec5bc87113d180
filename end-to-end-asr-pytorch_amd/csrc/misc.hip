// Small data-movement / elementwise kernels around the hot ops (all HBM-bound, vectorised where aligned).
#include "las_common.h"

namespace {

// out[d1][d0][:] = in[d0][d1][:]   (batch-major <-> time-major activations)
__global__ __launch_bounds__(256) void transpose01_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          int D0, int D1, int F) {
    const long row = blockIdx.x;                 // over D0*D1 input rows
    const int d0 = row / D1, d1 = row % D1;
    const float* s = in + row * F;
    float* o = out + ((long)d1 * D0 + d0) * F;
    if ((F & 3) == 0) {
        for (int i = threadIdx.x; i < F / 4; i += 256) ((float4*)o)[i] = ((const float4*)s)[i];
    } else {
        for (int i = threadIdx.x; i < F; i += 256) o[i] = s[i];
    }
}

// out = dy * (1 - y^2)        (backward of y = tanh(.), reference asr.py:316); out16 (optional): its bf16 twin
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                       float* __restrict__ out, bf16_t* __restrict__ out16, long n) {
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * 256 * 4;
    for (; i + 3 < n; i += stride) {
        const float4 a = *(const float4*)(dy + i), b = *(const float4*)(y + i);
        const float4 o = make_float4(a.x * (1.f - b.x * b.x), a.y * (1.f - b.y * b.y), a.z * (1.f - b.z * b.z), a.w * (1.f - b.w * b.w));
        *(float4*)(out + i) = o;
        if (out16) store4_ct(out16 + i, o.x, o.y, o.z, o.w);
    }
    if (i < n && i + 3 >= n)
        for (long k = i; k < n; ++k) { const float o = dy[k] * (1.f - y[k] * y[k]); out[k] = o; if (out16) out16[k] = f2bf(o); }
}

// out16 = bf16(in): the bf16 twin of an activation whose producer does not write one itself (HBM-bound: 6 bytes per element)
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long n) {
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
    const long stride = (long)gridDim.x * 256 * 8;
    for (; i + 7 < n; i += stride) {
        const float4 a = *(const float4*)(in + i), b = *(const float4*)(in + i + 4);
        *(uint4*)(out + i) = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
    }
    if (i < n && i + 7 >= n)
        for (long k = i; k < n; ++k) out[k] = f2bf(in[k]);
}

// lens[b] = #frames whose feature sum != 0   (reference solver.py:134, done on the host there).
// grid (B, T-chunks of 64 frames): a wave sums 16 frames, integer atomics add the chunk counts (lens zeroed first).
constexpr int IL_FRAMES = 64;
__global__ __launch_bounds__(256) void infer_lengths_kernel(const float* __restrict__ x, int T, int D,
                                                            int32_t* __restrict__ lens) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t0 = blockIdx.y * IL_FRAMES, t1 = min(t0 + IL_FRAMES, T);
    int cnt = 0;
    for (int t = t0 + w; t < t1; t += 4) {
        const float* r = x + ((long)b * T + t) * D;
        float s = 0.f;
        for (int i = lane; i < D; i += 64) s += r[i];
        s = wave_sum(s);
        if (s != 0.f) ++cnt;
    }
    if (lane == 0 && cnt) atomicAdd(&lens[b], cnt);
}

// out[b] = #nonzero labels in y[b,:] (int64)   (reference solver.py:136,159)
__global__ __launch_bounds__(64) void count_nonzero_i64_kernel(const long long* __restrict__ y, int L,
                                                               int32_t* __restrict__ out) {
    const int b = blockIdx.x;
    float c = 0.f;
    for (int i = threadIdx.x; i < L; i += 64) c += (y[(long)b * L + i] != 0) ? 1.f : 0.f;
    c = wave_sum(c);
    if (threadIdx.x == 0) out[b] = (int)c;
}


// out[c][r] = in[r][c], 32x32 tiles through LDS
__global__ __launch_bounds__(256) void transpose2d_kernel(const float* __restrict__ in, float* __restrict__ out, int R,
                                                          int C) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < R && c0 + tx < C) tile[i][tx] = in[(long)(r0 + i) * C + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < C && r0 + tx < R) out[(long)(c0 + i) * R + r0 + tx] = tile[tx][i];
}

__device__ __forceinline__ unsigned drop_hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__global__ __launch_bounds__(256) void dropout_rows_kernel(const float* __restrict__ in, long ldi, float* __restrict__ out,
                                                           long ldo, int R, int N, float p, float scale, unsigned seed) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)R * N) return;
    const int r = (int)(idx / N), i = (int)(idx - (long)r * N);
    const unsigned h = drop_hash32(seed ^ drop_hash32((unsigned)idx * 0x9e3779b9u + 0x85ebca6bu));
    const float u = (float)(h >> 8) * (1.f / 16777216.f);
    out[(long)r * ldo + i] = u >= p ? in[(long)r * ldi + i] * scale : 0.f;
}

}  // namespace

extern "C" int las_transpose01(const float* in, float* out, int D0, int D1, int F, void* stream) {
    LAS_CHECK_ARG(in && out && D0 > 0 && D1 > 0 && F > 0);
    hipLaunchKernelGGL(transpose01_kernel, dim3((unsigned)((long)D0 * D1)), dim3(256), 0, (hipStream_t)stream, in, out, D0, D1, F);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

static int tanh_bwd_launch(const float* dy, const float* y, float* out, void* out16, int64_t n, void* stream) {
    LAS_CHECK_ARG(dy && y && out && n >= 0);
    LAS_CHECK_ARG(!out16 || (((uintptr_t)out16) & 7) == 0);
    if (n == 0) return LAS_OK;
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, y, out, (bf16_t*)out16, (long)n);
    LAS_LAUNCH_OK();
    return LAS_OK;
}
extern "C" int las_tanh_bwd(const float* dy, const float* y, float* out, int64_t n, void* stream) {
    return tanh_bwd_launch(dy, y, out, nullptr, n, stream);
}
extern "C" int las_tanh_bwd_twin(const float* dy, const float* y, float* out, void* out_bf16, int64_t n, void* stream) {
    return tanh_bwd_launch(dy, y, out, out_bf16, n, stream);
}
extern "C" int las_cast_bf16(const float* in, void* out_bf16, int64_t n, void* stream) {
    LAS_CHECK_ARG(in && out_bf16 && n >= 0 && (((uintptr_t)in) & 15) == 0 && (((uintptr_t)out_bf16) & 15) == 0);
    if (n == 0) return LAS_OK;
    long blocks = (n / 8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out_bf16, (long)n);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

extern "C" int las_infer_lengths(const float* x, int B, int T, int D, int32_t* lens, void* stream) {
    LAS_CHECK_ARG(x && lens && B > 0 && T > 0 && D > 0);
    LAS_HIP(hipMemsetAsync(lens, 0, sizeof(int32_t) * B, (hipStream_t)stream));
    hipLaunchKernelGGL(infer_lengths_kernel, dim3(B, (T + IL_FRAMES - 1) / IL_FRAMES), dim3(256), 0, (hipStream_t)stream, x, T, D, lens);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

extern "C" int las_count_nonzero_i64(const int64_t* y, int B, int L, int32_t* out, void* stream) {
    LAS_CHECK_ARG(y && out && B > 0 && L > 0);
    hipLaunchKernelGGL(count_nonzero_i64_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, (const long long*)y, L, out);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

extern "C" int las_transpose2d(const float* in, float* out, int R, int C, void* stream) {
    LAS_CHECK_ARG(in && out && R > 0 && C > 0);
    hipLaunchKernelGGL(transpose2d_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, (hipStream_t)stream, in, out, R, C);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

extern "C" int las_dropout_rows(const float* in, int64_t ld_in, float* out, int64_t ld_out, int R, int N, float p, unsigned seed,
                                void* stream) {
    LAS_CHECK_ARG(in && out && R >= 0 && N > 0 && ld_in >= N && ld_out >= N && p >= 0.f && p < 1.f);
    if (R == 0) return LAS_OK;
    hipLaunchKernelGGL(dropout_rows_kernel, dim3((unsigned)(((long)R * N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in,
                       (long)ld_in, out, (long)ld_out, R, N, p, 1.f / (1.f - p), seed);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

extern "C" unsigned las_decoder_drop_seed(unsigned drop_seed, int step, int layer) {
    return drop_seed * 0x9E3779B1u + (unsigned)step * 131u + (unsigned)layer * 7919u + 1u;
}
