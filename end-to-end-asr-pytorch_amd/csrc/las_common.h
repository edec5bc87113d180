// Shared device/host helpers for liblas_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <math.h>
#include "../../include/las_hip.h"

#define LAS_WAVE 64

#define LAS_CHECK_ARG(cond)            \
    do {                               \
        if (!(cond)) return LAS_E_BADARG; \
    } while (0)

// Launch-error check: never synchronises; returns the positive hipError_t.
#define LAS_LAUNCH_OK()                              \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

#define LAS_HIP(call)                                \
    do {                                             \
        hipError_t e__ = (call);                     \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

static inline size_t las_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------- wave / block reductions
// Wave-wide reductions on DPP (plain VALU ops).  __shfl_xor compiles to ds_bpermute_b32: an LDS-crossbar round trip per
// step, six dependent ones per reduction -- measured as the dominant cost of kernels doing ten reductions per row.
// row_shr:1,2,4,8 leave each 16-lane row's total in its lane 15; row_bcast:15 / :31 fold the rows; lane 63 has the
// wave total, which is broadcast with v_readlane.  Lanes without a DPP source receive `keep`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float las_dpp(float keep, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += las_dpp<0x111, 0xf>(0.f, v);
    v += las_dpp<0x112, 0xf>(0.f, v);
    v += las_dpp<0x114, 0xf>(0.f, v);
    v += las_dpp<0x118, 0xf>(0.f, v);
    v += las_dpp<0x142, 0xa>(0.f, v);
    v += las_dpp<0x143, 0xc>(0.f, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, las_dpp<0x111, 0xf>(v, v));
    v = fmaxf(v, las_dpp<0x112, 0xf>(v, v));
    v = fmaxf(v, las_dpp<0x114, 0xf>(v, v));
    v = fmaxf(v, las_dpp<0x118, 0xf>(v, v));
    v = fmaxf(v, las_dpp<0x142, 0xa>(v, v));
    v = fmaxf(v, las_dpp<0x143, 0xc>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// Block-wide reductions; `red` is >= 32 floats of LDS; every thread gets the result.
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = -INFINITY;
    for (int i = 0; i < nw; ++i) r = fmaxf(r, red[i]);
    return r;
}

// 16-byte load the compiler keeps as ONE global_load_dwordx4: the pointer is asserted aligned and the load is
// unconditional.  Mask by selecting on the DATA afterwards (v = ok ? v : 0): a `ok ? *p : 0` select makes hipcc
// branch around every load, split it into dwords and wait per element (cdna_hip_programming.md, trap (c)).
__device__ __forceinline__ float4 ldg4(const float* p) {
    return *reinterpret_cast<const float4*>(__builtin_assume_aligned(p, 16));
}
__device__ __forceinline__ float4 sel4(bool ok, const float4& v) {
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// put(i, src[i]) for i in [0,n): U independent loads per thread are issued before the first put, so a fill costs
// ceil(n / (threads*U)) memory round trips instead of one per element-stride.  Loads are unconditional from clamped
// indices (see ldg4 above for why).
template <int U, typename F>
__device__ __forceinline__ void fill_batched(const float* __restrict__ src, int n, F put) {
    const int nt = blockDim.x;
    for (int i0 = threadIdx.x; i0 < n; i0 += nt * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[min(i0 + u * nt, n - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i0 + u * nt < n) put(i0 + u * nt, v[u]);
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ---------------------------------------------------------------- bf16 helpers
typedef unsigned short bf16_t;
// fp32 -> bf16, round-to-nearest-even, NaN stays NaN: a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950
// (MI355X_MICROARCH.md "Correctness boundaries").  Use the packed forms below where two or more values convert.
__device__ __forceinline__ bf16_t f2bf(float f) {
    const __bf16 h = (__bf16)f;
    return __builtin_bit_cast(unsigned short, h);
}
// ---- the saved s = tanh(psi + q + u) of the location-aware energies (asr.py:453), bf16 mode: 16 bits per element instead of
// fp32 (s is the largest thing the decoder saves: L*B*T'*A elements, written once, read by the BPTT loop and by the sums
// behind it).  The code is x = copysign(1 - |s|, s) as bf16 (RNE), not bf16(s): the backward pass needs
//   1 - s^2 = t (2 - t), t = |x|   relative error 2^-8 -- also where tanh saturates, which bf16(s) would round to 0 or 2^-8
//   s = sign(x) (1 - t)            absolute error <= 2^-9 (1 - |s|)      (d w_e += d e * s)
// fp32 mode keeps plain fp32 s.  las_decoder_s_elem_bytes(prec) is the element size callers allocate with.
__device__ __forceinline__ bf16_t las_s16_enc(float s) { return f2bf(copysignf(1.f - fabsf(s), s)); }
__device__ __forceinline__ float las_s16_t(unsigned x) { return __uint_as_float((x & 0x7fffu) << 16); }
__device__ __forceinline__ float las_s16_ds(unsigned x) { const float t = las_s16_t(x); return t * (2.f - t); }
__device__ __forceinline__ float las_s16_s(unsigned x) {
    return __uint_as_float(__float_as_uint(1.f - las_s16_t(x)) | ((x & 0x8000u) << 16));
}
// element i of a saved-s array in either format -> s and 1 - s^2
__device__ __forceinline__ void las_s_load(const void* base, long i, int s16, float& s, float& ds) {
    if (s16) { const unsigned x = ((const bf16_t*)base)[i]; s = las_s16_s(x); ds = las_s16_ds(x); }
    else { s = ((const float*)base)[i]; ds = 1.f - s * s; }
}
__device__ __forceinline__ void las_s_store(void* base, long i, int s16, float s) {
    if (s16) ((bf16_t*)base)[i] = las_s16_enc(s);
    else ((float*)base)[i] = s;
}
typedef __attribute__((ext_vector_type(2))) float las_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 las_bf16x2;
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {       // one v_cvt_pk_bf16_f32
    const las_f32x2 v = {a, b};
    const las_bf16x2 h = __builtin_convertvector(v, las_bf16x2);
    return __builtin_bit_cast(unsigned, h);
}
// 4 converted values stored as one 8-byte (bf16) / 16-byte (f32) access
__device__ __forceinline__ void store4_ct(unsigned short* dst, float a, float b, float c, float d) {
    *(uint2*)dst = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
}
__device__ __forceinline__ void store4_ct(float* dst, float a, float b, float c, float d) {
    *(float4*)dst = make_float4(a, b, c, d);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA bf16 A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;    // 16x16 accumulator fragment

// ---- run-time switches.  The product library reads only the documented FALLBACKS (README: LAS_LSTM_NO_XL, LAS_LSTM_NO_GR,
// LAS_DEC_NO_PK, LAS_DEC_NO_XL, LAS_LOC_POST_VALU): each selects an older, independently tested kernel form, which the parity
// tests also run.  A/B measurement knobs exist in the diagnostic build only (`make stamps`, -DLAS_DIAG): in liblas_hip.so
// LAS_AB_KNOB(...) is a compile-time nullptr and the environment is not consulted.
#include <cstdlib>
static inline const char* las_fallback(const char* name) { return getenv(name); }
// CUs of the current device (cached).  The persistent kernels need every workgroup co-resident, one per CU: their geometry
// is sized from this, never from a literal 256 -- on a partitioned or smaller device a grid that does not fit would spin
// until its timeout instead of taking the per-step path.  (The XCD-grouping arithmetic still assumes 8 XCDs of CUs/8; the
// kernels verify placement at run time and fall back to the placement-independent protocol.)
static inline int las_cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else return 256;                 // (no device: size queries made without a GPU answer for an MI355X)
    }
    return n;
}
#ifdef LAS_DIAG
#define LAS_AB_KNOB(name) getenv(name)
#else
#define LAS_AB_KNOB(name) ((const char*)nullptr)
#endif
