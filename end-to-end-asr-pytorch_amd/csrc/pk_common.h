// Shared device primitives of the persistent decoder loops (decoder_pk.hip forward, decoder_pk_bwd.hip backward): the
// cross-CU hand-off (MI355X_MICROARCH.md "Valid forms", row 1: sc1 stores, every storing wave drains vmcnt(0), workgroup
// barrier, ONE lane adds to an agent-scope counter; consumers poll with sc1 loads from one wave, barrier, then read the
// bytes with sc1 loads only), tile pulls, and the diagnostic cycle stamps.
#pragma once
#include "las_mma.h"

namespace {

constexpr int PNT = 512, PNW = PNT / 64;
constexpr unsigned PK_SPIN = 1u << 22;
constexpr int CLW = 64;                                           // words per counter line: every counter on its own 256 bytes
constexpr size_t PK_MIN_LDS = 84 * 1024;                          // > 80 KiB: one workgroup per CU
constexpr size_t PK_LDS_CAP = 160 * 1024;

// ---- in-kernel cycle stamps (diagnostic build only: make stamps -> liblas_hip_stamps.so; the product build has none) ----
#ifdef LAS_PK_STAMPS
#define PK_STAMP_DECL unsigned st_acc[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = (unsigned)__builtin_amdgcn_s_memtime()
#define PK_STAMP(i)                                                          \
    do {                                                                     \
        const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime();        \
        st_acc[i] += now_ - st_last;                                         \
        st_last = now_;                                                      \
    } while (0)
#define PK_STAMP_FLUSH(dbg)                                                                          \
    do {                                                                                             \
        if (threadIdx.x == 0 && (dbg))                                                               \
            for (int i_ = 0; i_ < 20; ++i_) (dbg)[(long)blockIdx.x * 20 + i_] = st_acc[i_];          \
    } while (0)
#else
#define PK_STAMP_DECL
#define PK_STAMP(i)
#define PK_STAMP_FLUSH(dbg)
#endif

// ---- hand-off primitives ------------------------------------------------------------------------------------------
// The LAST wave polls: lane l < n watches counter cnt0 + l*stride (one 4-byte sc1 load per lane and poll).  Result through
// the LDS word `flag` (callers alternate between two words so that a fast wave cannot overwrite one still being read).
__device__ __forceinline__ bool pk_block_wait(unsigned* cnt0, int stride, int n, unsigned target, unsigned* abort_word, int* flag) {
    if (threadIdx.x >= PNT - 64) {
        const int lane = threadIdx.x & 63;
        unsigned* p = cnt0 + (long)min(lane, n - 1) * stride;
        unsigned spins = 0;
        bool ok = true;
        while (__builtin_amdgcn_ballot_w64(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0) {
                if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = false; break; }
                if (spins > PK_SPIN) { __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
            }
        }
        if (lane == 0) *flag = ok ? 1 : 0;
    }
    __syncthreads();
    return *flag != 0;
}
// Publish: every storing wave drains its stores, workgroup barrier, one lane adds to the counter.
__device__ __forceinline__ void pk_signal(unsigned* cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == PNT - 64) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
    return __uint_as_float(__hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store((unsigned*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// exchange store of the pair (a, b) at consecutive columns, sc1 (write-through): 4 bytes (bf16) / 8 bytes (f32)
__device__ __forceinline__ void st_pair_sc1(bf16_t* p, float a, float b) {
    __hip_atomic_store((unsigned*)p, pack_bf16x2(a, b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_pair_sc1(float* p, float a, float b) {
    const unsigned long long v = (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
    __hip_atomic_store((unsigned long long*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Pull rows x cols (16-byte vectors, sc1 = L1 bypass) of a published tile into LDS; every load of a thread is issued
// before its first LDS write; out-of-range lanes read through the buffer descriptor's bounds check (0, no branch).
template <typename T, int VEC, int UNR>
__device__ __forceinline__ void pk_pull(const T* __restrict__ src, int rows, int cols, int src_ld, T* __restrict__ lds, int ld) {
    const int vpr = cols / VEC, total = rows * vpr;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, rows * src_ld * (int)sizeof(T), 0x00020000);
    for (int i0 = threadIdx.x; i0 < total; i0 += PNT * UNR) {
        u32x4 v[UNR];
        int dst[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int i = i0 + u * PNT;
            const int r = i / vpr, c = (i - r * vpr) * VEC;
            dst[u] = i < total ? r * ld + c : -1;
            const int off = i < total ? (r * src_ld + c) * (int)sizeof(T) : 0x7ffffff0;
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (dst[u] >= 0) *(u32x4*)(lds + dst[u]) = v[u];
    }
}

// four consecutive LDS values as floats: one 8-byte (bf16) / 16-byte (f32) read
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
    const uint2 v = *(const uint2*)p;
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xffff0000u));
}
__device__ __forceinline__ float4 ld4(const float* p) { return *(const float4*)p; }


}  // namespace
