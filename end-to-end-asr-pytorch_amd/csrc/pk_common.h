// Shared device primitives of the persistent decoder loops (decoder_pk.hip forward, decoder_pk_bwd.hip backward): the
// cross-CU hand-off (MI355X_MICROARCH.md "Valid forms", row 1: sc1 stores, every storing wave drains vmcnt(0), workgroup
// barrier, ONE lane adds to an agent-scope counter; consumers poll with sc1 loads from one wave, barrier, then read the
// bytes with sc1 loads only), tile pulls, and the diagnostic cycle stamps.
#pragma once
#include <type_traits>
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
// 1: every part of the utterance reported the same hardware XCC id; 0: not (or the id mapping is not the XCD-grouped one);
// -1: timeout.  One rendezvous before the first step through agent-scope atomics on words 32..34 of a zeroed 256-byte line of the utterance.
__device__ __forceinline__ int pk_utt_local(unsigned* line, int nparts, bool try_local, unsigned* abort_word, int* flag) {
    if (!try_local) return 0;
    unsigned* w = line + 32;
    if (threadIdx.x == PNT - 64) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        x &= 15u;
        const unsigned o1 = __hip_atomic_fetch_max(w + 1, x + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned o2 = __hip_atomic_fetch_max(w + 2, 16u - x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::"v"(o1), "v"(o2) : "memory");          // both maxima performed before the arrival below
        __hip_atomic_fetch_add(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!pk_block_wait(w, 0, 1, (unsigned)nparts, abort_word, flag)) return -1;
    const unsigned mx = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                   mn = __hip_atomic_load(w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return mx + mn == 17u ? 1 : 0;
}
__device__ __forceinline__ float ld_sc1(const float* p) {
    return __uint_as_float(__hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store((unsigned*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ---- tagged granules (MI355X_MICROARCH.md "R2": a naturally aligned 16-byte store is observed untorn by 16-byte sc1 loads
// on gfx950; not an architectural guarantee) ------------------------------------------------------------------------
// A hand-off without flag, drain or atomic: the payload travels as 16-byte granules {<= 12 bytes of data, tag = step + 1}
// written by ONE lane with ONE sc1 store; the rings are zeroed before the launch, so a tag can only be matched by this
// launch's store of that step.  The consumer sweeps its granules (one 16-byte sc1 load per granule and pass, consecutive
// lanes <-> consecutive granules) until every tag matches: in steady state the first pass does, and a hop costs one L2
// round trip instead of drain + barrier + atomic + poll + barrier + pull (the flag form was ~2 us per hop here).
constexpr int GR_OOB = 0x7ffffff0;                                // an offset beyond every ring: returns zeros, no memory traffic
__device__ __forceinline__ u32x4 pk_gr_poll(__amdgpu_buffer_rsrc_t rs, int off) {
    asm volatile("" : "+v"(off));                                 // identical polls must neither be merged nor hoisted
    return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
}
// Sweep SW granules per lane (byte offsets goff, GR_OOB = none) until each carries `tag`; uniform over the wave, branch-free.
// Sweep SW granules per lane (byte offsets goff, GR_OOB = none) until each carries `tag`; uniform over the wave.  A matched
// granule is handed to on_hit(u, granule) at once (an exec-masked LDS write at the call sites): keeping the matched
// payloads in registers until the end of the sweep cost 4 SW registers at the point of the roles' highest pressure.
template <int SW, typename F>
__device__ __forceinline__ bool pk_gr_sweep(__amdgpu_buffer_rsrc_t rs, const int (&goff)[SW], unsigned tag, unsigned* abort_word, F&& on_hit) {
    int off[SW];
    u32x4 v[SW];
#pragma unroll
    for (int u = 0; u < SW; ++u) {
        off[u] = goff[u];
        v[u] = pk_gr_poll(rs, off[u]);
    }
    unsigned spins = 0;
    while (true) {
        bool need = false;
#pragma unroll
        for (int u = 0; u < SW; ++u) {
            const bool hit = off[u] != GR_OOB && v[u][3] == tag;
            if (hit) on_hit(u, v[u]);
            off[u] = hit ? GR_OOB : off[u];
            need = need || off[u] != GR_OOB;
        }
        if (__builtin_amdgcn_ballot_w64(need) == 0ull) return true;
#pragma unroll
        for (int u = 0; u < SW; ++u) v[u] = pk_gr_poll(rs, off[u]);
        if ((++spins & 255u) == 0) {
            if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || spins > PK_SPIN) {
                __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
}
__device__ __forceinline__ void pk_gr_store(__amdgpu_buffer_rsrc_t rs, int off, unsigned w0, unsigned w1, unsigned w2, unsigned tag) {
    const u32x4 g = {w0, w1, w2, tag};
    __builtin_amdgcn_raw_buffer_store_b128(g, rs, off, 0, 16);     // sc1: the consumers sit on other XCDs
}
// ... or a plain store when producer and consumers share an XCD's L2 (checked at run time, pk_utt_local)
__device__ __forceinline__ void pk_gr_store_x(__amdgpu_buffer_rsrc_t rs, int off, unsigned w0, unsigned w1, unsigned w2, unsigned tag, bool local) {
    const u32x4 g = {w0, w1, w2, tag};
    if (local) __builtin_amdgcn_raw_buffer_store_b128(g, rs, off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(g, rs, off, 0, 16);
}
template <int CTRL>
__device__ __forceinline__ unsigned pk_dpp_u(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false); }
// Payload words of the granule that STARTS at this lane, from one value per lane in consecutive lanes of a 16-lane row
// (row_shl: lanes beyond the row read 0).  12-byte payload: 3 f32 / 6 bf16 per granule; 8-byte payload: 2 f32 / 4 bf16.
template <typename T> struct GrT;
template <> struct GrT<float>  { static constexpr int V12 = 3, V8 = 2; };
template <> struct GrT<bf16_t> { static constexpr int V12 = 6, V8 = 4; };
template <typename T>
__device__ __forceinline__ void pk_gr_gather12(float v, unsigned (&w)[3]) {
    if constexpr (sizeof(T) == 4) {
        w[0] = __float_as_uint(v); w[1] = pk_dpp_u<0x101>(w[0]); w[2] = pk_dpp_u<0x102>(w[0]);
    } else {
        w[0] = pack_bf16x2(v, __uint_as_float(pk_dpp_u<0x101>(__float_as_uint(v)))); w[1] = pk_dpp_u<0x102>(w[0]); w[2] = pk_dpp_u<0x104>(w[0]);
    }
}
template <typename T>
__device__ __forceinline__ void pk_gr_gather8(float v, unsigned (&w)[3]) {
    if constexpr (sizeof(T) == 4) {
        w[0] = __float_as_uint(v); w[1] = pk_dpp_u<0x101>(w[0]); w[2] = 0u;
    } else {
        w[0] = pack_bf16x2(v, __uint_as_float(pk_dpp_u<0x101>(__float_as_uint(v)))); w[1] = pk_dpp_u<0x102>(w[0]); w[2] = 0u;
    }
}
// The first NW payload words of a matched granule into an LDS row of T (4-byte aligned destination).
template <typename T, int NW>
__device__ __forceinline__ void pk_gr_scatter(T* dst, const u32x4& g) {
#pragma unroll
    for (int e = 0; e < NW; ++e) ((unsigned*)dst)[e] = g[e];
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
