// 128x128x32 MFMA tile machinery shared by the GEMM (gemm.hip) and the implicit-GEMM convolution (conv.hip):
// staging registers, global->register loaders, register->LDS stores in the [row][k] operand layout.
#pragma once
#include "las_common.h"

namespace las_tile {

constexpr int BM = 128, BN = 128, BK = 32, NT = 256;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int PREC> struct Elem;
template <> struct Elem<LAS_PREC_BF16> { typedef bf16_t T; static constexpr int PAD = 8; };   // 16 B
template <> struct Elem<LAS_PREC_F32>  { typedef float  T; static constexpr int PAD = 4; };   // 16 B

template <typename T> __device__ __forceinline__ T cvt(float f);
template <> __device__ __forceinline__ bf16_t cvt<bf16_t>(float f) { return f2bf(f); }
template <> __device__ __forceinline__ float cvt<float>(float f) { return f; }

// Stage-in registers for one operand tile (128 rows x 32 k): 4 float4 per thread.
// KCONT: element (r,k) at src[r*ld + k]; thread -> row = tid/8 + 32*p, k4 = (tid%8)*4.
// else : element (r,k) at src[k*ld + r]; thread -> k = 4*(tid/32) + p, r4 = (tid%32)*4: a 4x4 block that is
//        transposed in registers so the LDS writes are 4 consecutive k per row (8/16-byte stores).
// VEC (host-checked: 16-B aligned base, ld % 4 == 0, and K % 4 == 0 (KCONT) / rows % 4 == 0 (k-strided)): every
// load is ONE unconditional global_load_dwordx4 from a clamped, always-valid address; out-of-range vectors are
// zeroed on the data.  No branch, no per-load wait (cdna_hip_programming.md trap (c)).  Otherwise: guarded scalars.
// Staged tile fragment of one thread: 16 scalars (plain registers; a float4[4] that is read "transposed" gets
// parked in scratch by hipcc).  v[4*p + i] = i-th element of load p.
struct Frag16 { float v[16]; };

template <bool KCONT, bool VEC>
__device__ __forceinline__ void g_load(Frag16& reg, const float* __restrict__ src, long ld, int row0, int k0,
                                       int rows, int K) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int r, k;
        if (KCONT) { r = row0 + (tid >> 3) + 32 * p; k = k0 + (tid & 7) * 4; }
        else       { k = k0 + (tid >> 5) * 4 + p;    r = row0 + (tid & 31) * 4; }
        if (VEC) {
            const bool ok = r < rows && k < K;
            const int rc = KCONT ? min(r, rows - 1) : min(r, rows - 4);
            const int kc = KCONT ? min(k, K - 4) : min(k, K - 1);
            const float* q = KCONT ? src + (long)rc * ld + kc : src + (long)kc * ld + rc;
            const float4 t = ldg4(q);
            reg.v[4 * p + 0] = ok ? t.x : 0.f; reg.v[4 * p + 1] = ok ? t.y : 0.f;
            reg.v[4 * p + 2] = ok ? t.z : 0.f; reg.v[4 * p + 3] = ok ? t.w : 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rr = KCONT ? r : r + i, kk = KCONT ? k + i : k;
                const bool ok = rr < rows && kk < K;
                const float* q = KCONT ? src + (long)min(rr, rows - 1) * ld + min(kk, K - 1)
                                       : src + (long)min(kk, K - 1) * ld + min(rr, rows - 1);
                const float v = *q;
                reg.v[4 * p + i] = ok ? v : 0.f;
            }
        }
    }
}

// Interior tiles (all 128 rows and all 32 k inside the matrix, aligned): base pointer precomputed once per thread,
// four unconditional 16-byte loads, no clamping, no masking -- the generic loader above spends more VALU issue
// slots on address arithmetic and selects than the tile's 16 MFMAs take.
template <bool KCONT>
__device__ __forceinline__ void g_load_fast(Frag16& reg, const float* __restrict__ base, long ld, int k0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float4 t = KCONT ? ldg4(base + (long)(32 * p) * ld + k0) : ldg4(base + (long)(k0 + p) * ld);
        reg.v[4 * p + 0] = t.x; reg.v[4 * p + 1] = t.y; reg.v[4 * p + 2] = t.z; reg.v[4 * p + 3] = t.w;
    }
}

template <bool KCONT, typename T, int LDS_LD>
__device__ __forceinline__ void s_store(const Frag16& reg, T* __restrict__ tile) {
    const int tid = threadIdx.x;
    if (KCONT) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = (tid >> 3) + 32 * p, k = (tid & 7) * 4;
            store4_ct(tile + r * LDS_LD + k, reg.v[4 * p], reg.v[4 * p + 1], reg.v[4 * p + 2], reg.v[4 * p + 3]);
        }
    } else {
        const int k = (tid >> 5) * 4, r = (tid & 31) * 4;      // load p = k offset, element i = row offset
#pragma unroll
        for (int i = 0; i < 4; ++i)
            store4_ct(tile + (r + i) * LDS_LD + k, reg.v[i], reg.v[4 + i], reg.v[8 + i], reg.v[12 + i]);
    }
}


// ---- bf16 operand tiles.  A k-contiguous source keeps the [row][k] image (ds_read_b128 fragments).  A k-strided
// ("transposed") source used to be transposed in registers and scattered into that image: 8-byte stores whose
// 32 lanes sit 4 rows = 320 bytes apart, i.e. on 2 of the 32 write banks (16-way conflicts; the weight-gradient
// shapes, both operands k-strided, ran at 150-200 TFLOP/s).  It now stays k-major in LDS, exactly as it comes from
// memory ([k][row], 256 contiguous bytes per k row and wave: conflict-free stores), and the MFMA fragments are taken
// with gfx950's transposing read ds_read_b64_tr_b16 (guide T10): per 16-lane group a 4 (k) x 16 (rows) block, lane i
// receiving the 4 k values of row i.  The k slot of a lane (fr, fq) is {4 fq .. 4 fq + 3} u {16 + 4 fq .. 16 + 4 fq + 3}
// (the sum over k does not care, both operands use the same map): one transposing read then covers 8 consecutive k
// rows per 32-lane half, and with 288-byte rows (128 + 16 elements) those 8 x 32-byte pieces tile the 64 banks exactly.
// The [row][k] image stores its eight 4-element k chunks in that slot order so that its b128 read needs no change.
constexpr int LDK = BM + 16;                 // k-major image row stride (elements); 32 * LDK <= BM * (BK + 8)
typedef short v4s __attribute__((ext_vector_type(4)));

template <bool KCONT>
__device__ __forceinline__ void tile_store_bf16(const Frag16& reg, bf16_t* __restrict__ tile) {
    constexpr int LD = BK + Elem<LAS_PREC_BF16>::PAD;
    const int tid = threadIdx.x;
    if (KCONT) {
        const int c = tid & 7, pos = c < 4 ? 8 * c : 8 * (c - 4) + 4;     // logical chunk c -> slot position
#pragma unroll
        for (int p = 0; p < 4; ++p)
            store4_ct(tile + ((tid >> 3) + 32 * p) * LD + pos, reg.v[4 * p], reg.v[4 * p + 1], reg.v[4 * p + 2], reg.v[4 * p + 3]);
    } else {
        const int k = (tid >> 5) * 4, r = (tid & 31) * 4;                  // load p = k offset, 4 consecutive rows each
#pragma unroll
        for (int p = 0; p < 4; ++p)
            store4_ct(tile + (k + p) * LDK + r, reg.v[4 * p], reg.v[4 * p + 1], reg.v[4 * p + 2], reg.v[4 * p + 3]);
    }
}

// MFMA 16x16x32 fragment of rows rb .. rb + 15 (EXEC must be full for the transposing reads: call sites are uniform)
template <bool KCONT>
__device__ __forceinline__ bf16x8 tile_frag_bf16(const bf16_t* __restrict__ tile, int rb) {
    constexpr int LD = BK + Elem<LAS_PREC_BF16>::PAD;
    const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
    if (KCONT) return *(const bf16x8*)(tile + (rb + fr) * LD + fq * 8);
    const int q = fr >> 2, p = fr & 3;                                      // lane 4q+p of its group: block row q, columns 4p..
    typedef v4s __attribute__((address_space(3))) * lds_v4s;
    const bf16_t* a0 = tile + (4 * fq + q) * LDK + rb + 4 * p;
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)a0);
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(a0 + 16 * LDK));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}



// ---- bf16 SOURCE operands (activation twins written by their producers, the bf16 shadow of the weights): half the
// L2 -> LDS bytes of an fp32 source and no conversion while staging.  One thread stages two 16-byte vectors per tile.
// KCONT: vector = 8 consecutive k of one row (row = tid/4 + 64 p, k8 = 8 (tid % 4)); k-strided: 8 consecutive rows of
// one k (k = tid/16 + 16 p, r8 = 8 (tid % 16)).  Callers guarantee 16-byte aligned bases, ld % 8 == 0 and K % 8 == 0
// (KCONT) / rows % 8 == 0 (k-strided); vectors outside the matrix come from clamped addresses and are zeroed.
struct Frag8h { u32x4_t v[2]; };
template <bool KCONT>
__device__ __forceinline__ void g_load_bf16(Frag8h& reg, const bf16_t* __restrict__ src, long ld, int row0, int k0, int rows, int K) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        int r, k;
        if (KCONT) { r = row0 + (tid >> 2) + 64 * p; k = k0 + (tid & 3) * 8; }
        else       { k = k0 + (tid >> 4) + 16 * p;  r = row0 + (tid & 15) * 8; }
        const bool ok = r < rows && k < K;
        const int rc = KCONT ? min(r, rows - 1) : min(r, rows - 8);
        const int kc = KCONT ? min(k, K - 8) : min(k, K - 1);
        const bf16_t* q = KCONT ? src + (long)rc * ld + kc : src + (long)kc * ld + rc;
        const u32x4_t t = *reinterpret_cast<const u32x4_t*>(__builtin_assume_aligned(q, 16));
        reg.v[p] = ok ? t : (u32x4_t){0u, 0u, 0u, 0u};
    }
}
template <bool KCONT>
__device__ __forceinline__ void tile_store_bf16_src(const Frag8h& reg, bf16_t* __restrict__ tile) {
    constexpr int LD = BK + Elem<LAS_PREC_BF16>::PAD;
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (KCONT) {
            // the [row][k] image keeps its eight 4-element k chunks in the slot order of the transposing read (see above):
            // logical chunk c sits at position (c < 4 ? 8 c : 8 (c - 4) + 4); my vector is chunks 2 v and 2 v + 1
            const int r = (tid >> 2) + 64 * p, v = tid & 3, c0 = 2 * v, c1 = 2 * v + 1;
            const int p0 = c0 < 4 ? 8 * c0 : 8 * (c0 - 4) + 4, p1 = c1 < 4 ? 8 * c1 : 8 * (c1 - 4) + 4;
            *(uint2*)(tile + r * LD + p0) = make_uint2(reg.v[p][0], reg.v[p][1]);
            *(uint2*)(tile + r * LD + p1) = make_uint2(reg.v[p][2], reg.v[p][3]);
        } else {
            const int k = (tid >> 4) + 16 * p, r = (tid & 15) * 8;
            *(u32x4_t*)(tile + k * LDK + r) = reg.v[p];
        }
    }
}

}  // namespace las_tile
