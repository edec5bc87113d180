// Shared MFMA helpers for the skinny (M = batch <= 128) products of the recurrent kernels.
#pragma once
#include "las_common.h"

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// Operand ("compute") type per precision mode.
template <int PREC> struct CT;
template <> struct CT<LAS_PREC_BF16> { typedef bf16_t T; static constexpr int VEC = 8; static constexpr int KSTEP = 32; };
template <> struct CT<LAS_PREC_F32>  { typedef float  T; static constexpr int VEC = 4; static constexpr int KSTEP = 16; };

template <typename T> __device__ __forceinline__ T to_ct(float f);
template <> __device__ __forceinline__ bf16_t to_ct<bf16_t>(float f) { return f2bf(f); }
template <> __device__ __forceinline__ float to_ct<float>(float f) { return f; }

// acc[bt] += A(rows bt*16 + (lane&15), k) * B(row (lane&15), k) over k-steps [0, nks); both operands live in
// LDS as [row][k] (k contiguous).  16x16 MFMA tiles: A rows = batch, B rows = output columns.
// C/D layout of acc[bt][r]: column = lane&15, row = bt*16 + (lane>>4)*4 + r.
template <int PREC, int NB>
__device__ __forceinline__ void mma_rows(f32x4 (&acc)[NB], const typename CT<PREC>::T* __restrict__ Al, int lda,
                                         const typename CT<PREC>::T* __restrict__ Bl, int ldb, int nks) {
    const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
    if constexpr (PREC == LAS_PREC_BF16) {
        for (int ks = 0; ks < nks; ++ks) {
            const bf16x8 b = *(const bf16x8*)(Bl + fr * ldb + ks * 32 + fq * 8);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) {
                const bf16x8 a = *(const bf16x8*)(Al + (bt * 16 + fr) * lda + ks * 32 + fq * 8);
                acc[bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[bt], 0, 0, 0);
            }
        }
    } else {
        for (int ks = 0; ks < nks; ++ks) {        // lane holds k = 16*ks + 4*fq + {0..3}; MFMA j uses element j
            const float4 b = *(const float4*)(Bl + fr * ldb + ks * 16 + fq * 4);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) {
                const float4 a = *(const float4*)(Al + (bt * 16 + fr) * lda + ks * 16 + fq * 4);
                acc[bt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[bt], 0, 0, 0);
                acc[bt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[bt], 0, 0, 0);
                acc[bt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[bt], 0, 0, 0);
                acc[bt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[bt], 0, 0, 0);
            }
        }
    }
}

// tanh through one v_exp: |abs err| ~1e-7, used where the argument is already O(1) noise-limited.
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = __expf(2.f * x);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);      // v_rcp (1 ulp) + v_mul: a division, __fdividef included, is 7-12 instructions here
}

// Cell pointwise backward of the EARLIER decode step, fused into the skinny product that yields its recurrent dh
// (skinny.hip mode 3): pointers are that step's slabs.
struct las_skinny_pw {
    const float* dh_ext; long ld_ext;     // upstream gradient d loss / d h_top of that step, [B][ld_ext]
    float* dc_carry;                      // [B][C] running d c (updated in place)
    const float* gates; const float* c_t; const float* c_prev;   // saved [B][4C], [B][C], [B][C]
    float* dgates;                        // [B][4C] out
};

__device__ __forceinline__ float fast_sig(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

int las_skinny_launch_pk(int prec, const float* x0, long ldx0, const float* w0, long ldw0, int K0, const float* x1,
                         long ldx1, const float* w1, long ldw1, int K1, const float* x2, long ldx2, const float* w2,
                         long ldw2, int K2, int B, int N, const float* bias0, const float* bias1, int mode, float* out,
                         long ldo, int accumulate, int C, const float* c_prev, float* h_out, float* c_out,
                         float* gates_out, const las_skinny_pw* pw, const void* wpk, hipStream_t st);

inline int las_pick_nb(int B) { return B <= 16 ? 1 : B <= 32 ? 2 : B <= 64 ? 4 : B <= 128 ? 8 : 0; }

#define LAS_NB_SWITCH(NBV, CALL)                        \
    switch (NBV) {                                      \
        case 1: { constexpr int NB_ = 1; CALL; } break; \
        case 2: { constexpr int NB_ = 2; CALL; } break; \
        case 4: { constexpr int NB_ = 4; CALL; } break; \
        case 8: { constexpr int NB_ = 8; CALL; } break; \
        default: return LAS_E_UNSUPPORTED;              \
    }
