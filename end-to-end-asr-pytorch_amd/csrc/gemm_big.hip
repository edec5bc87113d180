// Large-tile MFMA GEMM for the big bf16-source products of the step (gfx950): the x*W_ih^T projections of the BiLSTM layers
// (reference src/asr.py:473-481), the proj Linear of asr.py:307,316, and their backward products dX = dY*W and
// dW += dY^T*X.  Same contract as gemm.hip's kernel (C = alpha*opA(A)*opB(B) + beta*C + bias, optional tanh, optional bf16
// copy of the result; split-K slices added with float atomics), both operands bf16 twins (las_gemm_ex).
//
// Why a second kernel: the 128x128x32 tile of gemm.hip moves one L2 byte per 64 flop and stages through registers; at the
// C5 shapes (8192 gate columns, K = 2048 / 4096) it sits at ~14 % of the bf16 MFMA peak in the step.  Here:
//   * tile BM x BN x 64 with BM = 256, BN = 256 | 128, 512 threads = 8 waves (2 x 4 or 4 x 2), one workgroup per CU;
//   * operands go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`: no staging registers, no ds_write), 1 KiB per wave
//     instruction, a ring of NS stages; the loads of the stages ahead stay in flight ACROSS the per-stage barrier
//     (counted s_waitcnt vmcnt(N) + raw s_barrier: cdna_hip_programming.md "Pipelining across barriers");
//   * out-of-range rows / the K tail need no branch: the buffer resource's bounds check returns zeros for them;
//   * LDS images (the DMA writes lane-linearly, so the layout is chosen on the SOURCE address, guide rule 21):
//       k-contiguous operand  -> chunks of 8 rows x 128 B, the eight 16-byte pieces of a row XOR-permuted by (row >> 1) & 7:
//                                full 128-byte lines from memory, conflict-free ds_read_b128 fragments;
//       k-strided operand     -> blocks of [8 k][16 rows] (256 B), odd k-blocks with their k rows 0-3 <-> 4-7 swapped, read
//                                with ds_read_b64_tr_b16 (guide T10): the two 16-lane groups of a half hit disjoint banks;
//   * SWAP (plain stores): the MFMA takes B's fragment as its A operand, so a lane ends up with FOUR CONSECUTIVE n of one
//     row m: the epilogue is one 16-byte store per accumulator tile instead of four 4-byte ones (and a float4 C / bias
//     read); the split-K instances keep the plain roles (64-byte runs per atomic instruction);
//   * XCD-aware, grouped tile order: the workgroups that share an XCD (ids congruent mod 8) walk one contiguous run of
//     tiles in groups of GM m-tiles x all n-tiles, so the ~32 tiles an XCD works on at a time form a compact block whose A
//     and B panels are re-read from that XCD's L2.
#include "las_common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_vp;
typedef short v4s __attribute__((ext_vector_type(4)));
typedef v4s __attribute__((address_space(3))) * lds_v4s;

constexpr int BKS = 64;                 // k per stage
constexpr unsigned OOB = 0x80000000u;   // a byte offset no resource of < 2 GiB contains: the load returns zeros

struct BigArgs {
    int M, N, K;
    float alpha, beta;
    const bf16_t* A; long lda;
    const bf16_t* B; long ldb;
    float* C; long ldc;
    const float* bias;
    int act, ksplit;
    bf16_t* C16; long ldc16;
};

template <int BM, int BN, int WM> struct Geo {
    static constexpr int WN = 8 / WM, WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NJ = WTN / 16;
    static constexpr int CA = BM / 64, CB = BN / 64;              // 1-KiB chunks per wave, stage and operand
    static constexpr int STAGE = (BM + BN) * BKS * 2;             // bytes
};

// per-lane byte offset (from the workgroup's operand base) of the 16 bytes this lane moves for chunk `c` of a stage
//   k-contiguous: chunk = rows 8c .. 8c+7 x 64 k; lane l: row 8c + (l >> 3), stored slot l & 7 holds logical piece (l & 7) ^ f(row)
//   k-strided   : chunk = blocks (kb, 4 nb4 .. 4 nb4 + 3), c = kb * (R / 64) + nb4; lane l: block l >> 4, position (l & 15) >> 1
template <bool KC, int R>
__device__ __forceinline__ unsigned src_off(int c, int lane, long ld, int rows_left, int* k_local) {
    if (KC) {
        const int row = 8 * c + (lane >> 3);
        const int piece = (lane & 7) ^ ((((c & 1) << 3) + (lane >> 3)) >> 1);
        *k_local = 8 * piece;
        return (unsigned)(((long)row * ld + 8 * piece) * 2);          // rows beyond the matrix fall outside the resource
    } else {
        constexpr int NB4 = R / 64;
        const int kb = c / NB4, nb4 = c - kb * NB4;
        const int pos = (lane & 15) >> 1, kr = pos ^ ((kb & 1) << 2);
        const int col = 64 * nb4 + 16 * (lane >> 4) + 8 * (lane & 1);
        *k_local = 8 * kb + kr;
        return col < rows_left ? (unsigned)(((long)(8 * kb + kr) * ld + col) * 2) : OOB;   // k beyond K falls outside the resource
    }
}

// MFMA 16x16x32 fragment (rows rb*16 .. +15 of the operand tile, k-step ks of the stage)
template <bool KC, int R>
__device__ __forceinline__ bf16x8 frag(const unsigned char* __restrict__ tile, int rb, int ks, int fr, int fq) {
    if (KC) {
        const unsigned char* p = tile + (rb * 16 + fr) * 128 + ((((ks << 2) + fq) ^ (fr >> 1)) << 4);
        return *(const bf16x8*)p;
    } else {
        const int kb = 4 * ks + fq;
        const unsigned char* blk = tile + (kb * (R / 16) + rb) * 256;
        const int q = fr >> 2, p = fr & 3;
        const unsigned char* a0 = blk + ((q ^ ((fq & 1) << 2)) << 5) + 8 * p;
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)a0);
        const unsigned char* a1 = blk + (((q + 4) ^ ((fq & 1) << 2)) << 5) + 8 * p;
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)a1);
        return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Stage s of the K loop -> ring slot s % NS: CA + CB LDS-DMA instructions per wave.  (A plain function, not a lambda inside
// the kernel: hipcc's host pass silently drops the kernel's launch stub when a lambda that uses device builtins is CALLED.)
template <int BM, int BN, int CA, int CB, int NS>
__device__ __forceinline__ void issue_stage(int s, int nst, unsigned char* smem, int wave, const bf16_t* Ab, const bf16_t* Bb,
                                            long bytesA, long bytesB, long stepA, long stepB, const unsigned* voA,
                                            const unsigned* voB, const unsigned* vtA, const unsigned* vtB) {
    unsigned char* slot = smem + (s % NS) * ((BM + BN) * BKS * 2);
    const bool tail = s == nst - 1;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)Ab + s * stepA), 0, (int)(bytesA - s * stepA), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)Bb + s * stepB), 0, (int)(bytesB - s * stepB), 0x00020000);
#pragma unroll
    for (int i = 0; i < CA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_vp)(slot + (wave * CA + i) * 1024), 16, tail ? vtA[i] : voA[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < CB; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_vp)(slot + BM * 128 + (wave * CB + i) * 1024), 16, tail ? vtB[i] : voB[i], 0, 0, 0);
}

template <int BM, int BN, int WM, int NS, bool AKC, bool BKC, bool SWAP>
__global__ __launch_bounds__(512) void gemm_big_kernel(BigArgs a, int ntm, int ntn, int gm) {
    typedef Geo<BM, BN, WM> G;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;

    // ---- which tile (and K slice): XCD-contiguous runs, grouped GM x ntn inside a run
    const int ntiles = ntm * ntn, nwg = ntiles * a.ksplit;
    int t;
    {
        const int orig = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int zs = t / ntiles;
    t -= zs * ntiles;
    int tm, tn;
    {
        const int per = gm * ntn, g = t / per, w = t - g * per;
        const int gh = min(gm, ntm - g * gm);                       // the last group may be shorter
        tm = g * gm + w % gh; tn = w / gh;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int nst_all = (a.K + BKS - 1) / BKS, per_z = (nst_all + a.ksplit - 1) / a.ksplit;
    const int st0 = zs * per_z, st1 = min(nst_all, st0 + per_z);
    if (st0 >= st1) return;
    const int kbeg = st0 * BKS, Keff = min(a.K, st1 * BKS) - kbeg, nst = st1 - st0;

    // ---- buffer resources over what is left of each operand from this tile's first element
    const bf16_t* Ab = AKC ? a.A + (long)m0 * a.lda + kbeg : a.A + (long)kbeg * a.lda + m0;
    const bf16_t* Bb = BKC ? a.B + (long)n0 * a.ldb + kbeg : a.B + (long)kbeg * a.ldb + n0;
    const long bytesA = AKC ? ((long)(a.M - m0 - 1) * a.lda + Keff) * 2 : ((long)(Keff - 1) * a.lda + (a.M - m0)) * 2;
    const long bytesB = BKC ? ((long)(a.N - n0 - 1) * a.ldb + Keff) * 2 : ((long)(Keff - 1) * a.ldb + (a.N - n0)) * 2;
    // (the resource is re-based per stage with scalar arithmetic: base += stage bytes, num_records -= the same, so the
    // bounds check is exact for every stage and the per-lane offsets never change)

    unsigned voA[G::CA], voB[G::CB], vtA[G::CA], vtB[G::CB];
    const int ktail = (nst - 1) * BKS;                              // k_local + ktail >= Keff: beyond the K tail (last stage only)
#pragma unroll
    for (int i = 0; i < G::CA; ++i) {
        int kl;
        voA[i] = src_off<AKC, BM>(wave * G::CA + i, lane, a.lda, a.M - m0, &kl);
        vtA[i] = (AKC && kl + ktail >= Keff) ? OOB : voA[i];
    }
#pragma unroll
    for (int i = 0; i < G::CB; ++i) {
        int kl;
        voB[i] = src_off<BKC, BN>(wave * G::CB + i, lane, a.ldb, a.N - n0, &kl);
        vtB[i] = (BKC && kl + ktail >= Keff) ? OOB : voB[i];
    }
    const long stepA = AKC ? BKS * 2 : BKS * a.lda * 2, stepB = BKC ? BKS * 2 : BKS * a.ldb * 2;      // bytes per stage

#define LAS_ISSUE(s_) issue_stage<BM, BN, G::CA, G::CB, NS>(s_, nst, smem, wave, Ab, Bb, bytesA, bytesB, stepA, stepB, voA, voB, vtA, vtB)

    f32x4 acc[G::MI][G::NJ];
#pragma unroll
    for (int i = 0; i < G::MI; ++i)
#pragma unroll
        for (int j = 0; j < G::NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wm = (wave / G::WN) * G::WTM, wn = (wave % G::WN) * G::WTN;
    constexpr int LPS = G::CA + G::CB;                              // LDS-DMA instructions per wave and stage

    // ---- prologue: NS - 1 stages in flight
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nst) LAS_ISSUE(s);
    for (int s = 0; s < nst; ++s) {
        // stage s has landed (mine: counted wait; everyone's: the barrier), and nobody still reads stage s - 1
        const int ahead = min(NS - 2, nst - 1 - s);                 // younger stages that may stay in flight
        if (NS >= 4 && ahead >= 2) wait_vm<2 * LPS>();
        else if (NS >= 3 && ahead >= 1) wait_vm<LPS>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");                              // (no LDS read of this stage may move above the barrier)
        if (s + NS - 1 < nst) LAS_ISSUE(s + NS - 1);                    // into the slot stage s - 1 has just vacated
        const unsigned char* As = smem + (s % NS) * G::STAGE;
        const unsigned char* Bs = As + BM * 128;
#pragma unroll
        for (int ks = 0; ks < BKS / 32; ++ks) {
            bf16x8 af[G::MI], bfr[G::NJ];
#pragma unroll
            for (int j = 0; j < G::NJ; ++j) bfr[j] = frag<BKC, BN>(Bs, (wn >> 4) + j, ks, fr, fq);
#pragma unroll
            for (int i = 0; i < G::MI; ++i) af[i] = frag<AKC, BM>(As, (wm >> 4) + i, ks, fr, fq);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < G::MI; ++i)
#pragma unroll
                for (int j = 0; j < G::NJ; ++j)
                    acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    }

#undef LAS_ISSUE

    // ---- epilogue
    if (SWAP) {
        // D rows = n, columns = m: lane (fr, fq) holds C[m = .. + fr][n = .. + 4 fq .. + 3]
#pragma unroll
        for (int j = 0; j < G::NJ; ++j) {
            const int n = n0 + wn + j * 16 + fq * 4;
            if (n >= a.N) continue;                                 // (N % 4 == 0: a quad is inside or outside as a whole)
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias) bv = *(const float4*)(a.bias + n);
#pragma unroll
            for (int i = 0; i < G::MI; ++i) {
                const int m = m0 + wm + i * 16 + fr;
                if (m >= a.M) continue;
                float* c = a.C + (long)m * a.ldc + n;
                float4 v = make_float4(a.alpha * acc[i][j][0] + bv.x, a.alpha * acc[i][j][1] + bv.y,
                                       a.alpha * acc[i][j][2] + bv.z, a.alpha * acc[i][j][3] + bv.w);
                if (a.beta != 0.f) {
                    const float4 o = *(const float4*)c;
                    v.x += a.beta * o.x; v.y += a.beta * o.y; v.z += a.beta * o.z; v.w += a.beta * o.w;
                }
                if (a.act == LAS_ACT_TANH) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
                else if (a.act == LAS_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *(float4*)c = v;
                if (a.C16) *(uint2*)(a.C16 + (long)m * a.ldc16 + n) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
            }
        }
    } else {
        // D rows = m, columns = n (split-K): C was pre-scaled by beta; the slices are added with float atomics
#pragma unroll
        for (int j = 0; j < G::NJ; ++j) {
            const int n = n0 + wn + j * 16 + fr;
            if (n >= a.N) continue;
#pragma unroll
            for (int i = 0; i < G::MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm + i * 16 + fq * 4 + r;
                    if (m < a.M) atomicAdd(a.C + (long)m * a.ldc + n, a.alpha * acc[i][j][r]);
                }
        }
    }
}

template <int BM, int BN, int WM, int NS, bool SWAP>
int launch_big(int ta, int tb, const BigArgs& a, hipStream_t st) {
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
    const int gm = ntm < 4 ? ntm : 4;
    const size_t lds = (size_t)NS * Geo<BM, BN, WM>::STAGE;
    const dim3 grid(ntm * ntn * a.ksplit);
#define LAS_BIG_GO(AK, BK_)                                                                                        \
    {                                                                                                              \
        auto k = gemm_big_kernel<BM, BN, WM, NS, AK, BK_, SWAP>;                                                   \
        LAS_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));        \
        hipLaunchKernelGGL(k, grid, dim3(512), lds, st, a, ntm, ntn, gm);                                          \
    }
    if (!ta && tb) LAS_BIG_GO(true, true)
    else if (!ta && !tb) LAS_BIG_GO(true, false)
    else if (ta && !tb) LAS_BIG_GO(false, false)
    else LAS_BIG_GO(false, true)
#undef LAS_BIG_GO
    LAS_LAUNCH_OK();
    return LAS_OK;
}

__global__ __launch_bounds__(256) void big_scale2d_kernel(float beta, int N, float* __restrict__ C, long ldc) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < N) { float* p = C + (long)blockIdx.y * ldc + c; *p = beta != 0.f ? beta * (*p) : 0.f; }
}

}  // namespace

// Called by gemm.hip's dispatcher.  Returns LAS_E_UNSUPPORTED when the shape is not one this kernel is for (the caller
// then runs the 128^2 kernel); `cfg` != 0 forces a configuration (tools/bench_gemm.py: 1 = 256x256 two stages,
// 2 = 256x128 three stages).
int las_gemm_big(int transA, int transB, int M, int N, int K, float alpha, const void* A, int64_t lda, const void* B,
                 int64_t ldb, float beta, float* C, int64_t ldc, const float* bias, int act, void* C16, int64_t ldc16,
                 int cfg, hipStream_t st) {
    if (M < 256 || N < 128 || K < 256) return LAS_E_UNSUPPORTED;
    // 16-byte vectors of 8 bf16 along each operand's contiguous dimension; float4 rows of C / bias
    if ((lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || (K & 7) || (transA && (M & 7)) || (!transB && (N & 7)))
        return LAS_E_UNSUPPORTED;
    if ((N & 3) || (ldc & 3) || ((uintptr_t)C & 15) || (bias && ((uintptr_t)bias & 15)) || (C16 && ((ldc16 & 3) || ((uintptr_t)C16 & 7))))
        return LAS_E_UNSUPPORTED;
    const long spanA = (transA ? (long)K * lda : (long)M * lda) * 2, spanB = (transB ? (long)N * ldb : (long)K * ldb) * 2;
    if (spanA >= (1l << 31) || spanB >= (1l << 31)) return LAS_E_UNSUPPORTED;          // 32-bit buffer offsets
    const int cus = las_cu_count();
    const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256), t128 = (long)((M + 255) / 256) * ((N + 127) / 128);
    BigArgs a{M, N, K, alpha, beta, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, bias, act, 1, (bf16_t*)C16, ldc16};
    // split-K for the weight-gradient shapes (few output tiles, K = T*B in the tens of thousands)
    const bool can_split = !bias && act == LAS_ACT_NONE && !C16;
    // Which tile (tools/bench_gemm.py on the C2 / C5 shapes, profiles/r03_gemm_shapes.txt): 256x256 once it fills every CU
    // (C5: 1.0-1.27 PFLOP/s against 0.6-0.78 for the 128^2 kernel; C2 layer 1: 729 / 645 against 526 / 557); 256x128 (three
    // stages in flight) for grids in between; split-K (256x128, plain MFMA roles, float atomics) for the weight-gradient
    // shapes when the slices can be cut so that the grid fills whole rounds of CUs.
    int which = cfg;
    if (which == 0) {
        if (t256 >= cus) which = 1;
        else if (t128 >= (long)cus * 3 / 4) which = 2;
        else if (can_split && K >= 4096) which = 2;
        else return LAS_E_UNSUPPORTED;
    }
    long tiles = which == 1 ? t256 : t128;
    if (can_split && tiles < cus * 3 / 4) {
        which = 2;
        tiles = t128;
        // slices: the count (at least 8 stages each) that wastes the least of the last round of workgroups
        const int nst = (K + BKS - 1) / BKS;
        int best = 1;
        double best_u = 0.0;
        for (int ks = 1; ks <= 16 && ks * 8 <= nst; ++ks) {
            const long wgs = tiles * ks;
            const double u = (double)wgs / (double)(((wgs + cus - 1) / cus) * cus);
            if (u > best_u + 0.02) { best_u = u; best = ks; }
        }
        if (cfg == 0 && best_u < 0.7) return LAS_E_UNSUPPORTED;
        a.ksplit = best;
    }
    if (a.ksplit > 1) {
        if (beta != 1.f) {
            hipLaunchKernelGGL(big_scale2d_kernel, dim3((N + 255) / 256, M), dim3(256), 0, st, beta, N, C, (long)ldc);
            LAS_LAUNCH_OK();
        }
        return launch_big<256, 128, 4, 3, false>(transA, transB, a, st);
    }
    return which == 1 ? launch_big<256, 256, 2, 2, true>(transA, transB, a, st)
                      : launch_big<256, 128, 4, 3, true>(transA, transB, a, st);
}
