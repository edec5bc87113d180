// Persistent attend-and-spell decoder loop, backward through time: ONE launch for all L steps (location-aware attention,
// one Speller layer, no dropout) instead of three launches per step.  Hand-written counterpart of what autograd derives
// for reference src/asr.py:84-107; fills the same las_dec_bwd_state buffers as decoder_bwd.hip's per-step kernels
// (dgates, dxin's context half, dq_pre, de, df), so the post-loop contractions (att_loc_post, att_conv_wgrad, the weight
// gradient GEMMs) run unchanged.
//
// Per step t (newest first), with the hand-off protocol of pk_common.h between two roles:
//   CELL workgroup (U hidden units, batch slice bs): cell pointwise backward -> its [Bs x 4U] slice of dgates_t; K-SPLIT
//     product of that slice with its 4U rows of [W_ih(ctx part) | W_hh] -> a partial [Bs x (E + C)] of d ctx_t and of the
//     recurrent d h_{t-1}, published as "pieces" (a reduce-scatter, as in the persistent LSTM's BPTT: nothing has to be
//     gathered before the MFMAs).  d h_{t-1} = sum of the pieces addressed to it + d q_pre_t W_phi (its own 16 columns).
//   ATTENTION workgroup (utterance b, part c): enc[b][all T'][E-slice] in LDS as in the forward loop.
//     d ctx[E-slice] = sum of the cell pieces;  d a_part[t'] = enc[t'][slice] . d ctx[slice]  (+ for its own T'-chunk the
//     location-conv path: the transposed conv of the NEXT step's d f) -> all-gather over the utterance's parts ->
//     d a, the softmax dot, d e for its T'-chunk -> d z = d e w_e (1 - s^2), d q partial, d u = d z (1 - u^2),
//     d f = d u W_lp (MFMA) -> published for the previous step's conv path; d q partials all-gathered, each part
//     reduces an a-slice, applies (1 - q^2) and publishes d q_pre to the cell role.
// Hand-offs on the chain per step: pieces (cell -> attention), d a parts, d q partials, d q_pre (attention -> cell).
#include "pk_common.h"
#include "decoder_pk.h"
#include <stdlib.h>
#include <stdio.h>

extern "C" int las_gemm(int prec, int transA, int transB, int M, int N, int K, float alpha, const float* A, int64_t lda,
                        int64_t strideA, const float* B, int64_t ldb, int64_t strideB, float beta, float* C, int64_t ldc,
                        int64_t strideC, const float* bias, int act, int batch, void* stream);

namespace {

constexpr int LOC_C = 10, LOC_K = 100, LOC_W = 2 * LOC_K + 1;
constexpr int LWP = 208, NSEG = 4, SEGW = LWP / NSEG;
constexpr float ATT_SCALE = 2.0f;
constexpr int MAXB = 32, MAXNS = 4;

struct PbSync {                         // zeroed before every launch
    unsigned abort_[CLW];
    unsigned cnt_p[MAXNS][CLW];         // per batch slice: piece sets published (NCT per step)
    unsigned cnt_da[MAXB][CLW];         // per utterance: d a parts published (NCH per step)
    unsigned cnt_dqp[MAXB][CLW];        // per utterance: d q partials published
    unsigned cnt_df[MAXB][CLW];         // per utterance: d f chunks published
    unsigned cnt_dq[MAXB][CLW];         // per utterance: d q_pre slices published
};

struct PbGeom {
    int U, NCT, NS, Bs, NB, NCELL;
    int NCH, TC, ES, MT, NTW;
    int xl;                             // attention parts of an utterance on block ids that are congruent mod 8 (one XCD, observed)
    int NX;                             // piece row: [0,E) d ctx, [E,E+C) d h; padded to 16
    int Ap;                             // A padded to the MFMA k-step
    int AS;                             // d q_pre columns reduced per attention part
    size_t lds;
};

struct PbArgs {
    int B, Tp, E, A, C, L;
    int loc;                            // 1: location-aware attention; 0: dot attention (no conv / u / d f phases; d e -> d q through psi)
    PbGeom g;
    const float* enc; const float* psi; const int32_t* lens;
    const float* w_ih; const float* w_hh; const float* w_phi; const float* conv_w; const float* w_lp; const float* w_e;
    // saved by the forward loop
    const float* att; const float* q; const float* gates; const float* cs; const float* f; const void* s;      // (s: fp32, or the 16-bit code of las_common.h in bf16 mode)
    const float* g_htop;
    // outputs (las_dec_bwd_state)
    float* dgates; float* dxin; float* dq_pre; float* de; float* df;
    // exchange rings (2 slots each)
    void* px;                           // [2][NCT][B][NX / V8] 16-byte granules {V8 values of a K-split piece (4 bf16 | 2 f32), 0, tag = step count}
    float* dax;                         // [2][B][NCH][Tp]
    float* dqx;                         // [2][B][NCH][Ap]
    float* dfx;                         // [2][B][10][Tp]
    void* dqq;                          // [2][B][Ap / V8] 16-byte granules {V8 values of d q_pre (4 bf16 | 2 f32), 0, tag = step count}
    PbSync* sync; int* status;
    unsigned long long* dbg;
};

// four consecutive values of the compute type as one store
__device__ __forceinline__ void st4_sc1(bf16_t* p, float a, float b, float c, float d) {
    const unsigned long long v = (unsigned long long)pack_bf16x2(a, b) | ((unsigned long long)pack_bf16x2(c, d) << 32);
    __hip_atomic_store((unsigned long long*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st4_sc1(float* p, float a, float b, float c, float d) {
    st_pair_sc1(p, a, b);
    st_pair_sc1(p + 2, c, d);
}
__device__ __forceinline__ float ldb_sc1(__amdgpu_buffer_rsrc_t rs, int byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, 16));
}
// ---- exchanges among the parts of ONE utterance (d a, d q partials, d f): when all its parts sit on one XCD (checked at
// run time, pb_utt_local) they use the LSTM's L2-local form (lstm.hip): the payload as PLAIN stores (write-through L1, the
// line stays in the XCD's L2, where the readers' L1-bypassing loads find it), the signal as a plain store of the step
// count to the part's own progress word (no atomic: an agent-scope atomic and its poll go to the memory side), the wait a
// poll of the NCH words.  Otherwise: sc1 stores and one atomic counter, as between the roles.
__device__ __forceinline__ void st_x(float* p, float v, bool local) {
    if (local) *p = v;
    else st_sc1(p, v);
}
__device__ __forceinline__ void pk_signal_x(unsigned* line, int part, unsigned steps, bool local) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == PNT - 64) {
        if (local) line[1 + part] = steps;
        else __hip_atomic_fetch_add(line, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ bool pk_block_wait_x(unsigned* line, int nparts, unsigned steps, bool local, unsigned* abort_word, int* flag) {
    return local ? pk_block_wait(line + 1, 1, nparts, steps, abort_word, flag)
                 : pk_block_wait(line, 0, 1, (unsigned)nparts * steps, abort_word, flag);
}
__device__ __forceinline__ float ct2f(float v) { return v; }
__device__ __forceinline__ float ct2f(bf16_t v) { return bf2f(v); }

// ---- cell role ---------------------------------------------------------------------------------------------------
template <int PREC, int NB>
__device__ __forceinline__ void pb_cell_role(const PbArgs& a, char* smem) {
    typedef typename CT<PREC>::T T;
    constexpr int VEC = CT<PREC>::VEC, KSTEP = CT<PREC>::KSTEP;
    const PbGeom& g = a.g;
    const int j = blockIdx.x / g.NS, bs = blockIdx.x - j * g.NS;
    const int U = g.U, j0 = j * U, b0 = bs * g.Bs, Bl = min(g.Bs, a.B - b0);
    const int B = a.B, C = a.C, E = a.E, A = a.A, XI = C + E, NX = g.NX, Ap = g.Ap;
    const int K4 = 4 * U, ldk = K4 + VEC, ldq = Ap + VEC, NKS = K4 / KSTEP, NTN = NX / 16;
    T* WT = (T*)smem;                                   // [NX][ldk]   rows n: [0,E) W_ih[my rows][C + n], [E,E+C) W_hh[my rows][n - E]; k = gate*U + unit
    T* Dl = WT + (size_t)NX * ldk;                      // [NB*16][ldk] my d gates of this step
    T* WqT = Dl + NB * 16 * ldk;                        // [16][ldq]   W_phi[a][j0 + u] as rows u, k = a
    T* Ql = WqT + 16 * ldq;                             // [NB*16][ldq] d q_pre rows of my batch slice
    T* Pl = Ql + NB * 16 * ldq;                         // [NCT][NB*16][U] pieces addressed to me
    float* Gl = (float*)(Pl + (size_t)g.NCT * NB * 16 * U);      // [PNW][NB*16][17]

    for (int i = threadIdx.x; i < NX * ldk; i += PNT) {
        const int k = i % ldk, n = i / ldk, gi = k / U, u = k - gi * U;
        float v = 0.f;
        if (k < K4 && j0 + u < C) {
            const long wr = (long)gi * C + j0 + u;
            if (n < E) v = a.w_ih[wr * XI + C + n];
            else if (n - E < C) v = a.w_hh[wr * C + (n - E)];
        }
        WT[i] = to_ct<T>(v);
    }
    for (int i = threadIdx.x; i < NB * 16 * ldk; i += PNT) Dl[i] = (T)0;
    for (int i = threadIdx.x; i < 16 * ldq; i += PNT) {
        const int k = i % ldq, u = i / ldq;
        WqT[i] = to_ct<T>((u < U && j0 + u < C && k < A) ? a.w_phi[(long)k * C + j0 + u] : 0.f);
    }
    for (int i = threadIdx.x; i < NB * 16 * ldq; i += PNT) Ql[i] = (T)0;
    __syncthreads();

    const int er = threadIdx.x / U, en = threadIdx.x - er * U, ej = j0 + en;
    const bool ev = er < Bl && ej < C && threadIdx.x < NB * 16 * U;
    float dc_carry = 0.f;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
    unsigned* abort_word = a.sync->abort_;
    PK_STAMP_DECL;

    for (int t = a.L - 1, n = 1; t >= 0; --t, ++n) {
        // saved activations of my element
        float gi_ = 0.f, gf_ = 0.f, gg_ = 0.f, go_ = 0.f, ct = 0.f, cp = 0.f, dh = 0.f;
        if (ev) {
            const long ro = (long)t * B + b0 + er;
            const float* gp = a.gates + ro * 4 * C;
            gi_ = gp[ej]; gf_ = gp[C + ej]; gg_ = gp[2 * C + ej]; go_ = gp[3 * C + ej];
            ct = a.cs[(ro + B) * C + ej]; cp = a.cs[ro * C + ej];
            dh = a.g_htop[ro * C + ej];
        }
        PK_STAMP(0);
        if (t + 1 < a.L) {
            // ---- recurrent d h_t = sum of the pieces of step t+1 addressed to my units + d q_pre_{t+1} W_phi
            // (tagged granules, as d q_pre below: the pieces of step t+1 were stored long ago, the first pass matches)
            {
                constexpr int V8 = GrT<T>::V8;
                const int NGX = NX / V8, gpr = U / V8, total = g.NCT * Bl * gpr;
                __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)a.px + (size_t)((t + 1) & 1) * g.NCT * B * NGX * 16), 0,
                                                                             g.NCT * B * NGX * 16, 0x00020000);
                auto sweep = [&](auto swv) -> bool {
                    constexpr int SW = decltype(swv)::value;
                    int off[SW], dst[SW];
#pragma unroll
                    for (int u = 0; u < SW; ++u) {
                        const int i = threadIdx.x + u * PNT, pj = i / (Bl * gpr), rem = i - pj * (Bl * gpr), r = rem / gpr, gc = rem - r * gpr;
                        off[u] = i < total ? ((pj * B + b0 + r) * NGX + (E + j0) / V8 + gc) * 16 : GR_OOB;
                        dst[u] = (pj * NB * 16 + r) * U + gc * V8;
                    }
                    return pk_gr_sweep<SW>(rp, off, (unsigned)(n - 1), abort_word, [&](int u, const u32x4& gv) { pk_gr_scatter<T, 2>(Pl + dst[u], gv); });
                };
                const bool ok = total <= 2 * PNT ? sweep(std::integral_constant<int, 2>{}) : total <= 4 * PNT ? sweep(std::integral_constant<int, 4>{})
                                                                                                             : sweep(std::integral_constant<int, 8>{});
                if (!ok) {
                    if (threadIdx.x == 0) *a.status = LAS_E_TIMEOUT;
                    return;
                }
            }
            // ... and summed over the producers NOW, before the wait for d q_pre: the pieces of step t+1 are long there, while the
            // d q_pre term is on the loop's chain (as one sum behind the d q_pre product this was 40 dependent LDS reads per thread
            // between the attention role's publication and the cells' own)
            __syncthreads();
            float vp = 0.f;
            if (ev) {
                float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
                int pj = 0;
                for (; pj + 3 < g.NCT; pj += 4) {
                    v0 += ct2f(Pl[((size_t)pj * NB * 16 + er) * U + en]);
                    v1 += ct2f(Pl[((size_t)(pj + 1) * NB * 16 + er) * U + en]);
                    v2 += ct2f(Pl[((size_t)(pj + 2) * NB * 16 + er) * U + en]);
                    v3 += ct2f(Pl[((size_t)(pj + 3) * NB * 16 + er) * U + en]);
                }
                for (; pj < g.NCT; ++pj) v0 += ct2f(Pl[((size_t)pj * NB * 16 + er) * U + en]);
                vp = (v0 + v1) + (v2 + v3);
            }
            PK_STAMP(1);
            // d q_pre of step t+1 arrives as tagged granules (pk_common.h): no counter, no poll-then-pull -- the sweep IS the
            // pull, one L2 round trip once the last part has stored (this hand-off is on the loop's chain; as flag + data it
            // cost a drain, a barrier, an atomic, a poll round trip, a barrier and the pull's round trip)
            {
                constexpr int V8 = GrT<T>::V8;
                const int NGr = Ap / V8, NGu = min(NGr, g.NCH * g.AS / V8), total = Bl * NGu;
                __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)a.dqq + ((size_t)((t + 1) & 1) * B + b0) * NGr * 16), 0,
                                                                             Bl * NGr * 16, 0x00020000);
                auto sweep = [&](auto swv) -> bool {
                    constexpr int SW = decltype(swv)::value;
                    int off[SW], dst[SW];
#pragma unroll
                    for (int u = 0; u < SW; ++u) {
                        const int i = threadIdx.x + u * PNT, r = i / NGu, gc = i - r * NGu;
                        off[u] = i < total ? (r * NGr + gc) * 16 : GR_OOB;
                        dst[u] = r * ldq + gc * V8;
                    }
                    return pk_gr_sweep<SW>(rq, off, (unsigned)(n - 1), abort_word, [&](int u, const u32x4& gv) { pk_gr_scatter<T, 2>(Ql + dst[u], gv); });
                };
                const bool ok = total <= 2 * PNT ? sweep(std::integral_constant<int, 2>{}) : total <= 4 * PNT ? sweep(std::integral_constant<int, 4>{})
                                                                                                             : sweep(std::integral_constant<int, 8>{});
                if (!ok) {
                    if (threadIdx.x == 0) *a.status = LAS_E_TIMEOUT;
                    return;
                }
            }
            PK_STAMP(2);
            __syncthreads();
            f32x4 qa[NB];
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) qa[bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int ks = wave; ks < Ap / KSTEP; ks += PNW) mma_rows<PREC, NB>(qa, Ql + ks * KSTEP, ldq, WqT + ks * KSTEP, ldq, 1);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Gl[(wave * NB * 16 + bt * 16 + fq * 4 + r) * 17 + fr] = qa[bt][r];
            __syncthreads();
            if (ev) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < PNW; ++w) v += Gl[(w * NB * 16 + er) * 17 + en];
                dh += v + vp;
            }
            PK_STAMP(3);
        }
        // ---- cell pointwise backward (reference asr.py:353 LSTMCell, gate order i,f,g,o)
        float dg[4] = {0.f, 0.f, 0.f, 0.f};
        if (ev) {
            const float tc = fast_tanh(ct);
            const float dc = dh * go_ * (1.f - tc * tc) + dc_carry;
            dg[0] = dc * gg_ * gi_ * (1.f - gi_);
            dg[1] = dc * cp * gf_ * (1.f - gf_);
            dg[2] = dc * gi_ * (1.f - gg_ * gg_);
            dg[3] = dh * tc * go_ * (1.f - go_);
            dc_carry = dc * gf_;
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) Dl[er * ldk + gi * U + en] = to_ct<T>(dg[gi]);
        }
        __syncthreads();
        // ---- K-split product: my d gates slice x my rows of [W_ih(ctx) | W_hh], transposed so that a lane ends up with 4
        // consecutive output columns of one batch row (one 8-byte store per tile)
        {
            constexpr int V8 = GrT<T>::V8;
            const int NGX = NX / V8;
            __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)a.px + ((size_t)(t & 1) * g.NCT + j) * B * NGX * 16), 0,
                                                                         B * NGX * 16, 0x00020000);
            for (int tile = wave; tile < NTN; tile += PNW) {
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) {
                    f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
                    mma_rows<PREC, 1>(acc, WT + (size_t)tile * 16 * ldk, ldk, Dl + bt * 16 * ldk, ldk, NKS);
                    // a lane's four columns of one batch row leave as tagged granules: payload + step count in ONE 16-byte sc1
                    // store (bf16: one granule; f32: two), no drain, barrier or counter behind them
                    const int row = bt * 16 + fr, g0 = ((b0 + row) * NGX + (tile * 16 + fq * 4) / V8) * 16;
                    if constexpr (PREC == LAS_PREC_BF16) {
                        pk_gr_store(rp, row < Bl ? g0 : GR_OOB, pack_bf16x2(acc[0][0], acc[0][1]), pack_bf16x2(acc[0][2], acc[0][3]), 0u, (unsigned)n);
                    } else {
                        pk_gr_store(rp, row < Bl ? g0 : GR_OOB, __float_as_uint(acc[0][0]), __float_as_uint(acc[0][1]), 0u, (unsigned)n);
                        pk_gr_store(rp, row < Bl ? g0 + 16 : GR_OOB, __float_as_uint(acc[0][2]), __float_as_uint(acc[0][3]), 0u, (unsigned)n);
                    }
                }
            }
        }
        PK_STAMP(4);
        if (ev) {
            float* go = a.dgates + ((long)t * B + b0 + er) * 4 * C;
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) __builtin_nontemporal_store(dg[gi], &go[gi * C + ej]);
        }
        PK_STAMP(5);
    }
    PK_STAMP_FLUSH(a.dbg);
}

// ---- attention role ----------------------------------------------------------------------------------------------
// 16-byte vectors per channel row of the d f window (frames r0 - K .. r0 + TC + K and a margin): see pb_att_role
__host__ __device__ inline int pb_window_vecs(int TCq) {
    const int need = TCq + (2 * LOC_K + 12) / 4;
    return need + ((TCq - need) % 16 + 16) % 16;
}

template <int PREC, int MT, int NTW, bool LOC>      // LOC: location-aware attention (false: dot -- no conv / u / d f phases); a
// compile-time switch: as a run-time flag it cost the location-aware BPTT loop 0.7 ms at c3 (registers, 16 spilled)
__device__ __forceinline__ void pb_att_role(const PbArgs& a, char* smem) {
    typedef typename CT<PREC>::T T;
    constexpr int VEC = CT<PREC>::VEC, KSTEP = CT<PREC>::KSTEP, LDK = KSTEP + VEC;
    const PbGeom& g = a.g;
    // block -> (utterance b, part c).  XCD-grouped: block ids congruent mod 8 share an XCD (observed, not contractual; NCELL is a
    // multiple of 8 then), so utterance b = 8 slot + x takes the ids == x (mod 8); ids without an utterance exit at once
    int b, c;
    {
        const int id = blockIdx.x - g.NCELL;
        if (g.xl) { const int m = id >> 3; b = (m / g.NCH) * 8 + (id & 7); c = m % g.NCH; }
        else { b = id / g.NCH; c = id - b * g.NCH; }
    }
    if (b >= a.B) return;
    const int B = a.B, Tp = a.Tp, E = a.E, A = a.A, C = a.C, XI = C + E, Ap = g.Ap, NX = g.NX;
    const int len = a.lens[b];
    const int TC = g.TC, r0 = c * TC, TCr = max(0, min(TC, Tp - r0)), tcv = max(0, min(TCr, len - r0));
    // enc rows: whole MFMA k-steps, 16-byte aligned, and a row stride (ESP + one vector) that spreads the 16 rows of an A
    // fragment over all 64 banks (bf16: 52 dwords, f32: == 4 mod 16)
    const int ES = g.ES, e0 = c * ES, ESr = max(0, min(ES, E - e0)), ESP = (ES + KSTEP - 1) / KSTEP * KSTEP, ESp = ESP + VEC;
    const int Tp16 = (Tp + 15) & ~15;
    const int TCq = (TC + 3) / 4, Tp4 = (Tp + 3) & ~3, lda_ = Ap + VEC;
    const int a0 = c * g.AS, ASr = max(0, min(g.AS, A - a0));          // my d q_pre columns
    // d f window of the conv path: frames r0 - K .. r0 + TC + K; an ODD number of 16-byte vectors per channel row, so that the
    // ten channels' windows do not sit on the same LDS banks (with 256-float rows the conv's reads were 6-way conflicts)
    // row length in 16-byte vectors == TCq (mod 16): thread (channel cc, frame quad qd) then reads vector cc * TCq + qd (mod 16) of
    // the 16 bank groups, i.e. the 16 lanes a b128 read serves at once hit 16 different groups (with 65 vectors per row the
    // lanes of channel cc + 1 sat on those of channel cc: two-way conflicts on every window read)
    const int WN = 4 * pb_window_vecs(TCq);
    // LDS (every float array 16-byte aligned)
    T* enc_l = (T*)smem;                                         // [Tp16][ESp], zero beyond the utterance / my slice
    T* dctxT_l = enc_l + (size_t)Tp16 * ESp;                     // [ESP] d ctx of my slice as an MFMA operand row
    float* cwf_l = (float*)(smem + ((((size_t)Tp16 * ESp + ESP) * sizeof(T) + 15) & ~(size_t)15));      // [10][LWP] FLIPPED taps, zero padded
    float* att_l = cwf_l + LOC_C * LWP;                          // [Tp4]
    float* da_l = att_l + Tp4;                                   // [Tp4]
    float* dctx_l = da_l + Tp4;                                  // [ES]
    float* de_l = dctx_l + ES;                                   // [MT*16]
    float* ct_l = de_l + MT * 16;                                // [MT*16] conv-path term of my frames
    float* dqp_l = ct_l + MT * 16;                               // [4][NTW*128] d q partials per lane row group
    float* dfa_l = dqp_l + 4 * NTW * PNW * 16;                   // [MT*16][16] d f accumulators (LDS float adds)
    float* red = dfa_l + MT * 16 * 16;                           // [64]
    int* flag = (int*)(red + 64);                                // [4]
    T* Wt = (T*)(flag + 4);                                      // [NTW*128][LDK] W_lp rows (a x channels) for u = tanh(W_lp f)
    T* WlT = Wt + NTW * PNW * 16 * LDK;                          // [16][lda_] W_lp^T rows (channel x a) for d f = d u W_lp
    T* Ft = WlT + 16 * lda_;                                     // [MT*16][LDK] f of my frames
    // phase A: d f window + conv partials; phase E: d u tile.  The 16-byte alignment is applied to the OFFSET from smem: a
    // pointer rounded through uintptr_t loses its LDS address space, and every access through it (the conv loop's window
    // reads, the piece sums, the d u tile) compiled to FLAT loads / stores -- 11 000 cycles for the conv loop alone
    const size_t scratch_off = ((size_t)((char*)(Ft + MT * 16 * LDK) - smem) + 15) & ~(size_t)15;
    float* dfn_l = (float*)(smem + scratch_off);                 // [10][WN]
    float* f4_l = dfn_l + LOC_C * WN;                            // [NSEG][10][TCq][4]
    T* Du = (T*)dfn_l;                                           // [MT*16][lda_]   (aliases the two above)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;

    // ---- prologue
    for (int i = threadIdx.x; i < ESP; i += PNT) dctxT_l[i] = (T)0;
    for (int i = threadIdx.x; i < Tp16 * ESp; i += PNT) {
        const int tp = i / ESp, col = i - tp * ESp;
        enc_l[i] = to_ct<T>((col < ESr && tp < len) ? a.enc[((long)b * Tp + tp) * E + e0 + col] : 0.f);
    }
    for (int i = threadIdx.x; i < NTW * PNW * 16 * LDK; i += PNT) {
        const int k = i % LDK, aa = i / LDK;
        Wt[i] = to_ct<T>((LOC && k < LOC_C && aa < A) ? a.w_lp[(long)aa * LOC_C + k] : 0.f);
    }
    for (int i = threadIdx.x; i < 16 * lda_; i += PNT) {
        const int k = i % lda_, cc = i / lda_;
        WlT[i] = to_ct<T>((LOC && cc < LOC_C && k < A) ? a.w_lp[(long)k * LOC_C + cc] : 0.f);
    }
    for (int i = threadIdx.x; i < MT * 16 * LDK; i += PNT) Ft[i] = (T)0;
    for (int i = threadIdx.x; i < LOC_C * LWP; i += PNT) {
        const int cc = i / LWP, k = i - cc * LWP;                                // flipped: w'[c][k'] = w[c][2K - k']
        cwf_l[i] = (LOC && k < LOC_W) ? a.conv_w[cc * LOC_W + (LOC_W - 1 - k)] : 0.f;
    }
    for (int i = threadIdx.x; i < MT * 16; i += PNT) ct_l[i] = 0.f;
    __syncthreads();
    float wev[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) { const int aa = (wave + PNW * j) * 16 + fr; wev[j] = (LOC && aa < A) ? a.w_e[aa] : 0.f; }
    unsigned* abort_word = a.sync->abort_;
    unsigned nwait = 0;
    const int loc_ = pk_utt_local(&a.sync->cnt_da[b][0], g.NCH, g.xl != 0, abort_word, flag + 2);
    if (loc_ < 0) {
        if (threadIdx.x == 0) *a.status = LAS_E_TIMEOUT;
        return;
    }
    const bool local = loc_ > 0;            // my utterance's parts share an XCD: L2-local exchanges among them
    const int cv_sg = threadIdx.x / (LOC_C * TCq), cv_cc = (threadIdx.x - cv_sg * (LOC_C * TCq)) / TCq,
              cv_qd = threadIdx.x - cv_sg * (LOC_C * TCq) - cv_cc * TCq;
    // window origin: dfn_l[cc][x] = d f_next[cc][r0 - LOC_K - sh + x], sh chosen so that frame r0's window start is 16-byte aligned
    float sv[MT][NTW][4];
    auto load_s = [&](int t_) {                                  // saved s of step t_ for my elements (clamped addresses)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tp = min(r0 + mt * 16 + fq * 4 + r, Tp - 1);
                const long so = (((long)t_ * B + b) * Tp + tp) * A;
#pragma unroll
                for (int j = 0; j < NTW; ++j) {            // bf16 mode: sv = 1 - |s| (all phase E needs: 1 - s^2 = sv (2 - sv))
                    const int aa = min((wave + PNW * j) * 16 + fr, A - 1);
                    if constexpr (PREC == LAS_PREC_BF16) sv[mt][j][r] = las_s16_t(((const bf16_t*)a.s)[so + aa]);
                    else sv[mt][j][r] = ((const float*)a.s)[so + aa];
                }
            }
    };
    float pf_att[2] = {0.f, 0.f}, pf_f = 0.f, pf_q = 0.f;
    const int pf_cc = (int)threadIdx.x / TC, pf_tt = (int)threadIdx.x - pf_cc * TC;
    auto prefetch = [&](int t_) {                                // saved att (step t_ + 1's slot), f, q of step t_
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = threadIdx.x + j * PNT;
            if (i < Tp) pf_att[j] = a.att[((long)(t_ + 1) * B + b) * Tp + i];
        }
        if (LOC && (int)threadIdx.x < LOC_C * TC)
            pf_f = pf_tt < TCr ? a.f[(((long)t_ * B + b) * LOC_C + pf_cc) * Tp + r0 + pf_tt] : 0.f;
        pf_q = a.q[((long)t_ * B + b) * A + min(a0 + (int)threadIdx.x, A - 1)];
    };
    prefetch(a.L - 1);
    if (!LOC) {                                            // dot attention: d e reaches d q through psi (e = psi . q), the same every step
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tp = min(r0 + mt * 16 + fq * 4 + r, Tp - 1);
                const float* __restrict__ pp = a.psi + ((long)b * Tp + tp) * A;
#pragma unroll
                for (int j = 0; j < NTW; ++j) { const int aa = (wave + PNW * j) * 16 + fr; sv[mt][j][r] = aa < A ? pp[aa] : 0.f; }
            }
    }
    PK_STAMP_DECL;

    for (int t = a.L - 1, n = 1; t >= 0; --t, ++n) {
        // ---- (A) everything that needs no d ctx_t: attention, u = tanh(W_lp f_t), the conv path of d f_{t+1}
        // the step's saved att / f / q were requested one step ago (HBM-cold reads: waiting for them here was 3 700
        // cycles of every step); now into LDS, and the next step's are requested
        for (int i = threadIdx.x, j = 0; i < Tp; i += PNT, ++j)
            att_l[i] = j < 2 ? pf_att[j] : a.att[((long)(t + 1) * B + b) * Tp + i];
        if ((int)threadIdx.x < LOC_C * TC) Ft[pf_tt * LDK + pf_cc] = to_ct<T>(pf_f);
        for (int i = threadIdx.x + PNT; LOC && i < LOC_C * TC; i += PNT) {          // (TC > 51: not a geometry this loop is given)
            const int cc = i / TC, tt = i - cc * TC;
            Ft[tt * LDK + cc] = to_ct<T>(tt < TCr ? a.f[(((long)t * B + b) * LOC_C + cc) * Tp + r0 + tt] : 0.f);
        }
        const float qq_f = pf_q;                                             // (for phase F)
        if (t > 0) prefetch(t - 1);
        PK_STAMP(9);
        if (LOC && t + 1 < a.L) {
            // d f_{t+1} of the frames around my chunk (published by my utterance's parts at the end of step t+1)
            if (!pk_block_wait_x(&a.sync->cnt_df[b][0], g.NCH, (unsigned)(n - 1), local, abort_word, flag + (nwait++ & 1))) {
                if (threadIdx.x == 0) *a.status = LAS_E_TIMEOUT;
                return;
            }
            // (16-byte loads: chunks start on multiples of 4 frames and the exchange rows have stride Tp4; frames beyond T' are
            // never written and read as the zeros of the initial memset, frames < 0 are whole vectors)
            {
                const float* src = a.dfx + ((size_t)((t + 1) & 1) * B + b) * LOC_C * Tp4;
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, LOC_C * Tp4 * 4, 0x00020000);
                const int wq = TCq + (2 * LOC_K + 12) / 4;             // (the vectors the conv reads; the row is longer, see WN)
                for (int i = threadIdx.x; i < LOC_C * wq; i += PNT) {
                    const int cc = i / wq, x4 = i - cc * wq, tp = r0 - LOC_K + 4 * x4;
                    u32x4 v = {0u, 0u, 0u, 0u};
                    if (tp >= 0 && tp < Tp4) v = __builtin_amdgcn_raw_buffer_load_b128(rs, (cc * Tp4 + tp) * 4, 0, 16);
                    *(u32x4*)(dfn_l + cc * WN + 4 * x4) = v;
                }
            }
            __syncthreads();
            PK_STAMP(10);
            // conv path: ct[t'] = sum_c sum_k' w'[c][k'] d f_{t+1}[c][t' + k' - K]   (the forward conv's form, flipped taps)
            for (int i = threadIdx.x; i < NSEG * LOC_C * TCq; i += PNT) {
                int sg = cv_sg, cc = cv_cc, qd = cv_qd;
                if (i >= PNT) { sg = i / (LOC_C * TCq); const int rem = i - sg * (LOC_C * TCq); cc = rem / TCq; qd = rem - cc * TCq; }
                if (4 * qd >= TCr) { *(float4*)(f4_l + (size_t)i * 4) = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
                const float* w = cwf_l + cc * LWP + sg * SEGW;
                const float* p = dfn_l + cc * WN + 4 * qd + sg * SEGW;
                float4 lo = *(const float4*)p;
                float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll 4
                for (int k = 0; k < SEGW; k += 4) {
                    const float4 wv = *(const float4*)(w + k), hi = *(const float4*)(p + k + 4);
                    o0 = fmaf(wv.x, lo.x, fmaf(wv.y, lo.y, fmaf(wv.z, lo.z, fmaf(wv.w, lo.w, o0))));
                    o1 = fmaf(wv.x, lo.y, fmaf(wv.y, lo.z, fmaf(wv.z, lo.w, fmaf(wv.w, hi.x, o1))));
                    o2 = fmaf(wv.x, lo.z, fmaf(wv.y, lo.w, fmaf(wv.z, hi.x, fmaf(wv.w, hi.y, o2))));
                    o3 = fmaf(wv.x, lo.w, fmaf(wv.y, hi.x, fmaf(wv.z, hi.y, fmaf(wv.w, hi.z, o3))));
                    lo = hi;
                }
                *(float4*)(f4_l + (size_t)i * 4) = make_float4(o0, o1, o2, o3);
            }
            PK_STAMP(17);
            __syncthreads();
            float cpart = 0.f;
            if (threadIdx.x < 4 * TC) {                       // thread = (frame, tap segment): its 10 channel partials
                const int tt = threadIdx.x >> 2, sg = threadIdx.x & 3;
#pragma unroll
                for (int cc = 0; cc < LOC_C; ++cc) cpart += f4_l[(((size_t)sg * LOC_C + cc) * TCq + (tt >> 2)) * 4 + (tt & 3)];
            }
            cpart += las_dpp<0x111, 0xf>(0.f, cpart);        // + lane - 1, + lane - 2: lane 4 tt + 3 ends with the frame's total
            cpart += las_dpp<0x112, 0xf>(0.f, cpart);
            if (threadIdx.x < 4 * TC && (threadIdx.x & 3) == 3) ct_l[threadIdx.x >> 2] = cpart;
        }
        __syncthreads();
        PK_STAMP(18);
        PK_STAMP(0);
        // 1 - u^2 of my elements (u = tanh(F W_lp^T) on the MFMA), kept in registers
        // (bf16 mode: kept as bf16 PAIRS, 18 registers instead of 36 -- d u = d z (1 - u^2) is rounded to bf16 for the d f product
        // anyway; the registers are what lets the piece sweep below run without scratch spills)
        constexpr int UMW = PREC == LAS_PREC_BF16 ? 2 : 4;
        unsigned um[MT][NTW][UMW];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
                if (LOC) mma_rows<PREC, 1>(acc, Ft + mt * 16 * LDK, LDK, Wt + (wave + PNW * j) * 16 * LDK, LDK, 1);
                float uu[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float u = fast_tanh(acc[0][r]);
                    uu[r] = 1.f - u * u;
                }
                if constexpr (PREC == LAS_PREC_BF16) {
                    um[mt][j][0] = pack_bf16x2(uu[0], uu[1]);
                    um[mt][j][1] = pack_bf16x2(uu[2], uu[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) um[mt][j][r] = __float_as_uint(uu[r]);
                }
#pragma unroll
                for (int r = 0; r < UMW; ++r) asm volatile("" : "+v"(um[mt][j][r]));
            }
        PK_STAMP(1);
        // ---- (B) d ctx of my E-slice: sum of the cell role's pieces of this step
        PK_STAMP(2);
        {
            // the pieces arrive as tagged granules: the sweep is wait and pull in one (one L2 round trip once the last cell has
            // stored); a matched granule goes to psum_l[producer][ES] as fp32 (in the scratch region, free between the conv path
            // and the d u tile); then thread (column, quarter of the producers), then 4 partials.  ONE granule per lane and pass
            // over the list: two at once cost registers the role does not have here (it holds 1 - u^2 of its 36 elements).
            // (LDS float atomics for the sums instead: 12 900 cycles -- they retire about one LANE per 3 cycles, cycle stamps)
            float* psum_l = dfn_l;                                           // [NCT][ES]
            constexpr int V8 = GrT<T>::V8;
            const int NGX = NX / V8, gpe = ES / V8, total = g.NCT * gpe;
            __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)a.px + (size_t)(t & 1) * g.NCT * B * NGX * 16), 0,
                                                                         g.NCT * B * NGX * 16, 0x00020000);
            bool ok = true;
            // two granules per lane and sweep (the 960 of the c3 shape are then ONE pass, one L2 round trip on the chain, instead of two)
            for (int i0 = 0; i0 < total && ok; i0 += 2 * PNT) {
                int off[2], dst[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = i0 + u * PNT + threadIdx.x, pj = i / gpe, gc = i - pj * gpe, col = e0 + gc * V8;
                    dst[u] = pj * ES + gc * V8;
                    off[u] = (i < total && col < E) ? ((pj * B + b) * NGX + col / V8) * 16 : GR_OOB;
                    if (i < total && col >= E) {                             // beyond E (last part): not d ctx columns, zeros
#pragma unroll
                        for (int e = 0; e < V8; ++e) psum_l[dst[u] + e] = 0.f;
                    }
                }
                ok = pk_gr_sweep<2>(rp, off, (unsigned)n, abort_word, [&](int u, const u32x4& gv) {
                    float* o = psum_l + dst[u];
                    if constexpr (PREC == LAS_PREC_BF16)
                        *(float4*)o = make_float4(__uint_as_float(gv[0] << 16), __uint_as_float(gv[0] & 0xffff0000u),
                                                  __uint_as_float(gv[1] << 16), __uint_as_float(gv[1] & 0xffff0000u));
                    else
                        *(float2*)o = make_float2(__uint_as_float(gv[0]), __uint_as_float(gv[1]));
                });
            }
            if (!ok) {
                if (threadIdx.x == 0) *a.status = LAS_E_TIMEOUT;
                return;
            }
            __syncthreads();
            float part4 = 0.f;
            if (threadIdx.x < 4 * ES) {
                const int e = threadIdx.x >> 2, qd = threadIdx.x & 3;
                // (five reads requested before the first add: as a rolled loop over the producers this was ten dependent LDS round
                // trips on the chain; all ten at once do not fit the role's registers)
                constexpr int PQ = 5;                                        // producers per round trip and thread
#pragma unroll 1
                for (int p0 = qd; p0 < g.NCT; p0 += 4 * PQ) {
                    float pv4[PQ];
#pragma unroll
                    for (int k = 0; k < PQ; ++k) pv4[k] = psum_l[min(p0 + 4 * k, g.NCT - 1) * ES + e];
#pragma unroll
                    for (int k = 0; k < PQ; ++k) part4 += (p0 + 4 * k < g.NCT) ? pv4[k] : 0.f;
                }
            }
            part4 += las_dpp<0x111, 0xf>(0.f, part4);
            part4 += las_dpp<0x112, 0xf>(0.f, part4);
            if (threadIdx.x < 4 * ES && (threadIdx.x & 3) == 3) { dctx_l[threadIdx.x >> 2] = part4; dctxT_l[threadIdx.x >> 2] = to_ct<T>(part4); }
        }
        __syncthreads();
        if (LOC) load_s(t);           // requested only now (its 36 registers would be live across the piece sweep): in flight during
                                        // the d a product, its exchange and the softmax backward; first used in phase E
        PK_STAMP(11);
        // ---- (C) d a over my E-slice for every frame of the utterance (+ the conv path for my own frames)
        // on the matrix cores: A = 16 frames x my slice of enc (LDS), B = the d ctx row for all 16 columns (ldb = 0: every lane of
        // a row group ends with the same four frames' sums); a wave per 16-frame tile.  (A thread per frame walking its row with
        // scalar FMAs was 3 900 cycles of every step, on the chain between the cells' pieces and the softmax backward.)
        for (int tile = wave; tile * 16 < len; tile += PNW) {
            f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
            mma_rows<PREC, 1>(acc, enc_l + (size_t)tile * 16 * ESp, ESp, dctxT_l, 0, ESP / KSTEP);
            const int tp = tile * 16 + fq * 4 + fr;                          // lane fr < 4 of a row group stores row fr
            float v = fr == 0 ? acc[0][0] : fr == 1 ? acc[0][1] : fr == 2 ? acc[0][2] : acc[0][3];
            if (fr < 4 && tp < len) {
                if (tp >= r0 && tp < r0 + TCr) v += ct_l[tp - r0];
                st_x(a.dax + (((size_t)(t & 1) * B + b) * g.NCH + c) * Tp + tp, v, local);
            }
        }
        PK_STAMP(12);
        pk_signal_x(&a.sync->cnt_da[b][0], c, (unsigned)n, local);
        PK_STAMP(3);
        if (threadIdx.x < ESr) a.dxin[((long)t * B + b) * XI + C + e0 + threadIdx.x] = dctx_l[threadIdx.x];
        // ---- (D) all parts' d a: softmax backward for my frames
        if (!pk_block_wait_x(&a.sync->cnt_da[b][0], g.NCH, (unsigned)n, local, abort_word, flag + (nwait++ & 1))) {
            if (threadIdx.x == 0) *a.status = LAS_E_TIMEOUT;
            return;
        }
        PK_STAMP(4);
        float part = 0.f;
        __amdgpu_buffer_rsrc_t rs_da = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dax + ((size_t)(t & 1) * B + b) * g.NCH * Tp), 0,
                                                                         g.NCH * Tp * 4, 0x00020000);
        for (int tp = threadIdx.x; tp < len; tp += PNT) {
            float v = 0.f, pv_[16];
#pragma unroll
            for (int pc = 0; pc < 16; ++pc) pv_[pc] = pc < g.NCH ? ldb_sc1(rs_da, (pc * Tp + tp) * 4) : 0.f;
#pragma unroll
            for (int pc = 0; pc < 16; ++pc) v += pv_[pc];
            da_l[tp] = v;
            part = fmaf(att_l[tp], v, part);
        }
        const float dot = block_sum(part, red);              // (its barriers also publish da_l)
        if (threadIdx.x < MT * 16) {
            const int tt = threadIdx.x, tp = r0 + tt;
            const float v = tt < tcv ? ATT_SCALE * att_l[tp] * (da_l[tp] - dot) : 0.f;
            de_l[tt] = v;
            if (tt < TCr) __builtin_nontemporal_store(v, &a.de[((long)t * B + b) * Tp + tp]);
        }
        __syncthreads();
        PK_STAMP(5);
        // ---- (E) energy backward: d z = d e w_e (1 - s^2); d q partial = sum over my frames; d u = d z (1 - u^2) -> LDS
        float dq[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) dq[j] = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tt = mt * 16 + fq * 4 + r;
                const float de = de_l[tt];
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    // (frames beyond the utterance: the saved s is undefined there -- the forward loop never writes it --, and 0 * NaN
                    // would poison d q, d f and every gradient behind them; a select, not a product with d e = 0)
                    const float s_ = sv[mt][j][r];
                    const float ds_ = PREC == LAS_PREC_BF16 ? s_ * (2.f - s_) : 1.f - s_ * s_;
                    const float dz = tt < tcv ? (LOC ? de * wev[j] * ds_ : de * s_) : 0.f;      // (dot: s_ holds psi)
                    dq[j] += dz;
                    if (!LOC) continue;
                    const int aa = (wave + PNW * j) * 16 + fr;
                    float umv;
                    if constexpr (PREC == LAS_PREC_BF16) umv = (r & 1) ? __uint_as_float(um[mt][j][r >> 1] & 0xffff0000u) : __uint_as_float(um[mt][j][r >> 1] << 16);
                    else umv = __uint_as_float(um[mt][j][r]);
                    if (aa < Ap) Du[tt * lda_ + aa] = to_ct<T>(dz * umv);
                }
            }
        PK_STAMP(13);
#pragma unroll
        for (int j = 0; j < NTW; ++j) dqp_l[fq * (NTW * PNW * 16) + (wave + PNW * j) * 16 + fr] = dq[j];
        __syncthreads();
        for (int aa = threadIdx.x; aa < A; aa += PNT) {
            const float v = (dqp_l[aa] + dqp_l[NTW * PNW * 16 + aa]) + (dqp_l[2 * NTW * PNW * 16 + aa] + dqp_l[3 * NTW * PNW * 16 + aa]);
            st_x(a.dqx + (((size_t)(t & 1) * B + b) * g.NCH + c) * Ap + aa, v, local);
        }
        PK_STAMP(14);
        // d f[c][t'] = sum_a d u[t'][a] W_lp[a][c]: MFMA over k = a, the waves split the k-steps, LDS float adds combine them
        // (one wave per 16-frame tile walks all k-steps and keeps the sums in registers: with the k-steps split over the waves
        // the LDS float adds that combined them cost 14 900 cycles a step, cycle stamps)
        if (LOC && wave < MT) {
            f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
            mma_rows<PREC, 1>(acc, Du + wave * 16 * lda_, lda_, WlT, lda_, Ap / KSTEP);
#pragma unroll
            for (int r = 0; r < 4; ++r) dfa_l[(wave * 16 + fq * 4 + r) * 16 + fr] = acc[0][r];
        }
        __syncthreads();
        PK_STAMP(15);
        for (int i = threadIdx.x; LOC && i < LOC_C * TC; i += PNT) {
            const int cc = i / TC, tt = i - cc * TC;
            if (tt < TCr) st_x(a.dfx + (((size_t)(t & 1) * B + b) * LOC_C + cc) * Tp4 + r0 + tt, tt < tcv ? dfa_l[tt * 16 + cc] : 0.f, local);
        }
        PK_STAMP(16);
        pk_signal_x(&a.sync->cnt_dqp[b][0], c, (unsigned)n, local);
        if (LOC && threadIdx.x == PNT - 64) {             // (the same drain + barrier covers the d f stores)
            if (local) a.sync->cnt_df[b][1 + c] = (unsigned)n;
            else __hip_atomic_fetch_add(&a.sync->cnt_df[b][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        PK_STAMP(6);
        for (int i = threadIdx.x; LOC && i < LOC_C * TC; i += PNT) {
            const int cc = i / TC, tt = i - cc * TC;
            if (tt < tcv) __builtin_nontemporal_store(dfa_l[tt * 16 + cc], &a.df[(((long)t * B + b) * LOC_C + cc) * Tp + r0 + tt]);
        }
        // ---- (F) d q_pre of my a-slice: sum of the parts' partials, times (1 - q^2); to the cell role
        if (!pk_block_wait_x(&a.sync->cnt_dqp[b][0], g.NCH, (unsigned)n, local, abort_word, flag + (nwait++ & 1))) {
            if (threadIdx.x == 0) *a.status = LAS_E_TIMEOUT;
            return;
        }
        PK_STAMP(7);
        {
            const int k = threadIdx.x, aa = a0 + k;
            float v = 0.f;
            if (k < ASr) {
                __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dqx + ((size_t)(t & 1) * B + b) * g.NCH * Ap), 0,
                                                                                g.NCH * Ap * 4, 0x00020000);
                const float qq = qq_f;
                float pv_[16];
#pragma unroll
                for (int pc = 0; pc < 16; ++pc) pv_[pc] = pc < g.NCH ? ldb_sc1(rs_q, (pc * Ap + aa) * 4) : 0.f;
#pragma unroll
                for (int pc = 0; pc < 16; ++pc) v += pv_[pc];
                v *= 1.f - qq * qq;
            }
            // to the cells as tagged granules: lane k (a multiple of V8) gathers the V8 values that start at it over DPP and
            // stores {payload, tag = step count} with ONE 16-byte sc1 store; every one of my AS columns goes out (zeros beyond A)
            {
                constexpr int V8 = GrT<T>::V8;
                const int NGr = Ap / V8;
                unsigned w2[3];
                pk_gr_gather8<T>(k < ASr ? v : 0.f, w2);
                __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)a.dqq + ((size_t)(t & 1) * B + b) * NGr * 16), 0, NGr * 16, 0x00020000);
                pk_gr_store(rq, (k < g.AS && k % V8 == 0 && (a0 + k) / V8 < NGr) ? ((a0 + k) / V8) * 16 : GR_OOB, w2[0], w2[1], 0u, (unsigned)n);
            }
            if (k < ASr) a.dq_pre[((long)t * B + b) * A + aa] = v;
        }
        PK_STAMP(8);
    }
    PK_STAMP_FLUSH(a.dbg);
}

template <int PREC, int NB, int MT, int NTW, bool LOC>
__global__ __launch_bounds__(PNT) void dec_pk_bwd_kernel(PbArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < a.g.NCELL) pb_cell_role<PREC, NB>(a, smem);
    else pb_att_role<PREC, MT, NTW, LOC>(a, smem);
}

// ---- host side ---------------------------------------------------------------------------------------------------
size_t pb_cell_lds(int prec, const PbGeom& g) {
    const int sz = prec == LAS_PREC_BF16 ? 2 : 4, vec = prec == LAS_PREC_BF16 ? 8 : 4;
    return ((size_t)(g.NX + g.NB * 16) * (4 * g.U + vec) + (size_t)(16 + g.NB * 16) * (g.Ap + vec) + (size_t)g.NCT * g.NB * 16 * g.U) * sz +
           sizeof(float) * PNW * g.NB * 16 * 17 + 64;
}
size_t pb_att_lds(int prec, const PbGeom& g, int Tp, int A) {
    const int sz = prec == LAS_PREC_BF16 ? 2 : 4, vec = prec == LAS_PREC_BF16 ? 8 : 4, ldk = prec == LAS_PREC_BF16 ? 40 : 20;
    const size_t Tp4 = (Tp + 3) & ~3, TCq = (g.TC + 3) / 4, WN = 4 * (size_t)pb_window_vecs((int)TCq), lda_ = g.Ap + vec;
    const size_t scratchA = sizeof(float) * (LOC_C * WN + NSEG * LOC_C * TCq * 4), scratchE = (size_t)g.MT * 16 * lda_ * sz;
    const size_t ks = prec == LAS_PREC_BF16 ? 32 : 16, ESP = (g.ES + ks - 1) / ks * ks, Tp16 = ((size_t)Tp + 15) & ~(size_t)15;
    return (Tp16 * (ESP + vec) + ESP) * sz + 16 +
           sizeof(float) * (LOC_C * LWP + 2 * Tp4 + g.ES + 2 * g.MT * 16 + 4 * g.NTW * PNW * 16 + g.MT * 16 * 16 + 64 + 4) +
           ((size_t)g.NTW * PNW * 16 * ldk + 16 * lda_ + (size_t)g.MT * 16 * ldk) * sz + 16 + (scratchA > scratchE ? scratchA : scratchE) + 64;
}

bool pb_geom(const las_dec_dims* d, PbGeom& best) {
    if (!d || d->NL != 1 || d->dropout != 0.f || d->L < 1) return false;      // (dot and location-aware attention)
    if (d->B < 1 || d->B > MAXB || d->A > 512 || d->A < 1 || d->Tp < 1 || (d->C & 1)) return false;
    if (d->prec != LAS_PREC_BF16 && d->prec != LAS_PREC_F32) return false;
    if (las_fallback("LAS_DEC_NO_PK") || LAS_AB_KNOB("LAS_DEC_NO_PK_BWD")) return false;
    const int ks = d->prec == LAS_PREC_BF16 ? 32 : 16, vec = d->prec == LAS_PREC_BF16 ? 8 : 4;
    int want_ns = 0, want_u = 0;
    if (const char* e = LAS_AB_KNOB("LAS_DEC_PKB_CFG")) sscanf(e, "%d,%d", &want_ns, &want_u);      // (A/B measurements)
    static const int cand[4][2] = {{2, 8}, {1, 8}, {2, 16}, {1, 16}};
    for (int ci = 0; ci < 4; ++ci) {
        PbGeom g{};
        g.NS = cand[ci][0]; g.U = cand[ci][1];
        if (want_ns && (g.NS != want_ns || g.U != want_u)) continue;
        if (g.NS > 1 && d->B < 8) continue;
        if ((4 * g.U) % ks) continue;                    // my d gates slice must be whole k-steps
        g.Bs = (d->B + g.NS - 1) / g.NS;
        g.NS = (d->B + g.Bs - 1) / g.Bs;
        g.NB = g.Bs <= 16 ? 1 : 2;
        if (g.Bs > 32) continue;
        g.NCT = (d->C + g.U - 1) / g.U;
        g.NCELL = g.NCT * g.NS;
        g.NX = (d->E + d->C + 15) / 16 * 16;
        g.Ap = (d->A + ks - 1) / ks * ks;
        if ((g.U * (d->prec == LAS_PREC_BF16 ? 2 : 4)) % 16 || d->E % vec) continue;     // piece pulls are 16-byte vectors
        if (pb_cell_lds(d->prec, g) > PK_LDS_CAP) continue;
        g.NCH = (las_cu_count() - g.NCELL) / d->B;
        if (g.NCH > 16) g.NCH = 16;
        if (g.NCH < 1) continue;
        g.TC = ((d->Tp + g.NCH - 1) / g.NCH + 3) / 4 * 4;         // chunks start on multiples of 4 frames (16-byte window loads)
        if (g.TC > 64) continue;
        g.MT = g.TC <= 32 ? 2 : g.TC <= 48 ? 3 : 4;
        g.NTW = d->A <= 128 ? 1 : d->A <= 384 ? 3 : 4;
        if (g.NTW == 4 && g.MT == 4) continue;
        g.ES = ((d->E + g.NCH - 1) / g.NCH + vec - 1) / vec * vec;
        if (g.ES / 4 > PNT || g.ES > PNT) continue;
        { const int v8 = d->prec == LAS_PREC_BF16 ? 4 : 2; g.AS = ((d->A + g.NCH - 1) / g.NCH + v8 - 1) / v8 * v8; }      // whole d q_pre granules per part
        if (g.AS > PNT) continue;
        if (pb_att_lds(d->prec, g, d->Tp, d->A) > PK_LDS_CAP) continue;
        // XCD-grouped attention blocks: utterance b on the ids == b (mod 8) behind the cells (NCELL a multiple of 8), if the
        // padded grid still is one workgroup per CU
        g.xl = (!las_fallback("LAS_DEC_NO_XL") && g.NCELL % 8 == 0 && g.NCELL + 8 * ((d->B + 7) / 8) * g.NCH <= las_cu_count()) ? 1 : 0;
        g.lds = pb_cell_lds(d->prec, g);
        const size_t al = pb_att_lds(d->prec, g, d->Tp, d->A);
        if (al > g.lds) g.lds = al;
        if (g.lds < PK_MIN_LDS) g.lds = PK_MIN_LDS;
        best = g;
        return true;
    }
    return false;
}

struct PbWs { size_t sync, dbg, px, dax, dqx, dfx, dqq, total; };
PbWs pb_ws(const las_dec_dims* d, const PbGeom& g) {
    PbWs w;
    size_t o = 0;
    w.sync = o; o += las_align(sizeof(PbSync));
    w.dbg = o; o += las_align(sizeof(unsigned long long) * 256 * 20);
    w.px = o; o += las_align((size_t)2 * g.NCT * d->B * (g.NX / (d->prec == LAS_PREC_BF16 ? 4 : 2)) * 16);       // granules of 4 bf16 | 2 f32
    w.dax = o; o += las_align(sizeof(float) * 2 * d->B * g.NCH * d->Tp);
    w.dqx = o; o += las_align(sizeof(float) * 2 * d->B * g.NCH * g.Ap);
    w.dfx = o; o += las_align(sizeof(float) * 2 * d->B * LOC_C * ((d->Tp + 3) & ~3));
    w.dqq = o; o += las_align((size_t)2 * d->B * (g.Ap / (d->prec == LAS_PREC_BF16 ? 4 : 2)) * 16);       // granules of 4 bf16 | 2 f32
    w.total = o;
    return w;
}

}  // namespace

size_t las_dec_pk_bwd_ws_bytes(const las_dec_dims* d) {
    PbGeom g;
    if (!pb_geom(d, g)) return 0;
    return pb_ws(d, g).total;
}

int las_dec_pk_bwd(const las_dec_dims* d, const las_dec_params* p, const float* enc, const float* psi, const int32_t* enc_len,
                   const las_dec_state* st, const float* g_htop, las_dec_bwd_state* bw, hipStream_t stream) {
    PbGeom g;
    if (!pb_geom(d, g)) return LAS_E_UNSUPPORTED;
    LAS_CHECK_ARG(bw->pk_ws && bw->pk_status && bw->de && psi);
    if (d->loc) LAS_CHECK_ARG(st->f && st->s && bw->df && p->conv_w && p->w_lp && p->w_e);
    const PbWs w = pb_ws(d, g);
    char* ws = (char*)bw->pk_ws;
    const int B = d->B, C = d->C, E = d->E, L = d->L;
    LAS_HIP(hipMemsetAsync(ws, 0, w.total, stream));
    PbArgs a{};
    a.B = B; a.Tp = d->Tp; a.E = E; a.A = d->A; a.C = C; a.L = L; a.g = g; a.loc = d->loc ? 1 : 0;
    a.enc = enc; a.psi = psi; a.lens = enc_len;
    a.w_ih = p->w_ih[0]; a.w_hh = p->w_hh[0]; a.w_phi = p->w_phi; a.conv_w = p->conv_w; a.w_lp = p->w_lp; a.w_e = p->w_e;
    a.att = st->att; a.q = st->q; a.gates = st->gates; a.cs = st->cs; a.f = st->f; a.s = st->s; a.g_htop = g_htop;
    a.dgates = bw->dgates; a.dxin = bw->dxin; a.dq_pre = bw->dq_pre; a.de = bw->de; a.df = bw->df;
    a.px = ws + w.px; a.dax = (float*)(ws + w.dax); a.dqx = (float*)(ws + w.dqx); a.dfx = (float*)(ws + w.dfx); a.dqq = ws + w.dqq;
    a.sync = (PbSync*)(ws + w.sync); a.status = bw->pk_status; a.dbg = (unsigned long long*)(ws + w.dbg);
    const int grid = g.NCELL + (g.xl ? 8 * ((B + 7) / 8) * g.NCH : B * g.NCH);
    int launched = 0;
#define LAS_PB_GO(P_, N_, M_, W_)                                                                                  \
    {                                                                                                             \
        auto k = a.loc ? dec_pk_bwd_kernel<P_, N_, M_, W_, true> : dec_pk_bwd_kernel<P_, N_, M_, W_, false>;                                                               \
        LAS_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds));     \
        hipLaunchKernelGGL(k, dim3(grid), dim3(PNT), g.lds, stream, a);                                           \
        LAS_LAUNCH_OK();                                                                                          \
        launched = 1;                                                                                             \
    }
#define LAS_PB_W(P_, N_, M_)                                                                                       \
    {                                                                                                             \
        if (g.NTW == 1) LAS_PB_GO(P_, N_, M_, 1) else if (g.NTW == 3) LAS_PB_GO(P_, N_, M_, 3) else LAS_PB_GO(P_, N_, M_, 4) \
    }
#define LAS_PB_M(P_, N_)                                                                                           \
    {                                                                                                             \
        if (g.MT == 2) LAS_PB_W(P_, N_, 2) else if (g.MT == 3) LAS_PB_W(P_, N_, 3) else LAS_PB_W(P_, N_, 4)        \
    }
    if (d->prec == LAS_PREC_BF16) { if (g.NB == 1) LAS_PB_M(LAS_PREC_BF16, 1) else LAS_PB_M(LAS_PREC_BF16, 2) }
    else { if (g.NB == 1) LAS_PB_M(LAS_PREC_F32, 1) else LAS_PB_M(LAS_PREC_F32, 2) }
#undef LAS_PB_M
#undef LAS_PB_W
#undef LAS_PB_GO
    return launched ? LAS_OK : LAS_E_BADARG;
}

// d xin's embedding half for every step: d gates [L*B x 4C] x W_ih[:, 0:C] -- one GEMM, read by the embedding-row sums only
// (a parameter gradient: las_decoder_bwd_parts runs it with them, off the caller's main stream)
int las_dec_pk_bwd_emb(const las_dec_dims* d, const las_dec_params* p, las_dec_bwd_state* bw, hipStream_t stream) {
    const int B = d->B, C = d->C, XI = C + d->E, L = d->L;
    return las_gemm(d->prec, 0, 0, L * B, C, 4 * C, 1.f, bw->dgates, 4 * C, 0, p->w_ih[0], XI, 0, 0.f, bw->dxin, XI, 0, nullptr, 0, 1,
                    (void*)stream);
}
