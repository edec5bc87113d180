// Persistent attend-and-spell decoder loop (forward) for gfx950: ONE launch runs all L teacher-forced steps of reference
// src/asr.py:84-107 (location-aware Attention.forward :443-457, LSTMCell Speller :352-357) instead of four launches per
// step.  Same arithmetic and the same saved tensors (las_dec_state) as decoder.hip's per-step kernels, so the BPTT of
// decoder_bwd.hip consumes its output unchanged.
//
// The loop is a chain of L dependent steps whose arithmetic is a few microseconds; per-step launches cost ~40 us.  Here
// every workgroup lives for the whole loop (one per CU, all co-resident) and the step's four data hand-offs go through
// the L2 / memory fabric as tagged 16-byte granules (pk_common.h: {payload, tag = step + 1} written with one sc1 store,
// swept by the consumer with sc1 loads until every tag matches -- no flag, no drain, no atomic).  Two roles:
//
//   CELL workgroup (unit slice j of U hidden units, batch slice bs): keeps its 4U rows of [W_hh | W_ih(ctx part)] in LDS
//     for all steps.  Per step: q_t tile = tanh(W_phi h_{t-1}) (published to the attention workgroups), gates = W_hh
//     h_{t-1} + W_ih ctx_t + xe_t (xe = W_ih(emb part) emb_t + b_ih: ONE GEMM over all steps before the launch), cell
//     update, publish h_t, all-gather h_t from the other unit slices of its batch slice.
//   ATTENTION workgroup (utterance b, part c of NCH): keeps psi[b][its T'-chunk] in REGISTERS (fp32, each thread owns
//     fixed (frame, a) elements for the whole loop) and enc[b][all T'][its E-slice] in LDS -- nothing of psi / enc is
//     re-read from HBM or L2 after the prologue (SURVEY.md 8d prices the per-step re-read at 13.5 MB).  Per step:
//     location conv + tanh(W_lp f) from the previous attention (before q_t arrives, off the critical path), energies of
//     its T'-chunk once q_t is there, all-gather of the utterance's energies among its NCH parts, softmax (redundantly
//     per part), context for its E-slice (complete sums: no cross-workgroup reduction), published to the cell role.
//
// Hand-offs per step: ctx -> cell, h all-gather (cell), q -> attention, e all-gather (attention).  Spins are bounded; a
// timeout sets *status = LAS_E_TIMEOUT and every workgroup exits.
#include "pk_common.h"
#include "decoder_pk.h"
#include <stdlib.h>
#include <stdio.h>

extern "C" int las_gemm(int prec, int transA, int transB, int M, int N, int K, float alpha, const float* A, int64_t lda,
                        int64_t strideA, const float* B, int64_t ldb, int64_t strideB, float beta, float* C, int64_t ldc,
                        int64_t strideC, const float* bias, int act, int batch, void* stream);

namespace {

constexpr int LOC_C = 10, LOC_K = 100, LOC_W = 2 * LOC_K + 1;     // reference asr.py:395-398
constexpr int LWP = 208, NSEG = 4, SEGW = LWP / NSEG;              // taps padded with zeros, walked in 4 segments of 52
constexpr float ATT_SCALE = 2.0f;                                 // reference asr.py:410
constexpr int MAXB = 32;

struct PkSync {                         // zeroed before every launch
    unsigned abort_[CLW];
    unsigned utt[MAXB][CLW];            // per utterance: words 32..34 = the placement rendezvous of its attention parts
};

struct PkGeom {
    int U, NCT, NS, Bs, NB, NCELL;      // cell role: units per workgroup, unit slices, batch slices, rows per slice, 16-row tiles
    int xl;                             // attention parts of an utterance on block ids congruent mod 8 (one XCD, observed)
    int NQC;                            // q column tiles (16 wide)
    int NCH, TC, ES;                    // attention role: parts per utterance, frames per part, context columns per part
    int Cp, Ep, Cx, Ex;                 // k extents padded to the MFMA k-step; exchange row strides (padded to a vector)
    int MT, NTW;                        // attention role: 16-frame tiles of a T'-chunk, 16-wide a-tiles per wave
    int HG, CG, QG, TCG;                // granules: h per utterance, ctx per part, q per utterance, energies per part
    size_t lds;
};

struct PkArgs {
    int B, Tp, E, A, C, L;
    int loc;                            // 1: location-aware attention (asr.py:443-457); 0: dot attention (asr.py:426-432): no conv / u
                                        // phases, e = psi . q, nothing of f / s saved
    PkGeom g;
    const float* psi; const float* enc; const int32_t* lens;
    const float* xe;                    // [L][B][4C] W_ih[:, 0:C] emb_t + b_ih
    const float* w_ih; const float* w_hh; const float* b_hh; const float* w_phi;
    const float* conv_w; const float* w_lp; const float* w_e; const float* b_e;
    float* q; float* att; float* xin; float* hs; float* cs; float* gates; float* f; void* s;      // (s: fp32, or the 16-bit code of las_common.h in bf16 mode)
    u32x4 *hxg, *cxg, *qxg, *exg;       // granule rings [2][B][HG], [2][B][NCH*CG], [2][B][QG], [2][B][NCH*TCG]
    PkSync* sync; int* status;
    unsigned long long* dbg;            // [grid][12] cycle sums per phase (stamps build only)
};

// ---- cell role ---------------------------------------------------------------------------------------------------
// LDS: Wl [4U][ld] rows flat = gate*U + unit, columns [0,Cp) = W_hh row, [Cp,Cp+Ep) = W_ih row (context part);
//      Xl [NB*16][ld] batch rows: [0,Cp) = h_{t-1}, [Cp,Cp+Ep) = ctx_t;  Wq [16][Cp+VEC] my q tile's rows of W_phi;
//      Gl [8 waves][NB*16][17] accumulators.  Wave w: 16-row tile w % NTILE of Wl, k-steps ks = w / NTILE (mod KP).
template <int PREC, int NB>
__device__ __forceinline__ void pk_cell_role(const PkArgs& a, char* smem) {
    typedef typename CT<PREC>::T T;
    constexpr int VEC = CT<PREC>::VEC, KSTEP = CT<PREC>::KSTEP;
    const PkGeom& g = a.g;
    const int j = blockIdx.x / g.NS, bs = blockIdx.x - j * g.NS;
    const int U = g.U, j0 = j * U, b0 = bs * g.Bs, Bl = min(g.Bs, a.B - b0);
    const int B = a.B, C = a.C, E = a.E, A = a.A, XI = C + E, Cp = g.Cp, Ep = g.Ep, ld = Cp + Ep + VEC, ldq = Cp + VEC;
    const int NTILE = 4 * U / 16, KP = PNW / NTILE;
    T* Wl = (T*)smem;
    T* Xl = Wl + 4 * U * ld;
    T* Wq = Xl + NB * 16 * ld;
    float* Gl = (float*)(Wq + 16 * ldq);
    const bool has_q = j < g.NQC;

    for (int i = threadIdx.x; i < 4 * U * ld; i += PNT) {
        const int k = i % ld, row = i / ld, gi = row / U, n = row - gi * U;
        float v = 0.f;
        if (j0 + n < C) {
            const long wr = (long)gi * C + j0 + n;
            if (k < C) v = a.w_hh[wr * C + k];
            else if (k >= Cp && k - Cp < E) v = a.w_ih[wr * XI + C + (k - Cp)];
        }
        Wl[i] = to_ct<T>(v);
    }
    for (int i = threadIdx.x; i < NB * 16 * ld; i += PNT) Xl[i] = (T)0;
    for (int i = threadIdx.x; i < 16 * ldq; i += PNT) {
        const int k = i % ldq, r = i / ldq;
        Wq[i] = to_ct<T>((has_q && j * 16 + r < A && k < C) ? a.w_phi[(long)(j * 16 + r) * C + k] : 0.f);
    }
    __syncthreads();

    // my pointwise element: (batch row, unit)
    const int er = threadIdx.x / U, en = threadIdx.x - er * U, ej = j0 + en;
    const bool ev = er < Bl && ej < C && threadIdx.x < NB * 16 * U;
    float c_state = 0.f, bias[4];
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) bias[gi] = ev ? a.b_hh[gi * C + ej] : 0.f;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
    const int tile = wave % NTILE, kp = wave / NTILE;
    unsigned* abort_word = a.sync->abort_;
    constexpr int V12 = GrT<T>::V12, V8 = GrT<T>::V8;
    // my shares of the two sweeps (granule i of wave w, slot u of lane l: w * Q + 64 u + l): ring offsets, LDS destinations
    // and the number of payload words inside the row (the last part / unit group may reach beyond E / C) are fixed.
    constexpr int SWM = 4;
    const int nctx = Bl * g.NCH * g.CG, Qc = (nctx + PNW - 1) / PNW, swc = (Qc + 63) / 64;
    const int nh = Bl * g.HG, Qh = (nh + PNW - 1) / PNW, swh = (Qh + 63) / 64;
    int c_off[SWM], c_dst[SWM], c_nw[SWM], h_off[SWM], h_dst[SWM], h_nw[SWM];
#pragma unroll
    for (int u = 0; u < SWM; ++u) {
        const int idx = lane + 64 * u;
        {
            const int i = wave * Qc + idx;
            const bool ok = idx < Qc && i < nctx;
            const int r = ok ? i / (g.NCH * g.CG) : 0, rest = ok ? i - r * (g.NCH * g.CG) : 0, cc = rest / g.CG, k = rest - cc * g.CG;
            const int col = cc * g.ES + k * V12;
            c_off[u] = ok ? ((b0 + r) * g.NCH * g.CG + rest) * 16 : GR_OOB;
            c_dst[u] = r * ld + Cp + col;
            c_nw[u] = ok ? max(0, min(3, (min(E, (cc + 1) * g.ES) - col + V12 / 3 - 1) / (V12 / 3))) : 0;
        }
        {
            const int i = wave * Qh + idx;
            const bool ok = idx < Qh && i < nh;
            const int r = ok ? i / g.HG : 0, gq = ok ? i - r * g.HG : 0;
            h_off[u] = ok ? ((b0 + r) * g.HG + gq) * 16 : GR_OOB;
            h_dst[u] = r * ld + gq * V8;
            h_nw[u] = ok ? max(0, min(2, (C - gq * V8 + V8 / 2 - 1) / (V8 / 2))) : 0;
        }
    }
    const int slot_c = B * g.NCH * g.CG, slot_h = B * g.HG, slot_q = B * g.QG;
    auto fail = [&]() { *a.status = LAS_E_TIMEOUT; };
    PK_STAMP_DECL;

    for (int t = 0; t < a.L; ++t) {
        const unsigned tag = (unsigned)t + 1u;
        // ---- (1) q_t tile = tanh(W_phi[16 j .. ] h_{t-1}) for my batch slice; t = 0: h = 0
        if (has_q) {
            f32x4 qa[NB];
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) qa[bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (t > 0)
                for (int ks = wave; ks < Cp / KSTEP; ks += PNW) mma_rows<PREC, NB>(qa, Xl + ks * KSTEP, ld, Wq + ks * KSTEP, ldq, 1);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Gl[(wave * NB * 16 + bt * 16 + fq * 4 + r) * 17 + fr] = qa[bt][r];
            __syncthreads();
            if (threadIdx.x < NB * 16 * 16) {
                const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < PNW; ++w) v += Gl[(w * NB * 16 + row) * 17 + col];
                const bool okq = row < Bl && j * 16 + col < A;
                const float qv_ = okq ? fast_tanh(v) : 0.f;
                if (okq) __builtin_nontemporal_store(qv_, a.q + ((long)t * B + b0 + row) * A + j * 16 + col);      // saved for the backward pass
                unsigned w[3];
                pk_gr_gather8<float>(qv_, w);                    // q travels in f32: 2 values per granule
                __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)(a.qxg + (t & 1) * slot_q), 0, slot_q * 16, 0x00020000);
                pk_gr_store(rq, (okq && !(col & 1)) ? ((b0 + row) * g.QG + (j * 16 + col) / 2) * 16 : GR_OOB, w[0], w[1], 0u, tag);
            }
        }
        PK_STAMP(0);
        // ---- (2) recurrent half of the gates; the embedding half + b_ih comes precomputed
        float xe[4];
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) xe[gi] = ev ? a.xe[((long)t * B + b0 + er) * 4 * C + gi * C + ej] : 0.f;
        f32x4 acc[NB];
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) acc[bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (t > 0)
            for (int ks = kp; ks < Cp / KSTEP; ks += KP) mma_rows<PREC, NB>(acc, Xl + ks * KSTEP, ld, Wl + tile * 16 * ld + ks * KSTEP, ld, 1);
        PK_STAMP(1);
        // ---- (3) context of this step from the attention workgroups of my batch rows
        {
            __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.cxg + (t & 1) * slot_c), 0, slot_c * 16, 0x00020000);
            auto sweep = [&](auto swv) -> bool {
                constexpr int SW = decltype(swv)::value;
                int off[SW];
#pragma unroll
                for (int u = 0; u < SW; ++u) off[u] = c_off[u];
                return pk_gr_sweep<SW>(rc, off, tag, abort_word, [&](int u, const u32x4& gv) {
#pragma unroll
                    for (int e = 0; e < 3; ++e)
                        if (e < c_nw[u]) ((unsigned*)(Xl + c_dst[u]))[e] = gv[e];
                });
            };
            const bool ok = swc <= 1 ? sweep(std::integral_constant<int, 1>{}) : swc == 2 ? sweep(std::integral_constant<int, 2>{})
                          : swc == 3 ? sweep(std::integral_constant<int, 3>{}) : sweep(std::integral_constant<int, 4>{});
            if (!ok) { fail(); return; }
        }
        PK_STAMP(2);
        __syncthreads();
        for (int ks = kp; ks < Ep / KSTEP; ks += KP) mma_rows<PREC, NB>(acc, Xl + Cp + ks * KSTEP, ld, Wl + tile * 16 * ld + Cp + ks * KSTEP, ld, 1);
#pragma unroll
        for (int bt = 0; bt < NB; ++bt)
#pragma unroll
            for (int r = 0; r < 4; ++r) Gl[(wave * NB * 16 + bt * 16 + fq * 4 + r) * 17 + fr] = acc[bt][r];
        __syncthreads();
        PK_STAMP(3);
        // ---- (4) cell update (reference asr.py:353: nn.LSTMCell, gate order i,f,g,o); publish h_t first
        float pre[4];
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const int flat = gi * U + en, tl = flat >> 4, col = flat & 15;
            float v = xe[gi] + bias[gi];
            for (int k = 0; k < KP; ++k) v += Gl[((tl + NTILE * k) * NB * 16 + min(er, NB * 16 - 1)) * 17 + col];
            pre[gi] = v;
        }
        const float ig = fast_sig(pre[0]), fg = fast_sig(pre[1]), gg = fast_tanh(pre[2]), og = fast_sig(pre[3]);
        const float cn = fg * c_state + ig * gg;
        const float hn = ev ? og * fast_tanh(cn) : 0.f;
        c_state = cn;
        {
            unsigned w[3];
            pk_gr_gather8<T>(hn, w);                             // units en .. en + V8 - 1 of my row (U is a multiple of V8)
            __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)(a.hxg + (t & 1) * slot_h), 0, slot_h * 16, 0x00020000);
            pk_gr_store(rh, (ev && en % V8 == 0) ? ((b0 + er) * g.HG + ej / V8) * 16 : GR_OOB, w[0], w[1], 0u, tag);
        }
        PK_STAMP(4);
        if (ev) {
            const long ro = (long)t * B + b0 + er;
            __builtin_nontemporal_store(hn, &a.hs[(ro + B) * C + ej]);          // slot t + 1
            __builtin_nontemporal_store(cn, &a.cs[(ro + B) * C + ej]);
            float* go = a.gates + ro * 4 * C;
            __builtin_nontemporal_store(ig, &go[ej]);
            __builtin_nontemporal_store(fg, &go[C + ej]);
            __builtin_nontemporal_store(gg, &go[2 * C + ej]);
            __builtin_nontemporal_store(og, &go[3 * C + ej]);
        }
        PK_STAMP(5);
        // ---- (5) all-gather h_t of my batch slice for the next step
        if (t + 1 < a.L) {
            __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)(a.hxg + (t & 1) * slot_h), 0, slot_h * 16, 0x00020000);
            auto sweep = [&](auto swv) -> bool {
                constexpr int SW = decltype(swv)::value;
                int off[SW];
#pragma unroll
                for (int u = 0; u < SW; ++u) off[u] = h_off[u];
                return pk_gr_sweep<SW>(rh, off, tag, abort_word, [&](int u, const u32x4& gv) {
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        if (e < h_nw[u]) ((unsigned*)(Xl + h_dst[u]))[e] = gv[e];
                });
            };
            const bool ok = swh <= 1 ? sweep(std::integral_constant<int, 1>{}) : swh == 2 ? sweep(std::integral_constant<int, 2>{})
                          : swh == 3 ? sweep(std::integral_constant<int, 3>{}) : sweep(std::integral_constant<int, 4>{});
            if (!ok) { fail(); return; }
            PK_STAMP(6);
            __syncthreads();
            PK_STAMP(7);
        }
    }
    PK_STAMP_FLUSH(a.dbg);
}

// ---- attention role ----------------------------------------------------------------------------------------------
// MT = 16-frame tiles of my T'-chunk, NTW = 16-wide a-tiles per wave (wave w owns a-tiles w, w+8, ...).  u = tanh(F W_lp^T)
// is an MFMA product (F: frames x 10 channels from the location conv, zero padded to one k-step), and psi / u / s live in
// registers in that product's output layout: lane (fr = lane & 15, fq = lane >> 4) holds frames 16 mt + 4 fq + r, r = 0..3,
// of a = 16 (wave + 8 j) + fr.
template <int PREC, int MT, int NTW, bool LOC>      // LOC: location-aware attention (false: dot -- no conv / u / d f phases); a
// compile-time switch: as a run-time flag it cost the location-aware BPTT loop 0.7 ms at c3 (registers, 16 spilled)
__device__ __forceinline__ void pk_att_role(const PkArgs& a, char* smem) {
    typedef typename CT<PREC>::T T;
    constexpr int VEC = CT<PREC>::VEC, KSTEP = CT<PREC>::KSTEP, LDK = KSTEP + VEC;
    const PkGeom& g = a.g;
    // block -> (utterance b, part c); XCD-grouped as in the BPTT loop (decoder_pk_bwd.hip): ids congruent mod 8 share an XCD
    int b, c;
    {
        const int id = blockIdx.x - g.NCELL;
        if (g.xl) { const int m = id >> 3; b = (m / g.NCH) * 8 + (id & 7); c = m % g.NCH; }
        else { b = id / g.NCH; c = id - b * g.NCH; }
    }
    if (b >= a.B) return;
    const int B = a.B, Tp = a.Tp, E = a.E, A = a.A, C = a.C, XI = C + E;
    const int len = a.lens[b];
    const int TC = g.TC, r0 = c * TC, TCr = max(0, min(TC, Tp - r0));
    const int ES = g.ES, e0 = c * ES, ESr = max(0, min(ES, E - e0)), ESp = ES + 4;
    const int NQ4 = ES / 4, NTG = min(32, PNT / NQ4);
    const int TCq = (TC + 3) / 4, A4 = (A + 3) & ~3, Tp4 = (Tp + 3) & ~3;
    // every float array below starts 16-byte aligned (sizes are multiples of 4 floats): the location conv reads 16 bytes a time
    T* enc_l = (T*)smem;                                         // [Tp][ESp]
    float* part_l = (float*)(smem + (((size_t)Tp * ESp * sizeof(T) + 15) & ~(size_t)15));     // [NTG][ES]
    float* cw_l = part_l + NTG * ES;                             // [10][LWP] conv taps, zero padded to 208
    float* att_l = cw_l + LOC_C * LWP;                           // [<=3 | LOC_K | Tp | LOC_K + 16] previous attention, zero margins
    float* f4_l = att_l + Tp4 + 2 * LOC_K + 20;                  // [4 tap segments][10][TCq][4] partial conv sums
    float* we_l = f4_l + NSEG * LOC_C * TCq * 4;                 // [A4]
    float* q_l = we_l + A4;                                      // [A4]
    float* e_l = q_l + A4;                                       // [Tp4]
    float* ep_l = e_l + Tp4;                                     // [8 waves][MT*16] partial energies
    float* red = ep_l + PNW * MT * 16;                           // [64]
    T* Ft = (T*)(red + 64 + 4);                                     // [MT*16][LDK] location features of my frames (frames x channels)
    T* Wt = Ft + MT * 16 * LDK;                                  // [NTW*8*16][LDK] W_lp rows (a x channels), zero padded
    // my copy of the attention is shifted so that frame r0 sits on a 16-byte boundary: att0[LOC_K + t'] = att[t']
    float* att0 = att_l + ((4 - ((r0 + LOC_K) & 3)) & 3);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;

    // ---- prologue: everything of psi / enc this workgroup will ever need
    float pv[MT][NTW][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tp = min(r0 + mt * 16 + fq * 4 + r, Tp - 1), aa = min((wave + PNW * j) * 16 + fr, A - 1);
                pv[mt][j][r] = a.psi[((long)b * Tp + tp) * A + aa];
            }
    for (int i = threadIdx.x; i < Tp * ESp; i += PNT) {
        const int tp = i / ESp, col = i - tp * ESp;
        enc_l[i] = to_ct<T>((col < ESr && tp < len) ? a.enc[((long)b * Tp + tp) * E + e0 + col] : 0.f);
    }
    for (int i = threadIdx.x; i < NTW * PNW * 16 * LDK; i += PNT) {
        const int k = i % LDK, aa = i / LDK;
        Wt[i] = to_ct<T>((LOC && k < LOC_C && aa < A) ? a.w_lp[(long)aa * LOC_C + k] : 0.f);
    }
    for (int i = threadIdx.x; i < MT * 16 * LDK; i += PNT) Ft[i] = (T)0;
    for (int i = threadIdx.x; i < LOC_C * LWP; i += PNT) cw_l[i] = 0.f;
    __syncthreads();
    if (LOC) {
        fill_batched<4>(a.conv_w, LOC_C * LOC_W, [&](int i, float v) { const int cc = i / LOC_W; cw_l[cc * LWP + (i - cc * LOC_W)] = v; });
        fill_batched<2>(a.w_e, A, [&](int i, float v) { we_l[i] = v; });
    } else {
        for (int i = threadIdx.x; i < A; i += PNT) we_l[i] = 0.f;
    }
    for (int i = threadIdx.x; i < Tp4 + 2 * LOC_K + 20; i += PNT) att_l[i] = 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < len; i += PNT) att0[LOC_K + i] = 1.f / (float)len;      // reference asr.py:444-449
    __syncthreads();
    float wev[NTW];                                              // my a-columns' energy weights (0 beyond A)
#pragma unroll
    for (int j = 0; j < NTW; ++j) { const int aa = (wave + PNW * j) * 16 + fr; wev[j] = aa < A ? we_l[aa] : 0.f; }
    const float be = LOC ? a.b_e[0] : 0.f;
    unsigned* abort_word = a.sync->abort_;
    // the energies all-gather stays inside my utterance: plain granule stores when its parts share an XCD's L2 (checked once)
    const int loc_ = pk_utt_local(&a.sync->utt[b][0], g.NCH, g.xl != 0, abort_word, (int*)(red + 64));
    if (loc_ < 0) { *a.status = LAS_E_TIMEOUT; return; }
    const bool local = loc_ > 0;
    constexpr int V12 = GrT<T>::V12;
    // my shares of the two sweeps (q of my utterance: QG granules of 2 f32; its energies: NCH * TCG granules of 2 f32)
    constexpr int SWA = 2;
    const int ne = g.NCH * g.TCG;
    int q_off[SWA], q_dst[SWA], e_off[SWA], e_dst[SWA];
    bool e_two[SWA];
#pragma unroll
    for (int u = 0; u < SWA; ++u) {
        const int i = threadIdx.x + u * PNT;
        q_off[u] = i < g.QG ? (b * g.QG + i) * 16 : GR_OOB;
        q_dst[u] = 2 * i;
        const int cc = i / g.TCG, k = i - cc * g.TCG;
        e_off[u] = i < ne ? (b * ne + i) * 16 : GR_OOB;
        e_dst[u] = cc * TC + 2 * k;
        e_two[u] = 2 * k + 1 < TC;
    }
    const int slot_c = B * g.NCH * g.CG, slot_q = B * g.QG, slot_e = B * ne;
    auto fail = [&]() { *a.status = LAS_E_TIMEOUT; };
    // this thread's (first) conv work item and reduction element: fixed for the whole loop
    const int cv_sg = threadIdx.x / (LOC_C * TCq), cv_cc = (threadIdx.x - cv_sg * (LOC_C * TCq)) / TCq,
              cv_qd = threadIdx.x - cv_sg * (LOC_C * TCq) - cv_cc * TCq;
    const int rd_cc = threadIdx.x / TC, rd_tt = threadIdx.x - rd_cc * TC;
    PK_STAMP_DECL;

    for (int t = 0; t < a.L; ++t) {
        // ---- (A) location features of my frames from the previous attention, u = tanh(W_lp f): needs no q_t.
        // Work item = (channel, 4 consecutive frames, tap segment): 16 FMAs per three 16-byte LDS reads (the scalar form
        // -- two 4-byte reads per FMA -- was bound by LDS instruction issue: 9 400 cycles a step, cycle stamps)
        float uv[MT][NTW][4];
        if (LOC) {
        for (int i = threadIdx.x; i < NSEG * LOC_C * TCq; i += PNT) {
            int sg = cv_sg, cc = cv_cc, qd = cv_qd;                          // (the first item's split is loop-invariant)
            if (i >= PNT) { sg = i / (LOC_C * TCq); const int rem = i - sg * (LOC_C * TCq); cc = rem / TCq; qd = rem - cc * TCq; }
            if (4 * qd >= TCr) { *(float4*)(f4_l + (size_t)i * 4) = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
            const float* w = cw_l + cc * LWP + sg * SEGW;
            const float* p = att0 + r0 + 4 * qd + sg * SEGW;                 // p[k] = prev[r0 + 4 qd + (52 sg + k) - LOC_K]
            float4 lo = *(const float4*)p;
            float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll 4
            for (int k = 0; k < SEGW; k += 4) {
                const float4 wv = *(const float4*)(w + k), hi = *(const float4*)(p + k + 4);
                // (explicit fma: the library is built with -ffp-contract=off, a*b+c would be two instructions)
                o0 = fmaf(wv.x, lo.x, fmaf(wv.y, lo.y, fmaf(wv.z, lo.z, fmaf(wv.w, lo.w, o0))));
                o1 = fmaf(wv.x, lo.y, fmaf(wv.y, lo.z, fmaf(wv.z, lo.w, fmaf(wv.w, hi.x, o1))));
                o2 = fmaf(wv.x, lo.z, fmaf(wv.y, lo.w, fmaf(wv.z, hi.x, fmaf(wv.w, hi.y, o2))));
                o3 = fmaf(wv.x, lo.w, fmaf(wv.y, hi.x, fmaf(wv.z, hi.y, fmaf(wv.w, hi.z, o3))));
                lo = hi;
            }
            *(float4*)(f4_l + (size_t)i * 4) = make_float4(o0, o1, o2, o3);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < LOC_C * TC; i += PNT) {
            int cc = rd_cc, tt = rd_tt;
            if (i >= PNT) { cc = i / TC; tt = i - cc * TC; }
            float v = 0.f;
#pragma unroll
            for (int sg = 0; sg < NSEG; ++sg) v += f4_l[(((size_t)sg * LOC_C + cc) * TCq + (tt >> 2)) * 4 + (tt & 3)];
            Ft[tt * LDK + cc] = to_ct<T>(v);
            if (tt < TCr) __builtin_nontemporal_store(v, &a.f[(((long)t * B + b) * LOC_C + cc) * Tp + r0 + tt]);
        }
        __syncthreads();
        PK_STAMP(0);
        // u = tanh(F W_lp^T): one MFMA k-step per (frame tile, a tile); kept in registers until q_t arrives
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
                mma_rows<PREC, 1>(acc, Ft + mt * 16 * LDK, LDK, Wt + (wave + PNW * j) * 16 * LDK, LDK, 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    uv[mt][j][r] = fast_tanh(acc[0][r]);
                    asm volatile("" : "+v"(uv[mt][j][r]));        // done HERE, before the wait for q_t (not sunk behind it)
                }
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) uv[mt][j][r] = 0.f;
        }
        PK_STAMP(1);
        // ---- (B) q_t from the cell workgroups of my batch slice
        const unsigned tag = (unsigned)t + 1u;
        {
            __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)(a.qxg + (t & 1) * slot_q), 0, slot_q * 16, 0x00020000);
            auto sweep = [&](auto swv) -> bool {
                constexpr int SW = decltype(swv)::value;
                int off[SW];
#pragma unroll
                for (int u = 0; u < SW; ++u) off[u] = q_off[u];
                return pk_gr_sweep<SW>(rq, off, tag, abort_word, [&](int u, const u32x4& gv) {
                    q_l[q_dst[u]] = __uint_as_float(gv[0]); q_l[q_dst[u] + 1] = __uint_as_float(gv[1]);
                });
            };
            const bool ok = g.QG <= PNT ? sweep(std::integral_constant<int, 1>{}) : sweep(std::integral_constant<int, 2>{});
            if (!ok) { fail(); return; }
        }
        __syncthreads();
        PK_STAMP(2);
        float qv[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) { const int aa = (wave + PNW * j) * 16 + fr; qv[j] = aa < A ? q_l[aa] : 0.f; }
        // ---- (C) energies of my frames: e = w_e . tanh(psi + q + u) + b_e   (reference asr.py:453); s overwrites u
        float er_[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    if (LOC) {
                        const float sv = fast_tanh(pv[mt][j][r] + qv[j] + uv[mt][j][r]);
                        uv[mt][j][r] = sv;
                        acc = fmaf(wev[j], sv, acc);
                    } else {
                        acc = fmaf(pv[mt][j][r], qv[j], acc);      // dot attention: e = psi . q (asr.py:428); columns >= A: q = 0
                    }
                }
                // sum over the 16 a's of my lane row (DPP row_shr 1, 2, 4, 8: lane 15 of the row ends with the total)
                acc += las_dpp<0x111, 0xf>(0.f, acc);
                acc += las_dpp<0x112, 0xf>(0.f, acc);
                acc += las_dpp<0x114, 0xf>(0.f, acc);
                acc += las_dpp<0x118, 0xf>(0.f, acc);
                er_[mt][r] = acc;
            }
        if (fr == 15) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) ep_l[wave * MT * 16 + mt * 16 + fq * 4 + r] = er_[mt][r];
        }
        __syncthreads();
        if (threadIdx.x < MT * 16) {                             // (whole waves: the granule gather runs on DPP)
            const bool act = (int)threadIdx.x < TCr;
            float v = be;
#pragma unroll
            for (int w = 0; w < PNW; ++w) v += ep_l[w * MT * 16 + threadIdx.x];
            v = (act && r0 + (int)threadIdx.x < len) ? v : 0.f;
            unsigned w2[3];
            pk_gr_gather8<float>(v, w2);                         // energies travel in f32: 2 frames per granule
            __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void*)(a.exg + (t & 1) * slot_e), 0, slot_e * 16, 0x00020000);
            // (every part publishes all its TCG granules, zeros beyond the utterance: the readers wait for each of them)
            pk_gr_store_x(re, (!(threadIdx.x & 1) && (int)threadIdx.x / 2 < g.TCG) ? ((b * g.NCH + c) * g.TCG + (int)threadIdx.x / 2) * 16 : GR_OOB, w2[0], w2[1], 0u, tag, local);
        }
        PK_STAMP(3);
        PK_STAMP(4);
        // ---- (D) all energies of my utterance; masked softmax(2 e) (reference asr.py:454-455), redundantly per part
        {
            __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void*)(a.exg + (t & 1) * slot_e), 0, slot_e * 16, 0x00020000);
            auto sweep = [&](auto swv) -> bool {
                constexpr int SW = decltype(swv)::value;
                int off[SW];
#pragma unroll
                for (int u = 0; u < SW; ++u) off[u] = e_off[u];
                return pk_gr_sweep<SW>(re, off, tag, abort_word, [&](int u, const u32x4& gv) {
                    if (e_dst[u] < Tp4) e_l[e_dst[u]] = ATT_SCALE * __uint_as_float(gv[0]);
                    if (e_two[u] && e_dst[u] + 1 < Tp4) e_l[e_dst[u] + 1] = ATT_SCALE * __uint_as_float(gv[1]);   // (odd TC: the last granule's second frame belongs to the next part)
                });
            };
            const bool ok = ne <= PNT ? sweep(std::integral_constant<int, 1>{}) : sweep(std::integral_constant<int, 2>{});
            if (!ok) { fail(); return; }
        }
        __syncthreads();
        PK_STAMP(5);
        // every WAVE takes the maximum and the sum over the whole utterance for itself (<= 8 elements per lane, DPP
        // reductions): no workgroup reduction, no barrier before the one the context needs anyway (two block reductions
        // and libm expf were 3 600 cycles of the step)
        // (T' <= 512: the wave's <= 8 elements per lane are loaded ONCE, all requests before the first use; the two loops over
        // LDS that stood here -- each iteration waiting out the LDS latency behind the previous one's max / exp -- were most of
        // this phase's 3 700 cycles)
        float m = -INFINITY, sum = 0.f;
        if (Tp <= 512) {
            float ev[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int i = lane + 64 * k; const float v = e_l[min(i, Tp4 - 1)]; ev[k] = i < len ? v : -INFINITY; }
#pragma unroll
            for (int k = 0; k < 8; ++k) m = fmaxf(m, ev[k]);
            m = wave_max(m);
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += __expf(ev[k] - m);            // (exp(-inf) = 0 beyond the utterance)
        } else {
            for (int i = lane; i < len; i += 64) m = fmaxf(m, e_l[i]);
            m = wave_max(m);
            for (int i = lane; i < len; i += 64) sum += __expf(e_l[i] - m);
        }
        sum = wave_sum(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
        for (int i = threadIdx.x; i < Tp; i += PNT) att0[LOC_K + i] = i < len ? __expf(e_l[i] - m) * inv : 0.f;
        __syncthreads();
        PK_STAMP(6);
        // ---- (E) context of my E-slice over the RAW encoder features (reference asr.py:457); complete sums
        {
            const int cq = threadIdx.x % NQ4, tg = threadIdx.x / NQ4;
            if (tg < NTG) {
                // four frames per iteration, their eight LDS reads requested before the first FMA (one frame per iteration paid
                // the LDS latency ~14 times in a row: most of this phase's 3 600 cycles)
                float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
                int tp = tg;
                for (; tp + 3 * NTG < len; tp += 4 * NTG) {
                    float w[4];
                    float4 v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { w[j] = att0[LOC_K + tp + j * NTG]; v[j] = ld4(enc_l + (size_t)(tp + j * NTG) * ESp + cq * 4); }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        c0 = fmaf(w[j], v[j].x, c0); c1 = fmaf(w[j], v[j].y, c1); c2 = fmaf(w[j], v[j].z, c2); c3 = fmaf(w[j], v[j].w, c3);
                    }
                }
                for (; tp < len; tp += NTG) {
                    const float w = att0[LOC_K + tp];
                    const float4 v = ld4(enc_l + (size_t)tp * ESp + cq * 4);
                    c0 = fmaf(w, v.x, c0); c1 = fmaf(w, v.y, c1); c2 = fmaf(w, v.z, c2); c3 = fmaf(w, v.w, c3);
                }
                *(float4*)(part_l + tg * ES + cq * 4) = make_float4(c0, c1, c2, c3);
            }
        }
        __syncthreads();
        {
            // 12 consecutive columns in lanes 0..11 of every 16-lane row: a granule's 3 f32 / 6 bf16 come over DPP row shifts
            const int kk = threadIdx.x >> 4, l = threadIdx.x & 15, col = 12 * kk + l;
            const bool cv = l < 12 && col < ES;
            float v = 0.f;
            if (cv) {                                            // NTG <= 32 partials: every read requested before the first add
                float pp[32];
#pragma unroll
                for (int tg = 0; tg < 32; ++tg) pp[tg] = part_l[min(tg, NTG - 1) * ES + col];
#pragma unroll
                for (int tg = 0; tg < 32; ++tg) v += tg < NTG ? pp[tg] : 0.f;
            }
            v = (cv && col < ESr) ? v : 0.f;
            unsigned w3[3];
            pk_gr_gather12<T>(v, w3);
            __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.cxg + (t & 1) * slot_c), 0, slot_c * 16, 0x00020000);
            const int gk = kk * (12 / V12) + l / V12;              // granule gk = columns [V12 gk, V12 gk + V12) of my slice
            pk_gr_store(rc, (l < 12 && l % V12 == 0 && gk < g.CG) ? ((b * g.NCH + c) * g.CG + gk) * 16 : GR_OOB, w3[0], w3[1], w3[2], tag);
            PK_STAMP(7);
            // saved for the backward pass (after the hand-off): the context, and the attention map from part 0
            if (cv && col < ESr) a.xin[((long)t * B + b) * XI + C + e0 + col] = v;
            if (c == 0)
                for (int i = threadIdx.x; i < Tp; i += PNT) a.att[((long)(t + 1) * B + b) * Tp + i] = att0[LOC_K + i];
        }
        // s = tanh(psi + q + u) of my frames, saved for the backward pass: stored only now, after both hand-offs of the
        // step, so that neither waits for these stores (write-only stream: non-temporal)
        if (LOC) {
            if constexpr (PREC == LAS_PREC_BF16) {          // (pk_geom: A is even then)
                // the 16-bit code, two columns per store: an even lane writes the pair (fr, fr + 1) of rows 0 and 2, its odd neighbour
                // the same pair of rows 1 and 3 (codes swapped over DPP) -- 18 four-byte stores per lane instead of 36 two-byte ones
                const bool odd = fr & 1;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        const int aa = (wave + PNW * j) * 16 + (fr & ~1);
                        unsigned c[4], n[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            c[r] = las_s16_enc(uv[mt][j][r]);
                            n[r] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)c[r], 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
                        }
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const int tt = mt * 16 + fq * 4 + 2 * k + (odd ? 1 : 0), tp = r0 + tt;
                            const unsigned w = odd ? (n[2 * k + 1] | (c[2 * k + 1] << 16)) : (c[2 * k] | (n[2 * k] << 16));      // (static indices: registers)
                            if (tt < TCr && tp < len && aa < A)
                                __builtin_nontemporal_store(w, (unsigned*)((bf16_t*)a.s + (((long)t * B + b) * Tp + tp) * A + aa));
                        }
                    }
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int tt = mt * 16 + fq * 4 + r, tp = r0 + tt;
                        if (tt < TCr && tp < len) {
                            const long so = (((long)t * B + b) * Tp + tp) * A;
#pragma unroll
                            for (int j = 0; j < NTW; ++j) {
                                const int aa = (wave + PNW * j) * 16 + fr;
                                if (aa < A) {
                                    if constexpr (PREC == LAS_PREC_BF16) __builtin_nontemporal_store(las_s16_enc(uv[mt][j][r]), (bf16_t*)a.s + so + aa);
                                    else __builtin_nontemporal_store(uv[mt][j][r], (float*)a.s + so + aa);
                                }
                            }
                        }
                    }
            }
        }
        PK_STAMP(8);
    }
    PK_STAMP_FLUSH(a.dbg);
}

template <int PREC, int NB, int MT, int NTW, bool LOC>
__global__ __launch_bounds__(PNT) void dec_pk_fwd_kernel(PkArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < a.g.NCELL) pk_cell_role<PREC, NB>(a, smem);
    else pk_att_role<PREC, MT, NTW, LOC>(a, smem);
}

// ---- host side ---------------------------------------------------------------------------------------------------
size_t cell_lds(int prec, const PkGeom& g, int C, int E) {
    const int sz = prec == LAS_PREC_BF16 ? 2 : 4, vec = prec == LAS_PREC_BF16 ? 8 : 4;
    const size_t ld = g.Cp + g.Ep + vec;
    return (size_t)(4 * g.U + g.NB * 16) * ld * sz + (size_t)16 * (g.Cp + vec) * sz + sizeof(float) * PNW * g.NB * 16 * 17 + 64;
}
size_t att_lds(int prec, const PkGeom& g, int Tp, int A) {
    const int sz = prec == LAS_PREC_BF16 ? 2 : 4, ldk = prec == LAS_PREC_BF16 ? 40 : 20;
    int ntg = PNT / (g.ES / 4);
    if (ntg > 32) ntg = 32;
    const size_t A4 = (A + 3) & ~3, Tp4 = (Tp + 3) & ~3, TCq = (g.TC + 3) / 4;
    return (size_t)Tp * (g.ES + 4) * sz + 16 + (size_t)(g.MT * 16 + g.NTW * PNW * 16) * ldk * sz +
           sizeof(float) * ((size_t)ntg * g.ES + LOC_C * LWP + (Tp4 + 2 * LOC_K + 20) + NSEG * LOC_C * TCq * 4 + 2 * A4 + Tp4 +
                            PNW * g.MT * 16 + 64 + 4) + 64;
}

bool pk_geom(const las_dec_dims* d, PkGeom& best) {
    if (!d || d->NL != 1 || d->dropout != 0.f || d->L < 1) return false;      // (dot and location-aware attention)
    if (d->B < 1 || d->B > MAXB || d->A > 512 || d->A < 1 || d->Tp < 1 || (d->C & 1)) return false;
    if (d->prec != LAS_PREC_BF16 && d->prec != LAS_PREC_F32) return false;
    if (d->prec == LAS_PREC_BF16 && d->loc && (d->A & 1)) return false;       // (the saved s is written as column pairs)
    if (las_fallback("LAS_DEC_NO_PK")) return false;
    const int ks = d->prec == LAS_PREC_BF16 ? 32 : 16, vec = d->prec == LAS_PREC_BF16 ? 8 : 4;
    int want_ns = 0, want_u = 0;
    if (const char* e = LAS_AB_KNOB("LAS_DEC_PK_CFG")) sscanf(e, "%d,%d", &want_ns, &want_u);     // (A/B measurements)
    static const int cand[6][2] = {{2, 8}, {2, 16}, {1, 8}, {1, 16}, {2, 4}, {1, 4}};
    for (int ci = 0; ci < 6; ++ci) {
        PkGeom g{};
        g.NS = cand[ci][0]; g.U = cand[ci][1];
        if (want_ns && (g.NS != want_ns || g.U != want_u)) continue;
        if (g.NS > 1 && d->B < 8) continue;
        g.Bs = (d->B + g.NS - 1) / g.NS;
        g.NS = (d->B + g.Bs - 1) / g.Bs;
        g.NB = g.Bs <= 16 ? 1 : 2;
        if (g.Bs > 32) continue;
        g.NCT = (d->C + g.U - 1) / g.U;
        g.NCELL = g.NCT * g.NS;
        g.NQC = (d->A + 15) / 16;
        if (g.NQC > g.NCT) continue;
        g.Cp = (d->C + ks - 1) / ks * ks; g.Ep = (d->E + ks - 1) / ks * ks;
        g.Cx = (d->C + vec - 1) / vec * vec; g.Ex = (d->E + vec - 1) / vec * vec;
        if (cell_lds(d->prec, g, d->C, d->E) > PK_LDS_CAP) continue;
        g.NCH = (las_cu_count() - g.NCELL) / d->B;
        if (g.NCH > 16) g.NCH = 16;
        if (g.NCH < 1) continue;
        g.TC = (d->Tp + g.NCH - 1) / g.NCH;
        if (g.TC > 64) continue;
        g.MT = g.TC <= 32 ? 2 : g.TC <= 48 ? 3 : 4;
        g.NTW = d->A <= 128 ? 1 : d->A <= 384 ? 3 : 4;
        if (g.NTW == 4 && g.MT == 4) continue;          // (register budget of the attention role: 2 x 4 MT NTW values per lane)
        g.ES = ((d->E + g.NCH - 1) / g.NCH + 3) / 4 * 4;
        if (g.ES / 4 > PNT || g.ES > PNT) continue;
        if (att_lds(d->prec, g, d->Tp, d->A) > PK_LDS_CAP) continue;
        // granule counts (12-byte payload: 3 f32 / 6 bf16; 8-byte payload: 2 f32 / 4 bf16) and the sweeps' limits
        const int v12 = d->prec == LAS_PREC_BF16 ? 6 : 3, v8 = d->prec == LAS_PREC_BF16 ? 4 : 2;
        if ((d->E & 1) || g.U % v8 != 0) continue;
        g.HG = (d->C + v8 - 1) / v8;
        g.CG = (g.ES + v12 - 1) / v12;
        g.QG = (d->A + 1) / 2;
        g.TCG = (g.TC + 1) / 2;
        if (g.Bs * g.HG > 4 * PNT || g.Bs * g.NCH * g.CG > 4 * PNT || g.QG > 2 * PNT || g.NCH * g.TCG > 2 * PNT) continue;
        if ((g.ES + 11) / 12 * 16 > PNT) continue;
        g.xl = (!las_fallback("LAS_DEC_NO_XL") && g.NCELL % 8 == 0 && g.NCELL + 8 * ((d->B + 7) / 8) * g.NCH <= las_cu_count()) ? 1 : 0;
        g.lds = cell_lds(d->prec, g, d->C, d->E);
        const size_t al = att_lds(d->prec, g, d->Tp, d->A);
        if (al > g.lds) g.lds = al;
        if (g.lds < PK_MIN_LDS) g.lds = PK_MIN_LDS;
        best = g;
        return true;
    }
    return false;
}

struct WsLayout { size_t sync, hxg, cxg, qxg, exg, dbg, xe, total; };
WsLayout ws_layout(const las_dec_dims* d, const PkGeom& g) {
    WsLayout w;
    size_t o = 0;
    w.sync = o; o += las_align(sizeof(PkSync));
    w.dbg = o; o += las_align(sizeof(unsigned long long) * 256 * 20);      // (read by tools/pk_stamps.py in the stamps build)
    w.hxg = o; o += las_align((size_t)16 * 2 * d->B * g.HG);
    w.cxg = o; o += las_align((size_t)16 * 2 * d->B * g.NCH * g.CG);
    w.qxg = o; o += las_align((size_t)16 * 2 * d->B * g.QG);
    w.exg = o; o += las_align((size_t)16 * 2 * d->B * g.NCH * g.TCG);
    w.xe = o; o += las_align(sizeof(float) * (size_t)d->L * d->B * 4 * d->C);
    w.total = o;
    return w;
}

}  // namespace

size_t las_dec_pk_fwd_ws_bytes(const las_dec_dims* d) {
    PkGeom g;
    if (!pk_geom(d, g)) return 0;
    return ws_layout(d, g).total;
}

int las_dec_pk_fwd(const las_dec_dims* d, const las_dec_params* p, const float* enc, const float* psi,
                   const int32_t* enc_len, las_dec_state* st, hipStream_t stream) {
    PkGeom g;
    if (!pk_geom(d, g)) return LAS_E_UNSUPPORTED;
    LAS_CHECK_ARG(st->pk_ws && st->pk_status);
    if (d->loc) LAS_CHECK_ARG(st->f && st->s && p->conv_w && p->w_lp && p->w_e && p->b_e);
    const WsLayout w = ws_layout(d, g);
    char* ws = (char*)st->pk_ws;
    const int B = d->B, C = d->C, E = d->E, XI = C + E, L = d->L;
    // abort word and the granule rings (no tag of an earlier launch may match)
    LAS_HIP(hipMemsetAsync(ws, 0, w.xe, stream));
    // xe[t][b][:] = W_ih[:, 0:C] emb(tok[t][b]) + b_ih for every step: one MFMA GEMM instead of L skinny k-segments
    float* xe = (float*)(ws + w.xe);
    int rc = las_gemm(d->prec, 0, 1, L * B, 4 * C, C, 1.f, st->xin, XI, 0, p->w_ih[0], XI, 0, 0.f, xe, 4 * C, 0, p->b_ih[0], 0, 1,
                      (void*)stream);
    if (rc) return rc;
    PkArgs a{};
    a.B = B; a.Tp = d->Tp; a.E = E; a.A = d->A; a.C = C; a.L = L; a.g = g; a.loc = d->loc ? 1 : 0;
    a.psi = psi; a.enc = enc; a.lens = enc_len; a.xe = xe;
    a.w_ih = p->w_ih[0]; a.w_hh = p->w_hh[0]; a.b_hh = p->b_hh[0]; a.w_phi = p->w_phi;
    a.conv_w = p->conv_w; a.w_lp = p->w_lp; a.w_e = p->w_e; a.b_e = p->b_e;
    a.q = st->q; a.att = st->att; a.xin = st->xin; a.hs = st->hs; a.cs = st->cs; a.gates = st->gates; a.f = st->f; a.s = st->s;
    a.hxg = (u32x4*)(ws + w.hxg); a.cxg = (u32x4*)(ws + w.cxg); a.qxg = (u32x4*)(ws + w.qxg); a.exg = (u32x4*)(ws + w.exg);
    a.sync = (PkSync*)(ws + w.sync); a.status = st->pk_status;
    a.dbg = (unsigned long long*)(ws + w.dbg);
    const int grid = g.NCELL + (g.xl ? 8 * ((B + 7) / 8) * g.NCH : B * g.NCH);
#define LAS_PK_GO(P_, N_, M_, W_)                                                                                  \
    {                                                                                                             \
        auto k = a.loc ? dec_pk_fwd_kernel<P_, N_, M_, W_, true> : dec_pk_fwd_kernel<P_, N_, M_, W_, false>;                                                               \
        LAS_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds));     \
        hipLaunchKernelGGL(k, dim3(grid), dim3(PNT), g.lds, stream, a);                                           \
        LAS_LAUNCH_OK();                                                                                          \
        return LAS_OK;                                                                                            \
    }
#define LAS_PK_W(P_, N_, M_)                                                                                       \
    {                                                                                                             \
        if (g.NTW == 1) LAS_PK_GO(P_, N_, M_, 1) else if (g.NTW == 3) LAS_PK_GO(P_, N_, M_, 3) else LAS_PK_GO(P_, N_, M_, 4) \
    }
#define LAS_PK_AI(P_, N_)                                                                                          \
    {                                                                                                             \
        if (g.MT == 2) LAS_PK_W(P_, N_, 2) else if (g.MT == 3) LAS_PK_W(P_, N_, 3) else LAS_PK_W(P_, N_, 4)        \
    }
    if (d->prec == LAS_PREC_BF16) { if (g.NB == 1) LAS_PK_AI(LAS_PREC_BF16, 1) else LAS_PK_AI(LAS_PREC_BF16, 2) }
    else { if (g.NB == 1) LAS_PK_AI(LAS_PREC_F32, 1) else LAS_PK_AI(LAS_PREC_F32, 2) }
#undef LAS_PK_W
#undef LAS_PK_AI
#undef LAS_PK_GO
    return LAS_E_BADARG;
}
