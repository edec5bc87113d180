// Skinny MFMA product for the per-step decoder matmuls (M = batch <= 128), with fused epilogues.
//
//   pre[b][n] = sum_s sum_k X_s[b][k] * W_s[row(n)][k]  (+ bias0[row] + bias1[row])
//
// Replaces, one decode step at a time: nn.LSTMCell (reference src/asr.py:329-331, called :353-355:
// gates = W_ih x + b_ih + W_hh h + b_hh followed by the cell pointwise), tanh(phi(h)) (:383,422), the
// per-step char_trans when sampling (:92), and in backward the dgates*W products (with transposed weight
// copies so every operand row is k-contiguous).
//
// One workgroup = 16 output columns x all batch rows; K is split over the 4 waves (partials meet in LDS).
// Operands are staged fp32 -> operand type into LDS as [row][k]; up to 3 (X,W) segments are concatenated
// along k (e.g. [emb|ctx] with W_ih and h with W_hh).  Epilogues: store (+accumulate), tanh, LSTM cell.
#include "las_mma.h"

namespace {

constexpr int NT = 256;

struct Seg {
    const float* x; long ldx;
    const float* w; long ldw;
    int K;
};
struct SkinnyArgs {
    Seg seg[3];
    int ns, B, N;
    const float* bias0; const float* bias1;
    int mode;                 // 0 store, 1 tanh, 2 lstm cell, 3 product = dh of the PREVIOUS decode step -> its cell backward
    float* out; long ldo; int accumulate;
    // cell mode: N = 4*C gate rows (i,f,g,o blocks of C); tile column c -> gate c>>2, unit blk*4 + (c&3)
    int C;
    const float* c_prev; float* h_out; float* c_out; float* gates_out;
    // mode 3 (N = C): out[b][n] is d loss / d h of the earlier step through the recurrence; with the upstream gradient
    // dh_ext it goes straight through that step's cell pointwise backward (decoder_bwd.hip cell_pw_bwd) for element (b,n)
    const float* pw_dh_ext; long pw_ld_ext; float* pw_dc_carry; const float* pw_gates; const float* pw_c_t;
    const float* pw_c_prev; float* pw_dgates;
    // optional: the weight operand pre-packed in MFMA fragment order (las_skinny_pack_weights): for block b and k-step g
    // (over the concatenated segments) 64 lanes x 8 bf16 contiguous, i.e. ONE fully coalesced 1 KB read per wave and
    // k-step instead of two 16-byte pieces from each of 16 rows 4*K bytes apart; zero where the row or k is out of range
    const bf16_t* wpk;
};

// Stage `nrows` rows x n columns (fp32 source, row r at base + rowidx(r)*ldsrc) into tile[r][dcol..dcol+n) with all
// 256 threads; 8 independent 16-byte loads are in flight per thread before the first LDS write (the serial
// load->write->load chain of a row-at-a-time copy costs a full memory latency per row).
template <typename T, typename RowFn>
__device__ __forceinline__ void stage_block(T* __restrict__ tile, int ld, int dcol, const float* __restrict__ base,
                                            long ldsrc, int nrows, int n, RowFn rowidx) {
    constexpr int U = 8;
    const bool vec = ((n & 3) == 0) && ((ldsrc & 3) == 0) && ((((uintptr_t)base) & 15) == 0);
    if (vec) {
        const int nv = n >> 2, total = nrows * nv;
        for (int i0 = threadIdx.x; i0 < total; i0 += NT * U) {
            float4 v[U];
            int row[U], col[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * NT;
                row[u] = -1;
                if (i < total) {
                    const int r = i / nv;
                    col[u] = (i - r * nv) << 2;
                    const long sr = rowidx(r);
                    if (sr >= 0) { row[u] = r; v[u] = *(const float4*)(base + sr * ldsrc + col[u]); }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (row[u] >= 0) store4_ct(tile + row[u] * ld + dcol + col[u], v[u].x, v[u].y, v[u].z, v[u].w);
        }
    } else {
        const int total = nrows * n;
        for (int i = threadIdx.x; i < total; i += NT) {
            const int r = i / n, c = i - r * n;
            const long sr = rowidx(r);
            if (sr >= 0) tile[r * ld + dcol + c] = to_ct<T>(base[sr * ldsrc + c]);
        }
    }
}

// Epilogue operands of a thread's FIRST output element (bias, old output when accumulating, previous cell state):
// requested at kernel entry so that their memory round trip overlaps the weight stream instead of following the
// reduction barrier.
struct SkinnyPre { float bias[4]; float old; float pw[4]; };
template <int NWAVES, int MODE = -1>
__device__ __forceinline__ SkinnyPre skinny_prefetch(const SkinnyArgs& a, int blk) {
    const int mode = MODE >= 0 ? (MODE == 0 ? (a.mode & 1) : MODE) : a.mode;      // MODE 0 covers store (0) and tanh (1)
    SkinnyPre p;
    p.bias[0] = p.bias[1] = p.bias[2] = p.bias[3] = 0.f; p.old = 0.f;
    const int e = threadIdx.x;
    p.pw[0] = p.pw[1] = p.pw[2] = p.pw[3] = 0.f;
    if (mode == 3) {
        const int b = min(e >> 4, a.B - 1), n = min(blk * 16 + (e & 15), a.N - 1);
        const long i = (long)b * a.N + n;
        const float* g = a.pw_gates + (long)b * 4 * a.N;
#pragma unroll
        for (int q = 0; q < 4; ++q) p.bias[q] = g[q * a.N + n];                 // i, f, g, o of the earlier step
        p.old = a.pw_dh_ext[(long)b * a.pw_ld_ext + n];
        p.pw[0] = a.pw_c_t[i]; p.pw[1] = a.pw_c_prev[i]; p.pw[2] = a.pw_dc_carry[i];
    } else if (mode != 2) {
        const int b = min(e >> 4, a.B - 1), n = min(blk * 16 + (e & 15), a.N - 1);
        if (a.bias0) p.bias[0] = a.bias0[n];
        if (a.bias1) p.bias[0] += a.bias1[n];
        if (a.accumulate) p.old = a.out[(long)b * a.ldo + n];
    } else {
        const int b = min(e >> 2, a.B - 1), u = min(blk * 4 + (e & 3), a.C - 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (a.bias0) p.bias[g] = a.bias0[g * a.C + u];
            if (a.bias1) p.bias[g] += a.bias1[g * a.C + u];
        }
        p.old = a.c_prev[(long)b * a.C + u];
    }
    return p;
}

// MODE >= 0 compiles only that epilogue into the kernel: these kernels are launch-bound and their duration follows their
// code size, so an instantiation per mode is worth more than one generic body.
template <int NB, int NWAVES, int MODE = -1>
__device__ __forceinline__ void skinny_epilogue(const SkinnyArgs& a, const float* __restrict__ Gl, int blk, const SkinnyPre& pre0) {
    const int mode = MODE >= 0 ? (MODE == 0 ? (a.mode & 1) : MODE) : a.mode;
    auto gsum = [&](int b, int c) -> float {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) v += Gl[(w * NB * 16 + b) * 17 + c];
        return v;
    };
    auto wrow = [&](int c) -> int {
        if (mode == 2) { const int u = blk * 4 + (c & 3); return u < a.C ? (c >> 2) * a.C + u : -1; }
        const int n = blk * 16 + c;
        return n < a.N ? n : -1;
    };
    if (mode == 3) {
        for (int e = threadIdx.x; e < a.B * 16; e += NWAVES * 64) {
            const int b = e >> 4, c = e & 15, n = wrow(c);
            if (n < 0) continue;
            const bool first = e == (int)threadIdx.x;
            const long i = (long)b * a.N + n;
            const float* g = a.pw_gates + (long)b * 4 * a.N;
            const float ig = first ? pre0.bias[0] : g[n], fg = first ? pre0.bias[1] : g[a.N + n];
            const float gg = first ? pre0.bias[2] : g[2 * a.N + n], og = first ? pre0.bias[3] : g[3 * a.N + n];
            const float dh = (first ? pre0.old : a.pw_dh_ext[(long)b * a.pw_ld_ext + n]) + gsum(b, c);
            const float ct = first ? pre0.pw[0] : a.pw_c_t[i], cp = first ? pre0.pw[1] : a.pw_c_prev[i];
            const float dcc = first ? pre0.pw[2] : a.pw_dc_carry[i];
            const float tc = fast_tanh(ct);
            const float dc = dh * og * (1.f - tc * tc) + dcc;
            float* d = a.pw_dgates + (long)b * 4 * a.N;
            d[n] = dc * gg * ig * (1.f - ig);
            d[a.N + n] = dc * cp * fg * (1.f - fg);
            d[2 * a.N + n] = dc * ig * (1.f - gg * gg);
            d[3 * a.N + n] = dh * tc * og * (1.f - og);
            a.pw_dc_carry[i] = dc * fg;
        }
    } else if (mode != 2) {
        for (int e = threadIdx.x; e < a.B * 16; e += NWAVES * 64) {
            const int b = e >> 4, c = e & 15, row = wrow(c);
            if (row < 0) continue;
            const bool first = e == (int)threadIdx.x;
            float v = gsum(b, c);
            float* o = a.out + (long)b * a.ldo + row;
            if (first) {
                v += pre0.bias[0] + pre0.old;
            } else {
                if (a.bias0) v += a.bias0[row];
                if (a.bias1) v += a.bias1[row];
                if (a.accumulate) v += *o;
            }
            if (mode == 1) v = fast_tanh(v);
            *o = v;
        }
    } else {
        for (int e = threadIdx.x; e < a.B * 4; e += NWAVES * 64) {
            const int b = e >> 2, jj = e & 3, u = blk * 4 + jj;
            if (u >= a.C) continue;
            const bool first = e == (int)threadIdx.x;
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = g * 4 + jj, row = g * a.C + u;
                float v = gsum(b, c);
                if (first) {
                    v += pre0.bias[g];
                } else {
                    if (a.bias0) v += a.bias0[row];
                    if (a.bias1) v += a.bias1[row];
                }
                pre[g] = v;
            }
            const float ig = fast_sig(pre[0]), fg = fast_sig(pre[1]), gg = fast_tanh(pre[2]), og = fast_sig(pre[3]);
            const float cp = first ? pre0.old : a.c_prev[(long)b * a.C + u];
            const float cn = fg * cp + ig * gg;
            a.c_out[(long)b * a.C + u] = cn;
            a.h_out[(long)b * a.C + u] = og * fast_tanh(cn);
            float* go = a.gates_out + (long)b * 4 * a.C;
            go[u] = ig; go[a.C + u] = fg; go[2 * a.C + u] = gg; go[3 * a.C + u] = og;
        }
    }
}

template <int PREC, int NB>
__global__ __launch_bounds__(NT) void skinny_kernel(SkinnyArgs a, int KC, int NCK) {
    typedef typename CT<PREC>::T T;
    constexpr int VEC = CT<PREC>::VEC, KSTEP = CT<PREC>::KSTEP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // k layout: each segment padded to KSTEP; walked in NCK chunks of KC (KC % (4*KSTEP) == 0)
    int koff[4];
    koff[0] = 0;
    for (int s = 0; s < 3; ++s) koff[s + 1] = koff[s] + (s < a.ns ? (a.seg[s].K + KSTEP - 1) / KSTEP * KSTEP : 0);
    const int ld = KC + VEC;
    T* Al = (T*)smem;                         // [NB*16][ld]
    T* Bl = Al + NB * 16 * ld;                // [16][ld]
    float* Gl = (float*)(Bl + 16 * ld);       // [4][NB*16][17]
    const int blk = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const SkinnyPre pre0 = skinny_prefetch<4>(a, blk);

    // W row of tile column c
    auto wrow = [&](int c) -> int {
        if (a.mode == 2) { const int u = blk * 4 + (c & 3); return u < a.C ? (c >> 2) * a.C + u : -1; }
        const int n = blk * 16 + c;
        return n < a.N ? n : -1;
    };
    const int q = KC / KSTEP / 4;             // k-steps per wave per chunk
    f32x4 acc[NB];
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) acc[bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ck = 0; ck < NCK; ++ck) {
        if (ck > 0) __syncthreads();
        // zero both tiles (pad columns, rows beyond B / beyond N), then drop the real pieces in
        {
            typedef T T8 __attribute__((ext_vector_type(16 / sizeof(T))));
            T8 z;
#pragma unroll
            for (int i = 0; i < (int)(16 / sizeof(T)); ++i) z[i] = (T)0;
            const int nvec = (NB * 16 + 16) * ld / (int)(16 / sizeof(T));
            for (int i = threadIdx.x; i < nvec; i += NT) ((T8*)Al)[i] = z;
        }
        __syncthreads();
        const int k0 = ck * KC;
        for (int sidx = 0; sidx < a.ns; ++sidx) {
            const int lo = max(k0, koff[sidx]), hi = min(k0 + KC, koff[sidx] + a.seg[sidx].K);
            if (hi <= lo) continue;
            const int so = lo - koff[sidx];
            stage_block<T>(Al, ld, lo - k0, a.seg[sidx].x + so, a.seg[sidx].ldx, a.B, hi - lo, [](int r) -> long { return r; });
            stage_block<T>(Bl, ld, lo - k0, a.seg[sidx].w + so, a.seg[sidx].ldw, 16, hi - lo, [&](int c) -> long { return wrow(c); });
        }
        __syncthreads();
        mma_rows<PREC, NB>(acc, Al + wave * q * KSTEP, ld, Bl + wave * q * KSTEP, ld, q);
    }
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int bt = 0; bt < NB; ++bt)
#pragma unroll
        for (int r = 0; r < 4; ++r) Gl[(wave * NB * 16 + bt * 16 + fq * 4 + r) * 17 + fr] = acc[bt][r];
    __syncthreads();
    skinny_epilogue<NB, 4>(a, Gl, blk, pre0);
}

// ---- direct variant: both operands go global -> registers -> MFMA, no LDS staging, no barriers before the
// reduction.  Weights are streamed once and not shared between waves, activations are L2 hits: the LDS round trip
// is pure overhead at M = batch (cdna_hip_programming.md, "GEMV / M <= 16 decode weights").  Requires every
// segment to start 16-byte aligned with K_s % 4 == 0 (each lane's two float4 halves are either inside or outside).
// bf16 only: lane (fr = lane&15, fq = lane>>4) holds k = 32*ks + 8*fq + {0..7} of row fr.
// DW = waves per workgroup: 16 for NB <= 2, fewer for more batch tiles (register budget per wave)
template <int NB, int DW, int MODE>
__global__ __launch_bounds__(DW * 64) void skinny_direct_kernel(SkinnyArgs a) {
    __shared__ float Gl[DW * NB * 16 * 17];
    const int blk = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
    auto wrow = [&](int c) -> int {
        if (MODE == 2) { const int u = blk * 4 + (c & 3); return u < a.C ? (c >> 2) * a.C + u : -1; }
        const int n = blk * 16 + c;
        return n < a.N ? n : -1;
    };
    const int row = wrow(fr);
    const SkinnyPre pre0 = skinny_prefetch<DW, MODE>(a, blk);
    f32x4 acc[NB];
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) acc[bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto pack = [](const float4& lo, const float4& hi) -> bf16x8 {       // 4 x v_cvt_pk_bf16_f32
        const u32x4 r = {pack_bf16x2(lo.x, lo.y), pack_bf16x2(lo.z, lo.w), pack_bf16x2(hi.x, hi.y), pack_bf16x2(hi.z, hi.w)};
        return __builtin_bit_cast(bf16x8, r);
    };
    // global k-step index over the concatenated segments; wave w takes k-steps w, w+DW, ...  U of them are resolved
    // to (segment, k) and requested together, so a wave pays one memory round trip per U k-steps whatever the number
    // of segments.  Every load is unconditional from a clamped (always valid) address; out-of-range pieces are
    // zeroed on the data.
    const int rowc = row >= 0 ? row : 0;
    int nks_s[3], cum[4];
    cum[0] = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) { nks_s[i] = i < a.ns ? (a.seg[i].K + 31) / 32 : 0; cum[i + 1] = cum[i] + nks_s[i]; }
    const int total = cum[3];
    // A ROLLED, software-pipelined loop: the operands of k-step g + DW are requested before the MFMAs of k-step g, so
    // two round trips overlap while the body stays ~1 KB (an unrolled three-k-step body was slower: code size).
    struct Frag { float4 bw[2]; float4 ax[NB][2]; bool okw[2], oka[2]; };
    const bool packed = a.wpk != nullptr;
    auto load = [&](Frag& f, int g) {
        const int gc = min(g, total - 1);
        const int si = gc >= cum[2] ? 2 : (gc >= cum[1] ? 1 : 0);
        const Seg& sg = a.seg[si];
        const int k = (gc - cum[si]) * 32 + fq * 8, kmax = sg.K - 4;
        const bool live = g < total;
        f.oka[0] = live && k < sg.K;
        f.oka[1] = live && k + 4 < sg.K;
        f.okw[0] = f.oka[0] && row >= 0;
        f.okw[1] = f.oka[1] && row >= 0;
        if (packed) {
            f.bw[0] = ldg4((const float*)(a.wpk + (((long)blk * total + gc) * 64 + lane) * 8));
        } else {
            const float* __restrict__ wp = sg.w + (long)rowc * sg.ldw;
            f.bw[0] = ldg4(wp + min(k, kmax));
            f.bw[1] = ldg4(wp + min(k + 4, kmax));
        }
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) {
            const int b = min(bt * 16 + fr, a.B - 1);
            const float* xp = sg.x + (long)b * sg.ldx;
            f.ax[bt][0] = ldg4(xp + min(k, kmax));
            f.ax[bt][1] = ldg4(xp + min(k + 4, kmax));
        }
    };
    auto mma = [&](const Frag& f) {
        // (a packed fragment needs no mask: it is zero where it must be, and a k-step past the end meets zeroed A)
        const bf16x8 bf = packed ? __builtin_bit_cast(bf16x8, f.bw[0]) : pack(sel4(f.okw[0], f.bw[0]), sel4(f.okw[1], f.bw[1]));
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) {
            const bool okb = bt * 16 + fr < a.B;
            acc[bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                pack(sel4(okb && f.oka[0], f.ax[bt][0]), sel4(okb && f.oka[1], f.ax[bt][1])), bf, acc[bt], 0, 0, 0);
        }
    };
    if (wave < total) {
        Frag cur, nxt;
        load(cur, wave);
#pragma unroll 1
        for (int g = wave; g < total; g += DW) {
            load(nxt, g + DW);                            // (clamped, zeroed on the data when past the end)
            mma(cur);
            cur = nxt;
        }
    }
#pragma unroll
    for (int bt = 0; bt < NB; ++bt)
#pragma unroll
        for (int r = 0; r < 4; ++r) Gl[(wave * NB * 16 + bt * 16 + fq * 4 + r) * 17 + fr] = acc[bt][r];
    __syncthreads();
    skinny_epilogue<NB, DW, MODE>(a, Gl, blk, pre0);
}

// grid (blocks of 16 output rows, k-steps), 64 threads: lane (fr, fq) converts W[row(fr)][k0 + 8 fq .. +7] of its segment
__global__ __launch_bounds__(64) void skinny_pack_kernel(SkinnyArgs a, int total, bf16_t* __restrict__ out) {
    const int blk = blockIdx.x, g = blockIdx.y, lane = threadIdx.x, fr = lane & 15, fq = lane >> 4;
    int row;
    if (a.mode == 2) { const int u = blk * 4 + (fr & 3); row = u < a.C ? (fr >> 2) * a.C + u : -1; }
    else { const int n = blk * 16 + fr; row = n < a.N ? n : -1; }
    int cum[4];
    cum[0] = 0;
    for (int i = 0; i < 3; ++i) cum[i + 1] = cum[i] + (i < a.ns ? (a.seg[i].K + 31) / 32 : 0);
    const int si = g >= cum[2] ? 2 : (g >= cum[1] ? 1 : 0);
    const Seg& sg = a.seg[si];
    const int k = (g - cum[si]) * 32 + fq * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (row >= 0 && k + j < sg.K) ? sg.w[(long)row * sg.ldw + k + j] : 0.f;
    const u32x4 r = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    *(u32x4*)(out + (((long)blk * total + g) * 64 + lane) * 8) = r;
}

constexpr size_t SKINNY_LDS_TARGET = 72 * 1024;      // keep >= 2 workgroups per CU

// choose the k-chunking: returns LDS bytes, sets KC and NCK
size_t skinny_plan(int prec, const SkinnyArgs& a, int NB, int& KC, int& NCK) {
    const int sz = prec == LAS_PREC_BF16 ? 2 : 4, vec = prec == LAS_PREC_BF16 ? 8 : 4, ks = prec == LAS_PREC_BF16 ? 32 : 16;
    int k = 0;
    for (int s = 0; s < a.ns; ++s) k += (a.seg[s].K + ks - 1) / ks * ks;
    const int quantum = 4 * ks;
    const size_t fixed = sizeof(float) * 4 * NB * 16 * 17;
    for (NCK = 1;; ++NCK) {
        KC = ((k + NCK - 1) / NCK + quantum - 1) / quantum * quantum;
        const size_t lds = (size_t)(NB * 16 + 16) * (KC + vec) * sz + fixed;
        if (lds <= SKINNY_LDS_TARGET || KC == quantum) return lds;
    }
}

template <int PREC, int NB>
int launch(const SkinnyArgs& a, size_t lds, int KC, int NCK, int grid, hipStream_t st) {
    auto k = skinny_kernel<PREC, NB>;
    if (lds > 64 * 1024) LAS_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, st, a, KC, NCK);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

}  // namespace

// Internal C++ entry used by the decoder driver (decoder.hip) and by the public wrappers below.
int las_skinny_launch_pw(int prec, const float* x0, long ldx0, const float* w0, long ldw0, int K0, const float* x1,
                         long ldx1, const float* w1, long ldw1, int K1, const float* x2, long ldx2, const float* w2,
                         long ldw2, int K2, int B, int N, const float* bias0, const float* bias1, int mode, float* out,
                         long ldo, int accumulate, int C, const float* c_prev, float* h_out, float* c_out,
                         float* gates_out, const las_skinny_pw* pw, hipStream_t st) {
    return las_skinny_launch_pk(prec, x0, ldx0, w0, ldw0, K0, x1, ldx1, w1, ldw1, K1, x2, ldx2, w2, ldw2, K2, B, N, bias0, bias1,
                                mode, out, ldo, accumulate, C, c_prev, h_out, c_out, gates_out, pw, nullptr, st);
}

int las_skinny_launch_pk(int prec, const float* x0, long ldx0, const float* w0, long ldw0, int K0, const float* x1,
                         long ldx1, const float* w1, long ldw1, int K1, const float* x2, long ldx2, const float* w2,
                         long ldw2, int K2, int B, int N, const float* bias0, const float* bias1, int mode, float* out,
                         long ldo, int accumulate, int C, const float* c_prev, float* h_out, float* c_out,
                         float* gates_out, const las_skinny_pw* pw, const void* wpk, hipStream_t st) {
    SkinnyArgs a{};
    a.wpk = prec == LAS_PREC_BF16 ? (const bf16_t*)wpk : nullptr;
    a.seg[0] = Seg{x0, ldx0, w0, ldw0, K0};
    a.seg[1] = Seg{x1, ldx1, w1, ldw1, K1};
    a.seg[2] = Seg{x2, ldx2, w2, ldw2, K2};
    a.ns = x2 ? 3 : (x1 ? 2 : 1);
    a.B = B; a.N = N; a.bias0 = bias0; a.bias1 = bias1; a.mode = mode; a.out = out; a.ldo = ldo;
    a.accumulate = accumulate; a.C = C; a.c_prev = c_prev; a.h_out = h_out; a.c_out = c_out; a.gates_out = gates_out;
    if (pw) {
        if (mode != 0 || accumulate || bias0 || bias1) return LAS_E_BADARG;
        a.mode = 3;
        a.pw_dh_ext = pw->dh_ext; a.pw_ld_ext = pw->ld_ext; a.pw_dc_carry = pw->dc_carry; a.pw_gates = pw->gates;
        a.pw_c_t = pw->c_t; a.pw_c_prev = pw->c_prev; a.pw_dgates = pw->dgates;
    }
    const int NB = las_pick_nb(B);
    if (NB == 0 || B <= 0 || N <= 0 || K0 <= 0) return LAS_E_UNSUPPORTED;
    if (prec != LAS_PREC_BF16 && prec != LAS_PREC_F32) return LAS_E_BADARG;
    if (prec == LAS_PREC_BF16) {
        bool ok = true;
        for (int i = 0; i < a.ns; ++i) {
            const Seg& g = a.seg[i];
            ok = ok && (g.K % 4 == 0) && (g.ldx % 4 == 0) && (g.ldw % 4 == 0) && ((((uintptr_t)g.x) & 15) == 0) &&
                 ((((uintptr_t)g.w) & 15) == 0);
        }
        if (ok) {
            const int grid_d = mode == 2 ? (C + 3) / 4 : (N + 15) / 16;
#define LAS_SKD_GO(M_) LAS_NB_SWITCH(NB, { constexpr int DW_ = NB_ <= 2 ? 16 : (NB_ == 4 ? 8 : 4); hipLaunchKernelGGL((skinny_direct_kernel<NB_, DW_, M_>), dim3(grid_d), dim3(DW_ * 64), 0, st, a); LAS_LAUNCH_OK(); return LAS_OK; })
            if (a.mode == 2) { LAS_SKD_GO(2); } else if (a.mode == 3) { LAS_SKD_GO(3); } else { LAS_SKD_GO(0); }
#undef LAS_SKD_GO
        }
    }
    int KC = 0, NCK = 0;
    const size_t lds = skinny_plan(prec, a, NB, KC, NCK);
    if (lds > 160 * 1024) return LAS_E_UNSUPPORTED;
    const int grid = mode == 2 ? (C + 3) / 4 : (N + 15) / 16;
    if (prec == LAS_PREC_BF16) { LAS_NB_SWITCH(NB, return (launch<LAS_PREC_BF16, NB_>(a, lds, KC, NCK, grid, st))); }
    else { LAS_NB_SWITCH(NB, return (launch<LAS_PREC_F32, NB_>(a, lds, KC, NCK, grid, st))); }
    return LAS_E_BADARG;
}

int las_skinny_launch(int prec, const float* x0, long ldx0, const float* w0, long ldw0, int K0, const float* x1,
                      long ldx1, const float* w1, long ldw1, int K1, const float* x2, long ldx2, const float* w2,
                      long ldw2, int K2, int B, int N, const float* bias0, const float* bias1, int mode, float* out,
                      long ldo, int accumulate, int C, const float* c_prev, float* h_out, float* c_out,
                      float* gates_out, hipStream_t st) {
    return las_skinny_launch_pw(prec, x0, ldx0, w0, ldw0, K0, x1, ldx1, w1, ldw1, K1, x2, ldx2, w2, ldw2, K2, B, N, bias0, bias1,
                                mode, out, ldo, accumulate, C, c_prev, h_out, c_out, gates_out, nullptr, st);
}

extern "C" int las_lstm_cell_fwd(int prec, const float* x, int64_t ldx, int Kx, const float* h_prev, const float* c_prev,
                                 const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int B, int C,
                                 float* h_out, float* c_out, float* gates_out, void* stream) {
    LAS_CHECK_ARG(x && h_prev && c_prev && w_ih && w_hh && h_out && c_out && gates_out && B > 0 && C > 0 && Kx > 0);
    return las_skinny_launch(prec, x, ldx, w_ih, Kx, Kx, h_prev, C, w_hh, C, C, nullptr, 0, nullptr, 0, 0, B, 4 * C, b_ih,
                             b_hh, 2, nullptr, 0, 0, C, c_prev, h_out, c_out, gates_out, (hipStream_t)stream);
}

extern "C" int las_skinny_linear(int prec, const float* x, int64_t ldx, const float* w, int64_t ldw, int B, int N, int K,
                                 const float* bias, int act, int accumulate, float* out, int64_t ldo, void* stream) {
    LAS_CHECK_ARG(x && w && out && B > 0 && N > 0 && K > 0);
    return las_skinny_launch(prec, x, ldx, w, ldw, K, nullptr, 0, nullptr, 0, 0, nullptr, 0, nullptr, 0, 0, B, N, bias,
                             nullptr, act ? 1 : 0, out, ldo, accumulate, 0, nullptr, nullptr, nullptr, nullptr,
                             (hipStream_t)stream);
}

// ---- pre-packed weight operand (bf16 mode): same segment list / row mapping as the product that will read it
extern "C" size_t las_skinny_pack_bytes(int N, int K0, int K1, int K2, int cell_mode, int C) {
    const long nblk = cell_mode ? (C + 3) / 4 : (N + 15) / 16;
    const long total = (K0 + 31) / 32 + (K1 > 0 ? (K1 + 31) / 32 : 0) + (K2 > 0 ? (K2 + 31) / 32 : 0);
    return (size_t)(nblk * total * 64 * 8 * sizeof(bf16_t));
}

extern "C" int las_skinny_pack_weights(const float* w0, int64_t ldw0, int K0, const float* w1, int64_t ldw1, int K1,
                                       const float* w2, int64_t ldw2, int K2, int N, int cell_mode, int C, void* out,
                                       void* stream) {
    LAS_CHECK_ARG(w0 && out && K0 > 0 && N > 0 && (!cell_mode || (C > 0 && N == 4 * C)));
    SkinnyArgs a{};
    a.seg[0] = Seg{nullptr, 0, w0, ldw0, K0};
    a.seg[1] = Seg{nullptr, 0, w1, ldw1, w1 ? K1 : 0};
    a.seg[2] = Seg{nullptr, 0, w2, ldw2, w2 ? K2 : 0};
    a.ns = w2 ? 3 : (w1 ? 2 : 1);
    a.N = N; a.C = C; a.mode = cell_mode ? 2 : 0;
    const int nblk = cell_mode ? (C + 3) / 4 : (N + 15) / 16;
    int total = 0;
    for (int i = 0; i < a.ns; ++i) total += (a.seg[i].K + 31) / 32;
    hipLaunchKernelGGL(skinny_pack_kernel, dim3(nblk, total), dim3(64), 0, (hipStream_t)stream, a, total, (bf16_t*)out);
    LAS_LAUNCH_OK();
    return LAS_OK;
}
