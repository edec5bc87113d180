// Persistent (Bi)LSTM recurrence for gfx950: forward and back-propagation-through-time.
//
// Replaces the sequential half of nn.LSTM(bidirectional, packed) at reference src/asr.py:473-481 and
// its autograd backward, including pack_padded_sequence/pad_packed_sequence (:480,483: zero output at
// t >= len, state only advances inside the utterance) and the drop/concat time down-sampling
// (:487-497) which is folded into the output addressing.
//
// Design (time-major activations):
//   * x*W_ih^T for all timesteps and both directions is ONE MFMA GEMM done beforehand (las_gemm);
//     this kernel receives xproj[T][B][ND*4H] and runs only the recurrence.
//   * grid = ND * G workgroups, one per CU, all co-resident.  Workgroup (d,g) owns U (<=16) hidden units
//     of direction d: the 4*U rows of W_hh it needs stay in LDS for all T steps (weight-stationary),
//     the cell state c stays in registers.
//   * per step: h_{t-1}[B,H] (published by the G workgroups of direction d) is pulled into LDS, wave w
//     computes gate w's pre-activations with MFMA (A = h tile [16 batch x K], B = W rows [K x 16 units]),
//     accumulators cross LDS once, the 256 threads do the pointwise cell update for (batch, unit-pair)
//     and publish their slice of h_t.
//   * hand-off between CUs (MI355X_MICROARCH "Valid forms", row 1): every published byte is written by an
//     sc1 (write-through) store, every storing wave drains vmcnt(0), workgroup barrier, ONE lane adds to
//     an agent-scope counter; consumers poll that counter with an sc1 load from one lane, workgroup
//     barrier, then read the bytes with sc1 buffer loads (L1 bypass) only.  One workgroup per CU is
//     enforced by the LDS request.  Spins are bounded; a timeout sets *status and every workgroup exits.
//   * the published history doubles as the saved activations for BPTT.
// Backward mirrors this with dgates[B,4H] as the exchanged quantity and W_hh^T columns resident.
#include <type_traits>
#include "las_mma.h"
#include <stdlib.h>

namespace {

constexpr int NT = 256;
constexpr unsigned SPIN_LIMIT = 1u << 22;
constexpr size_t MIN_LDS = 84 * 1024;            // > 80 KiB: at most one workgroup per CU

constexpr int MAX_SLICES = 16;
constexpr int CNT_STRIDE = 128;     // words per (direction, slice) group, lines of their own (pollers of one group must
                                    // not share a line with another group's atomics): word 0 arrival counter, words 32-34
                                    // placement rendezvous, words 64-95 per-producer progress words (L2-local mode)
constexpr int FLAG_OFS = 64;
// The exchanged h lives in a ring of HX_SLOTS steps per direction (slot = step & 3), not in a [T] history: no workgroup
// can be more than one step ahead of the slowest reader of its group, and lines that are rewritten every four steps stay
// resident in the L2 -- a store to a line the L2 does not hold has to wait for its fill before the consumers' reads are
// served, which cost 0.25 us per step once T*B*H outgrew the memory-side cache (T = 1200: 2.20 vs 1.95 us).
constexpr int HX_SLOTS = 4;
struct SyncWords {          // zeroed by hipMemsetAsync before every launch
    unsigned cnt[2 * MAX_SLICES * CNT_STRIDE];   // arrivals per (direction, batch slice)
    unsigned abort_;                // set on spin timeout
    unsigned pad[63];
};

// one-v_exp activations for the cell pointwise (abs err ~1e-7; the recurrence is fp32 throughout)
// (v_rcp_f32, 1 ulp: __fdividef compiles to the full IEEE division sequence -- v_div_scale, v_rcp, four FMAs, v_div_fmas,
// v_div_fixup -- on this toolchain)
__device__ __forceinline__ float fsig(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }

// One lane polls `cnt >= target` (sc1 loads); returns false on timeout / abort.
__device__ __forceinline__ bool wait_counter(unsigned* cnt, unsigned target, unsigned* abort_word) {
    unsigned spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023u) == 0) {
            if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
            if (spins > SPIN_LIMIT) {
                __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
    return true;
}

// The thread that polls and signals: lane 0 of the LAST wave.  A wave's vector-memory results return in issue order,
// and with <= 16-row batch slices the pointwise work (and with it the cold x-projection prefetch and the
// backward-only stores) lives in waves 0-1; the last wave's queue holds nothing but its share of the pull, so its
// poll is never stuck behind a cold HBM access.
#define LAS_SYNC_THREAD (NT - 64)

// L2-local mode (the group's G <= 32 workgroups share one XCD): no atomics at all.  Producer g keeps a progress word
// (steps published so far, a plain store that lands in the XCD's L2); the last wave polls all G words at once, lane l
// the word of producer l, with L1-bypassing loads that hit that L2.  (An agent-scope atomic and its poll go to the
// memory side whatever their scope bits say: 1.8 us per step for a single workgroup handing off to itself.)
__device__ __forceinline__ bool wait_flags(unsigned* flags, int G, unsigned target, unsigned* abort_word) {
    unsigned* p = flags + min((int)(threadIdx.x & 63), G - 1);
    unsigned spins = 0;
    while (__builtin_amdgcn_ballot_w64(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023u) == 0) {
            if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
            if (spins > SPIN_LIMIT) {
                __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
    return true;
}

// Block-wide wait for `steps` publications of every workgroup of the group; result broadcast through LDS word `flag`.
__device__ __forceinline__ bool block_wait(unsigned* cnt, int G, unsigned steps, unsigned* abort_word, int* flag, bool local) {
    if (local) {
        if (threadIdx.x >= LAS_SYNC_THREAD && threadIdx.x < NT) {
            const bool ok = wait_flags(cnt + FLAG_OFS, G, steps, abort_word);
            if (threadIdx.x == LAS_SYNC_THREAD) *flag = ok ? 1 : 0;
        }
    } else if (threadIdx.x == LAS_SYNC_THREAD) *flag = wait_counter(cnt, (unsigned)G * steps, abort_word) ? 1 : 0;
    __syncthreads();
    return *flag != 0;
}

// Publish: every storing wave has drained its stores; one lane signals (`steps` = publications so far, this one
// included).  `local`: every workgroup of the group sits on ONE XCD, checked at run time by group_local below.
__device__ __forceinline__ void block_signal(unsigned* cnt, bool local, int g, unsigned steps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == LAS_SYNC_THREAD) {
        if (local) cnt[FLAG_OFS + g] = steps;
        else __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Exchange stores.  Across XCDs every published byte is an sc1 (write-through) store; inside one XCD a plain store
// (write-through L1, line KEPT in the shared L2) is what the consumers' L1-bypassing loads hit at L2 speed
// (MI355X_MICROARCH.md, handoff-payload row: 104-122 vs 62-73 GB/s per block).
__device__ __forceinline__ void st_pair_x(bf16_t* p, float a, float b, bool local) {
    if (local) *(unsigned*)p = pack_bf16x2(a, b);
    else __hip_atomic_store((unsigned*)p, pack_bf16x2(a, b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_pair_x(float* p, float a, float b, bool local) {
    const unsigned long long v = (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
    if (local) *(unsigned long long*)p = v;
    else __hip_atomic_store((unsigned long long*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Pull a [rows x cols] tile (row stride `src_ld` elements, published by other CUs) into LDS rows of stride `ld`:
// 16 bytes per lane per load, sc1 (L1 bypass).  All of a thread's loads (up to 8) are issued before the first LDS
// write; out-of-range lanes read through the buffer descriptor's bounds check (returns 0, no branch).
template <typename T, int VEC, int U>
__device__ __forceinline__ void pull_tile_sc1(const T* __restrict__ src, int rows, int cols, int src_ld, int col0,
                                              T* __restrict__ lds, int ld) {
    const int vpr = cols / VEC;                                  // vectors per row
    const int total = rows * vpr;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, rows * src_ld * (int)sizeof(T), 0x00020000);
    for (int i0 = threadIdx.x; i0 < total; i0 += NT * U) {
        u32x4 v[U];
        int dst[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * NT;
            const int r = i / vpr, c = (i - r * vpr) * VEC;
            dst[u] = i < total ? r * ld + c : -1;
            const int off = i < total ? (r * src_ld + col0 + c) * (int)sizeof(T) : 0x7ffffff0;
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (dst[u] >= 0) *(u32x4*)(lds + dst[u]) = v[u];
    }
}

// acc[bt] += A(tile rows, k) * Wfrag over KS k-steps with BOTH operands in registers: the A fragments come straight
// from global memory (sc1 buffer loads of the tile other CUs just published), the B fragments (weights) were loaded
// once at kernel start.  Lane (fr = lane&15, fq = lane>>4) holds k = 32*ks + 8*fq + {0..7} of row fr.
// Out-of-range rows / columns read through the descriptor's bounds check (0, no branch).  bf16 only.
template <int NB, int KS>
__device__ __forceinline__ void mma_direct(f32x4 (&acc)[NB], const bf16_t* __restrict__ tile, int rows, int row_ld,
                                           int col0, int cols, const bf16x8 (&wfrag)[KS]) {
    const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)tile, 0, rows * row_ld * 2, 0x00020000);
    u32x4 afr[NB][KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) {
            const int r = bt * 16 + fr, c = ks * 32 + fq * 8;
            const int off = (r < rows && c < cols) ? (r * row_ld + col0 + c) * 2 : 0x7ffffff0;
            afr[bt][ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
        }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int bt = 0; bt < NB; ++bt)
            acc[bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, afr[bt][ks]), wfrag[ks], acc[bt], 0, 0, 0);
}

struct LstmArgs {
    int T, B, H, ND, U, G;
    int NS, Bs;               // batch slices (independent sub-recurrences) and rows per slice
    int sr, concat, T_out, F_out;
    int y_is_hf;
    int wdirect;              // backward K-split: W_hh fragments straight from global memory (no LDS slab)
    int xl;                   // XCD-grouped launch: grid = 8 * G * ceil(groups / 8), see lstm_role
    int nt;                   // forward: non-temporal stores for the saved activations (C2 step -0.2 ms; the same for the
                              // backward kernel's dgf measured neutral)
    int pf;                   // forward: a fifth wave prefetches the x-projection rows into the L2 two steps ahead
    int pdelay;               // (diagnostic build) granule kernels: units of 64 cycles to sleep before a step's first poll
    int fxI;                  // forward, granule kernel: > 0 = the layer's input projection is FUSED: `xproj` is the layer input x
                              // [T][B][fxI] (fxI <= 96) and the kernel forms x W_ih^T itself (three extra MFMAs per step); 0: xproj
    const float* w_ih;        // [ND*4H][fxI] (fused input projection only)
    void* tw;                 // bf16 twin of the launch's main output (forward: y; backward: d gates), written by the granule /
                              // 32-unit kernels next to the fp32 stores (the GEMMs behind the layer read it); null: none
};

// Which (direction d, unit slice g, batch slice bs) a workgroup works on.  A GROUP = the G workgroups of one
// (d, bs): they exchange h / dgates every step and share a counter.  Default: consecutive ids.  xl: workgroups are
// dealt round-robin to the 8 XCDs (ids congruent mod 8 share one: observed, not contractual), so group `grp` takes
// ids == grp (mod 8) and all its hand-offs stay inside one XCD's L2; ids whose group does not exist exit at once.
struct Role { int d, g, bs; bool idle; };
__device__ __forceinline__ Role lstm_role(const LstmArgs& a) {
    Role r;
    const int bid = blockIdx.x;
    if (a.xl) {
        const int label = bid & 7, li = bid >> 3, grp = (li / a.G) * 8 + label;
        r.g = li % a.G; r.idle = grp >= a.ND * a.NS; r.d = grp / a.NS; r.bs = grp - r.d * a.NS;
    } else {
        r.d = bid / (a.G * a.NS); r.g = (bid % (a.G * a.NS)) / a.NS; r.bs = bid % a.NS; r.idle = false;
    }
    return r;
}

// Placement is only ever a speed assumption: before the first step the G workgroups of a group meet once through the
// placement-independent protocol (agent-scope atomics) and compare their hardware XCC ids.  1: all on one XCD -> the
// L2-local exchange (plain stores, L2 atomics); 0: not -> the sc1 protocol; -1: timeout.  `w` = the second 128-byte
// line of the group's counter slot (zeroed with it).
__device__ __forceinline__ int group_local(const LstmArgs& a, unsigned* cnt, unsigned* abort_word, int* flag) {
    if (!a.xl) return 0;
    if (threadIdx.x == LAS_SYNC_THREAD) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        x &= 15u;
        unsigned* w = cnt + 32;
        const unsigned o1 = __hip_atomic_fetch_max(w + 1, x + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned o2 = __hip_atomic_fetch_max(w + 2, 16u - x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::"v"(o1), "v"(o2) : "memory");          // both maxima performed before the arrival below
        __hip_atomic_fetch_add(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int r = -1;
        if (wait_counter(w, (unsigned)a.G, abort_word))
            r = (__hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                 __hip_atomic_load(w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 17u ? 1 : 0;
        *flag = r;
    }
    __syncthreads();
    const int r = *flag;
    __syncthreads();
    return r;
}

__device__ __forceinline__ long y_offset(const LstmArgs& a, int t, int b, int d, int j, bool& ok) {
    // layer output / its gradient, time-major [T_out][B][F_out]; concat: reference asr.py:493-494, drop: :491
    if (a.sr == 1) { ok = true; return ((long)t * a.B + b) * a.F_out + d * a.H + j; }
    const int u = t / a.sr, r = t - u * a.sr;
    if (a.concat) { ok = u < a.T_out; return ((long)u * a.B + b) * a.F_out + r * (a.ND * a.H) + d * a.H + j; }
    ok = (r == 0);
    return ((long)u * a.B + b) * a.F_out + d * a.H + j;
}

// ------------------------------------------------------------------------------------------------ forward
template <int PREC, int NB, int KS>     // KS > 0: register-resident operands (bf16), KS k-steps per wave; 0: LDS path
__global__ __launch_bounds__(KS > 16 ? NT : NT + 64) void lstm_fwd_kernel(LstmArgs a, const float* __restrict__ xproj,
                                                      const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                      const float* __restrict__ w_hh, const int32_t* __restrict__ lens,
                                                      float* __restrict__ y, float* __restrict__ hf,
                                                      typename CT<PREC>::T* __restrict__ hx, float* __restrict__ gates,
                                                      float* __restrict__ cs, SyncWords* sync, int* status) {
    typedef typename CT<PREC>::T T;
    constexpr int VEC = CT<PREC>::VEC, KSTEP = CT<PREC>::KSTEP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = a.H, B = a.B, U = a.U, ND = a.ND;
    const int Kp = (H + KSTEP - 1) / KSTEP * KSTEP, ld = Kp + VEC;
    const int Hx = (H + VEC - 1) / VEC * VEC;           // exchange row stride (pad columns are caller-zeroed)
    const Role role = lstm_role(a);
    if (role.idle) return;
    const int d = role.d, g = role.g, bs = role.bs, j0 = g * U;
    const int b0 = bs * a.Bs, Bl = min(a.Bs, B - b0);  // my batch slice: rows [b0, b0+Bl)
    constexpr bool WREG = KS > 0;                       // weights in registers, fetched from global memory once
    T* Wl = (T*)smem;                                   // [4][16][ld]   (LDS path only)
    T* Hl = Wl + (WREG ? 0 : 4 * 16 * ld);              // [NB*16][ld]
    float* Gl = (float*)(Hl + NB * 16 * ld);            // [4][NB*16][17]
    int* lensl = (int*)(Gl + 4 * NB * 16 * 17);         // [NB*16]
    int* flag = lensl + NB * 16;

    // ---- one-time staging: W_hh rows of my units (zero padded), zero h tile, lens
    const bool pfw = threadIdx.x >= NT;                 // the optional fifth wave (a.pf), see below
    if (!pfw) {
        if constexpr (!WREG)
        for (int i = threadIdx.x; i < 4 * 16 * ld; i += NT) {
            const int k = i % ld, n = (i / ld) % 16, gi = i / (ld * 16);
            float v = 0.f;
            if (k < H && n < U && j0 + n < H) v = w_hh[((long)d * 4 * H + gi * H + j0 + n) * H + k];
            Wl[i] = PREC == LAS_PREC_BF16 ? (T)f2bf(v) : (T)v;
        }
        for (int i = threadIdx.x; i < NB * 16 * ld; i += NT) Hl[i] = (T)0;
        for (int i = threadIdx.x; i < NB * 16; i += NT) lensl[i] = i < Bl ? lens[b0 + i] : 0;
    }
    __syncthreads();
    if (pfw) {
        // Prefetch wave.  A wave's loads return in issue order, so the HBM-cold read of the x-projection that a
        // pointwise wave issues at the top of a step (~3 600 cycles) holds back the pull of h it issues 1 700 cycles
        // later (cycle stamps: the pull took 1 900 cycles, all of it that wait).  This wave touches every 64-byte
        // segment the workgroup will read two steps from now, so those reads hit the L2; it never waits for its own
        // loads inside a step and only keeps pace through the workgroup's barriers.
        const int lane_ = threadIdx.x & 63, segs = NB * 16 * 4;
        unsigned* cnt_ = &sync->cnt[(d * MAX_SLICES + bs) * CNT_STRIDE];
        const int gl_ = group_local(a, cnt_, &sync->abort_, flag);
        if (gl_ < 0) return;
        constexpr int NQ = (NB * 16 * 4 + 63) / 64;
        float sink = 0.f, pa[NQ], pb[NQ];               // two sets: a value is consumed two steps after its request,
#pragma unroll                                          // so this wave never stalls on the way to a barrier
        for (int q = 0; q < NQ; ++q) pa[q] = pb[q] = 0.f;
        auto touch = [&](int s, float (&pr)[NQ]) {
            const int s3 = min(s + 3, a.T - 1), t3 = d == 0 ? s3 : a.T - 1 - s3;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int sg = lane_ + 64 * q, row = sg >> 2, gi = sg & 3;
                sink += pr[q];
                pr[q] = (sg < segs && row < Bl) ? xproj[((long)t3 * B + b0 + row) * (ND * 4 * H) + d * 4 * H + gi * H + j0] : 0.f;
            }
        };
        auto step = [&](int s, float (&pr)[NQ]) -> bool {
            touch(s, pr);
            if (s > 0) {
                if (!block_wait(cnt_, a.G, (unsigned)s, &sync->abort_, flag, gl_ == 1)) return false;
                __syncthreads();                        // pull
            }
            __syncthreads();                            // accumulators in LDS
            __syncthreads();                            // block_signal
            return true;
        };
        for (int s = 0; s < a.T; s += 2) {
            if (!step(s, pa)) return;
            if (s + 1 < a.T && !step(s + 1, pb)) return;
        }
        if (sink == 1.2345e38f) *status = 0;            // (keeps the loads)
        return;
    }
    // register-resident weight fragments of my gate (wave w <-> gate w): lane (fr, fq) holds k = 32 ks + 8 fq + {0..7}
    // of W_hh row (gate w, unit j0 + fr), straight from global memory (no LDS slab: at H = 1024 the four gates' rows of
    // 16 units are 128 KB of bf16, which together with the h tile would not fit the CU's 160 KB)
    bf16x8 wfrag[KS > 0 ? KS : 1];
    if constexpr (KS > 0) {
        const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6, n_ = lane_ & 15;
        const bool rowok = n_ < U && j0 + n_ < H;
        const float* wrow = w_hh + ((long)d * 4 * H + wave_ * H + min(j0 + n_, H - 1)) * H;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c = ks * 32 + (lane_ >> 4) * 8;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = wrow[min(c + e, H - 1)];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (rowok && c + e < H) ? v[e] : 0.f;
            const u32x4 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
            wfrag[ks] = __builtin_bit_cast(bf16x8, pk);
        }
    }

    // ---- my pointwise elements: ONE (batch row, unit) per thread and 256-element pass (NB passes): the cell update is
    // VALU-issue bound with one wave per SIMD (five v_exp/v_rcp chains per element), so it is spread over all the
    // lanes the slice can fill (12 rows x 16 units = waves 0-2; the polling wave stays free when Bs <= 12)
    constexpr int PE = NB;
    float c_state[PE];
    float bias[PE][4];
    int eb[PE], en[PE];
    bool ev[PE];
#pragma unroll
    for (int p = 0; p < PE; ++p) {
        const int e = threadIdx.x + p * NT;
        eb[p] = e >> 4; en[p] = e & 15;
        ev[p] = eb[p] < Bl && en[p] < U && (j0 + en[p] < H);
        c_state[p] = 0.f;
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const int j = j0 + en[p];
            bias[p][gi] = ev[p] ? b_ih[d * 4 * H + gi * H + j] + b_hh[d * 4 * H + gi * H + j] : 0.f;
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
    const int ND4H = ND * 4 * H;
    unsigned* cnt = &sync->cnt[(d * MAX_SLICES + bs) * CNT_STRIDE];
    const int gl = group_local(a, cnt, &sync->abort_, flag);
    if (gl < 0) { if (threadIdx.x == 0) *status = LAS_E_TIMEOUT; return; }
    const bool local = gl == 1;

    for (int s = 0; s < a.T; ++s) {
        const int t = d == 0 ? s : a.T - 1 - s;
        // (a) prefetch x-projection of my elements (independent of the recurrence)
        float xp[PE][4];
#pragma unroll
        for (int p = 0; p < PE; ++p)
#pragma unroll
            for (int gi = 0; gi < 4; ++gi)
                xp[p][gi] = ev[p] ? xproj[((long)t * B + b0 + eb[p]) * ND4H + d * 4 * H + gi * H + j0 + en[p]] : 0.f;
        // (b,c) wait for h_{t-1} of every unit of my direction, pull it into LDS
        f32x4 acc[NB];
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) acc[bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            if (!block_wait(cnt, a.G, (unsigned)s, &sync->abort_, flag, local)) { if (threadIdx.x == 0) *status = LAS_E_TIMEOUT; return; }
            // every wave needs the whole h tile: one shared pull into LDS (a direct global->register read per wave
            // fetches it four times in fragment-shaped pieces: slower across XCDs, and 2.9 vs 2.06 us inside one)
            // (one pass: all of a thread's loads are in flight before its first LDS write)
            pull_tile_sc1<T, VEC, (KS > 16 ? 8 : 2) * NB>(hx + (((long)d * HX_SLOTS + ((s - 1) & (HX_SLOTS - 1))) * B + b0) * Hx, Bl, Hx, Hx, 0, Hl, ld);
            __syncthreads();
            // (d) gate pre-activations: wave w <-> gate w
            if constexpr (KS > 0) {               // weights from registers, h from LDS
                // every fragment is requested before the first MFMA (unconditionally: a k-step beyond Kp re-reads the
                // last one against a zero weight fragment); issued one by one behind a branch each, the ten
                // ds_read -> mfma pairs of a wave took 1 460 cycles per step, all of it LDS latency (now 570)
                // (KS > 16: in chunks of CH k-steps, the h fragments of a whole row would not fit beside the weights)
                const int lane_ = threadIdx.x & 63;
                constexpr int CH = KS <= 16 ? KS : 8;
#pragma unroll
                for (int k0 = 0; k0 < KS; k0 += CH) {
                    bf16x8 av[CH][NB];
#pragma unroll
                    for (int ks = 0; ks < CH; ++ks)
#pragma unroll
                        for (int bt = 0; bt < NB; ++bt)
                            av[ks][bt] = *(const bf16x8*)((const bf16_t*)Hl + (bt * 16 + (lane_ & 15)) * ld + min((k0 + ks) * 32, Kp - 32) + (lane_ >> 4) * 8);
#pragma unroll
                    for (int ks = 0; ks < CH; ++ks)
#pragma unroll
                        for (int bt = 0; bt < NB; ++bt)
                            acc[bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ks][bt], wfrag[k0 + ks], acc[bt], 0, 0, 0);
                }
            } else {
                mma_rows<PREC, NB>(acc, Hl, ld, Wl + wave * 16 * ld, ld, Kp / KSTEP);
            }
        }
        // (e) accumulators -> LDS  (C/D layout: col = lane&15 = unit, row = (lane>>4)*4 + r = batch)
#pragma unroll
        for (int bt = 0; bt < NB; ++bt)
#pragma unroll
            for (int r = 0; r < 4; ++r) Gl[(wave * NB * 16 + bt * 16 + fq * 4 + r) * 17 + fr] = acc[bt][r];
        __syncthreads();
        // (f) pointwise cell update; publish h_t FIRST, signal, then write what only the backward pass reads
        float hv[PE], gv[PE][4];
#pragma unroll
        for (int p = 0; p < PE; ++p) {
            const int bl = min(eb[p], NB * 16 - 1), n = en[p], j = j0 + n;
            const bool mq = ev[p] && t < lensl[bl];
            const float pi = Gl[(0 * NB * 16 + bl) * 17 + n] + xp[p][0] + bias[p][0];
            const float pf = Gl[(1 * NB * 16 + bl) * 17 + n] + xp[p][1] + bias[p][1];
            const float pg = Gl[(2 * NB * 16 + bl) * 17 + n] + xp[p][2] + bias[p][2];
            const float po = Gl[(3 * NB * 16 + bl) * 17 + n] + xp[p][3] + bias[p][3];
            const float ig = fsig(pi), fg = fsig(pf), gg = ftanh(pg), og = fsig(po);
            const float cn = fg * c_state[p] + ig * gg;
            const float hn = og * ftanh(cn);
            c_state[p] = mq ? cn : c_state[p];
            hv[p] = mq ? hn : 0.f;
            gv[p][0] = mq ? ig : 0.f; gv[p][1] = mq ? fg : 0.f; gv[p][2] = mq ? gg : 0.f; gv[p][3] = mq ? og : 0.f;
            // the unit pair (n, n+1) goes out as one store from the even lane; its partner's h comes over DPP (row_shl:1)
            const float hnext = las_dpp<0x101, 0xf>(0.f, hv[p]);
            if (ev[p] && !(n & 1)) st_pair_x(hx + (((long)d * HX_SLOTS + (s & (HX_SLOTS - 1))) * B + b0 + bl) * Hx + j, hv[p], hnext, local);
        }
        // (g) publish: only the exchange stores are outstanding here
        block_signal(cnt, local, g, (unsigned)s + 1u);
#pragma unroll
        for (int p = 0; p < PE; ++p) {
            if (!ev[p]) continue;
            const int bl = eb[p], b = b0 + bl, j = j0 + en[p];
            const bool m = t < lensl[bl];
            const long ro = (long)t * B + b;
            if (a.nt) {
                // (write-only streams read by later kernels: non-temporal, so they neither wait for nor keep L2 lines)
                __builtin_nontemporal_store(hv[p], &hf[ro * (ND * H) + d * H + j]);
                if (!a.y_is_hf) {
                    bool ok;
                    const long yo = y_offset(a, t, b, d, j, ok);
                    if (ok) __builtin_nontemporal_store(hv[p], &y[yo]);
                }
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) __builtin_nontemporal_store(gv[p][gi], &gates[ro * ND4H + d * 4 * H + gi * H + j]);
                __builtin_nontemporal_store(m ? c_state[p] : 0.f, &cs[ro * (ND * H) + d * H + j]);
                continue;
            }
            hf[ro * (ND * H) + d * H + j] = hv[p];
            if (!a.y_is_hf) {
                bool ok;
                const long yo = y_offset(a, t, b, d, j, ok);
                if (ok) y[yo] = hv[p];
            }
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) gates[ro * ND4H + d * 4 * H + gi * H + j] = gv[p][gi];
            cs[ro * (ND * H) + d * H + j] = m ? c_state[p] : 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward, tagged granules
// The same recurrence with a hand-off that has NO flag, NO drain and ONE workgroup barrier per step (bf16, weights in
// registers).  Differences from lstm_fwd_kernel:
//   * MFMA roles are swapped: A = the wave's 16 weight rows ordered (unit, gate) -- wave w owns units 4w .. 4w+3 of the
//     workgroup's 16 --, B = the h tile.  Lane (fr, fq) then receives acc[0..3] = the i, f, g, o pre-activations of unit
//     4w + fq for batch row fr: the cell update runs on the accumulators, no trip through LDS, no barrier.
//   * h_t leaves as self-validating 16-byte granules {6 batch rows of ONE unit as bf16, tag = step + 1} written by one
//     lane with one store (the six rows are gathered over DPP row shifts); the ring of four steps is zeroed before the
//     launch, so a tag can only be matched by this launch's store of that step.  A consumer sweeps its share of the
//     H * ceil(Bs / 6) granules (consecutive lanes <-> consecutive granules) until every tag matches, writes the rows into
//     the h tile in LDS (double-buffered: the one barrier per step separates a tile's writers from its readers) and goes
//     on.  MI355X_MICROARCH.md "R2": a naturally aligned 16-byte store is observed untorn by 16-byte sc1 loads on gfx950
//     (not an architectural guarantee); against the flag form it saves the vmcnt(0) drain, the signal store, the poll's
//     round trip and three barriers (handoff-1to1 vs handoff-flag rows of the guide's price table).
//   * In steady state the FIRST pass of a sweep matches (cycle stamps: 1.00-1.03 passes per step): a step pays one L2
//     round trip for the hand-off.  (Several passes in flight were tried; the compiler's register renaming of the
//     re-issued loads forces a full wait per iteration, and there is nothing left for them to win.)
//   * The four compute waves touch global memory ONLY for the hand-off.  A wave's vector-memory operations complete in
//     order, so x-projection loads (HBM) and the saved-activation stores in front of a sweep delayed it: 1.85 us per step
//     with them, 1.57 without the stores, 1.45 without both (H = 320, B = 24).  A fifth wave does that I/O: it loads the
//     x-projection rows three steps ahead and hands them over in LDS, and stores h / y / gates / c of the step before
//     from LDS; the barrier of the step is its only synchronisation with the compute waves.
__device__ __forceinline__ int gr_chunks(int rows) { return (rows >> 4) * 3 + ((rows & 15) + 5) / 6; }
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
// One poll of a granule: an sc1 (L1-bypassing) 16-byte load the compiler can neither merge with an identical poll nor
// hoist out of the polling loop (its offset passes through an empty volatile asm), but still counts in vmcnt.
__device__ __forceinline__ u32x4 gr_poll(__amdgpu_buffer_rsrc_t rs, int off) {
    asm volatile("" : "+v"(off));
    return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
}
#ifdef LAS_PK_STAMPS            // diagnostic build only (make stamps): cycle sums per phase of a step, printed by workgroup 0
#define GR_ST_DECL unsigned gst[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gst_last = (unsigned)__builtin_amdgcn_s_memtime()
#define GR_ST(i) do { const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime(); gst[i] += now_ - gst_last; gst_last = now_; } while (0)
#define GR_ST_PRINT(T) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) printf("gr stamps wave %d T=%d: sweep %u barrier %u mfma/cell %u cell/mfma+store %u passes %u barrierB %u\n", (int)(threadIdx.x >> 6), T, gst[0] / T, gst[1] / T, gst[2] / T, gst[3] / T, gst[4], gst[5] / T); } while (0)
#else
#define GR_ST_DECL
#define GR_ST(i)
#define GR_ST_PRINT(T)
#endif
constexpr int GR_XLD = 16 * 4 + 4;        // floats per batch row of the x-projection / gate tiles in LDS: [16 units][4 gates] + pad
constexpr int GR_HLD = 16 * 2 + 2;        // ... of the {h, c} tile

// FX (one batch tile, narrow layer input: the bottom layer's 80 fbank dims): the input projection is fused.  The I/O wave hands
// over x_t as a bf16 tile instead of the x-projection's rows, the compute waves hold their 16 rows of W_ih as three k-steps of
// fragments and start the step's accumulators with x_t W_ih^T.  The projection GEMM of the bottom layer is pure output traffic
// (K = 80: 295 MB written at the C2 shape, 118 us) that this kernel then read back from HBM, 245 KB per step.
constexpr int FX_KS = 3, FX_LD = FX_KS * 32 + 8;        // k-steps of the fused projection (I <= 96), bf16 row stride of the x tile
template <int NB, int KS, int SWM = 4, bool FX = false>     // SWM: 16-byte granules a lane sweeps per step (4: per-lane lists in registers; more: lists in LDS)
__global__ __launch_bounds__(NT + 64) void lstm_fwd_gr_kernel(LstmArgs a, const float* __restrict__ xproj,
                                                      const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                      const float* __restrict__ w_hh, const int32_t* __restrict__ lens,
                                                      float* __restrict__ y, float* __restrict__ hf,
                                                      u32x4* __restrict__ ring, float* __restrict__ gates,
                                                      float* __restrict__ cs, SyncWords* sync, int* status) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = a.H, B = a.B, ND = a.ND;
    // (row stride = 16 (mod 32) elements: the 16-lane groups of a ds_read_b128 -- rows 0-3 and 12-15 at one k quarter, rows 4-11 at
    // the next -- then hit 16 different 16-byte slots of the bank row; with a pad of 8 seven of the 16 lanes collided at H = 320)
    const int Kp = (H + 31) / 32 * 32, ld = Kp + 16;
    const Role role = lstm_role(a);
    if (role.idle) return;
    const int d = role.d, g = role.g, bs = role.bs, j0 = g * 16;
    const int b0 = bs * a.Bs, Bl = min(a.Bs, B - b0);
    bf16_t* Hl = (bf16_t*)smem;                         // [2][NB*16][ld]; pad rows / columns stay zero
    float* Xl = (float*)(Hl + 2 * NB * 16 * ld);        // [2][NB*16][GR_XLD]  x-projection of the step (I/O wave -> compute)
    float* Sg = Xl + 2 * NB * 16 * GR_XLD;              // [2][NB*16][GR_XLD]  post-activation gates (compute -> I/O wave)
    float* Sh = Sg + 2 * NB * 16 * GR_XLD;              // [2][NB*16][GR_HLD]  {h, c}
    int* lensl = (int*)(Sh + 2 * NB * 16 * GR_HLD);     // [NB*16]
    int* flag = lensl + NB * 16;
    int* tab = flag + 4;                                // SWM > 4: [NT][SWM][2] sweep lists (ring offset, h-tile destination)
    const bool io = threadIdx.x >= NT;                  // the fifth wave
    if (!io) {
        for (int i = threadIdx.x; i < NB * 16 * ld; i += NT) ((unsigned*)Hl)[i] = 0u;      // 2 tiles of bf16 = NB*16*ld words
        for (int i = threadIdx.x; i < NB * 16; i += NT) lensl[i] = i < Bl ? lens[b0 + i] : 0;
    }
    __syncthreads();
    unsigned* cnt = &sync->cnt[(d * MAX_SLICES + bs) * CNT_STRIDE];
    const int gl = group_local(a, cnt, &sync->abort_, flag);       // only decides the store flavour here
    if (gl < 0) { if (threadIdx.x == 0) *status = LAS_E_TIMEOUT; return; }
    const bool local = gl == 1;
    const int ND4H = ND * 4 * H;
    const int lane = threadIdx.x & 63;
    if (io) {
        // ---- I/O wave.  Item = 4 consecutive units of one (batch row, gate): lane + 64 q -> row = item / 16,
        // gate = (item / 4) % 4, quarter = item % 4; 16-byte buffer accesses (H % 4 == 0), masked lanes use an
        // out-of-range offset instead of a branch, so the wave's code is straight-line and its waits are exact counts.
        constexpr int NQ = FX ? 6 : NB * 4;              // FX: item = 4 consecutive input dims of one batch row: 16 x 24 items
        constexpr int OOB = 0x7ffffff0;
        const long nrow = (long)a.T * B;
        __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xproj, 0, (int)(nrow * (FX ? a.fxI : ND4H) * 4), 0x00020000);
        __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)gates, 0, (int)(nrow * ND4H * 4), 0x00020000);
        __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)hf, 0, (int)(nrow * ND * H * 4), 0x00020000);
        __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)cs, 0, (int)(nrow * ND * H * 4), 0x00020000);
        __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)((long)a.T_out * B * a.F_out * 4), 0x00020000);
        __amdgpu_buffer_rsrc_t ry16 = __builtin_amdgcn_make_buffer_rsrc(a.tw, 0, a.tw ? (int)((long)a.T_out * B * a.F_out * 2) : 0, 0x00020000);
        auto tstep = [&](int s) { const int sc = min(s, a.T - 1); return d == 0 ? sc : a.T - 1 - sc; };
        auto xload = [&](int s, u32x4 (&xr)[NQ]) {
            const int t = tstep(s);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if constexpr (FX) {
                    const int it = lane + 64 * q, row = it / 24, k0 = (it - row * 24) * 4;
                    const int off = (row < Bl && k0 < a.fxI) ? (int)((((long)t * B + b0 + row) * a.fxI + k0) * 4) : OOB;      // (beyond the input width: zeros)
                    xr[q] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
                } else {
                    const int it = lane + 64 * q, row = it >> 4, gi = (it >> 2) & 3, u0 = (it & 3) * 4;
                    const int off = (row < Bl && j0 + u0 < H) ? (int)((((long)t * B + b0 + row) * ND4H + d * 4 * H + gi * H + j0 + u0) * 4) : OOB;
                    xr[q] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
                }
            }
        };
        auto xstore = [&](int s, const u32x4 (&xr)[NQ]) {
            if constexpr (FX) {                          // x_t as a bf16 tile [16][FX_LD] (in Xl's space), zeros beyond the rows / the width
                bf16_t* xb = (bf16_t*)Xl + (s & 1) * 16 * FX_LD;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int it = lane + 64 * q, row = it / 24, k0 = (it - row * 24) * 4;
                    if (row < 16)
                        *(u32x2*)(xb + row * FX_LD + k0) = (u32x2){pack_bf16x2(__uint_as_float(xr[q][0]), __uint_as_float(xr[q][1])),
                                                                 pack_bf16x2(__uint_as_float(xr[q][2]), __uint_as_float(xr[q][3]))};
                }
            } else {
            unsigned* xl = (unsigned*)Xl + (s & 1) * NB * 16 * GR_XLD;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int it = lane + 64 * q, row = it >> 4, gi = (it >> 2) & 3, u0 = (it & 3) * 4;
                unsigned* p = xl + row * GR_XLD + u0 * 4 + gi;
                p[0] = xr[q][0]; p[4] = xr[q][1]; p[8] = xr[q][2]; p[12] = xr[q][3];
            }
            }
        };
        auto sflush = [&](int s, bool valid) {           // saved activations of step s: LDS -> global (non-temporal)
            const int t = tstep(s);
            const unsigned* sg = (const unsigned*)Sg + (s & 1) * NB * 16 * GR_XLD;
            const unsigned* sh = (const unsigned*)Sh + (s & 1) * NB * 16 * GR_HLD;
#pragma unroll
            for (int q = 0; q < NB * 4; ++q) {
                const int it = lane + 64 * q, row = it >> 4, gi = (it >> 2) & 3, u0 = (it & 3) * 4;
                const unsigned* p = sg + row * GR_XLD + u0 * 4 + gi;
                const u32x4 v = {p[0], p[4], p[8], p[12]};
                const int off = (valid && row < Bl && j0 + u0 < H) ? (int)((((long)t * B + b0 + row) * ND4H + d * 4 * H + gi * H + j0 + u0) * 4) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(v, rg, off, 0, 2);
            }
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int it = lane + 64 * q, row = it >> 2, u0 = (it & 3) * 4;
                const unsigned* p = sh + row * GR_HLD + u0 * 2;
                const u32x4 hv4 = {p[0], p[2], p[4], p[6]}, cv4 = {p[1], p[3], p[5], p[7]};
                const bool ok = valid && row < Bl && j0 + u0 < H;
                const int b = b0 + row;
                const int off = ok ? (int)((((long)t * B + b) * (ND * H) + d * H + j0 + u0) * 4) : OOB;
                bool yok = false;
                const long yo = y_offset(a, t, b, d, j0 + u0, yok);
                const int offy = (ok && yok && !a.y_is_hf) ? (int)(yo * 4) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(hv4, rh, off, 0, 2);
                __builtin_amdgcn_raw_buffer_store_b128(cv4, rc, off, 0, 2);
                __builtin_amdgcn_raw_buffer_store_b128(hv4, ry, offy, 0, 2);
                // y's bf16 twin (an empty resource when there is none: every offset is out of range)
                const int off16 = a.y_is_hf ? off : ((ok && yok) ? (int)(yo * 4) : OOB);
                const u32x2 h16 = {pack_bf16x2(__uint_as_float(hv4[0]), __uint_as_float(hv4[1])), pack_bf16x2(__uint_as_float(hv4[2]), __uint_as_float(hv4[3]))};
                __builtin_amdgcn_raw_buffer_store_b64(h16, ry16, off16 == OOB ? OOB : off16 >> 1, 0, 0);
            }
        };
        // D register sets: a value is used D phases after its request.  Three sufficed while the x-projection came out of the
        // L2 / MALL; at T = 1 200 (295 MB of it at the C2 shape) the rows come from HBM and a 3.2 us lead left the step waiting:
        // 1.04 us per step at T = 600, 1.17 at 1 200, 1.28 at 2 400 (tools/sweep_lstm.py).  Six sets for one batch tile: 1.11 at
        // T = 1 200, c3 15.53 -> 15.37 ms (four sets: 15.42; nine: slower -- the loads queue in front of the saved-activation stores).
        constexpr int D = NB == 1 ? 6 : 3;
        u32x4 xs[D][NQ];
#pragma unroll
        for (int k = 0; k < D; ++k) xload(k, xs[k]);
        xstore(0, xs[0]); xload(D, xs[0]);
        __syncthreads();                                 // barrier(0)
        // phase p (between barrier(p) and barrier(p+1)): hand over step p+1's x-projection, request step p+1+D's, store step
        // p-1.  (The only back edge follows the D-th phase, so the compiler's wait counts at the loop head stay exact.)
        for (int s = 0;; s += D) {
            bool done = false;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                if (done) break;
                xstore(s + k + 1, xs[(k + 1) % D]); xload(s + k + 1 + D, xs[(k + 1) % D]);
                sflush(max(s + k - 1, 0), s + k > 0);
                __syncthreads();
                done = s + k + 1 >= a.T;
            }
            if (done) break;
        }
        sflush(a.T - 1, true);
        return;
    }
    const int wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
    // A operand: row fr of the wave's tile = (unit 4 wave + fr / 4, gate fr % 4); lane holds k = 32 ks + 8 fq + {0..7}
    bf16x8 wfrag[KS];
    {
        const int jw = j0 + 4 * wave + (fr >> 2), gi = fr & 3;
        const bool rowok = jw < H;
        const float* wrow = w_hh + ((long)d * 4 * H + gi * H + min(jw, H - 1)) * H;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c = ks * 32 + fq * 8;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = wrow[min(c + e, H - 1)];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (rowok && c + e < H) ? v[e] : 0.f;
            const u32x4 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
            wfrag[ks] = __builtin_bit_cast(bf16x8, pk);
        }
    }
    // FX: the same 16 rows of W_ih, k = 32 ks + 8 fq + {0..7} of the input width (zeros beyond it)
    bf16x8 xfrag[FX ? FX_KS : 1];
    if constexpr (FX) {
        const int jw = j0 + 4 * wave + (fr >> 2), gi = fr & 3, I_ = a.fxI;
        const bool rowok = jw < H;
        const float* wrow = a.w_ih + ((long)d * 4 * H + gi * H + min(jw, H - 1)) * I_;
#pragma unroll
        for (int ks = 0; ks < FX_KS; ++ks) {
            const int c = ks * 32 + fq * 8;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (rowok && c + e < I_) ? wrow[min(c + e, I_ - 1)] : 0.f;
            const u32x4 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
            xfrag[ks] = __builtin_bit_cast(bf16x8, pk);
        }
    }
    // my cell elements: unit je (ul within the workgroup), batch rows bt * 16 + fr
    const int ul = 4 * wave + fq, je = j0 + ul;
    const bool evu = je < H;
    float c_state[NB];
    f32x4 bias;
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) bias[gi] = evu ? b_ih[d * 4 * H + gi * H + je] + b_hh[d * 4 * H + gi * H + je] : 0.f;
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) c_state[bt] = 0.f;
    const int nck = gr_chunks(a.Bs), nckl = gr_chunks(Bl), total = H * nckl;
    const long slot_stride = (long)a.NS * H * nck;                      // granules per ring slot of one direction
    u32x4* ringg = ring + (long)d * HX_SLOTS * slot_stride + (long)bs * H * nck;
    const int cc = fr >= 12 ? 2 : fr >= 6 ? 1 : 0;
    const bool writer = evu && fr == cc * 6;
    // my share of the sweep: the workgroup's H * nckl granules in four contiguous quarters, one per wave (equal work, and a
    // wave instruction reads consecutive granules); slot u of lane l is granule wave * Q + 64 u + l.  Offsets in the ring
    // slot, destinations in the h tile and the spare-row flags are the same every step.
    constexpr int SWMAX = SWM;                           // host side: H * chunks(Bs) <= SWM * 256
    constexpr bool TAB = SWM > 4;                        // (H = 1024: 128 registers hold the weight fragments; the lists live in LDS)
    const int Q = (total + 3) / 4;
    const int nsw = (Q + 63) / 64;
    int g_off[TAB ? 1 : SWMAX], g_dst[TAB ? 1 : SWMAX];  // g_dst: (destination << 1) | six-row flag, or -1
    int* tabw = tab + (int)threadIdx.x * SWMAX * 2;
#pragma unroll
    for (int u = 0; u < SWMAX; ++u) {
        const int idx = lane + 64 * u, i = wave * Q + idx;
        const bool ok = idx < Q && i < total;
        const int j = ok ? i / nckl : 0, ch = ok ? i - j * nckl : 0, bt = ch / 3, c3 = ch - bt * 3;
        const int o_ = ok ? (j * nck + ch) * 16 : 0x7ffffff0;
        const int d_ = ok ? ((((bt * 16 + c3 * 6) * ld + j) << 1) | (c3 < 2 ? 1 : 0)) : -1;
        if constexpr (TAB) { tabw[2 * u] = o_; tabw[2 * u + 1] = d_; }
        else { g_off[u] = o_; g_dst[u] = d_; }
    }

    GR_ST_DECL;
    for (int s = 0; s < a.T; ++s) {
        const int t = d == 0 ? s : a.T - 1 - s;
        bf16_t* buf = Hl + (s & 1) * NB * 16 * ld;
        if (s > 0) {
            // ---- sweep: the granules of step s-1, until every tag reads s.  Branch-free: a granule that has matched is
            // re-requested at an out-of-range offset (returns zeros without memory traffic), the loop is uniform over the wave.
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(ringg + ((s - 1) & (HX_SLOTS - 1)) * slot_stride), 0,
                                                                          H * nck * 16, 0x00020000);
            auto sweep = [&](auto swc) -> bool {
                constexpr int SW = decltype(swc)::value;
                constexpr int OOB = 0x7ffffff0;
                u32x4 v[SW], got[TAB ? 1 : SW];
                int off[SW];
                for (int dl = 0; dl < (a.pdelay & 255); ++dl) __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int u = 0; u < SW; ++u) {
                    if constexpr (TAB) off[u] = tabw[2 * u]; else { off[u] = g_off[u]; got[u] = (u32x4){0u, 0u, 0u, 0u}; }
                    v[u] = gr_poll(rs, off[u]);
                }
                unsigned spins = 0;
                while (true) {
                    bool need = false;
#pragma unroll
                    for (int u = 0; u < SW; ++u) {
                        const bool hit = off[u] != OOB && v[u][3] == (unsigned)s;
                        off[u] = hit ? OOB : off[u];
                        if constexpr (TAB) {
                            if (hit) {                   // (lists in LDS: no registers to park the payload in; the loop runs once in steady state)
                                const int d2 = tabw[2 * u + 1];
                                bf16_t* dst = buf + (d2 >> 1);
                                dst[0] = (bf16_t)(v[u][0] & 0xffffu); dst[ld] = (bf16_t)(v[u][0] >> 16);
                                dst[2 * ld] = (bf16_t)(v[u][1] & 0xffffu); dst[3 * ld] = (bf16_t)(v[u][1] >> 16);
                                if (d2 & 1) { dst[4 * ld] = (bf16_t)(v[u][2] & 0xffffu); dst[5 * ld] = (bf16_t)(v[u][2] >> 16); }
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 3; ++e) got[u][e] = hit ? v[u][e] : got[u][e];
                        }
                        need = need || off[u] != OOB;
                    }
#ifdef LAS_PK_STAMPS
                    ++gst[4];
#endif
                    if (__builtin_amdgcn_ballot_w64(need) == 0ull) break;
#pragma unroll
                    for (int u = 0; u < SW; ++u) v[u] = gr_poll(rs, off[u]);
                    if ((++spins & 255u) == 0) {
                        if (__hip_atomic_load(&sync->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || spins > SPIN_LIMIT) {
                            __hip_atomic_store(&sync->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            *status = LAS_E_TIMEOUT;
                            return false;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < SW; ++u) {
                    if constexpr (TAB) continue;
                    const int d2 = g_dst[u];
                    if (d2 < 0) continue;
                    bf16_t* dst = buf + (d2 >> 1);
                    dst[0] = (bf16_t)(got[u][0] & 0xffffu); dst[ld] = (bf16_t)(got[u][0] >> 16);
                    dst[2 * ld] = (bf16_t)(got[u][1] & 0xffffu); dst[3 * ld] = (bf16_t)(got[u][1] >> 16);
                    if (d2 & 1) { dst[4 * ld] = (bf16_t)(got[u][2] & 0xffffu); dst[5 * ld] = (bf16_t)(got[u][2] >> 16); }
                }
                return true;
            };
            bool ok_;
            if constexpr (SWM > 4) ok_ = nsw <= 6 ? sweep(std::integral_constant<int, 6>{}) : sweep(std::integral_constant<int, SWM>{});
            else ok_ = nsw <= 2 ? sweep(std::integral_constant<int, 2>{})
                     : nsw == 3 ? sweep(std::integral_constant<int, 3>{}) : sweep(std::integral_constant<int, 4>{});
            if (!ok_) return;
        }
        GR_ST(0);
        __syncthreads();                                 // barrier(s): h tile complete, x-projection of step s in Xl
GR_ST(1);
        // ---- gate pre-activations of my four units: x-projection + bias + W_hh h_{t-1}
        f32x4 acc[NB], acc2[NB];
        const float* xl = Xl + (s & 1) * NB * 16 * GR_XLD;
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) {
            if constexpr (FX) {                          // x_t W_ih^T: three MFMAs on the bf16 x tile the I/O wave handed over
                const bf16_t* xb = (const bf16_t*)Xl + (s & 1) * 16 * FX_LD;
                acc[bt] = bias;
#pragma unroll
                for (int ks = 0; ks < FX_KS; ++ks)
                    acc[bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xfrag[ks], *(const bf16x8*)(xb + fr * FX_LD + ks * 32 + fq * 8), acc[bt], 0, 0, 0);
            } else {
                acc[bt] = *(const f32x4*)(xl + (bt * 16 + fr) * GR_XLD + ul * 4) + bias;
            }
            acc2[bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (s > 0) {
            if constexpr (KS > 16) {
                // 128 registers hold the weights: the h fragments come in chunks of 4 k-steps, the NEXT chunk requested before
                // the current one's MFMAs (one chunk = 4 LDS reads per lane; without the overlap every chunk waited out the
                // LDS latency: 1 900 cycles per step for 32 MFMAs)
                constexpr int CH = 4, NCH = KS / CH;
                bf16x8 hc[2][CH];
#pragma unroll
                for (int ks = 0; ks < CH; ++ks) hc[0][ks] = *(const bf16x8*)(buf + fr * ld + min(ks * 32, Kp - 32) + fq * 8);
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (c + 1 < NCH) {
#pragma unroll
                        for (int ks = 0; ks < CH; ++ks)
                            hc[(c + 1) & 1][ks] = *(const bf16x8*)(buf + fr * ld + min(((c + 1) * CH + ks) * 32, Kp - 32) + fq * 8);
                    }
                    __builtin_amdgcn_sched_barrier(0);       // (keep the next chunk's reads in front of this chunk's MFMAs)
#pragma unroll
                    for (int ks = 0; ks < CH; ++ks) {
                        if (ks & 1) acc2[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag[c * CH + ks], hc[c & 1][ks], acc2[0], 0, 0, 0);
                        else acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag[c * CH + ks], hc[c & 1][ks], acc[0], 0, 0, 0);
                    }
                }
            } else {
            constexpr int CH = KS <= 10 ? KS : 8;
#pragma unroll
            for (int k0 = 0; k0 < KS; k0 += CH) {
                bf16x8 hv_[CH][NB];
#pragma unroll
                for (int ks = 0; ks < CH; ++ks)
#pragma unroll
                    for (int bt = 0; bt < NB; ++bt)
                        // (predicating the read on `batch row < Bl` -- 6 of 16 rows in use -- made this phase SLOWER: 580 -> 800 cycles)
                        hv_[ks][bt] = *(const bf16x8*)(buf + (bt * 16 + fr) * ld + min((k0 + ks) * 32, Kp - 32) + fq * 8);
                // (round 3: without this fence hipcc sinks every read to just in front of its MFMA -- `ds_read; s_waitcnt lgkmcnt(1);
                // v_mfma` ten times over, each MFMA behind a full LDS latency: 58 cycles per MFMA in the ISA of round 2's build)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < CH; ++ks)
#pragma unroll
                    for (int bt = 0; bt < NB; ++bt) {
                        if (ks & 1) acc2[bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag[k0 + ks], hv_[ks][bt], acc2[bt], 0, 0, 0);
                        else acc[bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag[k0 + ks], hv_[ks][bt], acc[bt], 0, 0, 0);
                    }
            }
            }
        }
#ifdef LAS_PK_STAMPS
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc2[0][0]));
#endif
        GR_ST(2);
        // ---- cell update on the accumulators; publish h_t, then leave the saved activations to the I/O wave
        u32x4* slot = ringg + (s & (HX_SLOTS - 1)) * slot_stride;
        float* sg = Sg + (s & 1) * NB * 16 * GR_XLD;
        float* sh = Sh + (s & 1) * NB * 16 * GR_HLD;
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) {
            const int row = bt * 16 + fr;
            const bool mq = evu && row < Bl && t < lensl[row];
            const f32x4 pre = acc[bt] + acc2[bt];
            const float ig = fsig(pre[0]), fg = fsig(pre[1]), gg = ftanh(pre[2]), og = fsig(pre[3]);
            const float cn = fg * c_state[bt] + ig * gg;
            const float hn = og * ftanh(cn);
            c_state[bt] = mq ? cn : c_state[bt];
            const float hv = mq ? hn : 0.f;
            const float hnext = las_dpp<0x101, 0xf>(0.f, hv);                      // row_shl:1 -> batch row fr + 1
            const unsigned p0 = pack_bf16x2(hv, hnext), p1 = dpp_u<0x102>(p0), p2 = dpp_u<0x104>(p0);
            {
                // (one rsrc per workgroup, the granule's place as a per-lane offset: a per-lane rsrc costs a waterfall loop)
                const u32x4 gr = {p0, p1, p2, (unsigned)s + 1u};
                __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc((void*)slot, 0, H * nck * 16, 0x00020000);
                const int woff = (writer && row < Bl) ? je * nck * 16 + (bt * 3 + cc) * 16 : 0x7ffffff0;
                if (local) __builtin_amdgcn_raw_buffer_store_b128(gr, ws, woff, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(gr, ws, woff, 0, 16);
            }
            *(f32x4*)(sg + row * GR_XLD + ul * 4) = mq ? (f32x4){ig, fg, gg, og} : (f32x4){0.f, 0.f, 0.f, 0.f};
            *(float2*)(sh + row * GR_HLD + ul * 2) = make_float2(hv, mq ? c_state[bt] : 0.f);
        }
        GR_ST(3);
    }
    GR_ST_PRINT(a.T);
    __syncthreads();                                     // barrier(T): the I/O wave stores the last step
}

// ------------------------------------------------------------------------------------------------ backward
// dh_rec[b][j] = sum_m dgates_next[b][m] * W_hh[m][j]; K = 4H is walked in NC chunks so the pulled
// dgates tile fits in LDS for large H; wave w takes a quarter of each chunk's k-steps.
template <int PREC, int NB, int KS>     // KS > 0: register-resident operands (bf16, NC == 1), KS k-steps per wave
__global__ __launch_bounds__(NT) void lstm_bwd_kernel(LstmArgs a, int NC, int K4p, const float* __restrict__ dy,
                                                      const float* __restrict__ gates, const float* __restrict__ cs,
                                                      const float* __restrict__ w_hh, const int32_t* __restrict__ lens,
                                                      typename CT<PREC>::T* __restrict__ dgx, float* __restrict__ dgf,
                                                      SyncWords* sync, int* status) {
    typedef typename CT<PREC>::T T;
    constexpr int VEC = CT<PREC>::VEC, KSTEP = CT<PREC>::KSTEP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = a.H, B = a.B, U = a.U, ND = a.ND, K4 = 4 * H;
    const int KC = K4p / NC;                              // chunk width; K4p = 4H zero-padded so KC % (4*KSTEP) == 0
    const int ldw = K4p + VEC, ldc = KC + VEC;
    const Role role = lstm_role(a);
    if (role.idle) return;
    const int d = role.d, g = role.g, bs = role.bs, j0 = g * U;
    const int b0 = bs * a.Bs, Bl = min(a.Bs, B - b0);
    T* Wl = (T*)smem;                                     // [16][ldw]   W_hh^T columns of my units
    T* Dl = Wl + 16 * ldw;                                // [NB*16][ldc] dgates_next chunk
    float* Gl = (float*)(Dl + NB * 16 * ldc);             // [4][NB*16][17] per-wave partial sums
    int* lensl = (int*)(Gl + 4 * NB * 16 * 17);
    int* flag = lensl + NB * 16;

    for (int i = threadIdx.x; i < 16 * ldw; i += NT) {
        const int m = i % ldw, n = i / ldw;
        float v = 0.f;
        if (m < K4 && n < U && j0 + n < H) v = w_hh[((long)d * K4 + m) * H + j0 + n];
        Wl[i] = PREC == LAS_PREC_BF16 ? (T)f2bf(v) : (T)v;
    }
    for (int i = threadIdx.x; i < NB * 16 * ldc; i += NT) Dl[i] = (T)0;
    for (int i = threadIdx.x; i < NB * 16; i += NT) lensl[i] = i < Bl ? lens[b0 + i] : 0;
    __syncthreads();
    // register-resident W_hh^T fragments of my k-quarter (wave w <-> k-steps [w*KS, (w+1)*KS))
    bf16x8 wfrag[KS > 0 ? KS : 1];
    if constexpr (KS > 0) {
        const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            wfrag[ks] = *(const bf16x8*)((const bf16_t*)Wl + (lane_ & 15) * ldw + (wave_ * KS + ks) * 32 + (lane_ >> 4) * 8);
    }

    constexpr int PP = (NB * 16 * 8 + NT - 1) / NT;
    const int half = U / 2;
    float dc_carry[PP][2];
    int eb[PP], en[PP];
    bool ev[PP];
#pragma unroll
    for (int p = 0; p < PP; ++p) {
        const int e = threadIdx.x + p * NT;
        eb[p] = e / half; en[p] = (e % half) * 2;
        ev[p] = (e < NB * 16 * half) && eb[p] < Bl && (j0 + en[p] < H);
        dc_carry[p][0] = dc_carry[p][1] = 0.f;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
    const int ND4H = ND * K4, NDH = ND * H;
    unsigned* cnt = &sync->cnt[(d * MAX_SLICES + bs) * CNT_STRIDE];
    const int kq = KC / KSTEP / 4;                        // k-steps per wave per chunk
    const int gl = group_local(a, cnt, &sync->abort_, flag);
    if (gl < 0) { if (threadIdx.x == 0) *status = LAS_E_TIMEOUT; return; }
    const bool local = gl == 1;

    for (int s = 0; s < a.T; ++s) {
        const int t = d == 0 ? a.T - 1 - s : s;           // reverse of the forward processing order
        const int tn = d == 0 ? t + 1 : t - 1;            // step handled just before (later in forward order)
        const int tp = d == 0 ? t - 1 : t + 1;            // previous step in forward order (c_{prev})
        // (a) prefetch saved activations and the incoming gradient
        float2 sg[PP][4], sc[PP], scp[PP], sdy[PP];
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            sc[p] = scp[p] = sdy[p] = make_float2(0.f, 0.f);
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) sg[p][gi] = make_float2(0.f, 0.f);
            if (!ev[p]) continue;
            const int bl = eb[p], b = b0 + bl, j = j0 + en[p];
            if (t >= lensl[bl]) continue;
            const long ro = (long)t * B + b;
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) sg[p][gi] = *(const float2*)(gates + ro * ND4H + d * K4 + gi * H + j);
            sc[p] = *(const float2*)(cs + ro * NDH + d * H + j);
            if (tp >= 0 && tp < lensl[bl]) scp[p] = *(const float2*)(cs + ((long)tp * B + b) * NDH + d * H + j);
            bool ok;
            const long yo = y_offset(a, t, b, d, j, ok);
            if (ok) sdy[p] = *(const float2*)(dy + yo);
        }
        // (b,c,d) dh_rec from the previous step's dgates
        f32x4 acc[NB];
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) acc[bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            if (!block_wait(cnt, a.G, (unsigned)s, &sync->abort_, flag, local)) { if (threadIdx.x == 0) *status = LAS_E_TIMEOUT; return; }
            const T* src = dgx + (((long)d * a.T + tn) * B + b0) * K4;
            if constexpr (KS > 0) {
                mma_direct<NB, KS>(acc, (const bf16_t*)src, Bl, K4, wave * KS * 32, K4 - wave * KS * 32, wfrag);
            } else
            for (int c = 0; c < NC; ++c) {
                if (c > 0) __syncthreads();               // previous chunk fully consumed
                // rows have stride K4 in memory; this chunk = real columns [c*KC, c*KC + kreal)
                pull_tile_sc1<T, VEC, 8>(src, Bl, min(KC, K4 - c * KC), K4, c * KC, Dl, ldc);
                __syncthreads();
                // A = Dl rows (batch) x k ; B = Wl rows (unit) x k, offset to this chunk/wave quarter
                mma_rows<PREC, NB>(acc, Dl + wave * kq * KSTEP, ldc, Wl + c * KC + wave * kq * KSTEP, ldw, kq);
            }
        }
#pragma unroll
        for (int bt = 0; bt < NB; ++bt)
#pragma unroll
            for (int r = 0; r < 4; ++r) Gl[(wave * NB * 16 + bt * 16 + fq * 4 + r) * 17 + fr] = acc[bt][r];
        __syncthreads();
        // (f) pointwise BPTT; publish dgates_t FIRST, signal, then write the fp32 copy for the weight-gradient GEMMs
        float dg[PP][4][2];
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            if (!ev[p]) continue;
            const int bl = eb[p], n = en[p], j = j0 + n;
            const bool m = t < lensl[bl];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float dh_rec = Gl[(0 * NB * 16 + bl) * 17 + n + q] + Gl[(1 * NB * 16 + bl) * 17 + n + q] +
                                     Gl[(2 * NB * 16 + bl) * 17 + n + q] + Gl[(3 * NB * 16 + bl) * 17 + n + q];
                const float ig = q ? sg[p][0].y : sg[p][0].x, fg = q ? sg[p][1].y : sg[p][1].x;
                const float gg = q ? sg[p][2].y : sg[p][2].x, og = q ? sg[p][3].y : sg[p][3].x;
                const float ct = q ? sc[p].y : sc[p].x, cp = q ? scp[p].y : scp[p].x;
                const float dh = (q ? sdy[p].y : sdy[p].x) + dh_rec;
                const float tc = ftanh(ct);
                const float dc = dh * og * (1.f - tc * tc) + dc_carry[p][q];
                const bool mq = m && (j + q < H);
                dg[p][0][q] = mq ? dc * gg * ig * (1.f - ig) : 0.f;
                dg[p][1][q] = mq ? dc * cp * fg * (1.f - fg) : 0.f;
                dg[p][2][q] = mq ? dc * ig * (1.f - gg * gg) : 0.f;
                dg[p][3][q] = mq ? dh * tc * og * (1.f - og) : 0.f;
                dc_carry[p][q] = mq ? dc * fg : 0.f;
            }
#pragma unroll
            for (int gi = 0; gi < 4; ++gi)
                st_pair_x(dgx + (((long)d * a.T + t) * B + b0 + bl) * K4 + gi * H + j, dg[p][gi][0], dg[p][gi][1], local);
        }
        block_signal(cnt, local, g, (unsigned)s + 1u);
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            if (!ev[p]) continue;
            const long ro = (long)t * B + b0 + eb[p];
            const int j = j0 + en[p];
#pragma unroll
            for (int gi = 0; gi < 4; ++gi)
                *(float2*)(dgf + ro * ND4H + d * K4 + gi * H + j) = make_float2(dg[p][gi][0], dg[p][gi][1]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward, K-split
// Same ownership as above, but the recurrent product is split over K instead of N: workgroup g multiplies ITS OWN
// dgates_t [Bs x 64] (4 gates x 16 units, just produced by its pointwise pass, still in LDS) by its 64 rows of W_hh
// and gets a partial dh_{t-1} for ALL H columns; the 16-column piece of consumer c goes to c's inbox, and c sums the
// G pieces it receives.  The exchange is a reduce-scatter of [Bs x H] (as large as the forward's all-gather of h)
// instead of an all-gather of dgates [Bs x 4H]: 7.7 KB instead of 30 KB pulled per workgroup and step at H = 320,
// and nothing has to be fetched before the MFMAs.  Partial sums travel as bf16 pairs (rows 2r, 2r+1 of one column)
// in bf16 mode, as f32 in f32 mode.  The inboxes are a ring of 4 steps (a producer can be at most one step ahead of
// the slowest reader of its group); hand-off protocol as in the forward kernel.
constexpr int KS_SLOTS = 4;
__host__ __device__ inline int ks_words_per_tile(int prec, int NB) { return NB * 16 * (prec == LAS_PREC_BF16 ? 8 : 16); }

template <int PREC, int NB, int MT>     // MT = tiles (consumers) per wave = ceil(G / 4)
__global__ __launch_bounds__(NT) void lstm_bwd_ks_kernel(LstmArgs a, const float* __restrict__ dy,
                                                         const float* __restrict__ gates, const float* __restrict__ cs,
                                                         const float* __restrict__ w_hh, const int32_t* __restrict__ lens,
                                                         unsigned* __restrict__ pex, float* __restrict__ dgf,
                                                         SyncWords* sync, int* status) {
    typedef typename CT<PREC>::T T;
    constexpr int VEC = CT<PREC>::VEC, KSTEP = CT<PREC>::KSTEP;
    constexpr int KO = 64, LDK = KO + VEC;                // own dgates: k = gate*16 + unit
    constexpr int WPR = PREC == LAS_PREC_BF16 ? 8 : 16;   // exchange words per batch row of a 16-column piece
    constexpr int WPT = NB * 16 * WPR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = a.H, B = a.B, ND = a.ND, K4 = 4 * H, G = a.G;
    const Role role = lstm_role(a);
    if (role.idle) return;
    const int d = role.d, g = role.g, bs = role.bs, j0 = g * 16;
    const int b0 = bs * a.Bs, Bl = min(a.Bs, B - b0);
    const bool wdirect = PREC == LAS_PREC_BF16 && a.wdirect;   // weight fragments fetched from global memory, no LDS slab
    T* Wl = (T*)smem;                                     // [G*16][LDK]  W_hh[my 64 gate rows][every column], k contiguous
    T* Dl = Wl + (wdirect ? 0 : (size_t)G * 16 * LDK);    // [NB*16][LDK] my dgates of this step
    unsigned* Pl = (unsigned*)(Dl + NB * 16 * LDK);       // [G][WPT] inbox of the previous step
    int* lensl = (int*)(Pl + (size_t)G * WPT);
    int* flag = lensl + NB * 16;

    if (!wdirect)
    for (int i = threadIdx.x; i < G * 16 * KO; i += NT) {
        const int col = i % (G * 16), k = i / (G * 16), gi = k >> 4, n = k & 15;
        float v = 0.f;
        if (col < H && j0 + n < H) v = w_hh[((long)d * K4 + gi * H + j0 + n) * H + col];
        Wl[col * LDK + k] = PREC == LAS_PREC_BF16 ? (T)f2bf(v) : (T)v;
    }
    for (int i = threadIdx.x; i < NB * 16 * LDK; i += NT) Dl[i] = (T)0;
    for (int i = threadIdx.x; i < NB * 16; i += NT) lensl[i] = i < Bl ? lens[b0 + i] : 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
    // The product is taken transposed (A = W rows = output columns, B = dgates rows = batch), so a lane ends up with
    // FOUR CONSECUTIVE output columns of one batch row: one 8-byte (bf16) / 16-byte (f32) store per tile.
    // bf16: the weight fragments of this wave's tiles stay in registers for all T steps.
    bf16x8 wfrag[PREC == LAS_PREC_BF16 ? MT : 1][2];
    if constexpr (PREC == LAS_PREC_BF16) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int c = min(wave + 4 * i, G - 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (!wdirect) { wfrag[i][ks] = *(const bf16x8*)((const bf16_t*)Wl + (c * 16 + fr) * LDK + ks * 32 + fq * 8); continue; }
                // element e <-> k = 32 ks + 8 fq + e = gate (2 ks + fq/2), unit 8 (fq & 1) + e; output column 16 c + fr
                // (H = 1024: the [H][64] slab is 147 KB of LDS; one strided read per element instead, once per launch)
                const int gi = 2 * ks + (fq >> 1), n0 = (fq & 1) * 8, col = c * 16 + fr;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    v[e] = w_hh[((long)d * K4 + gi * H + min(j0 + n0 + e, H - 1)) * H + min(col, H - 1)];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (col < H && j0 + n0 + e < H) ? v[e] : 0.f;
                const u32x4 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
                wfrag[i][ks] = __builtin_bit_cast(bf16x8, pk);
            }
        }
    }

    // pointwise elements of this thread: the pair (row eb, units en, en+1)  (NB <= 2: at most one pair per thread)
    const int e = threadIdx.x, eb = e >> 3, en = (e & 7) * 2, j = j0 + en;
    const bool ev = e < NB * 16 * 8 && eb < Bl && j < H;
    float dc_carry[2] = {0.f, 0.f};
    const int ND4H = ND * K4, NDH = ND * H;
    unsigned* cnt = &sync->cnt[(d * MAX_SLICES + bs) * CNT_STRIDE];
    const long slot_words = (long)ND * a.NS * G * G * WPT;
    auto inbox = [&](int slot, int consumer, int producer) -> unsigned* {
        return pex + slot * slot_words + ((((long)d * a.NS + bs) * G + consumer) * G + producer) * WPT;
    };
    // saved activations / incoming gradient of one step, requested a step ahead of their use
    float2 sg[4], sc, scp, sdy;
    auto load_inputs = [&](int s) {
        const int t = d == 0 ? a.T - 1 - s : s, tp = d == 0 ? t - 1 : t + 1;
        sc = scp = sdy = make_float2(0.f, 0.f);
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) sg[gi] = make_float2(0.f, 0.f);
        if (!ev || s >= a.T || t >= lensl[eb]) return;
        const int b = b0 + eb;
        const long ro = (long)t * B + b;
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) sg[gi] = *(const float2*)(gates + ro * ND4H + d * K4 + gi * H + j);
        sc = *(const float2*)(cs + ro * NDH + d * H + j);
        if (tp >= 0 && tp < lensl[eb]) scp = *(const float2*)(cs + ((long)tp * B + b) * NDH + d * H + j);
        bool ok;
        const long yo = y_offset(a, t, b, d, j, ok);
        if (ok) sdy = *(const float2*)(dy + yo);
    };
    load_inputs(0);
    const int gl = group_local(a, cnt, &sync->abort_, flag);
    if (gl < 0) { if (threadIdx.x == 0) *status = LAS_E_TIMEOUT; return; }
    const bool local = gl == 1;

    for (int s = 0; s < a.T; ++s) {
        const int t = d == 0 ? a.T - 1 - s : s;           // reverse of the forward processing order
        float dh_rec[2] = {0.f, 0.f};
        if (s > 0) {
            if (!block_wait(cnt, G, (unsigned)s, &sync->abort_, flag, local)) { if (threadIdx.x == 0) *status = LAS_E_TIMEOUT; return; }
            bool direct_sum = false;
            if constexpr (PREC == LAS_PREC_BF16 && MT <= 8) direct_sum = local;
            if (direct_sum) {
                // L2-local group: each pointwise thread reads its word of every piece straight from the XCD's L2 (a wave
                // instruction covers 256 contiguous bytes of one piece) -- no LDS staging, one barrier less
                if constexpr (PREC == LAS_PREC_BF16 && MT <= 8) {
                    constexpr int NP = MT * 4;
                    const unsigned* src = inbox((s - 1) & (KS_SLOTS - 1), g, 0);
                    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, G * WPT * 4, 0x00020000);
                    const int w0 = ev ? eb * WPR + en / 2 : 0x1ffffff0;
                    unsigned w[NP];
#pragma unroll
                    for (int p = 0; p < NP; ++p) w[p] = __builtin_amdgcn_raw_buffer_load_b32(rs, (min(p, G - 1) * WPT + w0) * 4, 0, 16);
                    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const unsigned v = p < G ? w[p] : 0u;
                        s0[p & 3] += __uint_as_float(v << 16);
                        s1[p & 3] += __uint_as_float(v & 0xffff0000u);
                    }
                    dh_rec[0] = (s0[0] + s0[1]) + (s0[2] + s0[3]);
                    dh_rec[1] = (s1[0] + s1[1]) + (s1[2] + s1[3]);
                }
            } else {
            // (only the Bl batch rows of each piece that carry data are pulled)
            pull_tile_sc1<unsigned, 4, (MT > 8 ? 8 : 4)>(inbox((s - 1) & (KS_SLOTS - 1), g, 0), G, (Bl * WPR + 3) & ~3, WPT, 0, Pl, WPT);
            __syncthreads();
            if (ev) {                                     // sum of the G pieces: every LDS read is issued before the first add
                const unsigned* pw = Pl + eb * WPR + (PREC == LAS_PREC_BF16 ? en / 2 : en);
                float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
                if (PREC == LAS_PREC_BF16 && MT == 5 && G > 10) {      // (measured: pays for 11..20 pieces only)
                    constexpr int NP = MT * 4;            // >= G
                    unsigned w[NP];
#pragma unroll
                    for (int p = 0; p < NP; ++p) w[p] = pw[min(p, G - 1) * WPT];
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const unsigned v = p < G ? w[p] : 0u;
                        s0[p & 3] += __uint_as_float(v << 16);
                        s1[p & 3] += __uint_as_float(v & 0xffff0000u);
                    }
                } else {
                    int p = 0;
                    for (; p + 3 < G; p += 4) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if constexpr (PREC == LAS_PREC_BF16) {
                                const unsigned w = pw[(p + u) * WPT];
                                s0[u] += __uint_as_float(w << 16);
                                s1[u] += __uint_as_float(w & 0xffff0000u);
                            } else {
                                const uint2 w = *(const uint2*)(pw + (p + u) * WPT);
                                s0[u] += __uint_as_float(w.x);
                                s1[u] += __uint_as_float(w.y);
                            }
                        }
                    }
                    for (; p < G; ++p) {
                        if constexpr (PREC == LAS_PREC_BF16) {
                            const unsigned w = pw[p * WPT];
                            s0[0] += __uint_as_float(w << 16);
                            s1[0] += __uint_as_float(w & 0xffff0000u);
                        } else {
                            const uint2 w = *(const uint2*)(pw + p * WPT);
                            s0[0] += __uint_as_float(w.x);
                            s1[0] += __uint_as_float(w.y);
                        }
                    }
                }
                dh_rec[0] = (s0[0] + s0[1]) + (s0[2] + s0[3]);
                dh_rec[1] = (s1[0] + s1[1]) + (s1[2] + s1[3]);
            }
            }
        }
        // pointwise BPTT -> my dgates of this step, into LDS for the product below
        float dg[4][2];
        {
            const bool m = ev && t < lensl[min(eb, NB * 16 - 1)];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float ig = q ? sg[0].y : sg[0].x, fg = q ? sg[1].y : sg[1].x;
                const float gg = q ? sg[2].y : sg[2].x, og = q ? sg[3].y : sg[3].x;
                const float ct = q ? sc.y : sc.x, cp = q ? scp.y : scp.x;
                const float dh = (q ? sdy.y : sdy.x) + dh_rec[q];
                const float tc = ftanh(ct);
                const float dc = dh * og * (1.f - tc * tc) + dc_carry[q];
                const bool mq = m && (j + q < H);
                dg[0][q] = mq ? dc * gg * ig * (1.f - ig) : 0.f;
                dg[1][q] = mq ? dc * cp * fg * (1.f - fg) : 0.f;
                dg[2][q] = mq ? dc * ig * (1.f - gg * gg) : 0.f;
                dg[3][q] = mq ? dh * tc * og * (1.f - og) : 0.f;
                dc_carry[q] = mq ? dc * fg : 0.f;
            }
            if (ev) {
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    if constexpr (PREC == LAS_PREC_BF16) *(unsigned*)(Dl + eb * LDK + gi * 16 + en) = pack_bf16x2(dg[gi][0], dg[gi][1]);
                    else *(float2*)(Dl + eb * LDK + gi * 16 + en) = make_float2(dg[gi][0], dg[gi][1]);
                }
            }
        }
        __syncthreads();
        if (s + 1 < a.T) {
            // partial dh_{t-1}[:, 16 c .. 16 c + 15] for every consumer c; wave w takes c = w, w+4, ...
            const int slot = s & (KS_SLOTS - 1);
            if constexpr (PREC == LAS_PREC_BF16) {
                bf16x8 dv[NB][2];
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        dv[bt][ks] = *(const bf16x8*)((const bf16_t*)Dl + (bt * 16 + fr) * LDK + ks * 32 + fq * 8);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int c = wave + 4 * i;
                    if (c >= G) break;
                    unsigned* dst = inbox(slot, c, g);
#pragma unroll
                    for (int bt = 0; bt < NB; ++bt) {
                        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag[i][0], dv[bt][0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag[i][1], dv[bt][1], acc, 0, 0, 0);
                        // lane: batch row bt*16 + fr, output columns 16c + 4 fq + {0..3}
                        const int row = bt * 16 + fr;
                        if (row < Bl) {
                            const unsigned long long v = (unsigned long long)pack_bf16x2(acc[0], acc[1]) |
                                                         ((unsigned long long)pack_bf16x2(acc[2], acc[3]) << 32);
                            if (local) *(unsigned long long*)(dst + row * WPR + fq * 2) = v;
                            else __hip_atomic_store((unsigned long long*)(dst + row * WPR + fq * 2), v, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            } else {
                for (int c = wave; c < G; c += 4) {
                    unsigned* dst = inbox(slot, c, g);
#pragma unroll
                    for (int bt = 0; bt < NB; ++bt) {
                        f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
                        mma_rows<PREC, 1>(acc, Wl + (size_t)c * 16 * LDK, LDK, Dl + bt * 16 * LDK, LDK, KO / KSTEP);
                        const int row = bt * 16 + fr;
                        if (row < Bl) {
                            const u32x4 v = {__float_as_uint(acc[0][0]), __float_as_uint(acc[0][1]), __float_as_uint(acc[0][2]),
                                             __float_as_uint(acc[0][3])};
                            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, WPT * 4, 0x00020000);
                            if (local) __builtin_amdgcn_raw_buffer_store_b128(v, rs, (row * WPR + fq * 4) * 4, 0, 0);
                            else __builtin_amdgcn_raw_buffer_store_b128(v, rs, (row * WPR + fq * 4) * 4, 0, 16);
                        }
                    }
                }
            }
            block_signal(cnt, local, g, (unsigned)s + 1u);
        }
        // fp32 copy for the weight-gradient GEMMs, then next step's inputs
        if (ev) {
            const long ro = (long)t * B + b0 + eb;
#pragma unroll
            for (int gi = 0; gi < 4; ++gi)
                *(float2*)(dgf + ro * ND4H + d * K4 + gi * H + j) = make_float2(dg[gi][0], dg[gi][1]);
        }
        load_inputs(s + 1);
    }
}

// ------------------------------------------------------------------------------------------------ backward, K-split, tagged granules
// lstm_bwd_ks_kernel's reduce-scatter with the hand-off of lstm_fwd_gr_kernel (bf16): a producer's 16-column piece for
// consumer c leaves as 16-byte granules {4 consecutive columns of one batch row as bf16, tag = step + 1, 0} -- what a lane
// holds after the transposed MFMA -- into c's inbox [producer][row][4]; the consumer sweeps its G * Bl * 4 granules until
// every tag matches, spreads them as f32 into LDS [row][column][producer], and after the barrier the cell lane of
// (row, unit) adds its G partial sums with 16-byte LDS reads.  No flag, no drain; two barriers per step (sums complete,
// dgates tile complete).  The compute waves touch global memory only for the hand-off: a fifth wave loads gates / c /
// c_prev / dy three steps ahead into LDS and stores d gates (the weight-gradient GEMMs' operand) from LDS.
constexpr int BG_CLD = 16 * 4 + 4;                       // floats per batch row of the {c, c_prev, dy, -} tile
template <int NB, int MT, int SWM = 4>
__global__ __launch_bounds__(NT + 64) void lstm_bwd_gr_kernel(LstmArgs a, const float* __restrict__ dy,
                                                         const float* __restrict__ gates, const float* __restrict__ cs,
                                                         const float* __restrict__ w_hh, const int32_t* __restrict__ lens,
                                                         u32x4* __restrict__ ring, float* __restrict__ dgf,
                                                         SyncWords* sync, int* status) {
    constexpr int LDK = 64 + 16;                          // own dgates tile: k = gate * 16 + unit (row stride 160 B: conflict-free b128 fragments)
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = a.H, B = a.B, ND = a.ND, K4 = 4 * H, G = a.G;
    const Role role = lstm_role(a);
    if (role.idle) return;
    const int d = role.d, g = role.g, bs = role.bs, j0 = g * 16;
    const int b0 = bs * a.Bs, Bl = min(a.Bs, B - b0);
    constexpr int PS = MT * 4, RLD = 16 * PS + 4;         // partial sums [row][column][producer >= G: stays 0], rows padded (banks)
    bf16_t* Dl = (bf16_t*)smem;                           // [NB*16][LDK]        my dgates of this step (MFMA B operand)
    float* Red = (float*)(Dl + NB * 16 * LDK);            // [NB*16][RLD]
    float* Gi = Red + NB * 16 * RLD;                      // [2][NB*16][GR_XLD]  saved gates of the step   (I/O wave -> compute)
    float* Ci = Gi + 2 * NB * 16 * GR_XLD;                // [2][NB*16][BG_CLD]  {c, c_prev, dy, -}
    float* Do = Ci + 2 * NB * 16 * BG_CLD;                // [2][NB*16][GR_XLD]  d gates in f32            (compute -> I/O wave)
    int* lensl = (int*)(Do + 2 * NB * 16 * GR_XLD);
    int* flag = lensl + NB * 16;
    int* tab = flag + 4;                                  // SWM > 4: [NT][SWM][2] sweep lists
    const bool io = threadIdx.x >= NT;
    if (!io) {
        for (int i = threadIdx.x; i < NB * 16 * LDK / 2; i += NT) ((unsigned*)Dl)[i] = 0u;
        for (int i = threadIdx.x; i < NB * 16 * RLD; i += NT) Red[i] = 0.f;
        for (int i = threadIdx.x; i < NB * 16; i += NT) lensl[i] = i < Bl ? lens[b0 + i] : 0;
    }
    __syncthreads();
    unsigned* cnt = &sync->cnt[(d * MAX_SLICES + bs) * CNT_STRIDE];
    const int gl = group_local(a, cnt, &sync->abort_, flag);
    if (gl < 0) { if (threadIdx.x == 0) *status = LAS_E_TIMEOUT; return; }
    const bool local = gl == 1;
    const int ND4H = ND * K4, NDH = ND * H;
    const int lane = threadIdx.x & 63;
    auto tstep = [&](int s) { const int sc = min(s, a.T - 1); return d == 0 ? a.T - 1 - sc : sc; };   // reverse of the forward order
    if (io) {
        // ---- I/O wave (see lstm_fwd_gr_kernel): item = 4 consecutive units of one (batch row[, gate])
        constexpr int NQ = NB * 4;
        const long nrow = (long)a.T * B;
        __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)gates, 0, (int)(nrow * ND4H * 4), 0x00020000);
        __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)cs, 0, (int)(nrow * NDH * 4), 0x00020000);
        __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)((long)a.T_out * B * a.F_out * 4), 0x00020000);
        __amdgpu_buffer_rsrc_t ro_ = __builtin_amdgcn_make_buffer_rsrc((void*)dgf, 0, (int)(nrow * ND4H * 4), 0x00020000);
        __amdgpu_buffer_rsrc_t ro16 = __builtin_amdgcn_make_buffer_rsrc(a.tw, 0, a.tw ? (int)(nrow * ND4H * 2) : 0, 0x00020000);
        struct In { u32x4 g4[NQ], c[NB], cp[NB], y[NB]; };
        auto xload = [&](int s, In& x) {
            const int t = tstep(s), tp = d == 0 ? t - 1 : t + 1;
            const bool sv = s < a.T;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int it = lane + 64 * q, row = it >> 4, gi = (it >> 2) & 3, u0 = (it & 3) * 4;
                const bool ok = sv && row < Bl && j0 + u0 < H && t < lensl[min(row, NB * 16 - 1)];
                x.g4[q] = __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? (int)((((long)t * B + b0 + row) * ND4H + d * K4 + gi * H + j0 + u0) * 4) : OOB, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int it = lane + 64 * q, row = it >> 2, u0 = (it & 3) * 4, len = lensl[min(row, NB * 16 - 1)], b = b0 + row;
                const bool ok = sv && row < Bl && j0 + u0 < H && t < len;
                x.c[q] = __builtin_amdgcn_raw_buffer_load_b128(rc, ok ? (int)((((long)t * B + b) * NDH + d * H + j0 + u0) * 4) : OOB, 0, 0);
                x.cp[q] = __builtin_amdgcn_raw_buffer_load_b128(rc, (ok && tp >= 0 && tp < len) ? (int)((((long)tp * B + b) * NDH + d * H + j0 + u0) * 4) : OOB, 0, 0);
                bool yok = false;
                const long yo = y_offset(a, t, min(b, B - 1), d, j0 + u0, yok);
                x.y[q] = __builtin_amdgcn_raw_buffer_load_b128(ry, (ok && yok) ? (int)(yo * 4) : OOB, 0, 0);
            }
        };
        auto xstore = [&](int s, const In& x) {
            unsigned* gi_ = (unsigned*)Gi + (s & 1) * NB * 16 * GR_XLD;
            unsigned* ci_ = (unsigned*)Ci + (s & 1) * NB * 16 * BG_CLD;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int it = lane + 64 * q, row = it >> 4, gi = (it >> 2) & 3, u0 = (it & 3) * 4;
                unsigned* p = gi_ + row * GR_XLD + u0 * 4 + gi;
                p[0] = x.g4[q][0]; p[4] = x.g4[q][1]; p[8] = x.g4[q][2]; p[12] = x.g4[q][3];
            }
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int it = lane + 64 * q, row = it >> 2, u0 = (it & 3) * 4;
                unsigned* p = ci_ + row * BG_CLD + u0 * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) { p[4 * e + 0] = x.c[q][e]; p[4 * e + 1] = x.cp[q][e]; p[4 * e + 2] = x.y[q][e]; }
            }
        };
        auto sflush = [&](int s, bool valid) {           // d gates of step s: LDS -> global
            const int t = tstep(s);
            const unsigned* dg = (const unsigned*)Do + (s & 1) * NB * 16 * GR_XLD;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int it = lane + 64 * q, row = it >> 4, gi = (it >> 2) & 3, u0 = (it & 3) * 4;
                const unsigned* p = dg + row * GR_XLD + u0 * 4 + gi;
                const u32x4 v = {p[0], p[4], p[8], p[12]};
                const int off = (valid && row < Bl && j0 + u0 < H) ? (int)((((long)t * B + b0 + row) * ND4H + d * K4 + gi * H + j0 + u0) * 4) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(v, ro_, off, 0, 0);
                const u32x2 v16 = {pack_bf16x2(__uint_as_float(v[0]), __uint_as_float(v[1])), pack_bf16x2(__uint_as_float(v[2]), __uint_as_float(v[3]))};
                __builtin_amdgcn_raw_buffer_store_b64(v16, ro16, off == OOB ? OOB : off >> 1, 0, 0);      // the bf16 twin (empty resource: none)
            }
        };
        In x0, x1, x2;                                   // three sets: a value is used three phases after its request
        xload(0, x0); xload(1, x1); xload(2, x2);
        xstore(0, x0); xload(3, x0);
        xstore(1, x1); xload(4, x1);
        __syncthreads();                                 // I: inputs of steps 0 and 1 in place
        // step s: A(s); B(s); store step s's d gates, hand over step s+2's inputs (their buffer was last read before A(s),
        // the compute waves want it after B(s+1): nothing of this wave's work sits between two barriers of one step),
        // request step s+5's
        for (int s = 0;; s += 3) {
            __syncthreads();
            __syncthreads();
            sflush(s, true); xstore(s + 2, x2); xload(s + 5, x2);
            if (s + 1 >= a.T) break;
            __syncthreads();
            __syncthreads();
            sflush(s + 1, true); xstore(s + 3, x0); xload(s + 6, x0);
            if (s + 2 >= a.T) break;
            __syncthreads();
            __syncthreads();
            sflush(s + 2, true); xstore(s + 4, x1); xload(s + 7, x1);
            if (s + 3 >= a.T) break;
        }
        return;
    }
    const int wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
    // A operand (transposed product): rows = the 16 output columns of consumer c, k = my 64 gate columns; fragments of the
    // wave's consumers c = wave, wave + 4, ... stay in registers for all T steps.  Element e of k-step ks <-> k = 32 ks +
    // 8 fq + e = gate 2 ks + fq / 2, unit 8 (fq & 1) + e.
    bf16x8 wfrag[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int c = min(wave + 4 * i, G - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int gi = 2 * ks + (fq >> 1), n0 = (fq & 1) * 8, col = c * 16 + fr;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = w_hh[((long)d * K4 + gi * H + min(j0 + n0 + e, H - 1)) * H + min(col, H - 1)];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (col < H && j0 + n0 + e < H) ? v[e] : 0.f;
            const u32x4 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
            wfrag[i][ks] = __builtin_bit_cast(bf16x8, pk);
        }
    }
    // my cell elements: unit ul of the workgroup, batch rows bt * 16 + fr
    const int ul = 4 * wave + fq;
    const bool evu = j0 + ul < H;
    float dc_carry[NB];
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) dc_carry[bt] = 0.f;
    // inboxes: ring[slot][d][bs][consumer][producer][Bs][4] granules
    const int RW = a.Bs * 4;
    const long slot_stride = (long)ND * a.NS * G * G * RW;
    u32x4* ringg = ring + ((long)d * a.NS + bs) * G * G * RW;
    // my share of the sweep (see lstm_fwd_gr_kernel): granule i = (producer p, row, column group cg)
    constexpr int SWMAX = SWM;
    constexpr bool TAB = SWM > 4;
    const int total = G * Bl * 4, Q = (total + 3) / 4, nsw = (Q + 63) / 64;
    int g_off[TAB ? 1 : SWMAX], g_dst[TAB ? 1 : SWMAX];
    int* tabw = tab + (int)threadIdx.x * SWMAX * 2;
#pragma unroll
    for (int u = 0; u < SWMAX; ++u) {
        const int idx = lane + 64 * u, i = wave * Q + idx;
        const bool ok = idx < Q && i < total;
        const int p = ok ? i / (Bl * 4) : 0, rem = ok ? i - p * (Bl * 4) : 0, row = rem >> 2, cg = rem & 3;
        const int o_ = ok ? ((g * G + p) * RW + rem) * 16 : OOB;
        const int d_ = ok ? row * RLD + cg * 4 * PS + p : -1;
        if constexpr (TAB) { tabw[2 * u] = o_; tabw[2 * u + 1] = d_; }
        else { g_off[u] = o_; g_dst[u] = d_; }
    }

    // The part of the cell backward that does not depend on dh_{rec} is taken BEFORE the sweep, from inputs the I/O wave
    // delivered a barrier earlier: dc = dh * ka + carry, d gates = dc * {k0, k1, k2}, dh * k3, carry' = dc * kf.
    struct Pre { float dy, ka, k0, k1, k2, k3, kf; bool mq; };
    Pre pre[NB];
    auto precompute = [&](int s) {
        const int t = d == 0 ? a.T - 1 - s : s;
        const float* gi_ = Gi + (s & 1) * NB * 16 * GR_XLD;
        const float* ci_ = Ci + (s & 1) * NB * 16 * BG_CLD;
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) {
            const int row = bt * 16 + fr;
            const f32x4 g4 = *(const f32x4*)(gi_ + row * GR_XLD + ul * 4);
            const f32x4 c4 = *(const f32x4*)(ci_ + row * BG_CLD + ul * 4);
            const float ig = g4[0], fg = g4[1], gg = g4[2], og = g4[3], tc = ftanh(c4[0]);
            pre[bt].mq = evu && row < Bl && t < lensl[row];
            pre[bt].dy = c4[2];
            pre[bt].ka = og * (1.f - tc * tc);
            pre[bt].k0 = gg * ig * (1.f - ig);
            pre[bt].k1 = c4[1] * fg * (1.f - fg);
            pre[bt].k2 = ig * (1.f - gg * gg);
            pre[bt].k3 = tc * og * (1.f - og);
            pre[bt].kf = fg;
        }
    };
    __syncthreads();                                     // I: inputs of steps 0 and 1 in place
    precompute(0);
    GR_ST_DECL;
    for (int s = 0; s < a.T; ++s) {
        if (s > 0) {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(ringg + ((s - 1) & (KS_SLOTS - 1)) * slot_stride), 0,
                                                                          G * G * RW * 16, 0x00020000);
            auto sweep = [&](auto swc) -> bool {
                constexpr int SW = decltype(swc)::value;
                u32x4 v[SW];
                unsigned got[TAB ? 1 : SW][2];
                int off[SW];
                for (int dl = 0; dl < (a.pdelay & 255); ++dl) __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int u = 0; u < SW; ++u) {
                    if constexpr (TAB) off[u] = tabw[2 * u]; else { off[u] = g_off[u]; got[u][0] = got[u][1] = 0u; }
                    v[u] = gr_poll(rs, off[u]);
                }
                precompute(s);                           // (its inputs were delivered before B(s-1)): under the first poll's round trip
                unsigned spins = 0;
                while (true) {
                    bool need = false;
#pragma unroll
                    for (int u = 0; u < SW; ++u) {
                        const bool hit = off[u] != OOB && v[u][2] == (unsigned)s;
                        off[u] = hit ? OOB : off[u];
                        if constexpr (TAB) {
                            if (hit) {
                                float* dst = Red + tabw[2 * u + 1];
                                dst[0] = __uint_as_float(v[u][0] << 16); dst[PS] = __uint_as_float(v[u][0] & 0xffff0000u);
                                dst[2 * PS] = __uint_as_float(v[u][1] << 16); dst[3 * PS] = __uint_as_float(v[u][1] & 0xffff0000u);
                            }
                        } else {
                            got[u][0] = hit ? v[u][0] : got[u][0];
                            got[u][1] = hit ? v[u][1] : got[u][1];
                        }
                        need = need || off[u] != OOB;
                    }
#ifdef LAS_PK_STAMPS
                    ++gst[4];
#endif
                    if (__builtin_amdgcn_ballot_w64(need) == 0ull) break;
#pragma unroll
                    for (int u = 0; u < SW; ++u) v[u] = gr_poll(rs, off[u]);
                    if ((++spins & 255u) == 0) {
                        if (__hip_atomic_load(&sync->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || spins > SPIN_LIMIT) {
                            __hip_atomic_store(&sync->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            *status = LAS_E_TIMEOUT;
                            return false;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < SW; ++u) {
                    if constexpr (TAB) continue;
                    const int d2 = g_dst[u];
                    if (d2 < 0) continue;
                    float* dst = Red + d2;
                    dst[0] = __uint_as_float(got[u][0] << 16); dst[PS] = __uint_as_float(got[u][0] & 0xffff0000u);
                    dst[2 * PS] = __uint_as_float(got[u][1] << 16); dst[3 * PS] = __uint_as_float(got[u][1] & 0xffff0000u);
                }
                return true;
            };
            bool ok_;
            if constexpr (SWM > 4) ok_ = nsw <= 6 ? sweep(std::integral_constant<int, 6>{})
                                       : nsw <= 9 ? sweep(std::integral_constant<int, 9>{}) : sweep(std::integral_constant<int, SWM>{});
            else ok_ = nsw <= 2 ? sweep(std::integral_constant<int, 2>{})
                     : nsw == 3 ? sweep(std::integral_constant<int, 3>{}) : sweep(std::integral_constant<int, 4>{});
            if (!ok_) return;
        }
        GR_ST(0);
        __syncthreads();                                 // A(s): partial sums complete, inputs of step s in Gi / Ci
        GR_ST(1);
        // ---- the rest of the pointwise BPTT of my elements -> my dgates of this step
        float* do_ = Do + (s & 1) * NB * 16 * GR_XLD;
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) {
            const int row = bt * 16 + fr;
            float dh_rec = 0.f;
            if (s > 0) {
                f32x4 acc4 = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float* rp = Red + row * RLD + ul * PS;
                f32x4 part[MT];                              // every read is issued before the first add
#pragma unroll
                for (int p = 0; p < MT; ++p) part[p] = *(const f32x4*)(rp + 4 * p);
#pragma unroll
                for (int p = 0; p < MT; ++p) acc4 += part[p];
                dh_rec = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
            }
            const bool mq = pre[bt].mq;
            const float dh = pre[bt].dy + dh_rec;
            const float dc = dh * pre[bt].ka + dc_carry[bt];
            const f32x4 dg = {mq ? dc * pre[bt].k0 : 0.f, mq ? dc * pre[bt].k1 : 0.f, mq ? dc * pre[bt].k2 : 0.f, mq ? dh * pre[bt].k3 : 0.f};
            dc_carry[bt] = mq ? dc * pre[bt].kf : 0.f;
            *(f32x4*)(do_ + row * GR_XLD + ul * 4) = dg;
            const unsigned lo = pack_bf16x2(dg[0], dg[1]), hi = pack_bf16x2(dg[2], dg[3]);
            bf16_t* dl = Dl + row * LDK + ul;
            dl[0] = (bf16_t)(lo & 0xffffu); dl[16] = (bf16_t)(lo >> 16); dl[32] = (bf16_t)(hi & 0xffffu); dl[48] = (bf16_t)(hi >> 16);
        }
        GR_ST(2);
        __syncthreads();                                 // B(s): dgates tile complete, d gates of step s in Do
        GR_ST(5);
        if (s + 1 < a.T) {
            // partial dh_{t-1}[:, 16 c .. 16 c + 15] for every consumer c; wave w takes c = w, w + 4, ...
            __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc((void*)(ringg + (s & (KS_SLOTS - 1)) * slot_stride), 0,
                                                                          G * G * RW * 16, 0x00020000);
            bf16x8 dv[NB][2];
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) dv[bt][ks] = *(const bf16x8*)(Dl + (bt * 16 + fr) * LDK + ks * 32 + fq * 8);
            f32x4 acc[MT][NB];                           // all products first (independent accumulators), then the stores
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) {
                    acc[i][bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag[i][0], dv[bt][0], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    acc[i][bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag[i][1], dv[bt][1], acc[i][bt], 0, 0, 0);
                }
            // lane: batch row bt * 16 + fr, output columns 16 c + 4 fq + {0..3}
            auto put = [&](auto auxc) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int c = wave + 4 * i;
#pragma unroll
                    for (int bt = 0; bt < NB; ++bt) {
                        const int row = bt * 16 + fr;
                        const u32x4 gr = {pack_bf16x2(acc[i][bt][0], acc[i][bt][1]), pack_bf16x2(acc[i][bt][2], acc[i][bt][3]), (unsigned)s + 1u, 0u};
                        const int woff = (c < G && row < Bl) ? ((c * G + g) * RW + row * 4 + fq) * 16 : OOB;
                        __builtin_amdgcn_raw_buffer_store_b128(gr, ws, woff, 0, decltype(auxc)::value);
                    }
                }
            };
            if (local) put(std::integral_constant<int, 0>{}); else put(std::integral_constant<int, 16>{});
        }
        GR_ST(3);
    }
    GR_ST_PRINT(a.T);
}
#include "lstm_x32.h"

size_t bwd_gr_lds(int H, int NB) {
    const int G = (H + 15) / 16, mt = (G + 3) / 4, PS = mt <= 5 ? 20 : mt <= 8 ? 32 : 64, RLD = 16 * PS + 4;
    return (size_t)NB * 16 * (64 + 16) * 2 + sizeof(float) * NB * 16 * (RLD + 2 * (2 * GR_XLD + BG_CLD)) + sizeof(int) * (NB * 16 + 4) +
           (mt > 8 ? sizeof(int) * NT * 12 * 2 : 0);
}
size_t bwd_gr_ring_bytes(const LstmArgs& a) { return (size_t)16 * KS_SLOTS * a.ND * a.NS * a.G * a.G * a.Bs * 4; }

size_t bwd_ks_lds(int prec, int H, int NB, bool wdirect) {
    const int sz = prec == LAS_PREC_BF16 ? 2 : 4, vec = prec == LAS_PREC_BF16 ? 8 : 4, G = (H + 15) / 16;
    return (size_t)((wdirect ? 0 : G * 16) + NB * 16) * (64 + vec) * sz + sizeof(unsigned) * (size_t)G * ks_words_per_tile(prec, NB) +
           sizeof(int) * (NB * 16 + 4);
}
size_t bwd_ks_ring_bytes(int prec, int H, int ND, int NS, int NB) {
    const size_t G = (H + 15) / 16;
    return sizeof(unsigned) * KS_SLOTS * ND * NS * G * G * ks_words_per_tile(prec, NB);
}

size_t fwd_lds(int prec, int H, int NB, bool wreg) {      // wreg: register-resident weights, no LDS slab
    const int sz = prec == LAS_PREC_BF16 ? 2 : 4, vec = prec == LAS_PREC_BF16 ? 8 : 4, ks = prec == LAS_PREC_BF16 ? 32 : 16;
    const int Kp = (H + ks - 1) / ks * ks, ld = Kp + vec;
    return (size_t)((wreg ? 0 : 4 * 16) + NB * 16) * ld * sz + sizeof(float) * 4 * NB * 16 * 17 + sizeof(int) * (NB * 16 + 4);
}
int bwd_k4p(int prec, int H) {
    const int q = 4 * (prec == LAS_PREC_BF16 ? 32 : 16);
    return (4 * H + q - 1) / q * q;
}
size_t bwd_lds(int prec, int H, int NB, int NC) {
    const int sz = prec == LAS_PREC_BF16 ? 2 : 4, vec = prec == LAS_PREC_BF16 ? 8 : 4;
    const int K4p = bwd_k4p(prec, H), KC = K4p / NC;
    return (size_t)16 * (K4p + vec) * sz + (size_t)NB * 16 * (KC + vec) * sz + sizeof(float) * 4 * NB * 16 * 17 +
           sizeof(int) * (NB * 16 + 4);
}
constexpr size_t LDS_CAP = 160 * 1024;


int check_common(int T, int B, int H, int ND, int sr) {
    if (T <= 0 || B <= 0 || H <= 0 || (ND != 1 && ND != 2) || sr < 1) return LAS_E_BADARG;
    if (H % 2 != 0) return LAS_E_UNSUPPORTED;          // pair stores / float2 accesses
    return LAS_OK;
}

void fill_args(LstmArgs& a, int T, int B, int H, int ND, int U, int sr, int concat, int force_ns = 0) {
    a.T = T; a.B = B; a.H = H; a.ND = ND; a.U = U; a.G = (H + U - 1) / U;
    // batch slices: independent sub-recurrences of <= 16 rows each, as many as the chip has room for
    int ns = (B + 11) / 12;
    if (ns < 8 / ND) {
        // finer slices (>= 6 rows) as long as every (direction, slice) group still gets an XCD of its own: at C2 four
        // slices of 6 rows on all eight XCDs instead of two of 12 on four (fwd 1.91 -> 1.84, bwd 2.42 -> 2.34 us per
        // step, C2 step -0.4 ms)
        const int ns6 = (B + 5) / 6;
        ns = ns6 < 8 / ND ? (ns6 > ns ? ns6 : ns) : 8 / ND;
    }
    while (ns > 1 && ((long)ND * a.G * ns > las_cu_count() || ns > MAX_SLICES)) --ns;
    if (force_ns > 0) ns = force_ns;
    a.Bs = (B + ns - 1) / ns;
    a.NS = (B + a.Bs - 1) / a.Bs;
    a.sr = sr; a.concat = concat;
    if (sr == 1) { a.T_out = T; a.F_out = ND * H; }
    else if (concat) { a.T_out = T / sr; a.F_out = sr * ND * H; }
    else { a.T_out = (T + sr - 1) / sr; a.F_out = ND * H; }
    a.y_is_hf = 0; a.wdirect = 0; a.fxI = 0; a.w_ih = nullptr;
    // XCD-grouped launch when every XCD (32 CUs, one workgroup per CU) can hold the groups dealt to it
    const bool no_xl = las_fallback("LAS_LSTM_NO_XL") != nullptr;
    const int groups = ND * a.NS, gpl = (groups + 7) / 8;
    a.xl = (!no_xl && a.G * gpl <= las_cu_count() / 8) ? 1 : 0;
    const bool no_nt = LAS_AB_KNOB("LAS_LSTM_NO_NT") != nullptr;
    a.nt = no_nt ? 0 : 1;
    const bool no_pf = LAS_AB_KNOB("LAS_LSTM_NO_PF") != nullptr;
    a.pf = no_pf ? 0 : 1;
    const char* pd = LAS_AB_KNOB("LAS_LSTM_POLL_DELAY");
    a.pdelay = pd ? atoi(pd) : 0;
}
int lstm_grid(const LstmArgs& a) { return a.xl ? 8 * a.G * ((a.ND * a.NS + 7) / 8) : a.ND * a.G * a.NS; }

// The 32-unit geometry (lstm_x32.h): bf16, 512 < H <= 1024, slices of <= 16 rows, every group on an XCD of its own.
inline int host_gr_chunks(int rows) { return (rows >> 4) * 3 + ((rows & 15) + 5) / 6; }
bool use_x32(int prec, int T, int B, int H, int ND, int sr, int concat, LstmArgs& a) {
    if (prec != LAS_PREC_BF16 || H <= 512 || H > 1024 || (H & 3)) return false;
    if (las_fallback("LAS_LSTM_NO_GR") || las_fallback("LAS_LSTM_NO_XL") || las_fallback("LAS_LSTM_NO_X32")) return false;
    // slices of >= 6 rows while every (direction, slice) group still gets an XCD of its own: the sweeps shrink with the
    // slice (C5, B = 24: four slices of 6 instead of two of 12: forward 3.7 -> 3.1, backward 6.1 -> 4.8 us per step, C5 step
    // 55.3 -> 48.6 ms), at the price of occupying all eight XCDs
    int ns = (B + 5) / 6;
    if (ns > 8 / ND) ns = 8 / ND;
    if (ns < (B + 15) / 16) ns = (B + 15) / 16;
    if (const char* e = LAS_AB_KNOB("LAS_LSTM_X32_NS")) ns = atoi(e);      // (diagnostic build: force the slice count)
    fill_args(a, T, B, H, ND, 32, sr, concat, ns);
    if (!a.xl || a.Bs > 16 || a.NS > MAX_SLICES || ND * a.NS > 8) return false;
    if ((long)T * B * ND * 4 * H * 4 >= (1l << 31)) return false;
    if ((long)H * host_gr_chunks(a.Bs) > 8 * 256 || (long)a.G * a.Bs * 8 > X32_SWB * 256) return false;
    return true;
}

template <int PREC, int NB, int KS>
int launch_fwd(const LstmArgs& a, size_t lds, hipStream_t st, const float* xproj, const float* b_ih, const float* b_hh,
               const float* w_hh, const int32_t* lens, float* y, float* hf, void* hx, float* gates, float* cs,
               SyncWords* sync, int* status) {
    auto k = lstm_fwd_kernel<PREC, NB, KS>;
    LAS_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // (KS > 16: the weight fragments alone are 128 registers per lane -- one wave per SIMD, no prefetch wave)
    hipLaunchKernelGGL(k, dim3(lstm_grid(a)), dim3(a.pf && KS <= 16 ? NT + 64 : NT), lds, st, a, xproj, b_ih, b_hh, w_hh, lens, y, hf,
                       (typename CT<PREC>::T*)hx, gates, cs, sync, status);
    LAS_LAUNCH_OK();
    return LAS_OK;
}
size_t fwd_gr_lds(int H, int NB, int swm = 4) {
    const int Kp = (H + 31) / 32 * 32, ld = Kp + 16;
    return (size_t)2 * NB * 16 * ld * 2 + sizeof(float) * 2 * NB * 16 * (2 * GR_XLD + GR_HLD) + sizeof(int) * (NB * 16 + 4) +
           (swm > 4 ? sizeof(int) * NT * swm * 2 : 0);
}
size_t fwd_gr_ring_bytes(const LstmArgs& a) {
    return (size_t)16 * a.ND * HX_SLOTS * a.NS * a.H * ((a.Bs >> 4) * 3 + ((a.Bs & 15) + 5) / 6);
}
template <int NB, int KS, int SWM = 4, bool FX = false>
int launch_fwd_gr(const LstmArgs& a, size_t lds, hipStream_t st, const float* xproj, const float* b_ih, const float* b_hh,
                  const float* w_hh, const int32_t* lens, float* y, float* hf, void* hx, float* gates, float* cs,
                  SyncWords* sync, int* status) {
    auto k = lstm_fwd_gr_kernel<NB, KS, SWM, FX>;
    LAS_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if ((char*)hx != (char*)sync + sizeof(SyncWords))                    // (else: zeroed together with the sync words by the caller below)
        LAS_HIP(hipMemsetAsync(hx, 0, fwd_gr_ring_bytes(a), st));      // tags of earlier launches must not match
    hipLaunchKernelGGL(k, dim3(lstm_grid(a)), dim3(NT + 64), lds, st, a, xproj, b_ih, b_hh, w_hh, lens, y, hf,
                       (u32x4*)hx, gates, cs, sync, status);
    LAS_LAUNCH_OK();
    return LAS_OK;
}
template <int PREC, int NB, int KS>
int launch_bwd(const LstmArgs& a, int NC, int K4p, size_t lds, hipStream_t st, const float* dy, const float* gates,
               const float* cs, const float* w_hh, const int32_t* lens, void* dgx, float* dgf, SyncWords* sync,
               int* status) {
    auto k = lstm_bwd_kernel<PREC, NB, KS>;
    LAS_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(lstm_grid(a)), dim3(NT), lds, st, a, NC, K4p, dy, gates, cs, w_hh, lens,
                       (typename CT<PREC>::T*)dgx, dgf, sync, status);
    LAS_LAUNCH_OK();
    return LAS_OK;
}


// Which backward kernel a shape gets, its LDS request and the size of its exchange workspace (`dgx`).
struct BwdPlan { bool ks, wdirect, gr; int NB, NC, K4p; size_t lds, ws; };
int bwd_plan(int prec, int T, int B, int H, int ND, const LstmArgs& a, BwdPlan& p) {
    p.NB = las_pick_nb(a.Bs);
    if (p.NB == 0 || ND * ((H + 15) / 16) > las_cu_count()) return LAS_E_UNSUPPORTED;
    // K-split exchange (reduce-scatter of partial dh) whenever it fits: the [H][64] weight slab in LDS, or (bf16) the
    // weight fragments fetched from global memory into registers (ceil(G / 4) <= 16 tiles per wave: H <= 1024)
    p.ks = false; p.wdirect = false;
    if (p.NB <= 2 && !LAS_AB_KNOB("LAS_LSTM_BWD_GATHER")) {
        const int mt = (a.G + 3) / 4;
        if (bwd_ks_lds(prec, H, p.NB, false) <= LDS_CAP && (prec != LAS_PREC_BF16 || mt <= 16)) p.ks = true;
        else if (prec == LAS_PREC_BF16 && mt <= 16 && bwd_ks_lds(prec, H, p.NB, true) <= LDS_CAP) p.ks = p.wdirect = true;
    }
    // tagged-granule hand-off (lstm_bwd_gr_kernel): bf16, <= 8 consumers per wave, <= 1 024 granules per inbox and step
    // (more than 32 producers -- 512 < H <= 1024 --: one batch tile per slice, up to 12 granules per lane and step, lists in LDS)
    const int mt_ = (a.G + 3) / 4;
    p.gr = p.ks && prec == LAS_PREC_BF16 && (mt_ <= 8 || (mt_ <= 16 && p.NB == 1)) && H % 4 == 0 &&
           (long)a.G * a.Bs * 4 <= (mt_ > 8 ? 3072 : 1024) &&
           (long)T * B * ND * 4 * H * 4 < (1l << 31) && bwd_gr_lds(H, p.NB) <= LDS_CAP && !las_fallback("LAS_LSTM_NO_GR");
    if (p.gr) {
        p.wdirect = false;
        p.lds = bwd_gr_lds(H, p.NB);
        p.ws = bwd_gr_ring_bytes(a);
        p.NC = 1; p.K4p = 4 * H;
    } else if (p.ks) {
        p.lds = bwd_ks_lds(prec, H, p.NB, p.wdirect);
        p.ws = bwd_ks_ring_bytes(prec, H, ND, a.NS, p.NB);
        p.NC = 1; p.K4p = 4 * H;
    } else {
        const int ks = prec == LAS_PREC_BF16 ? 32 : 16;
        p.K4p = bwd_k4p(prec, H);
        p.NC = 1;                                              // chunks must keep whole k-steps per wave and be unpadded if >1
        while (p.NC <= 8 && (bwd_lds(prec, H, p.NB, p.NC) > LDS_CAP || (p.K4p / p.NC) % (4 * ks) != 0 || (p.NC > 1 && p.K4p != 4 * H))) p.NC *= 2;
        if (p.NC > 8) return LAS_E_UNSUPPORTED;
        p.lds = bwd_lds(prec, H, p.NB, p.NC);
        p.ws = (size_t)ND * T * B * 4 * H * (prec == LAS_PREC_BF16 ? 2 : 4);    // all-gather variant: dgates copy
    }
    if (p.lds < MIN_LDS) p.lds = MIN_LDS;
    return LAS_OK;
}

}  // namespace

extern "C" void las_lstm_out_shape(int T, int H, int ND, int sr, int concat, int* T_out, int* F_out) {
    LstmArgs a;
    fill_args(a, T, 1, H, ND, 16, sr, concat);
    *T_out = a.T_out; *F_out = a.F_out;
}

extern "C" size_t las_lstm_sync_bytes(void) { return sizeof(SyncWords); }

extern "C" size_t las_lstm_hx_bytes(int prec, int T, int B, int H, int ND) {
    LstmArgs a;
    if (check_common(T, B, H, ND, 1)) return 0;
    if (use_x32(prec, T, B, H, ND, 1, 0, a)) return las_align(fwd_gr_ring_bytes(a));
    fill_args(a, T, B, H, ND, 16, 1, 0);
    const size_t esz = prec == LAS_PREC_BF16 ? 2 : 4, vec = prec == LAS_PREC_BF16 ? 8 : 4;
    const size_t plain = (size_t)ND * HX_SLOTS * B * ((H + vec - 1) / vec * vec) * esz;     // [ND][4][B][Hx] ring of lstm_fwd_kernel
    const size_t gr = fwd_gr_ring_bytes(a);                                                 // granule ring of lstm_fwd_gr_kernel
    return las_align(plain > gr ? plain : gr);
}

extern "C" size_t las_lstm_bwd_ws_bytes(int prec, int T, int B, int H, int ND) {
    LstmArgs a;
    BwdPlan p;
    if (check_common(T, B, H, ND, 1)) return 0;
    if (use_x32(prec, T, B, H, ND, 1, 0, a)) return bwd_x32_ring_bytes(a);
    fill_args(a, T, B, H, ND, 16, 1, 0);
    return bwd_plan(prec, T, B, H, ND, a, p) == LAS_OK ? p.ws : 0;
}

extern "C" int las_lstm_bwd_is_ksplit(int prec, int T, int B, int H, int ND) {
    LstmArgs a;
    BwdPlan p;
    if (check_common(T, B, H, ND, 1)) return 0;
    if (use_x32(prec, T, B, H, ND, 1, 0, a)) return 3;
    fill_args(a, T, B, H, ND, 16, 1, 0);
    if (bwd_plan(prec, T, B, H, ND, a, p) != LAS_OK) return 0;
    return p.gr ? 2 : p.ks ? 1 : 0;
}

extern "C" int las_lstm_resident_wgs(int prec, int T, int B, int H, int ND) {
    LstmArgs a;
    (void)prec;
    if (check_common(T, B, H, ND, 1)) return 0;
    if (use_x32(prec, T, B, H, ND, 1, 0, a)) return a.ND * a.G * a.NS;
    fill_args(a, T, B, H, ND, 16, 1, 0);
    return a.ND * a.G * a.NS;            // (an XCD-grouped launch starts more, the ones without a group exit at once)
}

// 1: lstm_fwd_gr_kernel (tagged-granule hand-off), 0: lstm_fwd_kernel.  Same rule as in las_lstm_rec_fwd below.
static bool fwd_uses_gr(int prec, int T, int B, int H, int ND, const LstmArgs& a, int U, int NB, int KS) {
    // KS = 32 (512 < H <= 1024, one batch tile per slice): the weight fragments take 128 registers per lane, the sweep lists
    // move to LDS and a lane sweeps up to 8 granules per step
    const bool big = KS == 32 && NB == 1;
    return prec == LAS_PREC_BF16 && KS > 0 && (KS <= 16 || big) && U == 16 && H % 4 == 0 && (long)T * B * ND * 4 * H * 4 < (1l << 31) &&
           (long)H * ((a.Bs >> 4) * 3 + ((a.Bs & 15) + 5) / 6) <= (big ? 2048 : 1024) && !las_fallback("LAS_LSTM_NO_GR");
}
extern "C" int las_lstm_fwd_variant(int prec, int T, int B, int H, int ND) {
    LstmArgs a;
    if (check_common(T, B, H, ND, 1)) return 0;
    if (use_x32(prec, T, B, H, ND, 1, 0, a)) return 2;
    if (ND * ((H + 15) / 16) > las_cu_count()) return 0;
    fill_args(a, T, B, H, ND, 16, 1, 0);
    const int NB = las_pick_nb(a.Bs), ksteps = (H + 31) / 32;
    int KS = 0;
    if (prec == LAS_PREC_BF16 && NB >= 1 && NB <= 2 && !LAS_AB_KNOB("LAS_LSTM_NO_DIRECT"))
        KS = ksteps <= 8 ? 8 : ksteps <= 10 ? 10 : (ksteps <= 16 && NB == 1) ? 16 : 0;
    if (prec == LAS_PREC_BF16 && NB >= 1 && NB <= 2 && KS == 0 && ksteps <= 32 && fwd_lds(prec, H, NB, false) > LDS_CAP) KS = 32;     // (as las_lstm_rec_fwd)
    const bool u8 = !a.xl && ND * ((H + 7) / 8) <= las_cu_count() && H == 512;
    return fwd_uses_gr(prec, T, B, H, ND, a, u8 ? 8 : 16, NB, KS) ? 1 : 0;
}

static int rec_fwd_impl(int prec, const float* xproj, const float* b_ih, const float* b_hh, const float* w_hh,
                        const int32_t* lens, int T, int B, int H, int ND, int sr, int concat, float* y,
                        float* hf, void* tw, bool* tw_done, void* hx, float* gates, float* cs, void* sync, int* status, void* stream,
                        int fxI = 0, const float* w_ih = nullptr);

// y_bf16 (may be null): the bf16 twin of y, [T_out][B][F_out] -- the operand of the projection GEMM behind the layer.  The granule
// and 32-unit kernels write it next to the fp32 stores (no second pass over y); behind the other kernels it is one cast pass.
extern "C" int las_lstm_rec_fwd(int prec, const float* xproj, const float* b_ih, const float* b_hh, const float* w_hh,
                                const int32_t* lens, int T, int B, int H, int ND, int sr, int concat, float* y,
                                float* hf, void* y_bf16, void* hx, float* gates, float* cs, void* sync, int* status, void* stream) {
    bool in_kernel = false;
    int rc = rec_fwd_impl(prec, xproj, b_ih, b_hh, w_hh, lens, T, B, H, ND, sr, concat, y, hf, y_bf16, &in_kernel, hx, gates,
                          cs, sync, status, stream);
    if (rc == LAS_OK && y_bf16 && !in_kernel) {
        int T_out, F_out;
        las_lstm_out_shape(T, H, ND, sr, concat, &T_out, &F_out);
        rc = las_cast_bf16(y, y_bf16, (int64_t)T_out * B * F_out, stream);
    }
    return rc;
}

static int rec_fwd_impl(int prec, const float* xproj, const float* b_ih, const float* b_hh, const float* w_hh,
                        const int32_t* lens, int T, int B, int H, int ND, int sr, int concat, float* y,
                        float* hf, void* tw, bool* tw_done, void* hx, float* gates, float* cs, void* sync, int* status, void* stream,
                        int fxI, const float* w_ih) {
    LAS_CHECK_ARG(xproj && b_ih && b_hh && w_hh && lens && y && hf && hx && gates && cs && sync && status);
    int rc = check_common(T, B, H, ND, sr);
    if (rc) return rc;
    if (prec != LAS_PREC_BF16 && prec != LAS_PREC_F32) return LAS_E_BADARG;
    int U = 16;
    LstmArgs a;
    if (use_x32(prec, T, B, H, ND, sr, concat, a)) {
        if (fxI > 0 && (fxI > 32 * FX_KS || (fxI & 3) || (long)T * B * fxI * 4 >= (1l << 31))) return LAS_E_UNSUPPORTED;
        a.fxI = fxI; a.w_ih = w_ih;
        a.y_is_hf = (y == hf);
        a.tw = tw; *tw_done = true;
        if (a.y_is_hf && sr != 1) return LAS_E_BADARG;
        hipStream_t st = (hipStream_t)stream;
        const size_t ringb = fwd_gr_ring_bytes(a);
        const bool behind = (char*)hx == (char*)sync + sizeof(SyncWords);
        LAS_HIP(hipMemsetAsync(sync, 0, sizeof(SyncWords) + (behind ? ringb : 0), st));
        if (!behind) LAS_HIP(hipMemsetAsync(hx, 0, ringb, st));
        size_t lds = fwd_x32_lds(H);
        if (lds < MIN_LDS) lds = MIN_LDS;
#define LAS_X32_FWD(KS_)                                                                                                   \
    {                                                                                                                     \
        auto k = fxI > 0 ? lstm_fwd_x32_kernel<KS_, true> : lstm_fwd_x32_kernel<KS_, false>;                              \
        LAS_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));               \
        hipLaunchKernelGGL(k, dim3(lstm_grid(a)), dim3(X32_NT), lds, st, a, xproj, b_ih, b_hh, w_hh, lens, y, hf, (u32x4*)hx, \
                           gates, cs, (SyncWords*)sync, status);                                                          \
        LAS_LAUNCH_OK();                                                                                                  \
        return LAS_OK;                                                                                                    \
    }
        if ((H + 31) / 32 <= 24) LAS_X32_FWD(24) else LAS_X32_FWD(32)
#undef LAS_X32_FWD
    }
    if (ND * ((H + 15) / 16) > las_cu_count()) return LAS_E_UNSUPPORTED;
    // use more, smaller unit slices when the chip has room (shorter MFMA chains per step)
    fill_args(a, T, B, H, ND, U, sr, concat);
    if (!a.xl && ND * ((H + 7) / 8) <= las_cu_count() && H >= 512 && H <= 512) {      // (an XCD-grouped launch beats the finer slicing;
        U = 8;                                                            // beyond 512 two batch slices of 16-unit groups do)
        fill_args(a, T, B, H, ND, U, sr, concat);
    }
    const int NB = las_pick_nb(a.Bs);
    if (NB == 0) return LAS_E_UNSUPPORTED;
    // bf16: register-resident weight fragments for the common hidden sizes and small batch slices (VGPR budget:
    // 4 KS for the weights + the h fragments of a chunk); otherwise the weight slab lives in LDS
    const int ksteps = (H + 31) / 32;
    int KS = 0;
    if (prec == LAS_PREC_BF16 && NB <= 2 && !LAS_AB_KNOB("LAS_LSTM_NO_DIRECT")) {
        if (ksteps <= 8) KS = 8;
        else if (ksteps <= 10) KS = 10;
        else if (ksteps <= 16 && NB == 1) KS = 16;
    }
    if (prec == LAS_PREC_BF16 && NB <= 2 && KS == 0 && ksteps <= 32 && fwd_lds(prec, H, NB, false) > LDS_CAP) KS = 32;
    size_t lds = fwd_lds(prec, H, NB, KS > 0);
    if (lds > LDS_CAP) return LAS_E_UNSUPPORTED;
    a.y_is_hf = (y == hf);
    a.tw = tw;
    if (a.y_is_hf && sr != 1) return LAS_E_BADARG;
    if (lds < MIN_LDS) lds = MIN_LDS;
    hipStream_t st = (hipStream_t)stream;
    // one fill for the sync words and, when the caller placed it right behind them, the granule ring
    const bool gr_fwd = fwd_uses_gr(prec, T, B, H, ND, a, U, NB, KS);
    LAS_HIP(hipMemsetAsync(sync, 0, sizeof(SyncWords) + ((gr_fwd && (char*)hx == (char*)sync + sizeof(SyncWords)) ? fwd_gr_ring_bytes(a) : 0), st));
#define LAS_FWD_ARGS a, lds, st, xproj, b_ih, b_hh, w_hh, lens, y, hf, hx, gates, cs, (SyncWords*)sync, status
    if (fxI > 0 && !(gr_fwd && NB == 1 && KS >= 8 && KS <= 16 && fxI <= 32 * FX_KS && (fxI & 3) == 0 && (long)T * B * fxI * 4 < (1l << 31)))
        return LAS_E_UNSUPPORTED;                        // (las_lstm_fwd_fx_ok: the caller runs the projection GEMM instead)
    if (gr_fwd) {
        *tw_done = true;
        // tagged-granule hand-off (lstm_fwd_gr_kernel): no flag, no drain, one barrier per step
        lds = fwd_gr_lds(H, NB);
        if (lds < MIN_LDS) lds = MIN_LDS;
        if (fxI > 0) {                                   // fused input projection: `xproj` is the layer input
            a.fxI = fxI; a.w_ih = w_ih;
            if (KS == 8) return launch_fwd_gr<1, 8, 4, true>(LAS_FWD_ARGS);
            if (KS == 10) return launch_fwd_gr<1, 10, 4, true>(LAS_FWD_ARGS);
            return launch_fwd_gr<1, 16, 4, true>(LAS_FWD_ARGS);
        }
        if (KS == 8)  { LAS_NB_SWITCH(NB, return (launch_fwd_gr<NB_ <= 2 ? NB_ : 1, 8>(LAS_FWD_ARGS))); }
        if (KS == 10) { LAS_NB_SWITCH(NB, return (launch_fwd_gr<NB_ <= 2 ? NB_ : 1, 10>(LAS_FWD_ARGS))); }
        if (KS == 16) { return launch_fwd_gr<1, 16>(LAS_FWD_ARGS); }
        if (KS == 32) { lds = fwd_gr_lds(H, 1, 8); return launch_fwd_gr<1, 32, 8>(LAS_FWD_ARGS); }
    }
    if (prec == LAS_PREC_BF16) {
        if (KS == 8)  { LAS_NB_SWITCH(NB, return (launch_fwd<LAS_PREC_BF16, NB_ <= 2 ? NB_ : 1, 8>(LAS_FWD_ARGS))); }
        if (KS == 10) { LAS_NB_SWITCH(NB, return (launch_fwd<LAS_PREC_BF16, NB_ <= 2 ? NB_ : 1, 10>(LAS_FWD_ARGS))); }
        if (KS == 16) { return launch_fwd<LAS_PREC_BF16, 1, 16>(LAS_FWD_ARGS); }
        if (KS == 32) { LAS_NB_SWITCH(NB, return (launch_fwd<LAS_PREC_BF16, NB_ <= 2 ? NB_ : 1, 32>(LAS_FWD_ARGS))); }
        LAS_NB_SWITCH(NB, return (launch_fwd<LAS_PREC_BF16, NB_, 0>(LAS_FWD_ARGS)));
    } else {
        LAS_NB_SWITCH(NB, return (launch_fwd<LAS_PREC_F32, NB_, 0>(LAS_FWD_ARGS)));
    }
#undef LAS_FWD_ARGS
    return LAS_E_BADARG;
}

// The same with the layer's input projection FUSED (bf16 mode, granule kernel with one batch tile per slice, input width I <= 96
// and a multiple of 4: las_lstm_fwd_fx_ok): x [T][B][I] and w_ih [ND*4H][I] instead of xproj = x W_ih^T -- the bottom layer's
// projection GEMM (K = 80: pure output traffic) and the kernel's read-back of its 4H-wide rows disappear.
extern "C" int las_lstm_fwd_fx_ok(int prec, int T, int B, int H, int ND, int I) {
    if (prec != LAS_PREC_BF16 || I < 4 || I > 32 * FX_KS || (I & 3) || (long)T * B * I * 4 >= (1l << 31)) return 0;
    const int var = las_lstm_fwd_variant(prec, T, B, H, ND);
    if (var == 2) return 1;                              // the 32-unit kernel: one batch tile by construction
    if (var != 1) return 0;
    LstmArgs a;
    fill_args(a, T, B, H, ND, 16, 1, 0);
    const int ksteps = (H + 31) / 32;
    return las_pick_nb(a.Bs) == 1 && ksteps <= 16 && !(ND * ((H + 7) / 8) <= las_cu_count() && !a.xl && H == 512);
}
extern "C" int las_lstm_rec_fwd_fx(int prec, const float* x, int I, const float* w_ih, const float* b_ih, const float* b_hh,
                                   const float* w_hh, const int32_t* lens, int T, int B, int H, int ND, int sr, int concat, float* y,
                                   float* hf, void* y_bf16, void* hx, float* gates, float* cs, void* sync, int* status, void* stream) {
    LAS_CHECK_ARG(x && w_ih && I > 0);
    bool in_kernel = false;
    int rc = rec_fwd_impl(prec, x, b_ih, b_hh, w_hh, lens, T, B, H, ND, sr, concat, y, hf, y_bf16, &in_kernel, hx, gates, cs, sync,
                          status, stream, I, w_ih);
    if (rc == LAS_OK && y_bf16 && !in_kernel) {
        int T_out, F_out;
        las_lstm_out_shape(T, H, ND, sr, concat, &T_out, &F_out);
        rc = las_cast_bf16(y, y_bf16, (int64_t)T_out * B * F_out, stream);
    }
    return rc;
}

static int rec_bwd_impl(int prec, const float* dy, const float* gates, const float* cs, const float* w_hh,
                        const int32_t* lens, int T, int B, int H, int ND, int sr, int concat, void* dgx,
                        float* dgf, void* tw, bool* tw_done, void* sync, int* status, void* stream);

// dgf_bf16 (may be null): the bf16 twin of dgf, [T][B][ND*4H] -- the operand of the d x / d W_ih / d W_hh GEMMs behind the launch;
// written by the granule and 32-unit kernels themselves, one cast pass behind the others.
extern "C" int las_lstm_rec_bwd(int prec, const float* dy, const float* gates, const float* cs, const float* w_hh,
                                const int32_t* lens, int T, int B, int H, int ND, int sr, int concat, void* dgx,
                                float* dgf, void* dgf_bf16, void* sync, int* status, void* stream) {
    bool in_kernel = false;
    int rc = rec_bwd_impl(prec, dy, gates, cs, w_hh, lens, T, B, H, ND, sr, concat, dgx, dgf, dgf_bf16, &in_kernel, sync, status, stream);
    if (rc == LAS_OK && dgf_bf16 && !in_kernel) rc = las_cast_bf16(dgf, dgf_bf16, (int64_t)T * B * ND * 4 * H, stream);
    return rc;
}

static int rec_bwd_impl(int prec, const float* dy, const float* gates, const float* cs, const float* w_hh,
                        const int32_t* lens, int T, int B, int H, int ND, int sr, int concat, void* dgx,
                        float* dgf, void* tw, bool* tw_done, void* sync, int* status, void* stream) {
    LAS_CHECK_ARG(dy && gates && cs && w_hh && lens && dgx && dgf && sync && status);
    int rc = check_common(T, B, H, ND, sr);
    if (rc) return rc;
    if (prec != LAS_PREC_BF16 && prec != LAS_PREC_F32) return LAS_E_BADARG;
    LstmArgs a;
    if (use_x32(prec, T, B, H, ND, sr, concat, a)) {
        a.tw = tw; *tw_done = true;
        hipStream_t st = (hipStream_t)stream;
        const size_t ringb = bwd_x32_ring_bytes(a);
        const bool behind = (char*)dgx == (char*)sync + sizeof(SyncWords);
        LAS_HIP(hipMemsetAsync(sync, 0, sizeof(SyncWords) + (behind ? ringb : 0), st));
        if (!behind) LAS_HIP(hipMemsetAsync(dgx, 0, ringb, st));
        size_t lds = bwd_x32_lds();
        if (lds < MIN_LDS) lds = MIN_LDS;
#define LAS_X32_BWD(MT_)                                                                                                   \
    {                                                                                                                     \
        auto k = lstm_bwd_x32_kernel<MT_>;                                                                                \
        LAS_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));               \
        hipLaunchKernelGGL(k, dim3(lstm_grid(a)), dim3(X32_NT), lds, st, a, dy, gates, cs, w_hh, lens, (u32x4*)dgx, dgf,   \
                           (SyncWords*)sync, status);                                                                     \
        LAS_LAUNCH_OK();                                                                                                  \
        return LAS_OK;                                                                                                    \
    }
        if (((H + 15) / 16 + 7) / 8 <= 6) LAS_X32_BWD(6) else LAS_X32_BWD(8)
#undef LAS_X32_BWD
    }
    fill_args(a, T, B, H, ND, 16, sr, concat);
    BwdPlan p;
    rc = bwd_plan(prec, T, B, H, ND, a, p);
    if (rc) return rc;
    const int NB = p.NB, NC = p.NC, K4p = p.K4p;
    a.wdirect = p.wdirect ? 1 : 0;
    a.tw = tw;
    const size_t lds = p.lds;
    hipStream_t st = (hipStream_t)stream;
    const bool ring_behind = p.gr && (char*)dgx == (char*)sync + sizeof(SyncWords);      // one fill for both
    LAS_HIP(hipMemsetAsync(sync, 0, sizeof(SyncWords) + (ring_behind ? p.ws : 0), st));
    if (p.gr) {
        *tw_done = true;
        if (!ring_behind) LAS_HIP(hipMemsetAsync(dgx, 0, p.ws, st));      // tags of earlier launches must not match
#define LAS_GR_GO(N_, M_)                                                                                              \
    {                                                                                                                 \
        auto k = lstm_bwd_gr_kernel<N_, M_, (M_ > 8 ? 12 : 4)>;                                                                          \
        LAS_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));           \
        hipLaunchKernelGGL(k, dim3(lstm_grid(a)), dim3(NT + 64), lds, st, a, dy, gates, cs, w_hh, lens, (u32x4*)dgx, dgf, \
                           (SyncWords*)sync, status);                                                                 \
        LAS_LAUNCH_OK();                                                                                              \
        return LAS_OK;                                                                                                \
    }
        const int mt = (a.G + 3) / 4;
        if (NB == 1) { if (mt <= 5) LAS_GR_GO(1, 5) else if (mt <= 8) LAS_GR_GO(1, 8) else LAS_GR_GO(1, 16) }
        else { if (mt <= 5) LAS_GR_GO(2, 5) else LAS_GR_GO(2, 8) }
#undef LAS_GR_GO
    }
    if (p.ks) {
#define LAS_KS_GO(P_, N_, M_)                                                                                          \
    {                                                                                                                 \
        auto k = lstm_bwd_ks_kernel<P_, N_, M_>;                                                                      \
        LAS_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));           \
        hipLaunchKernelGGL(k, dim3(lstm_grid(a)), dim3(NT), lds, st, a, dy, gates, cs, w_hh, lens, (unsigned*)dgx, dgf, \
                           (SyncWords*)sync, status);                                                                 \
        LAS_LAUNCH_OK();                                                                                              \
        return LAS_OK;                                                                                                \
    }
        const int mt = (a.G + 3) / 4;
        if (prec == LAS_PREC_BF16) {
            if (NB == 1) { if (mt <= 5) LAS_KS_GO(LAS_PREC_BF16, 1, 5) else if (mt <= 8) LAS_KS_GO(LAS_PREC_BF16, 1, 8) else LAS_KS_GO(LAS_PREC_BF16, 1, 16) }
            else { if (mt <= 5) LAS_KS_GO(LAS_PREC_BF16, 2, 5) else if (mt <= 8) LAS_KS_GO(LAS_PREC_BF16, 2, 8) else LAS_KS_GO(LAS_PREC_BF16, 2, 16) }
        } else { if (NB == 1) LAS_KS_GO(LAS_PREC_F32, 1, 1) else LAS_KS_GO(LAS_PREC_F32, 2, 1) }
#undef LAS_KS_GO
    }
#define LAS_BWD_ARGS a, NC, K4p, lds, st, dy, gates, cs, w_hh, lens, dgx, dgf, (SyncWords*)sync, status
    if (prec == LAS_PREC_BF16) {
        const int kq = K4p / 32 / 4;                        // k-steps per wave
        const bool direct = NB <= 2 && NC == 1 && !LAS_AB_KNOB("LAS_LSTM_NO_DIRECT");
        if (direct && kq == 8)  { LAS_NB_SWITCH(NB, return (launch_bwd<LAS_PREC_BF16, NB_ <= 2 ? NB_ : 1, 8>(LAS_BWD_ARGS))); }
        if (direct && kq == 10) { LAS_NB_SWITCH(NB, return (launch_bwd<LAS_PREC_BF16, NB_ <= 2 ? NB_ : 1, 10>(LAS_BWD_ARGS))); }
        if (direct && kq == 16 && NB == 1) { return launch_bwd<LAS_PREC_BF16, 1, 16>(LAS_BWD_ARGS); }
        LAS_NB_SWITCH(NB, return (launch_bwd<LAS_PREC_BF16, NB_, 0>(LAS_BWD_ARGS)));
    } else {
        LAS_NB_SWITCH(NB, return (launch_bwd<LAS_PREC_F32, NB_, 0>(LAS_BWD_ARGS)));
    }
#undef LAS_BWD_ARGS
    return LAS_E_BADARG;
}
