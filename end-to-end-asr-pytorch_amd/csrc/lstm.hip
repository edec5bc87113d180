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
__device__ __forceinline__ float fsig(float x) { return __fdividef(1.f, 1.f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 1.f - __fdividef(2.f, __expf(2.f * x) + 1.f); }

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

void fill_args(LstmArgs& a, int T, int B, int H, int ND, int U, int sr, int concat) {
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
    while (ns > 1 && ((long)ND * a.G * ns > 256 || ns > MAX_SLICES)) --ns;
    a.Bs = (B + ns - 1) / ns;
    a.NS = (B + a.Bs - 1) / a.Bs;
    a.sr = sr; a.concat = concat;
    if (sr == 1) { a.T_out = T; a.F_out = ND * H; }
    else if (concat) { a.T_out = T / sr; a.F_out = sr * ND * H; }
    else { a.T_out = (T + sr - 1) / sr; a.F_out = ND * H; }
    a.y_is_hf = 0; a.wdirect = 0;
    // XCD-grouped launch when every XCD (32 CUs, one workgroup per CU) can hold the groups dealt to it
    const bool no_xl = getenv("LAS_LSTM_NO_XL") != nullptr;
    const int groups = ND * a.NS, gpl = (groups + 7) / 8;
    a.xl = (!no_xl && a.G * gpl <= 32) ? 1 : 0;
    const bool no_nt = getenv("LAS_LSTM_NO_NT") != nullptr;
    a.nt = no_nt ? 0 : 1;
    const bool no_pf = getenv("LAS_LSTM_NO_PF") != nullptr;
    a.pf = no_pf ? 0 : 1;
}
int lstm_grid(const LstmArgs& a) { return a.xl ? 8 * a.G * ((a.ND * a.NS + 7) / 8) : a.ND * a.G * a.NS; }

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
struct BwdPlan { bool ks, wdirect; int NB, NC, K4p; size_t lds, ws; };
int bwd_plan(int prec, int T, int B, int H, int ND, const LstmArgs& a, BwdPlan& p) {
    p.NB = las_pick_nb(a.Bs);
    if (p.NB == 0 || ND * ((H + 15) / 16) > 256) return LAS_E_UNSUPPORTED;
    // K-split exchange (reduce-scatter of partial dh) whenever it fits: the [H][64] weight slab in LDS, or (bf16) the
    // weight fragments fetched from global memory into registers (ceil(G / 4) <= 16 tiles per wave: H <= 1024)
    p.ks = false; p.wdirect = false;
    if (p.NB <= 2 && !getenv("LAS_LSTM_BWD_GATHER")) {
        const int mt = (a.G + 3) / 4;
        if (bwd_ks_lds(prec, H, p.NB, false) <= LDS_CAP && (prec != LAS_PREC_BF16 || mt <= 16)) p.ks = true;
        else if (prec == LAS_PREC_BF16 && mt <= 16 && bwd_ks_lds(prec, H, p.NB, true) <= LDS_CAP) p.ks = p.wdirect = true;
    }
    if (p.ks) {
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

extern "C" size_t las_lstm_bwd_ws_bytes(int prec, int T, int B, int H, int ND) {
    LstmArgs a;
    BwdPlan p;
    if (check_common(T, B, H, ND, 1)) return 0;
    fill_args(a, T, B, H, ND, 16, 1, 0);
    return bwd_plan(prec, T, B, H, ND, a, p) == LAS_OK ? p.ws : 0;
}

extern "C" int las_lstm_bwd_is_ksplit(int prec, int T, int B, int H, int ND) {
    LstmArgs a;
    BwdPlan p;
    if (check_common(T, B, H, ND, 1)) return 0;
    fill_args(a, T, B, H, ND, 16, 1, 0);
    return bwd_plan(prec, T, B, H, ND, a, p) == LAS_OK && p.ks ? 1 : 0;
}

extern "C" int las_lstm_rec_fwd(int prec, const float* xproj, const float* b_ih, const float* b_hh, const float* w_hh,
                                const int32_t* lens, int T, int B, int H, int ND, int sr, int concat, float* y,
                                float* hf, void* hx, float* gates, float* cs, void* sync, int* status, void* stream) {
    LAS_CHECK_ARG(xproj && b_ih && b_hh && w_hh && lens && y && hf && hx && gates && cs && sync && status);
    int rc = check_common(T, B, H, ND, sr);
    if (rc) return rc;
    if (prec != LAS_PREC_BF16 && prec != LAS_PREC_F32) return LAS_E_BADARG;
    int U = 16;
    if (ND * ((H + 15) / 16) > 256) return LAS_E_UNSUPPORTED;
    // use more, smaller unit slices when the chip has room (shorter MFMA chains per step)
    LstmArgs a;
    fill_args(a, T, B, H, ND, U, sr, concat);
    if (!a.xl && ND * ((H + 7) / 8) <= 256 && H >= 512 && H <= 512) {      // (an XCD-grouped launch beats the finer slicing;
        U = 8;                                                            // beyond 512 two batch slices of 16-unit groups do)
        fill_args(a, T, B, H, ND, U, sr, concat);
    }
    const int NB = las_pick_nb(a.Bs);
    if (NB == 0) return LAS_E_UNSUPPORTED;
    // bf16: register-resident weight fragments for the common hidden sizes and small batch slices (VGPR budget:
    // 4 KS for the weights + the h fragments of a chunk); otherwise the weight slab lives in LDS
    const int ksteps = (H + 31) / 32;
    int KS = 0;
    if (prec == LAS_PREC_BF16 && NB <= 2 && !getenv("LAS_LSTM_NO_DIRECT")) {
        if (ksteps <= 8) KS = 8;
        else if (ksteps <= 10) KS = 10;
        else if (ksteps <= 16 && NB == 1) KS = 16;
    }
    if (prec == LAS_PREC_BF16 && NB <= 2 && KS == 0 && ksteps <= 32 && fwd_lds(prec, H, NB, false) > LDS_CAP) KS = 32;
    size_t lds = fwd_lds(prec, H, NB, KS > 0);
    if (lds > LDS_CAP) return LAS_E_UNSUPPORTED;
    a.y_is_hf = (y == hf);
    if (a.y_is_hf && sr != 1) return LAS_E_BADARG;
    if (lds < MIN_LDS) lds = MIN_LDS;
    hipStream_t st = (hipStream_t)stream;
    LAS_HIP(hipMemsetAsync(sync, 0, sizeof(SyncWords), st));
#define LAS_FWD_ARGS a, lds, st, xproj, b_ih, b_hh, w_hh, lens, y, hf, hx, gates, cs, (SyncWords*)sync, status
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

extern "C" int las_lstm_rec_bwd(int prec, const float* dy, const float* gates, const float* cs, const float* w_hh,
                                const int32_t* lens, int T, int B, int H, int ND, int sr, int concat, void* dgx,
                                float* dgf, void* sync, int* status, void* stream) {
    LAS_CHECK_ARG(dy && gates && cs && w_hh && lens && dgx && dgf && sync && status);
    int rc = check_common(T, B, H, ND, sr);
    if (rc) return rc;
    if (prec != LAS_PREC_BF16 && prec != LAS_PREC_F32) return LAS_E_BADARG;
    LstmArgs a;
    fill_args(a, T, B, H, ND, 16, sr, concat);
    BwdPlan p;
    rc = bwd_plan(prec, T, B, H, ND, a, p);
    if (rc) return rc;
    const int NB = p.NB, NC = p.NC, K4p = p.K4p;
    a.wdirect = p.wdirect ? 1 : 0;
    const size_t lds = p.lds;
    hipStream_t st = (hipStream_t)stream;
    LAS_HIP(hipMemsetAsync(sync, 0, sizeof(SyncWords), st));
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
        const bool direct = NB <= 2 && NC == 1 && !getenv("LAS_LSTM_NO_DIRECT");
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
