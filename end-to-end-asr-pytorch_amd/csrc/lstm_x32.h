// Persistent BiLSTM recurrence / BPTT for 512 < H <= 1024 (BASELINE configs[4]: 6 x 1024) -- included by lstm.hip inside its
// anonymous namespace (reference: the sequential half of nn.LSTM at src/asr.py:473-481 and its autograd).
//
// Why a third geometry.  With 16 units per workgroup a (direction, batch slice) group at H = 1024 is 64 workgroups: two
// XCDs.  Every hand-off then crosses the fabric (sc1 stores to the memory side, loads that miss every L2), and the tagged
// granule kernels that run H <= 512 at 1.3 / 2.0 us per step inside ONE XCD's L2 need 3.6 / 6.8 us there (cycle stamps: the
// sweep alone is 4 500 - 12 000 cycles, 1.4 - 1.7 passes per step; tools/sweep_lstm.py).  Here a workgroup
// owns 32 units, so a group is <= 32 workgroups = one XCD, and the exchange is the L2-local one again.
//
// What that costs on the CU: the workgroup's slice of W_hh is 128 rows x H = 256 KB of bf16 at H = 1024 -- half the CU's
// register file.  Eight compute waves (two per SIMD, 256 registers each) hold 128 registers of weight fragments each; there
// is no room for a ninth, I/O-only wave as in the 16-unit kernels (three waves on one SIMD would cap every wave at 168
// registers).  The global I/O that must stay out of the sweeping waves' memory queue (a wave's vector-memory operations
// retire in order: a cold x-projection load or a saved-activation store in front of a poll delays the poll) goes to waves
// 4-7, which do it WHILE waves 0-3 sweep the granules of the previous step; all eight then meet at the step's barrier and
// compute.  Same data flow, same LDS hand-over tiles, one barrier per forward step and two per backward step as before.
//
//   forward : wave w owns units 4w .. 4w+3 of the workgroup's 32 (A operand = their 16 (unit, gate) rows, B = the h tile);
//             lane (fr, fq) gets the i, f, g, o pre-activations of unit 4w + fq for batch row fr in its accumulator.
//   backward: K-split as lstm_bwd_gr_kernel: partial d h[:, all H] from the workgroup's own 128 d-gate columns
//             (k = gate * 32 + unit), wave w takes the 16-column tiles w, w + 8, ...; tile ti goes to consumer ti / 2 as
//             granules {4 consecutive columns of one batch row as bf16, tag}; a consumer's inbox is
//             [producer][row][8 column groups].
// bf16 mode, one batch tile per slice (<= 16 rows), H % 4 == 0.  Anything else keeps the 16-unit kernels.

#ifdef LAS_X32_STAMPS            // (cycle sums per phase; s_memtime waits drain the LDS queue, so absolute numbers are inflated)
#define X32_ST_DECL GR_ST_DECL
#define X32_ST(i) GR_ST(i)
#else
#define X32_ST_DECL
#define X32_ST(i)
#endif
constexpr int X32_NT = 512;                 // eight waves
constexpr int X32_XLD = 32 * 4 + 4;         // floats per batch row of the [32 units][4 gates] tiles
constexpr int X32_HLD = 32 * 2 + 2;         // ... of the {h, c} tile

template <int KS, bool FX = false>                  // FX: the layer's input projection fused, as in lstm_fwd_gr_kernel (input width <= 96)
__global__ __launch_bounds__(X32_NT) void lstm_fwd_x32_kernel(LstmArgs a, const float* __restrict__ xproj,
                                                               const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                               const float* __restrict__ w_hh, const int32_t* __restrict__ lens,
                                                               float* __restrict__ y, float* __restrict__ hf,
                                                               u32x4* __restrict__ ring, float* __restrict__ gates,
                                                               float* __restrict__ cs, SyncWords* sync, int* status) {
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = a.H, B = a.B, ND = a.ND;
    const int Kp = (H + 31) / 32 * 32, ld = Kp + 16;
    const Role role = lstm_role(a);
    if (role.idle) return;
    const int d = role.d, g = role.g, bs = role.bs, j0 = g * 32;
    const int b0 = bs * a.Bs, Bl = min(a.Bs, B - b0);
    bf16_t* Hl = (bf16_t*)smem;                         // [2][16][ld]; pad rows / columns stay zero
    float* Xl = (float*)(Hl + 2 * 16 * ld);             // [2][16][X32_XLD]  x-projection of the step (waves 4-7 -> all)
    float* Sg = Xl + 2 * 16 * X32_XLD;                  // [2][16][X32_XLD]  post-activation gates (all -> waves 4-7)
    float* Sh = Sg + 2 * 16 * X32_XLD;                  // [2][16][X32_HLD]  {h, c}
    int* lensl = (int*)(Sh + 2 * 16 * X32_HLD);         // [16]
    int* flag = lensl + 16;
    for (int i = threadIdx.x; i < 16 * ld; i += X32_NT) ((unsigned*)Hl)[i] = 0u;      // both tiles
    for (int i = threadIdx.x; i < 16; i += X32_NT) lensl[i] = i < Bl ? lens[b0 + i] : 0;
    __syncthreads();
    unsigned* cnt = &sync->cnt[(d * MAX_SLICES + bs) * CNT_STRIDE];
    const int gl = group_local(a, cnt, &sync->abort_, flag);       // decides the store flavour only
    if (gl < 0) { if (threadIdx.x == 0) *status = LAS_E_TIMEOUT; return; }
    const bool local = gl == 1;
    const int ND4H = ND * 4 * H, NDH = ND * H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
    const bool io = wave >= 4;                          // waves 4-7: global I/O while waves 0-3 sweep
    const int il = threadIdx.x - 256;                   // index among the 256 I/O lanes

    // ---- weight fragments: row fr of my tile = (unit 4 wave + fr / 4, gate fr % 4); lane holds k = 32 ks + 8 fq + {0..7}
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
    bf16x8 xfrag[FX ? FX_KS : 1];                      // FX: the same 16 rows of W_ih over the input width
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
    const int ul = 4 * wave + fq, je = j0 + ul;
    const bool evu = je < H;
    float c_state = 0.f;
    f32x4 bias;
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) bias[gi] = evu ? b_ih[d * 4 * H + gi * H + je] + b_hh[d * 4 * H + gi * H + je] : 0.f;
    const int nck = gr_chunks(a.Bs), nckl = gr_chunks(Bl), total = H * nckl;
    const long slot_stride = (long)a.NS * H * nck;
    u32x4* ringg = ring + (long)d * HX_SLOTS * slot_stride + (long)bs * H * nck;
    const int cc = fr >= 12 ? 2 : fr >= 6 ? 1 : 0;
    const bool writer = evu && fr == cc * 6;
    // sweep lists of waves 0-3: the workgroup's H * nckl granules in four contiguous quarters
    constexpr int SWMAX = 8;
    const int Q = (total + 3) / 4, nsw = (Q + 63) / 64;
    int g_off[SWMAX], g_dst[SWMAX];
#pragma unroll
    for (int u = 0; u < SWMAX; ++u) {
        const int idx = lane + 64 * u, i = (wave & 3) * Q + idx;
        const bool ok = !io && idx < Q && i < total;
        const int j = ok ? i / nckl : 0, c3 = ok ? i - j * nckl : 0;
        g_off[u] = ok ? (j * nck + c3) * 16 : OOB;
        g_dst[u] = ok ? ((((c3 * 6) * ld + j) << 1) | (c3 < 2 ? 1 : 0)) : -1;
    }

    // ---- global I/O of waves 4-7.  x-projection item = 4 consecutive units of one (batch row, gate): 16 x 4 x 8 = 512 items,
    // two per I/O lane; {h, c, y} item = 4 consecutive units of one batch row: 128 items (I/O lanes < 128)
    const long nrow = (long)a.T * B;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xproj, 0, (int)(nrow * (FX ? a.fxI : ND4H) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)gates, 0, (int)(nrow * ND4H * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)hf, 0, (int)(nrow * NDH * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)cs, 0, (int)(nrow * NDH * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)((long)a.T_out * B * a.F_out * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry16 = __builtin_amdgcn_make_buffer_rsrc(a.tw, 0, a.tw ? (int)((long)a.T_out * B * a.F_out * 2) : 0, 0x00020000);
    auto tstep = [&](int s) __attribute__((always_inline)) { const int sc = min(s, a.T - 1); return d == 0 ? sc : a.T - 1 - sc; };
    auto xload = [&](int s, u32x4 (&xr)[2]) __attribute__((always_inline)) {
        const int t = tstep(s);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if constexpr (FX) {                          // item = 4 consecutive input dims of one batch row: 16 x 24 items
                const int it = il + 256 * q, row = it / 24, k0 = (it - row * 24) * 4;
                const int off = (io && it < 384 && row < Bl && k0 < a.fxI) ? (int)((((long)t * B + b0 + row) * a.fxI + k0) * 4) : OOB;
                xr[q] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
            } else {
            const int it = il + 256 * q, row = it >> 5, gi = (it >> 3) & 3, u0 = (it & 7) * 4;
            const int off = (io && row < Bl && j0 + u0 < H) ? (int)((((long)t * B + b0 + row) * ND4H + d * 4 * H + gi * H + j0 + u0) * 4) : OOB;
            xr[q] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
            }
        }
    };
    auto xstore = [&](int s, const u32x4 (&xr)[2]) __attribute__((always_inline)) {
        if constexpr (FX) {                              // x_t as a bf16 tile [16][FX_LD] (in Xl's space)
            bf16_t* xb = (bf16_t*)Xl + (s & 1) * 16 * FX_LD;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int it = il + 256 * q, row = it / 24, k0 = (it - row * 24) * 4;
                if (it < 384)
                    *(u32x2*)(xb + row * FX_LD + k0) = (u32x2){pack_bf16x2(__uint_as_float(xr[q][0]), __uint_as_float(xr[q][1])),
                                                             pack_bf16x2(__uint_as_float(xr[q][2]), __uint_as_float(xr[q][3]))};
            }
            return;
        }
        unsigned* xl = (unsigned*)Xl + (s & 1) * 16 * X32_XLD;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int it = il + 256 * q, row = it >> 5, gi = (it >> 3) & 3, u0 = (it & 7) * 4;
            unsigned* p = xl + row * X32_XLD + u0 * 4 + gi;
            p[0] = xr[q][0]; p[4] = xr[q][1]; p[8] = xr[q][2]; p[12] = xr[q][3];
        }
    };
    auto sflush = [&](int s) __attribute__((always_inline)) {                           // saved activations of step s: LDS -> global (non-temporal)
        const int t = tstep(s);
        const unsigned* sg = (const unsigned*)Sg + (s & 1) * 16 * X32_XLD;
        const unsigned* sh = (const unsigned*)Sh + (s & 1) * 16 * X32_HLD;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int it = il + 256 * q, row = it >> 5, gi = (it >> 3) & 3, u0 = (it & 7) * 4;
            const unsigned* p = sg + row * X32_XLD + u0 * 4 + gi;
            const u32x4 v = {p[0], p[4], p[8], p[12]};
            const int off = (row < Bl && j0 + u0 < H) ? (int)((((long)t * B + b0 + row) * ND4H + d * 4 * H + gi * H + j0 + u0) * 4) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(v, rg, off, 0, 2);
        }
        if (il < 128) {
            const int row = il >> 3, u0 = (il & 7) * 4;
            const unsigned* p = sh + row * X32_HLD + u0 * 2;
            const u32x4 hv4 = {p[0], p[2], p[4], p[6]}, cv4 = {p[1], p[3], p[5], p[7]};
            const bool ok = row < Bl && j0 + u0 < H;
            const int b = b0 + row;
            const int off = ok ? (int)((((long)t * B + b) * NDH + d * H + j0 + u0) * 4) : OOB;
            bool yok = false;
            const long yo = y_offset(a, t, min(b, B - 1), d, j0 + u0, yok);
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

    // One step = W(s), the barrier, the computation.  W(s), before the barrier: waves 0-3 sweep the granules of step s-1 into
    // the h tile; waves 4-7 hand over the x-projection of step s (requested three steps ago), request step s+3's and store
    // step s-2's saved activations (written before barrier(s-1); their buffer is rewritten after barrier(s)).  The two roles
    // run two separate loops around the same barriers, so that neither carries the other's registers.
    X32_ST_DECL;
    auto barrier_ok = [&](int s) __attribute__((always_inline)) -> bool {               // the step's barrier + (every 64 steps) a block-uniform abort check
        if ((s & 63) == 0 && threadIdx.x == 0) *flag = (int)__hip_atomic_load(&sync->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        return !((s & 63) == 0 && *flag != 0);
    };
    auto sweep_step = [&](int s) __attribute__((always_inline)) {
        bf16_t* buf = Hl + (s & 1) * 16 * ld;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(ringg + ((s - 1) & (HX_SLOTS - 1)) * slot_stride), 0,
                                                                      H * nck * 16, 0x00020000);
        auto sweep = [&](auto swc) __attribute__((always_inline)) -> bool {
            constexpr int SW = decltype(swc)::value;
            u32x4 v[SW];
            int off[SW];
            for (int dl = 0; dl < (a.pdelay & 255); ++dl) __builtin_amdgcn_s_sleep(1);      // (diagnostic build: delay the first poll)
#pragma unroll
            for (int u = 0; u < SW; ++u) { off[u] = g_off[u]; v[u] = gr_poll(rs, off[u]); }
            unsigned spins = 0;
            while (true) {
                bool need = false;
#pragma unroll
                for (int u = 0; u < SW; ++u) {
                    const bool hit = off[u] != OOB && v[u][3] == (unsigned)s;
                    off[u] = hit ? OOB : off[u];
                    if (hit) {
                        bf16_t* dst = buf + (g_dst[u] >> 1);
                        dst[0] = (bf16_t)(v[u][0] & 0xffffu); dst[ld] = (bf16_t)(v[u][0] >> 16);
                        dst[2 * ld] = (bf16_t)(v[u][1] & 0xffffu); dst[3 * ld] = (bf16_t)(v[u][1] >> 16);
                        if (g_dst[u] & 1) { dst[4 * ld] = (bf16_t)(v[u][2] & 0xffffu); dst[5 * ld] = (bf16_t)(v[u][2] >> 16); }
                    }
                    need = need || off[u] != OOB;
                }
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
            return true;
        };
        if (nsw <= 2) sweep(std::integral_constant<int, 2>{});
        else if (nsw <= 4) sweep(std::integral_constant<int, 4>{});
        else if (nsw <= 6) sweep(std::integral_constant<int, 6>{});
        else sweep(std::integral_constant<int, 8>{});
    };
    auto compute = [&](int s) __attribute__((always_inline)) {
        const int t = d == 0 ? s : a.T - 1 - s;
        const bf16_t* buf = Hl + (s & 1) * 16 * ld;
        // ---- gate pre-activations of my four units: x-projection + bias + W_hh h_{t-1}
        const float* xl = Xl + (s & 1) * 16 * X32_XLD;
        f32x4 acc;
        if constexpr (FX) {
            const bf16_t* xb = (const bf16_t*)Xl + (s & 1) * 16 * FX_LD;
            acc = bias;
#pragma unroll
            for (int ks = 0; ks < FX_KS; ++ks)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xfrag[ks], *(const bf16x8*)(xb + fr * FX_LD + ks * 32 + fq * 8), acc, 0, 0, 0);
        } else {
            acc = *(const f32x4*)(xl + fr * X32_XLD + ul * 4) + bias;
        }
        f32x4 acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (s > 0 && !(a.pdelay & 256)) {
            constexpr int CH = 8, NCH = KS / CH;
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
                __builtin_amdgcn_sched_barrier(0);           // (keep the next chunk's reads in front of this chunk's MFMAs: hipcc sinks them otherwise)
#pragma unroll
                for (int ks = 0; ks < CH; ++ks) {
                    if (ks & 1) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag[c * CH + ks], hc[c & 1][ks], acc2, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag[c * CH + ks], hc[c & 1][ks], acc, 0, 0, 0);
                }
            }
        }
#ifdef LAS_X32_STAMPS
        asm volatile("" : "+v"(acc[0]), "+v"(acc2[0]));
#endif
        X32_ST(2);
        // ---- cell update on the accumulators; publish h_t; leave the saved activations to waves 4-7
        u32x4* slot = ringg + (s & (HX_SLOTS - 1)) * slot_stride;
        float* sg = Sg + (s & 1) * 16 * X32_XLD;
        float* sh = Sh + (s & 1) * 16 * X32_HLD;
        const int row = fr;
        const bool mq = evu && row < Bl && t < lensl[row];
        const f32x4 pre = acc + acc2;
        const float ig = fsig(pre[0]), fg = fsig(pre[1]), gg = ftanh(pre[2]), og = fsig(pre[3]);
        const float cn = fg * c_state + ig * gg;
        const float hn = og * ftanh(cn);
        c_state = mq ? cn : c_state;
        const float hv = mq ? hn : 0.f;
        const float hnext = las_dpp<0x101, 0xf>(0.f, hv);                      // row_shl:1 -> batch row fr + 1
        const unsigned p0 = pack_bf16x2(hv, hnext), p1 = dpp_u<0x102>(p0), p2 = dpp_u<0x104>(p0);
        const u32x4 gr = {p0, p1, p2, (unsigned)s + 1u};
        __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc((void*)slot, 0, H * nck * 16, 0x00020000);
        const int woff = (writer && row < Bl) ? je * nck * 16 + cc * 16 : OOB;
        if (local) __builtin_amdgcn_raw_buffer_store_b128(gr, ws, woff, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(gr, ws, woff, 0, 16);
        *(f32x4*)(sg + row * X32_XLD + ul * 4) = mq ? (f32x4){ig, fg, gg, og} : (f32x4){0.f, 0.f, 0.f, 0.f};
        *(float2*)(sh + row * X32_HLD + ul * 2) = make_float2(hv, mq ? c_state : 0.f);
        X32_ST(3);
    };

    if (io) {
        u32x4 x0[2], x1[2], x2[2];                       // three register sets: a value is used three steps after its request
        xload(0, x0); xload(1, x1); xload(2, x2);
        auto io_step = [&](int s, u32x4 (&xs)[2]) __attribute__((always_inline)) -> bool {
            xstore(s, xs);
            xload(s + 3, xs);
            if (s >= 2) sflush(s - 2);
            X32_ST(0);
            if (!barrier_ok(s)) return false;
            X32_ST(1);
            compute(s);
            return true;
        };
        for (int s = 0; s < a.T; s += 3) {
            if (!io_step(s, x0)) return;
            if (s + 1 >= a.T) break;
            if (!io_step(s + 1, x1)) return;
            if (s + 2 >= a.T) break;
            if (!io_step(s + 2, x2)) return;
        }
        __syncthreads();                                 // barrier(T): the last two steps' activations are in LDS
        if (a.T >= 2) sflush(a.T - 2);
        sflush(a.T - 1);
    } else {
        for (int s = 0; s < a.T; ++s) {
            if (s > 0 && !(a.pdelay & 512)) sweep_step(s);
            X32_ST(0);
            if (!barrier_ok(s)) return;
            X32_ST(1);
            compute(s);
        }
        __syncthreads();
    }
#ifdef LAS_X32_STAMPS
    if (blockIdx.x < 16 && (lane == 0) && (wave == 0 || wave == 4)) printf("x32 fwd wg %d wave %d local %d T=%d: W %u barrier %u mfma %u cell %u\n", (int)blockIdx.x, wave, (int)local, a.T, gst[0] / a.T, gst[1] / a.T, gst[2] / a.T, gst[3] / a.T);
#endif
}

size_t fwd_x32_lds(int H) {
    const int Kp = (H + 31) / 32 * 32, ld = Kp + 16;
    return (size_t)2 * 16 * ld * 2 + sizeof(float) * 2 * 16 * (2 * X32_XLD + X32_HLD) + sizeof(int) * (16 + 4);
}

// ------------------------------------------------------------------------------------------------ backward
constexpr int X32_LDK = 128 + 16;                       // own d-gates tile: k = gate * 32 + unit
constexpr int X32_PS = 32, X32_RLD = 32 * X32_PS + 4;   // partial sums [row][column 32][producer 32], rows padded (banks)
constexpr int X32_SWB = 12;                             // granules a sweeping lane takes per step at most (G * Bs * 8 <= 3072)

template <int MT>                                       // 16-column tiles per wave = ceil(ceil(H / 16) / 8)
__global__ __launch_bounds__(X32_NT) void lstm_bwd_x32_kernel(LstmArgs a, const float* __restrict__ dy,
                                                               const float* __restrict__ gates, const float* __restrict__ cs,
                                                               const float* __restrict__ w_hh, const int32_t* __restrict__ lens,
                                                               u32x4* __restrict__ ring, float* __restrict__ dgf,
                                                               SyncWords* sync, int* status) {
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = a.H, B = a.B, ND = a.ND, K4 = 4 * H, G = a.G;
    const Role role = lstm_role(a);
    if (role.idle) return;
    const int d = role.d, g = role.g, bs = role.bs, j0 = g * 32;
    const int b0 = bs * a.Bs, Bl = min(a.Bs, B - b0);
    bf16_t* Dl = (bf16_t*)smem;                           // [16][X32_LDK]     my d gates of this step (MFMA B operand)
    float* Red = (float*)(Dl + 16 * X32_LDK);             // [16][X32_RLD]
    float* Gi = Red + 16 * X32_RLD;                       // [2][16][X32_XLD]  saved gates of the step   (waves 4-7 -> all)
    float* Ci = Gi + 2 * 16 * X32_XLD;                    // [2][16][X32_XLD]  {c, c_prev, dy, -}
    float* Do = Ci + 2 * 16 * X32_XLD;                    // [2][16][X32_XLD]  d gates in f32            (all -> waves 4-7)
    int* lensl = (int*)(Do + 2 * 16 * X32_XLD);
    int* flag = lensl + 16;
    int* tab = flag + 4;                                  // [256][X32_SWB][2] sweep lists of waves 0-3
    for (int i = threadIdx.x; i < 16 * X32_LDK / 2; i += X32_NT) ((unsigned*)Dl)[i] = 0u;
    for (int i = threadIdx.x; i < 16 * X32_RLD; i += X32_NT) Red[i] = 0.f;
    for (int i = threadIdx.x; i < 16; i += X32_NT) lensl[i] = i < Bl ? lens[b0 + i] : 0;
    __syncthreads();
    unsigned* cnt = &sync->cnt[(d * MAX_SLICES + bs) * CNT_STRIDE];
    const int gl = group_local(a, cnt, &sync->abort_, flag);
    if (gl < 0) { if (threadIdx.x == 0) *status = LAS_E_TIMEOUT; return; }
    const bool local = gl == 1;
    const int ND4H = ND * K4, NDH = ND * H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
    const bool io = wave >= 4;
    const int il = threadIdx.x - 256;
    auto tstep = [&](int s) __attribute__((always_inline)) { const int sc = min(s, a.T - 1); return d == 0 ? a.T - 1 - sc : sc; };   // reverse of the forward order

    // ---- weight fragments (transposed product): rows = the 16 output columns of tile ti = wave + 8 i, k = my 128 gate
    // columns; element e of k-step ks <-> gate ks, unit 8 fq + e
    const int ntile = (H + 15) / 16;
    bf16x8 wfrag[MT][4];
#pragma clang loop unroll(full)
    for (int i = 0; i < MT; ++i) {
        const int ti = min(wave + 8 * i, ntile - 1), col = ti * 16 + fr;
        const bool colok = col < H;
        const float* wc = w_hh + (long)d * K4 * H + min(col, H - 1);
#pragma clang loop unroll(full)
        for (int ks = 0; ks < 4; ++ks) {
            unsigned pk[4];
#pragma clang loop unroll(full)
            for (int e2 = 0; e2 < 4; ++e2) {             // two k (= units) per packed register
                const int u0 = j0 + 8 * fq + 2 * e2;
                const float v0 = wc[((long)ks * H + min(u0, H - 1)) * H], v1 = wc[((long)ks * H + min(u0 + 1, H - 1)) * H];
                pk[e2] = pack_bf16x2((colok && u0 < H) ? v0 : 0.f, (colok && u0 + 1 < H) ? v1 : 0.f);
            }
            wfrag[i][ks] = __builtin_bit_cast(bf16x8, (u32x4){pk[0], pk[1], pk[2], pk[3]});
        }
    }
    const int ul = 4 * wave + fq;
    const bool evu = j0 + ul < H;
    float dc_carry = 0.f;
    // inboxes: ring[slot][d][bs][consumer][producer][Bs][8] granules
    const int RW = a.Bs * 8;
    const long slot_stride = (long)ND * a.NS * G * G * RW;
    u32x4* ringg = ring + ((long)d * a.NS + bs) * G * G * RW;
    const int total = G * Bl * 8, Q = (total + 3) / 4, nsw = (Q + 63) / 64;
    int* tabw = tab + (int)(threadIdx.x & 255) * X32_SWB * 2;
    if (!io) {
#pragma unroll
        for (int u = 0; u < X32_SWB; ++u) {
            const int idx = lane + 64 * u, i = wave * Q + idx;
            const bool ok = idx < Q && i < total;
            const int p = ok ? i / (Bl * 8) : 0, rem = ok ? i - p * (Bl * 8) : 0, row = rem >> 3, cg = rem & 7;
            tabw[2 * u] = ok ? ((g * G + p) * RW + rem) * 16 : OOB;
            tabw[2 * u + 1] = ok ? row * X32_RLD + cg * 4 * X32_PS + p : 0;
        }
    }

    // ---- global I/O of waves 4-7 (items as in the forward kernel)
    const long nrow = (long)a.T * B;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)gates, 0, (int)(nrow * ND4H * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)cs, 0, (int)(nrow * NDH * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)((long)a.T_out * B * a.F_out * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro_ = __builtin_amdgcn_make_buffer_rsrc((void*)dgf, 0, (int)(nrow * ND4H * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro16 = __builtin_amdgcn_make_buffer_rsrc(a.tw, 0, a.tw ? (int)(nrow * ND4H * 2) : 0, 0x00020000);
    struct In { u32x4 g4[2], c, cp, y; };
    auto xload = [&](int s, In& x) __attribute__((always_inline)) {
        const int t = tstep(s), tp = d == 0 ? t - 1 : t + 1;
        const bool sv = io && s < a.T;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int it = il + 256 * q, row = it >> 5, gi = (it >> 3) & 3, u0 = (it & 7) * 4;
            const bool ok = sv && row < Bl && j0 + u0 < H && t < lensl[row & 15];
            x.g4[q] = __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? (int)((((long)t * B + b0 + row) * ND4H + d * K4 + gi * H + j0 + u0) * 4) : OOB, 0, 0);
        }
        {
            const int row = (il >> 3) & 15, u0 = (il & 7) * 4, len = lensl[row], b = b0 + row;
            const bool ok = sv && il < 128 && row < Bl && j0 + u0 < H && t < len;
            x.c = __builtin_amdgcn_raw_buffer_load_b128(rc, ok ? (int)((((long)t * B + b) * NDH + d * H + j0 + u0) * 4) : OOB, 0, 0);
            x.cp = __builtin_amdgcn_raw_buffer_load_b128(rc, (ok && tp >= 0 && tp < len) ? (int)((((long)tp * B + b) * NDH + d * H + j0 + u0) * 4) : OOB, 0, 0);
            bool yok = false;
            const long yo = y_offset(a, t, min(b, B - 1), d, j0 + u0, yok);
            x.y = __builtin_amdgcn_raw_buffer_load_b128(ry, (ok && yok) ? (int)(yo * 4) : OOB, 0, 0);
        }
    };
    auto xstore = [&](int s, const In& x) __attribute__((always_inline)) {
        unsigned* gi_ = (unsigned*)Gi + (s & 1) * 16 * X32_XLD;
        unsigned* ci_ = (unsigned*)Ci + (s & 1) * 16 * X32_XLD;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int it = il + 256 * q, row = it >> 5, gi = (it >> 3) & 3, u0 = (it & 7) * 4;
            unsigned* p = gi_ + row * X32_XLD + u0 * 4 + gi;
            p[0] = x.g4[q][0]; p[4] = x.g4[q][1]; p[8] = x.g4[q][2]; p[12] = x.g4[q][3];
        }
        if (il < 128) {
            const int row = il >> 3, u0 = (il & 7) * 4;
            unsigned* p = ci_ + row * X32_XLD + u0 * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { p[4 * e + 0] = x.c[e]; p[4 * e + 1] = x.cp[e]; p[4 * e + 2] = x.y[e]; }
        }
    };
    auto sflush = [&](int s) __attribute__((always_inline)) {                           // d gates of step s: LDS -> global
        const int t = tstep(s);
        const unsigned* dg = (const unsigned*)Do + (s & 1) * 16 * X32_XLD;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int it = il + 256 * q, row = it >> 5, gi = (it >> 3) & 3, u0 = (it & 7) * 4;
            const unsigned* p = dg + row * X32_XLD + u0 * 4 + gi;
            const u32x4 v = {p[0], p[4], p[8], p[12]};
            const int off = (row < Bl && j0 + u0 < H) ? (int)((((long)t * B + b0 + row) * ND4H + d * K4 + gi * H + j0 + u0) * 4) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(v, ro_, off, 0, 0);
            const u32x2 v16 = {pack_bf16x2(__uint_as_float(v[0]), __uint_as_float(v[1])), pack_bf16x2(__uint_as_float(v[2]), __uint_as_float(v[3]))};
            __builtin_amdgcn_raw_buffer_store_b64(v16, ro16, off == OOB ? OOB : off >> 1, 0, 0);      // the bf16 twin (empty resource: none)
        }
    };

    // the part of the cell backward that does not depend on the recurrent d h (see lstm_bwd_gr_kernel)
    struct Pre { float dy, ka, k0, k1, k2, k3, kf; bool mq; };
    Pre pre;
    auto precompute = [&](int s) __attribute__((always_inline)) {
        const int t = d == 0 ? a.T - 1 - s : s;
        const float* gi_ = Gi + (s & 1) * 16 * X32_XLD;
        const float* ci_ = Ci + (s & 1) * 16 * X32_XLD;
        const f32x4 g4 = *(const f32x4*)(gi_ + fr * X32_XLD + ul * 4);
        const f32x4 c4 = *(const f32x4*)(ci_ + fr * X32_XLD + ul * 4);
        const float ig = g4[0], fg = g4[1], gg = g4[2], og = g4[3], tc = ftanh(c4[0]);
        pre.mq = evu && fr < Bl && t < lensl[fr];
        pre.dy = c4[2];
        pre.ka = og * (1.f - tc * tc);
        pre.k0 = gg * ig * (1.f - ig);
        pre.k1 = c4[1] * fg * (1.f - fg);
        pre.k2 = ig * (1.f - gg * gg);
        pre.k3 = tc * og * (1.f - og);
        pre.kf = fg;
    };

    // W(s) (after B(s-1), before A(s)): waves 0-3 sweep the pieces of step s-1; waves 4-7 store step s-1's d gates (complete
    // since B(s-1)), hand over step s+1's inputs (their buffer was last read by precompute(s-1), two barriers ago; the
    // compute lanes read them in W(s+1), behind A(s) and B(s)) and request step s+4's.  Two role loops, same barriers.
    X32_ST_DECL;
    auto barrier_ok = [&](int s) __attribute__((always_inline)) -> bool {               // barrier A(s) + (every 64 steps) a block-uniform abort check
        if ((s & 63) == 0 && threadIdx.x == 0) *flag = (int)__hip_atomic_load(&sync->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        return !((s & 63) == 0 && *flag != 0);
    };
    auto sweep_step = [&](int s) __attribute__((always_inline)) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(ringg + ((s - 1) & (KS_SLOTS - 1)) * slot_stride), 0,
                                                                      G * G * RW * 16, 0x00020000);
        auto sweep = [&](auto swc) __attribute__((always_inline)) -> bool {
            constexpr int SW = decltype(swc)::value;
            u32x4 v[SW];
            int off[SW];
            for (int dl = 0; dl < (a.pdelay & 255); ++dl) __builtin_amdgcn_s_sleep(1);      // (diagnostic build: delay the first poll)
#pragma unroll
            for (int u = 0; u < SW; ++u) { off[u] = tabw[2 * u]; v[u] = gr_poll(rs, off[u]); }
            precompute(s);                               // under the first poll's round trip
            unsigned spins = 0;
            while (true) {
                bool need = false;
#pragma unroll
                for (int u = 0; u < SW; ++u) {
                    const bool hit = off[u] != OOB && v[u][2] == (unsigned)s;
                    off[u] = hit ? OOB : off[u];
                    if (hit) {
                        float* dst = Red + tabw[2 * u + 1];
                        dst[0] = __uint_as_float(v[u][0] << 16); dst[X32_PS] = __uint_as_float(v[u][0] & 0xffff0000u);
                        dst[2 * X32_PS] = __uint_as_float(v[u][1] << 16); dst[3 * X32_PS] = __uint_as_float(v[u][1] & 0xffff0000u);
                    }
                    need = need || off[u] != OOB;
                }
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
            return true;
        };
        if (nsw <= 3) sweep(std::integral_constant<int, 3>{});
        else if (nsw <= 6) sweep(std::integral_constant<int, 6>{});
        else if (nsw <= 9) sweep(std::integral_constant<int, 9>{});
        else sweep(std::integral_constant<int, X32_SWB>{});
    };
    auto compute = [&](int s) __attribute__((always_inline)) {                          // between A(s) and the next W: cell backward, B(s), products, publish
        // ---- the rest of the pointwise BPTT of my element -> my d gates of this step
        {
            float* do_ = Do + (s & 1) * 16 * X32_XLD;
            float dh_rec = 0.f;
            if (s > 0) {
                const float* rp = Red + fr * X32_RLD + ul * X32_PS;
                f32x4 part[8];                               // every read is issued before the first add
#pragma unroll
                for (int p = 0; p < 8; ++p) part[p] = *(const f32x4*)(rp + 4 * p);
                f32x4 acc4 = part[0];
#pragma unroll
                for (int p = 1; p < 8; ++p) acc4 += part[p];
                dh_rec = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
            }
            const bool mq = pre.mq;
            const float dh = pre.dy + dh_rec;
            const float dc = dh * pre.ka + dc_carry;
            const f32x4 dg = {mq ? dc * pre.k0 : 0.f, mq ? dc * pre.k1 : 0.f, mq ? dc * pre.k2 : 0.f, mq ? dh * pre.k3 : 0.f};
            dc_carry = mq ? dc * pre.kf : 0.f;
            *(f32x4*)(do_ + fr * X32_XLD + ul * 4) = dg;
            const unsigned lo = pack_bf16x2(dg[0], dg[1]), hi = pack_bf16x2(dg[2], dg[3]);
            bf16_t* dl = Dl + fr * X32_LDK + ul;
            dl[0] = (bf16_t)(lo & 0xffffu); dl[32] = (bf16_t)(lo >> 16); dl[64] = (bf16_t)(hi & 0xffffu); dl[96] = (bf16_t)(hi >> 16);
        }
        X32_ST(2);
        __syncthreads();                                 // B(s): d gates tile complete, d gates of step s in Do
        X32_ST(5);
        if (s + 1 < a.T) {
            // partial dh_{t-1}[:, 16 ti .. 16 ti + 15] for my tiles ti = wave + 8 i; tile ti belongs to consumer ti / 2
            __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc((void*)(ringg + (s & (KS_SLOTS - 1)) * slot_stride), 0,
                                                                          G * G * RW * 16, 0x00020000);
            bf16x8 dv[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dv[ks] = *(const bf16x8*)(Dl + fr * X32_LDK + ks * 32 + fq * 8);
            f32x4 acc[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag[i][0], dv[0], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int ks = 1; ks < 4; ++ks) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag[i][ks], dv[ks], acc[i], 0, 0, 0);
            }
            // lane: batch row fr, output columns 16 ti + 4 fq + {0..3}
            auto put = [&](auto auxc) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int ti = wave + 8 * i, c = ti >> 1, cg = (ti & 1) * 4 + fq;
                    const u32x4 gr = {pack_bf16x2(acc[i][0], acc[i][1]), pack_bf16x2(acc[i][2], acc[i][3]), (unsigned)s + 1u, 0u};
                    const int woff = (ti < ntile && c < G && fr < Bl) ? ((c * G + g) * RW + fr * 8 + cg) * 16 : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(gr, ws, woff, 0, decltype(auxc)::value);
                }
            };
            if (local) put(std::integral_constant<int, 0>{}); else put(std::integral_constant<int, 16>{});
        }
        X32_ST(3);
    };

    if (io) {
        In x0, x1, x2;                                   // three sets: a value is used three steps after its request
        xload(0, x0); xload(1, x1); xload(2, x2);
        xstore(0, x0);
        xload(3, x0);
        __syncthreads();                                 // inputs of step 0 in place
        // W(s) hands over step s+1 from set (s+1) % 3 and refills it with step s+4
        auto io_step = [&](int s, In& xs) __attribute__((always_inline)) -> bool {
            if (s > 0) sflush(s - 1);
            xstore(s + 1, xs);
            xload(s + 4, xs);
            precompute(s);
            X32_ST(0);
            if (!barrier_ok(s)) return false;
            X32_ST(1);
            compute(s);
            return true;
        };
        for (int s = 0; s < a.T; s += 3) {
            if (!io_step(s, x1)) return;
            if (s + 1 >= a.T) break;
            if (!io_step(s + 1, x2)) return;
            if (s + 2 >= a.T) break;
            if (!io_step(s + 2, x0)) return;
        }
        __syncthreads();
        sflush(a.T - 1);
    } else {
        __syncthreads();                                 // inputs of step 0 in place
        for (int s = 0; s < a.T; ++s) {
            if (s > 0) sweep_step(s); else precompute(s);
            X32_ST(0);
            if (!barrier_ok(s)) return;
            X32_ST(1);
            compute(s);
        }
        __syncthreads();
    }
#ifdef LAS_X32_STAMPS
    if (blockIdx.x < 16 && (lane == 0) && (wave == 0 || wave == 4)) printf("x32 bwd wg %d wave %d local %d T=%d: W %u barrierA %u cell %u barrierB %u products %u\n", (int)blockIdx.x, wave, (int)local, a.T, gst[0] / a.T, gst[1] / a.T, gst[2] / a.T, gst[5] / a.T, gst[3] / a.T);
#endif
}

size_t bwd_x32_lds() {
    return (size_t)16 * X32_LDK * 2 + sizeof(float) * 16 * (X32_RLD + 3 * 2 * X32_XLD) + sizeof(int) * (16 + 4) +
           sizeof(int) * 256 * X32_SWB * 2;
}
size_t bwd_x32_ring_bytes(const LstmArgs& a) { return (size_t)16 * KS_SLOTS * a.ND * a.NS * a.G * a.G * a.Bs * 8; }
