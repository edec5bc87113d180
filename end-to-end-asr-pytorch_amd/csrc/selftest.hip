// Start-up self-test of the property every tagged-granule hand-off rests on (pk_common.h, lstm.hip): a naturally aligned 16-byte
// store {payload, tag} by one lane is observed UNTORN by 16-byte sc1 loads of other workgroups -- measured on gfx950
// (MI355X_MICROARCH.md "R2"), not an architectural guarantee.  Workgroups play ping-pong in pairs through granules whose payload is
// a hash of their tag, with exactly the store / poll instructions of the product kernels (cross-XCD: sc1 store + sc1 poll; inside one
// XCD: plain store + sc1 poll); EVERY polled value -- also the ones carrying an older tag -- must be consistent with its own tag.
// A torn or inconsistent observation, or a pair that does not finish, makes the caller fall back to the counter-protocol kernels
// (`LAS_LSTM_NO_GR`, `LAS_DEC_NO_PK`): python side in _lib.py.
#include "pk_common.h"

namespace {

__device__ __forceinline__ unsigned st_hash(unsigned t, unsigned lane) { return (t * 2654435761u) ^ (lane * 40503u) ^ 0x9e3779b9u; }
__device__ __forceinline__ bool st_consistent(const u32x4& v, unsigned lane) {
    if (v[3] == 0u) return v[0] == 0u && v[1] == 0u && v[2] == 0u;           // never written: the zero fill
    const unsigned h = st_hash(v[3], lane);
    return v[0] == h && v[1] == ~h && v[2] == (h ^ 0x5a5a5a5au);
}

// grid = 2 * pairs workgroups of ONE wave; pair p = blocks p (side 0) and p + pairs (side 1): with pairs a multiple of 8 both sit on
// XCD p % 8 (round-robin placement: the `local` flavour), with cross the partner is pair (p + 1) % pairs' side 1 (another XCD).
// slots: [2 sides][pairs][64 lanes] granules, zeroed.  res[0] += inconsistent observations, res[1] += lanes that finished,
// res[2] += timeouts.
__global__ __launch_bounds__(64) void granule_selftest_kernel(u32x4* slots, unsigned* res, int iters, int pairs, int cross, int plain) {
    const int side = (int)blockIdx.x >= pairs, p = (int)blockIdx.x - side * pairs, lane = threadIdx.x;
    const int peer = cross ? (side ? (p + pairs - 1) % pairs : (p + 1) % pairs) : p;
    __amdgpu_buffer_rsrc_t mine = __builtin_amdgcn_make_buffer_rsrc((void*)(slots + ((size_t)side * pairs + p) * 64), 0, 64 * 16, 0x00020000);
    __amdgpu_buffer_rsrc_t theirs = __builtin_amdgcn_make_buffer_rsrc((void*)(slots + ((size_t)(1 - side) * pairs + peer) * 64), 0, 64 * 16, 0x00020000);
    unsigned bad = 0, timeouts = 0;
    bool alive = true;
    for (int t = 1; t <= iters && alive; ++t) {
        if (side == 0) {                                         // serve
            const unsigned h = st_hash((unsigned)t, lane);
            pk_gr_store_x(mine, lane * 16, h, ~h, h ^ 0x5a5a5a5au, (unsigned)t, plain != 0);
        }
        // wait for the partner's granule of this round, checking everything seen on the way
        unsigned spins = 0;
        while (true) {
            const u32x4 v = pk_gr_poll(theirs, lane * 16);
            if (!st_consistent(v, lane)) ++bad;
            if (v[3] == (unsigned)t) break;
            if (++spins > (1u << 22)) { ++timeouts; alive = false; break; }
        }
        alive = __builtin_amdgcn_ballot_w64(!alive) == 0ull;    // (a wave leaves together)
        if (side == 1 && alive) {                                // return
            const unsigned h = st_hash((unsigned)t, lane);
            pk_gr_store_x(mine, lane * 16, h, ~h, h ^ 0x5a5a5a5au, (unsigned)t, plain != 0);
        }
    }
    if (bad) atomicAdd(&res[0], bad);
    if (alive) atomicAdd(&res[1], 1u);
    if (timeouts) atomicAdd(&res[2], timeouts);
}

}  // namespace

// workspace: las_granule_selftest_bytes() bytes (zeroed by the call).  result[3] (host): inconsistent observations, lanes that
// finished (2 * pairs * 64 expected), timeouts -- over both flavours.  Synchronises the stream.  LAS_OK whatever the outcome of the
// test itself; the caller reads `result`.
extern "C" size_t las_granule_selftest_bytes(void) { return sizeof(u32x4) * 2 * 64 * 64 + 256; }
extern "C" int las_granule_selftest(int iters, void* workspace, unsigned* result, void* stream) {
    LAS_CHECK_ARG(workspace && result && iters > 0);
    hipStream_t st = (hipStream_t)stream;
    const int pairs = 64;
    unsigned* res = (unsigned*)((char*)workspace + sizeof(u32x4) * 2 * 64 * 64);
    result[0] = result[1] = result[2] = 0;
    for (int flavour = 0; flavour < 2; ++flavour) {             // 0: inside one XCD, plain stores; 1: across XCDs, sc1 stores
        LAS_HIP(hipMemsetAsync(workspace, 0, las_granule_selftest_bytes(), st));
        hipLaunchKernelGGL(granule_selftest_kernel, dim3(2 * pairs), dim3(64), 0, st, (u32x4*)workspace, res, iters, pairs, flavour, flavour == 0 ? 1 : 0);
        LAS_LAUNCH_OK();
        unsigned h[3];
        LAS_HIP(hipMemcpyAsync(h, res, sizeof(h), hipMemcpyDeviceToHost, st));
        LAS_HIP(hipStreamSynchronize(st));
        for (int i = 0; i < 3; ++i) result[i] += h[i];
    }
    return LAS_OK;
}
