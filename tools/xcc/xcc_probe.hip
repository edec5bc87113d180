// Which XCD does block b of consecutive launches land on?  (placement probe; prints HW_REG_XCC_ID of the first blocks)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int* out) {
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[blockIdx.x] = (int)(x & 15u);
    }
}
int main() {
    const int grids[] = {80, 80, 80, 300, 80, 240, 80, 13, 80, 80, 160, 160, 7, 160, 160};
    int* d;
    hipMalloc(&d, sizeof(int) * 4096);
    std::vector<int> h(4096);
    hipStream_t st;
    hipStreamCreate(&st);
    // back-to-back launches on one stream, results read after all of them
    int* dd;
    hipMalloc(&dd, sizeof(int) * 4096 * 16);
    int n = 0;
    for (int g : grids) { hipLaunchKernelGGL(probe, dim3(g), dim3(256), 0, st, dd + 4096 * n); ++n; }
    hipStreamSynchronize(st);
    n = 0;
    for (int g : grids) {
        hipMemcpy(h.data(), dd + 4096 * n, sizeof(int) * g, hipMemcpyDeviceToHost);
        printf("launch %2d grid %3d: xcc of blocks 0..15:", n, g);
        for (int b = 0; b < 16 && b < g; ++b) printf(" %d", h[b]);
        bool rr = true;
        for (int b = 8; b < g; ++b) rr = rr && h[b] == h[b - 8];
        printf("   (b and b+8 share an XCD: %s)\n", rr ? "yes" : "NO");
        ++n;
    }
    return 0;
}
