// What v_permlane32_swap / v_permlane16_swap (gfx950) do, lane by lane: prints, for a = lane and b = 100 + lane, the two results of each.
// Build: hipcc --offload-arch=gfx950 -O3 permlane_swap.hip -o permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
    unsigned a = threadIdx.x, b = threadIdx.x + 100;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1]; out[128 + threadIdx.x] = s[0]; out[192 + threadIdx.x] = s[1];
}
int main() {
    unsigned *d, h[256]; hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[4] = { "permlane32_swap[0]", "permlane32_swap[1]", "permlane16_swap[0]", "permlane16_swap[1]" };
    for (int r = 0; r < 4; ++r) { printf("%s:", names[r]); for (int i = 0; i < 64; ++i) printf(" %u", h[64 * r + i]); printf("\n"); }
    return 0;
}
