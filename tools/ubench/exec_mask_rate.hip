// Does a VALU instruction cost less when part of the wave is masked off?  (MI355X, wave64.)  Every wave runs ITER x 64 instructions of one form on 8 independent
// register chains with exec set to a given mask; W waves per SIMD.  Prints nominal cycles (2.4 GHz) per wave-instruction per SIMD for each mask.
// Build: hipcc --offload-arch=gfx950 -O3 exec_mask_rate.hip -o exec_mask_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <algorithm>
template <int OP> __global__ __launch_bounds__(64) void k(float *out, float a, int iters, uint64_t mask) {
    float f[8]; for (int i = 0; i < 8; ++i) f[i] = a + threadIdx.x + i;
    uint32_t lo = (uint32_t) mask, hi = (uint32_t) (mask >> 32);
    asm volatile("s_mov_b64 s[24:25], exec\n s_mov_b32 exec_lo, %0\n s_mov_b32 exec_hi, %1" : : "s"(lo), "s"(hi) : "s24", "s25");
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                                  "v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7"
                                  : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(a));
        if (OP == 1) asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n"
                                  "v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8"
                                  : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(a));
        if (OP == 2) asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                                  "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                                  : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(a));
        if (OP == 3) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                                  "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                                  : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(a) : "vcc");
    }
    asm volatile("s_mov_b64 exec, s[24:25]" : : : "s24", "s25");
    float s = 0; for (int i = 0; i < 8; ++i) s += f[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int OP> double run(float *d, int waves, uint64_t mask) {
    const int iters = 4096, grid = 256 * 4 * waves;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double best = 1e30;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(64), 0, 0, d, 1.0001f, iters, mask); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r) best = std::min(best, (double) ms);
    }
    return best * 1e-3 * 2.4e9 / ((double) iters * 16 * waves);   // cycles per wave-instruction per SIMD
}
int main() {
    float *d; hipMalloc(&d, 4096);
    const char *names[4] = { "v_fma_f32", "v_max_f32", "v_rcp_f32", "v_cndmask_b32" };
    struct { const char *what; uint64_t m; } masks[] = { { "all 64", ~0ull }, { "low 32", 0xffffffffull }, { "high 32", 0xffffffff00000000ull }, { "low 16", 0xffffull }, { "lanes 16-31", 0xffff0000ull },
        { "even lanes (32)", 0x5555555555555555ull }, { "one per 16 (4)", 0x0001000100010001ull }, { "20 scattered", 0x8421084210842108ull ^ 0x0000100000100001ull }, { "lane 0", 1ull }, { "none", 0ull } };
    for (int w : { 1, 4 }) {
        printf("-- %d wave(s) per SIMD: nominal cycles per wave-instruction per SIMD\n%-18s", w, "exec mask");
        for (auto n : names) printf(" %14s", n);
        printf("\n");
        for (auto &m : masks) {
            printf("%-18s %14.2f %14.2f %14.2f %14.2f\n", m.what, run<0>(d, w, m.m), run<1>(d, w, m.m), run<2>(d, w, m.m), run<3>(d, w, m.m));
        }
    }
    return 0;
}
