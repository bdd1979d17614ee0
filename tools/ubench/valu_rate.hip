// valu_rate.hip -- issue rate of the VALU instruction classes the shading kernels are made of (MI355X, wave64):
// every wave runs ITER x 64 instructions of one class on 8 independent register chains; blocks of 64 threads,
// W waves per SIMD resident (grid = 256 CUs x 4 SIMDs x W).  Prints cycles per wave-instruction per SIMD
// (wall time x clock / instructions issued on one SIMD).  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define ITER 2048
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP> __global__ __launch_bounds__(64) void k(float *out, float a, uint32_t ua, int iters) {
    float f[8]; uint32_t u[8]; uint64_t q[8];
    for (int i = 0; i < 8; ++i) { f[i] = a + threadIdx.x + i; u[i] = ua + threadIdx.x * 7 + i; q[i] = ((uint64_t) u[i] << 32) | u[i]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#define FMA(i)  asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(a));
#define MUL(i)  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(a));
#define ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ua));
#define XOR(i)  asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(ua));
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ua));
#define MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ua));
#define MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(u[i]), "v"(ua) : "vcc");
#define RCP(i)  asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
#define SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i]));
#define CND(i)  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(a) : "vcc");
#define LSHL(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
#define CMP(i)  asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(f[i]), "v"(a) : "vcc");
#define RDL(i)  asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(u[i]) : "s20");
#define DIVS(i) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(f[i]) : "v"(a) : "vcc");
#define MAX(i)  asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i]) : "v"(a));
#define FMAC(i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(f[i]) : "v"(a));
#define ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %1, 5" : "+v"(u[i]) : "v"(ua));
#define CVT(i)  asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(f[i]));
#define DEP(i)  asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[0]) : "v"(a));
            if (OP == 0) { REP8(FMA) } else if (OP == 1) { REP8(MUL) } else if (OP == 2) { REP8(ADDU) } else if (OP == 3) { REP8(XOR) }
            else if (OP == 4) { REP8(MULLO) } else if (OP == 5) { REP8(MULHI) } else if (OP == 6) { REP8(MAD64) } else if (OP == 7) { REP8(RCP) }
            else if (OP == 8) { REP8(SQRT) } else if (OP == 9) { REP8(CND) } else if (OP == 10) { REP8(LSHL) } else if (OP == 11) { REP8(PKFMA) }
            else if (OP == 12) { REP8(CMP) } else if (OP == 13) { REP8(RDL) } else if (OP == 14) { REP8(DIVS) } else if (OP == 15) { REP8(MAX) }
            else if (OP == 16) { REP8(FMAC) } else if (OP == 17) { REP8(ALIGN) } else if (OP == 18) { REP8(CVT) } else if (OP == 19) { REP8(DEP) }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += f[i] + (float) u[i] + (float) q[i];
    if (s == 1.2345f) out[threadIdx.x] = s;
}

template <int OP> double run(int waves_per_simd, float *d, double clock_hz) {
    const int grid = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<grid, 64>>>(d, 1.0001f, 3u, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<grid, 64>>>(d, 1.0001f, 3u, ITER); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double) waves_per_simd * ITER * 64.0;
    return ms * 1e-3 * clock_hz / instr_per_simd;
}

int main() {
    float *d; hipMalloc(&d, 4096);
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double hz = khz * 1e3;
    printf("clock %.0f MHz (nominal; the chip may run lower under load)\n", hz / 1e6);
    const char *names[] = { "v_fma_f32", "v_mul_f32", "v_add_u32", "v_xor_b32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_rcp_f32", "v_sqrt_f32",
                            "v_cndmask_b32", "v_lshlrev_b32", "v_pk_fma_f32", "v_cmp_gt_f32", "v_readlane_b32", "v_div_scale_f32", "v_max_f32", "v_fmac_f32",
                            "v_alignbit_b32", "v_cvt_f32_u32", "v_fma_f32 (dependent chain)" };
    printf("%-30s %8s %8s %8s %8s   cycles per wave-instruction per SIMD at W waves/SIMD\n", "instruction", "W=1", "W=2", "W=4", "W=8");
#define ROW(OP) { printf("%-30s", names[OP]); for (int w : {1, 2, 4, 8}) printf(" %8.2f", run<OP>(w, d, hz)); printf("\n"); }
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12) ROW(13) ROW(14) ROW(15) ROW(16) ROW(17) ROW(18) ROW(19)
    return 0;
}
