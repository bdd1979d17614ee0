#!/usr/bin/env python3
"""Writes valu_rate.hip: one kernel per VALU instruction form, its loop body ONE asm block of 64 instructions on 8 independent register
chains (no compiler-inserted s_nop / moves between the timed instructions).  {c} = the chain's register, {a} = a second VGPR of the same kind."""
OPS = [  # name, kind (f: float chain, u: uint chain, q: 64-bit chain), text, instructions per step
    ("v_fma_f32 v,v,v,v", "f", "v_fma_f32 {c}, {c}, {a}, {c}", 1),
    ("v_mul_f32 v,v,v", "f", "v_mul_f32 {c}, {c}, {a}", 1),
    ("v_add_f32 v,v,v", "f", "v_add_f32 {c}, {c}, {a}", 1),
    ("v_sub_f32 v,v,v", "f", "v_sub_f32 {c}, {c}, {a}", 1),
    ("v_fmac_f32 v,v,v", "f", "v_fmac_f32 {c}, {a}, {a}", 1),
    ("v_fmac_f32 c,a,b (three distinct VGPRs)", "f", "v_fmac_f32 {c}, {a}, %10", 1),
    ("v_fmac_f32 c,c,a", "f", "v_fmac_f32 {c}, {c}, {a}", 1),
    ("v_fma_f32 c,a,b,c (three distinct VGPRs)", "f", "v_fma_f32 {c}, {a}, %10, {c}", 1),
    ("v_mul_f32 c,a,a (one VGPR twice)", "f", "v_mul_f32 {c}, {a}, {a}", 1),
    ("v_mul_f32 c,a,b (dst not a source)", "f", "v_mul_f32 {c}, {a}, %10", 1),
    ("v_max_f32 c,a,b (dst not a source)", "f", "v_max_f32 {c}, {a}, %10", 1),
    ("v_fma_f32 v,s,v (SGPR operand)", "f", "v_fma_f32 {c}, {c}, s20, {c}", 1),
    ("v_fmac_f32 v,s,v (SGPR operand)", "f", "v_fmac_f32 {c}, s20, {a}", 1),
    ("v_mul_f32 v,s,v (SGPR operand)", "f", "v_mul_f32 {c}, s20, {c}", 1),
    ("v_mul_f32 v,2.0,v (inline constant)", "f", "v_mul_f32 {c}, 2.0, {c}", 1),
    ("v_mul_f32 v,literal,v", "f", "v_mul_f32 {c}, 0x40490fdb, {c}", 1),
    ("v_fma_f32 v,-v,v,v (neg modifier)", "f", "v_fma_f32 {c}, -{c}, {a}, {c}", 1),
    ("v_fma_f32 v,v,v,1.0 (inline constant)", "f", "v_fma_f32 {c}, {c}, {a}, 1.0", 1),
    ("v_mul_f32_e64 v,|v|,v (abs modifier)", "f", "v_mul_f32_e64 {c}, |{c}|, {a}", 1),
    ("v_max_f32", "f", "v_max_f32 {c}, {c}, {a}", 1),
    ("v_min_f32", "f", "v_min_f32 {c}, {c}, {a}", 1),
    ("v_med3_f32", "f", "v_med3_f32 {c}, {c}, {a}, {a}", 1),
    ("v_mov_b32", "f", "v_mov_b32 {c}, {a}", 1),
    ("v_mov_b32 v,s (SGPR source)", "f", "v_mov_b32 {c}, s20", 1),
    ("v_mov_b32 v,literal", "f", "v_mov_b32 {c}, 0x40490fdb", 1),
    ("v_max3_f32", "f", "v_max3_f32 {c}, {c}, {a}, %10", 1),
    ("v_min3_f32", "f", "v_min3_f32 {c}, {c}, {a}, %10", 1),
    ("v_add_u32 v,s,v (SGPR operand)", "u", "v_add_u32 {c}, s20, {c}", 1),
    ("v_and_b32 v,literal,v", "u", "v_and_b32 {c}, 0xffff00ff, {c}", 1),
    ("v_lshl_or_b32", "u", "v_lshl_or_b32 {c}, {c}, 3, {a}", 1),
    ("v_and_or_b32", "u", "v_and_or_b32 {c}, {c}, {a}, {a}", 1),
    ("v_ashrrev_i32", "u", "v_ashrrev_i32 {c}, 3, {c}", 1),
    ("v_cvt_f32_ubyte0", "f", "v_cvt_f32_ubyte0 {c}, {c}", 1),
    ("v_cvt_f32_i32", "f", "v_cvt_f32_i32 {c}, {c}", 1),
    ("v_rndne_f32", "f", "v_rndne_f32 {c}, {c}", 1),
    ("v_exp_f32", "f", "v_exp_f32 {c}, {c}", 1),
    ("v_log_f32", "f", "v_log_f32 {c}, {c}", 1),
    ("v_mul_legacy_f32", "f", "v_mul_legacy_f32 {c}, {c}, {a}", 1),
    ("v_cmp_lt_u32 vcc", "u", "v_cmp_lt_u32 vcc, {c}, {a}", 1),
    ("v_cmp_class_f32", "f", "v_cmp_class_f32 vcc, {c}, {a}", 1),
    ("v_add_co_u32 + v_addc_co_u32 (2 instr)", "u", "v_add_co_u32 {c}, vcc, {c}, {a}\\n v_addc_co_u32 {c}, vcc, {c}, {a}, vcc", 2),
    ("v_lshlrev_b64", "q", "v_lshlrev_b64 {c}, 3, {c}", 1),
    ("v_lshl_add_u64", "q", "v_lshl_add_u64 {c}, {c}, 3, {a}", 1),
    ("v_mbcnt_lo_u32_b32", "u", "v_mbcnt_lo_u32_b32 {c}, {a}, {c}", 1),
    ("v_add_u32", "u", "v_add_u32 {c}, {c}, {a}", 1),
    ("v_sub_u32", "u", "v_sub_u32 {c}, {c}, {a}", 1),
    ("v_xor_b32", "u", "v_xor_b32 {c}, {c}, {a}", 1),
    ("v_and_b32", "u", "v_and_b32 {c}, {c}, {a}", 1),
    ("v_or_b32", "u", "v_or_b32 {c}, {c}, {a}", 1),
    ("v_lshlrev_b32 v,1,v", "u", "v_lshlrev_b32 {c}, 1, {c}", 1),
    ("v_lshrrev_b32 v,3,v", "u", "v_lshrrev_b32 {c}, 3, {c}", 1),
    ("v_lshl_add_u32", "u", "v_lshl_add_u32 {c}, {c}, 3, {a}", 1),
    ("v_add3_u32", "u", "v_add3_u32 {c}, {c}, {a}, {a}", 1),
    ("v_xad_u32", "u", "v_xad_u32 {c}, {c}, {a}, {a}", 1),
    ("v_alignbit_b32", "u", "v_alignbit_b32 {c}, {c}, {a}, 5", 1),
    ("v_bfe_u32", "u", "v_bfe_u32 {c}, {c}, 3, 20", 1),
    ("v_bitop3_b32", "u", "v_bitop3_b32 {c}, {c}, {a}, {a} bitop3:0x96", 1),
    ("v_mul_lo_u32", "u", "v_mul_lo_u32 {c}, {c}, {a}", 1),
    ("v_mul_hi_u32", "u", "v_mul_hi_u32 {c}, {c}, {a}", 1),
    ("v_mul_u32_u24", "u", "v_mul_u32_u24 {c}, {c}, {a}", 1),
    ("v_mad_u32_u24", "u", "v_mad_u32_u24 {c}, {c}, {a}, {a}", 1),
    ("v_mad_u64_u32", "q", "v_mad_u64_u32 {c}, vcc, %9, %9, {c}", 1),
    ("v_rcp_f32", "f", "v_rcp_f32 {c}, {c}", 1),
    ("v_sqrt_f32", "f", "v_sqrt_f32 {c}, {c}", 1),
    ("v_rsq_f32", "f", "v_rsq_f32 {c}, {c}", 1),
    ("v_sin_f32", "f", "v_sin_f32 {c}, {c}", 1),
    ("v_div_scale_f32", "f", "v_div_scale_f32 {c}, vcc, {c}, {a}, {c}", 1),
    ("v_div_fmas_f32", "f", "v_div_fmas_f32 {c}, {c}, {a}, {c}", 1),
    ("v_div_fixup_f32", "f", "v_div_fixup_f32 {c}, {c}, {a}, {c}", 1),
    ("v_cmp_gt_f32 vcc", "f", "v_cmp_gt_f32 vcc, {c}, {a}", 1),
    ("v_cmp_gt_f32_e64 s[20:21]", "f", "v_cmp_gt_f32_e64 s[20:21], {c}, {a}", 1),
    ("v_cndmask_b32 (vcc)", "f", "v_cndmask_b32 {c}, {c}, {a}, vcc", 1),
    ("v_cndmask_b32_e64 (s[22:23])", "f", "v_cndmask_b32_e64 {c}, {c}, {a}, s[22:23]", 1),
    ("v_cmp + v_cndmask (2 instr)", "f", "v_cmp_gt_f32 vcc, {c}, {a}\\n v_cndmask_b32 {c}, {c}, {a}, vcc", 2),
    ("v_cvt_f32_u32", "f", "v_cvt_f32_u32 {c}, {c}", 1),
    ("v_cvt_u32_f32", "f", "v_cvt_u32_f32 {c}, {c}", 1),
    ("v_floor_f32", "f", "v_floor_f32 {c}, {c}", 1),
    ("v_fract_f32", "f", "v_fract_f32 {c}, {c}", 1),
    ("v_ldexp_f32", "f", "v_ldexp_f32 {c}, {c}, {a}", 1),
    ("v_readlane_b32", "u", "v_readlane_b32 s24, {c}, 3", 1),
    ("v_readfirstlane_b32", "u", "v_readfirstlane_b32 s24, {c}", 1),
    ("v_pk_fma_f32", "q", "v_pk_fma_f32 {c}, {c}, {a}, {c}", 1),
    ("v_pk_mul_f32", "q", "v_pk_mul_f32 {c}, {c}, {a}", 1),
    ("v_pk_add_f32", "q", "v_pk_add_f32 {c}, {c}, {a}", 1),
    ("v_pk_fma_f32 v,s[20:21],v,v (SGPR pair operand)", "q", "v_pk_fma_f32 {c}, s[20:21], {a}, {c}", 1),
    ("v_pk_fma_f32 v,s[20:21],v,v op_sel_hi:[0,1,1]", "q", "v_pk_fma_f32 {c}, s[20:21], {a}, {c} op_sel_hi:[0,1,1]", 1),
    ("v_pk_fma_f32 v,v,v,v op_sel:[0,1,0] op_sel_hi:[1,1,1] (VGPR broadcast)", "q", "v_pk_fma_f32 {c}, {c}, {a}, {c} op_sel:[0,1,0] op_sel_hi:[1,1,1]", 1),
    ("v_pk_mul_f32 v,v,s[20:21]", "q", "v_pk_mul_f32 {c}, {c}, s[20:21]", 1),
    ("v_mov_b64 v,s[20:21]", "q", "v_mov_b64 {c}, s[20:21]", 1),
    ("v_mov_b64 v,v", "q", "v_mov_b64 {c}, {a}", 1),
    ("v_fma_f64", "q", "v_fma_f64 {c}, {c}, {a}, {c}", 1),
    ("v_mul_f64", "q", "v_mul_f64 {c}, {c}, {a}", 1),
    ("v_add_f64", "q", "v_add_f64 {c}, {c}, {a}", 1),
    ("v_mov_b32 dpp quad_perm", "f", "v_mov_b32_dpp {c}, {a} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", 1),
    ("v_fma, v_max alternating", "f", "v_fma_f32 {c}, {c}, {a}, {c}\\n v_max_f32 {c}, {c}, {a}", 2),
    ("v_fma, v_fma(SGPR) alternating", "f", "v_fma_f32 {c}, {c}, {a}, {c}\\n v_fma_f32 {c}, {c}, s20, {c}", 2),
    ("v_fma + s_mov_b32 (VALU, SALU alternating; per pair)", "f", "v_fma_f32 {c}, {c}, {a}, {c}\\n s_mov_b32 s25, s20", 1),
    ("v_fma + s_and_b64 (VALU, SALU alternating; per pair)", "f", "v_fma_f32 {c}, {c}, {a}, {c}\\n s_and_b64 s[26:27], s[22:23], s[22:23]", 1),
    ("v_fma_f32 (ONE dependent chain)", "f", None, 1),
]
HEAD = r'''// GENERATED by gen_valu_rate.py -- issue rate of the VALU instruction forms the shading kernels are made of (MI355X, wave64).
// Every wave runs ITER x 64 instructions of one form on 8 independent register chains (one asm block per loop body); blocks of 64
// threads, W waves per SIMD resident (grid = 256 CUs x 4 SIMDs x W).  Prints NOMINAL cycles (2.4 GHz) per wave-instruction per SIMD
// = wall time x clock / instructions issued on one SIMD; best of 5 launches after a warm-up that brings the clocks up.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
template <int OP> __global__ __launch_bounds__(64) void k(float *out, float a, uint32_t ua, int iters) {
    float f[8]; uint32_t u[8]; uint64_t q[8]; float b = a * 1.5f + threadIdx.x; uint64_t qa = ((uint64_t) ua << 32) | 0x3f800100u;
    for (int i = 0; i < 8; ++i) { f[i] = a + threadIdx.x + i; u[i] = ua + threadIdx.x * 7 + i; q[i] = ((uint64_t) __float_as_uint(f[i]) << 32) | __float_as_uint(f[i] + 1.f); }
    asm volatile("s_mov_b32 s20, 0x3f800100\n s_mov_b32 s21, 0x3f800200\n s_mov_b64 s[22:23], 0x5555\n v_cmp_gt_f32 vcc, %0, %1" : : "v"(f[0]), "v"(a) : "s20", "s22", "s23", "vcc");
    for (int it = 0; it < iters; ++it) {
'''
TAIL = r'''    }
    float s = 0; for (int i = 0; i < 8; ++i) s += f[i] + (float) u[i] + (float) q[i];
    if (s == 1.2345f) out[threadIdx.x] = s + a + (float) ua + (float) qa;
}
template <int OP> double run(int waves_per_simd, float *d, double clock_hz, int per_step) {
    const int grid = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<grid, 64>>>(d, 1.0001f, 3u, 16);
    hipDeviceSynchronize();
    float ms = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0); k<OP><<<grid, 64>>>(d, 1.0001f, 3u, ITER); hipEventRecord(e1); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1); if (t < ms) ms = t;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1e-3 * clock_hz / ((double) waves_per_simd * ITER * 64.0 * per_step);
}
int main() {
    float *d; hipMalloc(&d, 4096);
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double hz = khz * 1e3;
    for (int i = 0; i < 100; ++i) k<0><<<4096, 64>>>(d, 1.0001f, 3u, ITER);   // clocks up before anything is timed
    hipDeviceSynchronize();
    printf("clock %.0f MHz nominal; NOMINAL cycles per wave64 instruction per SIMD at W resident waves per SIMD\n", hz / 1e6);
    printf("%-54s %7s %7s %7s %7s\n", "instruction form", "W=1", "W=2", "W=4", "W=8");
'''
def body(kind, text):
    regs = {"f": "f", "u": "u", "q": "q"}[kind]
    if text is None:
        lines = ["v_fma_f32 %0, %0, %8, %0"] * 64
    else:
        lines = [text.replace("{c}", "%%%d" % (j % 8)).replace("{a}", "%8") for j in range(64)]
    second = {"f": "a", "u": "ua", "q": "qa"}[kind]
    ops = ", ".join('"+v"(%s[%d])' % (regs, i) for i in range(8))
    return '            asm volatile("%s" : %s : "v"(%s), "v"(ua), "v"(b) : "vcc", "s20", "s21", "s24", "s25", "s26", "s27");' % ("\\n ".join(lines), ops, second)
out = [HEAD]
for i, (name, kind, text, n) in enumerate(OPS):
    out.append("        if (OP == %d) {\n%s\n        }\n" % (i, body(kind, text)))
out.append(TAIL)
for i, (name, kind, text, n) in enumerate(OPS):
    out.append('    { printf("%%-54s", "%s"); for (int w : {1, 2, 4, 8}) printf(" %%7.2f", run<%d>(w, d, hz, %d)); printf("\\n"); fflush(stdout); }\n' % (name, i, n))
out.append("    return 0;\n}\n")
open(__file__.replace("gen_valu_rate.py", "valu_rate.hip"), "w").write("".join(out))
