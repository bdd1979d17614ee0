// gather_nodes.hip -- what a divergent BVH node step costs on MI355X when the nodes come from the vector L1 / L2 (global loads)
// versus from LDS: every lane follows its own pseudo-random chain through a table of 64-byte records (the TLAS node format:
// LOADS x 16-byte loads per step, the next index depends on all of them), with FILL dependent VALU instructions per step standing in
// for the slab tests.  Blocks of BLOCK threads, `waves` waves resident per CU (grid sized to one round of resident blocks).
// Prints ns and shader cycles per node step per wave and the CU-wide node steps per cycle.
// Build: hipcc --offload-arch=gfx950 -O3 gather_nodes.hip -o gather_nodes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>

template <int SRC, int LOADS, int FILL, int ACTIVE>   // SRC 0: global, 1: LDS copy of the table (node-major), 2: LDS copy in four planes (piece k of node i at k * n + i);  ACTIVE: lanes out of 64 that work
__global__ void k(const uint4 *table, uint32_t mask, int steps, uint32_t table_vec4, float *out) {
    extern __shared__ uint4 lds[];
    const uint4 *t = table;
    if (SRC == 1) {
        for (uint32_t i = threadIdx.x; i < table_vec4; i += blockDim.x) lds[i] = table[i];
        __syncthreads();
        t = lds;
    }
    if (SRC == 2) {
        for (uint32_t i = threadIdx.x; i < table_vec4; i += blockDim.x) lds[(i & 3u) * (table_vec4 / 4) + (i >> 2)] = table[i];
        __syncthreads();
    }
    const uint32_t plane = table_vec4 / 4;
    if ((threadIdx.x & 63) >= ACTIVE) return;
    uint32_t idx = (blockIdx.x * 2654435761u + threadIdx.x * 40503u) & mask;
    float acc = 1.f + threadIdx.x;
    for (int s = 0; s < steps; ++s) {
        const uint4 *n = SRC == 2 ? lds + idx : t + (size_t) idx * 4;
        const uint32_t st = SRC == 2 ? plane : 1u;
        uint4 a = n[0], b = LOADS > 1 ? n[st] : a, c = LOADS > 2 ? n[2 * st] : a, d = LOADS > 3 ? n[3 * st] : a;
        // every word of every piece is consumed, so the loads stay 16 bytes wide; FILL instructions in four independent chains (the slab tests have ILP)
        const uint32_t h = (a.x ^ a.y ^ a.z ^ a.w) + (b.x ^ b.y ^ b.z ^ b.w) + (c.x ^ c.y ^ c.z ^ c.w) + (d.x ^ d.y ^ d.z ^ d.w);
        float f = __uint_as_float((h & 0x007fffffu) | 0x3f800000u);
        float c0 = acc, c1 = f, c2 = acc + 1.f, c3 = f + 1.f;
#pragma unroll
        for (int i = 0; i < FILL / 4; ++i) { c0 = __builtin_fmaf(c0, f, 0.5f); c1 = __builtin_fmaf(c1, f, 0.25f); c2 = __builtin_fmaf(c2, f, 0.125f); c3 = __builtin_fmaf(c3, f, 0.75f); }
        acc = (c0 + c1) + (c2 + c3);
        idx = (h ^ (uint32_t) s) & mask;
    }
    if (acc == 1.2345f) out[threadIdx.x] = acc + idx;
    if (idx == 0xfffffffeu) out[0] = 1.f;
}

template <int SRC, int LOADS, int FILL, int ACTIVE>
void run(const char *name, const uint4 *d_table, uint32_t n_nodes, int block, int waves_per_cu, float *d_out, double clock_hz) {
    const int blocks_per_cu = waves_per_cu * 64 / block;
    if (blocks_per_cu < 1) return;
    const size_t lds = SRC != 0 ? (size_t) n_nodes * 64 : 0;
    if (lds * blocks_per_cu > 160 * 1024) { printf("%-34s block %4d waves/CU %2d: table does not fit LDS %d times\n", name, block, waves_per_cu, blocks_per_cu); return; }
    if (lds > 64 * 1024) hipFuncSetAttribute((const void *) k<SRC, LOADS, FILL, ACTIVE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    // pad LDS so that exactly blocks_per_cu blocks are resident per CU
    size_t pad = (160 * 1024) / blocks_per_cu; pad -= pad % 1024; if (pad < lds) pad = lds;
    if (pad > 64 * 1024) hipFuncSetAttribute((const void *) k<SRC, LOADS, FILL, ACTIVE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) pad);
    const int grid = 256 * blocks_per_cu, steps = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SRC, LOADS, FILL, ACTIVE><<<grid, block, pad>>>(d_table, n_nodes - 1, 64, n_nodes * 4, d_out);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed: %s\n", name, hipGetErrorString(hipGetLastError())); return; }
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0); k<SRC, LOADS, FILL, ACTIVE><<<grid, block, pad>>>(d_table, n_nodes - 1, steps, n_nodes * 4, d_out); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double cyc = best * 1e-3 * clock_hz;                       // wall cycles of the launch
    const double steps_per_cu = (double) waves_per_cu * steps;      // wave-level node steps one CU executed
    printf("%-34s block %4d waves/CU %2d table %5u KB: %7.3f ms  %6.1f cyc per wave-step per CU (= 1 / CU throughput), %6.0f cyc latency per step of one wave\n",
           name, block, waves_per_cu, n_nodes * 64 / 1024, best, cyc / steps_per_cu, cyc / steps);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double clock_hz = p.clockRate * 1e3;
    printf("%s, %d CUs, clock %.0f MHz\n", p.name, p.multiProcessorCount, clock_hz / 1e6);
    float *d_out; hipMalloc(&d_out, 4096);
    for (uint32_t kb : { 16u, 64u, 1024u }) {
        const uint32_t n_nodes = kb * 1024 / 64;
        std::vector<uint32_t> h((size_t) n_nodes * 16);
        uint32_t s = 12345u; for (auto &w : h) { s = s * 1664525u + 1013904223u; w = s >> 3; }
        uint4 *d; hipMalloc(&d, h.size() * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int waves : { 4, 8, 16, 32 }) {
            run<0, 4, 0, 64>("global 4x16B, no VALU", d, n_nodes, 256, waves, d_out, clock_hz);
            run<0, 4, 55, 64>("global 4x16B, 55 VALU", d, n_nodes, 256, waves, d_out, clock_hz);
        }
        for (int waves : { 8, 16 }) {
            run<0, 4, 55, 32>("global 4x16B, 55 VALU, 32 lanes", d, n_nodes, 256, waves, d_out, clock_hz);
            run<0, 2, 55, 64>("global 2x16B, 55 VALU", d, n_nodes, 256, waves, d_out, clock_hz);
            run<0, 1, 55, 64>("global 1x16B, 55 VALU", d, n_nodes, 256, waves, d_out, clock_hz);
            run<0, 2, 0, 64>("global 2x16B, no VALU", d, n_nodes, 256, waves, d_out, clock_hz);
            run<0, 1, 0, 64>("global 1x16B, no VALU", d, n_nodes, 256, waves, d_out, clock_hz);
        }
        if (kb <= 64) {
            for (int waves : { 4, 8, 16, 32 }) {
                const int block = kb == 64 ? (waves <= 8 ? 256 : waves * 32) : 256;   // 64 KB: at most two copies per CU
                run<1, 4, 0, 64>("LDS 4x16B, no VALU", d, n_nodes, block, waves, d_out, clock_hz);
                run<1, 4, 55, 64>("LDS 4x16B, 55 VALU", d, n_nodes, block, waves, d_out, clock_hz);
                run<1, 4, 55, 32>("LDS 4x16B, 55 VALU, 32 lanes", d, n_nodes, block, waves, d_out, clock_hz);
                run<2, 4, 0, 64>("LDS planes 4x16B, no VALU", d, n_nodes, block, waves, d_out, clock_hz);
                run<2, 4, 55, 64>("LDS planes 4x16B, 55 VALU", d, n_nodes, block, waves, d_out, clock_hz);
                run<2, 4, 55, 32>("LDS planes 4x16B, 55 VALU, 32 lanes", d, n_nodes, block, waves, d_out, clock_hz);
            }
        }
        hipFree(d);
    }
    return 0;
}
