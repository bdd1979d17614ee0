// stream_pattern.hip -- what HBM delivers for the memory access pattern of the bounce kernel k_shade<MODE 1> WITHOUT its arithmetic:
// per lane read  u32 index + u32 hit id + five 16-byte records + one 8-byte record (112 B) from SoA arrays through the index,
//          write four 16-byte records + 16-byte hit + two u32 (104 B); 64 threads per block, 8 chunks per block like the kernel.
// Also a plain float4 copy for the device's streaming ceiling.  Prints GB/s of algorithmic bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Q { float4 *ray_a, *ray_b, *st_a, *res; uint4 *hit, *rng_a; uint2 *rng_b; uint32_t *hit_id, *qin, *qout; };

template <int CHUNKS, bool WRITE_ALL> __global__ __launch_bounds__(64) void k_pattern(Q q, uint32_t n) {
    for (int c = 0; c < CHUNKS; ++c) {
        uint32_t j = (blockIdx.x * CHUNKS + c) * 64 + threadIdx.x;
        if (j >= n) return;
        uint32_t l = q.qin[j];
        uint32_t hid = q.hit_id[l];
        float4 a = q.ray_a[l], b = q.ray_b[l], s = q.st_a[l], r = q.res[l]; uint4 h = q.hit[l], g = q.rng_a[l]; uint2 gi = q.rng_b[l];
        float x = a.x + b.y + s.z + r.w + __uint_as_float(h.x ^ g.y ^ gi.x ^ hid);
        q.ray_a[l] = make_float4(x, a.y, a.z, a.w); q.ray_b[l] = make_float4(b.x, x, b.z, b.w); q.st_a[l] = make_float4(s.x, s.y, x, s.w);
        q.rng_a[l] = make_uint4(g.x + 1, g.y, g.z, g.w); q.hit[l] = make_uint4(h.x, h.y + 1, h.z, h.w); q.hit_id[l] = hid + 1; q.qout[j] = l;
        if (WRITE_ALL || (g.x & 3) == 0) q.res[l] = make_float4(r.x, r.y, r.z, x);
    }
}
__global__ void k_copy(const float4 *a, float4 *b, size_t n) { size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; if (i < n) b[i] = a[i]; }
__global__ void k_read(const float4 *a, float *out, size_t n) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; float4 v = i < n ? a[i] : make_float4(0, 0, 0, 0);
    if (v.x + v.y + v.z + v.w == 1.2345f) out[0] = 1.f;
}
__global__ void k_write(float4 *b, size_t n) { size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; if (i < n) b[i] = make_float4(1, 2, 3, 4); }

int main() {
    const uint32_t n = 15u << 20;
    Q q; void *p;
#define AL(field, T) CK(hipMalloc(&p, (size_t) n * sizeof(T))); CK(hipMemset(p, 1, (size_t) n * sizeof(T))); q.field = (T *) p;
    AL(ray_a, float4) AL(ray_b, float4) AL(st_a, float4) AL(res, float4) AL(hit, uint4) AL(rng_a, uint4) AL(rng_b, uint2) AL(hit_id, uint32_t) AL(qin, uint32_t) AL(qout, uint32_t)
    std::vector<uint32_t> idx(n);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        // mode 0: identity index; mode 1: 7 % of the slots skipped (a compacted queue after one bounce: increasing, with gaps)
        uint32_t k = 0; for (uint32_t i = 0; i < n; ++i) { idx[i] = k; k += (mode == 1 && (i * 2654435761u >> 28) == 0) ? 2 : 1; if (k >= n) k = n - 1; }
        CK(hipMemcpy(q.qin, idx.data(), (size_t) n * 4, hipMemcpyHostToDevice));
        for (int variant = 0; variant < 2; ++variant) {
            float best = 1e9f;
            for (int it = 0; it < 8; ++it) {
                CK(hipEventRecord(e0));
                if (variant == 0) hipLaunchKernelGGL((k_pattern<8, true>), dim3((n + 511) / 512), dim3(64), 0, 0, q, n);
                else hipLaunchKernelGGL((k_pattern<8, false>), dim3((n + 511) / 512), dim3(64), 0, 0, q, n);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double bytes = (double) n * (112 + (variant == 0 ? 104 : 92));
            printf("pattern index=%s res-write=%s : %.3f ms  %.0f GB/s (algorithmic %d B/lane, %u lanes)\n", mode ? "gappy" : "identity", variant ? "25%" : "all", best, bytes / best / 1e6,
                   112 + (variant == 0 ? 104 : 92), n);
        }
    }
    const size_t m = (size_t) 64 << 20;   // 1 GiB of float4
    float4 *a, *b; float *o; CK(hipMalloc(&a, m * 16)); CK(hipMalloc(&b, m * 16)); CK(hipMalloc(&o, 4)); CK(hipMemset(a, 0, m * 16));
    for (int t = 0; t < 3; ++t) {
        float best = 1e9f;
        for (int it = 0; it < 8; ++it) {
            CK(hipEventRecord(e0));
            if (t == 0) hipLaunchKernelGGL(k_copy, dim3((unsigned) (m / 256)), dim3(256), 0, 0, a, b, m);
            else if (t == 1) hipLaunchKernelGGL(k_read, dim3((unsigned) (m / 256)), dim3(256), 0, 0, a, o, m);
            else hipLaunchKernelGGL(k_write, dim3((unsigned) (m / 256)), dim3(256), 0, 0, b, m);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%s 1 GiB float4: %.3f ms  %.0f GB/s\n", t == 0 ? "copy (read+write)" : t == 1 ? "read " : "write", best, (t == 0 ? 2.0 : 1.0) * m * 16 / best / 1e6);
    }
    return 0;
}
