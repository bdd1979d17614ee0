// dtof_shade_spec2.hip -- instantiations of k_shade (dtof_shade.h): every BSDF + blendbsdf / two-BSDF twosided (SPEC = 2: the BSDF chain loops over two records).
#include "dtof_shade.h"

namespace dtof {

void launch_shade_spec2(bool k4, const ShadeLaunch &L) {
    if (k4) launch_shade_variant<true, kMaxOffsets, true, 2>(L); else launch_shade_variant<true, 1, true, 2>(L);
}

// BSDF::eval_pdf_sample over arrays (dtof_bsdf_eval): the function the shade kernels call at every path vertex, on a flat local frame (wi and wo are given in
// it; uv = the given texture coordinates).  in: wi[3], wo[3], sample1, sample2[2], u, v (11 floats); out: value * cos[3], pdf, sampled wo[3], sample pdf, eta,
// delta flag, weight[3], null flag (14 floats)
__global__ void k_bsdf_eval(const uint8_t *scene, uint32_t shape_index, const float *in, float *out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SceneView sv = make_view(scene);
    const float *a = in + (size_t) i * 11;
    Surface si;
    si.p = mk(0, 0, 0); si.n = mk(0, 0, 1); si.sh_n = mk(0, 0, 1); si.sh_s = mk(1, 0, 0); si.sh_t = mk(0, 1, 0); si.dp_du = mk(1, 0, 0); si.dp_dv = mk(0, 1, 0);
    si.wi = mk(a[0], a[1], a[2]); si.u = a[9]; si.v = a[10]; si.shape = &sv.shapes[shape_index];
    BsdfOut bo;
    bsdf_eval_pdf_sample<2>(sv, si.shape, si, mk(a[3], a[4], a[5]), true, a[6], a[7], a[8], bo);
    float *w = out + (size_t) i * 14;
    w[0] = bo.val.x; w[1] = bo.val.y; w[2] = bo.val.z; w[3] = bo.pdf; w[4] = bo.wo.x; w[5] = bo.wo.y; w[6] = bo.wo.z; w[7] = bo.bs_pdf; w[8] = bo.bs_eta;
    w[9] = bo.bs_delta ? 1.f : 0.f; w[10] = bo.weight.x; w[11] = bo.weight.y; w[12] = bo.weight.z; w[13] = bo.bs_null ? 1.f : 0.f;
}
void launch_bsdf_eval(const uint8_t *scene, uint32_t shape_index, const float *in, float *out, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_bsdf_eval, dim3(nblk(n)), dim3(kBlock), 0, s, scene, shape_index, in, out, n);
}

}  // namespace dtof
