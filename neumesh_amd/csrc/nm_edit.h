// nm_edit.h -- device-only: the texture-editing blend of nm_render_rays (reference: editing/texture_neumesh/texture_neumesh.py:53-122).
#pragma once

#include "nm_kernels.h"
#include "nm_mlp.h"

// ------------------------------------------------------------------------ texture editing
// TextureEditableNeuMesh.forward (editing/texture_neumesh/texture_neumesh.py:79-121) for one reference model, on the
// mid-point list of nm_render_rays.  Per point: painted_k = mask[idx_k]; on_paint = sum_k w_k painted_k, on_rest =
// sum_k w_k (1 - painted_k) (k ascending); the point is in the region when on_paint > 0; shares on_rest / total and
// on_paint / total; reference weights w_k painted_k / (sum + 1e-8); the edited colour table interpolated with them
// (nm_gather_interp: the arithmetic of the K-NN kernel's own gathers).  Points outside the region get a zero row (their
// reference colour is evaluated with the rest of the list and not used).
struct NmRot3 {
    float m[9];  // row-major
};
// transform_direction (utils/geo_util.py:78-89): rot @ v
__device__ __forceinline__ void nm_rotate3(const NmRot3& r, float x, float y, float z, float* out) {
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = __fadd_rn(__fadd_rn(__fmul_rn(r.m[3 * i], x), __fmul_rn(r.m[3 * i + 1], y)), __fmul_rn(r.m[3 * i + 2], z));
}
__global__ __launch_bounds__(256) void nm_rotate_rows3_kernel(long long n, NmRot3 r, const float* __restrict__ in, float* __restrict__ out) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) nm_rotate3(r, in[q * 3], in[q * 3 + 1], in[q * 3 + 2], out + q * 3);
}
// (nab_in != nullptr: the reference model lives in a rotated frame -- nab_out[q] = rot @ nab_in[q] for its colour call)
__global__ __launch_bounds__(256) void nm_edit_prepare_kernel(long long P, NmSlotMap smap, const int* __restrict__ idx32, const float* __restrict__ w,
                                                              const unsigned char* __restrict__ mask, const float* __restrict__ table, int cdim,
                                                              float* __restrict__ ref_w, float* __restrict__ share, float* __restrict__ ft_ref,
                                                              NmRot3 rot, const float* __restrict__ nab_in, float* __restrict__ nab_out) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = q < P && nm_slot_valid(smap, q);
    int bi[8];
    float wk[8], rw[8];
    float on_paint = 0.f, on_rest = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        bi[k] = valid ? idx32[q * 8 + k] : 0;
        wk[k] = valid ? w[q * 8 + k] : 0.f;
        const bool painted = valid && mask[bi[k]] != 0;
        rw[k] = painted ? wk[k] : 0.f;
        on_paint = __fadd_rn(on_paint, rw[k]);
        on_rest = __fadd_rn(on_rest, painted ? 0.f : wk[k]);
    }
    const bool region = valid && on_paint > 0.f;
    const float total = __fadd_rn(on_paint, on_rest), den = __fadd_rn(on_paint, 1e-8f);
#pragma unroll
    for (int k = 0; k < 8; ++k) rw[k] = region ? __fdiv_rn(rw[k], den) : 0.f;
    if (valid) {
        share[q * 2] = region ? __fdiv_rn(on_rest, total) : 1.f;
        share[q * 2 + 1] = region ? __fdiv_rn(on_paint, total) : 0.f;
        *reinterpret_cast<float4*>(ref_w + q * 8) = make_float4(rw[0], rw[1], rw[2], rw[3]);
        *reinterpret_cast<float4*>(ref_w + q * 8 + 4) = make_float4(rw[4], rw[5], rw[6], rw[7]);
        if (nab_in) nm_rotate3(rot, nab_in[q * 3], nab_in[q * 3 + 1], nab_in[q * 3 + 2], nab_out + q * 3);
        if (!region)
            for (int c = 0; c < cdim; c += 4) *reinterpret_cast<float4*>(ft_ref + q * cdim + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    nm_gather_interp(table, cdim, bi, rw, region, q, ft_ref);   // (the whole wave takes part)
}
// colour = colour * share_rest + reference colour * share_paint where the point is in the region (texture_neumesh.py:118-121)
__global__ __launch_bounds__(256) void nm_edit_blend_kernel(long long P, NmSlotMap smap, int Pmid, const float* __restrict__ share,
                                                            const float* __restrict__ rgb_ref, float* __restrict__ rgb) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= P || !nm_slot_valid(smap, q)) return;
    const float a_rest = share[q * 2], a_paint = share[q * 2 + 1];
    if (!(a_paint > 0.f)) return;
    long long oq = q;   // ordered lists: the colours live at their (ray, sample) position
    if (smap.order) {
        long long ray;
        int sp;
        nm_slot_ray(smap, q, (q / smap.E) * smap.G, ray, sp);
        oq = ray * smap.P + sp;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[oq * 3 + c] = __fadd_rn(__fmul_rn(rgb[oq * 3 + c], a_rest), __fmul_rn(rgb_ref[oq * 3 + c], a_paint));
}

