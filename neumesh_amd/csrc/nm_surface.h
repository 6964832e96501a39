// nm_surface.h -- first-hit surface points of a ray batch: the per-ray bookkeeping kernels of nm_surface_hits.
//
// Reference: models/ray_casting.py:45-200 (root_finding_surface_points) with run_secant_method (:12-38):
//   N_steps proposals d_j = near (1 - t_j) + far t_j, t = linspace(0, 1, N_steps); val_j = sdf(o + d_j dir) - tau;
//   the FIRST j with val_j val_{j+1} < 0 (the minimum of sign(val_j val_{j+1}) (N_steps - j), :106-117) is the sign change;
//   a hit needs val_j > 0 there (outside -> inside) and val_0 > 0; the root is refined by N_secant_steps regula-falsi
//   steps between (d_j, val_j) and (d_{j+1}, val_{j+1}).
// Everything the routine returns depends on the proposal values through that first sign change alone, so a ray leaves the
// walk after the block of proposals that holds it (the field is evaluated in blocks of NM_SURF_BLOCK proposals per ray on the
// K-NN + geometry-MLP kernels, over the compacted list of rays still walking).  Arithmetic: fp32, one rounding per
// operation in the reference's order (no FMA contraction), so depths and masks are the reference's value for value.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nm_rays.h"

#define NM_SURF_BLOCK 16   // proposals per ray and walk step (four 16-ray x 4-sample tiles of the distance kernel)

struct NmSurfState {      // per ray (original ray index)
    int* idx;             // j of the first sign change, -1: none yet
    float* f_high;        // val_j           (outside value)
    float* f_low;         // val_{j+1}
    float* d_high;        // d_j
    float* d_low;         // d_{j+1}
    float* val0;          // val_0
    float* prev;          // last proposal value of the previous block
};

__global__ void nm_surf_init_kernel(long long R, NmSurfState st, int* ids) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    st.idx[r] = -1;
    st.f_high[r] = st.f_low[r] = st.d_high[r] = st.d_low[r] = st.val0[r] = st.prev[r] = 0.f;
    if (ids) ids[r] = (int)r;
}

// One walk step: proposals [k0, k0 + n) of the rays ids[0 .. nA) have the field values val[a * n + p] (sdf, tau not yet
// subtracted).  Records a ray's first sign change and flags the rays that keep walking.
__global__ void nm_surf_scan_kernel(const int* __restrict__ ids, int nA, const float* __restrict__ val, int n, int k0, int N, float tau,
                                    const float* __restrict__ nearfar, float near_s, float far_s, NmSurfState st,
                                    unsigned char* __restrict__ keep) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= nA) return;
    const int r = ids[a];
    const float* v = val + (size_t)a * n;
    float prev = k0 > 0 ? st.prev[r] : 0.f;
    int found = -1;
    float fh = 0.f, fl = 0.f;
    for (int p = 0; p < n; ++p) {
        const float cur = nm_sub(v[p], tau);
        if (k0 + p == 0) st.val0[r] = cur;
        else if (found < 0 && nm_mul(prev, cur) < 0.f) {   // product of proposals k0 + p - 1 and k0 + p
            found = k0 + p - 1;
            fh = prev;
            fl = cur;
        }
        prev = cur;
    }
    st.prev[r] = prev;
    if (found >= 0) {
        const float nr = nearfar ? nearfar[2 * r] : near_s, fr = nearfar ? nearfar[2 * r + 1] : far_s;
        st.idx[r] = found;
        st.f_high[r] = fh;
        st.f_low[r] = fl;
        st.d_high[r] = nm_lerp_depth(nr, fr, nm_linspace01(found, N));
        st.d_low[r] = nm_lerp_depth(nr, fr, nm_linspace01(found + 1 < N ? found + 1 : N - 1, N));
    }
    keep[a] = found < 0 ? 1 : 0;
}

// hit = first sign change exists, goes outside -> inside, first proposal outside
__global__ void nm_surf_hit_flags_kernel(long long R, NmSurfState st, unsigned char* __restrict__ hit, int* __restrict__ ids) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    hit[r] = (st.idx[r] >= 0 && st.f_high[r] > 0.f && st.val0[r] > 0.f) ? 1 : 0;
    ids[r] = (int)r;
}

// d_pred = -f_low (d_high - d_low) / (f_high - f_low) + d_low  (ray_casting.py:23-25), point = o + d_pred dir (:29)
__device__ __forceinline__ float nm_surf_estimate(float f_low, float f_high, float d_low, float d_high) {
    return nm_add(nm_div(nm_mul(-f_low, nm_sub(d_high, d_low)), nm_sub(f_high, f_low)), d_low);
}
__global__ void nm_surf_secant_points_kernel(const int* __restrict__ ids, int nH, NmSurfState st, const float* __restrict__ rays_o,
                                             const float* __restrict__ dirn, float* __restrict__ d_pred, float* __restrict__ xyz) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= nH) return;
    const int r = ids[a];
    const float d = nm_surf_estimate(st.f_low[r], st.f_high[r], st.d_low[r], st.d_high[r]);
    d_pred[a] = d;
#pragma unroll
    for (int c = 0; c < 3; ++c) xyz[3 * a + c] = nm_add(rays_o[3 * (size_t)r + c], nm_mul(d, dirn[3 * (size_t)r + c]));
}
// f_mid < 0: the root lies before d_pred -> (d_low, f_low) <- (d_pred, f_mid); else (d_high, f_high) <- (ray_casting.py:31-36)
__global__ void nm_surf_secant_update_kernel(const int* __restrict__ ids, int nH, NmSurfState st, const float* __restrict__ d_pred,
                                             const float* __restrict__ sdf_mid, float tau) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= nH) return;
    const int r = ids[a];
    const float f_mid = nm_sub(sdf_mid[a], tau);
    if (f_mid < 0.f) {
        st.d_low[r] = d_pred[a];
        st.f_low[r] = f_mid;
    } else {
        st.d_high[r] = d_pred[a];
        st.f_high[r] = f_mid;
    }
}
// outputs (ray_casting.py:177-192): hits get the last estimate; the others d = inf / far (0 where the ray starts inside), point = 1
__global__ void nm_surf_finish_kernel(long long R, NmSurfState st, const unsigned char* __restrict__ hit, const float* __restrict__ rays_o,
                                      const float* __restrict__ dirn, const float* __restrict__ nearfar, float far_s, int refine, int fill_inf,
                                      float* __restrict__ d_out, float* __restrict__ pt_out, unsigned char* __restrict__ mask,
                                      unsigned char* __restrict__ mask_sign_change) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const bool h = hit[r] != 0;
    float d = 1.0f, px = 1.0f, py = 1.0f, pz = 1.0f;
    if (h) {
        d = refine ? nm_surf_estimate(st.f_low[r], st.f_high[r], st.d_low[r], st.d_high[r]) : 1.0f;
        px = nm_add(rays_o[3 * r], nm_mul(d, dirn[3 * r]));
        py = nm_add(rays_o[3 * r + 1], nm_mul(d, dirn[3 * r + 1]));
        pz = nm_add(rays_o[3 * r + 2], nm_mul(d, dirn[3 * r + 2]));
    } else {
        d = fill_inf ? __int_as_float(0x7f800000) : (nearfar ? nearfar[2 * r + 1] : far_s);
    }
    if (!(st.val0[r] > 0.f)) d = 0.f;   // the first proposal is already inside
    d_out[r] = d;
    pt_out[3 * r] = px;
    pt_out[3 * r + 1] = py;
    pt_out[3 * r + 2] = pz;
    mask[r] = h ? 1 : 0;
    mask_sign_change[r] = st.idx[r] >= 0 ? 1 : 0;
}
__global__ void nm_surf_gather_ids_kernel(const int* __restrict__ src, int n, int* __restrict__ dst) {  // (rocprim::select writes to another buffer)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
