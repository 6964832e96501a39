// nm_kernels.h -- device kernels other than the MLPs: K-NN / projected-distance kernel,
// per-ray stage kernels, small utility kernels.  Device-only (included by nm_api.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nm_rays.h"

// Where the query points of a launch come from.
//   mode 0: explicit xyz[q][3]
//   mode 1: point (r, p) = rays_o[r] + depth[r*dstride + doff + p] * dirn[r]   (renderer.py:198,246,264,267)
//   mode 2: depth = near[r]*(1-t_p) + far[r]*t_p, t = linspace(0,1,P)          (renderer.py:79-86,193-198)
//           optionally stored to depth_out[r*dstride + doff + p]
struct NmPointSrc {
    int mode;
    int P;  // points per ray (modes 1,2)
    const float* xyz;
    const float* rays_o;
    const float* dirn;
    const float* depth;
    const float* nearfar;  // [R][2]
    float* depth_out;
    int dstride, doff;
};

__device__ __forceinline__ void nm_fetch_point(const NmPointSrc& s, long long q, float& x, float& y, float& z) {
    if (s.mode == 0) {
        x = s.xyz[q * 3];
        y = s.xyz[q * 3 + 1];
        z = s.xyz[q * 3 + 2];
        return;
    }
    const long long r = q / s.P;
    const int p = (int)(q - r * s.P);
    float d;
    if (s.mode == 1) {
        d = s.depth[r * s.dstride + s.doff + p];
    } else {
        d = nm_lerp_depth(s.nearfar[2 * r], s.nearfar[2 * r + 1], nm_linspace01(p, s.P));
        if (s.depth_out) s.depth_out[r * s.dstride + s.doff + p] = d;
    }
    x = nm_add(s.rays_o[3 * r], nm_mul(d, s.dirn[3 * r]));
    y = nm_add(s.rays_o[3 * r + 1], nm_mul(d, s.dirn[3 * r + 1]));
    z = nm_add(s.rays_o[3 * r + 2], nm_mul(d, s.dirn[3 * r + 2]));
}

// ----------------------------------------------------------------------------- plain K-NN
template <int K>
__global__ __launch_bounds__(256) void nm_knn_kernel(NmGridView g, NmPointSrc src, long long Q, int Kout,
                                                     long long* __restrict__ idx_out, float* __restrict__ d2_out) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    float x, y, z;
    nm_fetch_point(src, q, x, y, z);
    float bd[K];
    int bi[K];
    nm_knn_search<K>(g, x, y, z, bd, bi);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (k < Kout) {
            const bool ok = bi[k] != 0x7fffffff;
            idx_out[q * Kout + k] = ok ? (long long)bi[k] : -1ll;
            d2_out[q * Kout + k] = ok ? bd[k] : -1.0f;
        }
    }
}

// ------------------------------------------------- K-NN + weights + projected signed distance
// (models/mesh_grid.py:88-144 fused; nothing of shape [Q,8,3] is ever materialised)
// Any output pointer may be null.  ds_out is indexed by q (compact).
__global__ __launch_bounds__(256) void nm_distance_kernel(NmGridView g, NmPointSrc src, long long Q,
                                                          const float* __restrict__ verts,
                                                          const float* __restrict__ indicator, float w1,
                                                          float* __restrict__ ds_out, int* __restrict__ idx32_out,
                                                          long long* __restrict__ idx64_out,
                                                          float* __restrict__ w_out, float* __restrict__ grad_out) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    float x, y, z;
    nm_fetch_point(src, q, x, y, z);
    float bd[8], wk[8], gr[3];
    int bi[8];
    nm_knn_search<8>(g, x, y, z, bd, bi);
    const float ds = nm_projected_distance8(x, y, z, bd, bi, verts, indicator, w1, wk, grad_out ? gr : nullptr);
    if (ds_out) ds_out[q] = ds;
    if (idx32_out) {
        *reinterpret_cast<int4*>(idx32_out + q * 8) = make_int4(bi[0], bi[1], bi[2], bi[3]);
        *reinterpret_cast<int4*>(idx32_out + q * 8 + 4) = make_int4(bi[4], bi[5], bi[6], bi[7]);
    }
    if (idx64_out) {
#pragma unroll
        for (int k = 0; k < 8; ++k) idx64_out[q * 8 + k] = (long long)bi[k];
    }
    if (w_out) {
        *reinterpret_cast<float4*>(w_out + q * 8) = make_float4(wk[0], wk[1], wk[2], wk[3]);
        *reinterpret_cast<float4*>(w_out + q * 8 + 4) = make_float4(wk[4], wk[5], wk[6], wk[7]);
    }
    if (grad_out) {
        grad_out[q * 3] = gr[0];
        grad_out[q * 3 + 1] = gr[1];
        grad_out[q * 3 + 2] = gr[2];
    }
}

// ------------------------------------------------------------------------- per-ray kernels
__global__ void nm_rays_setup_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, long long R,
                                     float radius, float* __restrict__ dirn, float* __restrict__ nearfar) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    nm_ray_setup(rays_o + 3 * r, rays_d + 3 * r, radius, dirn + 3 * r, nearfar + 2 * r, nearfar + 2 * r + 1);
}

__global__ void nm_rays_bounds_kernel(const float* __restrict__ ds_probe, long long R, int G, float thresh,
                                      const float* __restrict__ nearfar0, float* __restrict__ nearfar) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    nm_ray_bounds(ds_probe + r * G, 1, G, thresh, nearfar0[2 * r], nearfar0[2 * r + 1], nearfar + 2 * r,
                  nearfar + 2 * r + 1);
}

__global__ void nm_rays_bypass_kernel(long long R, float near_bypass, float far_bypass, float* __restrict__ nearfar) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    if (near_bypass >= 0.f) nearfar[2 * r] = near_bypass;
    if (far_bypass >= 0.f) nearfar[2 * r + 1] = far_bypass;
}

// merge the m samples appended by the previous iteration, then draw n_new new ones
__global__ void nm_rays_upsample_kernel(float* __restrict__ d, float* __restrict__ sdf, long long R, int cap, int n,
                                        int m, int it, int n_new) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float* dr = d + r * cap;
    float* sr = sdf + r * cap;
    if (m > 0) nm_ray_merge(dr, sr, n - m, m);
    float w[NM_MAX_SAMPLES], cdf[NM_MAX_SAMPLES];
    nm_ray_upsample(dr, sr, n, it, n_new, dr + n, w, cdf);
}

// final merge + mid-point depths (renderer.py:255-258, :266)
__global__ void nm_rays_finalize_kernel(float* __restrict__ d, float* __restrict__ sdf, long long R, int cap, int n,
                                        int m, float* __restrict__ d_mid) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float* dr = d + r * cap;
    if (m > 0) nm_ray_merge(dr, sdf + r * cap, n - m, m);
    for (int j = 0; j + 1 < n; ++j) d_mid[r * cap + j] = nm_mul(0.5f, nm_add(dr[j + 1], dr[j]));
}

__global__ void nm_rays_composite_kernel(const float* __restrict__ sdf, const float* __restrict__ d, long long R,
                                         int cap, int N, float s, const float* __restrict__ rgb_mid,
                                         const float* __restrict__ nablas, int white_bkgd, float* __restrict__ rgb,
                                         float* __restrict__ depth, float* __restrict__ acc,
                                         float* __restrict__ normals) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float w[NM_MAX_SAMPLES];
    nm_ray_composite(sdf + r * cap, d + r * cap, N, s, rgb_mid + r * (long long)(N - 1) * 3,
                     nablas ? nablas + r * (long long)N * 3 : nullptr, white_bkgd, rgb + 3 * r, depth + r, acc + r,
                     normals ? normals + 3 * r : nullptr, w);
}

// ------------------------------------------------------------------------------- utilities
__global__ void nm_pack_weight_kernel(const float* __restrict__ src, int rows, int in_dim, int Kpad,
                                      float* __restrict__ dst) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * Kpad) return;
    const int n = e / Kpad, k = e % Kpad;
    dst[e] = k < in_dim ? src[(size_t)n * in_dim + k] : 0.f;
}

__global__ void nm_idx64_to_32_kernel(const long long* __restrict__ src, long long n, int* __restrict__ dst) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) dst[e] = (int)src[e];
}

__global__ void nm_copy_strided_kernel(const float* __restrict__ src, long long R, int n, int src_stride,
                                       float* __restrict__ dst) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= R * n) return;
    const long long r = e / n;
    dst[e] = src[r * src_stride + (e - r * n)];
}
