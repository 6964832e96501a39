// nm_distance.h -- inverse-distance weights and the indicator-blended projected signed distance
// of a query point w.r.t. its K nearest mesh vertices, plus the closed-form d ds / d xyz.
//
// Reference: models/mesh_grid.py:121-142 (MeshGrid.compute_distance_frnn after the FRNN call):
//     dis = sqrt(d2); w = 1/(dis+1e-7); w /= sum(w)                      (:123-125)
//     dir_k = x - v_k; r_k = ||dir_k||                                    (:134-135)
//     mid_k = (n_k*w1 + dir_k*r_k) / (w1 + r_k)                           (:136)
//     ds = sum_k w_k * (dir_k . mid_k)                                    (:137-142)
// d ds / d xyz with (indices, weights) constant -- they are detached at :121-122 -- is
//     sum_k w_k * [ (w1 n_k + 3 r_k dir_k)(w1 + r_k) - (w1 a_k + r_k^3) u_k ] / (w1 + r_k)^2,
//     a_k = dir_k . n_k, u_k = dir_k / r_k (0 where r_k = 0: torch.norm's subgradient).
// Shared by the device kernels and the host logic check (tests/hostcheck).
#pragma once

#include "nm_grid.h"

NM_HD float nm_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fsqrt_rn(x);
#else
    return std::sqrt(x);
#endif
}
NM_HD float nm_div(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fdiv_rn(a, b);
#else
    return a / b;
#endif
}

// inverse-distance weights of the K = 8 neighbours (mesh_grid.py:121-124), normalised
NM_HD void nm_weights8(const float (&d2)[8], float (&w_out)[8]) {
    float wsum = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float dis = nm_sqrt(d2[k]);
        w_out[k] = nm_div(1.0f, nm_add(dis, 1e-7f));
        wsum = nm_add(wsum, w_out[k]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) w_out[k] = nm_div(w_out[k], wsum);
}

// projected signed distance from normalised weights (and its gradient in closed form)
NM_HD float nm_projected_from_weights8(float qx, float qy, float qz, const int (&idx)[8], const float (&w_out)[8],
                                       const float* __restrict__ verts, const float* __restrict__ indicator,
                                       float w1, float* grad) {
    float ds = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = idx[k];
        const float vx = verts[3 * i], vy = verts[3 * i + 1], vz = verts[3 * i + 2];
        const float nx = indicator[3 * i], ny = indicator[3 * i + 1], nz = indicator[3 * i + 2];
        const float dx = nm_sub(qx, vx), dy = nm_sub(qy, vy), dz = nm_sub(qz, vz);
        const float r = nm_sqrt(nm_add(nm_add(nm_mul(dx, dx), nm_mul(dy, dy)), nm_mul(dz, dz)));
        const float den = nm_add(w1, r);
        const float mx = nm_div(nm_add(nm_mul(nx, w1), nm_mul(dx, r)), den);
        const float my = nm_div(nm_add(nm_mul(ny, w1), nm_mul(dy, r)), den);
        const float mz = nm_div(nm_add(nm_mul(nz, w1), nm_mul(dz, r)), den);
        const float f = nm_add(nm_add(nm_mul(dx, mx), nm_mul(dy, my)), nm_mul(dz, mz));
        ds = nm_add(ds, nm_mul(w_out[k], f));
        if (grad) {
            const float a = nm_add(nm_add(nm_mul(dx, nx), nm_mul(dy, ny)), nm_mul(dz, nz));
            const float inv_r = r > 0.f ? nm_div(1.0f, r) : 0.f;
            const float ux = dx * inv_r, uy = dy * inv_r, uz = dz * inv_r;
            const float c3 = 3.0f * r * r;               // d(r^3)/dx = 3 r^2 u
            const float tail = w1 * a + r * r * r;
            const float inv_den2 = nm_div(1.0f, den * den);
            gx += w_out[k] * ((w1 * nx + c3 * ux) * den - tail * ux) * inv_den2;
            gy += w_out[k] * ((w1 * ny + c3 * uy) * den - tail * uy) * inv_den2;
            gz += w_out[k] * ((w1 * nz + c3 * uz) * den - tail * uz) * inv_den2;
        }
    }
    if (grad) {
        grad[0] = gx;
        grad[1] = gy;
        grad[2] = gz;
    }
    return ds;
}

// K = 8 neighbours.  verts/indicator: [V,3] in ORIGINAL vertex order.
// w_out[8] normalised weights; returns ds; grad (may be nullptr) receives d ds / d xyz.
NM_HD float nm_projected_distance8(float qx, float qy, float qz, const float (&d2)[8], const int (&idx)[8],
                                   const float* __restrict__ verts, const float* __restrict__ indicator,
                                   float w1, float (&w_out)[8], float* grad) {
    nm_weights8(d2, w_out);
    return nm_projected_from_weights8(qx, qy, qz, idx, w_out, verts, indicator, w1, grad);
}
