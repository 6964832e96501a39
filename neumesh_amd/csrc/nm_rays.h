// nm_rays.h -- per-ray stages of the volumetric renderer (everything in render_rayschunk that is
// not a field query).  Each function handles ONE ray serially; on the device one lane owns one
// ray (the work is a few thousand flops per ray per stage, <1 % of the MLP time, and the
// reference's scans are order-sensitive: torch's CPU cumsum/cumprod run sequentially with a
// float64 accumulator, which a lane-serial loop reproduces exactly).  Shared with the host-side
// logic check in tests/hostcheck.
//
// Reference (file:line in the NeuMesh tree):
//   F.normalize(rays_d)                         models/renderer.py:153
//   near_far_from_sphere                        utils/rend_util.py:179-199
//   compute_bounded_near_far (reduction part)   models/renderer.py:88-102
//   coarse depths near*(1-t)+far*t              models/renderer.py:193-194 (and :79-80)
//   up-sampling iteration                       models/renderer.py:209-245
//   alpha_to_w                                  models/renderer.py:49-63
//   sample_pdf(det=True)                        utils/rend_util.py:276-319
//   cat + sort + gather                         models/renderer.py:255-258
//   d_mid                                       models/renderer.py:266
//   sdf_to_alpha                                models/renderer.py:17-24
//   compositing                                 models/renderer.py:302-333
#pragma once

#include "nm_distance.h"

#define NM_MAX_SAMPLES 256  // N_samples + N_importance upper bound
// The per-ray loops carry only a cheap serial chain (cumprod / running sums); the expensive part of an iteration (two
// sigmoids, correctly rounded divisions) is independent from one interval to the next.  Unrolled by four, the scheduler
// interleaves those chains (one lane per ray, two waves per CU: the kernels are bound by instruction latency).  The
// arithmetic and its order are unchanged.
#define NM_RAY_UNROLL 4
#define NM_STR2(x) #x
#define NM_STR(x) NM_STR2(x)
#if defined(__HIPCC__)
#define NM_UNROLL_RAY _Pragma(NM_STR(unroll NM_RAY_UNROLL))
#else
#define NM_UNROLL_RAY
#endif

NM_HD float nm_exp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return expf(x);
#else
    return std::exp(x);
#endif
}
NM_HD float nm_sigmoid(float x) { return nm_div(1.0f, nm_add(1.0f, nm_exp(-x))); }

NM_HD float nm_fma(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fmaf(a, b, c);
#else
    return std::fma(a, b, c);
#endif
}

// torch.linspace(0, 1, n)[i] in fp32, bit-for-bit: step = 1/(n-1); i < n/2: step*i, else
// fma(-step, n-1-i, 1)  (verified against torch in tests/test_oracle.py).
NM_HD float nm_linspace01(int i, int n) {
    if (n <= 1) return 0.f;
    const float step = nm_div(1.0f, (float)(n - 1));
    return (i < n / 2) ? nm_mul(step, (float)i) : nm_fma(-step, (float)(n - 1 - i), 1.0f);
}

// near*(1-t) + far*t
NM_HD float nm_lerp_depth(float near_, float far_, float t) {
    return nm_add(nm_mul(near_, nm_sub(1.0f, t)), nm_mul(far_, t));
}

// renderer.py:153 + rend_util.py:179-199
NM_HD void nm_ray_setup(const float* o, const float* d_in, float radius, float* dirn, float* near_, float* far_) {
    const float n = nm_sqrt(nm_add(nm_add(nm_mul(d_in[0], d_in[0]), nm_mul(d_in[1], d_in[1])), nm_mul(d_in[2], d_in[2])));
    const float den = fmaxf(n, 1e-12f);
    dirn[0] = nm_div(d_in[0], den);
    dirn[1] = nm_div(d_in[1], den);
    dirn[2] = nm_div(d_in[2], den);
    const float dot = nm_add(nm_add(nm_mul(o[0], dirn[0]), nm_mul(o[1], dirn[1])), nm_mul(o[2], dirn[2]));
    const float mid = -dot;
    *near_ = fmaxf(nm_sub(mid, radius), 0.0f);
    *far_ = fmaxf(nm_add(mid, radius), radius);
}

// renderer.py:93-102 given the smallest / largest probe depth with ds < thresh (1e10 / -1e10: none)
NM_HD void nm_ray_bounds_finish(float mn, float mx, float near0, float far0, float* near_, float* far_) {
    float n = (mn > 1e5f) ? near0 : mn;
    float f = (mx < -1e5f) ? far0 : mx;
    if (nm_sub(f, n) < 0.1f) {
        f = nm_add(f, 0.05f);
        n = nm_sub(n, 0.05f);
    }
    *near_ = n;
    *far_ = f;
}

// renderer.py:88-102: ds_probe[i] is the projected distance at depth lerp(near0, far0, t_i).
NM_HD void nm_ray_bounds(const float* ds_probe, int stride, int G, float thresh, float near0, float far0,
                         float* near_, float* far_) {
    float mn = 1e10f, mx = -1e10f;
    for (int i = 0; i < G; ++i) {
        if (ds_probe[(size_t)i * stride] < thresh) {
            const float d = nm_lerp_depth(near0, far0, nm_linspace01(i, G));
            mn = fminf(mn, d);
            mx = fmaxf(mx, d);
        }
    }
    nm_ray_bounds_finish(mn, mx, near0, far0, near_, far_);
}

// One up-sampling iteration (renderer.py:209-245 + rend_util.py:276-319, det=True):
// reads sorted d[0..n), sdf[0..n); writes n_new new depths to d_new[0..n_new).
// w and cdf are caller-provided scratch of >= n floats; w may alias sdf and cdf may alias w (the
// device kernel keeps a ray's arrays in LDS and reuses the sdf row for both).
// If radius != nullptr: radius[slot] is the distance from an earlier sample (identified by
// slot[j], the position it was generated at) to its K-th nearest vertex; bound_new[i] then receives
// an upper bound of the K-th-neighbour distance of the i-th new sample: radius of the sample just
// below it on the same ray + the depth gap (triangle inequality, the direction is a unit vector).
// u_rand (optional, [n_new]): the stratum positions of sample_pdf(det=False) (rend_util.py:300-302: the
// caller's torch.rand), in any order; nullptr = det=True, u = linspace(0, 1, n_new).
template <class SlotT = int>
NM_HD void nm_ray_upsample(const float* d, const float* sdf, int n, int it, int n_new, float* d_new,
                           float* w, float* cdf, const SlotT* slot = nullptr, const float* radius = nullptr,
                           float* bound_new = nullptr, const float* u_rand = nullptr) {
    const float s = (float)(256 << it);
    float prev_dot = 0.f;
    double T = 1.0;      // cumprod accumulator (float64, rounded to fp32 per element like torch CPU)
    double wsum = 0.0;
    float d_j = d[0], s_j = sdf[0];
    NM_UNROLL_RAY
    for (int j = 0; j + 1 < n; ++j) {
        const float d_n = d[j + 1], s_n = sdf[j + 1];
        const float dist = nm_sub(d_n, d_j);
        const float mid = nm_mul(nm_add(s_j, s_n), 0.5f);
        const float dot = nm_div(nm_sub(s_n, s_j), nm_add(dist, 1e-5f));
        float dv = fminf(prev_dot, dot);
        dv = fminf(fmaxf(dv, -10.0f), 0.0f);
        prev_dot = dot;
        const float h = nm_mul(nm_mul(dv, dist), 0.5f);
        const float pc = nm_sigmoid(nm_mul(nm_sub(mid, h), s));
        const float nc = nm_sigmoid(nm_mul(nm_add(mid, h), s));
        const float alpha = nm_div(nm_add(nm_sub(pc, nc), 1e-5f), nm_add(pc, 1e-5f));
        const float wj = nm_add(nm_mul(alpha, (float)T), 1e-5f);  // alpha_to_w, then sample_pdf's +1e-5
        T *= (double)nm_add(nm_sub(1.0f, alpha), 1e-10f);
        w[j] = wj;  // (sdf[j] is dead from here on)
        wsum += (double)wj;
        d_j = d_n;
        s_j = s_n;
    }
    const float sum = (float)wsum;
    double c = 0.0;
    float w_j = n > 1 ? w[0] : 0.f;
    cdf[0] = 0.f;
    NM_UNROLL_RAY
    for (int j = 0; j + 1 < n; ++j) {
        const float w_next = (j + 2 < n) ? w[j + 1] : 0.f;  // read before cdf[j + 1] may overwrite it
        c += (double)nm_div(w_j, sum);
        cdf[j + 1] = (float)c;
        w_j = w_next;
    }
    int lb = 0;  // searchsorted(cdf, u, right=False); u ascending => monotone lower bound
    for (int i = 0; i < n_new; ++i) {
        const float u = u_rand ? u_rand[i] : nm_linspace01(i, n_new);
        if (u_rand) {  // unordered u: binary lower bound
            int lo = 0, hi = n;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cdf[mid] < u) lo = mid + 1;
                else hi = mid;
            }
            lb = lo;
        }
        while (lb < n && cdf[lb] < u) ++lb;
        const int below = lb - 1 > 0 ? lb - 1 : 0;
        const int above = lb < n - 1 ? lb : n - 1;
        float den = nm_sub(cdf[above], cdf[below]);
        if (den < 1e-5f) den = 1.0f;
        const float t = nm_div(nm_sub(u, cdf[below]), den);
        const float dn = nm_add(d[below], nm_mul(t, nm_sub(d[above], d[below])));
        d_new[i] = dn;
        if (radius) {
            const float rb = radius[(int)slot[below]] + fabsf(dn - d[below]);
            const float ra = radius[(int)slot[above]] + fabsf(dn - d[above]);
            bound_new[i] = fminf(rb, ra);
        }
    }
}

// renderer.py:255-258: sorted d[0..n) + unsorted tail d[n..n+m) -> sorted d[0..n+m), sdf follows.
// Stable (tail elements go after equal prefix elements); equal depths are equal points and the
// field is deterministic per point, so the tie order cannot change any value.
// slot (optional): slot[j] = generation position of the sample now at sorted position j; the
// tail elements enter with their own position as slot id.
template <class SlotT = int>
NM_HD void nm_ray_merge(float* d, float* sdf, int n, int m, SlotT* slot = nullptr) {
    for (int t = n; t < n + m; ++t) {
        const float dv = d[t], sv = sdf[t];
        int p = t;
        while (p > 0 && d[p - 1] > dv) {
            d[p] = d[p - 1];
            sdf[p] = sdf[p - 1];
            if (slot) slot[p] = slot[p - 1];
            --p;
        }
        d[p] = dv;
        sdf[p] = sv;
        if (slot) slot[p] = (SlotT)t;
    }
}

// sdf_to_alpha (renderer.py:17-24) + visibility weights (renderer.py:49-63, :306): w[j], j < N-1.
// w may alias sdf (w[j] is written after sdf[j] and sdf[j+1] have been read).
NM_HD void nm_ray_weights(const float* sdf, int N, float s, float* w) {
    double T = 1.0;  // cumprod accumulator (float64, rounded to fp32 per element like torch CPU)
    float cdf_j = nm_sigmoid(nm_mul(sdf[0], s));
    NM_UNROLL_RAY
    for (int j = 0; j + 1 < N; ++j) {
        const float cdf_n = nm_sigmoid(nm_mul(sdf[j + 1], s));
        const float alpha = fmaxf(nm_div(nm_sub(cdf_j, cdf_n), nm_add(cdf_j, 1e-10f)), 0.0f);
        w[j] = nm_mul(alpha, (float)T);
        T *= (double)nm_add(nm_sub(1.0f, alpha), 1e-10f);
        cdf_j = cdf_n;
    }
}

// renderer.py:278 + :302-333.  sdf[N], d[N] sorted; rgb_mid [N-1,3]; nablas [N,3] or nullptr.
// out: rgb[3], depth, acc, normals[3] (if nablas).
// A mid-point whose weight is exactly 0 (alpha = 0 wherever the SDF does not decrease along the ray)
// contributes w*c = 0 for any finite colour, and x + 0 == x: its colour is never read, so the fused
// renderer does not evaluate the field there at all (nm_render_rays; bit-identical result).
NM_HD void nm_ray_composite(const float* sdf, const float* d, int N, float s, const float* rgb_mid,
                            const float* nablas, int white_bkgd, float* rgb, float* depth, float* acc,
                            float* normals, float* w_scratch, const float* evaluated_w = nullptr) {
    // evaluated_w (optional): the weight array the caller used to decide which mid-points to evaluate
    // (same function, same inputs => same values; reading the decision from it keeps "not evaluated"
    // and "not read" the same set by construction)
    nm_ray_weights(sdf, N, s, w_scratch);
    float r = 0.f, g = 0.f, b = 0.f, wsum = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
    NM_UNROLL_RAY
    for (int j = 0; j + 1 < N; ++j) {
        const float w = w_scratch[j];
        if ((evaluated_w ? evaluated_w[j] : w) != 0.0f) {
            r = nm_add(r, nm_mul(w, rgb_mid[3 * j + 0]));
            g = nm_add(g, nm_mul(w, rgb_mid[3 * j + 1]));
            b = nm_add(b, nm_mul(w, rgb_mid[3 * j + 2]));
        }
        wsum = nm_add(wsum, w);
        // (normals: a zero-weight term adds +-0 to the sums, i.e. nothing; where the caller skipped the evaluation
        //  -- evaluated_w given and 0 -- the nabla was never computed and must not be read)
        if (nablas && (!evaluated_w || evaluated_w[j] != 0.0f)) {
            const float ax = nablas[3 * j], ay = nablas[3 * j + 1], az = nablas[3 * j + 2];
            const float nn = fmaxf(nm_sqrt(nm_add(nm_add(nm_mul(ax, ax), nm_mul(ay, ay)), nm_mul(az, az))), 1e-12f);
            nx = nm_add(nx, nm_mul(nm_div(ax, nn), w));
            ny = nm_add(ny, nm_mul(nm_div(ay, nn), w));
            nz = nm_add(nz, nm_mul(nm_div(az, nn), w));
        }
    }
    const float den = nm_add(wsum, 1e-10f);
    float dep = 0.f;
    NM_UNROLL_RAY
    for (int j = 0; j + 1 < N; ++j) {
        const float dm = nm_mul(0.5f, nm_add(d[j + 1], d[j]));
        dep = nm_add(dep, nm_mul(nm_div(w_scratch[j], den), dm));
    }
    if (white_bkgd) {
        const float bg = nm_sub(1.0f, wsum);
        r = nm_add(r, bg);
        g = nm_add(g, bg);
        b = nm_add(b, bg);
    }
    rgb[0] = r; rgb[1] = g; rgb[2] = b;
    *depth = dep;
    *acc = wsum;
    if (nablas && normals) {
        normals[0] = nx; normals[1] = ny; normals[2] = nz;
    }
}
