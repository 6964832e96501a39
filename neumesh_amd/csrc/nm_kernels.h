// nm_kernels.h -- device kernels other than the MLPs: K-NN / projected-distance kernel,
// per-ray stage kernels, small utility kernels.  Device-only (included by nm_api.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <rocprim/block/block_radix_sort.hpp>   // stable block-level radix sort of the depth-bucket lists (nm_rays_order_sort_kernel)

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
    // warm start (modes 1,2; optional): bound[r*dstride + doff + p] = upper bound of the distance
    // from the point to its K-th nearest vertex (see nm_ray_upsample)
    const float* bound;
    // where the per-point outputs go: record index = q (compact) if out_stride == 0,
    // else r*out_stride + out_off + p (per-ray slots, so later stages can address them by slot)
    int out_stride, out_off;
    // lane -> (ray, sample) assignment by depth buckets (optional, see nm_rays_order_sort_kernel): groups of
    // order_rays adjacent rays, E = roundup(order_rays*P, 64) entries per group,
    // order[group*E + j] = (ray - group*order_rays)*P + p of the j-th sample of the group (0xFFFF = padding)
    const unsigned short* order;
    int order_rays;
    int out_by_slot;  // (with order) per-point outputs go to record index = position in the order list
    // mode 2 only: a wave walks `chain` consecutive 4-sample tiles of its 16 rays (0/1 = one tile) and
    // warm-starts every tile after the first from the tile before it (see nm_distance_kernel)
    int chain;
    // mode 0 only: queries per wave (64, 32, 16 or 8; 0 = 64).  Small point-wise calls (a training step's few 10^4 points)
    // are bound by ONE wave's serial traversal, which for scattered queries grows with the number of lanes that walk their
    // own path: fewer queries per wave = more, shorter waves on a chip that has room for them.
    int lanes;
    // modes 1, 2 (optional): ray r of this launch is ray ray_index[r] of the rays_o / dirn / nearfar arrays (a compacted list of
    // rays, nm_surface_hits); depths / bounds / outputs stay indexed by the launch's own r
    const int* ray_index;
    // mode 2 (optional): the P samples are proposals p_off .. p_off + P - 1 of a p_total-point linspace (0 = the P points themselves)
    int p_off, p_total;
    // small launches (nm_distance_kernel<false, true>): a wave whose traversal has spent `budget` work units (24 per node test, 7 per
    // staged vertex) gives up and appends its queries to defer_list (defer_count entries so far, room for defer_cap); they are
    // finished by a wave of their own each (nm_knn_split_kernel, nm_distance_deferred_kernel below).  budget = 0: never.
    int budget, defer_cap;
    int* defer_count;
    int* defer_list;
    float* defer_bound2;
};

#ifdef NM_TESTING
// test library only: per-wave life of the distance kernels (tools/knn_wave_times.py): log[0] = number of entries, then (start, end,
// wave index) per wave, written by lane 0 at the end of the kernel
__device__ long long* g_nm_wave_log = nullptr;
__device__ __forceinline__ void nm_wave_log_write(long long t0, long long wave) {
    long long* log = g_nm_wave_log;
    if (!log || (threadIdx.x & 63)) return;
    const long long i = (long long)atomicAdd(reinterpret_cast<unsigned long long*>(log), 1ull);
    if (i < (1 << 20)) {
        log[1 + 3 * i] = t0;
        log[2 + 3 * i] = (long long)__builtin_amdgcn_s_memrealtime();
        log[3 + 3 * i] = wave;
    }
}
#endif

// (r, p) = (ray, sample) of query q = r*P + p, as produced by nm_lane_query (mode 0: r = q, p = 0)
__device__ __forceinline__ long long nm_out_index(const NmPointSrc& s, long long q, long long r, int p) {
    if (s.mode == 0 || s.out_stride == 0) return q;
    return r * s.out_stride + s.out_off + p;
}

// squared warm-start bound of point q (+INF when there is none); inflated so that it stays an
// upper bound under fp32 rounding of the positions and of the candidate distances
__device__ __forceinline__ float nm_init_bound(const NmPointSrc& s, long long r, int p) {
    if (s.mode == 0 || !s.bound) return NM_INF_F;
    const float b = s.bound[r * s.dstride + s.doff + p] * 1.0001f + 1e-5f;
    return b * b;
}

__device__ __forceinline__ void nm_fetch_point(const NmPointSrc& s, long long r, int p, float& x, float& y, float& z, float& d) {
    if (s.mode == 0) {
        x = s.xyz[r * 3];
        y = s.xyz[r * 3 + 1];
        z = s.xyz[r * 3 + 2];
        d = 0.f;
        return;
    }
    const long long rr = s.ray_index ? (long long)s.ray_index[r] : r;
    if (s.mode == 1) {
        d = s.depth[r * s.dstride + s.doff + p];
    } else {
        d = nm_lerp_depth(s.nearfar[2 * rr], s.nearfar[2 * rr + 1], s.p_total > 0 ? nm_linspace01(s.p_off + p, s.p_total) : nm_linspace01(p, s.P));
        if (s.depth_out) s.depth_out[r * s.dstride + s.doff + p] = d;
    }
    x = nm_add(s.rays_o[3 * rr], nm_mul(d, s.dirn[3 * rr]));
    y = nm_add(s.rays_o[3 * rr + 1], nm_mul(d, s.dirn[3 * rr + 1]));
    z = nm_add(s.rays_o[3 * rr + 2], nm_mul(d, s.dirn[3 * rr + 2]));
}

#include "nm_knn.h"   // K-NN traversal and the kernels built on it

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

// Per-ray rows in LDS.  The up-sampling stages are serial per ray (ordered float64 scans, a data-
// dependent insertion sort), one lane owns one ray -- run directly on the [R][cap] global arrays
// every step of those loops is a dependent, uncoalesced global access (measured: 1.5-2.8 ms per
// launch of 65536 rays).  So a 64-ray workgroup first copies its rows into LDS with coalesced
// loads (row stride cap+1 words: lane-private rows fall into distinct banks), runs the unchanged
// serial code there, and copies the results back.  Slots are bytes in LDS (cap <= 256).
struct NmRayLds {
    float* d;             // [64][cap + 1]
    float* s;             // [64][cap + 1]  sdf, then weights / cdf (nm_ray_upsample aliases them)
    unsigned char* slot;  // [64][cap + 4]
    int S, SB;
};
__device__ __forceinline__ NmRayLds nm_ray_lds(float* base, int cap) {
    NmRayLds l;
    l.S = cap + 1;
    l.SB = cap + 4;
    l.d = base;
    l.s = base + 64 * l.S;
    l.slot = reinterpret_cast<unsigned char*>(base + 2 * 64 * l.S);
    return l;
}
static inline size_t nm_ray_lds_bytes(int cap) { return (size_t)2 * 64 * (cap + 1) * 4 + (size_t)64 * (cap + 4); }

#ifndef NM_RAY_IO_THREADS
#define NM_RAY_IO_THREADS 256   // threads per 64-ray workgroup of the upsample / finalize kernels (64 = the one-wave form of rounds 1-3)
#endif
// rows [0, n) of 64 rays: global -> LDS (slot == nullptr or first == true: identity slots).
// The (ray, sample) elements are walked as one flat range, eight per lane in flight: written as a row loop, every
// iteration waited for its own two loads (s_waitcnt vmcnt(0) before the LDS store) -- 128 dependent memory round trips
// per workgroup at one wave per SIMD, ~30 % of these kernels' time.
__device__ __forceinline__ void nm_ray_rows_load(const NmRayLds& l, const float* __restrict__ d, const float* __restrict__ sdf,
                                                 const int* __restrict__ slot, bool identity, long long r0, long long R,
                                                 int cap, int n) {
    const int lane = threadIdx.x, T = blockDim.x;   // (every thread of the workgroup moves data; threads 0..63 own the rays)
    const int rows = (int)((R - r0) < 64 ? (R - r0) : 64);
    const int total = rows * n;
    constexpr int U = 8;
    for (int e0 = 0; e0 < total; e0 += T * U) {
        float dv[U], sv[U];
        int sl[U], rr[U], jj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * T + lane;
            const bool ok = e < total;
            rr[u] = ok ? e / n : 0;
            jj[u] = ok ? e - rr[u] * n : -1;
            const long long g = (r0 + rr[u]) * cap + (ok ? jj[u] : 0);
            dv[u] = ok ? d[g] : 0.f;
            sv[u] = ok ? sdf[g] : 0.f;
            sl[u] = (ok && slot && !identity) ? slot[g] : jj[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (jj[u] < 0) continue;
            l.d[rr[u] * l.S + jj[u]] = dv[u];
            l.s[rr[u] * l.S + jj[u]] = sv[u];
            if (slot) l.slot[rr[u] * l.SB + jj[u]] = (unsigned char)sl[u];
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void nm_ray_rows_store(const NmRayLds& l, float* __restrict__ d, float* __restrict__ sdf,
                                                  int* __restrict__ slot, long long r0, long long R, int cap, int j0, int j1,
                                                  bool with_sdf) {
    const int lane = threadIdx.x, T = blockDim.x;
    __syncthreads();
    const int rows = (int)((R - r0) < 64 ? (R - r0) : 64), w = j1 - j0;
    for (int e = lane; e < rows * w; e += T) {   // flat (ray, sample) range: a 16-sample tail still fills every lane of the workgroup
        const int rr = e / w, j = j0 + (e - rr * w);
        const long long g = (r0 + rr) * cap;
        d[g + j] = l.d[rr * l.S + j];
        if (with_sdf) {
            sdf[g + j] = l.s[rr * l.S + j];
            if (slot) slot[g + j] = (int)l.slot[rr * l.SB + j];
        }
    }
}

// In-place merge of the sorted prefix d[0..n0) with the m <= MAXM samples appended behind it, for the
// usual case that the appended samples are themselves ascending (inverse-CDF samples of ascending u
// are): the tail is held in registers and the two runs are merged from the back, so every element
// moves once (<= n0 + m LDS moves, against ~m*n0/2 for the insertion sort of nm_ray_merge).  Same
// result as nm_ray_merge -- stable, tail after equal prefix elements, slot of a tail element = its
// position.  Returns false (nothing touched) if the tail is longer than MAXM or not ascending
// (perturb=True): the caller falls back to nm_ray_merge.
template <int MAXM>
__device__ __forceinline__ bool nm_ray_merge_sorted_tail(float* d, float* sdf, int n0, int m, unsigned char* slot) {
    if (m > MAXM) return false;
    float td[MAXM], ts[MAXM];
#pragma unroll
    for (int i = 0; i < MAXM; ++i) {
        td[i] = i < m ? d[n0 + i] : 0.f;
        ts[i] = i < m ? sdf[n0 + i] : 0.f;
    }
    bool ascending = true;
#pragma unroll
    for (int i = 1; i < MAXM; ++i) ascending = ascending && !(i < m && td[i] < td[i - 1]);
    if (!ascending) return false;
    int p = n0 - 1, t = m - 1;
    float dp = p >= 0 ? d[p] : 0.f;
    for (int k = n0 + m - 1; t >= 0; --k) {
        float tv = td[0], tsv = ts[0];
#pragma unroll
        for (int i = 1; i < MAXM; ++i) {  // register file has no dynamic index: select
            tv = (t == i) ? td[i] : tv;
            tsv = (t == i) ? ts[i] : tsv;
        }
        if (p >= 0 && dp > tv) {
            d[k] = dp;
            sdf[k] = sdf[p];
            if (slot) slot[k] = slot[p];
            --p;
            dp = p >= 0 ? d[p] : 0.f;
        } else {
            d[k] = tv;
            sdf[k] = tsv;
            if (slot) slot[k] = (unsigned char)(n0 + t);
            --t;
        }
    }
    return true;
}

// merge the m samples appended by the previous iteration, then draw n_new new ones (+ their
// warm-start bounds from the cached K-th-neighbour radius of the neighbouring samples).
// Launch: 64 rays per block with 64 ... 256 threads (thread t < 64 owns ray t through the serial stages; ALL threads move the rows
// between HBM and LDS: these kernels are bound by the few loads two 1-wave workgroups per CU keep in flight -- 3.8 GB per launch at
// 1.75 TB/s in round 3), nm_ray_lds_bytes(cap) dynamic LDS.
__global__ __launch_bounds__(256) void nm_rays_upsample_kernel(float* __restrict__ d, float* __restrict__ sdf, int* __restrict__ slot,
                                                              const float* __restrict__ radius, float* __restrict__ bound, long long R,
                                                              int cap, int n, int m, int it, int n_new,
                                                              const float* __restrict__ u_rand, const int* __restrict__ u_perm = nullptr) {
    extern __shared__ float nm_ray_smem[];
    const NmRayLds l = nm_ray_lds(nm_ray_smem, cap);
    const long long r0 = (long long)blockIdx.x * 64;
    const long long r = r0 + threadIdx.x;
    nm_ray_rows_load(l, d, sdf, slot, m == 0, r0, R, cap, n);
    const bool owner = threadIdx.x < 64 && r < R;      // this thread runs a ray's serial stages
    const int row = threadIdx.x & 63;
    float* dr = l.d + row * l.S;
    float* sr = l.s + row * l.S;
    unsigned char* sl = slot ? l.slot + row * l.SB : nullptr;
    __syncthreads();                                    // (rows loaded by other waves)
    if (owner && m > 0 && !nm_ray_merge_sorted_tail<16>(dr, sr, n - m, m, sl)) nm_ray_merge(dr, sr, n - m, m, sl);
    if (m > 0 || slot) nm_ray_rows_store(l, d, sdf, slot, r0, R, cap, 0, n, true);  // merged rows (+ identity slots)
    __syncthreads();
    if (owner)
        nm_ray_upsample(dr, sr, n, it, n_new, dr + n, sr, sr, sl, (sl && radius) ? radius + r * cap : nullptr,
                        bound ? bound + r * cap + n : nullptr,
                        u_rand ? u_rand + (u_perm ? (long long)u_perm[r] : r) * n_new : nullptr);   // (u_perm: the caller's index of sorted ray r)
    nm_ray_rows_store(l, d, sdf, nullptr, r0, R, cap, n, n + n_new, false);  // the new depths
}

// final merge + mid-point depths (renderer.py:255-258, :266) + warm-start bounds of the mid-points
__global__ __launch_bounds__(256) void nm_rays_finalize_kernel(float* __restrict__ d, float* __restrict__ sdf, int* __restrict__ slot,
                                                              const float* __restrict__ radius, long long R, int cap, int n, int m,
                                                              float* __restrict__ d_mid, float* __restrict__ bound_mid,
                                                              float s_val, float* __restrict__ w_mid, float w_eps) {
    extern __shared__ float nm_ray_smem[];
    const NmRayLds l = nm_ray_lds(nm_ray_smem, cap);
    const long long r0 = (long long)blockIdx.x * 64;
    const long long r = r0 + threadIdx.x;
    nm_ray_rows_load(l, d, sdf, slot, m == 0, r0, R, cap, n);
    const bool owner = threadIdx.x < 64 && r < R;
    const int row = threadIdx.x & 63;
    unsigned char* sl = slot ? l.slot + row * l.SB : nullptr;
    __syncthreads();
    if (owner && m > 0 && !nm_ray_merge_sorted_tail<16>(l.d + row * l.S, l.s + row * l.S, n - m, m, sl))
        nm_ray_merge(l.d + row * l.S, l.s + row * l.S, n - m, m, sl);
    if (m > 0 || slot) nm_ray_rows_store(l, d, sdf, slot, r0, R, cap, 0, n, true);
    __syncthreads();
    const int lane = threadIdx.x, T = blockDim.x;
    // visibility weights of the mid-points (in place of the sdf row), for the zero-weight skip of the
    // mid-point pass: the SAME function the compositing kernel evaluates later
    if (w_mid) {
        if (owner) nm_ray_weights(l.s + row * l.S, n, s_val, l.s + row * l.S);
        __syncthreads();
        for (int rr = 0; rr < 64 && r0 + rr < R; ++rr)
            for (int j = lane; j + 1 < n; j += T) {  // (w_eps = 0: the weights themselves; else weights below it count as 0)
                const float wv = l.s[rr * l.S + j];
                w_mid[(r0 + rr) * cap + j] = wv < w_eps ? 0.0f : wv;
            }
        __syncthreads();
    }
    // the sdf rows are no longer needed in LDS: reuse them for the radius rows (coalesced loads)
    const bool warm = slot && radius && bound_mid;
    if (warm) {
        for (int rr = 0; rr < 64 && r0 + rr < R; ++rr)
            for (int j = lane; j < n; j += T) l.s[rr * l.S + j] = radius[(r0 + rr) * cap + j];
        __syncthreads();
    }
    for (int rr = 0; rr < 64 && r0 + rr < R; ++rr) {
        const float* dr = l.d + rr * l.S;
        const float* rad = l.s + rr * l.S;
        const unsigned char* sr = l.slot + rr * l.SB;
        const long long g = (r0 + rr) * cap;
        for (int j = lane; j + 1 < n; j += T) {
            const float dm = nm_mul(0.5f, nm_add(dr[j + 1], dr[j]));
            d_mid[g + j] = dm;
            if (warm) bound_mid[g + j] = fminf(rad[sr[j]] + fabsf(dm - dr[j]), rad[sr[j + 1]] + fabsf(dr[j + 1] - dm));
        }
    }
}

// Depth-bucket assignment of the importance samples to waves.  The P new samples of a ray follow
// the ray's own density profile, so a (16 rays x 4 samples) tile of them can stretch over the whole
// depth range and its cooperative K-NN traversal has to cover the union of 64 far-apart searches
// (measured on the benchmark scene: 1778 node tests + 3609 vertex visits per wave).  Ordering the
// 64*P samples of 64 adjacent rays by depth and cutting the list into waves gives compact
// footprints again (840 + 1289).  The same holds, less dramatically, for the N-1 mid-points of the
// final sorted samples (16 rays x 4 consecutive ones: 828 + 1452; the 2032 mid-points of 16 rays
// ordered by depth: 559 + 856).  One workgroup per group of G rays; which lane evaluates which sample
// changes no value.
// A wave only needs its 64 samples to be NEAR each other in depth, so the list is ordered by BUCKET: 1024 depth buckets between the
// group's smallest and largest kept depth (finer than the vertex spacing on the benchmark scene), ids ascending inside a bucket.
// wgt (optional, [R][cap]): samples with wgt == 0 are dropped (padding behind the kept ones); counter (optional): += kept.
#define NM_ORDER_BUCKETS 1024
// The list -- buckets ascending, ids ascending inside every bucket -- by a STABLE block radix sort of the 11-bit bucket keys with the entries'
// ids as values (thread t holds ids [t * IPT, (t + 1) * IPT): id order, which a stable sort keeps inside a bucket).  Deterministic by
// construction (the same list on every run: ADVICE r4), and faster than what it replaces: rounds 1-3 sorted (depth, id) keys bitonically (6.5 ms
// per frame), round 4 built the buckets by histogram + atomic scatter and re-ordered small buckets only (2.1 ms, but the importance passes
// CROWD their buckets -- the new samples of a patch of adjacent rays cluster at the surface's depth, hundreds per bucket -- and those stayed in
// atomic order, different from run to run; ranking them as well cost 5.0 ms).  This kernel: 1.4 ms per frame (rocprofv3, round 5).
// IPT = entries per thread: E <= 256 * IPT (groups hold at most 8192 entries: nm_fine_group_rays / nm_mid_group_rays).
template <int IPT>
__global__ __launch_bounds__(256) void nm_rays_order_sort_kernel(const float* __restrict__ d, long long R, int cap, int off,
                                                                 int P, int G, unsigned short* __restrict__ order,
                                                                 const float* __restrict__ wgt, unsigned long long* __restrict__ counter) {
    using Sort = rocprim::block_radix_sort<unsigned short, 256, IPT, unsigned short>;
    __shared__ typename Sort::storage_type nm_sort_st;
    __shared__ unsigned nm_lo, nm_hi, nm_kept;
    const long long grp = blockIdx.x;
    const int n = G * P, E = (n + 63) & ~63;
    const int t = threadIdx.x;
    if (t == 0) { nm_lo = 0xffffffffu; nm_hi = 0u; nm_kept = 0u; }
    __syncthreads();
    float dep[IPT];
    unsigned lo = 0xffffffffu, hi = 0u, kept = 0u;
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int i = t * IPT + k;
        const int rl = i / P;
        const long long r = grp * G + rl;
        float v = __int_as_float(0x7fc00000);                                // NaN bit pattern = dropped
        if (i < n && r < R) {
            const long long g = r * cap + off + (i - rl * P);
            if (!(wgt && wgt[g] == 0.0f)) {
                v = d[g];
                ++kept;
                if (v != v) v = 3.0e38f;                                     // (a NaN depth still gets a place: the last bucket, outside the range)
                else {
                    const unsigned key = nm_float_key(v);
                    lo = key < lo ? key : lo;
                    hi = key > hi ? key : hi;
                }
            }
        }
        dep[k] = v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned l2 = (unsigned)__shfl_xor((int)lo, o), h2 = (unsigned)__shfl_xor((int)hi, o);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
        kept += (unsigned)__shfl_xor((int)kept, o);
    }
    if ((t & 63) == 0) { atomicMin(&nm_lo, lo); atomicMax(&nm_hi, hi); atomicAdd(&nm_kept, kept); }
    __syncthreads();
    const unsigned klo = nm_lo, khi = nm_hi;
    const unsigned ulo = klo ^ ((klo >> 31) ? 0x80000000u : 0xffffffffu), uhi = khi ^ ((khi >> 31) ? 0x80000000u : 0xffffffffu);
    const float dlo = __uint_as_float(ulo), dhi = klo <= khi ? __uint_as_float(uhi) : dlo;
    const float scale = (float)NM_ORDER_BUCKETS / fmaxf(dhi - dlo, 1e-30f);
    unsigned short keys[IPT], ids[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const float v = dep[k];
        const float x = (v - dlo) * scale;
        const int b = x >= (float)(NM_ORDER_BUCKETS - 1) ? NM_ORDER_BUCKETS - 1 : (x > 0.f ? (int)x : 0);   
        keys[k] = (v == v) ? (unsigned short)b : (unsigned short)NM_ORDER_BUCKETS;                           // dropped entries behind every bucket
        ids[k] = (unsigned short)(t * IPT + k);
    }
    Sort().sort(keys, ids, nm_sort_st, 0, 11);
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int pos = t * IPT + k;
        if (pos < E) order[grp * E + pos] = keys[k] < NM_ORDER_BUCKETS ? ids[k] : (unsigned short)0xffffu;
    }
    if (counter && t == 0 && nm_kept) atomicAdd(counter, (unsigned long long)nm_kept);
}

// sample points of a ray batch as an explicit [R,P,3] array (staged renderer: the field is queried
// through the model's Python methods between the per-ray stages)
__global__ void nm_rays_points_kernel(NmPointSrc src, long long Q, float* __restrict__ xyz) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    float x, y, z;
    const long long r = q / src.P;
    float dep;
    nm_fetch_point(src, r, (int)(q - r * src.P), x, y, z, dep);
    xyz[3 * q] = x;
    xyz[3 * q + 1] = y;
    xyz[3 * q + 2] = z;
}

__global__ __launch_bounds__(256) void nm_rays_composite_kernel(const float* __restrict__ sdf, const float* __restrict__ d, long long R,
                                         int cap, int N, float s, const float* __restrict__ rgb_mid,
                                         const float* __restrict__ nablas, int white_bkgd, float* __restrict__ rgb,
                                         float* __restrict__ depth, float* __restrict__ acc,
                                         float* __restrict__ normals, const float* __restrict__ evaluated_w,
                                         const int* __restrict__ perm) {
    // 64 rays per workgroup; the sdf / depth rows come into LDS through every thread of the workgroup (as in the up-sampling kernels;
    // rounds 1-3: one lane per ray reading its rows from global memory with a 256-float weight array in scratch), the weights replace
    // the sdf row in place, thread t < 64 runs ray t's serial sums.  Launch: 64 ... 256 threads, nm_ray_lds_bytes(cap) dynamic LDS.
    extern __shared__ float nm_ray_smem[];
    const NmRayLds l = nm_ray_lds(nm_ray_smem, cap);
    const long long r0 = (long long)blockIdx.x * 64;
    const long long r = r0 + threadIdx.x;
    nm_ray_rows_load(l, d, sdf, nullptr, true, r0, R, cap, N);
    __syncthreads();
    if (threadIdx.x >= 64 || r >= R) return;
    const long long ro = perm ? perm[r] : r;  // rays were processed in spatial order: results go back to the caller's order
    float* wrow = l.s + threadIdx.x * l.S;
    nm_ray_composite(wrow, l.d + threadIdx.x * l.S, N, s, rgb_mid + r * (long long)(N - 1) * 3,
                     nablas ? nablas + r * (long long)N * 3 : nullptr, white_bkgd, rgb + 3 * ro, depth + ro, acc + ro,
                     normals ? normals + 3 * ro : nullptr, wrow, evaluated_w ? evaluated_w + r * cap : nullptr);
}

// ---------------------------------------------------------- spatial processing order of the rays
// Every K-NN pass hands 16 or 64 CONSECUTIVE rays to a wave / a depth-bucket group, so their footprint
// is only compact if consecutive rays are neighbours in space in both image directions.  A caller's
// rays are row-major pixels (a 64-ray group = a 64x1 pixel strip); sorted by the Morton code of each
// ray's point of closest approach to the scene centre the same group is an ~8x8 pixel patch, whose
// cooperative traversals open ~40 % fewer nodes / vertices (host emulation, importance samples:
// 840 + 1289 -> 498 + 929 per wave).  Rays are independent, results are scattered back (perm).
__global__ void nm_ray_keys_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, long long R, float inv_extent,
                                   unsigned* __restrict__ keys, int* __restrict__ idx) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float ox = rays_o[3 * r], oy = rays_o[3 * r + 1], oz = rays_o[3 * r + 2];
    const float dx = rays_d[3 * r], dy = rays_d[3 * r + 1], dz = rays_d[3 * r + 2];
    const float dd = fmaxf(dx * dx + dy * dy + dz * dz, 1e-24f);
    const float t = -(ox * dx + oy * dy + oz * dz) / dd;
    const float c[3] = {ox + t * dx, oy + t * dy, oz + t * dz};
    unsigned code = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float u = c[a] * inv_extent * 0.5f + 0.5f;  // [-extent, extent] -> [0, 1]
        u = !(u > 0.f) ? 0.f : (u > 1.f ? 1.f : u);  // (NaN -> 0)
        unsigned q = (unsigned)(u * 1023.0f);
        q = (q | (q << 16)) & 0x030000ffu;
        q = (q | (q << 8)) & 0x0300f00fu;
        q = (q | (q << 4)) & 0x030c30c3u;
        q = (q | (q << 2)) & 0x09249249u;
        code |= q << a;
    }
    keys[r] = code;
    idx[r] = (int)r;
}
__global__ void nm_ray_gather_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const int* __restrict__ perm,
                                     long long R, float* __restrict__ o_s, float* __restrict__ d_s) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const long long s = perm[r];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        o_s[3 * r + a] = rays_o[3 * s + a];
        d_s[3 * r + a] = rays_d[3 * s + a];
    }
}

// rend_util.get_rays for a contiguous pixel range (utils/rend_util.py:95-118,123-176)
struct NmCamera {
    float r[12];
    float fx, fy, cx, cy, sk;
    int H, W;
};
__device__ __forceinline__ void nm_make_ray(const NmCamera& cam, long long p, long long i, float* __restrict__ rays_o, float* __restrict__ rays_d) {
    const float y = (float)(p / cam.W), x = (float)(p - (p / cam.W) * cam.W);
    // x_lift = (x - cx + cy*sk/fy - sk*y/fy) / fx * z,  y_lift = (y - cy) / fy * z,  z = 1
    const float xl = nm_div(nm_sub(nm_add(nm_sub(x, cam.cx), nm_div(nm_mul(cam.cy, cam.sk), cam.fy)), nm_div(nm_mul(cam.sk, y), cam.fy)), cam.fx);
    const float yl = nm_div(nm_sub(y, cam.cy), cam.fy);
    const float n = nm_sqrt(nm_add(nm_add(nm_mul(xl, xl), nm_mul(yl, yl)), 1.0f));
    const float dx = nm_div(xl, n), dy = nm_div(yl, n), dz = nm_div(1.0f, n);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        rays_d[3 * i + a] = nm_add(nm_add(nm_mul(cam.r[4 * a], dx), nm_mul(cam.r[4 * a + 1], dy)), nm_mul(cam.r[4 * a + 2], dz));
        rays_o[3 * i + a] = cam.r[4 * a + 3];
    }
}
__global__ void nm_make_rays_kernel(NmCamera cam, long long first, long long count, float* __restrict__ rays_o,
                                    float* __restrict__ rays_d) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    nm_make_ray(cam, first + i, i, rays_o, rays_d);
}
// pixel list instead of a pixel range (a rank's interleaved tiles of a sharded frame, a training step's random pixels)
__global__ void nm_make_rays_indexed_kernel(NmCamera cam, const long long* __restrict__ pixels, long long count, float* __restrict__ rays_o,
                                            float* __restrict__ rays_d) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    long long p = pixels[i];
    const long long np = (long long)cam.H * cam.W;
    p = p < 0 ? 0 : (p >= np ? np - 1 : p);   // (the host checks nothing on the device: out-of-frame entries are clamped)
    nm_make_ray(cam, p, i, rays_o, rays_d);
}

// ------------------------------------------------------------------------------- image assembly (render.py:183-184, 219-249)
__device__ __forceinline__ unsigned char nm_integerify(float v) {  // (uint8)(v * 255.0f), clamped where numpy's cast is undefined
    const float x = __fmul_rn(v, 255.0f);
    return (unsigned char)(x >= 256.0f ? 255 : (x > 0.0f ? (int)x : 0));   // (NaN -> 0)
}
__global__ void nm_depth_max_kernel(const float* __restrict__ depth, long long n, unsigned* __restrict__ max_bits) {
    float m = 0.f;  // depths are >= 0: their bit patterns order like unsigned integers
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) m = fmaxf(m, depth[i]);
    m = nm_wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(max_bits, __float_as_uint(m));
}
__global__ void nm_assemble_kernel(const float* __restrict__ rgb, const float* __restrict__ depth, const float* __restrict__ normals,
                                   long long n, int bgr, unsigned char* __restrict__ rgb8, unsigned char* __restrict__ depth8,
                                   unsigned char* __restrict__ normal8, const float* __restrict__ depth_max) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (rgb8) {
        const unsigned char c0 = nm_integerify(rgb[3 * p]), c1 = nm_integerify(rgb[3 * p + 1]), c2 = nm_integerify(rgb[3 * p + 2]);
        rgb8[3 * p] = bgr ? c2 : c0;
        rgb8[3 * p + 1] = c1;
        rgb8[3 * p + 2] = bgr ? c0 : c2;
    }
    if (depth8) depth8[p] = nm_integerify(__fdiv_rn(depth[p], depth_max[0]));
    if (normal8) {
#pragma unroll
        for (int c = 0; c < 3; ++c) normal8[3 * p + c] = nm_integerify(__fadd_rn(__fmul_rn(normals[3 * p + c], 0.5f), 0.5f));
    }
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

// dst[r][j][0..3) = src[r][slot[r][j]][0..3)  (per-slot rows -> sorted sample order)
__global__ void nm_permute_rows3_kernel(const float* __restrict__ src, const int* __restrict__ slot, long long R, int cap,
                                        int n, float* __restrict__ dst) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= R * n) return;
    const long long r = e / n;
    const int j = (int)(e - r * n);
    const long long s = r * cap + slot[r * cap + j];
    dst[e * 3] = src[s * 3];
    dst[e * 3 + 1] = src[s * 3 + 1];
    dst[e * 3 + 2] = src[s * 3 + 2];
}

// dst[(perm ? perm[r] : r)][0..n) = src[r * src_stride + 0..n): per-ray rows of the workspace (processing order
// of the rays) out to a caller's [R][n] array (caller's ray order)
__global__ void nm_rows_out_kernel(const float* __restrict__ src, long long R, int n, int src_stride,
                                   const int* __restrict__ perm, float* __restrict__ dst) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= R * n) return;
    const long long r = e / n;
    const int j = (int)(e - r * n);
    const long long ro = perm ? (long long)perm[r] : r;
    dst[ro * n + j] = src[r * src_stride + j];
}
