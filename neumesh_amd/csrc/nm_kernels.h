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
    // warm start (modes 1,2; optional): bound[r*dstride + doff + p] = upper bound of the distance
    // from the point to its K-th nearest vertex (see nm_ray_upsample)
    const float* bound;
    // where the per-point outputs go: record index = q (compact) if out_stride == 0,
    // else r*out_stride + out_off + p (per-ray slots, so later stages can address them by slot)
    int out_stride, out_off;
    // lane -> (ray, sample) assignment by depth buckets (optional, see nm_rays_order_kernel): groups of
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

#define NM_KNN_BLOCK 256   // threads per workgroup of every kernel that runs the K-NN traversal (the leaf stage in LDS is sized by it)
// Waves per SIMD the K-NN kernels are compiled for (register budget 512 / waves).  Measured per kernel on the bench frame, round 3
// (tools/knn_variants.sh; K-NN ms per frame, one gpurun call): fine / mid-point passes 4: 90.6, 5: 88.7, 6: 95.3 (0 / 36 / 80 spilled
// registers -- the spills sit in the epilogue, once per point, and cost HBM write traffic rather than issue slots: 17.5 / 45 GB per frame
// at 4 / 6); chained coarse pass 3: +5.5 ms, 4 = 5 = 6; probe walk 4: 92.6, 5: 88.5, 6: 86.1.
#ifndef NM_KNN_WAVES
#define NM_KNN_WAVES 5
#endif
#ifndef NM_KNN_WAVES_CHAIN
#define NM_KNN_WAVES_CHAIN 4
#endif
#ifndef NM_KNN_WAVES_PROBE
#define NM_KNN_WAVES_PROBE 6
#endif

// ------------------------------------------------------------ wave-cooperative K-NN search
// The 64 queries of a wave are neighbours in space (consecutive samples of adjacent rays), so
// their K-NN searches open almost the same octree nodes.  The wave therefore runs ONE traversal:
// control flow and the node / vertex addresses are wave-uniform (scalar loads, no divergence),
// every lane evaluates its own box lower bound and its own candidate distances, a node is opened
// when ANY lane still needs it.  Measured on the 800x800 benchmark scene (tests/hostcheck
// emulation): 255 node tests + 468 vertex visits per 64 queries, versus 213 + 342 PER QUERY for
// lane-private traversals that additionally serialise on divergence.  Exactness is unchanged: a
// lane skips a subtree only on its own bound, scanning extra vertices cannot change a K-NN set.
#define NM_UNIFORM_I(x) __builtin_amdgcn_readfirstlane((int)(x))
// The index is read-only for the lifetime of a kernel: loading it through the CONSTANT address
// space lets the compiler use the scalar memory path (s_load_dwordx4 -> SGPRs) whenever the
// address is wave-uniform, which is always the case in the cooperative traversal.
#define NM_CONSTANT __attribute__((address_space(4)))
typedef unsigned nm_u32x4 __attribute__((ext_vector_type(4)));
typedef float nm_f32x4 __attribute__((ext_vector_type(4)));
typedef float nm_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ NmNode nm_ld_node(const NmNode* base, uint32_t i) {
    const nm_u32x4 NM_CONSTANT* pu = (const nm_u32x4 NM_CONSTANT*)(base + i);
    const nm_u32x4 h = pu[0], a = pu[1], b = pu[2], c = pu[3];
    NmNode n;
    n.first = h.x; n.end = h.y; n.parent = h.z; n.info = h.w;
    n.lox = __uint_as_float(a.x); n.loy = __uint_as_float(a.y); n.loz = __uint_as_float(a.z); n.ckx = a.w;
    n.hix = __uint_as_float(b.x); n.hiy = __uint_as_float(b.y); n.hiz = __uint_as_float(b.z); n.cky = b.w;
    n.ckz = c.x; n.om_lo = c.y; n.om_hi = c.z; n.pad = 0;
    return n;
}
__device__ __forceinline__ float4 nm_ld_vert(const float4* base, uint32_t i) {
    const nm_f32x4 v = ((const nm_f32x4 NM_CONSTANT*)(base))[i];
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float nm_uniform_f(float x) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
}

__device__ __forceinline__ float nm_wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float nm_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// BUDGET: returns true (nothing in kk is final then) as soon as the traversal has spent `budget` work units -- see NmPointSrc.budget.
// SUB: the traversal covers the subtree of the INTERNAL node `top` only (never climbs above it).
template <int K, bool BUDGET = false, bool SUB = false, int BLK = NM_KNN_BLOCK>
__device__ __forceinline__ bool nm_knn_search_packet(const NmGridView& g, float qx, float qy, float qz, bool active,
                                                     float rx, float ry, float rz, unsigned long long (&kk)[K], float init_d2, int budget = 0,
                                                     uint32_t top = 0u) {
    const unsigned long long act_mask = __builtin_amdgcn_ballot_w64(active);
    // ONE query in the wave (small point-wise launches, NmPointSrc.lanes = 1): the leaf scans turn from "this lane's query against
    // every staged vertex, one at a time" into "the query against THIS lane's vertex" -- 64 candidate distances per step, the few
    // that beat the current K-th best are inserted one by one into a list every lane keeps a copy of.  The packet centre IS the query
    // then (0.5 (q + q) = q exactly), so every lane computes with the same position and the list stays wave-uniform.  Such launches
    // live as long as their slowest wave, and the slowest ones hold a query near the medial axis of the object, for which almost
    // every leaf has to be scanned (tools/knn_wave_times.py: median wave 0.10 ms, slowest 1.2 ms = the whole launch).  Measured:
    // 8 k-point launches of a training step 0.26-0.33 -> 0.18-0.20 ms.  (The same idea for 2-16 queries per wave -- one broadcast
    // query at a time against 64 vertices -- gained 13 % at 4 queries and lost 2.5 x at 16: not kept.)
    const bool single = __popcll(act_mask) == 1;
    if (single) {
        const int src = __builtin_ctzll(act_mask);
        qx = rx; qy = ry; qz = rz;
        init_d2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(init_d2), src));
    }
#pragma unroll
    for (int k = 0; k < K; ++k) kk[k] = nm_key(init_d2, 0x7fffffff);
    const uint32_t kx = (uint32_t)NM_UNIFORM_I(nm_float_key(rx)), ky = (uint32_t)NM_UNIFORM_I(nm_float_key(ry)),
                   kz = (uint32_t)NM_UNIFORM_I(nm_float_key(rz));  // wave-uniform, kept in SGPRs
    NmNode rec = nm_ld_node(g.nodes, SUB ? top : 0u);
    int first = nm_octant(rec, kx, ky, kz);
    unsigned om = nm_visit_mask(rec, first);
    bool at_root = true;
    int work = 0;   // (BUDGET only; wave-uniform)
    for (;;) {
        if (BUDGET && work > budget) return true;
        // The list passes through one opaque definition per trip.  Without it the compiler carries the eight keys in TWO
        // register sets (one for this loop, one for the leaf scan below) and copies one into the other at every node --
        // 16 v_mov_b64 per trip, a quarter of the traversal's vector instructions; with it 8 (K-NN per frame 109 -> 103 ms).
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("" : "+v"(kk[k]));
        if (om == 0u) {
            if (at_root) break;
            const int c_prev = (int)((rec.info >> 8) & 7u);
            const uint32_t parent = rec.parent;
            rec = nm_ld_node(g.nodes, parent);
            at_root = parent == (SUB ? top : 0u);
            first = nm_octant(rec, kx, ky, kz);
            om = nm_visit_mask(rec, first) & ~((2u << nm_perm(c_prev ^ first)) - 1u);
            continue;
        }
        const int i = __builtin_ctz(om);
        om &= om - 1u;
        const int c = first ^ nm_perm(i);
        const uint32_t mask = rec.info & 255u;
        const NmNode crec = nm_ld_node(g.nodes, rec.first + (uint32_t)__popc(mask & ((1u << c) - 1u)));
        if (BUDGET) work += 24;
        // every lane tests (an inactive lane's result is masked out of the vote): no divergent region around the bound,
        // and the vote is a scalar compare of the mask (__any() goes through a vector select + compare)
        const bool nearer = nm_box_lb2(crec, qx, qy, qz) <= nm_key_d2(kk[K - 1]);
        if ((__builtin_amdgcn_ballot_w64(nearer) & act_mask) == 0ull) continue;
        const bool want = active && nearer;
        if ((crec.info & 255u) == 0u) {  // leaf
            // LDS-staged leaf scan: the wave fetches up to 64 vertices of the leaf with ONE coalesced vector load, every lane
            // scores them from LDS (same-address reads: broadcast).  A/B on the 800x800 frame, same call: scalar-path scan
            // (s_load_dwordx16 = 4 vertices per dependent load) 103.2 ms of K-NN per frame, this 100.9, vector load +
            // v_readlane broadcast 115.5; the leaf level keeps its optimum (~32 vertices per leaf: 101 vs 127-130 ms at ~120).
            __shared__ float4 nm_leaf_lds[BLK / 64][64];  // one stage per wave: every kernel that traverses is compiled
            float4* stage = nm_leaf_lds[threadIdx.x >> 6];
            const nm_f32x2 qyz = {qy, qz};          // with __launch_bounds__(BLK) and launched with that block size (NM_KNN_BLOCK; 64 for the pull kernels)
            const uint32_t ln = threadIdx.x & 63u;
            if (BUDGET) work += single ? 8 * (int)((crec.end - crec.first + 63u) >> 6) : 7 * (int)(crec.end - crec.first);
            if (single) {
                for (uint32_t p0 = crec.first; p0 < crec.end; p0 += 64) {
                    const uint32_t cnt = crec.end - p0 < 64u ? crec.end - p0 : 64u;
                    const float4 sv = g.sverts[p0 + (ln < cnt ? ln : 0u)];
                    unsigned long long key = nm_key(nm_dist2(qx, qy, qz, sv.x, sv.y, sv.z), nm_as_int(sv.w));
                    if (ln >= cnt) key = ~0ull;
                    for (unsigned long long cand = __builtin_amdgcn_ballot_w64(key < kk[K - 1]); cand; cand &= cand - 1ull) {
                        const int l = __builtin_ctzll(cand);
                        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(key & 0xffffffffull), l);
                        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(key >> 32), l);
                        const unsigned long long c = ((unsigned long long)hi << 32) | lo;
                        if (c < kk[K - 1]) nm_topk_insert<K>(kk, c);       // (wave-uniform: every lane holds the same list)
                    }
                }
                continue;
            }
            for (uint32_t p0 = crec.first; p0 < crec.end; p0 += 64) {
                const uint32_t cnt = crec.end - p0 < 64u ? crec.end - p0 : 64u;
                {   // staged as {y, z, index, x}: the scan below then finds (y, z) in an aligned register pair (one packed subtract /
                    // multiply for two axes without copies) and the index beside the register its squared distance is formed in
                    const float4 sv = g.sverts[p0 + (ln < cnt ? ln : 0u)];
                    stage[ln] = make_float4(sv.y, sv.z, sv.w, sv.x);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (uint32_t j0 = 0; j0 < cnt; j0 += 4) {
                    const float4 vv[4] = {stage[j0], stage[(j0 + 1) & 63u], stage[(j0 + 2) & 63u], stage[(j0 + 3) & 63u]};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (j0 + j < cnt) {
                            // declared arithmetic (dx*dx + dy*dy) + dz*dz, one rounding per operation; (y, z) as packed pairs
                            const nm_f32x2 dyz = qyz - nm_f32x2{vv[j].x, vv[j].y};
                            const nm_f32x2 syz = dyz * dyz;
                            const float dxv = nm_sub(qx, vv[j].w);
                            const unsigned long long key = nm_key(nm_add(nm_add(nm_mul(dxv, dxv), syz.x), syz.y), nm_as_int(vv[j].z));
                            if (want && key < kk[K - 1]) nm_topk_insert<K>(kk, key);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();  // (the next chunk overwrites the stage)
            }
        } else {
            rec = crec;
            at_root = false;
            first = nm_octant(rec, kx, ky, kz);
            om = nm_visit_mask(rec, first);
        }
    }
    return false;
}

// K-NN for the calling lane's query; the whole wave must call it (inactive lanes pass
// active=false).  Picks the cooperative traversal when the wave's queries are compact
// (bounding-box extent below a fraction of the root cube), lane-private traversals otherwise
// (e.g. randomly scattered points through the point-wise API).
template <int K, bool BUDGET = false, int BLK = NM_KNN_BLOCK>
__device__ __forceinline__ bool nm_knn_wave(const NmGridView& g, float qx, float qy, float qz, bool active,
                                            unsigned long long (&kk)[K], float init_d2 = NM_INF_F, int budget = 0) {
    // inactive lanes borrow an active lane's position so that they do not stretch the box
    const unsigned long long act = __ballot(active);
    if (act == 0ull) return false;
    const int src = __builtin_ctzll(act);
    const float sx = __shfl(qx, src), sy = __shfl(qy, src), sz = __shfl(qz, src);
    const float px = active ? qx : sx, py = active ? qy : sy, pz = active ? qz : sz;
    const float lox = nm_wave_min(px), hix = nm_wave_max(px);
    const float loy = nm_wave_min(py), hiy = nm_wave_max(py);
    const float loz = nm_wave_min(pz), hiz = nm_wave_max(pz);
    const float ext = nm_uniform_f(fmaxf(fmaxf(hix - lox, hiy - loy), hiz - loz));
    if (ext <= g.coop_extent) {
        return nm_knn_search_packet<K, BUDGET, false, BLK>(g, qx, qy, qz, active, nm_uniform_f(0.5f * (lox + hix)), nm_uniform_f(0.5f * (loy + hiy)),
                                               nm_uniform_f(0.5f * (loz + hiz)), kk, init_d2, budget);
    } else if (active) {
        nm_knn_search<K>(g, qx, qy, qz, kk, nullptr, init_d2);
    }
    return false;
}

// Lane -> query mapping.  Ray-structured launches (modes 1, 2) give each wave a tile of
// 16 adjacent rays x 4 consecutive samples (the most compact 64-query footprint, see above);
// importance samples, which are not regular in depth, go by depth buckets over 64 adjacent rays
// (s.order); point-wise launches (mode 0) take 64 consecutive points.
#ifndef NM_TILE_SAMPLES
// consecutive samples of a ray per tile (power of two); the tile has 64 / NM_TILE_SAMPLES adjacent rays.  Measured on the 800x800 frame
// (round 4, K-NN kernels per frame): 16 rays x 4 samples 83.6 ms, 32 x 2 84.5 ms, 64 x 1 (an 8x8 pixel patch at one depth, chained
// along the ray) 85.7 ms -- the coarse passes are not footprint-bound, the shape stays.
#define NM_TILE_SAMPLES 4
#endif
#define NM_TILE_RAYS (64 / NM_TILE_SAMPLES)
__host__ __device__ __forceinline__ int nm_chain_len(const NmPointSrc& s) { return (s.mode == 2 && !s.order && s.chain > 1) ? s.chain : 1; }
// `wave`: index of the 64-query packet within the launch (nm_launch_wave(): the wave's position in the grid; the pull kernels draw it from a counter)
__device__ __forceinline__ long long nm_launch_wave() { return ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; }
__device__ __forceinline__ bool nm_lane_query(const NmPointSrc& s, long long Q, long long& q, long long& r, int& p, int it, long long wave) {
    const int lane = threadIdx.x & 63;
    if (s.mode == 0) {
        const int L = s.lanes > 0 ? s.lanes : 64;
        q = r = wave * L + lane;
        p = 0;
        return lane < L && q < Q;
    }
    const long long R = Q / s.P;  // uniform
    if (s.order) {
        const long long wpg = ((long long)s.order_rays * s.P + 63) >> 6;  // waves per group
        const long long grp = wave / wpg;
        if (grp * s.order_rays >= R) { q = r = 0; p = 0; return false; }
        const unsigned id = s.order[wave * 64 + lane];  // wave*64 == grp*E + (wave - grp*wpg)*64
        const unsigned rl = id / (unsigned)s.P;
        r = grp * s.order_rays + rl;
        p = (int)(id - rl * (unsigned)s.P);
        q = r * s.P + p;
        return id != 0xffffu && r < R;
    }
    const int chain = nm_chain_len(s);
    const long long tiles_p = (s.P + NM_TILE_SAMPLES - 1) / NM_TILE_SAMPLES, groups_p = (tiles_p + chain - 1) / chain;
    const long long rb = wave / groups_p, sb = (wave - rb * groups_p) * chain + it;
    r = rb * NM_TILE_RAYS + lane / NM_TILE_SAMPLES;
    p = (int)(sb * NM_TILE_SAMPLES) + (lane % NM_TILE_SAMPLES);
    q = r * s.P + p;
    return r < R && p < s.P;
}
static inline unsigned nm_query_blocks(const NmPointSrc& s, long long Q) {
    long long waves;
    if (s.mode == 0) waves = (Q + (s.lanes > 0 ? s.lanes : 64) - 1) / (s.lanes > 0 ? s.lanes : 64);
    else if (s.order) waves = ((Q / s.P + s.order_rays - 1) / s.order_rays) * (((long long)s.order_rays * s.P + 63) / 64);
    else {
        const int chain = (s.mode == 2 && s.chain > 1) ? s.chain : 1;
        waves = ((Q / s.P + NM_TILE_RAYS - 1) / NM_TILE_RAYS) * (((s.P + NM_TILE_SAMPLES - 1) / NM_TILE_SAMPLES + chain - 1) / chain);
    }
    return (unsigned)((waves + 3) / 4);  // 4 waves per 256-thread block
}

// ------------------------------------------------ gather + interpolate the per-vertex codes
// interpolation(features, indices, weights) = sum_k features[idx_k] * w_k
// (models/frameworks/neumesh/neumesh.py:11-13), done by the wave that just found the neighbours:
// 8 lanes share one point, each lane owns a 16-byte chunk of the code vector (dim/4 chunks; chunks
// beyond 8 loop), so every table row is fetched as one contiguous 128-byte segment and the result
// is stored as a contiguous row.  k ascending, one rounding per multiply and per add.
__device__ __forceinline__ void nm_gather_interp(const float* __restrict__ table, int dim, const int (&bi)[8],
                                                 const float (&wk)[8], bool active, long long out_index,
                                                 float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 3, sub = lane & 7;
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
        const int pl = it * 8 + grp;  // lane that owns the point this 8-lane group works on
        const bool on = __shfl((int)active, pl) != 0;
        const long long o = __shfl(out_index, pl);
        int ii[8];
        float ww[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            ii[k] = __shfl(bi[k], pl);
            ww[k] = __shfl(wk[k], pl);
        }
        if (!on) continue;
        for (int chunk = sub; chunk < (dim >> 2); chunk += 8) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(table + (size_t)ii[k] * dim + 4 * chunk);
                a.x = __fadd_rn(a.x, __fmul_rn(v.x, ww[k]));
                a.y = __fadd_rn(a.y, __fmul_rn(v.y, ww[k]));
                a.z = __fadd_rn(a.z, __fmul_rn(v.z, ww[k]));
                a.w = __fadd_rn(a.w, __fmul_rn(v.w, ww[k]));
            }
            *reinterpret_cast<float4*>(out + o * dim + 4 * chunk) = a;
        }
    }
}

// stand-alone form for callers that bring their own neighbours (NeuMesh.forward_color)
__global__ __launch_bounds__(256) void nm_interp_kernel(const float* __restrict__ table, int dim,
                                                        const long long* __restrict__ idx64, const int* __restrict__ idx32,
                                                        const float* __restrict__ w, long long P, float* __restrict__ out) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = q < P;
    int bi[8];
    float wk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        bi[k] = active ? (idx64 ? (int)idx64[q * 8 + k] : idx32[q * 8 + k]) : 0;
        wk[k] = active ? w[q * 8 + k] : 0.f;
    }
    nm_gather_interp(table, dim, bi, wk, active, q, out);
}

// Upper bound of the squared K-th-neighbour distance of (x,y,z) from 8 KNOWN vertices (the neighbours of
// a nearby query): their largest exact distance to the new query.  Far from the surface the K-NN set
// barely changes between consecutive samples of a ray, so this is within a hair of the true radius,
// whereas "radius of the previous sample + step" (triangle inequality) over-covers the surface cap by a
// factor that grows with the distance to the surface.  `src_thread` (same wave) holds the indices.
// The neighbour lists live in LDS ([k][thread of the workgroup]) rather than in eight loop-carried registers per lane read through eight
// cross-lane shuffles: the chained kernels write a lane's list after every search and read the list of the lane they warm-start
// from (same wave: DS operations of a wave execute in order, no barrier needed).
template <int BLK>
__device__ __forceinline__ float nm_bound_from_neighbours_lds(const float* __restrict__ verts, const int (*nbr)[BLK], int src_thread,
                                                              bool usable, float x, float y, float z) {
    float worst = 0.f;
    bool ok = usable;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = nbr[k][src_thread];
        ok = ok && i != 0x7fffffff;
        if (ok) {
            const float vx = verts[3 * (size_t)i], vy = verts[3 * (size_t)i + 1], vz = verts[3 * (size_t)i + 2];
            worst = fmaxf(worst, nm_dist2(x, y, z, vx, vy, vz));
        }
    }
    return ok ? worst : NM_INF_F;
}

// ----------------------------------------------------------------------------- plain K-NN
template <int K>
__global__ __launch_bounds__(NM_KNN_BLOCK) void nm_knn_kernel(NmGridView g, NmPointSrc src, long long Q, int Kout,
                                                     long long* __restrict__ idx_out, float* __restrict__ d2_out) {
    long long q, r;
    int p;
    const bool active = nm_lane_query(src, Q, q, r, p, 0, nm_launch_wave());
    float x = 0.f, y = 0.f, z = 0.f, dep = 0.f;
    if (active) nm_fetch_point(src, r, p, x, y, z, dep);
    unsigned long long kk[K];
    nm_knn_wave<K>(g, x, y, z, active, kk);
    if (!active) return;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (k < Kout) {
            const int bi = nm_key_idx(kk[k]);
            const bool ok = bi != 0x7fffffff;
            idx_out[q * Kout + k] = ok ? (long long)bi : -1ll;
            d2_out[q * Kout + k] = ok ? nm_key_d2(kk[k]) : -1.0f;
        }
    }
}

// ------------------------------------------------- K-NN + weights + projected signed distance
// (models/mesh_grid.py:88-144 fused; nothing of shape [Q,8,3] is ever materialised)
// Any output pointer may be null.  ds_out is indexed by q (compact).
// Occupancy: NM_KNN_WAVES / NM_KNN_WAVES_CHAIN above.
// The outputs of one query from its neighbour keys (everything behind the search): shared by the traversal kernels and by the kernel
// that answers the deferred queries.  The whole wave calls it (the code gather is wave-cooperative).
// `list_pos`: position of this lane's query in the launch's lane list (packet index * 64 + lane): the record index of launches that store by list position
__device__ __forceinline__ void nm_distance_finish(const NmPointSrc& src, bool active, long long q, long long r, int p, long long list_pos, float x, float y, float z,
                                                   const float (&bd)[8], int (&bi)[8],   // squared distances, indices (bi is scratch afterwards)
                                                   const float* __restrict__ verts, const float* __restrict__ indicator, float w1,
                                                   float* __restrict__ ds_out, int* __restrict__ idx32_out, long long* __restrict__ idx64_out,
                                                   float* __restrict__ w_out, float* __restrict__ grad_out, float* __restrict__ radius_out,
                                                   const float* __restrict__ geo_table, int gdim, float* __restrict__ fg_out,
                                                   const float* __restrict__ col_table, int cdim, float* __restrict__ ft_out) {
    float wk[8], gr[3];
    float ds = 0.f;
    long long o = 0;
    if (active) {
        ds = nm_projected_distance8(x, y, z, bd, bi, verts, indicator, w1, wk, grad_out ? gr : nullptr);
        o = (src.order && src.out_by_slot) ? list_pos : nm_out_index(src, q, r, p);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            bi[k] = 0;
            wk[k] = 0.f;
        }
    }
    if (fg_out) nm_gather_interp(geo_table, gdim, bi, wk, active, o, fg_out);
    if (ft_out) nm_gather_interp(col_table, cdim, bi, wk, active, o, ft_out);
    if (!active) return;
    if (ds_out) ds_out[o] = ds;
    if (radius_out) radius_out[o] = nm_sqrt(bd[7]);
    if (idx32_out) {
        *reinterpret_cast<int4*>(idx32_out + o * 8) = make_int4(bi[0], bi[1], bi[2], bi[3]);
        *reinterpret_cast<int4*>(idx32_out + o * 8 + 4) = make_int4(bi[4], bi[5], bi[6], bi[7]);
    }
    if (idx64_out) {
#pragma unroll
        for (int k = 0; k < 8; ++k) idx64_out[o * 8 + k] = (long long)bi[k];
    }
    if (w_out) {
        *reinterpret_cast<float4*>(w_out + o * 8) = make_float4(wk[0], wk[1], wk[2], wk[3]);
        *reinterpret_cast<float4*>(w_out + o * 8 + 4) = make_float4(wk[4], wk[5], wk[6], wk[7]);
    }
    if (grad_out) {
        grad_out[o * 3] = gr[0];
        grad_out[o * 3 + 1] = gr[1];
        grad_out[o * 3 + 2] = gr[2];
    }
}

// nm_distance_body: the work of ONE wave on packet `wave` of the launch.  BLK = threads of the calling kernel's workgroup (sizes the LDS arrays).
template <bool CHAIN, bool BUDGET, int BLK>
__device__ __forceinline__ void nm_distance_body(const NmGridView& g, const NmPointSrc& src, long long Q, long long wave,
                                                 const float* __restrict__ verts,
                                                 const float* __restrict__ indicator, float w1,
                                                 float* __restrict__ ds_out, int* __restrict__ idx32_out,
                                                 long long* __restrict__ idx64_out,
                                                 float* __restrict__ w_out, float* __restrict__ grad_out,
                                                 float* __restrict__ radius_out,
                                                 const float* __restrict__ geo_table, int gdim, float* __restrict__ fg_out,
                                                 const float* __restrict__ col_table, int cdim, float* __restrict__ ft_out) {
    // Chained tiles (regular depth grids: probes, coarse samples): the wave walks `chain` consecutive
    // 4-sample tiles of its 16 rays; from the second tile on every lane starts its search from a
    // proven bound -- the K-th-neighbour radius of the LAST sample of the previous tile on the same
    // ray plus the depth gap to it (triangle inequality along a unit direction) -- instead of +INF.
    // (CHAIN = false is the plain single-tile kernel: no loop-carried state in its registers)
#ifdef NM_TESTING
    const long long nm_t0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    const int chain = CHAIN ? nm_chain_len(src) : 1;
    const int lane = threadIdx.x & 63;
    float prev_rad = NM_INF_F, prev_dep = 0.f;
    __shared__ int prev_bi[CHAIN ? 8 : 1][CHAIN ? BLK : 64];   // neighbours of each lane's previous sample (chained tiles only)
    for (int it = 0; it < chain; ++it) {
        long long q, r;
        int p;
        const bool active = nm_lane_query(src, Q, q, r, p, it, wave);
        float x = 0.f, y = 0.f, z = 0.f, dep = 0.f, init = NM_INF_F;
        if (active) {
            nm_fetch_point(src, r, p, x, y, z, dep);
            init = nm_init_bound(src, r, p);
        }
        if (CHAIN) {
            const float pr = __shfl(prev_rad, lane | (NM_TILE_SAMPLES - 1)), pd = __shfl(prev_dep, lane | (NM_TILE_SAMPLES - 1));
            if (it > 0 && pr < NM_INF_F) {
                const float b = (pr + fabsf(dep - pd)) * 1.0001f + 1e-5f;
                init = fminf(init, b * b);
            }
            // ... and from the exact distances to that sample's 8 neighbours (usually far tighter)
            const float nb = nm_bound_from_neighbours_lds(verts, prev_bi, threadIdx.x | (NM_TILE_SAMPLES - 1), it > 0 && active, x, y, z);
            init = fminf(init, nb);
        }
        float bd[8];
        int bi[8];
        unsigned long long kk[8];
        if (BUDGET) {
            if (nm_knn_wave<8, true, BLK>(g, x, y, z, active, kk, init, src.budget)) {   // (wave-uniform) over budget: hand the queries on
                const unsigned long long am = __builtin_amdgcn_ballot_w64(active);
                const int n = __popcll(am);
                int old = 0;
                if (lane == 0) old = atomicAdd(src.defer_count, n);
                old = __builtin_amdgcn_readfirstlane(old);
                if (old + n <= src.defer_cap) {
                    if (active) {
                        const int at = old + __popcll(am & ((1ull << lane) - 1ull));
                        src.defer_list[at] = (int)q;
                        src.defer_bound2[at] = nm_key_d2(kk[7]);   // K-th best so far (or the warm-start bound / +INF): proven
                    }
                    continue;
                }
                // the list is full: finish here after all.  What this wave reserved inside the list stays empty (-1) -- the counter has
                // moved past it, and the slots would otherwise hold entries of an earlier launch
                if (active && old + __popcll(am & ((1ull << lane) - 1ull)) < src.defer_cap) src.defer_list[old + __popcll(am & ((1ull << lane) - 1ull))] = -1;
                nm_knn_wave<8, false, BLK>(g, x, y, z, active, kk, init);
            }
        } else {
            nm_knn_wave<8, false, BLK>(g, x, y, z, active, kk, init);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            bd[k] = nm_key_d2(kk[k]);
            bi[k] = nm_key_idx(kk[k]);
        }
        prev_rad = (active && bi[7] != 0x7fffffff) ? nm_sqrt(bd[7]) : NM_INF_F;
        prev_dep = dep;
        if (CHAIN) {
#pragma unroll
            for (int k = 0; k < 8; ++k) prev_bi[k][threadIdx.x] = active ? bi[k] : 0x7fffffff;
        }
        nm_distance_finish(src, active, q, r, p, wave * 64 + lane, x, y, z, bd, bi, verts, indicator, w1, ds_out, idx32_out, idx64_out, w_out, grad_out, radius_out,
                           geo_table, gdim, fg_out, col_table, cdim, ft_out);
    }
#ifdef NM_TESTING
    nm_wave_log_write(nm_t0, wave);
#endif
}

template <bool CHAIN, bool BUDGET = false>
__global__ __launch_bounds__(NM_KNN_BLOCK, CHAIN ? NM_KNN_WAVES_CHAIN : NM_KNN_WAVES) void nm_distance_kernel(NmGridView g, NmPointSrc src, long long Q,
                                                          const float* __restrict__ verts,
                                                          const float* __restrict__ indicator, float w1,
                                                          float* __restrict__ ds_out, int* __restrict__ idx32_out,
                                                          long long* __restrict__ idx64_out,
                                                          float* __restrict__ w_out, float* __restrict__ grad_out,
                                                          float* __restrict__ radius_out,
                                                          const float* __restrict__ geo_table, int gdim, float* __restrict__ fg_out,
                                                          const float* __restrict__ col_table, int cdim, float* __restrict__ ft_out) {
    nm_distance_body<CHAIN, BUDGET, NM_KNN_BLOCK>(g, src, Q, nm_launch_wave(), verts, indicator, w1, ds_out, idx32_out, idx64_out, w_out, grad_out, radius_out,
                                                  geo_table, gdim, fg_out, col_table, cdim, ft_out);
}

// ------------------------------------------------------------------------------ pull kernels (K-NN beside the MLP kernels of another ray chunk)
// The K-NN kernels are bound by vector-instruction issue, the MLP kernels by the matrix pipe, and the two pipes of a SIMD run side by side
// (MI355X_MICROARCH.md, wave scheduling).  When a call is rendered as several ray chunks on several streams, one chunk's K-NN kernels can therefore
// run UNDER another chunk's MLP kernels -- if both are resident on the same SIMDs: two MLP workgroups per CU leave 512 - 2 x 192 = 128 registers per
// SIMD lane and 16 KB (geometry) / 8 KB (colour) of LDS, room for exactly one K-NN wave per SIMD.  A grid-mapped K-NN launch never leaves that
// room: its own pending workgroups refill every slot its waves free, and an MLP workgroup (192 registers on all four SIMDs of ONE CU + 72 KB of LDS at
// once) starves until the K-NN grid is exhausted.  The pull form:
//   * workgroup = ONE wave (64 threads: its registers and its 1.25 / 3 KB of LDS are freed the moment it exits); the launch has at most
//     (SIMDs of the chip) x (waves per SIMD the kernel is compiled for) of them, and every wave draws packet indices from a counter until none are left;
//   * NmYield (one per device, shared by all streams): `wanted` = MLP launches queued or running (raised / lowered by one-thread kernels around
//     them), occ[simd] = pull waves resident on that SIMD (from HW_REG_HW_ID / HW_REG_XCC_ID);
//   * while wanted > 0 a SIMD keeps at most `cap` pull waves: the others exit before their next packet, a wave that arrives on a full SIMD exits at
//     once -- so the MLP workgroups find room within one packet's time (~0.1 ms), wherever the launch order put them; a launch never gives up
//     its last `min_alive` waves (the SIMD counts are shared by every K-NN launch in flight);
//   * with wanted == 0 the launch fills the chip like the grid-mapped form.
// Which wave evaluates which packet changes no result bit (every packet's outputs depend on its own queries only).
struct NmYield {
    int wanted;          // MLP launches that want room (queued or running)
    int pad[15];
    int occ[2048 * 4];   // pull waves per SIMD, index = nm_simd_key()
};
struct NmPull {
    unsigned long long* next;   // next[0]: packet counter of THIS launch, next[1]: its waves still at work (both zeroed by the host, stream-ordered)
    long long npackets;
    NmYield* y;                 // nullptr: never yield
    int cap;                    // pull waves a SIMD keeps while MLP launches want room
    int min_alive;              // a wave only leaves while at least this many waves of its launch stay at work: the SIMD counts are shared by every K-NN
                                // launch in flight, and a launch whose waves all sat beside another launch's must not be left without workers
};
// (xcc, se, sh, cu, simd) of the calling wave -> [0, 8192)
__device__ __forceinline__ int nm_simd_key() {
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_REG_HW_ID: simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13]
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11));   // HW_REG_XCC_ID [3:0]
    return (int)((((xcc & 7u) << 8 | ((hw >> 13) & 7u) << 5 | ((hw >> 12) & 1u) << 4 | ((hw >> 8) & 15u)) << 2) | ((hw >> 4) & 3u));
}
// true: this wave leaves (its occ entry and its share of the launch's alive count are already given back)
__device__ __forceinline__ bool nm_pull_should_leave(const NmPull& pl, int key) {
    if (!pl.y) return false;
    int leave = 0;
    if ((threadIdx.x & 63) == 0 && __hip_atomic_load(&pl.y->wanted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0 &&
        __hip_atomic_load(&pl.y->occ[key], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > pl.cap) {
        unsigned long long* alive = pl.next + 1;
        if ((long long)atomicAdd(alive, ~0ull) - 1 >= (long long)pl.min_alive) {   // (atomicAdd of -1)
            if (atomicSub(&pl.y->occ[key], 1) > pl.cap) leave = 1;
            else atomicAdd(&pl.y->occ[key], 1);
        }
        if (!leave) atomicAdd(alive, 1ull);
    }
    return __builtin_amdgcn_readfirstlane(leave) != 0;
}
__device__ __forceinline__ long long nm_pull_next(const NmPull& pl) {
    unsigned long long w = 0;
    if ((threadIdx.x & 63) == 0) w = atomicAdd(pl.next, 1ull);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(w & 0xffffffffull)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(w >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
#define NM_PULL_LOOP(BODY)                                                                           \
    const int nm_key_ = pl.y ? nm_simd_key() : 0;                                                     \
    if (pl.y && (threadIdx.x & 63) == 0) {                                                            \
        atomicAdd(&pl.y->occ[nm_key_], 1);                                                            \
        atomicAdd(pl.next + 1, 1ull);                                                                 \
    }                                                                                                 \
    for (;;) {                                                                                        \
        if (nm_pull_should_leave(pl, nm_key_)) return;                                                \
        const long long wave = nm_pull_next(pl);                                                      \
        if (wave >= pl.npackets) break;                                                               \
        BODY;                                                                                         \
    }                                                                                                 \
    if (pl.y && (threadIdx.x & 63) == 0) {                                                            \
        atomicSub(&pl.y->occ[nm_key_], 1);                                                            \
        atomicAdd(pl.next + 1, ~0ull);                                                                \
    }

template <bool CHAIN>
__global__ __launch_bounds__(64, CHAIN ? NM_KNN_WAVES_CHAIN : NM_KNN_WAVES) void nm_distance_pull_kernel(NmGridView g, NmPointSrc src, long long Q, NmPull pl,
                                                          const float* __restrict__ verts,
                                                          const float* __restrict__ indicator, float w1,
                                                          float* __restrict__ ds_out, int* __restrict__ idx32_out,
                                                          long long* __restrict__ idx64_out,
                                                          float* __restrict__ w_out, float* __restrict__ grad_out,
                                                          float* __restrict__ radius_out,
                                                          const float* __restrict__ geo_table, int gdim, float* __restrict__ fg_out,
                                                          const float* __restrict__ col_table, int cdim, float* __restrict__ ft_out) {
    NM_PULL_LOOP((nm_distance_body<CHAIN, false, 64>(g, src, Q, wave, verts, indicator, w1, ds_out, idx32_out, idx64_out, w_out, grad_out, radius_out,
                                                     geo_table, gdim, fg_out, col_table, cdim, ft_out)))
}
__global__ void nm_yield_add_kernel(NmYield* y, int delta) { atomicAdd(&y->wanted, delta); }

// ------------------------------------------------------------------------ the deferred queries of a small launch
// A small launch (a training batch: 10^4 ... 10^5 points) lives as long as its slowest wave, and the slowest waves hold queries near the
// medial axis of the object -- every surface patch about equally far, box bounds prune little, the traversal degenerates into a walk over
// most of the index by ONE wave (tools/knn_wave_times.py: wave life p50 0.11 ms, p99 0.5 ms, max 1.2 ms = the launch).  Such waves give
// up after NmPointSrc.budget work units (nm_distance_kernel<false, true>), leaving each query with the K-th best distance found so far --
// a PROVEN upper bound -- and the query is finished by a whole wave of its own: lane (c1, c2) searches the level-2 subtree with child
// digits (c1, c2) with a lane-private traversal started from that bound (64 subtrees in parallel instead of one after the other), the
// 64 sorted lists are merged by eight wave-wide minima, and nm_distance_deferred_kernel computes the outputs (weights, projected
// distance, gathers) from the merged keys.  Same candidate arithmetic and (d2, index) order, subtrees skipped only on the proven
// bound: exact.  (First built as an exhaustive scan of all vertices by the whole chip: 26 k wave instructions per query, twice what the
// hardest traversal itself costs -- 0.4-0.7 ms for the deferred queries of one launch; dropped.)

// nm_knn_search restricted to the subtree of node `top` (never climbs above it); kk already holds the starting list
template <int K>
__device__ __forceinline__ void nm_knn_search_subtree(const NmGridView& g, uint32_t top, float qx, float qy, float qz, unsigned long long (&kk)[K]) {
    const uint32_t kx = nm_float_key(qx), ky = nm_float_key(qy), kz = nm_float_key(qz);
    NmNode rec = g.nodes[top];
    if (nm_box_lb2(rec, qx, qy, qz) > nm_key_d2(kk[K - 1])) return;
    if ((rec.info & 255u) == 0u) {   // the subtree is one leaf
        for (uint32_t p = rec.first; p < rec.end; ++p) {
            const float4 v = g.sverts[p];
            const unsigned long long key = nm_key(nm_dist2(qx, qy, qz, v.x, v.y, v.z), nm_as_int(v.w));
            if (key < kk[K - 1]) nm_topk_insert<K>(kk, key);
        }
        return;
    }
    int first = nm_octant(rec, kx, ky, kz);
    unsigned om = nm_visit_mask(rec, first);
    bool at_top = true;
    for (;;) {
        if (om == 0u) {
            if (at_top) break;
            const int c_prev = (int)((rec.info >> 8) & 7u);
            const uint32_t parent = rec.parent;
            rec = g.nodes[parent];
            at_top = parent == top;
            first = nm_octant(rec, kx, ky, kz);
            om = nm_visit_mask(rec, first) & ~((2u << nm_perm(c_prev ^ first)) - 1u);
            continue;
        }
        const int i = nm_ctz(om);
        om &= om - 1u;
        const int c = first ^ nm_perm(i);
        const uint32_t mask = rec.info & 255u;
        const NmNode crec = g.nodes[rec.first + (uint32_t)nm_popc(mask & ((1u << c) - 1u))];
        if (nm_box_lb2(crec, qx, qy, qz) > nm_key_d2(kk[K - 1])) continue;
        if ((crec.info & 255u) == 0u) {
            for (uint32_t p = crec.first; p < crec.end; ++p) {
                const float4 v = g.sverts[p];
                const unsigned long long key = nm_key(nm_dist2(qx, qy, qz, v.x, v.y, v.z), nm_as_int(v.w));
                if (key < kk[K - 1]) nm_topk_insert<K>(kk, key);
            }
        } else {
            rec = crec;
            at_top = false;
            first = nm_octant(rec, kx, ky, kz);
            om = nm_visit_mask(rec, first);
        }
    }
}

__device__ __forceinline__ void nm_query_rp(const NmPointSrc& s, long long q, long long& r, int& p) {
    if (s.mode == 0) { r = q; p = 0; return; }
    r = q / s.P;
    p = (int)(q - r * s.P);
}

__device__ __forceinline__ unsigned long long nm_wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)(v & 0xffffffffull), o), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), o);
        const unsigned long long w = ((unsigned long long)hi << 32) | lo;
        v = w < v ? w : v;
    }
    return v;
}

// unit of work = (64 consecutive deferred queries) x (level-2 subtree c1, c2): keys[(slot * 64 + 8 c1 + c2) * 8 + k] = the K = 8 best keys of
// query `slot` inside that subtree that beat its bound (placeholders where there are none)
// (only subtrees that hold a candidate write their list; bit `sub` of mask[slot] says so -- zeroed by the host before the launch)
__global__ __launch_bounds__(NM_KNN_BLOCK) void nm_knn_subtree_kernel(NmGridView g, NmPointSrc src, const float* __restrict__ bound2,
                                                                      unsigned long long* __restrict__ keys, unsigned long long* __restrict__ mask) {
    int n = *src.defer_count;
    if (n > src.defer_cap) n = src.defer_cap;
    if (n <= 0) return;
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long units = (long long)((n + 63) >> 6) * 64;
    const NmNode root = nm_ld_node(g.nodes, 0);
    const uint32_t rmask = root.info & 255u;
    for (long long u = wave0; u < units; u += nwaves) {
        const int tile = (int)(u >> 6), sub = (int)(u & 63), c1 = sub >> 3, c2 = sub & 7;
        const int slot = tile * 64 + lane;
        const int qd = slot < n ? src.defer_list[slot] : -1;
        const bool active = qd >= 0;
        float x = 0.f, y = 0.f, z = 0.f, dep = 0.f, b2 = NM_INF_F;
        if (active) {
            long long r;
            int p;
            nm_query_rp(src, (long long)qd, r, p);
            nm_fetch_point(src, r, p, x, y, z, dep);
            b2 = bound2[slot];
        }
        unsigned long long kk[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) kk[k] = nm_key(b2, 0x7fffffff);
        // the unit's subtree (wave-uniform): level-2 node (c1, c2); a level-1 LEAF belongs to c2 = 0, a root that is a leaf to unit 0
        uint32_t top = 0xffffffffu;
        bool leaf = false;
        if (rmask == 0u) {
            if (sub == 0) { top = 0u; leaf = true; }
        } else if ((rmask >> c1) & 1u) {
            const uint32_t i1 = root.first + (uint32_t)__popc(rmask & ((1u << c1) - 1u));
            const NmNode n1 = nm_ld_node(g.nodes, i1);
            const uint32_t m1 = n1.info & 255u;
            if (m1 == 0u) {
                if (c2 == 0) { top = i1; leaf = true; }
            } else if ((m1 >> c2) & 1u) {
                top = n1.first + (uint32_t)__popc(m1 & ((1u << c2) - 1u));
                leaf = (nm_ld_node(g.nodes, top).info & 255u) == 0u;
            }
        }
        if (top != 0xffffffffu) {
            if (leaf) {
                const NmNode t = nm_ld_node(g.nodes, top);
                for (uint32_t p = t.first; p < t.end; ++p) {
                    const float4 v = nm_ld_vert(g.sverts, p);
                    const unsigned long long key = nm_key(nm_dist2(x, y, z, v.x, v.y, v.z), nm_as_int(v.w));
                    if (active && key < kk[7]) nm_topk_insert<8>(kk, key);
                }
            } else {
                // packet centre: the mean position of the tile's queries is not needed -- any point orders the children validly; lane 0's
                const float rx = nm_uniform_f(x), ry = nm_uniform_f(y), rz = nm_uniform_f(z);
                nm_knn_search_packet<8, false, true>(g, x, y, z, active, rx, ry, rz, kk, b2, 0, top);
            }
        }
        if (active && nm_key_idx(kk[0]) != 0x7fffffff) {
            unsigned long long* o = keys + ((long long)slot * 64 + sub) * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = kk[k];
            atomicOr(mask + slot, 1ull << sub);
        }
    }
}

__global__ __launch_bounds__(NM_KNN_BLOCK) void nm_distance_deferred_kernel(NmGridView g, NmPointSrc src, const unsigned long long* __restrict__ keys,
                                                                            const unsigned long long* __restrict__ mask,
                                                                            const float* __restrict__ verts, const float* __restrict__ indicator, float w1,
                                                                            float* __restrict__ ds_out, int* __restrict__ idx32_out,
                                                                            long long* __restrict__ idx64_out, float* __restrict__ w_out,
                                                                            float* __restrict__ grad_out, float* __restrict__ radius_out,
                                                                            const float* __restrict__ geo_table, int gdim, float* __restrict__ fg_out,
                                                                            const float* __restrict__ col_table, int cdim, float* __restrict__ ft_out) {
    int n = *src.defer_count;
    if (n > src.defer_cap) n = src.defer_cap;
    const long long slot = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (((slot >> 6) << 6) >= n) return;   // (whole wave)
    const int qd = slot < n ? src.defer_list[slot] : -1;
    const bool active = qd >= 0;
    long long q = 0, r = 0;
    int p = 0;
    float x = 0.f, y = 0.f, z = 0.f, dep = 0.f;
    unsigned long long kk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) kk[k] = nm_key(NM_INF_F, 0x7fffffff);
    if (active) {
        q = (long long)qd;
        nm_query_rp(src, q, r, p);
        nm_fetch_point(src, r, p, x, y, z, dep);
        const unsigned long long* in = keys + slot * 64 * 8;
        for (unsigned long long m = mask[slot]; m; m &= m - 1ull) {
            const int c = __builtin_ctzll(m);
            for (int k = 0; k < 8; ++k) {          // each subtree's list is ascending
                const unsigned long long key = in[c * 8 + k];
                if (!(key < kk[7])) break;
                nm_topk_insert<8>(kk, key);
            }
        }
    }
    float bd[8];
    int bi[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        bd[k] = nm_key_d2(kk[k]);
        bi[k] = nm_key_idx(kk[k]);
    }
    nm_distance_finish(src, active, q, r, p, slot, x, y, z, bd, bi, verts, indicator, w1, ds_out, idx32_out, idx64_out, w_out, grad_out, radius_out,
                       geo_table, gdim, fg_out, col_table, cdim, ft_out);
}

// ---------------------------------------------------- bounded near/far straight from the probes
// compute_bounded_near_far (renderer.py:66-102) needs, per ray, only the FIRST and the LAST of the
// P regular probes whose projected distance is below the threshold (min / max of the masked depths;
// the depths increase with the probe index).  Every point inside the object has ds < 0, so for a
// ray that crosses it most probes lie BETWEEN those two and are never evaluated here: a wave owns 16
// rays (8 with S = 8), walks their probes forward (S per ray per step, warm-started from the step before, as in the
// chained tiles above) until every ray has its first hit, then backward from the far end until every
// ray has its last one.  Results are the reference's exactly: the same probes decide, the skipped
// ones cannot change a min / max.  Replaces a P-probe K-NN pass + the reduction kernel + the
// [R,P] probe array.
// S = probes per ray and step (4: 16 rays per wave, 8: 8 rays per wave -- half as many serial steps per wave, up to 4 more probes
// per ray and walk; nm_render_rays picks)
template <int S, int BLK>
__device__ __forceinline__ void nm_probe_bounds_body(const NmGridView& g, long long wave, const float* __restrict__ rays_o,
                                                     const float* __restrict__ dirn, const float* __restrict__ nearfar0,
                                                     long long R, int P, float thresh, const float* __restrict__ verts,
                                                     const float* __restrict__ indicator, float w1,
                                                     float* __restrict__ nearfar,
                                                     unsigned long long* __restrict__ searched) {
    constexpr int LOG_S = S == 8 ? 3 : 2;
    constexpr unsigned SMASK = (1u << S) - 1u;
    const int lane = threadIdx.x & 63, sub = lane & (S - 1), quad = lane & ~(S - 1);
    const long long r = wave * (64 / S) + (lane >> LOG_S);
    const bool valid = r < R;
    float ox = 0.f, oy = 0.f, oz = 0.f, dx = 0.f, dy = 0.f, dz = 0.f, n0 = 0.f, f0 = 1.f;
    if (valid) {
        ox = rays_o[3 * r]; oy = rays_o[3 * r + 1]; oz = rays_o[3 * r + 2];
        dx = dirn[3 * r]; dy = dirn[3 * r + 1]; dz = dirn[3 * r + 2];
        n0 = nearfar0[2 * r]; f0 = nearfar0[2 * r + 1];
    }
    const int T = (P + S - 1) >> LOG_S;
    int first_idx = -1, last_idx = -1;
    unsigned n_searched = 0;  // probes this wave searched (profiling: one atomic per wave at the end)
    // one step: probe p of this lane's ray; returns ds (and the K-th-neighbour radius for the next warm start)
    __shared__ int nbr[8][BLK];  // neighbours of each lane's last probe
#pragma unroll
    for (int k = 0; k < 8; ++k) nbr[k][threadIdx.x] = 0x7fffffff;
    const int wave_base = threadIdx.x & ~63;
    // src_lane: the lane of this ray whose last probe is the closest one already evaluated (-1: none)
    auto probe = [&](int p, bool act, float init, int src_lane, float& dep, float& rad) -> float {
        dep = nm_lerp_depth(n0, f0, nm_linspace01(p < P ? p : P - 1, P));
        const float x = nm_add(ox, nm_mul(dep, dx)), y = nm_add(oy, nm_mul(dep, dy)), z = nm_add(oz, nm_mul(dep, dz));
        if (src_lane >= 0) init = fminf(init, nm_bound_from_neighbours_lds(verts, nbr, wave_base | src_lane, act, x, y, z));
        if (searched) n_searched += (unsigned)__popcll(__ballot(act));
        unsigned long long kk[8];
        nm_knn_wave<8, false, BLK>(g, x, y, z, act, kk, init);
        float bd[8], wk[8];
        int bi[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            bd[k] = nm_key_d2(kk[k]);
            bi[k] = nm_key_idx(kk[k]);
            nbr[k][threadIdx.x] = act ? bi[k] : 0x7fffffff;
        }
        rad = (act && bi[7] != 0x7fffffff) ? nm_sqrt(bd[7]) : NM_INF_F;
        return act ? nm_projected_distance8(x, y, z, bd, bi, verts, indicator, w1, wk, nullptr) : NM_INF_F;
    };
    // ---- forward: first hit
    {
        float prev_rad = NM_INF_F, prev_dep = 0.f;
        for (int t = 0; t < T; ++t) {
            if (!__any(valid && first_idx < 0)) break;
            const int p = S * t + sub;
            const bool act = valid && first_idx < 0 && p < P;
            const float pr = __shfl(prev_rad, quad | (S - 1)), pd = __shfl(prev_dep, quad | (S - 1));
            float dep, rad, init = NM_INF_F;
            // (the depth is needed for the bound before the search: same formula as inside probe())
            const float dep_here = nm_lerp_depth(n0, f0, nm_linspace01(p < P ? p : P - 1, P));
            if (t > 0 && pr < NM_INF_F) {
                const float b = (pr + fabsf(dep_here - pd)) * 1.0001f + 1e-5f;
                init = b * b;
            }
            const float ds = probe(p, act, init, t > 0 ? (quad | (S - 1)) : -1, dep, rad);
            prev_rad = rad;
            prev_dep = dep;
            const unsigned hm = (unsigned)((__ballot(act && ds < thresh) >> quad) & SMASK);
            if (hm && first_idx < 0) first_idx = S * t + __builtin_ctz(hm);
        }
    }
    // ---- backward: last hit (strictly after the first one; none => the first one is also the last)
    {
        float prev_rad = NM_INF_F, prev_dep = 0.f;
        bool started = false;
        for (int t = T - 1; t >= 0; --t) {
            if (first_idx >= 0 && last_idx < 0 && S * t + (S - 1) <= first_idx) last_idx = first_idx;  // nothing left above the first hit
            if (!__any(valid && first_idx >= 0 && last_idx < 0)) break;
            const int p = S * t + sub;
            const bool act = valid && first_idx >= 0 && last_idx < 0 && p < P && p > first_idx;
            const float pr = __shfl(prev_rad, quad), pd = __shfl(prev_dep, quad);
            float dep, rad, init = NM_INF_F;
            const float dep_here = nm_lerp_depth(n0, f0, nm_linspace01(p < P ? p : P - 1, P));
            if (started && pr < NM_INF_F) {
                const float b = (pr + fabsf(dep_here - pd)) * 1.0001f + 1e-5f;
                init = b * b;
            }
            const float ds = probe(p, act, init, started ? quad : -1, dep, rad);
            prev_rad = rad;
            prev_dep = dep;
            started = true;
            const unsigned hm = (unsigned)((__ballot(act && ds < thresh) >> quad) & SMASK);
            if (hm && first_idx >= 0 && last_idx < 0) last_idx = S * t + (31 - __builtin_clz(hm));
            if (first_idx >= 0 && last_idx < 0 && S * t <= first_idx) last_idx = first_idx;  // this tile held the first hit
        }
        if (first_idx >= 0 && last_idx < 0) last_idx = first_idx;
    }
    if (valid && sub == 0) {
        const float mn = first_idx >= 0 ? nm_lerp_depth(n0, f0, nm_linspace01(first_idx, P)) : 1e10f;
        const float mx = first_idx >= 0 ? nm_lerp_depth(n0, f0, nm_linspace01(last_idx, P)) : -1e10f;
        nm_ray_bounds_finish(mn, mx, n0, f0, nearfar + 2 * r, nearfar + 2 * r + 1);
    }
    if (searched && lane == 0 && n_searched) atomicAdd(searched, (unsigned long long)n_searched);
}
template <int S>
__global__ __launch_bounds__(NM_KNN_BLOCK, NM_KNN_WAVES_PROBE) void nm_probe_bounds_kernel(NmGridView g, const float* __restrict__ rays_o,
                                                                 const float* __restrict__ dirn, const float* __restrict__ nearfar0,
                                                                 long long R, int P, float thresh, const float* __restrict__ verts,
                                                                 const float* __restrict__ indicator, float w1,
                                                                 float* __restrict__ nearfar,
                                                                 unsigned long long* __restrict__ searched) {
    nm_probe_bounds_body<S, NM_KNN_BLOCK>(g, nm_launch_wave(), rays_o, dirn, nearfar0, R, P, thresh, verts, indicator, w1, nearfar, searched);
}
// pull form (see nm_distance_pull_kernel)
template <int S>
__global__ __launch_bounds__(64, NM_KNN_WAVES_PROBE) void nm_probe_bounds_pull_kernel(NmGridView g, NmPull pl, const float* __restrict__ rays_o,
                                                                 const float* __restrict__ dirn, const float* __restrict__ nearfar0,
                                                                 long long R, int P, float thresh, const float* __restrict__ verts,
                                                                 const float* __restrict__ indicator, float w1,
                                                                 float* __restrict__ nearfar,
                                                                 unsigned long long* __restrict__ searched) {
    NM_PULL_LOOP((nm_probe_bounds_body<S, 64>(g, wave, rays_o, dirn, nearfar0, R, P, thresh, verts, indicator, w1, nearfar, searched)))
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
// Rounds 1-3 sorted the keys exactly (bitonic network in LDS, ~80 passes over up to 8192 keys: 6.5 ms per frame, the
// largest per-ray kernel after round 4's other changes).  A wave only needs its 64 samples to be NEAR each other in depth, so
// round 4 orders by BUCKET: 1024 depth buckets between the group's smallest and largest kept depth (finer than the vertex
// spacing on the benchmark scene), histogram + scan + scatter, ids ascending inside EVERY bucket (the list is the same on every run).
// wgt (optional, [R][cap]): samples with wgt == 0 are dropped (padding behind the kept ones); counter (optional): += kept.
#define NM_ORDER_BUCKETS 1024
static inline size_t nm_order_lds_bytes(int n) { return (size_t)((n + 63) & ~63) * (4 + 2) + 64; }   // depth + id per list entry
__global__ __launch_bounds__(256) void nm_rays_order_kernel(const float* __restrict__ d, long long R, int cap, int off,
                                                            int P, int G, unsigned short* __restrict__ order,
                                                            const float* __restrict__ wgt, unsigned long long* __restrict__ counter) {
    extern __shared__ float nm_order_smem[];
    __shared__ unsigned nm_hist[NM_ORDER_BUCKETS];    // counts, then running scatter positions
    __shared__ unsigned nm_start[NM_ORDER_BUCKETS];   // first list position of each bucket
    __shared__ unsigned nm_lo, nm_hi;                 // order-preserving keys of the smallest / largest kept depth
    __shared__ unsigned nm_wsum[4];
    const long long grp = blockIdx.x;
    const int n = G * P, E = (n + 63) & ~63;
    float* dep = nm_order_smem;                                              // [E] depth of entry i (NaN bit pattern = dropped)
    unsigned short* ids = reinterpret_cast<unsigned short*>(dep + E);        // [E] the list
    const int t = threadIdx.x;
    for (int b = t; b < NM_ORDER_BUCKETS; b += 256) nm_hist[b] = 0u;
    if (t == 0) { nm_lo = 0xffffffffu; nm_hi = 0u; }
    __syncthreads();
    unsigned lo = 0xffffffffu, hi = 0u;
    for (int i = t; i < E; i += 256) {
        const int rl = i / P;
        const long long r = grp * G + rl;
        float v = __int_as_float(0x7fc00000);
        if (i < n && r < R) {
            const long long g = r * cap + off + (i - rl * P);
            if (!(wgt && wgt[g] == 0.0f)) {
                v = d[g];
                if (v != v) v = 3.0e38f;                                     // (a NaN depth still gets a place: the last bucket -- but it must not
                else {                                                       //  stretch the bucket range and collapse every real depth into bucket 0)
                    const unsigned k = nm_float_key(v);
                    lo = k < lo ? k : lo;
                    hi = k > hi ? k : hi;
                }
            }
        }
        dep[i] = v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned l2 = (unsigned)__shfl_xor((int)lo, o), h2 = (unsigned)__shfl_xor((int)hi, o);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((t & 63) == 0) { atomicMin(&nm_lo, lo); atomicMax(&nm_hi, hi); }
    __syncthreads();
    const unsigned klo = nm_lo, khi = nm_hi;
    // bucket of a depth: linear in the order-preserving key space would follow the float spacing, so go through the values
    const unsigned ulo = klo ^ ((klo >> 31) ? 0x80000000u : 0xffffffffu), uhi = khi ^ ((khi >> 31) ? 0x80000000u : 0xffffffffu);
    const float dlo = __uint_as_float(ulo), dhi = klo <= khi ? __uint_as_float(uhi) : dlo;
    const float scale = (float)NM_ORDER_BUCKETS / fmaxf(dhi - dlo, 1e-30f);
    auto bucket = [&](float v) -> int {
        const float x = (v - dlo) * scale;
        return x >= (float)(NM_ORDER_BUCKETS - 1) ? NM_ORDER_BUCKETS - 1 : (x > 0.f ? (int)x : 0);
    };
    for (int i = t; i < E; i += 256) {
        const float v = dep[i];
        if (v == v) atomicAdd(&nm_hist[bucket(v)], 1u);
    }
    __syncthreads();
    {   // exclusive scan of the bucket counts: 4 buckets per thread, wave scan, 4 wave totals
        unsigned c[4], tot = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { c[u] = nm_hist[4 * t + u]; tot += c[u]; }
        unsigned incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned up = (unsigned)__shfl_up((int)incl, o);
            if ((t & 63) >= o) incl += up;
        }
        if ((t & 63) == 63) nm_wsum[t >> 6] = incl;
        __syncthreads();
        unsigned base = incl - tot;
        for (int w = 0; w < (t >> 6); ++w) base += nm_wsum[w];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            nm_start[4 * t + u] = base;
            nm_hist[4 * t + u] = base;
            base += c[u];
        }
    }
    __syncthreads();
    const unsigned kept = nm_wsum[0] + nm_wsum[1] + nm_wsum[2] + nm_wsum[3];
    for (int i = t; i < E; i += 256) {
        const float v = dep[i];
        if (v == v) ids[atomicAdd(&nm_hist[bucket(v)], 1u)] = (unsigned short)i;
    }
    __syncthreads();
    // ids ascending inside a bucket (the scatter above lands in atomic order, which differs from run to run): buckets hold a handful of
    // entries -- insertion sort; a crowded bucket (many samples at one depth) gets a shell sort by its thread, so that the list, and with it
    // which lane evaluates which sample, is the same on every run
    for (int b = t; b < NM_ORDER_BUCKETS; b += 256) {
        const unsigned s0 = nm_start[b], s1 = nm_hist[b], cnt = s1 - s0;
        unsigned gap = 1u;
        while (cnt > 64u && gap < cnt / 3u) gap = 3u * gap + 1u;
        for (; gap >= 1u; gap /= 3u) {
            for (unsigned a = s0 + gap; a < s1; ++a) {
                const unsigned short x = ids[a];
                unsigned q = a;
                while (q >= s0 + gap && ids[q - gap] > x) { ids[q] = ids[q - gap]; q -= gap; }
                ids[q] = x;
            }
            if (gap == 1u) break;
        }
    }
    __syncthreads();
    for (int i = t; i < E; i += 256) order[grp * E + i] = (unsigned)i < kept ? ids[i] : (unsigned short)0xffffu;
    if (counter && t == 0 && kept) atomicAdd(counter, (unsigned long long)kept);
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
