// nm_knn.h -- exact K-NN over the octree (wave-cooperative traversal), the fused K-NN + weights + projected distance + code gather kernels,
// the deferred tails of small launches, the probe walk.  Device-only (included by nm_kernels.h).
#pragma once

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
            const nm_f32x2 qyz = {qy, qz};          // with __launch_bounds__(BLK) and launched with that block size (NM_KNN_BLOCK)
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
// `wave`: index of the 64-query packet within the launch (nm_launch_wave(): the wave's position in the grid)
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
template <int K, int BLK = NM_KNN_BLOCK>
__global__ __launch_bounds__(BLK) void nm_knn_kernel(NmGridView g, NmPointSrc src, long long Q, int Kout,
                                                     long long* __restrict__ idx_out, float* __restrict__ d2_out) {
    long long q, r;
    int p;
    const bool active = nm_lane_query(src, Q, q, r, p, 0, nm_launch_wave());
    float x = 0.f, y = 0.f, z = 0.f, dep = 0.f;
    if (active) nm_fetch_point(src, r, p, x, y, z, dep);
    unsigned long long kk[K];
    nm_knn_wave<K, false, BLK>(g, x, y, z, active, kk);
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

template <bool CHAIN, bool BUDGET = false, int BLK = NM_KNN_BLOCK>
__global__ __launch_bounds__(BLK, CHAIN ? NM_KNN_WAVES_CHAIN : NM_KNN_WAVES) void nm_distance_kernel(NmGridView g, NmPointSrc src, long long Q,
                                                          const float* __restrict__ verts,
                                                          const float* __restrict__ indicator, float w1,
                                                          float* __restrict__ ds_out, int* __restrict__ idx32_out,
                                                          long long* __restrict__ idx64_out,
                                                          float* __restrict__ w_out, float* __restrict__ grad_out,
                                                          float* __restrict__ radius_out,
                                                          const float* __restrict__ geo_table, int gdim, float* __restrict__ fg_out,
                                                          const float* __restrict__ col_table, int cdim, float* __restrict__ ft_out) {
    nm_distance_body<CHAIN, BUDGET, BLK>(g, src, Q, nm_launch_wave(), verts, indicator, w1, ds_out, idx32_out, idx64_out, w_out, grad_out, radius_out,
                                                  geo_table, gdim, fg_out, col_table, cdim, ft_out);
}


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
template <int S, int BLK = NM_KNN_BLOCK>
__global__ __launch_bounds__(BLK, NM_KNN_WAVES_PROBE) void nm_probe_bounds_kernel(NmGridView g, const float* __restrict__ rays_o,
                                                                 const float* __restrict__ dirn, const float* __restrict__ nearfar0,
                                                                 long long R, int P, float thresh, const float* __restrict__ verts,
                                                                 const float* __restrict__ indicator, float w1,
                                                                 float* __restrict__ nearfar,
                                                                 unsigned long long* __restrict__ searched) {
    nm_probe_bounds_body<S, BLK>(g, nm_launch_wave(), rays_o, dirn, nearfar0, R, P, thresh, verts, indicator, w1, nearfar, searched);
}
