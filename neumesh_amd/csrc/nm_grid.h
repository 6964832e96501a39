// nm_grid.h -- sparse-octree spatial index over the mesh vertices + exact K-NN traversal.
//
// Replaces what the reference gets from the external FRNN CUDA package
// (models/mesh_grid.py:64-74 build, :109-119 query: K nearest, r=100 => unbounded, sorted).
//
// Structure (built once per mesh, nm_grid_build.h):
//   * root cube [origin, origin+root_size)^3 enclosing all vertices, subdivided L times;
//   * vertices sorted by the Morton code of their level-L cell ("leaf"), ties by vertex index,
//     stored as float4 {x, y, z, bitcast(index)} so a candidate costs one 16-byte load;
//   * only NON-EMPTY nodes exist.  One 64-byte record per node (NmNode):
//       first : internal -> index of its first child record; leaf -> first vertex (sorted array)
//       end   : leaf -> one past its last vertex
//       parent: index of the parent record (the traversal walks back up through it: no stack)
//       info  : bits 0-7 child-occupancy mask, bits 8-10 the node's own child digit
//       lo/hi : the node's TIGHT bounding box (fp32), already expanded by the rounding slack
//       ck*   : box centre as order-preserving integers; om_* : the occupancy mask pre-permuted into
//               near-first visiting order for each of the 8 octants the query can lie in
//     Children of a node are stored contiguously in child-digit order, so child c lives at
//     first + popcount(mask & ((1 << c) - 1)).
//     A mesh is a 2-D surface: inside a cell its vertices fill a thin slab, so the tight box
//     gives a far better lower bound than the cell cube for queries that are not right on the
//     surface (the 256-probe near/far search, every sample more than a cell away from the mesh).
//
// Query (this file, shared by the device kernels and by the host-side logic check in
// tests/hostcheck): depth-first, nearest-child-first traversal with an exact box lower bound,
// pruned against the current K-th best.  The traversal is STACKLESS: the child order at a
// node is (octant of q relative to the node centre) XOR a fixed permutation, so on the way
// back up the position in the parent's order is recomputed from the child's Morton digit; the
// whole state is {node record, remaining-children mask} in registers.
//
// Exactness: candidate distances use the declared arithmetic (fp32, dx = q - v,
// d2 = (dx*dx + dy*dy) + dz*dz, no FMA), order is (d2, index) ascending.  A subtree is
// skipped only if a conservative lower bound of every fp32 d2 inside it exceeds the K-th best.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NM_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define NM_HD inline
struct float4 { float x, y, z, w; };
#endif

#define NM_MAX_LEVEL 8
// Automatic leaf level: the smallest depth whose occupied leaves hold <= NM_LEAF_TARGET vertices on average.  The
// traversal is bound by the latency of its scalar node loads, not by vertex arithmetic, so fat leaves win: measured
// on the 800x800 frame (V = 1.4e5) K-NN 146 / 110 / 115 / 151 ms at ~120 / 32 / 8 / 2.5 vertices per leaf, and on the
// V = 1e6 stress mesh 31.4 / 29.7 / 37.9 ms at ~88 / 22 / 5.5 per leaf.
#define NM_LEAF_TARGET 40.0
#define NM_INF_F 3.402823466e+38f

struct alignas(64) NmNode {  // 64 bytes
    uint32_t first, end, parent, info;
    float lox, loy, loz;
    uint32_t ckx;             // box centre as order-preserving integer keys (nm_float_key)
    float hix, hiy, hiz;
    uint32_t cky;
    uint32_t ckz;
    uint32_t om_lo, om_hi;    // child masks in near-first visiting order for octant 0-3 / 4-7, one byte each
    uint32_t pad;
};

struct NmGridView {
    int L;                     // leaf level of the octree the records were built from
    int V;                     // number of vertices
    float coop_extent;         // waves whose queries fit a box of this edge search cooperatively
    int n_nodes;               // number of node records
    const NmNode* nodes;       // node records, root = 0, levels stored one after another
    const float4* sverts;      // [V + 4] sorted vertices (.w = bit pattern of the original
                               // index), padded with 4 far-away dummies for batched scans
};

NM_HD float nm_mul(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fmul_rn(a, b);
#else
    return a * b;  // host build uses -ffp-contract=off
#endif
}
NM_HD float nm_add(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fadd_rn(a, b);
#else
    return a + b;
#endif
}
NM_HD float nm_sub(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fsub_rn(a, b);
#else
    return a - b;
#endif
}

// declared distance arithmetic
NM_HD float nm_dist2(float qx, float qy, float qz, float vx, float vy, float vz) {
    const float dx = nm_sub(qx, vx), dy = nm_sub(qy, vy), dz = nm_sub(qz, vz);
    return nm_add(nm_add(nm_mul(dx, dx), nm_mul(dy, dy)), nm_mul(dz, dz));
}

NM_HD int nm_as_int(float f) {
    union { float f; int i; } u;
    u.f = f;
    return u.i;
}
NM_HD float nm_as_float(int i) {
    union { float f; int i; } u;
    u.i = i;
    return u.f;
}

NM_HD int nm_ctz(unsigned v) { return __builtin_ctz(v); }

NM_HD int nm_popc(unsigned v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(v);
#else
    return __builtin_popcount(v);
#endif
}

// near-first child permutation {0,1,2,4,3,5,6,7} packed in nibbles; it is an involution.
NM_HD int nm_perm(int i) { return (int)((0x76534210u >> (4 * i)) & 7u); }

// Occupancy mask re-ordered into near-first VISITING order for a query whose octant is `first`:
// bit i of the result = mask bit (first ^ perm(i)).  XOR-ing child digits by `first` is three
// conditional bit-block swaps; perm only exchanges ordinals 3 and 4.
NM_HD unsigned nm_ordered_mask(unsigned mask, int first) {
    unsigned m = mask & 255u;
    if (first & 1) m = ((m & 0x55u) << 1) | ((m & 0xAAu) >> 1);
    if (first & 2) m = ((m & 0x33u) << 2) | ((m & 0xCCu) >> 2);
    if (first & 4) m = ((m & 0x0Fu) << 4) | ((m & 0xF0u) >> 4);
    return (m & 0xE7u) | ((m & 0x08u) << 1) | ((m & 0x10u) >> 1);
}

// conservative lower bound of the fp32 squared distance from q to any vertex inside the node:
// distance to its (slack-expanded) tight box, times (1 - 1e-5) for the rounding of this
// expression and of the candidate distances themselves.
NM_HD float nm_box_lb2(const NmNode& n, float qx, float qy, float qz) {
    const float ax = fmaxf(fmaxf(n.lox - qx, qx - n.hix), 0.0f);
    const float ay = fmaxf(fmaxf(n.loy - qy, qy - n.hiy), 0.0f);
    const float az = fmaxf(fmaxf(n.loz - qz, qz - n.hiz), 0.0f);
    return (ax * ax + ay * ay + az * az) * 0.99999f;
}

// float -> unsigned with the same ordering (negative floats: all bits flipped; others: sign set)
NM_HD uint32_t nm_float_key(float f) {
    const uint32_t u = (uint32_t)nm_as_int(f);
    return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}

// octant of the (key-converted) point relative to the node's box centre
NM_HD int nm_octant(const NmNode& n, uint32_t kx, uint32_t ky, uint32_t kz) {
    return (kx >= n.ckx ? 1 : 0) | (ky >= n.cky ? 2 : 0) | (kz >= n.ckz ? 4 : 0);
}

// children still to visit (bit i = i-th child in near-first order) for a query in octant `first`
NM_HD unsigned nm_visit_mask(const NmNode& n, int first) {
    return ((first & 4 ? n.om_hi : n.om_lo) >> (8 * (first & 3))) & 255u;
}

// ---- top-K list as packed keys: (bits of d2) << 32 | index.  d2 >= 0, so the unsigned 64-bit
// order of the keys IS the (d2, index) lexicographic order: one compare per test.
NM_HD unsigned long long nm_key(float d2, int idx) {
    return ((unsigned long long)(uint32_t)nm_as_int(d2) << 32) | (unsigned long long)(uint32_t)idx;
}
NM_HD float nm_key_d2(unsigned long long k) { return nm_as_float((int)(uint32_t)(k >> 32)); }
NM_HD int nm_key_idx(unsigned long long k) { return (int)(uint32_t)k; }

// insert `key`, known to be smaller than the current last entry; branch-free bubble
template <int K>
NM_HD void nm_topk_insert(unsigned long long (&kk)[K], unsigned long long key) {
    kk[K - 1] = key;
#pragma unroll
    for (int p = K - 1; p > 0; --p) {
        const unsigned long long a = kk[p - 1], b = kk[p];
        const bool sw = b < a;
        kk[p - 1] = sw ? b : a;
        kk[p] = sw ? a : b;
    }
}

// Exact K-NN of (qx,qy,qz).  On return kk holds the K best keys ascending ((d2, index) order);
// unfilled slots (V < K) keep d2 = init, index = INT32_MAX.
// init_d2: every slot starts at this squared distance with index INT32_MAX; pass +INF for a cold
// search, or a PROVEN upper bound of the K-th neighbour's squared distance for a warm start (at
// least K real vertices then beat the placeholders, so none survives).
// STATS (host logic check only): stats[0] += node records tested, stats[1] += vertices scanned.
template <int K, bool STATS = false>
NM_HD void nm_knn_search(const NmGridView& g, float qx, float qy, float qz, unsigned long long (&kk)[K],
                         long long* stats = nullptr, float init_d2 = NM_INF_F) {
#pragma unroll
    for (int k = 0; k < K; ++k) kk[k] = nm_key(init_d2, 0x7fffffff);
    const uint32_t kx = nm_float_key(qx), ky = nm_float_key(qy), kz = nm_float_key(qz);
    NmNode rec = g.nodes[0];
    int first = nm_octant(rec, kx, ky, kz);
    unsigned om = nm_visit_mask(rec, first);  // children still to visit, near-first
    bool at_root = true;
    for (;;) {
        if (om == 0u) {
            if (at_root) break;
            const int c_prev = (int)((rec.info >> 8) & 7u);
            const uint32_t parent = rec.parent;
            rec = g.nodes[parent];
            at_root = parent == 0u;
            first = nm_octant(rec, kx, ky, kz);
            om = nm_visit_mask(rec, first) & ~((2u << nm_perm(c_prev ^ first)) - 1u);
            continue;
        }
        const int i = nm_ctz(om);
        om &= om - 1u;
        const int c = first ^ nm_perm(i);
        const uint32_t mask = rec.info & 255u;
        const NmNode crec = g.nodes[rec.first + (uint32_t)nm_popc(mask & ((1u << c) - 1u))];
        if (STATS) stats[0] += 1;
        if (nm_box_lb2(crec, qx, qy, qz) > nm_key_d2(kk[K - 1])) continue;
        if ((crec.info & 255u) == 0u) {  // leaf
            if (STATS) stats[1] += (long long)(crec.end - crec.first);
            for (uint32_t p = crec.first; p < crec.end; ++p) {
                const float4 v = g.sverts[p];
                const unsigned long long key = nm_key(nm_dist2(qx, qy, qz, v.x, v.y, v.z), nm_as_int(v.w));
                if (key < kk[K - 1]) nm_topk_insert<K>(kk, key);
            }
        } else {
            rec = crec;
            at_root = false;
            first = nm_octant(rec, kx, ky, kz);
            om = nm_visit_mask(rec, first);
        }
    }
}
