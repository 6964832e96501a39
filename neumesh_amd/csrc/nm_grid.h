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
//       c     : box centre (orders the children near-first)
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
#define NM_INF_F 3.402823466e+38f

struct alignas(64) NmNode {  // 64 bytes
    uint32_t first, end, parent, info;
    float lox, loy, loz, cx;
    float hix, hiy, hiz, cy;
    float cz, pad0, pad1, pad2;
};

struct NmGridView {
    int L;                     // leaf level of the octree the records were built from
    int V;                     // number of vertices
    float coop_extent;         // waves whose queries fit a box of this edge search cooperatively
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

NM_HD int nm_octant(const NmNode& n, float qx, float qy, float qz) {
    return (qx >= n.cx ? 1 : 0) | (qy >= n.cy ? 2 : 0) | (qz >= n.cz ? 4 : 0);
}

template <int K>
NM_HD bool nm_topk_accepts(const float (&bd)[K], const int (&bi)[K], float d, int idx) {
    return (d < bd[K - 1]) || (d == bd[K - 1] && idx < bi[K - 1]);
}

// insert (d, idx), known to be lexicographically smaller than the current last entry.
template <int K>
NM_HD void nm_topk_insert(float (&bd)[K], int (&bi)[K], float d, int idx) {
    bd[K - 1] = d;
    bi[K - 1] = idx;
#pragma unroll
    for (int p = K - 1; p > 0; --p) {
        const bool sw = (bd[p] < bd[p - 1]) || (bd[p] == bd[p - 1] && bi[p] < bi[p - 1]);
        const float d0 = bd[p - 1], d1 = bd[p];
        const int i0 = bi[p - 1], i1 = bi[p];
        bd[p - 1] = sw ? d1 : d0;
        bd[p] = sw ? d0 : d1;
        bi[p - 1] = sw ? i1 : i0;
        bi[p] = sw ? i0 : i1;
    }
}

// Exact K-NN of (qx,qy,qz).  On return bd/bi hold the K best ascending by (d2, index);
// unfilled slots (V < K) keep d2 = +INF, index = INT32_MAX.
// STATS (host logic check only): stats[0] += node records tested, stats[1] += vertices scanned.
// init_d2: every slot starts at this squared distance with index INT32_MAX; pass +INF for a cold
// search, or a PROVEN upper bound of the K-th neighbour's squared distance for a warm start (at
// least K real vertices then beat the placeholders, so none survives).
template <int K, bool STATS = false>
NM_HD void nm_knn_search(const NmGridView& g, float qx, float qy, float qz, float (&bd)[K], int (&bi)[K],
                         long long* stats = nullptr, float init_d2 = NM_INF_F) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        bd[k] = init_d2;
        bi[k] = 0x7fffffff;
    }
    NmNode rec = g.nodes[0];
    int first = nm_octant(rec, qx, qy, qz);
    unsigned om = nm_ordered_mask(rec.info & 255u, first);  // children still to visit, near-first
    bool at_root = true;
    for (;;) {
        if (om == 0u) {
            if (at_root) break;
            const int c_prev = (int)((rec.info >> 8) & 7u);
            const uint32_t parent = rec.parent;
            rec = g.nodes[parent];
            at_root = parent == 0u;
            first = nm_octant(rec, qx, qy, qz);
            om = nm_ordered_mask(rec.info & 255u, first) & ~((2u << nm_perm(c_prev ^ first)) - 1u);
            continue;
        }
        const int i = nm_ctz(om);
        om &= om - 1u;
        const int c = first ^ nm_perm(i);
        const uint32_t mask = rec.info & 255u;
        const NmNode crec = g.nodes[rec.first + (uint32_t)nm_popc(mask & ((1u << c) - 1u))];
        if (STATS) stats[0] += 1;
        if (nm_box_lb2(crec, qx, qy, qz) > bd[K - 1]) continue;
        if ((crec.info & 255u) == 0u) {  // leaf
            if (STATS) stats[1] += (long long)(crec.end - crec.first);
            for (uint32_t p = crec.first; p < crec.end; ++p) {
                const float4 v = g.sverts[p];
                const float d = nm_dist2(qx, qy, qz, v.x, v.y, v.z);
                const int idx = nm_as_int(v.w);
                if (nm_topk_accepts<K>(bd, bi, d, idx)) nm_topk_insert<K>(bd, bi, d, idx);
            }
        } else {
            rec = crec;
            at_root = false;
            first = nm_octant(rec, qx, qy, qz);
            om = nm_ordered_mask(rec.info & 255u, first);
        }
    }
}
