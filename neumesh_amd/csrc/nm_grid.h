// nm_grid.h -- sparse-octree spatial index over the mesh vertices + exact K-NN traversal.
//
// Replaces what the reference gets from the external FRNN CUDA package
// (models/mesh_grid.py:64-74 build, :109-119 query: K nearest, r=100 => unbounded, sorted).
//
// Structure (built once per mesh, nm_grid_build.cpp):
//   * root cube [origin, origin+root_size)^3 enclosing all vertices, subdivided L times;
//   * vertices sorted by the Morton code of their level-L cell ("leaf"), ties by vertex index,
//     stored as float4 {x, y, z, bitcast(index)} so a candidate costs one 16-byte load;
//   * leaf_start[8^L + 1]: CSR offsets of each leaf into the sorted array (dense);
//   * mask[(8^L - 1)/7]: for every internal node (levels 0..L-1, Morton order, level offset
//     (8^l - 1)/7) one byte whose bit c says "child c holds at least one vertex".
//     Empty space costs one byte test, not a cell visit.
//
// Query (this file, shared by the device kernels and by the host-side logic check in
// tests/hostcheck): depth-first, nearest-child-first traversal with an exact box lower bound,
// pruned against the current K-th best.  The traversal is STACKLESS: the child order at a
// node is (octant of q relative to the node centre) XOR a fixed permutation, so on the way
// back up the position in the parent's order is recomputed from the child's Morton digit; the
// whole state is {level, Morton code, cell coords, next child ordinal} in scalar registers.
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

struct NmGridView {
    float ox, oy, oz;          // min corner of the root cube
    float root_size;           // edge length of the root cube
    float slack;               // absolute slack subtracted from box distances (rounding of the
                               // cell assignment / box corners), ~2e-6 * coordinate scale
    int L;                     // leaf level (1..NM_MAX_LEVEL)
    int V;                     // number of vertices
    const uint8_t* mask;       // internal-node child masks, levels 0..L-1
    const uint32_t* leaf_start;  // [8^L + 1]
    const float4* sverts;      // [V] sorted vertices, .w = bit pattern of the original index
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

NM_HD uint32_t nm_level_offset(int level) { return ((1u << (3 * level)) - 1u) / 7u; }

// near-first child permutation {0,1,2,4,3,5,6,7} packed in nibbles; it is an involution.
NM_HD int nm_perm(int i) { return (int)((0x76534210u >> (4 * i)) & 7u); }

NM_HD float nm_cell_size(const NmGridView& g, int level) {
#if defined(__HIP_DEVICE_COMPILE__)
    return ldexpf(g.root_size, -level);
#else
    return std::ldexp(g.root_size, -level);
#endif
}

// conservative lower bound of the fp32 squared distance from q to anything assigned to the
// level-`level` cell (cx,cy,cz): per-axis gap shrunk by a relative 1e-6 and the absolute slack.
NM_HD float nm_box_lb2(const NmGridView& g, float qx, float qy, float qz, int cx, int cy, int cz,
                       float cs) {
    const float lox = g.ox + (float)cx * cs, loy = g.oy + (float)cy * cs, loz = g.oz + (float)cz * cs;
    float ax = fmaxf(fmaxf(lox - qx, qx - (lox + cs)), 0.0f);
    float ay = fmaxf(fmaxf(loy - qy, qy - (loy + cs)), 0.0f);
    float az = fmaxf(fmaxf(loz - qz, qz - (loz + cs)), 0.0f);
    ax = fmaxf(ax * 0.999999f - g.slack, 0.0f);
    ay = fmaxf(ay * 0.999999f - g.slack, 0.0f);
    az = fmaxf(az * 0.999999f - g.slack, 0.0f);
    return (ax * ax + ay * ay + az * az) * 0.999999f;
}

NM_HD int nm_octant(const NmGridView& g, float qx, float qy, float qz, int ix, int iy, int iz, float cs) {
    const float mx = g.ox + ((float)ix + 0.5f) * cs;
    const float my = g.oy + ((float)iy + 0.5f) * cs;
    const float mz = g.oz + ((float)iz + 0.5f) * cs;
    return (qx >= mx ? 1 : 0) | (qy >= my ? 2 : 0) | (qz >= mz ? 4 : 0);
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
template <int K>
NM_HD void nm_knn_search(const NmGridView& g, float qx, float qy, float qz, float (&bd)[K], int (&bi)[K]) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        bd[k] = NM_INF_F;
        bi[k] = 0x7fffffff;
    }
    const int L = g.L;
    int level = 0;
    uint32_t code = 0;
    int ix = 0, iy = 0, iz = 0;
    int child = 0;  // next child ordinal in near-first order
    uint32_t mask = g.mask[0];
    int first = nm_octant(g, qx, qy, qz, 0, 0, 0, g.root_size);
    for (;;) {
        if (child == 8) {
            if (level == 0) break;
            const int c_prev = (int)(code & 7u);
            code >>= 3;
            ix >>= 1;
            iy >>= 1;
            iz >>= 1;
            --level;
            first = nm_octant(g, qx, qy, qz, ix, iy, iz, nm_cell_size(g, level));
            mask = g.mask[nm_level_offset(level) + code];
            child = nm_perm(c_prev ^ first) + 1;
            continue;
        }
        const int c = first ^ nm_perm(child);
        ++child;
        if (!((mask >> c) & 1u)) continue;
        const int cx = (ix << 1) | (c & 1), cy = (iy << 1) | ((c >> 1) & 1), cz = (iz << 1) | ((c >> 2) & 1);
        const float cs = nm_cell_size(g, level + 1);
        if (nm_box_lb2(g, qx, qy, qz, cx, cy, cz, cs) > bd[K - 1]) continue;
        const uint32_t ccode = (code << 3) | (uint32_t)c;
        if (level + 1 == L) {
            const uint32_t beg = g.leaf_start[ccode], end = g.leaf_start[ccode + 1];
            for (uint32_t p = beg; p < end; ++p) {
                const float4 v = g.sverts[p];
                const float d = nm_dist2(qx, qy, qz, v.x, v.y, v.z);
                const int idx = nm_as_int(v.w);
                if (nm_topk_accepts<K>(bd, bi, d, idx)) nm_topk_insert<K>(bd, bi, d, idx);
            }
        } else {
            ++level;
            code = ccode;
            ix = cx;
            iy = cy;
            iz = cz;
            mask = g.mask[nm_level_offset(level) + code];
            first = nm_octant(g, qx, qy, qz, ix, iy, iz, cs);
            child = 0;
        }
    }
}
