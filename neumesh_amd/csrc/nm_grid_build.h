// nm_grid_build.h -- host-side construction of the sparse-octree index (see nm_grid.h).
// One-off per mesh (reference: models/mesh_grid.py:64-74 builds FRNN's grid once and caches it).
// Deterministic: vertices are ordered by (leaf Morton code, vertex index).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "nm_grid.h"

struct NmHostGrid {
    float ox = 0, oy = 0, oz = 0, root_size = 1, slack = 0;
    int L = 1;
    int V = 0;
    int occupied_leaves = 0;
    std::vector<uint8_t> mask;         // (8^L - 1)/7
    std::vector<uint32_t> leaf_start;  // 8^L + 1
    std::vector<float4> sverts;        // V
};

static inline uint32_t nm_spread3(uint32_t v) {  // 8 bits -> every third bit
    v &= 0xffu;
    v = (v | (v << 8)) & 0x00f00fu;
    v = (v | (v << 4)) & 0x0c30c3u;
    v = (v | (v << 2)) & 0x249249u;
    return v;
}
static inline uint32_t nm_morton(uint32_t x, uint32_t y, uint32_t z) {
    return nm_spread3(x) | (nm_spread3(y) << 1) | (nm_spread3(z) << 2);
}

static inline uint32_t nm_leaf_code(const NmHostGrid& g, int L, float x, float y, float z) {
    const int n = 1 << L;
    const float inv = (float)n / g.root_size;
    int ix = (int)std::floor((x - g.ox) * inv), iy = (int)std::floor((y - g.oy) * inv), iz = (int)std::floor((z - g.oz) * inv);
    ix = std::min(std::max(ix, 0), n - 1);
    iy = std::min(std::max(iy, 0), n - 1);
    iz = std::min(std::max(iz, 0), n - 1);
    return nm_morton((uint32_t)ix, (uint32_t)iy, (uint32_t)iz);
}

// verts: [V,3].  leaf_level 0 => automatic.  Returns false on invalid input (NaN/Inf, V<1).
static inline bool nm_build_host_grid(const float* verts, int64_t V, int leaf_level, NmHostGrid& g) {
    if (V < 1 || V > 0x7ffffff0LL) return false;
    float lo[3] = {NM_INF_F, NM_INF_F, NM_INF_F}, hi[3] = {-NM_INF_F, -NM_INF_F, -NM_INF_F};
    for (int64_t i = 0; i < V; ++i)
        for (int a = 0; a < 3; ++a) {
            const float c = verts[3 * i + a];
            if (!std::isfinite(c)) return false;
            lo[a] = std::min(lo[a], c);
            hi[a] = std::max(hi[a], c);
        }
    float ext = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
    float amax = 0.f;
    for (int a = 0; a < 3; ++a) amax = std::max(amax, std::max(std::fabs(lo[a]), std::fabs(hi[a])));
    if (!(ext > 0.f)) ext = std::max(1e-3f, 1e-3f * amax);  // all vertices coincide
    g.root_size = ext * 1.001f + 1e-6f * std::max(amax, 1.0f);
    g.ox = 0.5f * (lo[0] + hi[0]) - 0.5f * g.root_size;
    g.oy = 0.5f * (lo[1] + hi[1]) - 0.5f * g.root_size;
    g.oz = 0.5f * (lo[2] + hi[2]) - 0.5f * g.root_size;
    g.slack = 2e-6f * (amax + g.root_size);
    g.V = (int)V;

    std::vector<uint32_t> codes((size_t)V);
    int L = leaf_level;
    if (L <= 0) {
        // smallest depth with <= 12 vertices per occupied leaf on average
        for (L = 1; L < NM_MAX_LEVEL; ++L) {
            for (int64_t i = 0; i < V; ++i) codes[(size_t)i] = nm_leaf_code(g, L, verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
            std::vector<uint32_t> s(codes);
            std::sort(s.begin(), s.end());
            const size_t occ = (size_t)(std::unique(s.begin(), s.end()) - s.begin());
            if ((double)V / (double)occ <= 12.0) break;
        }
    }
    L = std::min(std::max(L, 1), NM_MAX_LEVEL);
    g.L = L;
    for (int64_t i = 0; i < V; ++i) codes[(size_t)i] = nm_leaf_code(g, L, verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);

    std::vector<uint32_t> order((size_t)V);
    for (int64_t i = 0; i < V; ++i) order[(size_t)i] = (uint32_t)i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        return codes[a] != codes[b] ? codes[a] < codes[b] : a < b;
    });

    const size_t n_leaves = (size_t)1 << (3 * L);
    g.leaf_start.assign(n_leaves + 1, 0u);
    g.sverts.resize((size_t)V);
    for (size_t p = 0; p < (size_t)V; ++p) {
        const uint32_t i = order[p];
        float4 v;
        v.x = verts[3 * i];
        v.y = verts[3 * i + 1];
        v.z = verts[3 * i + 2];
        v.w = nm_as_float((int)i);
        g.sverts[p] = v;
        g.leaf_start[codes[i] + 1]++;
    }
    g.occupied_leaves = 0;
    for (size_t c = 0; c < n_leaves; ++c) {
        if (g.leaf_start[c + 1]) g.occupied_leaves++;
        g.leaf_start[c + 1] += g.leaf_start[c];
    }
    // child masks, bottom-up
    g.mask.assign(nm_level_offset(L), 0);
    for (int level = L - 1; level >= 0; --level) {
        const size_t n = (size_t)1 << (3 * level);
        const uint32_t off = nm_level_offset(level);
        for (size_t m = 0; m < n; ++m) {
            uint8_t bits = 0;
            for (int c = 0; c < 8; ++c) {
                const size_t child = (m << 3) | (size_t)c;
                bool occ;
                if (level + 1 == L) occ = g.leaf_start[child + 1] > g.leaf_start[child];
                else occ = g.mask[nm_level_offset(level + 1) + child] != 0;
                if (occ) bits |= (uint8_t)(1u << c);
            }
            g.mask[off + m] = bits;
        }
    }
    return true;
}

static inline NmGridView nm_host_view(const NmHostGrid& g) {
    NmGridView v;
    v.ox = g.ox; v.oy = g.oy; v.oz = g.oz;
    v.root_size = g.root_size;
    v.slack = g.slack;
    v.L = g.L;
    v.V = g.V;
    v.mask = g.mask.data();
    v.leaf_start = g.leaf_start.data();
    v.sverts = g.sverts.data();
    return v;
}
