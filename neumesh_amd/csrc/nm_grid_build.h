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
    std::vector<NmNode> nodes;   // all levels, root first
    std::vector<float4> sverts;  // V + 4 (padding)
};

NM_HD uint32_t nm_spread3(uint32_t v) {  // 8 bits -> every third bit
    v &= 0xffu;
    v = (v | (v << 8)) & 0x00f00fu;
    v = (v | (v << 4)) & 0x0c30c3u;
    v = (v | (v << 2)) & 0x249249u;
    return v;
}
static inline uint32_t nm_compact3(uint32_t v) {  // inverse of nm_spread3
    v &= 0x249249u;
    v = (v | (v >> 2)) & 0x0c30c3u;
    v = (v | (v >> 4)) & 0x00f00fu;
    v = (v | (v >> 8)) & 0x0000ffu;
    return v;
}
NM_HD uint32_t nm_morton(uint32_t x, uint32_t y, uint32_t z) {
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
    g.slack = 4e-6f * (amax + g.root_size);
    g.V = (int)V;

    std::vector<uint32_t> codes((size_t)V);
    int L = leaf_level;
    if (L <= 0) {
        // smallest depth with <= NM_LEAF_TARGET vertices per occupied leaf on average
        for (L = 1; L < NM_MAX_LEVEL; ++L) {
            for (int64_t i = 0; i < V; ++i) codes[(size_t)i] = nm_leaf_code(g, L, verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
            std::vector<uint32_t> s(codes);
            std::sort(s.begin(), s.end());
            const size_t occ = (size_t)(std::unique(s.begin(), s.end()) - s.begin());
            if ((double)V / (double)occ <= NM_LEAF_TARGET) break;
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
    g.sverts.assign((size_t)V + 4, float4{NM_INF_F, NM_INF_F, NM_INF_F, nm_as_float(0x7fffffff)});
    for (size_t p = 0; p < (size_t)V; ++p) {
        const uint32_t i = order[p];
        float4 v;
        v.x = verts[3 * i];
        v.y = verts[3 * i + 1];
        v.z = verts[3 * i + 2];
        v.w = nm_as_float((int)i);
        g.sverts[p] = v;
    }

    // per level: sorted unique node codes, tight boxes, first-child / first-vertex index
    struct Lvl {
        std::vector<uint32_t> code, first;          // Morton code; first child (local) or first vertex
        std::vector<float> lo, hi;                  // tight boxes, 3 floats per node
    };
    std::vector<Lvl> lv((size_t)L + 1);
    {
        Lvl& leaf = lv[(size_t)L];
        for (size_t p = 0; p < (size_t)V; ++p) {
            const uint32_t c = codes[order[p]];
            if (leaf.code.empty() || leaf.code.back() != c) {
                leaf.code.push_back(c);
                leaf.first.push_back((uint32_t)p);
                for (int a = 0; a < 3; ++a) { leaf.lo.push_back(NM_INF_F); leaf.hi.push_back(-NM_INF_F); }
            }
            const size_t n = leaf.code.size() - 1;
            const float xyz[3] = {g.sverts[p].x, g.sverts[p].y, g.sverts[p].z};
            for (int a = 0; a < 3; ++a) {
                leaf.lo[3 * n + a] = std::min(leaf.lo[3 * n + a], xyz[a]);
                leaf.hi[3 * n + a] = std::max(leaf.hi[3 * n + a], xyz[a]);
            }
        }
        g.occupied_leaves = (int)leaf.code.size();
    }
    for (int l = L - 1; l >= 0; --l) {
        const Lvl& ch = lv[(size_t)l + 1];
        Lvl& cur = lv[(size_t)l];
        for (size_t j = 0; j < ch.code.size(); ++j) {
            const uint32_t pc = ch.code[j] >> 3;
            if (cur.code.empty() || cur.code.back() != pc) {
                cur.code.push_back(pc);
                cur.first.push_back((uint32_t)j);
                for (int a = 0; a < 3; ++a) { cur.lo.push_back(NM_INF_F); cur.hi.push_back(-NM_INF_F); }
            }
            const size_t n = cur.code.size() - 1;
            for (int a = 0; a < 3; ++a) {
                cur.lo[3 * n + a] = std::min(cur.lo[3 * n + a], ch.lo[3 * j + a]);
                cur.hi[3 * n + a] = std::max(cur.hi[3 * n + a], ch.hi[3 * j + a]);
            }
        }
    }
    std::vector<uint32_t> off((size_t)L + 2, 0u);
    for (int l = 0; l <= L; ++l) off[(size_t)l + 1] = off[(size_t)l] + (uint32_t)lv[(size_t)l].code.size();
    g.nodes.assign((size_t)off[(size_t)L + 1], NmNode{});
    for (int l = 0; l <= L; ++l) {
        const Lvl& cur = lv[(size_t)l];
        size_t parent_local = 0;
        for (size_t n = 0; n < cur.code.size(); ++n) {
            NmNode r{};
            uint32_t mask = 0;
            if (l == L) {
                r.first = cur.first[n];
                r.end = (n + 1 < cur.code.size()) ? cur.first[n + 1] : (uint32_t)V;
            } else {
                const Lvl& ch = lv[(size_t)l + 1];
                r.first = off[(size_t)l + 1] + cur.first[n];
                r.end = 0;
                for (size_t j = cur.first[n]; j < ch.code.size() && (ch.code[j] >> 3) == cur.code[n]; ++j) mask |= 1u << (ch.code[j] & 7u);
            }
            if (l == 0) {
                r.parent = 0;
            } else {
                const Lvl& par = lv[(size_t)l - 1];
                while (par.code[parent_local] != (cur.code[n] >> 3)) ++parent_local;
                r.parent = off[(size_t)l - 1] + (uint32_t)parent_local;
            }
            r.info = mask | ((cur.code[n] & 7u) << 8);
            // tight box, expanded so that it certainly contains every fp32 vertex assigned to it
            // and absorbs the rounding of (box - q) for queries within ~10x the scene scale
            float blo[3], bhi[3];
            for (int a = 0; a < 3; ++a) {
                const float e = g.slack + 1e-6f * std::max(std::fabs(cur.lo[3 * n + a]), std::fabs(cur.hi[3 * n + a]));
                blo[a] = cur.lo[3 * n + a] - e;
                bhi[a] = cur.hi[3 * n + a] + e;
            }
            r.lox = blo[0]; r.loy = blo[1]; r.loz = blo[2];
            r.hix = bhi[0]; r.hiy = bhi[1]; r.hiz = bhi[2];
            r.ckx = nm_float_key(0.5f * (blo[0] + bhi[0]));
            r.cky = nm_float_key(0.5f * (blo[1] + bhi[1]));
            r.ckz = nm_float_key(0.5f * (blo[2] + bhi[2]));
            r.om_lo = r.om_hi = 0;
            for (int fo = 0; fo < 8; ++fo) {
                const uint32_t om = nm_ordered_mask(mask, fo);
                if (fo < 4) r.om_lo |= om << (8 * fo);
                else r.om_hi |= om << (8 * (fo - 4));
            }
            r.pad = 0;
            g.nodes[(size_t)off[(size_t)l] + n] = r;
        }
    }
    return true;
}

static inline NmGridView nm_host_view(const NmHostGrid& g) {
    NmGridView v;
    v.L = g.L;
    v.V = g.V;
    v.coop_extent = 0.75f * g.root_size;
    v.n_nodes = (int)g.nodes.size();
    v.nodes = g.nodes.data();
    v.sverts = g.sverts.data();
    return v;
}
