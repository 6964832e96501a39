// tests/hostcheck/hostcheck.cpp -- TEST INFRASTRUCTURE ONLY.
// Compiles the host/device-shared headers of neumesh_amd/csrc (octree K-NN traversal, projected
// distance, per-ray stages) with g++ so their LOGIC can be checked against the oracle on a
// machine without a GPU.  The product never loads this library: libneumesh_hip.so runs the same
// headers on the device and fails loudly when no GPU is present.
//   g++ -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC -I../../neumesh_amd/csrc
#include <cstdint>
#include <cstring>
#include <vector>

#include "nm_grid_build.h"
#include "nm_rays.h"

struct HostGridHandle {
    NmHostGrid g;
    std::vector<float> verts;
};

template <int K>
static void knn_t(const NmGridView& v, const float* q, int64_t Q, int64_t* idx, float* d2) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < Q; ++i) {
        unsigned long long kk[K];
        nm_knn_search<K>(v, q[3 * i], q[3 * i + 1], q[3 * i + 2], kk);
        for (int k = 0; k < K; ++k) {
            const bool ok = nm_key_idx(kk[k]) != 0x7fffffff;
            idx[i * K + k] = ok ? nm_key_idx(kk[k]) : -1;
            d2[i * K + k] = ok ? nm_key_d2(kk[k]) : -1.0f;
        }
    }
}

extern "C" {

void* hc_grid_create(const float* verts, int64_t V, int leaf_level) {
    auto* h = new HostGridHandle();
    h->verts.assign(verts, verts + 3 * V);
    if (!nm_build_host_grid(verts, V, leaf_level, h->g)) {
        delete h;
        return nullptr;
    }
    return h;
}
void hc_grid_destroy(void* p) { delete (HostGridHandle*)p; }
int hc_grid_level(void* p) { return ((HostGridHandle*)p)->g.L; }
int hc_grid_occupied(void* p) { return ((HostGridHandle*)p)->g.occupied_leaves; }

int hc_knn(void* p, const float* q, int64_t Q, int K, int64_t* idx, float* d2) {
    const NmGridView v = nm_host_view(((HostGridHandle*)p)->g);
    switch (K) {
        case 1: knn_t<1>(v, q, Q, idx, d2); break;
        case 4: knn_t<4>(v, q, Q, idx, d2); break;
        case 8: knn_t<8>(v, q, Q, idx, d2); break;
        case 16: knn_t<16>(v, q, Q, idx, d2); break;
        case 32: knn_t<32>(v, q, Q, idx, d2); break;
        default: return 1;
    }
    return 0;
}

// warm-started search: bound[i] = upper bound of the distance from q[i] to its 8th nearest vertex
// (inflated exactly as nm_init_bound does on the device)
int hc_knn_warm(void* p, const float* q, int64_t Q, const float* bound, int64_t* idx, float* d2, double* out) {
    const NmGridView v = nm_host_view(((HostGridHandle*)p)->g);
    long long st[2] = {0, 0};
    for (int64_t i = 0; i < Q; ++i) {
        unsigned long long kk[8];
        const float b = bound[i] * 1.0001f + 1e-5f;
        nm_knn_search<8, true>(v, q[3 * i], q[3 * i + 1], q[3 * i + 2], kk, st, b * b);
        for (int k = 0; k < 8; ++k) {
            idx[i * 8 + k] = nm_key_idx(kk[k]) != 0x7fffffff ? nm_key_idx(kk[k]) : -1;
            d2[i * 8 + k] = nm_key_d2(kk[k]);
        }
    }
    out[0] = (double)st[0] / (double)(Q > 0 ? Q : 1);
    out[1] = (double)st[1] / (double)(Q > 0 ? Q : 1);
    return 0;
}

// average traversal cost per query: out[0] = node records tested, out[1] = vertices scanned
int hc_knn_stats(void* p, const float* q, int64_t Q, double* out) {
    const NmGridView v = nm_host_view(((HostGridHandle*)p)->g);
    long long st[2] = {0, 0};
    for (int64_t i = 0; i < Q; ++i) {
        unsigned long long kk[8];
        nm_knn_search<8, true>(v, q[3 * i], q[3 * i + 1], q[3 * i + 2], kk, st);
    }
    out[0] = (double)st[0] / (double)(Q > 0 ? Q : 1);
    out[1] = (double)st[1] / (double)(Q > 0 ? Q : 1);
    return 0;
}

// Host emulation of the wave-cooperative ("packet") traversal of nm_kernels.h: W queries share
// ONE traversal (a node is opened if ANY query needs it; every query tests every vertex of an
// opened leaf that it needs).  Checks exactness of the scheme and counts the shared work.
// out[0] = node records tested per packet, out[1] = vertices scanned per packet.
int hc_knn_packet(void* p, const float* q, int64_t Q, int Wd, int64_t* idx, float* d2, double* out) {
    const NmGridView g = nm_host_view(((HostGridHandle*)p)->g);
    long long nodes_t = 0, verts_t = 0, packets = 0, insert_events = 0, deferred_rounds = 0, lane_inserts = 0, marked_sum = 0;
    long long rejected_all = 0, scalar_sep = 0, scalar_box = 0;
    std::vector<unsigned long long> kk((size_t)Wd * 8);
    for (int64_t base = 0; base < Q; base += Wd) {
        const int n = (int)std::min<int64_t>(Wd, Q - base);
        ++packets;
        for (int l = 0; l < n * 8; ++l) kk[(size_t)l] = nm_key(NM_INF_F, 0x7fffffff);
        float cx0 = 0, cy0 = 0, cz0 = 0;
        for (int l = 0; l < n; ++l) { cx0 += q[3 * (base + l)]; cy0 += q[3 * (base + l) + 1]; cz0 += q[3 * (base + l) + 2]; }
        cx0 /= n; cy0 /= n; cz0 /= n;
        const uint32_t kx = nm_float_key(cx0), ky = nm_float_key(cy0), kz = nm_float_key(cz0);
        NmNode rec = g.nodes[0];
        int first = nm_octant(rec, kx, ky, kz);
        unsigned om = nm_visit_mask(rec, first);
        bool at_root = true;
        for (;;) {
            if (om == 0) {
                if (at_root) break;
                const int c_prev = (int)((rec.info >> 8) & 7u);
                const uint32_t parent = rec.parent;
                rec = g.nodes[parent];
                at_root = parent == 0u;
                first = nm_octant(rec, kx, ky, kz);
                om = nm_visit_mask(rec, first) & ~((2u << nm_perm(c_prev ^ first)) - 1u);
                continue;
            }
            const int i = __builtin_ctz(om);
            om &= om - 1;
            const int c = first ^ nm_perm(i);
            const uint32_t mask = rec.info & 255u;
            const NmNode crec = g.nodes[rec.first + (uint32_t)nm_popc(mask & ((1u << c) - 1u))];
            ++nodes_t;
            {   // out[6..8] (round 6 study): would a PACKET-level scalar pre-test have rejected this child without the 64-lane bound?
                // axis-separating form (integer compares only -- gfx950 has no scalar float ALU): the child's box misses the packet's AABB grown by
                // the largest K-th radius of the packet on some axis
                float lo[3] = {NM_INF_F, NM_INF_F, NM_INF_F}, hi[3] = {-NM_INF_F, -NM_INF_F, -NM_INF_F}, r2 = 0.f;
                for (int l = 0; l < n; ++l) {
                    const float* ql = q + 3 * (base + l);
                    for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], ql[a]); hi[a] = std::max(hi[a], ql[a]); }
                    r2 = std::max(r2, nm_key_d2(kk[(size_t)l * 8 + 7]));
                }
                const float rm = r2 < NM_INF_F ? std::sqrt(r2) * 1.00001f : NM_INF_F;
                const bool sep = crec.lox > hi[0] + rm || crec.hix < lo[0] - rm || crec.loy > hi[1] + rm || crec.hiy < lo[1] - rm ||
                                 crec.loz > hi[2] + rm || crec.hiz < lo[2] - rm;
                // full box-to-box lower bound at packet level (what a scalar FLOAT unit could do)
                const float ax = std::max(std::max(crec.lox - hi[0], lo[0] - crec.hix), 0.f), ay = std::max(std::max(crec.loy - hi[1], lo[1] - crec.hiy), 0.f),
                            az = std::max(std::max(crec.loz - hi[2], lo[2] - crec.hiz), 0.f);
                const bool boxbox = (ax * ax + ay * ay + az * az) * 0.99999f > r2;
                scalar_sep += sep ? 1 : 0;
                scalar_box += boxbox ? 1 : 0;
            }
            bool any = false;
            std::vector<char> want((size_t)n);
            for (int l = 0; l < n; ++l) {
                const float* ql = q + 3 * (base + l);
                want[(size_t)l] = nm_box_lb2(crec, ql[0], ql[1], ql[2]) <= nm_key_d2(kk[(size_t)l * 8 + 7]);
                any |= want[(size_t)l] != 0;
            }
            if (!any) { ++rejected_all; continue; }
            if ((crec.info & 255u) == 0u) {
                verts_t += crec.end - crec.first;
                // instruction-cost model of the leaf scan (out[2..4]): insert events of the one-vertex-at-a-time scan (a vertex costs the
                // wave one list insertion when ANY lane inserts it) against the deferred scan (per 64-vertex chunk every lane first marks the
                // vertices below its threshold at chunk entry, then the wave runs max-over-lanes(marked) insertion rounds)
                for (uint32_t p0 = crec.first; p0 < crec.end; p0 += 64) {
                    const uint32_t p1 = std::min<uint32_t>(crec.end, p0 + 64);
                    int max_marked = 0;
                    for (int l = 0; l < n; ++l) {
                        if (!want[(size_t)l]) continue;
                        const float* ql = q + 3 * (base + l);
                        const unsigned long long thr = kk[(size_t)l * 8 + 7];
                        int marked = 0;
                        for (uint32_t pp = p0; pp < p1; ++pp) {
                            const float4 v = g.sverts[pp];
                            marked += nm_key(nm_dist2(ql[0], ql[1], ql[2], v.x, v.y, v.z), nm_as_int(v.w)) < thr ? 1 : 0;
                        }
                        max_marked = std::max(max_marked, marked);
                        marked_sum += marked;
                    }
                    deferred_rounds += max_marked;
                }
                for (uint32_t pp = crec.first; pp < crec.end; ++pp) {
                    const float4 v = g.sverts[pp];
                    bool any_ins = false;
                    for (int l = 0; l < n; ++l) {
                        if (!want[(size_t)l]) continue;
                        const float* ql = q + 3 * (base + l);
                        const unsigned long long key = nm_key(nm_dist2(ql[0], ql[1], ql[2], v.x, v.y, v.z), nm_as_int(v.w));
                        unsigned long long(&k8)[8] = *reinterpret_cast<unsigned long long(*)[8]>(&kk[(size_t)l * 8]);
                        if (key < k8[7]) { nm_topk_insert<8>(k8, key); any_ins = true; ++lane_inserts; }
                    }
                    insert_events += any_ins ? 1 : 0;
                }
            } else {
                rec = crec;
                at_root = false;
                first = nm_octant(rec, kx, ky, kz);
                om = nm_visit_mask(rec, first);
            }
        }
        for (int l = 0; l < n; ++l)
            for (int k = 0; k < 8; ++k) {
                const unsigned long long key = kk[(size_t)l * 8 + k];
                const bool ok = nm_key_idx(key) != 0x7fffffff;
                idx[(base + l) * 8 + k] = ok ? nm_key_idx(key) : -1;
                d2[(base + l) * 8 + k] = ok ? nm_key_d2(key) : -1.0f;
            }
    }
    out[0] = (double)nodes_t / (double)(packets ? packets : 1);
    out[1] = (double)verts_t / (double)(packets ? packets : 1);
    out[2] = (double)insert_events / (double)(packets ? packets : 1);     // per packet: vertices at which at least one lane inserts
    out[3] = (double)deferred_rounds / (double)(packets ? packets : 1);   // per packet: insertion rounds of the deferred (chunk-wise) scan
    out[4] = (double)lane_inserts / (double)(Q ? Q : 1);                  // per query: list insertions
    out[5] = (double)marked_sum / (double)(Q ? Q : 1);                    // per query: vertices below the chunk-entry threshold
    out[6] = (double)rejected_all / (double)(packets ? packets : 1);      // per packet: node tests that NO lane passed
    out[7] = (double)scalar_sep / (double)(packets ? packets : 1);        // ... of which an axis-separating packet pre-test would have caught
    out[8] = (double)scalar_box / (double)(packets ? packets : 1);        // ... and a packet box-to-box lower bound
    return 0;
}

int hc_compute_distance(void* p, const float* q, int64_t Q, const float* indicator, float w1, float* ds,
                        int64_t* idx, float* w, float* grad) {
    auto* h = (HostGridHandle*)p;
    const NmGridView v = nm_host_view(h->g);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < Q; ++i) {
        float bd[8], wk[8], g[3];
        int bi[8];
        unsigned long long kk[8];
        nm_knn_search<8>(v, q[3 * i], q[3 * i + 1], q[3 * i + 2], kk);
        for (int k = 0; k < 8; ++k) {
            bd[k] = nm_key_d2(kk[k]);
            bi[k] = nm_key_idx(kk[k]);
        }
        ds[i] = nm_projected_distance8(q[3 * i], q[3 * i + 1], q[3 * i + 2], bd, bi, h->verts.data(), indicator, w1, wk, g);
        for (int k = 0; k < 8; ++k) {
            idx[i * 8 + k] = bi[k];
            w[i * 8 + k] = wk[k];
        }
        for (int a = 0; a < 3; ++a) grad[3 * i + a] = g[a];
    }
    return 0;
}

void hc_linspace01(int n, float* out) {
    for (int i = 0; i < n; ++i) out[i] = nm_linspace01(i, n);
}

void hc_ray_setup(const float* o, const float* d, int64_t R, float radius, float* dirn, float* nf) {
    for (int64_t r = 0; r < R; ++r) nm_ray_setup(o + 3 * r, d + 3 * r, radius, dirn + 3 * r, nf + 2 * r, nf + 2 * r + 1);
}

void hc_ray_bounds(const float* ds_probe, int64_t R, int G, float thresh, const float* nf0, float* nf) {
    for (int64_t r = 0; r < R; ++r)
        nm_ray_bounds(ds_probe + r * G, 1, G, thresh, nf0[2 * r], nf0[2 * r + 1], nf + 2 * r, nf + 2 * r + 1);
}

// d, sdf: [R, cap] with the first n valid; writes d[:, n:n+n_new]
// sample_pdf(det=False): the caller's uniform randoms u [R][n_new]
void hc_ray_upsample_u(float* d, const float* sdf, int64_t R, int cap, int n, int it, int n_new, const float* u) {
    float w[NM_MAX_SAMPLES], cdf[NM_MAX_SAMPLES];
    for (int64_t r = 0; r < R; ++r)
        nm_ray_upsample<int>(d + r * cap, sdf + r * cap, n, it, n_new, d + r * cap + n, w, cdf, nullptr, nullptr, nullptr, u + r * n_new);
}

void hc_ray_upsample(float* d, const float* sdf, int64_t R, int cap, int n, int it, int n_new) {
    for (int64_t r = 0; r < R; ++r) {
        float w[NM_MAX_SAMPLES], cdf[NM_MAX_SAMPLES];
        nm_ray_upsample(d + r * cap, sdf + r * cap, n, it, n_new, d + r * cap + n, w, cdf);
    }
}

void hc_ray_merge(float* d, float* sdf, int64_t R, int cap, int n, int m) {
    for (int64_t r = 0; r < R; ++r) nm_ray_merge(d + r * cap, sdf + r * cap, n, m);
}

// up-sampling with slot tracking and warm-start bounds, as nm_rays_upsample_kernel does
void hc_ray_upsample_slots(float* d, float* sdf, int* slot, const float* radius, float* bound, int64_t R, int cap, int n,
                           int m, int it, int n_new) {
    for (int64_t r = 0; r < R; ++r) {
        float w[NM_MAX_SAMPLES], cdf[NM_MAX_SAMPLES];
        int* sl = slot + r * cap;
        if (m > 0) nm_ray_merge(d + r * cap, sdf + r * cap, n - m, m, sl);
        else for (int j = 0; j < n; ++j) sl[j] = j;
        nm_ray_upsample(d + r * cap, sdf + r * cap, n, it, n_new, d + r * cap + n, w, cdf, sl, radius + r * cap, bound + r * cap + n);
    }
}

void hc_ray_composite(const float* sdf, const float* d, int64_t R, int N, float s, const float* rgb_mid,
                      const float* nablas, int white, float* rgb, float* depth, float* acc, float* normals) {
    for (int64_t r = 0; r < R; ++r) {
        float w[NM_MAX_SAMPLES];
        nm_ray_composite(sdf + r * N, d + r * N, N, s, rgb_mid + r * (N - 1) * 3, nablas ? nablas + r * N * 3 : nullptr,
                         white, rgb + 3 * r, depth + r, acc + r, normals ? normals + 3 * r : nullptr, w);
    }
}
}
