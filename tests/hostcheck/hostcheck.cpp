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
        float bd[K];
        int bi[K];
        nm_knn_search<K>(v, q[3 * i], q[3 * i + 1], q[3 * i + 2], bd, bi);
        for (int k = 0; k < K; ++k) {
            const bool ok = bi[k] != 0x7fffffff;
            idx[i * K + k] = ok ? bi[k] : -1;
            d2[i * K + k] = ok ? bd[k] : -1.0f;
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

int hc_compute_distance(void* p, const float* q, int64_t Q, const float* indicator, float w1, float* ds,
                        int64_t* idx, float* w, float* grad) {
    auto* h = (HostGridHandle*)p;
    const NmGridView v = nm_host_view(h->g);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < Q; ++i) {
        float bd[8], wk[8], g[3];
        int bi[8];
        nm_knn_search<8>(v, q[3 * i], q[3 * i + 1], q[3 * i + 2], bd, bi);
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
void hc_ray_upsample(float* d, const float* sdf, int64_t R, int cap, int n, int it, int n_new) {
    for (int64_t r = 0; r < R; ++r) {
        float w[NM_MAX_SAMPLES], cdf[NM_MAX_SAMPLES];
        nm_ray_upsample(d + r * cap, sdf + r * cap, n, it, n_new, d + r * cap + n, w, cdf);
    }
}

void hc_ray_merge(float* d, float* sdf, int64_t R, int cap, int n, int m) {
    for (int64_t r = 0; r < R; ++r) nm_ray_merge(d + r * cap, sdf + r * cap, n, m);
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
