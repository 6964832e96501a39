/*
 * oracle/knn_ref.c -- TEST INFRASTRUCTURE ONLY (the parity oracle, never the product path).
 *
 * CPU restatement of the K-nearest-vertex search that NeuMesh obtains from the
 * external FRNN CUDA package:
 *     reference call sites: models/mesh_grid.py:64-74 (grid build, result discarded)
 *                           models/mesh_grid.py:109-119 (query: K=8, r=100.0,
 *                           return_sorted=True -> squared distances ascending + indices)
 * FRNN itself (github.com/lxxue/FRNN) is not vendored in /root/reference and is not
 * pinned to any version, so the arithmetic is DECLARED here:
 *     dx = q.x - v.x (IEEE fp32), d2 = (dx*dx + dy*dy) + dz*dz, no FMA contraction,
 *     the K smallest by (d2, vertex index) ascending; with r = 100 and a scene inside
 *     the unit sphere the radius never excludes anything, so this is exact K-NN.
 * "parity unpinned": the reference ships no golden vectors for this boundary.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC (see Makefile).
 */
#include <stdint.h>
#include <stddef.h>

#define NM_ORACLE_MAX_K 64

static inline int lex_less(float da, int64_t ia, float db, int64_t ib) {
    return (da < db) || (da == db && ia < ib);
}

/* q: [Q,3], v: [V,3]; out idx: [Q,K] int64, d2: [Q,K] f32. Slots beyond V (V<K) get
 * idx=-1, d2=-1 (FRNN pads with -1; out of contract for NeuMesh, V >= K always). */
int nm_oracle_knn(const float* q, int64_t Q, const float* v, int64_t V, int K,
                  int64_t* idx_out, float* d2_out) {
    if (K < 1 || K > NM_ORACLE_MAX_K || Q < 0 || V < 0) return 1;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < Q; ++i) {
        float bd[NM_ORACLE_MAX_K];
        int64_t bi[NM_ORACLE_MAX_K];
        int n = 0;
        const float qx = q[3 * i + 0], qy = q[3 * i + 1], qz = q[3 * i + 2];
        for (int64_t j = 0; j < V; ++j) {
            const float dx = qx - v[3 * j + 0];
            const float dy = qy - v[3 * j + 1];
            const float dz = qz - v[3 * j + 2];
            const float xx = dx * dx;
            const float yy = dy * dy;
            const float zz = dz * dz;
            const float s = xx + yy;
            const float d = s + zz;
            if (n == K && !lex_less(d, j, bd[K - 1], bi[K - 1])) continue;
            int p = (n < K) ? n : K - 1;
            while (p > 0 && lex_less(d, j, bd[p - 1], bi[p - 1])) {
                bd[p] = bd[p - 1];
                bi[p] = bi[p - 1];
                --p;
            }
            bd[p] = d;
            bi[p] = j;
            if (n < K) ++n;
        }
        for (int k = 0; k < K; ++k) {
            idx_out[i * K + k] = (k < n) ? bi[k] : -1;
            d2_out[i * K + k] = (k < n) ? bd[k] : -1.0f;
        }
    }
    return 0;
}

/* Re-rank a candidate list with the pinned arithmetic: cand: [Q,C] int64 vertex ids
 * (e.g. from a float64 kd-tree), select the K best by (d2, index). Used only to give the
 * CPU timing baseline an O(log V) search; validated against nm_oracle_knn in tests. */
int nm_oracle_rerank(const float* q, int64_t Q, const float* v, const int64_t* cand, int C,
                     int K, int64_t* idx_out, float* d2_out) {
    if (K < 1 || K > NM_ORACLE_MAX_K || C < K) return 1;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < Q; ++i) {
        float bd[NM_ORACLE_MAX_K];
        int64_t bi[NM_ORACLE_MAX_K];
        int n = 0;
        const float qx = q[3 * i + 0], qy = q[3 * i + 1], qz = q[3 * i + 2];
        for (int c = 0; c < C; ++c) {
            const int64_t j = cand[i * C + c];
            const float dx = qx - v[3 * j + 0];
            const float dy = qy - v[3 * j + 1];
            const float dz = qz - v[3 * j + 2];
            const float xx = dx * dx;
            const float yy = dy * dy;
            const float zz = dz * dz;
            const float s = xx + yy;
            const float d = s + zz;
            if (n == K && !lex_less(d, j, bd[K - 1], bi[K - 1])) continue;
            int p = (n < K) ? n : K - 1;
            while (p > 0 && lex_less(d, j, bd[p - 1], bi[p - 1])) {
                bd[p] = bd[p - 1];
                bi[p] = bi[p - 1];
                --p;
            }
            bd[p] = d;
            bi[p] = j;
            if (n < K) ++n;
        }
        for (int k = 0; k < K; ++k) {
            idx_out[i * K + k] = bi[k];
            d2_out[i * K + k] = bd[k];
        }
    }
    return 0;
}
