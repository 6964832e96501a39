// nm_train.h -- device + launch code of the TRAINING form of the field: forward with every
// intermediate kept, and the closed-form backward pass -- no autograd graph, no torch-op recomputation.
//
// Reference semantics (file:line in the NeuMesh tree):
//   NeuMesh.forward / forward_with_nablas            models/frameworks/neumesh/neumesh.py:113-153
//   _forward_density (+ nabla by autograd.grad)      neumesh.py:204-237
//   _forward_color                                   neumesh.py:239-260
//   MeshGrid.compute_distance_frnn (weights detached) models/mesh_grid.py:88-144
//   what the trainer differentiates                  models/trainer.py:75-81,186-209 (image, eikonal, mask, indicator terms)
//
// nabla = d sdf / d xyz reaches xyz only through ds (the K-NN weights are detached, mesh_grid.py:120-122), so
//   nabla = alpha * g,   alpha = d sdf / d ds (forward-mode tangent through the geometry MLP),   g = d ds / d xyz (closed form).
// The geometry MLP therefore runs on a PAIR (h, t = dh/dds) per point:
//   z = W h + b,  u = W t,  h' = softplus(z),  t' = softplus'(z) * u
// and its reverse pass, for cotangents (H, T) on (h', t'), is
//   Z = H * softplus'(z) + T * softplus''(z) * u,   U = T * softplus'(z),
//   dW += Z^T h + U^T t,  db += sum Z,   cotangent of h = Z W,  of t = U W.
// A cotangent on nabla (eikonal loss, normals, the colour MLP's nabla input) enters as T on the last layer -- the
// "second derivative" of the reference's create_graph=True pass is this reverse pass, first order in everything.
//
// Layout: value and tangent rows are STACKED: every [rows, W] activation array holds the P value rows first and the
// P tangent rows after them, so each layer is ONE product with M = 2P in each direction (nm_gemm.h), and the weight
// gradient's two terms are one reduction over 2P rows.  All arrays fp32; the GEMMs run on the fp32 matrix pipe.
// Gradients are ACCUMULATED into the caller's buffers (atomic adds: tables are scattered through the neighbour
// lists, weight gradients are split over the points), so summation order is not fixed from run to run.
#pragma once

#include <hip/hip_runtime.h>

#include "nm_gemm.h"

#define NM_T_BETA 100.0f
#define NM_T_THRESH 20.0f

struct NmTrainDims {
    int W, Dg, Dc, G, Cd;              // hidden width, geometry / colour depth, code widths
    int md, mfg, mft, mv;              // embedder bands (>= 0)
    int ch_d, ch_v;                    // 1 + 2 md, 3 + 6 mv
    int K0, K0p;                       // geometry input: ch_d + G (1 + 2 mfg), padded to 16
    int Kt;                            // tangent input width: ch_d padded to 16
    int Kc0, Kc0p;                     // colour input: [nabla 3] + ch_d + ch_v + Cd (1 + 2 mft), padded to 16
    int use_nabla;
    int off_d, off_v, off_ft;          // column offsets inside the colour input
};

static inline int nm_t_pad16(int k) { return (k + 15) & ~15; }

static inline NmTrainDims nm_train_dims(const nm_field_desc* d) {
    NmTrainDims t;
    t.W = d->W; t.Dg = d->D_density; t.Dc = d->D_color; t.G = d->geometry_dim; t.Cd = d->color_dim;
    t.md = d->multires_d > 0 ? d->multires_d : 0; t.mfg = d->multires_fg > 0 ? d->multires_fg : 0;
    t.mft = d->multires_ft > 0 ? d->multires_ft : 0; t.mv = d->multires_view > 0 ? d->multires_view : 0;
    t.ch_d = 1 + 2 * t.md; t.ch_v = 3 + 6 * t.mv;
    t.K0 = t.ch_d + t.G * (1 + 2 * t.mfg); t.K0p = nm_t_pad16(t.K0);
    t.Kt = nm_t_pad16(t.ch_d);
    t.use_nabla = d->enable_nablas_input ? 1 : 0;
    t.off_d = t.use_nabla ? 3 : 0; t.off_v = t.off_d + t.ch_d; t.off_ft = t.off_v + t.ch_v;
    t.Kc0 = t.off_ft + t.Cd * (1 + 2 * t.mft); t.Kc0p = nm_t_pad16(t.Kc0);
    return t;
}

// ------------------------------------------------------------------------------------------------ workspace
struct NmTrainWs {
    long long P;
    int *idx;                                    // [P,8]
    float *w, *ds, *gds, *xyz, *fg, *ft;         // [P,8] [P] [P,3] [P,3] [P,G] [P,Cd]
    float *X0, *T0;                              // [P,K0p] [P,Kt]
    float *ZU[8], *HT[8];                        // [2P,W] per geometry layer: pre-activations (z | u), activations (h | t)
    float *sdf, *alpha, *nabla;                  // [P] [P] [P,3]
    float *C0, *HC[8], *rgb;                     // [P,Kc0p], [P,W] per colour layer (post-ReLU), [P,3]
    // backward temporaries
    float *DA, *DB;                              // [2P,W] x 2 (ping-pong cotangents)
    float *DX0, *DT0, *DC0;                      // [P,K0p] [P,Kt] [P,Kc0p]
    float *dds, *dnab, *gvec;                    // [P] [P,3] [P,3]
    float *W0p, *Wc0p, *dW0p, *dWc0p;            // zero-padded first-layer weights [W,K0p] [W,Kc0p] and their gradients
    size_t bytes;
};

static inline NmTrainWs nm_train_carve(void* base, long long P, const NmTrainDims& t) {
    NmTrainWs s;
    s.P = P;
    size_t off = 0;
    auto take = [&](size_t n_floats) {
        float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
        off += ((n_floats * 4 + 255) / 256) * 256;
        return p;
    };
    const size_t p = (size_t)P, W = (size_t)t.W;
    s.idx = reinterpret_cast<int*>(take(p * 8));
    s.w = take(p * 8); s.ds = take(p); s.gds = take(p * 3); s.xyz = take(p * 3); s.fg = take(p * t.G); s.ft = take(p * t.Cd);
    s.X0 = take(p * t.K0p); s.T0 = take(p * t.Kt);
    for (int l = 0; l < 8; ++l) { s.ZU[l] = l < t.Dg ? take(2 * p * W) : nullptr; s.HT[l] = l < t.Dg ? take(2 * p * W) : nullptr; }
    s.sdf = take(p); s.alpha = take(p); s.nabla = take(p * 3);
    s.C0 = take(p * t.Kc0p);
    for (int l = 0; l < 8; ++l) s.HC[l] = l < t.Dc ? take(p * W) : nullptr;
    s.rgb = take(p * 3);
    s.DA = take(2 * p * W); s.DB = take(2 * p * W);
    s.DX0 = take(p * t.K0p); s.DT0 = take(p * t.Kt); s.DC0 = take(p * t.Kc0p);
    s.dds = take(p); s.dnab = take(p * 3); s.gvec = take(p * 3);
    s.W0p = take(W * t.K0p); s.Wc0p = take(W * t.Kc0p); s.dW0p = take(W * t.K0p); s.dWc0p = take(W * t.Kc0p);
    s.bytes = off;
    return s;
}

// ------------------------------------------------------------------------------------------------ small helpers
__device__ __forceinline__ float nm_t_softplus(float z) {
    const float bz = NM_T_BETA * z;
    return bz > NM_T_THRESH ? z : log1pf(expf(bz)) * (1.0f / NM_T_BETA);
}
// softplus'(z) = sigmoid(beta z) (1 above the threshold), softplus''(z) = beta s (1 - s) (0 above it): torch's softplus_backward
__device__ __forceinline__ void nm_t_softplus_d(float z, float& s1, float& s2) {
    const float bz = NM_T_BETA * z;
    if (bz > NM_T_THRESH) { s1 = 1.f; s2 = 0.f; return; }
    const float e = expf(bz);
    s1 = e / (e + 1.0f);
    s2 = NM_T_BETA * e / ((e + 1.0f) * (e + 1.0f));
}

// copy [rows, cols] -> [rows, ld] zero-padded (dir = 0), or add the [rows, cols] corner of a padded array into dst (dir = 1)
__global__ void nm_t_pad_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols, int ld, int dir) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)rows * ld) return;
    const int r = (int)(i / ld), c = (int)(i % ld);
    if (dir == 0) dst[i] = c < cols ? src[(long long)r * cols + c] : 0.f;
    else if (c < cols) dst[(long long)r * cols + c] += src[i];
}

// ------------------------------------------------------------------------------------------------ forward
// Embedder.forward (models/base.py:52-70): [x, sin(x f0), cos(x f0), sin(x f1), ...], f_i = 2^i.
// One workgroup = 8 points x 32 lanes.  Geometry input X0 = [emb(ds) | emb(fg)], tangent input T0 = d emb(ds) / d ds,
// colour input C0 = [nabla | emb(ds) | emb(view) | emb(ft)] (the nabla columns are written by the geometry head).
__global__ __launch_bounds__(256) void nm_t_embed_kernel(NmTrainDims t, long long P, const float* __restrict__ ds, const float* __restrict__ fg,
                                                         const float* __restrict__ ft, const float* __restrict__ view,
                                                         float* __restrict__ X0, float* __restrict__ T0, float* __restrict__ C0,
                                                         const float* __restrict__ xyz, float* __restrict__ xyz_keep) {
    const long long p = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int j = threadIdx.x & 31;
    if (p >= P) return;
    if (j < 3) xyz_keep[p * 3 + j] = xyz[p * 3 + j];      // the backward pass of the distance stage needs the positions again
    const float d = ds[p];
    float* x0 = X0 + p * t.K0p;
    float* t0 = T0 + p * t.Kt;
    float* c0 = C0 ? C0 + p * t.Kc0p : nullptr;
    for (int c = j; c < t.Kt; c += 32) {          // ds embedding and its derivative
        float e = 0.f, de = 0.f;
        if (c == 0) { e = d; de = 1.f; }
        else if (c < t.ch_d) {
            const int b = (c - 1) >> 1;
            const float f = (float)(1 << b), a = d * f;
            if ((c - 1) & 1) { e = cosf(a); de = -f * sinf(a); }
            else { e = sinf(a); de = f * cosf(a); }
        }
        t0[c] = de;
        if (c < t.ch_d) {
            x0[c] = e;
            if (c0) c0[t.off_d + c] = e;
        }
    }
    for (int q = j; q < t.G; q += 32) {           // geometry code embedding
        const float x = fg[p * t.G + q];
        x0[t.ch_d + q] = x;
        for (int b = 0; b < t.mfg; ++b) {
            const float a = x * (float)(1 << b);
            x0[t.ch_d + (1 + 2 * b) * t.G + q] = sinf(a);
            x0[t.ch_d + (2 + 2 * b) * t.G + q] = cosf(a);
        }
    }
    for (int c = t.K0 + j; c < t.K0p; c += 32) x0[c] = 0.f;
    if (!c0) return;
    for (int c = j; c < t.ch_v; c += 32) {        // view direction embedding
        float e;
        if (c < 3) e = view[p * 3 + c];
        else {
            const int b = (c - 3) / 6, r = (c - 3) % 6;
            const float a = view[p * 3 + r % 3] * (float)(1 << b);
            e = r < 3 ? sinf(a) : cosf(a);
        }
        c0[t.off_v + c] = e;
    }
    for (int q = j; q < t.Cd; q += 32) {          // colour code embedding
        const float x = ft[p * t.Cd + q];
        c0[t.off_ft + q] = x;
        for (int b = 0; b < t.mft; ++b) {
            const float a = x * (float)(1 << b);
            c0[t.off_ft + (1 + 2 * b) * t.Cd + q] = sinf(a);
            c0[t.off_ft + (2 + 2 * b) * t.Cd + q] = cosf(a);
        }
    }
    for (int c = t.Kc0 + j; c < t.Kc0p; c += 32) c0[c] = 0.f;
}

// (z | u) -> (h | t) = (softplus(z) | softplus'(z) u);  n = P * W elements, tangent rows start at `toff` floats
__global__ void nm_t_softplus_kernel(const float* __restrict__ ZU, float* __restrict__ HT, long long n, long long toff, int tangent) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float4 z = *reinterpret_cast<const float4*>(ZU + i);
    float4 h, s;
    float s2;
    h.x = nm_t_softplus(z.x); h.y = nm_t_softplus(z.y); h.z = nm_t_softplus(z.z); h.w = nm_t_softplus(z.w);
    *reinterpret_cast<float4*>(HT + i) = h;
    if (!tangent) return;
    nm_t_softplus_d(z.x, s.x, s2); nm_t_softplus_d(z.y, s.y, s2); nm_t_softplus_d(z.z, s.z, s2); nm_t_softplus_d(z.w, s.w, s2);
    const float4 u = *reinterpret_cast<const float4*>(ZU + toff + i);
    *reinterpret_cast<float4*>(HT + toff + i) = make_float4(s.x * u.x, s.y * u.y, s.z * u.z, s.w * u.w);
}

__device__ __forceinline__ float nm_t_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// density head (neumesh.py:217): sdf = h . wd + bd, alpha = t . wd, nabla = alpha * g.  One wave per point.
__global__ __launch_bounds__(256) void nm_t_geo_head_kernel(NmTrainDims t, long long P, const float* __restrict__ HT, const float* __restrict__ wd,
                                                            const float* __restrict__ bd, const float* __restrict__ gds, int tangent,
                                                            float* __restrict__ sdf, float* __restrict__ alpha, float* __restrict__ nabla,
                                                            float* __restrict__ C0, float* __restrict__ sdf_out, float* __restrict__ nabla_out) {
    const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= P) return;
    float s = 0.f, a = 0.f;
    for (int c = lane; c < t.W; c += 64) {
        const float wv = wd[c];
        s += HT[p * t.W + c] * wv;
        if (tangent) a += HT[(P + p) * t.W + c] * wv;
    }
    s = nm_t_wave_sum(s);
    a = nm_t_wave_sum(a);
    if (lane == 0) {
        sdf[p] = s + bd[0];
        sdf_out[p] = s + bd[0];
        if (tangent) alpha[p] = a;
    }
    if (tangent && lane < 3) {
        const float nv = a * gds[p * 3 + lane];
        nabla[p * 3 + lane] = nv;
        if (nabla_out) nabla_out[p * 3 + lane] = nv;
        if (C0 && t.use_nabla) C0[p * t.Kc0p + lane] = nv;
    }
}

// colour head (neumesh.py:103,259): rgb = sigmoid(h Wr^T + br)
__global__ __launch_bounds__(256) void nm_t_col_head_kernel(NmTrainDims t, long long P, const float* __restrict__ HC, const float* __restrict__ Wr,
                                                            const float* __restrict__ br, float* __restrict__ rgb, float* __restrict__ rgb_out) {
    const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= P) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int c = lane; c < t.W; c += 64) {
        const float h = HC[p * t.W + c];
        a0 += h * Wr[c]; a1 += h * Wr[t.W + c]; a2 += h * Wr[2 * t.W + c];
    }
    a0 = nm_t_wave_sum(a0); a1 = nm_t_wave_sum(a1); a2 = nm_t_wave_sum(a2);
    if (lane < 3) {
        const float z = (lane == 0 ? a0 : lane == 1 ? a1 : a2) + br[lane];
        const float c = 1.0f / (1.0f + expf(-z));
        rgb[p * 3 + lane] = c;
        rgb_out[p * 3 + lane] = c;
    }
}

// ------------------------------------------------------------------------------------------------ backward
// colour head: zr = g_rgb * rgb (1 - rgb);  cotangent of the last hidden layer (ReLU mask applied) -> DZ;  dWr, dbr accumulated.
// A wave walks a strip of points and keeps its share of dWr in registers (W <= 256: 4 columns per lane).
__global__ __launch_bounds__(256) void nm_t_col_head_bwd_kernel(NmTrainDims t, long long P, const float* __restrict__ g_rgb, const float* __restrict__ rgb,
                                                                const float* __restrict__ HC, const float* __restrict__ Wr,
                                                                float* __restrict__ DZ, float* __restrict__ dWr, float* __restrict__ dbr) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), waves = (long long)gridDim.x * 4;
    for (int c0 = 0; c0 < t.W; c0 += 256) {
        float acc[4][3], wr[4][3];
        float db[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = c0 + lane + 64 * q;
#pragma unroll
            for (int j = 0; j < 3; ++j) { acc[q][j] = 0.f; wr[q][j] = c < t.W ? Wr[j * t.W + c] : 0.f; }
        }
        for (long long p = wave; p < P; p += waves) {
            float z[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float r = rgb[p * 3 + j];
                z[j] = g_rgb ? g_rgb[p * 3 + j] * r * (1.0f - r) : 0.f;
                db[j] += z[j];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = c0 + lane + 64 * q;
                if (c >= t.W) continue;
                const float h = HC[p * t.W + c];
                DZ[p * t.W + c] = h > 0.f ? z[0] * wr[q][0] + z[1] * wr[q][1] + z[2] * wr[q][2] : 0.f;
                acc[q][0] += z[0] * h; acc[q][1] += z[1] * h; acc[q][2] += z[2] * h;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = c0 + lane + 64 * q;
            if (c >= t.W) continue;
#pragma unroll
            for (int j = 0; j < 3; ++j) atomicAdd(dWr + j * t.W + c, acc[q][j]);
        }
        if (c0 == 0 && lane < 3) atomicAdd(dbr + lane, lane == 0 ? db[0] : lane == 1 ? db[1] : db[2]);
    }
}

// column sums of D[rows, W] -> out[W] (bias gradients)
#define NM_T_COLSUM_ROWS 64
__global__ __launch_bounds__(256) void nm_t_colsum_kernel(const float* __restrict__ D, long long rows, int W, float* __restrict__ out) {
    const long long r0 = (long long)blockIdx.x * NM_T_COLSUM_ROWS, r1 = r0 + NM_T_COLSUM_ROWS < rows ? r0 + NM_T_COLSUM_ROWS : rows;
    for (int c = threadIdx.x; c < W; c += 256) {
        float s = 0.f;
        for (long long r = r0; r < r1; ++r) s += D[r * W + c];
        atomicAdd(out + c, s);
    }
}

// cotangent of an embedded code vector -> cotangent of the code (32 lanes = code dims, up to 64 dims in two rounds), scattered into
// the vertex table through the point's neighbour list (interpolation, neumesh.py:11-13: code = sum_k w_k table[idx_k]).
__device__ __forceinline__ void nm_t_code_bwd(const float* __restrict__ dE, const float* __restrict__ code, int dim, int bands,
                                              const int* __restrict__ idx, const float* __restrict__ w, float* __restrict__ dTable, int j) {
    for (int q = j; q < dim; q += 32) {
        const float x = code[q];
        float g = dE[q];
        for (int b = 0; b < bands; ++b) {
            const float f = (float)(1 << b), a = x * f;
            g += dE[(1 + 2 * b) * dim + q] * f * cosf(a) - dE[(2 + 2 * b) * dim + q] * f * sinf(a);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(dTable + (long long)idx[k] * dim + q, w[k] * g);
    }
}

// cotangent of emb(ds) (and of its derivative, for the tangent input) -> cotangent of ds; lanes 0..31 of one point
__device__ __forceinline__ float nm_t_ds_emb_bwd(const float* __restrict__ dE, const float* __restrict__ dT, int ch_d, float d, int j) {
    float g = 0.f;
    for (int c = j; c < ch_d; c += 32) {
        if (c == 0) { g += dE[0]; continue; }          // emb = ds, derivative 1, second derivative 0
        const int b = (c - 1) >> 1;
        const float f = (float)(1 << b), a = d * f;
        const float sn = sinf(a), cs = cosf(a);
        if ((c - 1) & 1) { g += dE[c] * (-f * sn); if (dT) g += dT[c] * (-f * f * cs); }    // cos band
        else { g += dE[c] * (f * cs); if (dT) g += dT[c] * (-f * f * sn); }                 // sin band
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) g += __shfl_xor(g, o);
    return g;
}

// colour input cotangent DC0 [P,Kc0p] -> dnab += nabla columns, dds += emb(ds) columns, colour table scatter
__global__ __launch_bounds__(256) void nm_t_col_input_bwd_kernel(NmTrainDims t, long long P, const float* __restrict__ DC0, const float* __restrict__ ds,
                                                                 const float* __restrict__ ft, const int* __restrict__ idx, const float* __restrict__ w,
                                                                 float* __restrict__ dnab, float* __restrict__ dds, float* __restrict__ dFt) {
    const long long p = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int j = threadIdx.x & 31;
    if (p >= P) return;
    const float* dc = DC0 + p * t.Kc0p;
    if (t.use_nabla && j < 3) dnab[p * 3 + j] += dc[j];
    const float g = nm_t_ds_emb_bwd(dc + t.off_d, nullptr, t.ch_d, ds[p], j);
    if (j == 0) dds[p] += g;
    if (dFt) nm_t_code_bwd(dc + t.off_ft, ft + p * t.Cd, t.Cd, t.mft, idx + p * 8, w + p * 8, dFt, j);
}

// geometry input cotangents DX0 [P,K0p], DT0 [P,Kt] -> dds +=, geometry table scatter
__global__ __launch_bounds__(256) void nm_t_geo_input_bwd_kernel(NmTrainDims t, long long P, const float* __restrict__ DX0, const float* __restrict__ DT0,
                                                                 const float* __restrict__ ds, const float* __restrict__ fg, const int* __restrict__ idx,
                                                                 const float* __restrict__ w, float* __restrict__ dds, float* __restrict__ dFg) {
    const long long p = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int j = threadIdx.x & 31;
    if (p >= P) return;
    const float g = nm_t_ds_emb_bwd(DX0 + p * t.K0p, DT0 ? DT0 + p * t.Kt : nullptr, t.ch_d, ds[p], j);
    if (j == 0) dds[p] += g;
    if (dFg) nm_t_code_bwd(DX0 + p * t.K0p + t.ch_d, fg + p * t.G, t.G, t.mfg, idx + p * 8, w + p * 8, dFg, j);
}

// density head backward fused with the last layer's activation backward.  Per point: S = g_sdf, N = g_nabla (+ the colour
// input's nabla cotangent, already in dnab);  A = N . g (cotangent of alpha), gvec = alpha N (cotangent of g = d ds / d xyz);
// cotangents of the last (h, t): H = S wd, T = A wd  ->  (Z | U) of that layer.  dwd, dbd accumulated (strip per wave).
__global__ __launch_bounds__(256) void nm_t_geo_head_bwd_kernel(NmTrainDims t, long long P, const float* __restrict__ g_sdf, const float* __restrict__ g_nabla,
                                                                const float* __restrict__ dnab, const float* __restrict__ gds, const float* __restrict__ alpha,
                                                                const float* __restrict__ wd, const float* __restrict__ ZU, const float* __restrict__ HT,
                                                                int tangent, float* __restrict__ DZU, float* __restrict__ gvec,
                                                                float* __restrict__ dwd, float* __restrict__ dbd) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), waves = (long long)gridDim.x * 4;
    for (int c0 = 0; c0 < t.W; c0 += 256) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f}, wv[4];
        float db = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) wv[q] = c0 + lane + 64 * q < t.W ? wd[c0 + lane + 64 * q] : 0.f;
        for (long long p = wave; p < P; p += waves) {
            const float S = g_sdf ? g_sdf[p] : 0.f;
            float A = 0.f;
            if (tangent) {
                float n[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    n[k] = (g_nabla ? g_nabla[p * 3 + k] : 0.f) + (dnab ? dnab[p * 3 + k] : 0.f);
                    A += n[k] * gds[p * 3 + k];
                }
                if (c0 == 0 && lane < 3) gvec[p * 3 + lane] = alpha[p] * (lane == 0 ? n[0] : lane == 1 ? n[1] : n[2]);
            }
            db += S;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = c0 + lane + 64 * q;
                if (c >= t.W) continue;
                const float H = S * wv[q], T = A * wv[q];
                float s1, s2;
                nm_t_softplus_d(ZU[p * t.W + c], s1, s2);
                acc[q] += S * HT[p * t.W + c];
                if (tangent) {
                    acc[q] += A * HT[(P + p) * t.W + c];
                    DZU[p * t.W + c] = H * s1 + T * s2 * ZU[(P + p) * t.W + c];
                    DZU[(P + p) * t.W + c] = T * s1;
                } else {
                    DZU[p * t.W + c] = H * s1;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (c0 + lane + 64 * q < t.W) atomicAdd(dwd + c0 + lane + 64 * q, acc[q]);
        if (c0 == 0 && lane == 0) atomicAdd(dbd, db);
    }
}

// (H | T) cotangents of a layer's activations -> (Z | U) cotangents of its pre-activations (in place allowed)
__global__ void nm_t_softplus_bwd_kernel(const float* __restrict__ DHT, const float* __restrict__ ZU, float* __restrict__ DZU, long long n, long long toff,
                                         int tangent) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float4 z = *reinterpret_cast<const float4*>(ZU + i);
    const float4 H = *reinterpret_cast<const float4*>(DHT + i);
    float s1[4], s2[4];
    nm_t_softplus_d(z.x, s1[0], s2[0]); nm_t_softplus_d(z.y, s1[1], s2[1]); nm_t_softplus_d(z.z, s1[2], s2[2]); nm_t_softplus_d(z.w, s1[3], s2[3]);
    if (!tangent) {
        *reinterpret_cast<float4*>(DZU + i) = make_float4(H.x * s1[0], H.y * s1[1], H.z * s1[2], H.w * s1[3]);
        return;
    }
    const float4 T = *reinterpret_cast<const float4*>(DHT + toff + i);
    const float4 u = *reinterpret_cast<const float4*>(ZU + toff + i);
    *reinterpret_cast<float4*>(DZU + i) = make_float4(H.x * s1[0] + T.x * s2[0] * u.x, H.y * s1[1] + T.y * s2[1] * u.y,
                                                      H.z * s1[2] + T.z * s2[2] * u.z, H.w * s1[3] + T.w * s2[3] * u.w);
    *reinterpret_cast<float4*>(DZU + toff + i) = make_float4(T.x * s1[0], T.y * s1[1], T.z * s1[2], T.w * s1[3]);
}

// projected distance backward (mesh_grid.py:125-142; weights and neighbours detached): with d = x - v_k, r = |d|, a = d . n_k, D = w1 + r,
//   ds = sum_k w_k f_k,  f = (w1 a + r^3) / D,   g = d ds / d x = sum_k w_k [ (w1 n + 3 r^2 u) D - (w1 a + r^3) u ] / D^2,  u = d / r.
// Cotangents dds (of ds) and gvec (of g) -> indicator vectors (scatter) and the indicator weight w1.  One thread per point.
__global__ __launch_bounds__(256) void nm_t_distance_bwd_kernel(long long P, const float* __restrict__ xyz, const int* __restrict__ idx, const float* __restrict__ w,
                                                                const float* __restrict__ verts, const float* __restrict__ indicator, float w1,
                                                                const float* __restrict__ dds, const float* __restrict__ gvec,
                                                                float* __restrict__ dInd, float* __restrict__ dw1) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float gw1 = 0.f;
    if (p < P) {
        const float x = xyz[p * 3], y = xyz[p * 3 + 1], z = xyz[p * 3 + 2];
        const float S = dds[p];
        const float gx = gvec ? gvec[p * 3] : 0.f, gy = gvec ? gvec[p * 3 + 1] : 0.f, gz = gvec ? gvec[p * 3 + 2] : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const long long i = idx[p * 8 + k];
            const float wk = w[p * 8 + k];
            const float dx = x - verts[3 * i], dy = y - verts[3 * i + 1], dz = z - verts[3 * i + 2];
            const float nx = indicator[3 * i], ny = indicator[3 * i + 1], nz = indicator[3 * i + 2];
            const float r2 = dx * dx + dy * dy + dz * dz, r = sqrtf(r2), D = w1 + r, iD = 1.0f / D, iD2 = iD * iD;
            const float ir = r > 0.f ? 1.0f / r : 0.f;
            const float ux = dx * ir, uy = dy * ir, uz = dz * ir;
            const float a = dx * nx + dy * ny + dz * nz;
            const float A = gx * nx + gy * ny + gz * nz, B = gx * ux + gy * uy + gz * uz;
            const float tail = w1 * a + r * r2, lead = w1 * A + 3.0f * r2 * B;
            // d/dn: from ds  S w1 d / D;  from g  (w1 gvec D - w1 d B) / D^2
            const float cn = S * w1 * iD - w1 * B * iD2, cg = w1 * iD;
            if (dInd) {
                atomicAdd(dInd + 3 * i, wk * (cn * dx + cg * gx));
                atomicAdd(dInd + 3 * i + 1, wk * (cn * dy + cg * gy));
                atomicAdd(dInd + 3 * i + 2, wk * (cn * dz + cg * gz));
            }
            // d/dw1: from ds  S (a r - r^3) / D^2;  from g  (A D + lead - a B) / D^2 - 2 (lead D - tail B) / D^3
            gw1 += wk * (S * (a * r - r * r2) * iD2 + (A * D + lead - a * B) * iD2 - 2.0f * (lead * D - tail * B) * iD2 * iD);
        }
    }
    if (!dw1) return;
    gw1 = nm_t_wave_sum(gw1);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = gw1;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dw1, part[0] + part[1] + part[2] + part[3]);
}

// ------------------------------------------------------------------------------------------------ differentiable compositing
// renderer.py:264-333 on given sample SDFs (sdf_to_alpha :17-24, alpha_to_w :49-63, the weighted sums :299-333) with its reverse pass.
// One lane per ray, serial over the samples (the transmittance is a running product).  s = forward_s() is read from device memory
// (it is a function of the trained ln_s: its gradient is returned, and reading it on the host would drain the stream).
//   c_i = sigmoid(s sdf_i),  a_i = max((c_i - c_{i+1}) / (c_i + 1e-10), 0),  T_0 = 1,  T_{i+1} = T_i (1 - a_i + 1e-10),  w_i = a_i T_i
//   rgb = sum w_i rad_i (+ 1 - acc on a white background),  acc = sum w_i,  depth = sum w_i d_i / (acc + 1e-10),
//   normals = sum w_i nabla_i / max(|nabla_i|, 1e-12)
__device__ __forceinline__ float nm_t_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(64) void nm_t_composite_fwd_kernel(long long R, int N, const float* __restrict__ sdf, const float* __restrict__ s_ptr,
                                                                const float* __restrict__ dmid, int dmid_stride, const float* __restrict__ rad,
                                                                const float* __restrict__ nab, int white, float* __restrict__ rgb,
                                                                float* __restrict__ depth, float* __restrict__ acc, float* __restrict__ normals,
                                                                float* __restrict__ cdf, float* __restrict__ alpha, float* __restrict__ w_out,
                                                                float* __restrict__ trans) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float s = s_ptr[0];
    const float* sd = sdf + r * N;
    float c0 = nm_t_sigmoid(sd[0] * s), T = 1.0f;
    float cr = 0.f, cg = 0.f, cb = 0.f, A = 0.f, D = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
    for (int i = 0; i < N - 1; ++i) {
        const float c1 = nm_t_sigmoid(sd[i + 1] * s);
        const float a = fmaxf((c0 - c1) / (c0 + 1e-10f), 0.f);
        const float w = a * T;
        const long long m = r * (N - 1) + i;
        cdf[r * N + i] = c0; alpha[m] = a; w_out[m] = w; trans[m] = T;
        if (rad) { cr += w * rad[m * 3]; cg += w * rad[m * 3 + 1]; cb += w * rad[m * 3 + 2]; }
        A += w;
        D += w * dmid[r * dmid_stride + i];
        if (nab) {
            const float x = nab[(r * N + i) * 3], y = nab[(r * N + i) * 3 + 1], z = nab[(r * N + i) * 3 + 2];
            const float inv = 1.0f / fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f);
            nx += w * x * inv; ny += w * y * inv; nz += w * z * inv;
        }
        T = T * (1.0f - a + 1e-10f);
        c0 = c1;
    }
    cdf[r * N + N - 1] = c0;
    if (white) { cr += 1.0f - A; cg += 1.0f - A; cb += 1.0f - A; }
    rgb[r * 3] = cr; rgb[r * 3 + 1] = cg; rgb[r * 3 + 2] = cb;
    acc[r] = A;
    depth[r] = D / (A + 1e-10f);
    if (normals) { normals[r * 3] = nx; normals[r * 3 + 1] = ny; normals[r * 3 + 2] = nz; }
}

__global__ __launch_bounds__(64) void nm_t_composite_bwd_kernel(long long R, int N, const float* __restrict__ sdf, const float* __restrict__ s_ptr,
                                                                const float* __restrict__ dmid, int dmid_stride, const float* __restrict__ rad,
                                                                const float* __restrict__ nab, int white, const float* __restrict__ cdf,
                                                                const float* __restrict__ alpha, const float* __restrict__ w_in,
                                                                const float* __restrict__ trans, const float* __restrict__ acc,
                                                                const float* __restrict__ depth, const float* __restrict__ g_rgb,
                                                                const float* __restrict__ g_depth, const float* __restrict__ g_acc,
                                                                const float* __restrict__ g_normals, float* __restrict__ g_sdf,
                                                                float* __restrict__ g_rad, float* __restrict__ g_nab, float* __restrict__ g_s) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float gs = 0.f;
    if (r < R) {
        const float s = s_ptr[0];
        const float gr = g_rgb ? g_rgb[r * 3] : 0.f, gg = g_rgb ? g_rgb[r * 3 + 1] : 0.f, gb = g_rgb ? g_rgb[r * 3 + 2] : 0.f;
        const float gA = (g_acc ? g_acc[r] : 0.f) - (white ? gr + gg + gb : 0.f);
        const float gD = g_depth ? g_depth[r] : 0.f;
        const float gnx = g_normals ? g_normals[r * 3] : 0.f, gny = g_normals ? g_normals[r * 3 + 1] : 0.f, gnz = g_normals ? g_normals[r * 3 + 2] : 0.f;
        const float iA = 1.0f / (acc[r] + 1e-10f), dep = depth[r];
        float G = 0.f;            // cotangent of the transmittance entering the next interval
        float carry = 0.f;        // what interval i+1 contributed to the cotangent of c_{i+1} (as its first operand)
        for (int i = N - 2; i >= 0; --i) {
            const long long m = r * (N - 1) + i;
            const float a = alpha[m], w = w_in[m], T = trans[m];
            float wbar = gA + gD * (dmid[r * dmid_stride + i] - dep) * iA;
            if (rad) {
                wbar += gr * rad[m * 3] + gg * rad[m * 3 + 1] + gb * rad[m * 3 + 2];
                if (g_rad) { g_rad[m * 3] = w * gr; g_rad[m * 3 + 1] = w * gg; g_rad[m * 3 + 2] = w * gb; }
            }
            if (nab) {
                const float x = nab[(r * N + i) * 3], y = nab[(r * N + i) * 3 + 1], z = nab[(r * N + i) * 3 + 2];
                const float len = sqrtf(x * x + y * y + z * z), inv = 1.0f / fmaxf(len, 1e-12f);
                const float hx = x * inv, hy = y * inv, hz = z * inv;
                const float dot = gnx * hx + gny * hy + gnz * hz;
                wbar += dot;
                if (g_nab) {
                    // F.normalize: v / max(|v|, eps); below eps the denominator is the constant
                    const float k = len > 1e-12f ? dot : 0.f;
                    g_nab[(r * N + i) * 3] = w * (gnx - k * hx) * inv;
                    g_nab[(r * N + i) * 3 + 1] = w * (gny - k * hy) * inv;
                    g_nab[(r * N + i) * 3 + 2] = w * (gnz - k * hz) * inv;
                }
            }
            const float abar = (wbar - G) * T;
            G = wbar * a + G * (1.0f - a + 1e-10f);
            const float c0 = cdf[r * N + i], c1 = cdf[r * N + i + 1], den = c0 + 1e-10f;
            const bool on = (c0 - c1) / den >= 0.f;           // clamp_min passes the gradient at the kink too
            const float to_c1 = on ? -abar / den : 0.f, to_c0 = on ? abar * (c1 + 1e-10f) / (den * den) : 0.f;
            const float cb1 = carry + to_c1;                  // cotangent of c_{i+1} is complete
            const float k1 = cb1 * c1 * (1.0f - c1);
            g_sdf[r * N + i + 1] = k1 * s;
            gs += k1 * sdf[r * N + i + 1];
            carry = to_c0;
        }
        const float c0 = cdf[r * N], k0 = carry * c0 * (1.0f - c0);
        g_sdf[r * N] = k0 * s;
        gs += k0 * sdf[r * N];
        if (g_nab)      // the last sample's nabla is not used by the normals (renderer.py:326 takes [:N-1])
            g_nab[(r * N + N - 1) * 3] = g_nab[(r * N + N - 1) * 3 + 1] = g_nab[(r * N + N - 1) * 3 + 2] = 0.f;
    }
    if (!g_s) return;
    gs = nm_t_wave_sum(gs);
    if ((threadIdx.x & 63) == 0) atomicAdd(g_s, gs);
}
