// nm_mlp.h -- device-only: the fused "gather - interpolate - embed - MLP" kernels on fp32 MFMA.
//
// Reference semantics (file:line in the NeuMesh tree):
//   interpolation            models/frameworks/neumesh/neumesh.py:11-13
//   Embedder.forward         models/base.py:52-70 (per band: sin(x f) over all dims, cos(x f) ...)
//   _forward_density         neumesh.py:204-237  (177 -> 256 x3 softplus(beta=100) -> 1)
//   nabla                    neumesh.py:225-232  (autograd there; forward-mode tangent here)
//   _forward_color           neumesh.py:239-260  (207 -> 256 x4 ReLU -> 3 sigmoid)
//
// Tiling (gfx950, wave64): one workgroup = 4 waves = a 64-row x 256-column activation tile that
// lives in LDS (row stride 260 floats => ds_read_b128 of a column block is bank-conflict free).
// Wave w owns output columns [64w, 64w+64) for all 64 rows = 2x2 tiles of
// v_mfma_f32_32x32x2_f32 (4 independent accumulators: enough to saturate the fp32 matrix pipe
// from one wave per SIMD).  A operands come from LDS (16-byte reads, 8 k-values per lane per
// step), B operands straight from the packed weights in L2 (each wave reads a disjoint quarter
// of every layer, 32 contiguous bytes per lane per step; no LDS staging needed at the fp32 MFMA
// rate of 64 cycles/instruction).  The 2 lane-halves of a 32x32x2 MFMA take k-blocks
// [16J,16J+8) and [16J+8,16J+16): the sum over k is a permuted but fixed fp32 fma chain.
// With nabla: rows 0-31 are the points' activations h, rows 32-63 their tangents
// t = d h / d ds; both run through the same weights, t_out = (W t_in) * softplus'(z).
// LDS per workgroup: 66.8 KB => two workgroups per CU, so one workgroup's VALU phases
// (gather / sin-cos embedding / softplus epilogue) overlap the other's MFMA phase.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define NM_ROWS 64
#define NM_W 256
#define NM_LDS_STRIDE 260
#define NM_MAX_LAYERS 8

typedef float nm_f32x16 __attribute__((ext_vector_type(16)));

// Debug hook, compiled in only with -DNM_TESTING (the separate test / measurement library, neumesh_amd/build.py):
// workgroups 4096..4127 of an MLP launch record the shader clock at their phase boundaries (slot 0
// start, 1 after the prologue, then after each layer's MFMA loop and after its epilogue, last = end),
// 16 stamps per workgroup.  The production build has no trace of it.
#ifdef NM_TESTING
__device__ long long* g_nm_phase_log = nullptr;
__device__ __forceinline__ void nm_phase_stamp(int slot) {
    long long* log = g_nm_phase_log;
    if (log && blockIdx.x >= 4096u && blockIdx.x < 4128u && threadIdx.x == 0 && slot < 16)
        log[(blockIdx.x - 4096u) * 16 + slot] = (long long)clock64();
}
#else
__device__ __forceinline__ void nm_phase_stamp(int) {}
#endif

struct NmLayer {
    const float* W;  // packed [256][Kpad], zero padded
    const float* b;  // [256]
    int Kpad;        // multiple of 16
};

// Which K-NN record (ds / idx / w / grad produced by nm_distance_kernel) belongs to point q:
// compact (record q) when stride == 0; otherwise ray r = q / P, sample p = q % P and the record
// lives in ray r's slot array: r*stride + (slot ? slot[r*stride + off + p] : off + p).
// The slot indirection lets the final 128-sample pass reuse the records of the coarse and
// up-sampling passes (same points, hence bit-identical records) instead of searching again.
struct NmRecMap {
    int P, stride, off;
    const int* slot;
    // by_list (nm_mlp_h2.h geometry kernel with a point list): the list entry names (ray, sample position p); the
    // record is ray*stride + slot[ray*stride + off + p] and the outputs go to ray*out_stride + out_off + p
    int by_list;
};
__device__ __forceinline__ long long nm_rec_index(const NmRecMap& m, long long q) {
    if (m.stride == 0) return q;
    const long long r = q / m.P;
    const long long p = q - r * m.P;
    return r * m.stride + (m.slot ? (long long)m.slot[r * m.stride + m.off + p] : m.off + p);
}

// (q / P, q % P) for q = base + local, local < 2^16: one 64-bit division per workgroup (base is
// uniform), a 32-bit one per use -- the 64-bit software division costs ~120 instructions per lane.
struct NmDivBase {
    long long r0;
    unsigned p0, P;
};
__device__ __forceinline__ NmDivBase nm_div_base(long long base, int P) {
    NmDivBase d;
    d.P = (unsigned)P;
    if (P == 1) {  // compact lists (the mid-point pass): no division at all
        d.r0 = base;
        d.p0 = 0u;
        return d;
    }
    d.r0 = base / P;
    d.p0 = (unsigned)(base - d.r0 * P);
    return d;
}
__device__ __forceinline__ void nm_div_local(const NmDivBase& d, int local, long long& r, int& p) {
    const unsigned t = d.p0 + (unsigned)local;
    const unsigned dr = t / d.P;
    r = d.r0 + dr;
    p = (int)(t - dr * d.P);
}
__device__ __forceinline__ long long nm_rec_index_local(const NmRecMap& m, const NmDivBase& d, long long base, int local) {
    if (m.stride == 0) return base + local;
    long long r;
    int p;
    nm_div_local(d, local, r, p);
    return r * m.stride + (m.slot ? (long long)m.slot[r * m.stride + m.off + p] : m.off + p);
}

// Point lists addressed through a depth-bucket order (nm_rays_order_sort_kernel): point q = position in
// the list; groups of G rays own E consecutive positions, valid entries first, 0xFFFF = no point.
// order == nullptr: plain lists (every q < npts is a point).
struct NmSlotMap {
    const unsigned short* order;
    int G, P, E;
};
__device__ __forceinline__ bool nm_slot_valid(const NmSlotMap& m, long long q) { return !m.order || m.order[q] != 0xffffu; }
// (ray, sample) of position q; ray0 = first ray of q's group = (q / E) * G, which is the same for a whole
// workgroup tile (tiles never straddle groups), so the caller computes it once from the tile base
__device__ __forceinline__ void nm_slot_ray(const NmSlotMap& m, long long q, long long ray0, long long& ray, int& p) {
    const unsigned id = m.order[q];
    const unsigned rl = id / (unsigned)m.P;
    ray = ray0 + rl;
    p = (int)(id - rl * (unsigned)m.P);
}

struct NmGeoParams {
    NmLayer layer[NM_MAX_LAYERS];
    int D;
    const float* wd;  // density_linear weight [256]
    float bd;
    int multires_d, multires_fg, gdim;
    int d_emb, in_dim;  // 1+2*multires_d ; d_emb + gdim*(1+2*multires_fg)
};

struct NmColParams {
    NmLayer layer[NM_MAX_LAYERS];
    int D;
    const float* wrgb;  // [3][256]
    float brgb[3];
    int multires_d, multires_ft, multires_view, cdim, use_nabla;
    int d_emb, in_dim;
};

// softplus(beta=100, threshold=20) and its derivative (torch: x*beta > 20 ? x : log1p(exp(x*beta))/beta;
// backward x*beta > 20 ? 1 : z/(z+1), z = exp(x*beta)) on the hardware transcendental units:
// z = 2^(t*log2 e) (v_exp_f32), log1p(z) = log2(1+z)*ln 2 (v_log_f32) with the series z - z^2/2
// below 2^-10, z/(z+1) through v_rcp_f32.  A dozen instructions instead of ~150 for the libm
// expf/log1pf pair; absolute error vs float64 2.5e-8 (libm fp32: 1.6e-8), measured in
// tests/test_hostlogic.py::test_fast_softplus_formula.  The epilogue of every geometry layer
// evaluates this 16 K times per workgroup, so its cost is what decides whether the kernel is
// MFMA-bound.
__device__ __forceinline__ float nm_softplus100(float x, float* grad) {
    // branch- and select-free: z = e^min(100x, 21), y = max(x, log(1 + z) / 100), g = z / (1 + z).
    // Above torch's threshold (100x > 20, where it returns x and 1) the clamp makes log(1 + z)/100 <= 0.21 and
    // y = x from x = 0.21 on; in (0.2, 0.21] log(1 + z)/100 = x + e^-100x/100 = x to 2e-11, and g = 1 - 2e-9 (1.0f
    // or its fp32 neighbour).  For small z the rounding of 1 + z costs <= 6e-8 ABSOLUTE in log(1 + z), i.e.
    // <= 6e-10 in y -- below the fp32 spacing of the O(0.01..1) values the next layer sums
    // (tests/test_hostlogic.py::test_fast_softplus_formula).
    const float z = __builtin_amdgcn_exp2f(fminf(x * 144.269504f, 30.2965958f));
    const float u = 1.0f + z;
    if (grad) *grad = z * __builtin_amdgcn_rcpf(u);
    return fmaxf(x, __builtin_amdgcn_logf(u) * 0.0069314718f);
}

// One dense layer on the LDS tile: act[64][K] -> act[64][256] (in place).
// ACT: 0 softplus100 (geometry), 1 ReLU (colour).  TANGENT: rows 32-63 are tangents (no bias,
// multiplied by the activation derivative of the matching row 0-31).  k_hi: number of leading
// input columns that are non-zero for rows 32-63 (multiple of 16; = Kpad unless layer 0 of the
// tangent pass, where only the d-embedding columns are live).
template <int ACT, bool TANGENT>
__device__ __forceinline__ void nm_mlp_layer(float* act, const float* __restrict__ W, const float* __restrict__ bias,
                                             int Kpad, int k_hi, int stamp_slot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int n0 = wave * 64;
    nm_f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    const float* a0p = act + li * NM_LDS_STRIDE + 8 * h;
    const float* a1p = act + (32 + li) * NM_LDS_STRIDE + 8 * h;
    const float* b0p = W + (size_t)(n0 + li) * Kpad + 8 * h;
    const float* b1p = W + (size_t)(n0 + 32 + li) * Kpad + 8 * h;
    const int nJ = Kpad >> 4;
    const int nJ1 = k_hi >> 4;
    // operands of step J+1 are fetched while the 16-32 MFMAs of step J run
    float4 nb00 = *reinterpret_cast<const float4*>(b0p), nb01 = *reinterpret_cast<const float4*>(b0p + 4);
    float4 nb10 = *reinterpret_cast<const float4*>(b1p), nb11 = *reinterpret_cast<const float4*>(b1p + 4);
    float4 na00 = *reinterpret_cast<const float4*>(a0p), na01 = *reinterpret_cast<const float4*>(a0p + 4);
    float4 na10 = *reinterpret_cast<const float4*>(a1p), na11 = *reinterpret_cast<const float4*>(a1p + 4);
    for (int J = 0; J < nJ; ++J) {
        const float4 b00 = nb00, b01 = nb01, b10 = nb10, b11 = nb11;
        const float4 a00 = na00, a01 = na01, a10 = na10, a11 = na11;
        if (J + 1 < nJ) {
            const int o = 16 * (J + 1);
            nb00 = *reinterpret_cast<const float4*>(b0p + o);
            nb01 = *reinterpret_cast<const float4*>(b0p + o + 4);
            nb10 = *reinterpret_cast<const float4*>(b1p + o);
            nb11 = *reinterpret_cast<const float4*>(b1p + o + 4);
            na00 = *reinterpret_cast<const float4*>(a0p + o);
            na01 = *reinterpret_cast<const float4*>(a0p + o + 4);
            if (J + 1 < nJ1) {
                na10 = *reinterpret_cast<const float4*>(a1p + o);
                na11 = *reinterpret_cast<const float4*>(a1p + o + 4);
            }
        }
        const float av0[8] = {a00.x, a00.y, a00.z, a00.w, a01.x, a01.y, a01.z, a01.w};
        const float bv0[8] = {b00.x, b00.y, b00.z, b00.w, b01.x, b01.y, b01.z, b01.w};
        const float bv1[8] = {b10.x, b10.y, b10.z, b10.w, b11.x, b11.y, b11.z, b11.w};
        if (J < nJ1) {
            const float av1[8] = {a10.x, a10.y, a10.z, a10.w, a11.x, a11.y, a11.z, a11.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv0[e], acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv1[e], acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv0[e], acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv1[e], acc11, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv0[e], acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv1[e], acc01, 0, 0, 0);
            }
        }
    }
    __syncthreads();  // every wave has finished reading the input tile
    nm_phase_stamp(stamp_slot);
    const float bias0 = bias[n0 + li], bias1 = bias[n0 + 32 + li];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;  // MFMA 32x32 C/D layout
        float* o0 = act + row * NM_LDS_STRIDE + n0 + li;
        float* o1 = act + (32 + row) * NM_LDS_STRIDE + n0 + li;
        const float z00 = acc00[reg] + bias0, z01 = acc01[reg] + bias1;
        if (TANGENT) {
            float g0, g1;
            float y0, y1;
            if (ACT == 0) {
                y0 = nm_softplus100(z00, &g0);
                y1 = nm_softplus100(z01, &g1);
            } else {
                y0 = fmaxf(z00, 0.f); g0 = z00 > 0.f ? 1.f : 0.f;
                y1 = fmaxf(z01, 0.f); g1 = z01 > 0.f ? 1.f : 0.f;
            }
            o0[0] = y0;
            o0[32] = y1;
            o1[0] = acc10[reg] * g0;
            o1[32] = acc11[reg] * g1;
        } else {
            const float z10 = acc10[reg] + bias0, z11 = acc11[reg] + bias1;
            if (ACT == 0) {
                o0[0] = nm_softplus100(z00, nullptr);
                o0[32] = nm_softplus100(z01, nullptr);
                o1[0] = nm_softplus100(z10, nullptr);
                o1[32] = nm_softplus100(z11, nullptr);
            } else {
                o0[0] = fmaxf(z00, 0.f);
                o0[32] = fmaxf(z01, 0.f);
                o1[0] = fmaxf(z10, 0.f);
                o1[32] = fmaxf(z11, 0.f);
            }
        }
    }
    __syncthreads();
    nm_phase_stamp(stamp_slot + 1);
}

// Same layer on the scalar ALUs (one output element per thread-iteration, plain fmaf chain in
// natural k order).  NOT part of the product path: used by nm_selfcheck_* to cross-check the
// MFMA tile code on the device (catches fragment-layout mistakes that a symmetric test misses).
template <int ACT, bool TANGENT>
__device__ void nm_mlp_layer_valu(float* act, float* tmp /*[64][256] global*/, const float* __restrict__ W,
                                  const float* __restrict__ bias, int Kpad) {
    for (int e = threadIdx.x; e < NM_ROWS * NM_W; e += blockDim.x) {
        const int row = e >> 8, n = e & 255;
        float s = 0.f;
        for (int k = 0; k < Kpad; ++k) s = fmaf(act[row * NM_LDS_STRIDE + k], W[(size_t)n * Kpad + k], s);
        tmp[e] = s;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NM_ROWS * NM_W; e += blockDim.x) {
        const int row = e >> 8, n = e & 255;
        float y;
        if (TANGENT && row >= 32) {
            const float z = tmp[(row - 32) * NM_W + n] + bias[n];
            float g;
            if (ACT == 0) nm_softplus100(z, &g); else g = z > 0.f ? 1.f : 0.f;
            y = tmp[e] * g;
        } else {
            const float z = tmp[e] + bias[n];
            y = ACT == 0 ? nm_softplus100(z, nullptr) : fmaxf(z, 0.f);
        }
        act[row * NM_LDS_STRIDE + n] = y;
    }
    __syncthreads();
}

// sin and cos for the positional encodings (models/base.py:52-70).  ocml's sincosf is a
// full-range routine (~280 instructions with its large-argument path) and cost ~20 K cycles of
// every workgroup's prologue; the encodings only see |x| < ~1e3 (codes * 2^b, ds * 2^7, view
// directions * 2^3), so: three-term Cody-Waite reduction by pi/2 (fma) + the single-precision
// minimax polynomials on [-pi/4, pi/4].  Max abs error vs float64 9.2e-8 for |x| <= 1e5 (libm
// fp32: 7e-8), checked in tests/test_hostlogic.py::test_fast_sincos_formula; larger arguments
// take the libm path.
#define NM_SINCOS_FAST_MAX 1.0e5f  // beyond: libm's sincosf (full range reduction)
__device__ __forceinline__ void nm_sincos_fast(float x, float* sn, float* cs);
__device__ __forceinline__ void nm_sincos(float x, float* sn, float* cs) {
    if (fabsf(x) > NM_SINCOS_FAST_MAX) {
        sincosf(x, sn, cs);
        return;
    }
    nm_sincos_fast(x, sn, cs);
}
// |x| <= NM_SINCOS_FAST_MAX (the caller has checked): three-term Cody-Waite reduction + degree-7 / degree-8 polynomials
__device__ __forceinline__ void nm_sincos_fast(float x, float* sn, float* cs) {
    const float k = rintf(x * 0.63661977236758134f);
    float r = fmaf(k, -1.57079637050628662109e+00f, x);
    r = fmaf(k, 4.37113882867379288655e-08f, r);
    r = fmaf(k, 1.71512451000588187280e-15f, r);
    const int q = (int)k;
    const float r2 = r * r;
    float ps = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(ps, r2, -1.6666654611e-1f);
    const float s = fmaf(ps * r2, r, r);
    float pc = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(pc, r2, 4.166664568298827e-2f);
    const float c = fmaf(pc * r2, r2, fmaf(r2, -0.5f, 1.0f));
    const bool swap = (q & 1) != 0;
    const float ss = swap ? c : s, cc = swap ? s : c;
    *sn = (q & 2) ? -ss : ss;
    *cs = ((q + 1) & 2) ? -cc : cc;
}

// writes x and its `bands` sin/cos bands for `dim`-wide feature vector chunk (4 values at
// feature position 4*chunk) into an embedding that starts at row[0]
__device__ __forceinline__ void nm_embed4(float* row, int dim, int bands, int chunk, float4 x) {
    const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = 4 * chunk + e;
        row[c] = xs[e];
        float f = 1.0f;
        for (int b = 0; b < bands; ++b) {
            float s, co;
            nm_sincos(xs[e] * f, &s, &co);
            row[dim * (1 + 2 * b) + c] = s;
            row[dim * (2 + 2 * b) + c] = co;
            f *= 2.0f;
        }
    }
}

// ------------------------------------------------------------------ geometry MLP kernel
// Points q in [0, npts): inputs ds[rec], fg_rec[rec][gdim] (interpolated geometry code, from the
// K-NN/distance kernel) and, with NABLA, grad[rec][3] = d ds/d xyz; rec = nm_rec_index(rmap, q).  Output sdf to sdf_out[(q / P) * stride + off + q % P]
// (P = samples per ray of this call; P = 1, stride = 1 for flat outputs) and
// nabla_out[q][3] = (d sdf/d ds) * grad[q]  (nabla_slotted: written at the sdf_out position instead of q).
template <bool NABLA, bool VALU_CHECK>
__global__ __launch_bounds__(256, 2) void nm_geo_mlp_kernel(NmGeoParams prm, const float* __restrict__ fg_rec,
                                                            const float* __restrict__ ds, const float* __restrict__ grad,
                                                            NmRecMap rmap, long long npts, float* __restrict__ sdf_out, int P,
                                                            int stride, int off, float* __restrict__ nabla_out,
                                                            float* __restrict__ valu_tmp, int nabla_slotted, NmSlotMap smap) {
    __shared__ __attribute__((aligned(16))) float act[NM_ROWS * NM_LDS_STRIDE + NM_ROWS];
    float* red = act + NM_ROWS * NM_LDS_STRIDE;
    constexpr int PTS = NABLA ? 32 : 64;
    const long long base = (long long)blockIdx.x * PTS;
    if (smap.order && smap.order[base] == 0xffffu) return;  // no point in this tile (valid entries lead each group)
    const NmDivBase rdiv = nm_div_base(base, rmap.stride ? rmap.P : 1), odiv = nm_div_base(base, P);
    nm_phase_stamp(0);
    const int Kpad0 = prm.layer[0].Kpad;
    const int t_hi = ((prm.d_emb + 15) >> 4) << 4;  // live tangent columns, rounded to 16

    // ---- prologue: build the input rows
    for (int task = threadIdx.x; task < PTS * 8; task += 256) {
        const int p = task >> 3, j = task & 7;
        const long long q = base + p;
        float* row = act + p * NM_LDS_STRIDE;
        float* trow = act + (32 + p) * NM_LDS_STRIDE;
        // zero padding columns (and the whole row of out-of-range points)
        if (q >= npts || !nm_slot_valid(smap, q)) {
            for (int c = j; c < Kpad0; c += 8) row[c] = 0.f;
            if (NABLA) for (int c = j; c < Kpad0; c += 8) trow[c] = 0.f;
            continue;
        }
        for (int c = prm.in_dim + j; c < Kpad0; c += 8) row[c] = 0.f;
        const long long rq = nm_rec_index_local(rmap, rdiv, base, p);
        const float dsv = ds[rq];
        if (j == 0) {
            row[0] = dsv;
            if (NABLA) trow[0] = 1.0f;
        }
        if (NABLA) for (int c = prm.d_emb + j; c < Kpad0; c += 8) trow[c] = 0.f;
        for (int b = j; b < prm.multires_d; b += 8) {
            const float f = (float)(1 << b);
            float s, co;
            nm_sincos(dsv * f, &s, &co);
            row[1 + 2 * b] = s;
            row[2 + 2 * b] = co;
            if (NABLA) {
                trow[1 + 2 * b] = f * co;
                trow[2 + 2 * b] = -f * s;
            }
        }
        // interpolated geometry code of the point: gathered by the K-NN kernel's epilogue
        // (nm_gather_interp), so this is one coalesced 16-byte load instead of a dependent
        // index -> 8-row gather chain in front of the MFMA phase
        for (int chunk = j; chunk < (prm.gdim >> 2); chunk += 8) {
            const float4 fg = *reinterpret_cast<const float4*>(fg_rec + rq * prm.gdim + 4 * chunk);
            nm_embed4(row + prm.d_emb, prm.gdim, prm.multires_fg, chunk, fg);
        }
    }
    __syncthreads();
    nm_phase_stamp(1);

    // ---- hidden layers
    for (int l = 0; l < prm.D; ++l) {
        const NmLayer L = prm.layer[l];
        const int k_hi = (NABLA && l == 0) ? t_hi : L.Kpad;
        if (VALU_CHECK) nm_mlp_layer_valu<0, NABLA>(act, valu_tmp + (size_t)blockIdx.x * NM_ROWS * NM_W, L.W, L.b, L.Kpad);
        else nm_mlp_layer<0, NABLA>(act, L.W, L.b, L.Kpad, k_hi, 2 + 2 * l);
    }

    // ---- density_linear (neumesh.py:101,218): 4 threads per row, interleaved columns
    {
        const int row = threadIdx.x >> 2, q4 = threadIdx.x & 3;
        const float* a = act + row * NM_LDS_STRIDE;
        float s = 0.f;
        for (int m = 0; m < 64; ++m) s = fmaf(a[q4 + 4 * m], prm.wd[q4 + 4 * m], s);
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (q4 == 0) red[row] = s;
    }
    __syncthreads();
    if (threadIdx.x < PTS) {
        const long long q = base + threadIdx.x;
        if (q < npts && nm_slot_valid(smap, q)) {
            const float sdf = red[threadIdx.x] + prm.bd;
            long long orow;
            int op;
            nm_div_local(odiv, (int)threadIdx.x, orow, op);
            const long long oidx = orow * stride + off + op;  // (ray, sample) addressed output position
            if (sdf_out) sdf_out[oidx] = sdf;
            if (NABLA && nabla_out) {
                const float dsdf = red[32 + threadIdx.x];
                const long long rq = nm_rec_index_local(rmap, rdiv, base, (int)threadIdx.x);
                const long long no = nabla_slotted ? oidx : q;
                nabla_out[no * 3 + 0] = dsdf * grad[rq * 3 + 0];
                nabla_out[no * 3 + 1] = dsdf * grad[rq * 3 + 1];
                nabla_out[no * 3 + 2] = dsdf * grad[rq * 3 + 2];
            }
        }
    }
    nm_phase_stamp(15);
}

// ------------------------------------------------------------------ colour MLP kernel
// rgb[q] = sigmoid(W_rgb relu(...relu(W_0 [nabla, embed_d(ds), embed_view(dir), embed_ft(ft)] + b_0)...) + b_rgb)
// dirs: view direction of point q is dirs[(q / dir_div) * 3 ..] (dir_div = samples per ray when
// every sample of a ray shares the ray direction, 1 for per-point directions).
template <bool VALU_CHECK>
__global__ __launch_bounds__(256, 2) void nm_col_mlp_kernel(NmColParams prm, const float* __restrict__ ft_rec,
                                                            const float* __restrict__ ds, const float* __restrict__ nabla,
                                                            const float* __restrict__ dirs, int dir_div, long long npts,
                                                            float* __restrict__ rgb_out, float* __restrict__ valu_tmp, NmSlotMap smap) {
    __shared__ __attribute__((aligned(16))) float act[NM_ROWS * NM_LDS_STRIDE + 3 * NM_ROWS];
    float* red = act + NM_ROWS * NM_LDS_STRIDE;
    const long long base = (long long)blockIdx.x * NM_ROWS;
    if (smap.order && smap.order[base] == 0xffffu) return;  // no point in this tile
    const NmDivBase ddiv = nm_div_base(base, dir_div);
    const long long ray0 = smap.order ? (base / smap.E) * smap.G : 0;  // uniform: one division per workgroup
    nm_phase_stamp(0);
    const int Kpad0 = prm.layer[0].Kpad;
    const int o_d = prm.use_nabla ? 3 : 0;             // start of embed_d
    const int o_v = o_d + prm.d_emb;                   // start of embed_view
    const int o_f = o_v + 3 * (1 + 2 * prm.multires_view);  // start of embed_ft

    for (int task = threadIdx.x; task < NM_ROWS * 8; task += 256) {
        const int p = task >> 3, j = task & 7;
        const long long q = base + p;
        float* row = act + p * NM_LDS_STRIDE;
        if (q >= npts || !nm_slot_valid(smap, q)) {
            for (int c = j; c < Kpad0; c += 8) row[c] = 0.f;
            continue;
        }
        for (int c = prm.in_dim + j; c < Kpad0; c += 8) row[c] = 0.f;
        const float dsv = ds[q];
        if (j == 0) {
            row[o_d] = dsv;
            if (prm.use_nabla) {
                row[0] = nabla[q * 3 + 0];
                row[1] = nabla[q * 3 + 1];
                row[2] = nabla[q * 3 + 2];
            }
        }
        for (int b = j; b < prm.multires_d; b += 8) {
            float s, co;
            nm_sincos(dsv * (float)(1 << b), &s, &co);
            row[o_d + 1 + 2 * b] = s;
            row[o_d + 2 + 2 * b] = co;
        }
        {
            long long ray;
            int unused_p;
            if (smap.order) nm_slot_ray(smap, q, ray0, ray, unused_p);
            else nm_div_local(ddiv, p, ray, unused_p);
            const float* dv = dirs + ray * 3;
            if (j == 1) {
                row[o_v] = dv[0];
                row[o_v + 1] = dv[1];
                row[o_v + 2] = dv[2];
            }
            for (int e = j; e < 3 * prm.multires_view; e += 8) {
                const int dim = e % 3, b = e / 3;
                float s, co;
                nm_sincos(dv[dim] * (float)(1 << b), &s, &co);
                row[o_v + 3 + 6 * b + dim] = s;
                row[o_v + 6 + 6 * b + dim] = co;
            }
        }
        for (int chunk = j; chunk < (prm.cdim >> 2); chunk += 8) {
            const float4 ft = *reinterpret_cast<const float4*>(ft_rec + q * prm.cdim + 4 * chunk);
            nm_embed4(row + o_f, prm.cdim, prm.multires_ft, chunk, ft);
        }
    }
    __syncthreads();
    nm_phase_stamp(1);

    for (int l = 0; l < prm.D; ++l) {
        const NmLayer L = prm.layer[l];
        if (VALU_CHECK) nm_mlp_layer_valu<1, false>(act, valu_tmp + (size_t)blockIdx.x * NM_ROWS * NM_W, L.W, L.b, L.Kpad);
        else nm_mlp_layer<1, false>(act, L.W, L.b, L.Kpad, L.Kpad, 2 + 2 * l);
    }

    {
        const int row = threadIdx.x >> 2, q4 = threadIdx.x & 3;
        const float* a = act + row * NM_LDS_STRIDE;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int m = 0; m < 64; ++m) {
            const float av = a[q4 + 4 * m];
            s0 = fmaf(av, prm.wrgb[q4 + 4 * m], s0);
            s1 = fmaf(av, prm.wrgb[256 + q4 + 4 * m], s1);
            s2 = fmaf(av, prm.wrgb[512 + q4 + 4 * m], s2);
        }
        s0 += __shfl_xor(s0, 1); s0 += __shfl_xor(s0, 2);
        s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2);
        s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2);
        if (q4 == 0) {
            red[3 * row] = s0;
            red[3 * row + 1] = s1;
            red[3 * row + 2] = s2;
        }
    }
    __syncthreads();
    if (threadIdx.x < NM_ROWS * 3) {
        const int p = threadIdx.x / 3, c = threadIdx.x % 3;
        const long long q = base + p;
        if (q < npts && nm_slot_valid(smap, q)) {
            const float z = red[threadIdx.x] + prm.brgb[c];
            long long oq = q;  // ordered lists: the colour goes back to its (ray, sample) position
            if (smap.order) {
                long long ray;
                int sp;
                nm_slot_ray(smap, q, ray0, ray, sp);
                oq = ray * smap.P + sp;
            }
            rgb_out[oq * 3 + c] = __fdiv_rn(1.0f, 1.0f + expf(-z));
        }
    }
    nm_phase_stamp(15);
}
