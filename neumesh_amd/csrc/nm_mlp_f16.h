// nm_mlp_f16.h -- the same fused embed + MLP kernels as nm_mlp.h, on the f16 matrix pipe with
// "split-half" operands (precision mode 1).
//
// Why: on gfx950 the fp32-input MFMA executes at the vector-FMA rate and does not overlap VALU work
// (measured: one workgroup per CU 91-108 TFLOP/s, two 107-124; removing VALU work from a phase does
// not shorten a workgroup's life), so ~125 of the 157 TFLOP/s are this path's ceiling.  The f16 MFMA
// (v_mfma_f32_32x32x16_f16) has 16x the rate and its own pipe.
//
// How: every fp32 value a is carried as two halves  a = h1 + h2 * 2^-11,  h1 = rne16(a),
// h2 = rne16((a - h1) * 2^11)  -- 22 significant bits, error <= 2^-22 |a| (fp32: 2^-24) -- and a
// product as  a*b = h1a*h1b + 2^-11 (h1a*h2b + h2a*h1b)  (+ h2a*h2b*2^-22, dropped): 3 f16 MFMAs with
// fp32 accumulation in two accumulators (main / 2^11-scaled), recombined in the epilogue.  The two
// halves occupy exactly the 4 bytes of the fp32 value, so the LDS tile (64 rows x 256 columns) and
// the packed weights keep their size; activations are split ONCE, by the epilogue that produces
// them, so the K loop is pure operand loads + MFMA (12 MFMAs = 384 cycles per 16 k-values per wave,
// against 2048 cycles for the fp32 form).
//
// Range: |values| must stay below 65504 (fp16); fp16 subnormals are exact on the matrix pipe
// (probe: tools/mfma_denorm.hip), so small values need no special case.  The tangent rows (d h / d ds, seeded
// with up to 2^7 * cos) are linear in their seed, so they are carried scaled by 2^-8 and the final
// d sdf / d ds is multiplied back by 2^8 (exact), which keeps them far from the fp16 range limit.
// An overflow shows up as a non-finite output.  nm_field_desc.mlp_precision selects
// the mode; the fp32 kernels of nm_mlp.h remain the reference.
#pragma once

#include "nm_mlp.h"

typedef _Float16 nm_h8 __attribute__((ext_vector_type(8)));

#define NM_TANGENT_SCALE 0.00390625f  // 2^-8
#define NM_H_STRIDE 264  // halves per tile row: 528 B = 33 16-byte slots -> conflict-free ds_read_b128
#define NM_H_PLANE (NM_ROWS * NM_H_STRIDE)
#ifndef NM_EXP_LDS_PAD
#define NM_EXP_LDS_PAD 0  // experiments: extra LDS halves per workgroup (forces 1 workgroup per CU)
#endif

struct NmLayerH {
    const _Float16* W;  // packed fragments: [col tile 8][k-step Kpad/16][plane 2][lane 64][8 halves]
    const float* b;     // [256] fp32
    int Kpad;
};

struct NmGeoParamsH {
    NmLayerH layer[NM_MAX_LAYERS];
    int D;
    const float* wd;
    float bd;
    int multires_d, multires_fg, gdim;
    int d_emb, in_dim;
};

struct NmColParamsH {
    NmLayerH layer[NM_MAX_LAYERS];
    int D;
    const float* wrgb;
    float brgb[3];
    int multires_d, multires_ft, multires_view, cdim, use_nabla;
    int d_emb, in_dim;
};

// a -> (h1, h2): a ~= h1 + h2 / 2048.  fp16 subnormals are fine on both sides: v_cvt_f16_f32 produces
// them and the matrix pipe consumes them exactly (probe: tools/mfma_denorm.hip on gfx950).
__device__ __forceinline__ void nm_split_half(float a, _Float16* h1, _Float16* h2) {
    const _Float16 p = (_Float16)a;
    *h1 = p;
    *h2 = (_Float16)((a - (float)p) * 2048.0f);
}
__device__ __forceinline__ void nm_store_split(_Float16* tile, int row, int col, float a) {
    _Float16 h1, h2;
    nm_split_half(a, &h1, &h2);
    tile[row * NM_H_STRIDE + col] = h1;
    tile[NM_H_PLANE + row * NM_H_STRIDE + col] = h2;
}
__device__ __forceinline__ float nm_load_split(const _Float16* tile, int row, int col) {
    return fmaf((float)tile[NM_H_PLANE + row * NM_H_STRIDE + col], 1.0f / 2048.0f, (float)tile[row * NM_H_STRIDE + col]);
}

// weights: fp32 [256][in_dim] -> split halves in MFMA-fragment order (one 1 KiB block per
// (column tile, k-step, plane): a wave's B operand load is 64 lanes x 16 contiguous bytes)
__global__ void nm_pack_weight_h_kernel(const float* __restrict__ src, int in_dim, int Kpad, _Float16* __restrict__ dst) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (n, k)
    if (e >= NM_W * Kpad) return;
    const int n = e / Kpad, k = e - n * Kpad;
    const float w = k < in_dim ? src[(size_t)n * in_dim + k] : 0.f;
    _Float16 h1, h2;
    nm_split_half(w, &h1, &h2);
    const int ct = n >> 5, lane = (n & 31) | (((k >> 3) & 1) << 5), ks = k >> 4, el = k & 7;
    const int KS = Kpad >> 4;
    const size_t base = ((size_t)(ct * KS + ks) * 2) * 64 * 8;
    dst[base + (size_t)lane * 8 + el] = h1;
    dst[base + 64 * 8 + (size_t)lane * 8 + el] = h2;
}

// Work split inside a workgroup (64 activation rows x 256 output columns per layer): a wave owns all
// 64 rows of CT 32-column tiles, so there are 8/CT waves.  CT = 2: 256 threads, two workgroups per CU
// = 2 waves per SIMD.  CT = 1 (512 threads, 4 waves per SIMD, half the accumulators per wave) was
// measured and is slower -- geometry MLP 1.81 vs 1.43 ms per 2^20 points: every wave re-reads the
// whole A tile from LDS, the 128-register budget spills, and four waves sharing one matrix pipe
// stretch each K phase more than the extra epilogue overlap returns -- so only CT = 2 is built.
#define NM_H_CT 2
#define NM_H_THREADS (64 * 8 / NM_H_CT)
#define NM_H_WAVES_PER_SIMD (4 / NM_H_CT)  // two workgroups per CU (LDS: 2 x 68 KiB)

// B operand (weights) of one k-step for the CT column tiles of a wave: 2*CT x 16 bytes per lane.
template <int CT>
struct NmBFrag {
    nm_h8 a[CT], b[CT];  // plane h1, plane h2
};
__device__ __forceinline__ const nm_h8* nm_b_base(const _Float16* W, int Kpad, int ctile) {
    return reinterpret_cast<const nm_h8*>(W) + (size_t)ctile * (Kpad >> 4) * 2 * 64 + (threadIdx.x & 63);
}
template <int CT>
__device__ __forceinline__ NmBFrag<CT> nm_ld_b(const nm_h8* const (&bp)[CT], int ks) {
    NmBFrag<CT> f;
    const int o = ks * 128;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        f.a[c] = bp[c][o];
        f.b[c] = bp[c][o + 64];
    }
    return f;
}
// the first two k-steps of a layer (issued early: before the input phase / the previous epilogue)
template <int CT>
__device__ __forceinline__ void nm_prefetch_b(const NmLayerH L, NmBFrag<CT>& p0, NmBFrag<CT>& p1) {
    const int wave = threadIdx.x >> 6;
    const nm_h8* bp[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) bp[c] = nm_b_base(L.W, L.Kpad, wave * CT + c);
    p0 = nm_ld_b<CT>(bp, 0);
    p1 = nm_ld_b<CT>(bp, 1);  // Kpad >= 32 always (in_dim >= 17)
}

template <int CT>
struct NmAccH {  // 2 row tiles x CT column tiles of a wave, main and 2^11-scaled accumulators
    nm_f32x16 hi[2][CT], lo[2][CT];
};

// the 6*CT MFMAs of one k-step: a[rt][plane], F = weights
#define NM_H_MFMAS(A, F)                                                                              \
    _Pragma("unroll") for (int rt_ = 0; rt_ < 2; ++rt_) _Pragma("unroll") for (int c_ = 0; c_ < CT; ++c_)   \
        c.hi[rt_][c_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[rt_][0], F.a[c_], c.hi[rt_][c_], 0, 0, 0);  \
    _Pragma("unroll") for (int rt_ = 0; rt_ < 2; ++rt_) _Pragma("unroll") for (int c_ = 0; c_ < CT; ++c_)   \
        c.lo[rt_][c_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[rt_][0], F.b[c_], c.lo[rt_][c_], 0, 0, 0);  \
    _Pragma("unroll") for (int rt_ = 0; rt_ < 2; ++rt_) _Pragma("unroll") for (int c_ = 0; c_ < CT; ++c_)   \
        c.lo[rt_][c_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[rt_][1], F.a[c_], c.lo[rt_][c_], 0, 0, 0);

// K loop of one layer, fully unrolled for a compile-time number of k-steps.
// The weights come straight from L2 (each wave owns its own output columns, nothing is shared
// inside the workgroup), ~700 cycles away, while one k-step is 6*CT MFMAs of 32 matrix-pipe cycles:
// the B fragments are fetched TWO steps ahead (three rotating register sets; the first two steps
// arrive in pre0/pre1, requested before the previous epilogue), the A fragments (LDS) one step.
// Straight-line code on purpose: in a rolled loop the compiler drains ALL outstanding loads once
// per iteration (s_waitcnt vmcnt(0) for the loop-carried ones) and, left alone, its scheduler sinks
// the prefetches down to their first use; unrolled, each step waits for exactly its own fragment,
// and the sched_barriers pin the issue points.
template <int KS, int CT>
__device__ __forceinline__ void nm_kloop_h(const _Float16* a0p, const _Float16* a1p, const nm_h8* const (&bp)[CT],
                                           const NmBFrag<CT>& pre0, const NmBFrag<CT>& pre1, NmAccH<CT>& c) {
    NmBFrag<CT> f[3];
    f[0] = pre0;
    f[1] = pre1;
    nm_h8 a[2][2][2];  // [buffer][row tile][plane]
    a[0][0][0] = *reinterpret_cast<const nm_h8*>(a0p);
    a[0][0][1] = *reinterpret_cast<const nm_h8*>(a0p + NM_H_PLANE);
    a[0][1][0] = *reinterpret_cast<const nm_h8*>(a1p);
    a[0][1][1] = *reinterpret_cast<const nm_h8*>(a1p + NM_H_PLANE);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 2 < KS) f[(ks + 2) % 3] = nm_ld_b<CT>(bp, ks + 2);
        if (ks + 1 < KS) {
            const int oa = (ks + 1) * 16;
            a[(ks + 1) & 1][0][0] = *reinterpret_cast<const nm_h8*>(a0p + oa);
            a[(ks + 1) & 1][0][1] = *reinterpret_cast<const nm_h8*>(a0p + NM_H_PLANE + oa);
            a[(ks + 1) & 1][1][0] = *reinterpret_cast<const nm_h8*>(a1p + oa);
            a[(ks + 1) & 1][1][1] = *reinterpret_cast<const nm_h8*>(a1p + NM_H_PLANE + oa);
        }
        __builtin_amdgcn_sched_barrier(0);
        NM_H_MFMAS(a[ks & 1], f[ks % 3])
    }
    __builtin_amdgcn_sched_barrier(0);
}

// any other layer width: rolled loop, fragments one step ahead
template <int CT>
__device__ __forceinline__ void nm_kloop_h_generic(int KS, const _Float16* a0p, const _Float16* a1p,
                                                   const nm_h8* const (&bp)[CT], const NmBFrag<CT>& pre0, NmAccH<CT>& c) {
    NmBFrag<CT> nf = pre0;
    nm_h8 na[2][2];
    na[0][0] = *reinterpret_cast<const nm_h8*>(a0p);
    na[0][1] = *reinterpret_cast<const nm_h8*>(a0p + NM_H_PLANE);
    na[1][0] = *reinterpret_cast<const nm_h8*>(a1p);
    na[1][1] = *reinterpret_cast<const nm_h8*>(a1p + NM_H_PLANE);
    for (int ks = 0; ks < KS; ++ks) {
        const NmBFrag<CT> F = nf;
        nm_h8 A[2][2];
        A[0][0] = na[0][0]; A[0][1] = na[0][1]; A[1][0] = na[1][0]; A[1][1] = na[1][1];
        if (ks + 1 < KS) {
            nf = nm_ld_b<CT>(bp, ks + 1);
            const int oa = (ks + 1) * 16;
            na[0][0] = *reinterpret_cast<const nm_h8*>(a0p + oa);
            na[0][1] = *reinterpret_cast<const nm_h8*>(a0p + NM_H_PLANE + oa);
            na[1][0] = *reinterpret_cast<const nm_h8*>(a1p + oa);
            na[1][1] = *reinterpret_cast<const nm_h8*>(a1p + NM_H_PLANE + oa);
        }
        NM_H_MFMAS(A, F)
    }
}

// One dense layer on the split-half LDS tile (in place), see the header comment.
// pre0/pre1: B fragments of this layer's first two k-steps on entry, of the next layer's on exit
// (requested before the epilogue so that they arrive while it runs).
template <int ACT, bool TANGENT, int CT>
__device__ __forceinline__ void nm_mlp_layer_h(_Float16* tile, const NmLayerH L, const bool has_next, const NmLayerH next,
                                               NmBFrag<CT>& pre0, NmBFrag<CT>& pre1, int stamp_slot) {
    const float* __restrict__ bias = L.b;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int n0 = wave * 32 * CT;
    const int KS = L.Kpad >> 4;
    const _Float16* a0p = tile + li * NM_H_STRIDE + 8 * h;
    const _Float16* a1p = tile + (32 + li) * NM_H_STRIDE + 8 * h;
    const nm_h8* bp[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) bp[c] = nm_b_base(L.W, L.Kpad, wave * CT + c);
    NmAccH<CT> c;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) c.hi[rt][ct] = c.lo[rt][ct] = nm_f32x16{0};
    switch (KS) {  // the widths of the reference configuration get the unrolled form
        case 16: nm_kloop_h<16, CT>(a0p, a1p, bp, pre0, pre1, c); break;  // hidden layers (W = 256)
        case 12: nm_kloop_h<12, CT>(a0p, a1p, bp, pre0, pre1, c); break;  // geometry input 177 -> 192
        case 13: nm_kloop_h<13, CT>(a0p, a1p, bp, pre0, pre1, c); break;  // colour input 207 -> 208
        default: nm_kloop_h_generic<CT>(KS, a0p, a1p, bp, pre0, c); break;
    }
    if (has_next) nm_prefetch_b<CT>(next, pre0, pre1);
    __syncthreads();  // every wave has finished reading the input tile
    nm_phase_stamp(stamp_slot);
    const float sc = 1.0f / 2048.0f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int col = n0 + 32 * ct + li;
        const float bv = bias[col];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;  // MFMA 32x32 C/D layout
            const float z0 = fmaf(c.lo[0][ct][reg], sc, c.hi[0][ct][reg]) + bv;
            const float t1 = fmaf(c.lo[1][ct][reg], sc, c.hi[1][ct][reg]);
            if (TANGENT) {
                float g0, y0;
                if (ACT == 0) {
                    y0 = nm_softplus100(z0, &g0);
                } else {
                    y0 = fmaxf(z0, 0.f);
                    g0 = z0 > 0.f ? 1.f : 0.f;
                }
                nm_store_split(tile, row, col, y0);
                nm_store_split(tile, 32 + row, col, t1 * g0);
            } else {
                const float z1 = t1 + bv;
                if (ACT == 0) {
                    nm_store_split(tile, row, col, nm_softplus100(z0, nullptr));
                    nm_store_split(tile, 32 + row, col, nm_softplus100(z1, nullptr));
                } else {
                    nm_store_split(tile, row, col, fmaxf(z0, 0.f));
                    nm_store_split(tile, 32 + row, col, fmaxf(z1, 0.f));
                }
            }
        }
    }
    __syncthreads();
    nm_phase_stamp(stamp_slot + 1);
}
#undef NM_H_MFMAS

// x and its sin/cos bands for 4 consecutive feature dims, written split into the tile row.
// Odd bands come from the even band below them by the double-angle identities (3 operations
// instead of a ~28-operation sincos): sin 2t = 2 s c, cos 2t = (c - s)(c + s); measured error
// <= 3e-7 absolute (direct evaluation: 7e-8), inside this mode's accuracy class (header comment).
__device__ __forceinline__ void nm_embed4_h(_Float16* tile, int row, int col0, int dim, int bands, int chunk, float4 x) {
    const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = 4 * chunk + e;
        nm_store_split(tile, row, col0 + c, xs[e]);
        float f = 1.0f;
        for (int b = 0; b < bands; b += 2) {
            float s, co;
            nm_sincos(xs[e] * f, &s, &co);
            nm_store_split(tile, row, col0 + dim * (1 + 2 * b) + c, s);
            nm_store_split(tile, row, col0 + dim * (2 + 2 * b) + c, co);
            if (b + 1 < bands) {
                nm_store_split(tile, row, col0 + dim * (3 + 2 * b) + c, 2.0f * s * co);
                nm_store_split(tile, row, col0 + dim * (4 + 2 * b) + c, (co - s) * (co + s));
            }
            f *= 4.0f;
        }
    }
}

// ------------------------------------------------------------------ geometry MLP (split-half)
template <bool NABLA>
__global__ __launch_bounds__(NM_H_THREADS, NM_H_WAVES_PER_SIMD) void nm_geo_mlp_h_kernel(NmGeoParamsH prm, const float* __restrict__ fg_rec,
                                                              const float* __restrict__ ds, const float* __restrict__ grad,
                                                              NmRecMap rmap, long long npts, float* __restrict__ sdf_out, int P,
                                                              int stride, int off, float* __restrict__ nabla_out,
                                                              int nabla_slotted, NmSlotMap smap) {
    __shared__ __attribute__((aligned(16))) _Float16 tile[2 * NM_H_PLANE + 2 * NM_ROWS + NM_EXP_LDS_PAD];
    float* red = reinterpret_cast<float*>(tile + 2 * NM_H_PLANE);
    constexpr int PTS = NABLA ? 32 : 64;
    const long long base = (long long)blockIdx.x * PTS;
    if (smap.order && smap.order[base] == 0xffffu) return;  // no point in this tile (valid entries lead each group)
    const NmDivBase rdiv = nm_div_base(base, rmap.stride ? rmap.P : 1), odiv = nm_div_base(base, P);
    nm_phase_stamp(0);
    NmBFrag<NM_H_CT> pre0, pre1;
    nm_prefetch_b<NM_H_CT>(prm.layer[0], pre0, pre1);  // in flight during the input phase
    const int Kpad0 = prm.layer[0].Kpad;
    // All global loads of the input phase are issued first (both task rounds), the embedding work
    // follows: written in program order each round exposed two dependent memory latencies.
    constexpr int ROUNDS = PTS * 8 / NM_H_THREADS;
    float in_ds[ROUNDS];
    float4 in_fg[ROUNDS][2];
    const int nchunk = prm.gdim >> 2;  // <= 16: at most two 16-byte chunks per lane
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int task = threadIdx.x + rd * NM_H_THREADS;
        const int p = task >> 3, j = task & 7;
        in_ds[rd] = 0.f;
        in_fg[rd][0] = in_fg[rd][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (base + p < npts && nm_slot_valid(smap, base + p)) {
            const long long rq = nm_rec_index_local(rmap, rdiv, base, p);
            in_ds[rd] = ds[rq];
            if (j < nchunk) in_fg[rd][0] = *reinterpret_cast<const float4*>(fg_rec + rq * prm.gdim + 4 * j);
            if (j + 8 < nchunk) in_fg[rd][1] = *reinterpret_cast<const float4*>(fg_rec + rq * prm.gdim + 4 * (j + 8));
        }
    }
#pragma unroll 1  // (one copy of the embedding code: the round's inputs are selected, not re-inlined)
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int task = threadIdx.x + rd * NM_H_THREADS;
        const int p = task >> 3, j = task & 7;
        const long long q = base + p;
        const int sel = ROUNDS > 1 ? rd : 0;
        const float dsv = sel ? in_ds[ROUNDS - 1] : in_ds[0];
        const float4 fg0 = sel ? in_fg[ROUNDS - 1][0] : in_fg[0][0], fg1 = sel ? in_fg[ROUNDS - 1][1] : in_fg[0][1];
        if (q >= npts || !nm_slot_valid(smap, q)) {
            for (int c = j; c < Kpad0; c += 8) {
                tile[p * NM_H_STRIDE + c] = (_Float16)0.0f;
                tile[NM_H_PLANE + p * NM_H_STRIDE + c] = (_Float16)0.0f;
                if (NABLA) {
                    tile[(32 + p) * NM_H_STRIDE + c] = (_Float16)0.0f;
                    tile[NM_H_PLANE + (32 + p) * NM_H_STRIDE + c] = (_Float16)0.0f;
                }
            }
            continue;
        }
        for (int c = prm.in_dim + j; c < Kpad0; c += 8) {
            tile[p * NM_H_STRIDE + c] = (_Float16)0.0f;
            tile[NM_H_PLANE + p * NM_H_STRIDE + c] = (_Float16)0.0f;
        }
        if (j == 0) {
            nm_store_split(tile, p, 0, dsv);
            if (NABLA) nm_store_split(tile, 32 + p, 0, NM_TANGENT_SCALE);
        }
        if (NABLA)
            for (int c = prm.d_emb + j; c < Kpad0; c += 8) {
                tile[(32 + p) * NM_H_STRIDE + c] = (_Float16)0.0f;
                tile[NM_H_PLANE + (32 + p) * NM_H_STRIDE + c] = (_Float16)0.0f;
            }
        for (int b = j; b < prm.multires_d; b += 8) {
            const float f = (float)(1 << b);
            float s, co;
            nm_sincos(dsv * f, &s, &co);
            nm_store_split(tile, p, 1 + 2 * b, s);
            nm_store_split(tile, p, 2 + 2 * b, co);
            if (NABLA) {
                nm_store_split(tile, 32 + p, 1 + 2 * b, (NM_TANGENT_SCALE * f) * co);
                nm_store_split(tile, 32 + p, 2 + 2 * b, -(NM_TANGENT_SCALE * f) * s);
            }
        }
        if (j < nchunk) nm_embed4_h(tile, p, prm.d_emb, prm.gdim, prm.multires_fg, j, fg0);
        if (j + 8 < nchunk) nm_embed4_h(tile, p, prm.d_emb, prm.gdim, prm.multires_fg, j + 8, fg1);
    }
    __syncthreads();
    nm_phase_stamp(1);
    for (int l = 0; l < prm.D; ++l)  // (kernel-argument loads with a uniform index: scalar)
        nm_mlp_layer_h<0, NABLA, NM_H_CT>(tile, prm.layer[l], l + 1 < prm.D, prm.layer[l + 1 < prm.D ? l + 1 : l], pre0, pre1, 2 + 2 * l);
    if (threadIdx.x < 256) {  // output projection: 4 lanes per row
        const int row = threadIdx.x >> 2, q4 = threadIdx.x & 3;
        float s = 0.f;
        for (int m = 0; m < 8; ++m) {  // columns 8*(q4 + 4*m) .. +8: 16-byte LDS reads, 4 lanes cover 64 B
            const int c0 = 8 * (q4 + 4 * m);
            const nm_h8 v1 = *reinterpret_cast<const nm_h8*>(tile + row * NM_H_STRIDE + c0);
            const nm_h8 v2 = *reinterpret_cast<const nm_h8*>(tile + NM_H_PLANE + row * NM_H_STRIDE + c0);
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf(fmaf((float)v2[e], 1.0f / 2048.0f, (float)v1[e]), prm.wd[c0 + e], s);
        }
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
                const float dsdf = red[32 + threadIdx.x] * (1.0f / NM_TANGENT_SCALE);
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

// ------------------------------------------------------------------ colour MLP (split-half)
__global__ __launch_bounds__(NM_H_THREADS, NM_H_WAVES_PER_SIMD) void nm_col_mlp_h_kernel(NmColParamsH prm, const float* __restrict__ ft_rec,
                                                              const float* __restrict__ ds, const float* __restrict__ nabla,
                                                              const float* __restrict__ dirs, int dir_div, long long npts,
                                                              float* __restrict__ rgb_out, NmSlotMap smap) {
    __shared__ __attribute__((aligned(16))) _Float16 tile[2 * NM_H_PLANE + 6 * NM_ROWS + NM_EXP_LDS_PAD];
    float* red = reinterpret_cast<float*>(tile + 2 * NM_H_PLANE);
    const long long base = (long long)blockIdx.x * NM_ROWS;
    if (smap.order && smap.order[base] == 0xffffu) return;  // no point in this tile
    const NmDivBase ddiv = nm_div_base(base, dir_div);
    const long long ray0 = smap.order ? (base / smap.E) * smap.G : 0;  // uniform: one division per workgroup
    nm_phase_stamp(0);
    NmBFrag<NM_H_CT> pre0, pre1;
    nm_prefetch_b<NM_H_CT>(prm.layer[0], pre0, pre1);  // in flight during the input phase
    const int Kpad0 = prm.layer[0].Kpad;
    const int o_d = prm.use_nabla ? 3 : 0;
    const int o_v = o_d + prm.d_emb;
    const int o_f = o_v + 3 * (1 + 2 * prm.multires_view);
    constexpr int ROUNDS = NM_ROWS * 8 / NM_H_THREADS;  // global loads of both task rounds first (see the geometry kernel)
    float in_ds[ROUNDS], in_nb[ROUNDS][3], in_dv[ROUNDS][3];
    float4 in_ft[ROUNDS][2];
    const int nchunk = prm.cdim >> 2;
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int task = threadIdx.x + rd * NM_H_THREADS;
        const int p = task >> 3, j = task & 7;
        const long long q = base + p;
        in_ds[rd] = 0.f;
        in_nb[rd][0] = in_nb[rd][1] = in_nb[rd][2] = 0.f;
        in_dv[rd][0] = in_dv[rd][1] = in_dv[rd][2] = 0.f;
        in_ft[rd][0] = in_ft[rd][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < npts && nm_slot_valid(smap, q)) {
            in_ds[rd] = ds[q];
            if (j == 0 && prm.use_nabla) {
                in_nb[rd][0] = nabla[q * 3 + 0];
                in_nb[rd][1] = nabla[q * 3 + 1];
                in_nb[rd][2] = nabla[q * 3 + 2];
            }
            long long ray;
            int unused_p;
            if (smap.order) nm_slot_ray(smap, q, ray0, ray, unused_p);
            else nm_div_local(ddiv, p, ray, unused_p);
            in_dv[rd][0] = dirs[ray * 3 + 0];
            in_dv[rd][1] = dirs[ray * 3 + 1];
            in_dv[rd][2] = dirs[ray * 3 + 2];
            if (j < nchunk) in_ft[rd][0] = *reinterpret_cast<const float4*>(ft_rec + q * prm.cdim + 4 * j);
            if (j + 8 < nchunk) in_ft[rd][1] = *reinterpret_cast<const float4*>(ft_rec + q * prm.cdim + 4 * (j + 8));
        }
    }
#pragma unroll 1
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int task = threadIdx.x + rd * NM_H_THREADS;
        const int p = task >> 3, j = task & 7;
        const long long q = base + p;
        const int sel = ROUNDS > 1 ? rd : 0;
        const float dsv = sel ? in_ds[ROUNDS - 1] : in_ds[0];
        const float nb[3] = {sel ? in_nb[ROUNDS - 1][0] : in_nb[0][0], sel ? in_nb[ROUNDS - 1][1] : in_nb[0][1], sel ? in_nb[ROUNDS - 1][2] : in_nb[0][2]};
        const float dv[3] = {sel ? in_dv[ROUNDS - 1][0] : in_dv[0][0], sel ? in_dv[ROUNDS - 1][1] : in_dv[0][1], sel ? in_dv[ROUNDS - 1][2] : in_dv[0][2]};
        const float4 ft0 = sel ? in_ft[ROUNDS - 1][0] : in_ft[0][0], ft1 = sel ? in_ft[ROUNDS - 1][1] : in_ft[0][1];
        if (q >= npts || !nm_slot_valid(smap, q)) {
            for (int c = j; c < Kpad0; c += 8) {
                tile[p * NM_H_STRIDE + c] = (_Float16)0.0f;
                tile[NM_H_PLANE + p * NM_H_STRIDE + c] = (_Float16)0.0f;
            }
            continue;
        }
        for (int c = prm.in_dim + j; c < Kpad0; c += 8) {
            tile[p * NM_H_STRIDE + c] = (_Float16)0.0f;
            tile[NM_H_PLANE + p * NM_H_STRIDE + c] = (_Float16)0.0f;
        }
        if (j == 0) {
            nm_store_split(tile, p, o_d, dsv);
            if (prm.use_nabla) {
                nm_store_split(tile, p, 0, nb[0]);
                nm_store_split(tile, p, 1, nb[1]);
                nm_store_split(tile, p, 2, nb[2]);
            }
        }
        for (int b = j; b < prm.multires_d; b += 8) {
            float s, co;
            nm_sincos(dsv * (float)(1 << b), &s, &co);
            nm_store_split(tile, p, o_d + 1 + 2 * b, s);
            nm_store_split(tile, p, o_d + 2 + 2 * b, co);
        }
        {
            if (j == 1) {
                nm_store_split(tile, p, o_v, dv[0]);
                nm_store_split(tile, p, o_v + 1, dv[1]);
                nm_store_split(tile, p, o_v + 2, dv[2]);
            }
            for (int e = j; e < 3 * prm.multires_view; e += 8) {
                const int b = e / 3, dim = e - 3 * b;
                float s, co;
                nm_sincos((dim == 0 ? dv[0] : dim == 1 ? dv[1] : dv[2]) * (float)(1 << b), &s, &co);
                nm_store_split(tile, p, o_v + 3 + 6 * b + dim, s);
                nm_store_split(tile, p, o_v + 6 + 6 * b + dim, co);
            }
        }
        if (j < nchunk) nm_embed4_h(tile, p, o_f, prm.cdim, prm.multires_ft, j, ft0);
        if (j + 8 < nchunk) nm_embed4_h(tile, p, o_f, prm.cdim, prm.multires_ft, j + 8, ft1);
    }
    __syncthreads();
    nm_phase_stamp(1);
    for (int l = 0; l < prm.D; ++l)
        nm_mlp_layer_h<1, false, NM_H_CT>(tile, prm.layer[l], l + 1 < prm.D, prm.layer[l + 1 < prm.D ? l + 1 : l], pre0, pre1, 2 + 2 * l);
    if (threadIdx.x < 256) {  // output projection: 4 lanes per row
        const int row = threadIdx.x >> 2, q4 = threadIdx.x & 3;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int m = 0; m < 8; ++m) {
            const int c0 = 8 * (q4 + 4 * m);
            const nm_h8 v1 = *reinterpret_cast<const nm_h8*>(tile + row * NM_H_STRIDE + c0);
            const nm_h8 v2 = *reinterpret_cast<const nm_h8*>(tile + NM_H_PLANE + row * NM_H_STRIDE + c0);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float av = fmaf((float)v2[e], 1.0f / 2048.0f, (float)v1[e]);
                s0 = fmaf(av, prm.wrgb[c0 + e], s0);
                s1 = fmaf(av, prm.wrgb[256 + c0 + e], s1);
                s2 = fmaf(av, prm.wrgb[512 + c0 + e], s2);
            }
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
