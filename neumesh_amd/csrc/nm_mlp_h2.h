// nm_mlp_h2.h -- fused embed + MLP kernels on the f16 matrix pipe (v_mfma_f32_32x32x16_f16).
//
// Why not the fp32 MFMA: on gfx950 the fp32-input MFMA executes at the vector-FMA rate and does not overlap
// VALU work (measured: 91-124 TFLOP/s of the 157), the f16 MFMA has 16x the rate and its own pipe.
//
// Split-half operands (NP = 3 products, the default, mlp_precision 2): every fp32 value a is carried as two
// halves  a = h1 + h2 * 2^-11,  h1 = rne16(a), h2 = rne16((a - h1) * 2^11)  -- 22 significant bits, error
// <= 2^-22 |a| (fp32: 2^-24) -- and a product as  a*b = h1a*h1b + 2^-11 (h1a*h2b + h2a*h1b)  (+ h2a*h2b*2^-22,
// dropped): 3 f16 MFMAs with fp32 accumulation in two accumulators (main / 2^11-scaled), recombined in the
// epilogue.  The two halves occupy exactly the 4 bytes of the fp32 value, so the LDS tile (64 rows x 256
// columns) and the packed weights keep their size; activations are split ONCE, by the epilogue that produces them.
// Range: |values| must stay below 65504 (fp16); fp16 subnormals are exact on the matrix pipe (tools/mfma_denorm.hip).
//
// One accumulator (NP = 6, mlp_precision 6): the same three products with the residual halves stored UNSCALED,
// h2 = rne16(a - h1) (fp16 subnormals, quantum 2^-24, are exact on the matrix pipe), so that all three accumulate into ONE
// fp32 accumulator: 64 instead of 128 accumulator registers per wave, no recombination and no 2^11 scaling in the epilogue
// (5.5 instead of 7 vector instructions per activation).  What it gives up: an operand keeps an ABSOLUTE precision of 2^-25
// instead of a relative 2^-22, so small operands lose bits -- harmless for the value rows (tools/mlp_error_budget.py: sdf error
// 6e-8 against 5e-8), but the tangent rows must enter at 2^-8 instead of 2^-15 to stay above it (d sdf / d ds error 2e-6
// against 5e-7; gate 5e-6).
//
// Single product (NP = 1, mlp_precision 4, never the default): only h1a*h1b -- plain fp16 operands (11 significant
// bits) with fp32 accumulation, one MFMA per product.  This is the "bf16 MLP"-class mode BASELINE configs[1] names
// (fp16 keeps 3 more mantissa bits than bf16 at the same matrix rate); it does NOT meet the 1e-4 RGB bound and is
// reported with its measured error beside the default (bench.py `f16_single`).
//
// Data flow (v2; what the phase stamps of the first layout showed -- matrix pipe 46 % busy, the rest in 2-byte LDS
// stores, a 14 k-cycle input phase and a separate output projection):
//  * The MFMA operands are swapped: D = W_tile (32 output columns x 16 k) * A^T (16 k x 32 points).
//    The 32x32 result layout then gives a lane ONE point and 16 output columns, and the weight rows
//    of a column tile are packed in the order that makes those 16 columns CONSECUTIVE
//    (MFMA row i = 4h + 8g + e  <->  column 16h + 4g + e of the tile).  The epilogue therefore
//    writes its activations with 16-byte LDS stores (4 per tile and plane-pair instead of 64
//    2-byte stores) and converts them in pairs.
//  * Layer-0 input columns are PERMUTED (the permutation is folded into the packed weights):
//    [code embedding | sin/cos pairs of ds | (view bands, view, nabla) | ds].  Every lane of the
//    input phase then owns 4 consecutive columns per embedding band (8-byte stores), the tangent
//    rows of the geometry MLP are non-zero only in the last k-steps of layer 0 (the others are
//    skipped for that row tile), and no column needs an odd-address store except the last few.
//  * The output projection (256 -> 1 / 3) is fused into the last hidden layer's epilogue: each
//    lane multiplies its 32 fp32 activations by the head weights straight from the accumulators,
//    a half-wave exchange + a 4-wave LDS reduction finish it.  The last layer's activations are
//    never split or stored, and the head sees full fp32 activations.
//  * Biases and head weights sit in LDS (one coalesced copy per workgroup).
//  * The input phase of the reference configuration (32-d codes, 2 / 8 / 4 embedding bands) is
//    compiled with constant trip counts (FIXED); other configurations take the same code with the
//    counts as run-time values.
//  * Every value that is split is also tracked for fp16 range: a lane that sees |v| >= 65504 raises
//    a flag in the field handle (nm_field_overflow), so a checkpoint that does not fit the
//    split-half format is detected instead of rendering garbage.
//
// Reference semantics: models/frameworks/neumesh/neumesh.py:204-260, models/base.py:52-70 (see nm_mlp.h).
#pragma once

#include "nm_mlp.h"

typedef _Float16 nm_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 nm_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 nm_h2 __attribute__((ext_vector_type(2)));

#define NM_H_STRIDE 264  // halves per tile row: 528 B = 33 16-byte slots -> conflict-free ds_read_b128
#define NM_H_PLANE (NM_ROWS * NM_H_STRIDE)

struct NmLayerH {
    const _Float16* W;  // packed fragments: [col tile 8][k-step Kpad/16][plane 2][lane 64][8 halves]
    const float* b;     // [256] fp32
    int Kpad;
};

// a -> (h1, h2): a ~= h1 + h2 / 2048.  fp16 subnormals are fine on both sides: v_cvt_f16_f32 produces
// them and the matrix pipe consumes them exactly (probe: tools/mfma_denorm.hip on gfx950).
__device__ __forceinline__ void nm_split_half(float a, _Float16* h1, _Float16* h2) {
    const _Float16 p = (_Float16)a;
    *h1 = p;
    *h2 = (_Float16)((a - (float)p) * 2048.0f);
}

// Work split inside a workgroup (64 activation rows x 256 output columns per layer): a wave owns all
// 64 rows of CT 32-column tiles, so there are 8/CT waves.  CT = 2: 256 threads, two workgroups per CU
// = 2 waves per SIMD.  CT = 1 (512 threads, 4 waves per SIMD, half the accumulators per wave) was
// measured slower (every wave re-reads the whole A tile from LDS, four waves share one matrix pipe).
#define NM_H_CT 2
#define NM_H_THREADS (64 * 8 / NM_H_CT)
#define NM_H_WAVES_PER_SIMD (4 / NM_H_CT)  // two workgroups per CU (LDS: 2 x 68 KiB)

// B operand (weights) of one k-step for the CT column tiles of a wave: 2*CT x 16 bytes per lane.
template <int CT>
struct NmBFrag {
    nm_h8 a[CT], b[CT];  // plane h1, plane h2
};
template <int CT>
struct NmAccH {  // 2 row tiles x CT column tiles of a wave, main and 2^11-scaled accumulators
    nm_f32x16 hi[2][CT], lo[2][CT];
};

#define NM_H2_BIAS_LAYERS 4  // biases of the first 4 layers live in LDS (deeper layers read them from L2)
#define NM_H2_FP16_MAX 65504.0f

// physical input column k of layer 0 -> logical column of the reference's concatenation:
// consecutive segments, segment s = physical [sum len[<s], +len[s]) = logical [src[s], +len[s])
struct NmColSeg {
    int n;
    int len[6], src[6];
};

// weights: fp32 [256][in_dim] (PyTorch layout, logical columns) -> split halves in MFMA A-operand
// fragment order [column tile 8][k-step Kpad/16][plane 2][lane 64][8 halves]; lane = (row i of the
// tile) | (k-half << 5) with row i = 4h + 8g + e holding output column 16h + 4g + e (header comment).
__global__ void nm_pack_weight_h2_kernel(const float* __restrict__ src, int in_dim, int Kpad, NmColSeg seg, float scale, float res_scale, _Float16* __restrict__ dst) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (n, k)
    if (e >= NM_W * Kpad) return;
    const int n = e / Kpad, k = e - n * Kpad;
    int kl = -1, base = 0;
    for (int s = 0; s < seg.n; ++s) {
        if (kl < 0 && k >= base && k < base + seg.len[s]) kl = seg.src[s] + (k - base);
        base += seg.len[s];
    }
    const float w = (kl >= 0 && kl < in_dim) ? src[(size_t)n * in_dim + kl] * scale : 0.f;
    const _Float16 h1 = (_Float16)w;
    const _Float16 h2 = (_Float16)((w - (float)h1) * res_scale);   // res_scale: 2048 (two accumulators) or 1 (one accumulator)
    const int ct = n >> 5, c = n & 31;
    const int i = 4 * (c >> 4) + 8 * ((c >> 2) & 3) + (c & 3);
    const int lane = i | (((k >> 3) & 1) << 5), ks = k >> 4, el = k & 7;
    const int KS = Kpad >> 4;
    const size_t o = ((size_t)(ct * KS + ks) * 2) * 64 * 8;
    dst[o + (size_t)lane * 8 + el] = h1;
    dst[o + 64 * 8 + (size_t)lane * 8 + el] = h2;
}

__global__ void nm_scale_copy_kernel(const float* __restrict__ src, float scale, int n, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] * scale;
}

struct NmGeoParamsH2 {
    NmLayerH layer[NM_MAX_LAYERS];  // log2 units (nm_softplus_l2): layer 0 weights and every bias x S
    int D;
    const float* wd;  // [256], x 1/S
    float bd;
    int multires_d, multires_fg, gdim;
    int fg_w, in_dim;  // fg_w = gdim*(1+2*multires_fg): width of the code-embedding block = first column of the ds block
};

struct NmColParamsH2 {
    NmLayerH layer[NM_MAX_LAYERS];
    int D;
    const float* wrgb;  // [3][256]
    float brgb[3];
    int multires_d, multires_ft, multires_view, cdim, use_nabla;
    int ft_w, in_dim;   // ft_w = cdim*(1+2*multires_ft)
};

// ------------------------------------------------------------------------- split + store helpers
// a -> (h1, h2), a ~= h1 + h2 / 2048 as nm_split_half, with the residual formed by ONE mixed-precision fma
// on the f16 value: (a - h1) * 2048 == fma(h1, -2048, a * 2048) exactly (every term is exact in fp32).
template <bool US = false>
__device__ __forceinline__ void nm_h2_split(float a, _Float16& h1, _Float16& h2) {
    h1 = (_Float16)a;
    h2 = US ? (_Float16)(a - (float)h1) : (_Float16)fmaf((float)h1, -2048.0f, a * 2048.0f);
}
// Two values at once, as packed pairs (low half = a): h1 = one v_cvt_pk_f16_f32, the residuals by v_fma_mix{lo,hi}_f16,
// which read their f16 operand straight from the packed pair and round the fp32 fma result to f16 once -- the same
// arithmetic as nm_h2_split, 4 instructions per pair instead of 6 (the compiler only finds the mix form now and then).
template <bool US = false>
__device__ __forceinline__ void nm_h2_split2(float a, float b, unsigned& p1, unsigned& p2) {
    p1 = __builtin_bit_cast(unsigned, nm_h2{(_Float16)a, (_Float16)b});
    const float a2 = US ? a : a * 2048.0f, b2 = US ? b : b * 2048.0f, m = US ? -1.0f : -2048.0f;   // (US: the residual as it is, one fma_mix per value)
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p1), "s"(m), "v"(a2));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(p1), "s"(m), "v"(b2));
    p2 = r;
}
// (mx: running max of |value| for the fp16-range check; pairs go through one max3)
template <int PLANE = NM_H_PLANE, bool US = false>
__device__ __forceinline__ void nm_h2_store1(_Float16* p, float a, float& mx) {
    _Float16 h1, h2;
    nm_h2_split<US>(a, h1, h2);
    p[0] = h1;
    p[PLANE] = h2;
    mx = fmaxf(mx, fabsf(a));
}
template <int PLANE = NM_H_PLANE, bool US = false>
__device__ __forceinline__ void nm_h2_store2(_Float16* p, float a, float b, float& mx) {  // p 4-byte aligned
    unsigned p1, p2;
    nm_h2_split2<US>(a, b, p1, p2);
    *reinterpret_cast<unsigned*>(p) = p1;
    *reinterpret_cast<unsigned*>(p + PLANE) = p2;
    mx = fmaxf(mx, fmaxf(fabsf(a), fabsf(b)));
}
template <int PLANE = NM_H_PLANE, bool US = false>
__device__ __forceinline__ void nm_h2_store4(_Float16* p, const float (&v)[4], float& mx) {  // p 8-byte aligned
    uint2 a, b;
    nm_h2_split2<US>(v[0], v[1], a.x, b.x);
    nm_h2_split2<US>(v[2], v[3], a.y, b.y);
    mx = fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1])));
    mx = fmaxf(mx, fmaxf(fabsf(v[2]), fabsf(v[3])));
    *reinterpret_cast<uint2*>(p) = a;
    *reinterpret_cast<uint2*>(p + PLANE) = b;
}
template <int PLANE = NM_H_PLANE, bool US = false>
__device__ __forceinline__ void nm_h2_store8(_Float16* p, const float (&v)[8], float& mx) {  // p 16-byte aligned
    uint4 a, b;
    nm_h2_split2<US>(v[0], v[1], a.x, b.x);
    nm_h2_split2<US>(v[2], v[3], a.y, b.y);
    nm_h2_split2<US>(v[4], v[5], a.z, b.z);
    nm_h2_split2<US>(v[6], v[7], a.w, b.w);
#pragma unroll
    for (int e = 0; e < 8; e += 2) mx = fmaxf(fmaxf(mx, fabsf(v[e])), fabsf(v[e + 1]));
    *reinterpret_cast<uint4*>(p) = a;
    *reinterpret_cast<uint4*>(p + PLANE) = b;
}
__device__ __forceinline__ void nm_h1_store8(_Float16* p, const float (&v)[8], float& mx) {  // main plane only (single-product mode)
    uint4 a;
    a.x = __builtin_bit_cast(unsigned, nm_h2{(_Float16)v[0], (_Float16)v[1]});
    a.y = __builtin_bit_cast(unsigned, nm_h2{(_Float16)v[2], (_Float16)v[3]});
    a.z = __builtin_bit_cast(unsigned, nm_h2{(_Float16)v[4], (_Float16)v[5]});
    a.w = __builtin_bit_cast(unsigned, nm_h2{(_Float16)v[6], (_Float16)v[7]});
#pragma unroll
    for (int e = 0; e < 8; e += 2) mx = fmaxf(fmaxf(mx, fabsf(v[e])), fabsf(v[e + 1]));
    *reinterpret_cast<uint4*>(p) = a;
}
// zero columns [c0, c1) of a tile row, both planes (c0 a multiple of 8: 16-byte stores, then singles)
template <int PLANE = NM_H_PLANE>
__device__ __forceinline__ void nm_h2_zero_cols(_Float16* row, int c0, int c1, int j) {
    const nm_h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    const int full = (c1 - c0) >> 3;
    for (int b = j; b < full; b += 8) {
        *reinterpret_cast<nm_h8*>(row + c0 + 8 * b) = z;
        *reinterpret_cast<nm_h8*>(row + PLANE + c0 + 8 * b) = z;
    }
    for (int c = c0 + 8 * full + j; c < c1; c += 8) {
        row[c] = (_Float16)0.0f;
        row[PLANE + c] = (_Float16)0.0f;
    }
}

// x and its sin/cos bands for 4 consecutive dims of a `dim`-wide code vector, into an embedding block
// that starts at blk[0]: 8-byte stores (dim is a multiple of 4).  Odd bands from the even band below
// by the double-angle identities, as nm_embed4_h.
template <bool FAST, int PLANE = NM_H_PLANE, bool US = false>
__device__ __forceinline__ void nm_h2_embed_chunk_t(_Float16* blk, int dim, int bands, int chunk, const float (&xs)[4], float& mx) {
    nm_h2_store4<PLANE, US>(blk + 4 * chunk, xs, mx);
    float f = 1.0f;
    for (int b = 0; b < bands; b += 2) {
        float s[4], c[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (FAST) nm_sincos_fast(xs[e] * f, &s[e], &c[e]);
            else nm_sincos(xs[e] * f, &s[e], &c[e]);
        }
        nm_h2_store4<PLANE, US>(blk + dim * (1 + 2 * b) + 4 * chunk, s, mx);
        nm_h2_store4<PLANE, US>(blk + dim * (2 + 2 * b) + 4 * chunk, c, mx);
        if (b + 1 < bands) {
            float s2[4], c2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s2[e] = 2.0f * s[e] * c[e];
                c2[e] = (c[e] - s[e]) * (c[e] + s[e]);
            }
            nm_h2_store4<PLANE, US>(blk + dim * (3 + 2 * b) + 4 * chunk, s2, mx);
            nm_h2_store4<PLANE, US>(blk + dim * (4 + 2 * b) + 4 * chunk, c2, mx);
        }
        f *= 4.0f;
    }
}
// One range test per chunk instead of one per sincos: the straight-line fast path (every argument within the polynomial
// reduction's range -- always, for trained codes) lets the four evaluations of a band overlap; same values either way.
template <int PLANE = NM_H_PLANE, bool US = false>
__device__ __forceinline__ void nm_h2_embed_chunk(_Float16* blk, int dim, int bands, int chunk, float4 x, float& mx) {
    const float xs[4] = {x.x, x.y, x.z, x.w};
    const float top = fmaxf(fmaxf(fabsf(xs[0]), fabsf(xs[1])), fmaxf(fabsf(xs[2]), fabsf(xs[3]))) * (float)(1 << ((bands > 0 ? bands - 1 : 0) & ~1));  // (largest frequency evaluated directly: the highest even band)
    if (top <= NM_SINCOS_FAST_MAX) nm_h2_embed_chunk_t<true, PLANE, US>(blk, dim, bands, chunk, xs, mx);
    else nm_h2_embed_chunk_t<false, PLANE, US>(blk, dim, bands, chunk, xs, mx);
}

// Softplus(beta = 100) in "log2 units".  With S = 100 / ln 2 the reference's y = softplus(z) = log2(1 + 2^(S z)) / S, so
// carrying m = S y between the geometry layers instead of y makes the activation m = log2(1 + 2^z') of the
// pre-activation z' = S z, and the next layer's z' = S (W y + b) = W m + S b needs NO rescaling of its weights:
// only layer 0's weights (x S), every bias (x S) and the density head (x 1/S) are scaled, once, when the field is
// packed (nm_field_pack).  Two multiplies per element less than softplus100 on unscaled values, and the exponent
// argument comes straight from the accumulator.  Clamp, threshold behaviour and gradient as nm_softplus100
// (30.2965958 = 21 log2 e <-> 100 z = 21); d m / d z' = d y / d z = 2^z' / (1 + 2^z'), so tangent rows carry S dy
// the same way and the scaled head returns d sdf.
#define NM_H2_S 144.26950408889634f
__device__ __forceinline__ float nm_softplus_l2(float zp, float* grad) {
    const float e = __builtin_amdgcn_exp2f(fminf(zp, 30.2965958f));
    const float u = 1.0f + e;
    if (grad) *grad = e * __builtin_amdgcn_rcpf(u);
    return fmaxf(zp, __builtin_amdgcn_logf(u));
}
// tangent rows enter layer 0 scaled by 2^-15 (~ 2^-8 / S: the same fp16 head room as the unscaled kernels' 2^-8)
#define NM_H2_TANGENT_SCALE 3.0517578125e-05f
// one-accumulator mode: 2^-8 -- the unscaled residual halves resolve 2^-25 absolute, so the tangent operands must not be small
// (largest tangent operand on the fixture scenes at this scale: 3; fp16 range 65504)
#define NM_H2_TANGENT_SCALE_1ACC 3.90625e-03f
template <int NP>
__device__ __forceinline__ constexpr float nm_h2_tscale() { return NP == 6 ? NM_H2_TANGENT_SCALE_1ACC : NM_H2_TANGENT_SCALE; }

// --------------------------------------------------------------------------------- K loop
// As nm_kloop_h (B fragments two k-steps ahead in three rotating register sets, A fragments one step
// ahead, fully unrolled, sched_barriers pinning the issue points) with the operands swapped -- the
// weight fragment is the MFMA's A operand, the activation fragment its B operand -- and KT0: the first
// k-step in which row tile 1 takes part (layer 0 of the tangent kernel: its tangent rows are zero before).
#define NM_H2_MFMAS(A, F, R1)                                                                             \
    _Pragma("unroll") for (int rt_ = 0; rt_ < (R1); ++rt_) _Pragma("unroll") for (int c_ = 0; c_ < CT; ++c_)  \
        c.hi[rt_][c_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a[c_], A[rt_][0], c.hi[rt_][c_], 0, 0, 0);   \
    if (NP == 3) {                                                                                                \
    _Pragma("unroll") for (int rt_ = 0; rt_ < (R1); ++rt_) _Pragma("unroll") for (int c_ = 0; c_ < CT; ++c_)  \
        c.lo[rt_][c_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.b[c_], A[rt_][0], c.lo[rt_][c_], 0, 0, 0);   \
    _Pragma("unroll") for (int rt_ = 0; rt_ < (R1); ++rt_) _Pragma("unroll") for (int c_ = 0; c_ < CT; ++c_)  \
        c.lo[rt_][c_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a[c_], A[rt_][1], c.lo[rt_][c_], 0, 0, 0);   \
    }                                                                                                             \
    if (NP == 6) { /* one accumulator: the cross products (unscaled residual halves) go where the main product went */ \
    _Pragma("unroll") for (int rt_ = 0; rt_ < (R1); ++rt_) _Pragma("unroll") for (int c_ = 0; c_ < CT; ++c_)  \
        c.hi[rt_][c_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.b[c_], A[rt_][0], c.hi[rt_][c_], 0, 0, 0);   \
    _Pragma("unroll") for (int rt_ = 0; rt_ < (R1); ++rt_) _Pragma("unroll") for (int c_ = 0; c_ < CT; ++c_)  \
        c.hi[rt_][c_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a[c_], A[rt_][1], c.hi[rt_][c_], 0, 0, 0);   \
    }

// B (weight) fragments are fetched with raw buffer loads: the (layer, column tile) block is a buffer
// resource in 4 SGPRs, the lane contributes a constant 32-bit byte offset (lane * 16) and the k-step /
// plane an immediate or scalar offset.  No per-k-step 64-bit VGPR addresses exist, so the unrolled K
// loops cannot spill them (global_load forms did: scratch stores between the loads of layer 0, which the
// in-order vmcnt waits of the loop then had to sit out).
typedef __amdgpu_buffer_rsrc_t nm_rsrc;
__device__ __forceinline__ nm_rsrc nm_b_rsrc(const _Float16* W, int Kpad, int ctile_uniform) {
    const _Float16* p = W + (size_t)ctile_uniform * (Kpad >> 4) * 2 * 64 * 8;
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (Kpad >> 4) * 2048, 0x00020000);
}
template <int CT, int NP>
__device__ __forceinline__ NmBFrag<CT> nm_ld_bu(const nm_rsrc (&ub)[CT], int lane, int ks) {
    NmBFrag<CT> f;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        f.a[c] = __builtin_bit_cast(nm_h8, __builtin_amdgcn_raw_buffer_load_b128(ub[c], lane * 16, ks * 2048, 0));
        if (NP != 1) f.b[c] = __builtin_bit_cast(nm_h8, __builtin_amdgcn_raw_buffer_load_b128(ub[c], lane * 16, ks * 2048 + 1024, 0));
        else f.b[c] = nm_h8{0, 0, 0, 0, 0, 0, 0, 0};   // (single-product mode: the residual plane is never read)
    }
    return f;
}
#define NM_H2_PRE 2  // B-fragment sets a layer receives pre-loaded (= the prefetch distance of the K loops)
template <int CT>
struct NmBPre {
    NmBFrag<CT> s[NM_H2_PRE];
};
// DEPTH: how many k-steps ahead the B fragments are requested (DEPTH + 1 rotating register sets, the first
// DEPTH arrive pre-loaded in `pre`): 2 everywhere.  (A distance of 4 in layer 0 of the tangent kernel, whose
// row-tile-1 accumulators only start their life at k-step KT0, was measured slower.)
template <int KS, int CT, int KT0, int DEPTH, int NP>
__device__ __forceinline__ void nm_kloop_h2(const _Float16* a0p, const _Float16* a1p, const nm_rsrc (&bp)[CT], const int lane,
                                            const NmBPre<CT>& pre, NmAccH<CT>& c) {
    NmBFrag<CT> f[DEPTH + 1];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) f[i] = pre.s[i];
    nm_h8 a[2][2][2];  // [buffer][row tile][plane]
    a[0][0][0] = *reinterpret_cast<const nm_h8*>(a0p);
    if (NP != 1) a[0][0][1] = *reinterpret_cast<const nm_h8*>(a0p + NM_H_PLANE);
    if (KT0 == 0) {
        a[0][1][0] = *reinterpret_cast<const nm_h8*>(a1p);
        if (NP != 1) a[0][1][1] = *reinterpret_cast<const nm_h8*>(a1p + NM_H_PLANE);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        __builtin_amdgcn_sched_barrier(0);
        if (KT0 > 0 && ks == KT0) {  // row tile 1 joins here: its accumulators start their life now
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) c.hi[1][ct] = c.lo[1][ct] = nm_f32x16{0};
        }
        if (ks + DEPTH < KS) f[(ks + DEPTH) % (DEPTH + 1)] = nm_ld_bu<CT, NP>(bp, lane, ks + DEPTH);
        if (ks + 1 < KS) {
            const int oa = (ks + 1) * 16;
            a[(ks + 1) & 1][0][0] = *reinterpret_cast<const nm_h8*>(a0p + oa);
            if (NP != 1) a[(ks + 1) & 1][0][1] = *reinterpret_cast<const nm_h8*>(a0p + NM_H_PLANE + oa);
            if (ks + 1 >= KT0) {
                a[(ks + 1) & 1][1][0] = *reinterpret_cast<const nm_h8*>(a1p + oa);
                if (NP != 1) a[(ks + 1) & 1][1][1] = *reinterpret_cast<const nm_h8*>(a1p + NM_H_PLANE + oa);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ks >= KT0) {
            NM_H2_MFMAS(a[ks & 1], f[ks % (DEPTH + 1)], 2)
        } else {
            NM_H2_MFMAS(a[ks & 1], f[ks % (DEPTH + 1)], 1)
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int CT, int NP>
__device__ __forceinline__ void nm_prefetch_bn(const NmLayerH L, NmBPre<CT>& pre, int n) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    nm_rsrc bp[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) bp[c] = nm_b_rsrc(L.W, L.Kpad, wave * CT + c);
#pragma unroll
    for (int i = 0; i < NM_H2_PRE; ++i)
        if (i < n) pre.s[i] = nm_ld_bu<CT, NP>(bp, lane, i);  // (every layer has >= 2 k-steps; layer 0 of any supported configuration >= 4)
}

// any other layer-0 width (run-time k-step counts): rolled loops, fragments one step ahead; k-steps
// [0, kt0) without row tile 1, then [kt0, KS) with it
template <int CT, int NP>
__device__ __forceinline__ void nm_kloop_h2_generic(int KS, int kt0, const _Float16* a0p, const _Float16* a1p,
                                                    const nm_rsrc (&bp)[CT], const int lane, const NmBPre<CT>& pre, NmAccH<CT>& c) {
    NmBFrag<CT> nf = pre.s[0];
    nm_h8 na[2][2];
    na[0][0] = *reinterpret_cast<const nm_h8*>(a0p);
    na[0][1] = *reinterpret_cast<const nm_h8*>(a0p + NM_H_PLANE);
    na[1][0] = *reinterpret_cast<const nm_h8*>(a1p);
    na[1][1] = *reinterpret_cast<const nm_h8*>(a1p + NM_H_PLANE);
    int ks = 0;
    for (; ks < kt0; ++ks) {
        const NmBFrag<CT> F = nf;
        nm_h8 A[2][2];
        A[0][0] = na[0][0]; A[0][1] = na[0][1]; A[1][0] = na[1][0]; A[1][1] = na[1][1];
        if (ks + 1 < KS) {
            nf = nm_ld_bu<CT, NP>(bp, lane, ks + 1);
            const int oa = (ks + 1) * 16;
            na[0][0] = *reinterpret_cast<const nm_h8*>(a0p + oa);
            na[0][1] = *reinterpret_cast<const nm_h8*>(a0p + NM_H_PLANE + oa);
            na[1][0] = *reinterpret_cast<const nm_h8*>(a1p + oa);
            na[1][1] = *reinterpret_cast<const nm_h8*>(a1p + NM_H_PLANE + oa);
        }
        NM_H2_MFMAS(A, F, 1)
    }
    for (; ks < KS; ++ks) {
        const NmBFrag<CT> F = nf;
        nm_h8 A[2][2];
        A[0][0] = na[0][0]; A[0][1] = na[0][1]; A[1][0] = na[1][0]; A[1][1] = na[1][1];
        if (ks + 1 < KS) {
            nf = nm_ld_bu<CT, NP>(bp, lane, ks + 1);
            const int oa = (ks + 1) * 16;
            na[0][0] = *reinterpret_cast<const nm_h8*>(a0p + oa);
            na[0][1] = *reinterpret_cast<const nm_h8*>(a0p + NM_H_PLANE + oa);
            na[1][0] = *reinterpret_cast<const nm_h8*>(a1p + oa);
            na[1][1] = *reinterpret_cast<const nm_h8*>(a1p + NM_H_PLANE + oa);
        }
        NM_H2_MFMAS(A, F, 2)
    }
}
#undef NM_H2_MFMAS

// One dense layer on the split-half LDS tile.  Not the last hidden layer: activations go back into the
// tile (in place).  LAST: the NOUT-wide head (density / rgb) is applied to the fp32 activations in
// registers and the per-row sums land in red[wave][row][NOUT]; the tile is not written.
//   cst / bias_row: this layer's bias = LDS row cst[bias_row][256], or bias_row < 0 -> L.b from global;
//   head_w: LDS [NOUT][256].
//   KSF / KT0F: compile-time k-step count of the layer and first k-step with non-zero row-tile-1 operands
//   (non-zero only in layer 0 of the tangent kernel); KSF = 0: run-time counts (L.Kpad, kt0), rolled loops.
//   DEPTH: B-fragment prefetch distance of this layer's K loop = number of sets `pre` holds on entry; on exit
//   `pre` holds the first two sets of `next` (requested before the epilogue so that they arrive while it runs).
template <int ACT, bool TANGENT, bool LAST, int NOUT, int CT, int KSF, int KT0F, int DEPTH, int NP>
__device__ __forceinline__ void nm_mlp_layer_h2(_Float16* tile, const NmLayerH L, const int kt0, const bool has_next, const NmLayerH next,
                                                NmBPre<CT>& pre, const float* cst, const int bias_row, const float* head_w,
                                                float* red, float& mx, int stamp_slot) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, h = lane >> 5;
    const int n0 = wave * 32 * CT;
    const _Float16* a0p = tile + li * NM_H_STRIDE + 8 * h;
    const _Float16* a1p = tile + (32 + li) * NM_H_STRIDE + 8 * h;
    nm_rsrc bp[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) bp[c] = nm_b_rsrc(L.W, L.Kpad, wave * CT + c);
    NmAccH<CT> c;
    // the main accumulators of the value rows start at the bias (one add per element less in the epilogue); tangent rows at 0
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int colb = n0 + 32 * ct + 16 * h;  // this lane's 16 columns of the tile: register r <-> column colb + r
        float4 b4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (bias_row >= 0) {  // (two address spaces: LDS for the first layers, global beyond -- never a generic pointer)
                b4[q] = *reinterpret_cast<const float4*>(cst + bias_row * NM_W + colb + 4 * q);
            } else {  // (buffer loads: a different instruction class, so the two paths are never merged into flat loads)
                const nm_rsrc rb = __builtin_amdgcn_make_buffer_rsrc((void*)L.b, 0, NM_W * 4, 0x00020000);
                b4[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rb, (colb + 4 * q) * 4, 0, 0));
            }
        }
        nm_f32x16 bv;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bv[4 * q + 0] = b4[q].x; bv[4 * q + 1] = b4[q].y; bv[4 * q + 2] = b4[q].z; bv[4 * q + 3] = b4[q].w;
        }
        c.hi[0][ct] = bv;
        c.lo[0][ct] = nm_f32x16{0};
        if (!(KSF > 0 && KT0F > 0)) {  // (else: row tile 1 is initialised at k-step KT0F -- tangent rows, zero)
            c.hi[1][ct] = TANGENT ? nm_f32x16{0} : bv;
            c.lo[1][ct] = nm_f32x16{0};
        }
    }
    if (KSF > 0) nm_kloop_h2<(KSF > 0 ? KSF : 1), CT, KT0F, DEPTH, NP>(a0p, a1p, bp, lane, pre, c);  // one straight-line loop, no run-time dispatch
    else nm_kloop_h2_generic<CT, NP>(L.Kpad >> 4, kt0, a0p, a1p, bp, lane, pre, c);
    if (has_next) nm_prefetch_bn<CT, NP>(next, pre, 2);
    if (!LAST) __syncthreads();  // every wave has finished reading the input tile
    nm_phase_stamp(stamp_slot);
    const float sc = 1.0f / 2048.0f;
    float so[2][NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) so[0][o] = so[1][o] = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {  // 8 columns at a time: one 16-byte store per row and plane, short live ranges
            const int col0 = n0 + 32 * ct + 16 * h + 8 * hf;  // this lane's columns: register 8*hf + r <-> column col0 + r
            float y0[8], y1[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float z0 = NP == 3 ? fmaf(c.lo[0][ct][8 * hf + r], sc, c.hi[0][ct][8 * hf + r]) : c.hi[0][ct][8 * hf + r];  // (bias: in the accumulator)
                const float z1 = NP == 3 ? fmaf(c.lo[1][ct][8 * hf + r], sc, c.hi[1][ct][8 * hf + r]) : c.hi[1][ct][8 * hf + r];  // (NP 6: one accumulator holds all three products)
                if (TANGENT) {
                    float g0;
                    if (ACT == 0) {
                        y0[r] = nm_softplus_l2(z0, &g0);
                    } else {
                        y0[r] = fmaxf(z0, 0.f);
                        g0 = z0 > 0.f ? 1.f : 0.f;
                    }
                    y1[r] = z1 * g0;
                } else {
                    if (ACT == 0) {
                        y0[r] = nm_softplus_l2(z0, nullptr);
                        y1[r] = nm_softplus_l2(z1, nullptr);
                    } else {
                        y0[r] = fmaxf(z0, 0.f);
                        y1[r] = fmaxf(z1, 0.f);
                    }
                }
            }
            if (!LAST) {
                if (NP == 3) {
                    nm_h2_store8(tile + li * NM_H_STRIDE + col0, y0, mx);
                    nm_h2_store8(tile + (32 + li) * NM_H_STRIDE + col0, y1, mx);
                } else if (NP == 6) {   // one accumulator: unscaled residual halves
                    nm_h2_store8<NM_H_PLANE, true>(tile + li * NM_H_STRIDE + col0, y0, mx);
                    nm_h2_store8<NM_H_PLANE, true>(tile + (32 + li) * NM_H_STRIDE + col0, y1, mx);
                } else {   // single product: the main halves only
                    nm_h1_store8(tile + li * NM_H_STRIDE + col0, y0, mx);
                    nm_h1_store8(tile + (32 + li) * NM_H_STRIDE + col0, y1, mx);
                }
            } else {
#pragma unroll
                for (int o = 0; o < NOUT; ++o) {
                    const float4 w0 = *reinterpret_cast<const float4*>(head_w + o * NM_W + col0);
                    const float4 w1 = *reinterpret_cast<const float4*>(head_w + o * NM_W + col0 + 4);
                    const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        so[0][o] = fmaf(y0[r], wv[r], so[0][o]);
                        so[1][o] = fmaf(y1[r], wv[r], so[1][o]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (!LAST) {
        __syncthreads();
    } else {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {  // the two half-waves hold the same points, different columns
            so[0][o] += __shfl_xor(so[0][o], 32);
            so[1][o] += __shfl_xor(so[1][o], 32);
        }
        if (h == 0) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                red[(wave * NM_ROWS + li) * NOUT + o] = so[0][o];
                red[(wave * NM_ROWS + 32 + li) * NOUT + o] = so[1][o];
            }
        }
        __syncthreads();
    }
    nm_phase_stamp(stamp_slot + 1);
}

__device__ __forceinline__ void nm_h2_raise(int* overflow, float mx) {
    if (overflow && !(mx < NM_H2_FP16_MAX)) *overflow = 1;  // (benign race: every writer stores 1)
}

// ------------------------------------------------------------------ geometry MLP (split-half, v2)
// Same contract as nm_geo_mlp_h_kernel.  FIXED: gdim = 32, multires_fg = 2, multires_d = 8 (Kpad0 = 192).
template <bool NABLA, bool FIXED, int NP>
__global__ __launch_bounds__(NM_H_THREADS, NM_H_WAVES_PER_SIMD) void nm_geo_mlp_h2_kernel(
    NmGeoParamsH2 prm, const float* __restrict__ fg_rec, const float* __restrict__ ds, const float* __restrict__ grad, NmRecMap rmap,
    long long npts, float* __restrict__ sdf_out, int P, int stride, int off, float* __restrict__ nabla_out, int nabla_slotted,
    NmSlotMap smap, int* __restrict__ overflow) {
    __shared__ __attribute__((aligned(16))) _Float16 tile[2 * NM_H_PLANE];
    __shared__ __attribute__((aligned(16))) float cst[(NM_H2_BIAS_LAYERS + 1) * NM_W];  // biases of layers 0..3 | density weights
    __shared__ float red[4 * NM_ROWS];
    constexpr int PTS = NABLA ? 32 : 64;
    const long long base = (long long)blockIdx.x * PTS;
    if (smap.order && smap.order[base] == 0xffffu) return;  // no point in this tile (valid entries lead each group)
    const NmDivBase rdiv = nm_div_base(base, rmap.stride ? rmap.P : 1), odiv = nm_div_base(base, P);
    // by_list: records and outputs addressed by the (ray, sample) the list entry names (the nablas of the sample points
    // whose visibility weight is not zero, evaluated after the sampling passes from their slot records)
    const bool by_list = rmap.by_list && smap.order;
    const long long ray0 = by_list ? (base / smap.E) * smap.G : 0;  // uniform: one division per workgroup
    auto locate = [&](int p_local, long long& rq, long long& oidx) {   // record index / (ray, sample) output index of point base + p_local
        if (by_list) {
            long long ray;
            int sp;
            nm_slot_ray(smap, base + p_local, ray0, ray, sp);
            rq = ray * rmap.stride + (rmap.slot ? (long long)rmap.slot[ray * rmap.stride + rmap.off + sp] : rmap.off + sp);
            oidx = ray * stride + off + sp;
        } else {
            rq = nm_rec_index_local(rmap, rdiv, base, p_local);
            long long orow;
            int op;
            nm_div_local(odiv, p_local, orow, op);
            oidx = orow * stride + off + op;
        }
    };
    nm_phase_stamp(0);
    constexpr int DEPTH0 = 2;  // layer-0 prefetch distance (nm_kloop_h2); the rolled loops use one set
    NmBPre<NM_H_CT> pre;
    nm_prefetch_bn<NM_H_CT, NP>(prm.layer[0], pre, DEPTH0);  // in flight during the input phase
    const int gdim = FIXED ? 32 : prm.gdim, mfg = FIXED ? 2 : prm.multires_fg, md = FIXED ? 8 : prm.multires_d;
    const int FG = FIXED ? 160 : prm.fg_w, in_dim = FG + 2 * md + 1;
    const int Kpad0 = FIXED ? 192 : prm.layer[0].Kpad;
    const int kt0 = NABLA ? (FG >> 4) : 0;  // tangent rows are zero before the ds block
    const int nchunk = gdim >> 2;           // <= 16: at most two 4-dim chunks per lane
    constexpr int ROUNDS = PTS * 8 / NM_H_THREADS;
    float in_ds[ROUNDS];
    float4 in_fg[ROUNDS][2];
    bool in_ok[ROUNDS];
    // all global loads of the input phase first (both task rounds), then the constants, then the embedding work
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int task = threadIdx.x + rd * NM_H_THREADS;
        const int p = task >> 3, j = task & 7;
        in_ds[rd] = 0.f;
        in_fg[rd][0] = in_fg[rd][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        in_ok[rd] = base + p < npts && nm_slot_valid(smap, base + p);
        if (in_ok[rd]) {
            long long rq, unused_o;
            locate(p, rq, unused_o);
            in_ds[rd] = ds[rq];
            if (j < nchunk) in_fg[rd][0] = *reinterpret_cast<const float4*>(fg_rec + rq * gdim + 4 * j);
            if (j + 8 < nchunk) in_fg[rd][1] = *reinterpret_cast<const float4*>(fg_rec + rq * gdim + 4 * (j + 8));
        }
    }
    nm_phase_stamp(10);
    // biases / head weights: requested now, stored to LDS after the embedding work (their latency under it, one wait less)
    float cst_v[NM_H2_BIAS_LAYERS + 1];
    {
        const int nb = prm.D < NM_H2_BIAS_LAYERS ? prm.D : NM_H2_BIAS_LAYERS;
#pragma unroll
        for (int l = 0; l < NM_H2_BIAS_LAYERS; ++l) cst_v[l] = l < nb ? prm.layer[l].b[threadIdx.x] : 0.f;
        cst_v[NM_H2_BIAS_LAYERS] = prm.wd[threadIdx.x];
    }
    nm_phase_stamp(11);
#ifdef NM_TESTING_INPUT_STAMPS   // (measurement build only: stamp 13 = the record loads have arrived; 14 = first task round embedded; 8 = all rounds)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    nm_phase_stamp(13);
#endif
    float mx = 0.f;
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
#ifdef NM_TESTING_INPUT_STAMPS
        if (rd == 1) nm_phase_stamp(14);
#endif
        const int task = threadIdx.x + rd * NM_H_THREADS;
        const int p = task >> 3, j = task & 7;
        const long long q = base + p;
        _Float16* vrow = tile + p * NM_H_STRIDE;
        _Float16* trow = tile + (32 + p) * NM_H_STRIDE;  // (NABLA only)
        if (!in_ok[rd]) {
            nm_h2_zero_cols(vrow, 0, Kpad0, j);
            if (NABLA) nm_h2_zero_cols(trow, 16 * kt0, Kpad0, j);
            continue;
        }
        const float dsv = in_ds[rd];
        constexpr bool US = NP == 6;                      // residual halves unscaled (one-accumulator mode)
        constexpr float TS = nm_h2_tscale<NP>();
        if (j < nchunk) nm_h2_embed_chunk<NM_H_PLANE, US>(vrow, gdim, mfg, j, in_fg[rd][0], mx);
        if (j + 8 < nchunk) nm_h2_embed_chunk<NM_H_PLANE, US>(vrow, gdim, mfg, j + 8, in_fg[rd][1], mx);
        for (int b = j; b < md; b += 8) {  // ds block: (sin, cos) pairs, then ds itself
            const float f = (float)(1 << b);
            float s, co;
            nm_sincos(dsv * f, &s, &co);
            nm_h2_store2<NM_H_PLANE, US>(vrow + FG + 2 * b, s, co, mx);
            if (NABLA) nm_h2_store2<NM_H_PLANE, US>(trow + FG + 2 * b, (TS * f) * co, -(TS * f) * s, mx);
        }
        if (j == 0) {
            nm_h2_store1<NM_H_PLANE, US>(vrow + FG + 2 * md, dsv, mx);
            if (NABLA) nm_h2_store1<NM_H_PLANE, US>(trow + FG + 2 * md, TS, mx);
        }
        for (int c = in_dim + j; c < Kpad0; c += 8) {  // padding columns
            vrow[c] = (_Float16)0.0f;
            vrow[NM_H_PLANE + c] = (_Float16)0.0f;
            if (NABLA) {
                trow[c] = (_Float16)0.0f;
                trow[NM_H_PLANE + c] = (_Float16)0.0f;
            }
        }
        if (NABLA)
            for (int c = 16 * kt0 + j; c < FG; c += 8) {  // tangent columns of the first live k-step below the ds block
                trow[c] = (_Float16)0.0f;
                trow[NM_H_PLANE + c] = (_Float16)0.0f;
            }
    }
#ifdef NM_TESTING_INPUT_STAMPS
    nm_phase_stamp(8);
#endif
#pragma unroll
    for (int l = 0; l <= NM_H2_BIAS_LAYERS; ++l) cst[l * NM_W + threadIdx.x] = cst_v[l];
    nm_phase_stamp(12);
    __syncthreads();
    nm_phase_stamp(1);
    // layer 0: straight-line K loop for the reference widths (FIXED), rolled loops otherwise; hidden layers are
    // always 256 wide (16 k-steps).  (kernel-argument loads with a uniform index: scalar)
    constexpr int KS0 = FIXED ? 12 : 0, KT0 = (FIXED && NABLA) ? 10 : 0;
    const float* head = cst + NM_H2_BIAS_LAYERS * NM_W;
    if (prm.D == 1) {
        nm_mlp_layer_h2<0, NABLA, true, 1, NM_H_CT, KS0, KT0, DEPTH0, NP>(tile, prm.layer[0], kt0, false, prm.layer[0], pre, cst, 0, head, red, mx, 2);
    } else {
        nm_mlp_layer_h2<0, NABLA, false, 1, NM_H_CT, KS0, KT0, DEPTH0, NP>(tile, prm.layer[0], kt0, true, prm.layer[1], pre, cst, 0, nullptr, red, mx, 2);
        for (int l = 1; l + 1 < prm.D; ++l)
            nm_mlp_layer_h2<0, NABLA, false, 1, NM_H_CT, 16, 0, 2, NP>(tile, prm.layer[l], 0, true, prm.layer[l + 1], pre,
                                                                    cst, l < NM_H2_BIAS_LAYERS ? l : -1, nullptr, red, mx, 2 + 2 * l);
        const int l = prm.D - 1;
        nm_mlp_layer_h2<0, NABLA, true, 1, NM_H_CT, 16, 0, 2, NP>(tile, prm.layer[l], 0, false, prm.layer[l], pre,
                                                               cst, l < NM_H2_BIAS_LAYERS ? l : -1, head, red, mx, 2 + 2 * l);
    }
    if (threadIdx.x < PTS) {
        const int t = threadIdx.x;
        const long long q = base + t;
        if (q < npts && nm_slot_valid(smap, q)) {
            const float sdf = ((red[t] + red[NM_ROWS + t]) + (red[2 * NM_ROWS + t] + red[3 * NM_ROWS + t])) + prm.bd;
            long long rq, oidx;  // record / (ray, sample) addressed output position
            locate(t, rq, oidx);
            if (sdf_out) sdf_out[oidx] = sdf;
            if (NABLA && nabla_out) {
                const float dsdf = ((red[32 + t] + red[NM_ROWS + 32 + t]) + (red[2 * NM_ROWS + 32 + t] + red[3 * NM_ROWS + 32 + t])) * (1.0f / nm_h2_tscale<NP>());
                const long long no = nabla_slotted ? oidx : q;
                nabla_out[no * 3 + 0] = dsdf * grad[rq * 3 + 0];
                nabla_out[no * 3 + 1] = dsdf * grad[rq * 3 + 1];
                nabla_out[no * 3 + 2] = dsdf * grad[rq * 3 + 2];
            }
        }
    }
    nm_h2_raise(overflow, mx);
    nm_phase_stamp(15);
}

// ------------------------------------------------------------------ colour MLP (split-half, v2)
// Physical input columns: [ft embedding | (sin, cos) pairs of ds | view bands | view | nabla | ds].
// FIXED: cdim = 32, multires_ft = 2, multires_d = 8, multires_view = 4, nabla input (207 -> Kpad0 = 208).
template <bool FIXED, int NP>
__global__ __launch_bounds__(NM_H_THREADS, NM_H_WAVES_PER_SIMD) void nm_col_mlp_h2_kernel(
    NmColParamsH2 prm, const float* __restrict__ ft_rec, const float* __restrict__ ds, const float* __restrict__ nabla,
    const float* __restrict__ dirs, int dir_div, long long npts, float* __restrict__ rgb_out, NmSlotMap smap, int* __restrict__ overflow) {
    __shared__ __attribute__((aligned(16))) _Float16 tile[2 * NM_H_PLANE];
    __shared__ __attribute__((aligned(16))) float cst[(NM_H2_BIAS_LAYERS + 3) * NM_W];  // biases of layers 0..3 | rgb weights [3][256]
    __shared__ float red[4 * NM_ROWS * 3];
    const long long base = (long long)blockIdx.x * NM_ROWS;
    if (smap.order && smap.order[base] == 0xffffu) return;  // no point in this tile
    const NmDivBase ddiv = nm_div_base(base, dir_div);
    const long long ray0 = smap.order ? (base / smap.E) * smap.G : 0;  // uniform: one division per workgroup
    nm_phase_stamp(0);
    constexpr int DEPTH0 = 2;
    NmBPre<NM_H_CT> pre;
    nm_prefetch_bn<NM_H_CT, NP>(prm.layer[0], pre, DEPTH0);  // in flight during the input phase
    const int cdim = FIXED ? 32 : prm.cdim, mft = FIXED ? 2 : prm.multires_ft, md = FIXED ? 8 : prm.multires_d;
    const int mv = FIXED ? 4 : prm.multires_view, use_nabla = FIXED ? 1 : prm.use_nabla;
    const int FT = FIXED ? 160 : prm.ft_w;
    const int o_d = FT, o_vb = o_d + 2 * md, o_v = o_vb + 6 * mv, o_n = o_v + 3, o_ds = o_n + (use_nabla ? 3 : 0);
    const int in_dim = o_ds + 1;
    const int Kpad0 = FIXED ? 208 : prm.layer[0].Kpad;
    const int nchunk = cdim >> 2;
    constexpr int ROUNDS = NM_ROWS * 8 / NM_H_THREADS;  // global loads of both task rounds first
    float in_ds[ROUNDS], in_x[ROUNDS];  // in_x: lane j < 3: view component j, 3 <= j < 6: nabla component j - 3
    float in_dv[ROUNDS][3];
    float4 in_ft[ROUNDS][2];
    bool in_ok[ROUNDS];
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int task = threadIdx.x + rd * NM_H_THREADS;
        const int p = task >> 3, j = task & 7;
        const long long q = base + p;
        in_ds[rd] = in_x[rd] = 0.f;
        in_dv[rd][0] = in_dv[rd][1] = in_dv[rd][2] = 0.f;
        in_ft[rd][0] = in_ft[rd][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        in_ok[rd] = q < npts && nm_slot_valid(smap, q);
        if (in_ok[rd]) {
            in_ds[rd] = ds[q];
            if (use_nabla && j >= 3 && j < 6) in_x[rd] = nabla[q * 3 + (j - 3)];
            if (j < nchunk) in_ft[rd][0] = *reinterpret_cast<const float4*>(ft_rec + q * cdim + 4 * j);
            if (j + 8 < nchunk) in_ft[rd][1] = *reinterpret_cast<const float4*>(ft_rec + q * cdim + 4 * (j + 8));
        }
        if (in_ok[rd]) {
            long long ray;
            int unused_p;
            if (smap.order) nm_slot_ray(smap, q, ray0, ray, unused_p);
            else nm_div_local(ddiv, p, ray, unused_p);
            in_dv[rd][0] = dirs[ray * 3 + 0];
            in_dv[rd][1] = dirs[ray * 3 + 1];
            in_dv[rd][2] = dirs[ray * 3 + 2];
        }
    }
    float cst_v[NM_H2_BIAS_LAYERS + 3];  // (requested now, stored to LDS after the embedding work)
    {
        const int nb = prm.D < NM_H2_BIAS_LAYERS ? prm.D : NM_H2_BIAS_LAYERS;
#pragma unroll
        for (int l = 0; l < NM_H2_BIAS_LAYERS; ++l) cst_v[l] = l < nb ? prm.layer[l].b[threadIdx.x] : 0.f;
#pragma unroll
        for (int o = 0; o < 3; ++o) cst_v[NM_H2_BIAS_LAYERS + o] = prm.wrgb[o * NM_W + threadIdx.x];
    }
    float mx = 0.f;
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int task = threadIdx.x + rd * NM_H_THREADS;
        const int p = task >> 3, j = task & 7;
        const long long q = base + p;
        _Float16* vrow = tile + p * NM_H_STRIDE;
        if (!in_ok[rd]) {
            nm_h2_zero_cols(vrow, 0, Kpad0, j);
            continue;
        }
        const float dsv = in_ds[rd];
        const float dv[3] = {in_dv[rd][0], in_dv[rd][1], in_dv[rd][2]};
        constexpr bool US = NP == 6;
        if (j < nchunk) nm_h2_embed_chunk<NM_H_PLANE, US>(vrow, cdim, mft, j, in_ft[rd][0], mx);
        if (j + 8 < nchunk) nm_h2_embed_chunk<NM_H_PLANE, US>(vrow, cdim, mft, j + 8, in_ft[rd][1], mx);
        for (int b = j; b < md; b += 8) {
            float s, co;
            nm_sincos(dsv * (float)(1 << b), &s, &co);
            nm_h2_store2<NM_H_PLANE, US>(vrow + o_d + 2 * b, s, co, mx);
        }
        for (int e = j; e < 3 * mv; e += 8) {  // view bands: [sin(v f_b) (3) | cos(v f_b) (3)] per band
            const int b = e / 3, dim = e - 3 * b;
            float s, co;
            nm_sincos((dim == 0 ? dv[0] : dim == 1 ? dv[1] : dv[2]) * (float)(1 << b), &s, &co);
            nm_h2_store1<NM_H_PLANE, US>(vrow + o_vb + 6 * b + dim, s, mx);
            nm_h2_store1<NM_H_PLANE, US>(vrow + o_vb + 6 * b + 3 + dim, co, mx);
        }
        if (j < 3) nm_h2_store1<NM_H_PLANE, US>(vrow + o_v + j, j == 0 ? dv[0] : j == 1 ? dv[1] : dv[2], mx);
        else if (j < 6 && use_nabla) nm_h2_store1<NM_H_PLANE, US>(vrow + o_n + (j - 3), in_x[rd], mx);
        else if (j == 6) nm_h2_store1<NM_H_PLANE, US>(vrow + o_ds, dsv, mx);
        for (int c = in_dim + j; c < Kpad0; c += 8) {
            vrow[c] = (_Float16)0.0f;
            vrow[NM_H_PLANE + c] = (_Float16)0.0f;
        }
    }
#pragma unroll
    for (int l = 0; l < NM_H2_BIAS_LAYERS + 3; ++l) cst[l * NM_W + threadIdx.x] = cst_v[l];
    __syncthreads();
    nm_phase_stamp(1);
    constexpr int KS0 = FIXED ? 13 : 0;
    const float* head = cst + NM_H2_BIAS_LAYERS * NM_W;
    if (prm.D == 1) {
        nm_mlp_layer_h2<1, false, true, 3, NM_H_CT, KS0, 0, 2, NP>(tile, prm.layer[0], 0, false, prm.layer[0], pre, cst, 0, head, red, mx, 2);
    } else {
        nm_mlp_layer_h2<1, false, false, 3, NM_H_CT, KS0, 0, 2, NP>(tile, prm.layer[0], 0, true, prm.layer[1], pre, cst, 0, nullptr, red, mx, 2);
        for (int l = 1; l + 1 < prm.D; ++l)
            nm_mlp_layer_h2<1, false, false, 3, NM_H_CT, 16, 0, 2, NP>(tile, prm.layer[l], 0, true, prm.layer[l + 1], pre,
                                                                    cst, l < NM_H2_BIAS_LAYERS ? l : -1, nullptr, red, mx, 2 + 2 * l);
        const int l = prm.D - 1;
        nm_mlp_layer_h2<1, false, true, 3, NM_H_CT, 16, 0, 2, NP>(tile, prm.layer[l], 0, false, prm.layer[l], pre,
                                                               cst, l < NM_H2_BIAS_LAYERS ? l : -1, head, red, mx, 2 + 2 * l);
    }
    if (threadIdx.x < NM_ROWS) {  // one thread per point: its three channels are one 12-byte store
        const int p = threadIdx.x;
        const long long q = base + p;
        if (q < npts && nm_slot_valid(smap, q)) {
            long long oq = q;  // ordered lists: the colour goes back to its (ray, sample) position
            if (smap.order) {
                long long ray;
                int sp;
                nm_slot_ray(smap, q, ray0, ray, sp);
                oq = ray * smap.P + sp;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float z = ((red[p * 3 + c] + red[(NM_ROWS + p) * 3 + c]) + (red[(2 * NM_ROWS + p) * 3 + c] + red[(3 * NM_ROWS + p) * 3 + c])) + prm.brgb[c];
                rgb_out[oq * 3 + c] = __fdiv_rn(1.0f, 1.0f + expf(-z));
            }
        }
    }
    nm_h2_raise(overflow, mx);
    nm_phase_stamp(15);
}
