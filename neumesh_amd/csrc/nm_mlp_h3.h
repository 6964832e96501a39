// nm_mlp_h3.h -- fused embed + MLP kernels, split-half f16 MFMA, v3: the epilogue of one half of the tile runs INSIDE
// the K loop of the other half (reference configuration only; every other configuration keeps the v2 kernels of nm_mlp_h2.h).
//
// Why (round-2 measurements, tools/coissue3.hip): the v2 kernels alternate pure-MFMA K loops with pure-vector epilogues and
// rely on a second workgroup of the CU being in the opposite phase; they reach 55 % of the matrix pipe.  The hardware
// itself hides vector work behind MFMAs when both come from the SAME instruction stream: with two waves per SIMD, one
// v_mfma_f32_32x32x16_f16 followed by up to 6 vector instructions costs what the MFMA alone costs (25.9 vs 25.1 ticks per
// MFMA of the SIMD; 8 fillers: 32.3).  The epilogue of a layer is ~3.2 vector instructions per MFMA of that layer.
//
// How: one workgroup = 8 waves = a 128-row x 256-column activation tile (two planes, 135 KB of LDS: one workgroup per CU,
// still two waves per SIMD).  Wave w owns the 32 output columns of column tile w for all rows: 4 row tiles x (main,
// scaled) accumulators = 128 VGPRs, as in v2.  The rows form two PAIRS of row tiles (rows 0-63 / 64-127; with nablas:
// the 64 points' value rows / their tangent rows).  Per layer l:
//     phase X_l:  K loop of pair 0, layer l        ||  epilogue of pair 1, layer l-1   (writes rows 64-127 in place)
//     phase Y_l:  K loop of pair 1, layer l        ||  epilogue of pair 0, layer l     (writes rows 0-63 in place)
// with one barrier after each phase.  A K loop only reads its own pair's rows, an epilogue only writes its own pair's
// rows, and a pair's rows are rewritten only after every wave has finished the K loop that read them: the in-place tile
// of v2 stays.  Weight fragments are read twice per layer (once per pair) but serve 128 rows instead of 64, so the L2
// traffic per point is v2's.  The arithmetic per accumulator is v2's sequence (hi += W1 A1, lo += W2 A1, lo += W1 A2 per
// k-step; epilogue and head as nm_mlp_layer_h2), so the value rows of the nabla kernel stay bit-identical to the
// forward-only kernel, which the lazy-nabla path of the renderer relies on.
//
// Reference semantics: models/frameworks/neumesh/neumesh.py:204-260, models/base.py:52-70 (see nm_mlp.h).
#pragma once

#include "nm_mlp_h2.h"

#define NM_H3_ROWS 128
#define NM_H3_THREADS 512
#define NM_H3_WAVES 8
#ifndef NM_H3_LB
#define NM_H3_LB 2  // waves per SIMD the kernels are compiled for (512 threads = 2 per SIMD; 1 only to read the unconstrained register demand)
#endif
// LDS tile: row-major, the two fp16 planes of a row side by side: [h1: 256 halves | h2: 256 halves | 8 halves of padding].
// Row stride 1040 B = 260 dwords = 4 (mod 64) banks: the conflict-free ds_read_b128 pattern of v2's 528-byte rows, and
// every offset inside a pair of row tiles (row tile + plane + k-step) fits the 16-bit immediate of the LDS instructions
// (with v2's plane-major layout at 128 rows the second plane is 67 KB away and every access needs its own address VGPR).
#define NM_H3_PLANE 256                         // halves from a row's h1 plane to its h2 plane
#define NM_H3_STRIDE (2 * 256 + 8)              // halves per tile row
#define NM_H3_TILE (NM_H3_ROWS * NM_H3_STRIDE)  // halves

struct NmAcc3 {  // one pair of row tiles: [row tile of the pair]
    nm_f32x16 hi[2], lo[2];
};
struct NmB3 {  // weight fragment of one k-step (one column tile): planes h1, h2
    nm_h8 a, b;
};
#ifndef NM_H3_DB
#define NM_H3_DB 2  // weight fragments are requested this many k-steps ahead (DB + 1 rotating register sets)
#endif
#ifndef NM_H3_DA
#define NM_H3_DA 1  // activation fragments: k-steps ahead (DA + 1 register sets)
#endif
#ifndef NM_H3_EXP_NOA
#define NM_H3_EXP_NOA 0  // measurement switch: activation fragments are read for the first k-step only
#endif
struct NmBPre3 {
    NmB3 s[NM_H3_DB];
};
// measurement switches (wrong results, timing only; tools/mlp_ab.py with a -D build): NM_H3_EXP_SAMEW = every k-step reads
// the fragment of k-step 0 (no L2 traffic), NM_H3_EXP_NOEPI = no epilogue work inside the K loops
#ifndef NM_H3_EXP_SAMEW
#define NM_H3_EXP_SAMEW 0
#endif
#ifndef NM_H3_EXP_NOEPI
#define NM_H3_EXP_NOEPI 0
#endif
__device__ __forceinline__ NmB3 nm_h3_ldb(nm_rsrc r, int lane16, int ks) {
    NmB3 f;
    if (NM_H3_EXP_SAMEW) ks = 0;
    f.a = __builtin_bit_cast(nm_h8, __builtin_amdgcn_raw_buffer_load_b128(r, lane16, ks * 2048, 0));
    f.b = __builtin_bit_cast(nm_h8, __builtin_amdgcn_raw_buffer_load_b128(r, lane16, ks * 2048 + 1024, 0));
    return f;
}
// the first two k-steps of a K loop that starts at k-step ks0 of layer L (requested a phase ahead)
__device__ __forceinline__ void nm_h3_prefetch(const NmLayerH L, int wave, int lane16, int ks0, NmBPre3& pre) {
    const nm_rsrc r = nm_b_rsrc(L.W, L.Kpad, wave);
#pragma unroll
    for (int i = 0; i < NM_H3_DB; ++i) pre.s[i] = nm_h3_ldb(r, lane16, ks0 + i);  // (every K loop has >= 2 k-steps; reading past a short one stays inside the layer's block)
}

// ----------------------------------------------------------------------------- epilogue, in chunks
// Chunk q = (row tile rt = q >> 1, half hf = q & 1) of a pair: 8 of this lane's 16 activations of the row tile, registers
// [8 hf, 8 hf + 8) <-> columns col0 + 8 hf + r of the lane's point.  MODE 0: value rows; 1: value rows that keep
// act'(z) in g for their tangent rows; 2: tangent rows (y = z g).  LAST: the NOUT-wide head instead of the tile store.
template <int ACT, int MODE, bool LAST, int NOUT>
struct NmEpi3 {
    const NmAcc3& c;
    float (&g)[2][16];
    _Float16* dst0;        // row of (pair, row tile 0, this lane's point) at column col0; row tile 1 is 32 rows further
    const float* head_w;   // LDS [NOUT][256] + col0 (LAST)
    float (&so)[2][NOUT];
    float& mx;
    __device__ __forceinline__ void chunk(const int q) {
        const int rt = q >> 1, hf = q & 1;
        const float sc = 1.0f / 2048.0f;
        float y[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float z = fmaf(c.lo[rt][8 * hf + r], sc, c.hi[rt][8 * hf + r]);  // (bias: in the accumulator)
            if (MODE == 2) {
                y[r] = z * g[rt][8 * hf + r];
            } else if (ACT == 0) {
                y[r] = nm_softplus_l2(z, MODE == 1 ? &g[rt][8 * hf + r] : nullptr);
            } else {
                y[r] = fmaxf(z, 0.f);
                if (MODE == 1) g[rt][8 * hf + r] = z > 0.f ? 1.f : 0.f;
            }
        }
        if (!LAST) {
            nm_h2_store8<NM_H3_PLANE>(dst0 + rt * 32 * NM_H3_STRIDE + 8 * hf, y, mx);
        } else {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                const float4 w0 = *reinterpret_cast<const float4*>(head_w + o * NM_W + 8 * hf);
                const float4 w1 = *reinterpret_cast<const float4*>(head_w + o * NM_W + 8 * hf + 4);
                const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                for (int r = 0; r < 8; ++r) so[rt][o] = fmaf(y[r], wv[r], so[rt][o]);
            }
        }
    }
};
struct NmNoEpi3 {
    __device__ __forceinline__ void chunk(int) {}
};

// ------------------------------------------------------------------------------------- K loop
// k-steps [KS0, KS1) of one layer for one pair of row tiles (6 MFMAs per k-step), in up to four blocks; after the
// k-steps of block b the epilogue chunks assigned to it are issued and the scheduler is free to interleave the two
// inside the block (sched_barriers only at block boundaries).  Weight fragments two k-steps ahead in three rotating
// register sets (the first two arrive in `pre`), activation fragments one k-step ahead.
#define NM_H3_MFMAS(A, F)                                                                                        \
    _Pragma("unroll") for (int rt_ = 0; rt_ < 2; ++rt_) c.hi[rt_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a, A[rt_][0], c.hi[rt_], 0, 0, 0); \
    _Pragma("unroll") for (int rt_ = 0; rt_ < 2; ++rt_) c.lo[rt_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.b, A[rt_][0], c.lo[rt_], 0, 0, 0); \
    _Pragma("unroll") for (int rt_ = 0; rt_ < 2; ++rt_) c.lo[rt_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a, A[rt_][1], c.lo[rt_], 0, 0, 0);

#ifndef NM_H3_SCHED
#define NM_H3_SCHED 1  // 1: sched_group_barrier pattern (one MFMA, then vector / memory work) inside a block; 0: scheduler's choice
#endif
#ifndef NM_H3_VPM
#define NM_H3_VPM 3    // vector instructions placed behind each MFMA by the pattern
#endif

template <int KS0, int KS1, class Epi>
__device__ __forceinline__ void nm_h3_kloop(const _Float16* ap0, const nm_rsrc bp, const int lane16, const NmBPre3& pre, NmAcc3& c, Epi& epi) {
    constexpr int N = KS1 - KS0, NB = N < 4 ? N : 4;
    const _Float16* ap1 = ap0 + 32 * NM_H3_STRIDE;
    constexpr int DB = NM_H3_DB, DA = NM_H3_DA;
    NmB3 f[DB + 1];
#pragma unroll
    for (int i = 0; i < DB; ++i) f[i] = pre.s[i];
    nm_h8 a[DA + 1][2][2];  // [buffer][row tile][plane]
#pragma unroll
    for (int i = 0; i < DA; ++i) {
        if (i < N) {
            a[i][0][0] = *reinterpret_cast<const nm_h8*>(ap0 + (KS0 + i) * 16);
            a[i][0][1] = *reinterpret_cast<const nm_h8*>(ap0 + NM_H3_PLANE + (KS0 + i) * 16);
            a[i][1][0] = *reinterpret_cast<const nm_h8*>(ap1 + (KS0 + i) * 16);
            a[i][1][1] = *reinterpret_cast<const nm_h8*>(ap1 + NM_H3_PLANE + (KS0 + i) * 16);
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = b * N / NB; j < (b + 1) * N / NB; ++j) {
            const int ks = KS0 + j;
            if (j + DB < N) f[(j + DB) % (DB + 1)] = nm_h3_ldb(bp, lane16, ks + DB);
            if (j + DA < N && !NM_H3_EXP_NOA) {
                const int oa = (ks + DA) * 16;
                a[(j + DA) % (DA + 1)][0][0] = *reinterpret_cast<const nm_h8*>(ap0 + oa);
                a[(j + DA) % (DA + 1)][0][1] = *reinterpret_cast<const nm_h8*>(ap0 + NM_H3_PLANE + oa);
                a[(j + DA) % (DA + 1)][1][0] = *reinterpret_cast<const nm_h8*>(ap1 + oa);
                a[(j + DA) % (DA + 1)][1][1] = *reinterpret_cast<const nm_h8*>(ap1 + NM_H3_PLANE + oa);
            }
            NM_H3_MFMAS(a[NM_H3_EXP_NOA ? 0 : j % (DA + 1)], f[j % (DB + 1)])
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (!NM_H3_EXP_NOEPI && q * NB / 4 == b) epi.chunk(q);
#if NM_H3_SCHED
        // desired issue order inside the block: every MFMA is followed by a few vector instructions of the chunk and one
        // memory operation (the block's loads first, the chunk's LDS stores last); what does not fit the pattern follows
#pragma unroll
        for (int m = 0; m < 6 * ((b + 1) * N / NB - b * N / NB); ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x402, NM_H3_VPM, 0);    // vector ALU / transcendental
            __builtin_amdgcn_sched_group_barrier(0x120, 1, 0);            // one LDS read or buffer load
        }
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
}
#undef NM_H3_MFMAS

// main accumulators of a pair of value rows start at the layer's bias (this lane's 16 columns), everything else at zero
__device__ __forceinline__ void nm_h3_init(NmAcc3& c, const bool value_rows, const float* cst, const int bias_row, const float* bias_global, const int col0) {
    nm_f32x16 bv = nm_f32x16{0};
    if (value_rows) {
        float4 b4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (bias_row >= 0) {  // (two address spaces: LDS for the first layers, global beyond -- never a generic pointer)
                b4[q] = *reinterpret_cast<const float4*>(cst + bias_row * NM_W + col0 + 4 * q);
            } else {
                const nm_rsrc rb = __builtin_amdgcn_make_buffer_rsrc((void*)bias_global, 0, NM_W * 4, 0x00020000);
                b4[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rb, (col0 + 4 * q) * 4, 0, 0));
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bv[4 * q + 0] = b4[q].x; bv[4 * q + 1] = b4[q].y; bv[4 * q + 2] = b4[q].z; bv[4 * q + 3] = b4[q].w;
        }
    }
    c.hi[0] = c.hi[1] = bv;
    c.lo[0] = c.lo[1] = nm_f32x16{0};
}

// per-row head sums of a pair -> red[wave][row][NOUT] (the two half-waves hold the same points, different columns)
template <int NOUT>
__device__ __forceinline__ void nm_h3_head_out(float (&so)[2][NOUT], float* red, int wave, int pair, int li, int h) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int o = 0; o < NOUT; ++o) so[rt][o] += __shfl_xor(so[rt][o], 32);
    if (h == 0) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int o = 0; o < NOUT; ++o) red[(wave * NM_H3_ROWS + 64 * pair + 32 * rt + li) * NOUT + o] = so[rt][o];
    }
}
template <int NOUT>
__device__ __forceinline__ float nm_h3_head_sum(const float* red, int row, int o) {
    float v[NM_H3_WAVES];
#pragma unroll
    for (int w = 0; w < NM_H3_WAVES; ++w) v[w] = red[(w * NM_H3_ROWS + row) * NOUT + o];
    return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}

// All layers of one MLP on the tile.  ACT: 0 softplus (log2 units), 1 ReLU.  TANGENT: pair 1 = tangent rows of pair 0's
// points (first non-zero k-step of layer 0: KT0).  KS0: k-steps of layer 0.  On exit red[] holds the head sums of all rows.
template <int ACT, bool TANGENT, int NOUT, int KS0, int KT0>
__device__ __forceinline__ void nm_h3_layers(_Float16* tile, const NmLayerH* layer, const int D, const float* cst, const float* head_w,
                                             float* red, float& mx, NmBPre3& pre /* k-steps 0, 1 of layer 0 (requested before the input phase) */) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, h = lane >> 5, lane16 = lane * 16;
    const int col0 = wave * 32 + 16 * h;  // this lane's 16 columns: register r <-> column col0 + r
    const _Float16* ap[2] = {tile + li * NM_H3_STRIDE + 8 * h, tile + (64 + li) * NM_H3_STRIDE + 8 * h};
    _Float16* dst[2] = {tile + li * NM_H3_STRIDE + col0, tile + (64 + li) * NM_H3_STRIDE + col0};
    constexpr int MODE_V = TANGENT ? 1 : 0, MODE_T = TANGENT ? 2 : 0;
    NmAcc3 c0, c1;
    float g[2][16];
    float so0[2][NOUT], so1[2][NOUT];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) so0[rt][o] = so1[rt][o] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) g[rt][r] = 0.f;
    }
    __syncthreads();  // the input phase has written the tile and the constants
    nm_phase_stamp(1);
    nm_h3_init(c0, true, cst, 0, layer[0].b, col0);
    {   // X_0: K loop of pair 0, nothing to overlap yet
        NmNoEpi3 none;
        nm_h3_kloop<0, KS0>(ap[0], nm_b_rsrc(layer[0].W, layer[0].Kpad, wave), lane16, pre, c0, none);
        nm_h3_prefetch(layer[0], wave, lane16, TANGENT ? KT0 : 0, pre);
    }
    __syncthreads();
    nm_phase_stamp(2);
    nm_h3_init(c1, !TANGENT, cst, 0, layer[0].b, col0);
    {   // Y_0: K loop of pair 1 (tangent rows: only their non-zero k-steps) || epilogue of pair 0
        NmEpi3<ACT, MODE_V, false, NOUT> e = {c0, g, dst[0], head_w + col0, so0, mx};
        nm_h3_kloop<(TANGENT ? KT0 : 0), KS0>(ap[1], nm_b_rsrc(layer[0].W, layer[0].Kpad, wave), lane16, pre, c1, e);
        nm_h3_prefetch(layer[1], wave, lane16, 0, pre);
    }
    __syncthreads();
    nm_phase_stamp(3);
    for (int l = 1; l < D; ++l) {
        const NmLayerH L = layer[l];
        const nm_rsrc bp = nm_b_rsrc(L.W, L.Kpad, wave);
        const int brow = l < NM_H2_BIAS_LAYERS ? l : -1;
        nm_h3_init(c0, true, cst, brow, L.b, col0);
        {   // X_l: K loop of pair 0 || epilogue of pair 1, layer l-1
            NmEpi3<ACT, MODE_T, false, NOUT> e = {c1, g, dst[1], head_w + col0, so1, mx};
            nm_h3_kloop<0, 16>(ap[0], bp, lane16, pre, c0, e);
            nm_h3_prefetch(L, wave, lane16, 0, pre);
        }
        __syncthreads();
        nm_phase_stamp(2 + 2 * l);
        nm_h3_init(c1, !TANGENT, cst, brow, L.b, col0);
        if (l + 1 < D) {  // Y_l: K loop of pair 1 || epilogue of pair 0
            NmEpi3<ACT, MODE_V, false, NOUT> e = {c0, g, dst[0], head_w + col0, so0, mx};
            nm_h3_kloop<0, 16>(ap[1], bp, lane16, pre, c1, e);
            nm_h3_prefetch(layer[l + 1], wave, lane16, 0, pre);
        } else {          // last layer: the head is applied to pair 0's activations in registers
            NmEpi3<ACT, MODE_V, true, NOUT> e = {c0, g, dst[0], head_w + col0, so0, mx};
            nm_h3_kloop<0, 16>(ap[1], bp, lane16, pre, c1, e);
        }
        __syncthreads();
        nm_phase_stamp(3 + 2 * l);
    }
    {   // pair 1 of the last layer: nothing left to overlap
        NmEpi3<ACT, MODE_T, true, NOUT> e = {c1, g, dst[1], head_w + col0, so1, mx};
#pragma unroll
        for (int q = 0; q < 4; ++q) e.chunk(q);
    }
    nm_h3_head_out<NOUT>(so0, red, wave, 0, li, h);
    nm_h3_head_out<NOUT>(so1, red, wave, 1, li, h);
    __syncthreads();
}

// ------------------------------------------------------------------ geometry MLP (reference configuration)
// gdim = 32, multires_fg = 2, multires_d = 8 (177 inputs, Kpad0 = 192), D >= 2.  Contract as nm_geo_mlp_h2_kernel<NABLA, true>;
// 128 points per workgroup (64 with nablas).
template <bool NABLA>
__global__ __launch_bounds__(NM_H3_THREADS, NM_H3_LB) void nm_geo_mlp_h3_kernel(
    NmGeoParamsH2 prm, const float* __restrict__ fg_rec, const float* __restrict__ ds, const float* __restrict__ grad, NmRecMap rmap,
    long long npts, float* __restrict__ sdf_out, int P, int stride, int off, float* __restrict__ nabla_out, int nabla_slotted,
    NmSlotMap smap, int* __restrict__ overflow) {
    __shared__ __attribute__((aligned(16))) _Float16 tile[NM_H3_TILE];
    __shared__ __attribute__((aligned(16))) float cst[(NM_H2_BIAS_LAYERS + 1) * NM_W];  // biases of layers 0..3 | density weights
    __shared__ float red[NM_H3_WAVES * NM_H3_ROWS];
    constexpr int PTS = NABLA ? 64 : 128;
    const long long base = (long long)blockIdx.x * PTS;
    // no point in this tile (valid entries lead each group of the list; a 128-point tile may span two groups)
    if (smap.order && smap.order[base] == 0xffffu && (PTS == 64 || base + 64 >= npts || smap.order[base + 64] == 0xffffu)) return;
    const NmDivBase rdiv = nm_div_base(base, rmap.stride ? rmap.P : 1), odiv = nm_div_base(base, P);
    const bool by_list = rmap.by_list && smap.order;
    long long ray0[2] = {0, 0};  // first ray of the list group of each 64-point half (uniform)
    if (smap.order) {
        ray0[0] = (base / smap.E) * smap.G;
        ray0[1] = ((base + 64) / smap.E) * smap.G;
    }
    auto locate = [&](int p_local, long long& rq, long long& oidx) {   // record index / output index of point base + p_local
        if (by_list) {
            long long ray;
            int sp;
            nm_slot_ray(smap, base + p_local, ray0[(p_local >> 6) & 1], ray, sp);
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
    NmBPre3 pre;
    nm_h3_prefetch(prm.layer[0], __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), (threadIdx.x & 63) * 16, 0, pre);  // in flight during the input phase
    constexpr int gdim = 32, mfg = 2, md = 8, FG = 160, in_dim = FG + 2 * md + 1, Kpad0 = 192, kt0 = FG >> 4;
    constexpr int ROUNDS = PTS * 8 / NM_H3_THREADS;
    float in_ds[ROUNDS];
    float4 in_fg[ROUNDS];
    bool in_ok[ROUNDS];
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {  // all record loads first
        const int task = threadIdx.x + rd * NM_H3_THREADS;
        const int p = task >> 3, j = task & 7;
        in_ds[rd] = 0.f;
        in_fg[rd] = make_float4(0.f, 0.f, 0.f, 0.f);
        in_ok[rd] = base + p < npts && nm_slot_valid(smap, base + p);
        if (in_ok[rd]) {
            long long rq, unused_o;
            locate(p, rq, unused_o);
            in_ds[rd] = ds[rq];
            in_fg[rd] = *reinterpret_cast<const float4*>(fg_rec + rq * gdim + 4 * j);
        }
    }
    float cst_v[NM_H2_BIAS_LAYERS + 1];  // biases / head weights: requested now, stored to LDS after the embedding work
    {
        const int nb = prm.D < NM_H2_BIAS_LAYERS ? prm.D : NM_H2_BIAS_LAYERS;
#pragma unroll
        for (int l = 0; l <= NM_H2_BIAS_LAYERS; ++l) cst_v[l] = 0.f;
        if (threadIdx.x < NM_W) {
#pragma unroll
            for (int l = 0; l < NM_H2_BIAS_LAYERS; ++l) cst_v[l] = l < nb ? prm.layer[l].b[threadIdx.x] : 0.f;
            cst_v[NM_H2_BIAS_LAYERS] = prm.wd[threadIdx.x];
        }
    }
    float mx = 0.f;
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int task = threadIdx.x + rd * NM_H3_THREADS;
        const int p = task >> 3, j = task & 7;
        _Float16* vrow = tile + p * NM_H3_STRIDE;
        _Float16* trow = tile + (64 + p) * NM_H3_STRIDE;  // (NABLA only)
        if (!in_ok[rd]) {
            nm_h2_zero_cols<NM_H3_PLANE>(vrow, 0, Kpad0, j);
            if (NABLA) nm_h2_zero_cols<NM_H3_PLANE>(trow, 16 * kt0, Kpad0, j);
            continue;
        }
        const float dsv = in_ds[rd];
        nm_h2_embed_chunk<NM_H3_PLANE>(vrow, gdim, mfg, j, in_fg[rd], mx);
        {   // ds block: (sin, cos) pair of band j, then ds itself
            const float f = (float)(1 << j);
            float s, co;
            nm_sincos(dsv * f, &s, &co);
            nm_h2_store2<NM_H3_PLANE>(vrow + FG + 2 * j, s, co, mx);
            if (NABLA) nm_h2_store2<NM_H3_PLANE>(trow + FG + 2 * j, (NM_H2_TANGENT_SCALE * f) * co, -(NM_H2_TANGENT_SCALE * f) * s, mx);
        }
        if (j == 0) {
            nm_h2_store1<NM_H3_PLANE>(vrow + FG + 2 * md, dsv, mx);
            if (NABLA) nm_h2_store1<NM_H3_PLANE>(trow + FG + 2 * md, NM_H2_TANGENT_SCALE, mx);
        }
        for (int c = in_dim + j; c < Kpad0; c += 8) {  // padding columns
            vrow[c] = (_Float16)0.0f;
            vrow[NM_H3_PLANE + c] = (_Float16)0.0f;
            if (NABLA) {
                trow[c] = (_Float16)0.0f;
                trow[NM_H3_PLANE + c] = (_Float16)0.0f;
            }
        }
    }
    if (threadIdx.x < NM_W) {
#pragma unroll
        for (int l = 0; l <= NM_H2_BIAS_LAYERS; ++l) cst[l * NM_W + threadIdx.x] = cst_v[l];
    }
    const float* head = cst + NM_H2_BIAS_LAYERS * NM_W;
    nm_h3_layers<0, NABLA, 1, 12, 10>(tile, prm.layer, prm.D, cst, head, red, mx, pre);
    if (threadIdx.x < PTS) {
        const int t = threadIdx.x;
        const long long q = base + t;
        if (q < npts && nm_slot_valid(smap, q)) {
            const float sdf = nm_h3_head_sum<1>(red, t, 0) + prm.bd;
            long long rq, oidx;
            locate(t, rq, oidx);
            if (sdf_out) sdf_out[oidx] = sdf;
            if (NABLA && nabla_out) {
                const float dsdf = nm_h3_head_sum<1>(red, 64 + t, 0) * (1.0f / NM_H2_TANGENT_SCALE);
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

// ------------------------------------------------------------------ colour MLP (reference configuration)
// cdim = 32, multires_ft = 2, multires_d = 8, multires_view = 4, nabla input (207 -> Kpad0 = 208), D >= 2.  Contract as
// nm_col_mlp_h2_kernel<true>; 128 points per workgroup.
__global__ __launch_bounds__(NM_H3_THREADS, NM_H3_LB) void nm_col_mlp_h3_kernel(
    NmColParamsH2 prm, const float* __restrict__ ft_rec, const float* __restrict__ ds, const float* __restrict__ nabla,
    const float* __restrict__ dirs, int dir_div, long long npts, float* __restrict__ rgb_out, NmSlotMap smap, int* __restrict__ overflow) {
    __shared__ __attribute__((aligned(16))) _Float16 tile[NM_H3_TILE];
    __shared__ __attribute__((aligned(16))) float cst[(NM_H2_BIAS_LAYERS + 3) * NM_W];  // biases of layers 0..3 | rgb weights [3][256]
    __shared__ float red[NM_H3_WAVES * NM_H3_ROWS * 3];
    const long long base = (long long)blockIdx.x * NM_H3_ROWS;
    if (smap.order && smap.order[base] == 0xffffu && (base + 64 >= npts || smap.order[base + 64] == 0xffffu)) return;  // no point in this tile
    const NmDivBase ddiv = nm_div_base(base, dir_div);
    long long ray0[2] = {0, 0};
    if (smap.order) {
        ray0[0] = (base / smap.E) * smap.G;
        ray0[1] = ((base + 64) / smap.E) * smap.G;
    }
    nm_phase_stamp(0);
    NmBPre3 pre;
    nm_h3_prefetch(prm.layer[0], __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), (threadIdx.x & 63) * 16, 0, pre);  // in flight during the input phase
    constexpr int cdim = 32, mft = 2, md = 8, mv = 4, FT = 160;
    constexpr int o_d = FT, o_vb = o_d + 2 * md, o_v = o_vb + 6 * mv, o_n = o_v + 3, o_ds = o_n + 3, in_dim = o_ds + 1, Kpad0 = 208;
    constexpr int ROUNDS = NM_H3_ROWS * 8 / NM_H3_THREADS;
    float in_ds[ROUNDS], in_x[ROUNDS];  // in_x: lane j < 3: view component j, 3 <= j < 6: nabla component j - 3
    float in_dv[ROUNDS][3];
    float4 in_ft[ROUNDS];
    bool in_ok[ROUNDS];
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int task = threadIdx.x + rd * NM_H3_THREADS;
        const int p = task >> 3, j = task & 7;
        const long long q = base + p;
        in_ds[rd] = in_x[rd] = 0.f;
        in_dv[rd][0] = in_dv[rd][1] = in_dv[rd][2] = 0.f;
        in_ft[rd] = make_float4(0.f, 0.f, 0.f, 0.f);
        in_ok[rd] = q < npts && nm_slot_valid(smap, q);
        if (in_ok[rd]) {
            in_ds[rd] = ds[q];
            if (j >= 3 && j < 6) in_x[rd] = nabla[q * 3 + (j - 3)];
            in_ft[rd] = *reinterpret_cast<const float4*>(ft_rec + q * cdim + 4 * j);
            long long ray;
            int unused_p;
            if (smap.order) nm_slot_ray(smap, q, ray0[(p >> 6) & 1], ray, unused_p);
            else nm_div_local(ddiv, p, ray, unused_p);
            in_dv[rd][0] = dirs[ray * 3 + 0];
            in_dv[rd][1] = dirs[ray * 3 + 1];
            in_dv[rd][2] = dirs[ray * 3 + 2];
        }
    }
    float cst_v[NM_H2_BIAS_LAYERS + 3];
    {
        const int nb = prm.D < NM_H2_BIAS_LAYERS ? prm.D : NM_H2_BIAS_LAYERS;
#pragma unroll
        for (int l = 0; l < NM_H2_BIAS_LAYERS + 3; ++l) cst_v[l] = 0.f;
        if (threadIdx.x < NM_W) {
#pragma unroll
            for (int l = 0; l < NM_H2_BIAS_LAYERS; ++l) cst_v[l] = l < nb ? prm.layer[l].b[threadIdx.x] : 0.f;
#pragma unroll
            for (int o = 0; o < 3; ++o) cst_v[NM_H2_BIAS_LAYERS + o] = prm.wrgb[o * NM_W + threadIdx.x];
        }
    }
    float mx = 0.f;
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int task = threadIdx.x + rd * NM_H3_THREADS;
        const int p = task >> 3, j = task & 7;
        _Float16* vrow = tile + p * NM_H3_STRIDE;
        if (!in_ok[rd]) {
            nm_h2_zero_cols<NM_H3_PLANE>(vrow, 0, Kpad0, j);
            continue;
        }
        const float dsv = in_ds[rd];
        const float dv[3] = {in_dv[rd][0], in_dv[rd][1], in_dv[rd][2]};
        nm_h2_embed_chunk<NM_H3_PLANE>(vrow, cdim, mft, j, in_ft[rd], mx);
        {
            float s, co;
            nm_sincos(dsv * (float)(1 << j), &s, &co);
            nm_h2_store2<NM_H3_PLANE>(vrow + o_d + 2 * j, s, co, mx);
        }
        for (int e = j; e < 3 * mv; e += 8) {  // view bands: [sin(v f_b) (3) | cos(v f_b) (3)] per band
            const int b = e / 3, dim = e - 3 * b;
            float s, co;
            nm_sincos((dim == 0 ? dv[0] : dim == 1 ? dv[1] : dv[2]) * (float)(1 << b), &s, &co);
            nm_h2_store1<NM_H3_PLANE>(vrow + o_vb + 6 * b + dim, s, mx);
            nm_h2_store1<NM_H3_PLANE>(vrow + o_vb + 6 * b + 3 + dim, co, mx);
        }
        if (j < 3) nm_h2_store1<NM_H3_PLANE>(vrow + o_v + j, j == 0 ? dv[0] : j == 1 ? dv[1] : dv[2], mx);
        else if (j < 6) nm_h2_store1<NM_H3_PLANE>(vrow + o_n + (j - 3), in_x[rd], mx);
        else if (j == 6) nm_h2_store1<NM_H3_PLANE>(vrow + o_ds, dsv, mx);
        for (int c = in_dim + j; c < Kpad0; c += 8) {
            vrow[c] = (_Float16)0.0f;
            vrow[NM_H3_PLANE + c] = (_Float16)0.0f;
        }
    }
    if (threadIdx.x < NM_W) {
#pragma unroll
        for (int l = 0; l < NM_H2_BIAS_LAYERS + 3; ++l) cst[l * NM_W + threadIdx.x] = cst_v[l];
    }
    const float* head = cst + NM_H2_BIAS_LAYERS * NM_W;
    nm_h3_layers<1, false, 3, 13, 0>(tile, prm.layer, prm.D, cst, head, red, mx, pre);
    if (threadIdx.x < NM_H3_ROWS) {  // one thread per point: its three channels are one 12-byte store
        const int p = threadIdx.x;
        const long long q = base + p;
        if (q < npts && nm_slot_valid(smap, q)) {
            long long oq = q;  // ordered lists: the colour goes back to its (ray, sample) position
            if (smap.order) {
                long long ray;
                int sp;
                nm_slot_ray(smap, q, ray0[(p >> 6) & 1], ray, sp);
                oq = ray * smap.P + sp;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float z = nm_h3_head_sum<3>(red, p, c) + prm.brgb[c];
                rgb_out[oq * 3 + c] = __fdiv_rn(1.0f, 1.0f + expf(-z));
            }
        }
    }
    nm_h2_raise(overflow, mx);
    nm_phase_stamp(15);
}
