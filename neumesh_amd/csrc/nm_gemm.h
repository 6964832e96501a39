// nm_gemm.h -- device + launcher: the fp32 GEMM of the training path (nm_train.h), on v_mfma_f32_32x32x2_f32.
//
// The training step (models/trainer.py:75-81,186-209) needs three products per linear layer, all with fp32
// operands and fp32 accumulation (the gradients are compared with the reference's autograd):
//   forward      Y[P,out]  = X[P,in]  . W[out,in]^T            A k-contiguous, B k-contiguous
//   input grad   dX[P,in]  = dY[P,out] . W[out,in]             A k-contiguous, B n-contiguous
//   weight grad  dW[out,in] += dY[P,out]^T . X[P,in]           A m-contiguous, B n-contiguous, K = P split over workgroups
// One kernel covers them: an operand is described by (base, leading dimension, which index is contiguous).
//
// Tiling (gfx950, wave64): workgroup = 4 waves = a 128 x 128 tile of C, wave = 64 x 64 = 2 x 2 MFMA tiles (64
// accumulator registers), K in steps of NM_G_BK = 32.  Both operand tiles go through LDS as [row][k] with a row stride of
// BK + 4 floats: a lane reads 8 k-values with two ds_read_b128 (144-byte stride: 16 lanes cover all 64 banks once), the two
// lane halves of a 32x32x2 MFMA take k in [0,8) and [8,16) of a 16-wide sub-step -- the sum over k is a fixed permutation.
// Global loads of step s+1 are in flight (registers) while step s computes; two LDS buffers, one barrier per step.
// The fp32 matrix pipe takes 64 cycles per instruction, a wave issues 64 of them per step against 16 LDS reads; at K = 256
// (64 flop per byte of A + C traffic) the products of a training step sit at the balance point of the matrix pipe
// (157 TFLOP/s) and HBM: measured 80 TFLOP/s.
#pragma once

#include <hip/hip_runtime.h>

#define NM_G_BM 128
#define NM_G_BN 128
#ifndef NM_G_BK
#define NM_G_BK 32                     // 16: 40 KB of LDS, three workgroups per CU; 32: 74 KB, two -- half the barriers: GEMMs of a training step 5.05 -> 4.8 ms
#endif
#define NM_G_LS (NM_G_BK + 4)          // LDS row stride (floats): 80 / 144 bytes, conflict-free ds_read_b128 over 16 lanes
#define NM_G_TPR (NM_G_BK / 4)         // threads per tile row (k-contiguous operand)
#define NM_G_RPP (256 / NM_G_TPR)      // tile rows per pass of the workgroup
#define NM_G_NV (128 / NM_G_RPP)       // float4 per thread and operand tile

typedef float nm_gacc __attribute__((ext_vector_type(16)));

struct NmGemm {
    const float* A; long long lda; int a_kc;    // a_kc = 1: A(m,k) = A[m*lda + k];  0: A(m,k) = A[k*lda + m]
    const float* B; long long ldb; int b_kc;    // b_kc = 1: B(k,n) = B[n*ldb + k];  0: B(k,n) = B[k*ldb + n]
    float* C; long long ldc;                    // C(m,n) = C[m*ldc + n]
    long long M, N, K;
    const float* bias; long long bias_rows;     // C(m,n) += bias[n] for m < bias_rows
    int relu;                                   // C = max(C, 0)
    const float* mask; long long ldmask;        // C(m,n) = 0 where mask[m*ldmask + n] <= 0  (ReLU backward)
    int atomic;                                 // atomicAdd into C (split-K partial sums; C zero-initialised by the caller)
    long long kchunk;                           // K range of one workgroup (multiple of 16); gridDim.z chunks
};

// rows r0.. of an operand tile into registers: 128 rows x BK k = NM_G_NV float4 per thread
template <bool KC>
__device__ __forceinline__ void nm_g_fetch(const float* __restrict__ base, long long ld, long long r0, long long R, long long k0,
                                           long long K1, float4 (&v)[NM_G_NV], int t) {
#pragma unroll
    for (int i = 0; i < NM_G_NV; ++i) {
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KC) {
            const long long r = r0 + t / NM_G_TPR + NM_G_RPP * i, k = k0 + (t % NM_G_TPR) * 4;
            if (r < R && k < K1) v[i] = *reinterpret_cast<const float4*>(base + r * ld + k);
        } else {
            const long long k = k0 + (t % NM_G_BK), r = r0 + (t / NM_G_BK) * 4 + (1024 / NM_G_BK) * i;
            if (r < R && k < K1) v[i] = *reinterpret_cast<const float4*>(base + k * ld + r);
        }
    }
}

template <bool KC>
__device__ __forceinline__ void nm_g_stash(float* __restrict__ tile, const float4 (&v)[NM_G_NV], int t) {
#pragma unroll
    for (int i = 0; i < NM_G_NV; ++i) {
        if (KC) {
            *reinterpret_cast<float4*>(tile + (t / NM_G_TPR + NM_G_RPP * i) * NM_G_LS + (t % NM_G_TPR) * 4) = v[i];
        } else {
            float* p = tile + ((t / NM_G_BK) * 4 + (1024 / NM_G_BK) * i) * NM_G_LS + (t % NM_G_BK);
            p[0] = v[i].x;
            p[NM_G_LS] = v[i].y;
            p[2 * NM_G_LS] = v[i].z;
            p[3 * NM_G_LS] = v[i].w;
        }
    }
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(256) void nm_gemm_kernel(NmGemm g) {
    __shared__ float lds[2][(NM_G_BM + NM_G_BN) * NM_G_LS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, h = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    const long long m0 = (long long)blockIdx.x * NM_G_BM, n0 = (long long)blockIdx.y * NM_G_BN;
    const long long kb = (long long)blockIdx.z * g.kchunk;
    const long long ke = (kb + g.kchunk < g.K) ? kb + g.kchunk : g.K;
    nm_gacc acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 va[NM_G_NV], vb[NM_G_NV];
    if (kb < ke) {
        nm_g_fetch<AKC>(g.A, g.lda, m0, g.M, kb, ke, va, t);
        nm_g_fetch<BKC>(g.B, g.ldb, n0, g.N, kb, ke, vb, t);
        nm_g_stash<AKC>(lds[0], va, t);
        nm_g_stash<BKC>(lds[0] + NM_G_BM * NM_G_LS, vb, t);
    }
    __syncthreads();
    int buf = 0;
    for (long long k = kb; k < ke; k += NM_G_BK) {
        const bool more = k + NM_G_BK < ke;
        if (more) {
            nm_g_fetch<AKC>(g.A, g.lda, m0, g.M, k + NM_G_BK, ke, va, t);
            nm_g_fetch<BKC>(g.B, g.ldb, n0, g.N, k + NM_G_BK, ke, vb, t);
        }
#pragma unroll
        for (int ks = 0; ks < NM_G_BK; ks += 16) {
            const float* as = lds[buf] + (64 * wm + li) * NM_G_LS + ks + 8 * h;
            const float* bs = lds[buf] + (NM_G_BM + 64 * wn + li) * NM_G_LS + ks + 8 * h;
            float a[2][8], b[2][8];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 x0 = *reinterpret_cast<const float4*>(as + 32 * i * NM_G_LS), x1 = *reinterpret_cast<const float4*>(as + 32 * i * NM_G_LS + 4);
                const float4 y0 = *reinterpret_cast<const float4*>(bs + 32 * i * NM_G_LS), y1 = *reinterpret_cast<const float4*>(bs + 32 * i * NM_G_LS + 4);
                a[i][0] = x0.x; a[i][1] = x0.y; a[i][2] = x0.z; a[i][3] = x0.w; a[i][4] = x1.x; a[i][5] = x1.y; a[i][6] = x1.z; a[i][7] = x1.w;
                b[i][0] = y0.x; b[i][1] = y0.y; b[i][2] = y0.z; b[i][3] = y0.w; b[i][4] = y1.x; b[i][5] = y1.y; b[i][6] = y1.z; b[i][7] = y1.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][e], b[0][e], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][e], b[1][e], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][e], b[0][e], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][e], b[1][e], acc[1][1], 0, 0, 0);
            }
        }
        if (more) {
            nm_g_stash<AKC>(lds[buf ^ 1], va, t);
            nm_g_stash<BKC>(lds[buf ^ 1] + NM_G_BM * NM_G_LS, vb, t);
        }
        __syncthreads();
        buf ^= 1;
    }
    if (kb >= ke) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long n = n0 + 64 * wn + 32 * j + li;
            if (n >= g.N) continue;
            const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= g.M) continue;
                float c = acc[i][j][r];
                if (m < g.bias_rows) c += bv;
                if (g.relu) c = fmaxf(c, 0.f);
                if (g.mask && !(g.mask[m * g.ldmask + n] > 0.f)) c = 0.f;
                if (g.atomic) atomicAdd(g.C + m * g.ldc + n, c);
                else g.C[m * g.ldc + n] = c;
            }
        }
}

// ------------------------------------------------------------------------------------------------ bf16 x 3 form
// The same product on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16: 16 x the fp32 pipe's rate) with fp32-grade operands: every fp32
// value is cut into three bf16 pieces BY TRUNCATION, a = b1 + b2 + b3 EXACTLY (8 + 8 + 8 significant bits; bf16 has fp32's exponent, so
// cotangents of 1e-9 need no scaling -- an f16 split would), and a product is the six piece products of weight >= 2^-16,
//   a.b = a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a3 b1 + a2 b2)      [dropped: a2 b3 + a3 b2 + a3 b3 < 2^-21 |a||b| in the worst case of
//                                                                  truncated pieces (|a2| < 2^-7 |a|, |a3| < 2^-15 |a|), ~2^-23 typically]
// accumulated in fp32 by the matrix pipe.  6 MFMAs per 16-k step and tile instead of 8 on a pipe that is 16 x faster.
// Operand tiles are split ONCE, when they are stashed: three bf16 planes [row][k] per operand in LDS (row stride 24 halves = 48 bytes:
// conflict-free ds_read_b128), K in steps of 16, two buffers: 73.7 KB, two workgroups per CU as before.
#define NM_G3_BK 16
#define NM_G3_LS 24
#ifndef NM_G3_DEPTH
#define NM_G3_DEPTH 2                  // steps whose operand rows are in flight (512 threads: 2 -> 70 us, 4 -> 73, 6 -> 77 per [65536,256]x[256,256];
                                       // 256 threads needed 4: there the ring hides the latency, here four waves per SIMD do)
#endif
#ifndef NM_G3_THREADS
#define NM_G3_THREADS 512              // 256: wave = 64 x 64 of the tile (2 x 2 MFMA tiles); 512: wave = 32 x 64 (1 x 2), four waves per SIMD
#endif
#define NM_G3_NV (512 / NM_G3_THREADS)   // float4 per thread and operand tile (128 rows x 16 k / 4 / threads)
#define NM_G3_RT (512 / NM_G3_THREADS)   // row tiles (32 rows) per wave
#ifndef NM_G3_PRODUCTS
#define NM_G3_PRODUCTS 6                 // piece products per fp32 product: 6, or 8 (+ a2 b3, a3 b2): measured 73 vs 82 us per product with the SAME error
                                         // against float64 (7.5e-7 of max |C| at K = 256: the fp32 accumulation of the sum dominates, not the dropped terms)
#endif
typedef __bf16 nm_bf8 __attribute__((ext_vector_type(8)));

// (loads are UNCONDITIONAL, from a clamped address, and masked when they are consumed: a branch around a load, or a select right behind
//  it, makes the compiler drain the whole ring of loads in flight)
template <bool KC>
__device__ __forceinline__ void nm_g3_fetch(const float* __restrict__ base, long long ld, long long r0, long long R, long long k0,
                                            long long K1, float4 (&v)[NM_G3_NV], int t) {
#pragma unroll
    for (int i = 0; i < NM_G3_NV; ++i) {
        long long r, k;
        if (KC) { r = r0 + t / 4 + (NM_G3_THREADS / 4) * i; k = k0 + (t % 4) * 4; }
        else { k = k0 + (t % NM_G3_BK); r = r0 + (t / NM_G3_BK) * 4 + (NM_G3_THREADS / 4) * i; }
        const bool ok = r < R && k < K1;
        const long long rc = ok ? r : 0, kc = ok ? k : 0;
        v[i] = *reinterpret_cast<const float4*>(KC ? base + rc * ld + kc : base + kc * ld + rc);
    }
}

// one value -> the upper halves of three floats whose sum is the value
__device__ __forceinline__ void nm_g3_cut(float a, unsigned& b1, unsigned& b2, unsigned& b3) {
    b1 = __float_as_uint(a) & 0xffff0000u;
    const float r1 = a - __uint_as_float(b1);          // exact
    b2 = __float_as_uint(r1) & 0xffff0000u;
    b3 = __float_as_uint(r1 - __uint_as_float(b2));    // exact, <= 8 significant bits: its upper half holds all of it
}

template <bool KC>
__device__ __forceinline__ void nm_g3_stash(unsigned short* __restrict__ tile, int plane_stride, const float4 (&v)[NM_G3_NV], int t,
                                            long long r0, long long R, long long k0, long long K1) {
#pragma unroll
    for (int i = 0; i < NM_G3_NV; ++i) {
        const int row = KC ? t / 4 + (NM_G3_THREADS / 4) * i : (t / NM_G3_BK) * 4 + (NM_G3_THREADS / 4) * i;
        const int col = KC ? (t % 4) * 4 : t % NM_G3_BK;
        const bool ok = r0 + row < R && k0 + col < K1;
        unsigned b[3][4];
        nm_g3_cut(ok ? v[i].x : 0.f, b[0][0], b[1][0], b[2][0]);
        nm_g3_cut(ok ? v[i].y : 0.f, b[0][1], b[1][1], b[2][1]);
        nm_g3_cut(ok ? v[i].z : 0.f, b[0][2], b[1][2], b[2][2]);
        nm_g3_cut(ok ? v[i].w : 0.f, b[0][3], b[1][3], b[2][3]);
        unsigned short* p = tile + row * NM_G3_LS + col;
        if (KC) {   // 4 consecutive k of one row: one 8-byte store per plane
#pragma unroll
            for (int q = 0; q < 3; ++q)
                *reinterpret_cast<uint2*>(p + q * plane_stride) = make_uint2((b[q][0] >> 16) | (b[q][1] & 0xffff0000u), (b[q][2] >> 16) | (b[q][3] & 0xffff0000u));
        } else {    // 4 consecutive rows at one k
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) p[q * plane_stride + e * NM_G3_LS] = (unsigned short)(b[q][e] >> 16);
        }
    }
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(NM_G3_THREADS) void nm_gemm3_kernel(NmGemm g) {
    constexpr int PLANE = (NM_G_BM + NM_G_BN) * NM_G3_LS;     // halves per plane: A rows, then B rows
    __shared__ __attribute__((aligned(16))) unsigned short lds[2][3 * PLANE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, h = lane >> 5;
    const int wm = wave % (128 / (32 * NM_G3_RT)), wn = wave / (128 / (32 * NM_G3_RT));   // the wave's rows 32 RT wm .., columns 64 wn ..
    const long long m0 = (long long)blockIdx.x * NM_G_BM, n0 = (long long)blockIdx.y * NM_G_BN;
    const long long kb = (long long)blockIdx.z * g.kchunk;
    const long long ke = (kb + g.kchunk < g.K) ? kb + g.kchunk : g.K;
    if (kb >= ke) return;
    nm_gacc acc[NM_G3_RT][2];
#pragma unroll
    for (int i = 0; i < NM_G3_RT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // The matrix-pipe time of a 16-k step (0.2 - 0.4 us per wave) is far below the latency of a global load, so the operand rows of the next
    // NM_G3_DEPTH steps are kept in flight in a ring of registers (one step ahead, as in the fp32 kernel, left this kernel at the load
    // latency per step).
    float4 ra[NM_G3_DEPTH][NM_G3_NV], rb[NM_G3_DEPTH][NM_G3_NV];
#pragma unroll
    for (int u = 0; u < NM_G3_DEPTH; ++u) {
        nm_g3_fetch<AKC>(g.A, g.lda, m0, g.M, kb + u * NM_G3_BK, ke, ra[u], t);
        nm_g3_fetch<BKC>(g.B, g.ldb, n0, g.N, kb + u * NM_G3_BK, ke, rb[u], t);
    }
    int buf = 0;
    for (long long k = kb; k < ke; k += NM_G3_DEPTH * NM_G3_BK) {
#pragma unroll
        for (int u = 0; u < NM_G3_DEPTH; ++u) {
            const long long ks = k + u * NM_G3_BK;
            if (ks >= ke) break;   // (uniform)
            // a wave that is a step ahead writes the OTHER buffer; nobody is two steps ahead (one barrier per step)
            nm_g3_stash<AKC>(lds[buf], PLANE, ra[u], t, m0, g.M, ks, ke);
            nm_g3_stash<BKC>(lds[buf] + NM_G_BM * NM_G3_LS, PLANE, rb[u], t, n0, g.N, ks, ke);
            nm_g3_fetch<AKC>(g.A, g.lda, m0, g.M, ks + NM_G3_DEPTH * NM_G3_BK, ke, ra[u], t);
            nm_g3_fetch<BKC>(g.B, g.ldb, n0, g.N, ks + NM_G3_DEPTH * NM_G3_BK, ke, rb[u], t);
            __syncthreads();
            const unsigned short* as = lds[buf] + (32 * NM_G3_RT * wm + li) * NM_G3_LS + 8 * h;
            const unsigned short* bs = lds[buf] + (NM_G_BM + 64 * wn + li) * NM_G3_LS + 8 * h;
            nm_bf8 a[NM_G3_RT][3], b[2][3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int i = 0; i < NM_G3_RT; ++i) a[i][q] = *reinterpret_cast<const nm_bf8*>(as + q * PLANE + 32 * i * NM_G3_LS);
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j][q] = *reinterpret_cast<const nm_bf8*>(bs + q * PLANE + 32 * j * NM_G3_LS);
            }
#pragma unroll
            for (int i = 0; i < NM_G3_RT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {   // small terms first
#if NM_G3_PRODUCTS == 8
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][1], acc[i][j], 0, 0, 0);
#endif
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                }
            buf ^= 1;
        }
    }
#pragma unroll
    for (int i = 0; i < NM_G3_RT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long n = n0 + 64 * wn + 32 * j + li;
            if (n >= g.N) continue;
            const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + 32 * NM_G3_RT * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= g.M) continue;
                float c = acc[i][j][r];
                if (m < g.bias_rows) c += bv;
                if (g.relu) c = fmaxf(c, 0.f);
                if (g.mask && !(g.mask[m * g.ldmask + n] > 0.f)) c = 0.f;
                if (g.atomic) atomicAdd(g.C + m * g.ldc + n, c);
                else g.C[m * g.ldc + n] = c;
            }
        }
}

// C = A . B with the operand layouts of `g`; split_k > 1: K is cut into that many chunks whose partial products are added
// atomically (C must have been zeroed, or hold the value to accumulate onto).
// bf16x3 = true: the bf16 x 3 kernel above; false: the fp32-pipe kernel.
// Shape contract of the bf16 x 3 kernel: it moves and masks operands as float4 along each operand's contiguous axis, testing the
// FIRST element of the four only -- so that axis' extent must be a multiple of 4: K for a k-contiguous operand (a_kc / b_kc), M (A) or N (B)
// otherwise.  The training path pads accordingly (K0p, Kc0p, W % 16); anything else is refused here instead of being read out of bounds.
static inline bool nm_gemm3_shape_ok(const NmGemm& g) {
    return (g.a_kc ? g.K % 4 == 0 : g.M % 4 == 0) && (g.b_kc ? g.K % 4 == 0 : g.N % 4 == 0);
}
static inline int nm_gemm_launch(NmGemm g, int split_k, hipStream_t stream, bool bf16x3 = false) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return 0;
    if (bf16x3 && !nm_gemm3_shape_ok(g)) return 2;
    long long chunks = split_k > 1 ? split_k : 1;
    long long kchunk = ((g.K + chunks - 1) / chunks + NM_G_BK - 1) / NM_G_BK * NM_G_BK;
    chunks = (g.K + kchunk - 1) / kchunk;
    g.kchunk = kchunk;
    if (chunks > 1) g.atomic = 1;
    const dim3 grid((unsigned)((g.M + NM_G_BM - 1) / NM_G_BM), (unsigned)((g.N + NM_G_BN - 1) / NM_G_BN), (unsigned)chunks);
    if (bf16x3) {
        if (g.a_kc && g.b_kc) hipLaunchKernelGGL((nm_gemm3_kernel<true, true>), grid, dim3(NM_G3_THREADS), 0, stream, g);
        else if (g.a_kc && !g.b_kc) hipLaunchKernelGGL((nm_gemm3_kernel<true, false>), grid, dim3(NM_G3_THREADS), 0, stream, g);
        else if (!g.a_kc && g.b_kc) hipLaunchKernelGGL((nm_gemm3_kernel<false, true>), grid, dim3(NM_G3_THREADS), 0, stream, g);
        else hipLaunchKernelGGL((nm_gemm3_kernel<false, false>), grid, dim3(NM_G3_THREADS), 0, stream, g);
        return hipGetLastError() == hipSuccess ? 0 : 1;
    }
    if (g.a_kc && g.b_kc) hipLaunchKernelGGL((nm_gemm_kernel<true, true>), grid, dim3(256), 0, stream, g);
    else if (g.a_kc && !g.b_kc) hipLaunchKernelGGL((nm_gemm_kernel<true, false>), grid, dim3(256), 0, stream, g);
    else if (!g.a_kc && g.b_kc) hipLaunchKernelGGL((nm_gemm_kernel<false, true>), grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((nm_gemm_kernel<false, false>), grid, dim3(256), 0, stream, g);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
