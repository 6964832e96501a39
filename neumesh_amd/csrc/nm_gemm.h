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

// C = A . B with the operand layouts of `g`; split_k > 1: K is cut into that many chunks whose partial products are added
// atomically (C must have been zeroed, or hold the value to accumulate onto).
static inline int nm_gemm_launch(NmGemm g, int split_k, hipStream_t stream) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return 0;
    long long chunks = split_k > 1 ? split_k : 1;
    long long kchunk = ((g.K + chunks - 1) / chunks + NM_G_BK - 1) / NM_G_BK * NM_G_BK;
    chunks = (g.K + kchunk - 1) / kchunk;
    g.kchunk = kchunk;
    if (chunks > 1) g.atomic = 1;
    const dim3 grid((unsigned)((g.M + NM_G_BM - 1) / NM_G_BM), (unsigned)((g.N + NM_G_BN - 1) / NM_G_BN), (unsigned)chunks);
    if (g.a_kc && g.b_kc) hipLaunchKernelGGL((nm_gemm_kernel<true, true>), grid, dim3(256), 0, stream, g);
    else if (g.a_kc && !g.b_kc) hipLaunchKernelGGL((nm_gemm_kernel<true, false>), grid, dim3(256), 0, stream, g);
    else if (!g.a_kc && g.b_kc) hipLaunchKernelGGL((nm_gemm_kernel<false, true>), grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((nm_gemm_kernel<false, false>), grid, dim3(256), 0, stream, g);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
