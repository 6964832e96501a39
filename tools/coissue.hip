// tools/coissue.hip -- probe (GPU box): can a SIMD issue vector-ALU work while v_mfma_f32_32x32x16_f16 executes?
//  (a) one wave per SIMD: 8 independent MFMAs per iteration with NV independent v_fma_f32 (or v_exp_f32) behind each;
//  (b) two waves per SIMD (one 512-thread workgroup): waves 0-3 MFMA only, waves 4-7 vector ALU only, alone and together.
// hipcc --offload-arch=gfx950 -O3 tools/coissue.hip -o tools/_build/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define MFMA(acc, w, a) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, a, acc, 0, 0, 0)

template <int NV, bool TRANS>
__device__ __forceinline__ void valu_block(float (&v)[8], float m, float c) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (TRANS) v[i % 8] = __builtin_amdgcn_exp2f(v[i % 8]);
        else v[i % 8] = fmaf(v[i % 8], m, c);
    }
}

// role: 0 = this wave does MFMA (+ NV valu per MFMA), 1 = valu only (8*NV... per iteration: 8 blocks of NV), 2 = idle
template <int NV, bool TRANS>
__global__ __launch_bounds__(512, 1) void k(float* out, long long* cyc, int iters, int role_lo, int role_hi) {
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? role_lo : role_hi;
    h8 a0, w0;
    for (int i = 0; i < 8; ++i) { a0[i] = (_Float16)(threadIdx.x * 0.001f + i); w0[i] = (_Float16)(0.25f * i); }
    f16v acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f16v{0};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.001f * threadIdx.x + i;
    const float m = 0.999f + 1e-9f * threadIdx.x, c = 1e-3f;
    __syncthreads();
    const long long t0 = clock64();
    if (role == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                MFMA(acc[j], w0, a0);
                valu_block<NV, TRANS>(v, m, c);
            }
        }
    } else if (role == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) valu_block<(NV > 0 ? NV : 8), TRANS>(v, m, c);
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) { s += v[i]; for (int r = 0; r < 16; ++r) s += acc[i][r]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NV, bool TRANS>
void run(const char* name, int role_lo, int role_hi, float* out, long long* cyc) {
    const int iters = 2000, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NV, TRANS>), dim3(blocks), dim3(512), 0, 0, out, cyc, iters, role_lo, role_hi);
    hipDeviceSynchronize();
    static long long h[256 * 8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double lo = 0, hi = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi) += (double)h[b * 8 + w];
    lo /= blocks * 4.0 * iters * 8; hi /= blocks * 4.0 * iters * 8;
    printf("%-58s waves 0-3: %7.1f ticks per (MFMA|block), waves 4-7: %7.1f\n", name, lo, hi);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    printf("one wave per SIMD, per MFMA (32x32x16 f16) followed by NV independent vector instructions:\n");
    run<0, false>("  NV=0", 0, 2, out, cyc);
    run<2, false>("  NV=2 fma", 0, 2, out, cyc);
    run<4, false>("  NV=4 fma", 0, 2, out, cyc);
    run<6, false>("  NV=6 fma", 0, 2, out, cyc);
    run<8, false>("  NV=8 fma", 0, 2, out, cyc);
    run<12, false>("  NV=12 fma", 0, 2, out, cyc);
    run<1, true>("  NV=1 exp2", 0, 2, out, cyc);
    run<2, true>("  NV=2 exp2", 0, 2, out, cyc);
    run<4, true>("  NV=4 exp2", 0, 2, out, cyc);
    printf("vector ALU alone (one wave per SIMD), per block of 8 instructions:\n");
    run<8, false>("  8 fma", 2, 1, out, cyc);
    run<8, true>("  8 exp2", 2, 1, out, cyc);
    printf("two waves per SIMD: waves 0-3 MFMA only, waves 4-7 blocks of 8 vector instructions:\n");
    run<0, false>("  MFMA || 8 fma   (per MFMA | per 8-fma block)", 0, 1, out, cyc);
    run<0, true>("  MFMA || 8 exp2", 0, 1, out, cyc);
    printf("two waves per SIMD, both MFMA + NV=6 fma:\n");
    run<6, false>("  both", 0, 0, out, cyc);
    run<0, false>("  both NV=0", 0, 0, out, cyc);
    return 0;
}
