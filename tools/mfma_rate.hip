// tools/mfma_rate.hip -- probe (GPU box): issue rate of v_mfma_f32_32x32x16_f16 in the accumulator pattern
// of the split-half K loop (per k-step: hi x4, lo x4, lo x4 on 8 accumulators), with register operands
// only, 1 or 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/_build/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, long long* cyc, int iters) {
    h8 a0, a1, b0, b1, w0, w1, w2, w3;
    for (int i = 0; i < 8; ++i) {
        a0[i] = (_Float16)(threadIdx.x * 0.001f + i); a1[i] = (_Float16)(i * 0.5f); b0[i] = (_Float16)(i + 1); b1[i] = (_Float16)(2 * i);
        w0[i] = (_Float16)(0.25f * i); w1[i] = (_Float16)(0.125f * i); w2[i] = (_Float16)(0.5f); w3[i] = (_Float16)(1.5f);
    }
    f16v hi[4] = {{0}, {0}, {0}, {0}}, lo[4] = {{0}, {0}, {0}, {0}};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // kernel pattern: rt x ct = 4 tiles; hi: A0 x Wa, lo: A0 x Wb, lo: A1 x Wa
            hi[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, a0, hi[0], 0, 0, 0);
            hi[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, a0, hi[1], 0, 0, 0);
            hi[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, b0, hi[2], 0, 0, 0);
            hi[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, b0, hi[3], 0, 0, 0);
            lo[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, a0, lo[0], 0, 0, 0);
            lo[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3, a0, lo[1], 0, 0, 0);
            lo[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, b0, lo[2], 0, 0, 0);
            lo[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3, b0, lo[3], 0, 0, 0);
            lo[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, a1, lo[0], 0, 0, 0);
            lo[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, a1, lo[1], 0, 0, 0);
            lo[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, b1, lo[2], 0, 0, 0);
            lo[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, b1, lo[3], 0, 0, 0);
        } else if (MODE == 1) {  // 12 MFMAs round-robin over all 8 accumulators (max dependency distance)
            hi[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, a0, hi[0], 0, 0, 0);
            lo[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, a0, lo[0], 0, 0, 0);
            hi[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, a0, hi[1], 0, 0, 0);
            lo[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3, a0, lo[1], 0, 0, 0);
            hi[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, b0, hi[2], 0, 0, 0);
            lo[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, b0, lo[2], 0, 0, 0);
            hi[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, b0, hi[3], 0, 0, 0);
            lo[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3, b0, lo[3], 0, 0, 0);
            lo[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, a1, lo[0], 0, 0, 0);
            lo[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, a1, lo[1], 0, 0, 0);
            lo[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, b1, lo[2], 0, 0, 0);
            lo[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, b1, lo[3], 0, 0, 0);
        } else {  // MODE 2: one accumulator, fully dependent chain (latency)
            for (int j = 0; j < 12; ++j) hi[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, a0, hi[0], 0, 0, 0);
        }
        // keep operands changing a little so nothing is hoisted
        a0[0] += (_Float16)0.001f;
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += hi[i][r] + lo[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int iters = 2000;
    float* out; long long* cyc;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 4096 * 8);
    for (int mode = 0; mode < 3; ++mode)
        for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
            const int blocks = 256 * wg_per_cu;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[512]; hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks;
            const double mf = (double)iters * 12;
            printf("mode %d, %d WG(4 waves)/CU: %.1f clock64 ticks per MFMA per wave (wall %.3f ms -> %.1f TFLOP/s)\n", mode, wg_per_cu,
                   avg / mf, ms, blocks * 4.0 * mf * 32768.0 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
