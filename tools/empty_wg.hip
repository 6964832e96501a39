// tools/empty_wg.hip -- what does a workgroup that exits at once cost?  The list-addressed MLP launches of a frame are padded: a tile whose first list
// entry is the padding value returns immediately (csrc/nm_mlp_h2.h), and with 60 % of the mid-points dropped that is ~3.9 M such workgroups per
// frame, each dispatched with the kernel's full resources (256 threads, 72 KB of LDS, 192 registers).  hipcc --offload-arch=gfx950 -O3 empty_wg.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int LDS, int REGS>
__global__ __launch_bounds__(256, 2) void probe(const unsigned short* __restrict__ order, float* __restrict__ out, int stride) {
    __shared__ float lds[LDS / 4];
    const long long base = (long long)blockIdx.x * stride;
    if (order[base] == 0xffffu) return;
    float acc[REGS];                                       // keeps the register allocation of a real kernel
#pragma unroll
    for (int i = 0; i < REGS; ++i) acc[i] = order[base + i] * 1.0f;
    lds[threadIdx.x] = acc[threadIdx.x % REGS];
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < REGS; ++i) s += acc[i] * lds[(threadIdx.x + i) & 255];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    const int n = 1 << 21, stride = 32;
    std::vector<unsigned short> h((size_t)n * stride, 0xffff);
    unsigned short* d;
    float* o;
    hipMalloc(&d, h.size() * 2);
    hipMalloc(&o, 1 << 20);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    auto run = [&](auto kern, const char* name) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(kern, dim3(n), dim3(256), 0, 0, d, o, stride);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("%-40s %d empty workgroups: %.3f ms = %.1f ns per workgroup (chip-wide)\n", name, n, ms, ms * 1e6 / n);
        }
    };
    run(probe<73728, 150>, "256 threads, 72 KB LDS, ~190 registers");
    run(probe<4096, 150>, "256 threads, 4 KB LDS, ~190 registers");
    run(probe<4096, 16>, "256 threads, 4 KB LDS, few registers");
    return 0;
}
