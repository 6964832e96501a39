// tools/coissue2.hip -- probe (GPU box): issue cost of LDS / buffer loads / scalar instructions placed between
// independent v_mfma_f32_32x32x16_f16 of ONE wave per SIMD (the K loop's situation), batched vs interleaved.
// hipcc --offload-arch=gfx950 -O3 tools/coissue2.hip -o tools/_build/coissue2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define MFMA(acc, w, a) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, a, acc, 0, 0, 0)

// MODE 0: 12 MFMA only.  1: 4 ds_read_b128 + 4 buffer loads in a batch, then 12 MFMA.  2: the same 8 loads, one after
// each of the first 8 MFMAs.  3: only the 4 ds_reads batched.  4: only the 4 buffer loads batched.  5: 8 s_nop-like salu batched.
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters, const float* gsrc) {
    __shared__ f4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = f4{1.f * i, 0, 0, 0};
    h8 a0, w0;
    for (int i = 0; i < 8; ++i) { a0[i] = (_Float16)(threadIdx.x * 0.001f + i); w0[i] = (_Float16)(0.25f * i); }
    f16v acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f16v{0};
    __syncthreads();
    const unsigned laddr = (threadIdx.x & 63) * 16;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)gsrc, 0, 1 << 20, 0x00020000);
    f4 d[8];
    for (int i = 0; i < 8; ++i) d[i] = f4{0, 0, 0, 0};
    float s = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#define LDSRD(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[i]) : "v"(laddr), "n"(1024 * (i)))
#define BUFRD(i) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(d[i]) : "v"(laddr), "s"(rs), "n"(1024 * ((i) - 4)))
        if (MODE == 1 || MODE == 3) { LDSRD(0); LDSRD(1); LDSRD(2); LDSRD(3); }
        if (MODE == 1 || MODE == 4) { BUFRD(4); BUFRD(5); BUFRD(6); BUFRD(7); }
        if (MODE == 5) { asm volatile("s_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0"); }
        MFMA(acc[0], w0, a0); if (MODE == 2) LDSRD(0);
        MFMA(acc[1], w0, a0); if (MODE == 2) LDSRD(1);
        MFMA(acc[2], w0, a0); if (MODE == 2) LDSRD(2);
        MFMA(acc[3], w0, a0); if (MODE == 2) LDSRD(3);
        MFMA(acc[4], w0, a0); if (MODE == 2) BUFRD(4);
        MFMA(acc[5], w0, a0); if (MODE == 2) BUFRD(5);
        MFMA(acc[6], w0, a0); if (MODE == 2) BUFRD(6);
        MFMA(acc[7], w0, a0); if (MODE == 2) BUFRD(7);
        MFMA(acc[0], w0, a0);
        MFMA(acc[1], w0, a0);
        MFMA(acc[2], w0, a0);
        MFMA(acc[3], w0, a0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        for (int i = 0; i < 8; ++i) s += d[i][0];
    }
    const long long t1 = clock64();
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, long long* cyc, const float* g) {
    const int iters = 2000, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, g);
    hipDeviceSynchronize();
    static long long h[256 * 4];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double t = 0;
    for (int i = 0; i < 1024; ++i) t += (double)h[i];
    printf("%-70s %7.1f ticks per 12-MFMA step (384 = matrix pipe full)\n", name, t / 1024 / iters);
}

int main() {
    float *out, *g; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 8); hipMalloc(&g, 1 << 20); hipMemset(g, 0, 1 << 20);
    run<0>("12 MFMA", out, cyc, g);
    run<1>("4 ds_read_b128 + 4 buffer_load_dwordx4 batched, then 12 MFMA", out, cyc, g);
    run<2>("the same 8 loads, one behind each of the first 8 MFMAs", out, cyc, g);
    run<3>("4 ds_read_b128 batched, then 12 MFMA", out, cyc, g);
    run<4>("4 buffer_load_dwordx4 batched, then 12 MFMA", out, cyc, g);
    run<5>("8 s_nop batched, then 12 MFMA", out, cyc, g);
    return 0;
}
