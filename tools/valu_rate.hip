// tools/valu_rate.hip -- probe (GPU box): issue cost (cycles per instruction per SIMD) of the vector instructions the
// K-NN traversal is made of, measured with 1 and with 6 waves per SIMD (64 independent instances per loop trip).
// hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/_build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, long long* cyc, int iters) {
    unsigned long long a = threadIdx.x * 0x9E3779B97F4A7C15ull, b = a ^ 0x1234567ull, c = b + 77;
    float f0 = threadIdx.x * 0.5f, f1 = 1.0f, f2 = 2.0f, f3 = 0.25f;
    unsigned u0 = threadIdx.x, u1 = 3, u2 = 5;
    unsigned long long m = 0;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) asm volatile(REP64("v_fma_f32 %0, %1, %2, %0\n") : "+v"(f0) : "v"(f1), "v"(f2));
        if (MODE == 1) asm volatile(REP64("v_cmp_lt_u64_e64 %0, %1, %2\n") : "=s"(m) : "v"(a), "v"(b));
        if (MODE == 2) asm volatile(REP64("v_cmp_lt_u32_e64 %0, %1, %2\n") : "=s"(m) : "v"(u0), "v"(u1));
        if (MODE == 3) asm volatile(REP64("v_mov_b64 %0, %1\n") : "=v"(c) : "v"(a));
        if (MODE == 4) asm volatile(REP64("v_cndmask_b32_e64 %0, %1, %2, %3\n") : "=v"(u2) : "v"(u0), "v"(u1), "s"((unsigned long long)iters | 0xf0f0ull));
        if (MODE == 5) asm volatile(REP64("v_pk_mul_f32 %0, %1, %1\n") : "=v"(c) : "v"(a));
        if (MODE == 6) asm volatile(REP64("v_max3_f32 %0, %1, %2, %0\n") : "+v"(f0) : "v"(f1), "v"(f2));
        if (MODE == 7) asm volatile(REP8("s_mov_b64 exec, %1\n v_mov_b64 %0, %2\n v_mov_b64 %0, %2\n") "s_mov_b64 exec, -1\n" : "=v"(c) : "s"((unsigned long long)iters | 0xf0f1ull), "v"(a));
        if (MODE == 8) asm volatile(REP8("v_cmp_lt_u64_e64 %0, %1, %2\n s_and_b64 %0, %0, exec\n s_cbranch_scc0 1f\n1:\n") : "=s"(m) : "v"(a), "v"(b) : "scc");
        if (MODE == 9) asm volatile(REP64("v_mov_b32 %0, %1\n") : "=v"(u2) : "v"(u0));
        if (MODE == 10) asm volatile(REP64("v_sub_f32 %0, %1, %2\n") : "=v"(f3) : "v"(f1), "v"(f2));
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + m + (unsigned long long)(f0 + f3) + u2;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_trip, unsigned long long* out, long long* cyc) {
    const int iters = 500;
    double res[2];
    for (int w = 0; w < 2; ++w) {
        const int blocks = 256 * (w ? 6 : 1);  // 4 waves per block, one per SIMD: 1 or 6 waves per SIMD
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        (void)hipDeviceSynchronize();
        static long long h[256 * 6 * 4];
        (void)hipMemcpy(h, cyc, blocks * 4 * sizeof(long long), hipMemcpyDeviceToHost);
        double t = 0;
        for (int i = 0; i < blocks * 4; ++i) t += (double)h[i];
        res[w] = t / (blocks * 4) / iters / per_trip / (w ? 6 : 1);
    }
    printf("%-62s %6.2f cycles/instr alone, %6.2f per SIMD with 6 waves\n", name, res[0], res[1]);
}

int main() {
    unsigned long long* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 6 * 256 * 8); (void)hipMalloc(&cyc, 256 * 6 * 4 * 8);
    run<0>("v_fma_f32 (dependent chain)", 64, out, cyc);
    run<10>("v_sub_f32 (independent)", 64, out, cyc);
    run<9>("v_mov_b32", 64, out, cyc);
    run<1>("v_cmp_lt_u64 -> sgpr", 64, out, cyc);
    run<2>("v_cmp_lt_u32 -> sgpr", 64, out, cyc);
    run<3>("v_mov_b64", 64, out, cyc);
    run<4>("v_cndmask_b32 (sgpr mask)", 64, out, cyc);
    run<5>("v_pk_mul_f32", 64, out, cyc);
    run<6>("v_max3_f32 (dependent chain)", 64, out, cyc);
    run<7>("[s_mov exec + 2 v_mov_b64] (per group of 3)", 8, out, cyc);
    run<8>("[v_cmp_lt_u64 -> s_and -> s_cbranch] (per group of 3)", 8, out, cyc);
    return 0;
}
