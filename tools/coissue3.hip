// tools/coissue3.hip -- probe (GPU box): does vector-ALU work issued BETWEEN the MFMAs of one wave hide behind them?
// The timed loop is ONE inline-assembly block on fixed physical registers (the compiler re-packs plain C++ fmas into
// v_pk_fma_f32 and regroups them: an earlier C++ probe of this question measured that, not the hardware).
// Per loop trip: 8 x [ v_mfma_f32_32x32x16_f16 (8 independent accumulators v[0:127]) + N fillers on v[160:167] ].
//   waves per SIMD: 1 (256-thread block, 1 block per CU) or 2 (512-thread block: both waves run the same mix)
// hipcc --offload-arch=gfx950 -O3 tools/coissue3.hip -o tools/_build/coissue3
#include <hip/hip_runtime.h>
#include <cstdio>

#define FMA4 "v_fma_f32 v160, v160, v168, v169\n v_fma_f32 v161, v161, v168, v169\n v_fma_f32 v162, v162, v168, v169\n v_fma_f32 v163, v163, v168, v169\n"
#define FMA2 "v_fma_f32 v164, v164, v168, v169\n v_fma_f32 v165, v165, v168, v169\n"
#define EXP1 "v_exp_f32 v166, v166\n"
#define EXP2 "v_exp_f32 v166, v166\n v_exp_f32 v167, v167\n"
#define MIX7 "v_fma_f32 v160, v160, v168, v169\n v_exp_f32 v166, v166\n v_fma_f32 v161, v161, v168, v169\n v_add_f32 v162, v162, v169\n v_log_f32 v167, v167\n v_max_f32 v163, v163, v164\n v_mul_f32 v165, v165, v168\n"
#define MF(lo, hi) "v_mfma_f32_32x32x16_f16 v[" #lo ":" #hi "], v[140:143], v[144:147], v[" #lo ":" #hi "]\n"
#define NOFILL ""
#define DSR "ds_read_b128 v[172:175], v180\n"
#define GLD "global_load_dwordx4 v[176:179], v[182:183], off\n"
#define WAITL "s_waitcnt lgkmcnt(2)\n"
#define WAITV "s_waitcnt vmcnt(4)\n"
#define TRIP(F) MF(0, 15) F MF(16, 31) F MF(32, 47) F MF(48, 63) F MF(64, 79) F MF(80, 95) F MF(96, 111) F MF(112, 127) F
#define NOMF(F) F F F F F F F F
// dependency patterns of the real K loops (per trip: 12 MFMAs; operands alternate between two fragment sets)
#define MG(lo, hi, a, b) "v_mfma_f32_32x32x16_f16 v[" #lo ":" #hi "], v[" #a ":" #a "+3], v[" #b ":" #b "+3], v[" #lo ":" #hi "]\n"
// v3: 4 accumulators, order hi0 hi1 lo0 lo1 lo0 lo1 (same accumulator again after ONE other MFMA)
#define V3STEP(F) MF(0, 15) F MF(16, 31) F MF(32, 47) F MF(48, 63) F MF(32, 47) F MF(48, 63) F
// v3 reordered: lo0 hi0 lo1 hi1 lo0 lo1
#define V3RSTEP(F) MF(32, 47) F MF(0, 15) F MF(48, 63) F MF(16, 31) F MF(32, 47) F MF(48, 63) F
// the kernels' real mix per k-step of 6 MFMAs: 2 weight loads (1 KiB each), 4 LDS reads, optionally 3 vector instructions per MFMA
#define KSTEP_MEM MF(0, 15) GLD MF(16, 31) DSR MF(32, 47) DSR MF(48, 63) GLD MF(32, 47) DSR MF(48, 63) DSR
#define KSTEP_ALL MF(0, 15) GLD FMA2 EXP1 MF(16, 31) DSR FMA2 EXP1 MF(32, 47) DSR FMA2 EXP1 MF(48, 63) GLD FMA2 EXP1 MF(32, 47) DSR FMA2 EXP1 MF(48, 63) DSR FMA2 EXP1
#define KSTEP_VALU MF(0, 15) FMA2 EXP1 MF(16, 31) FMA2 EXP1 MF(32, 47) FMA2 EXP1 MF(48, 63) FMA2 EXP1 MF(32, 47) FMA2 EXP1 MF(48, 63) FMA2 EXP1
#define KSTEP_GLD MF(0, 15) GLD MF(16, 31) MF(32, 47) MF(48, 63) GLD MF(32, 47) MF(48, 63)
// v2: 8 accumulators, hi x4, lo x4, lo x4
#define V2STEP(F) MF(0, 15) F MF(16, 31) F MF(32, 47) F MF(48, 63) F MF(64, 79) F MF(80, 95) F MF(96, 111) F MF(112, 127) F MF(64, 79) F MF(80, 95) F MF(96, 111) F MF(112, 127) F

#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
 "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
 "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95", \
 "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127", \
 "v140","v141","v142","v143","v144","v145","v146","v147","v172","v173","v174","v175","v176","v177","v178","v179","v180","v182","v183","v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","s40","scc"

#define LOOP(BODY_)                                                                        \
    asm volatile("v_mov_b32 v140, 0x3c003c00\n v_mov_b32 v141, v140\n v_mov_b32 v142, v140\n v_mov_b32 v143, v140\n" \
                 "v_mov_b32 v144, 0\n v_mov_b32 v145, 0\n v_mov_b32 v146, 0\n v_mov_b32 v147, 0\n"                    \
                 "v_mov_b32 v168, 0x3f7fbe77\n v_mov_b32 v169, 0x3a83126f\n"                                          \
                 "v_mov_b32 v160, 1.0\n v_mov_b32 v161, 1.0\n v_mov_b32 v162, 1.0\n v_mov_b32 v163, 1.0\n v_mov_b32 v164, 1.0\n v_mov_b32 v165, 1.0\n v_mov_b32 v166, 0.5\n v_mov_b32 v167, 2.0\n" \
                 "v_mov_b32 v180, %1\n v_mov_b32 v182, %2\n v_mov_b32 v183, %3\n"      \
                 "s_mov_b32 s40, %0\n"                                                    \
                 "1:\n" BODY_ "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 1b\n s_waitcnt vmcnt(0) lgkmcnt(0)\n" \
                 : : "s"(iters), "v"(ldsaddr), "v"(galo), "v"(gahi) : CLOB);

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
    __shared__ float4 sh[1024];
    sh[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned ldsaddr = (threadIdx.x & 63) * 16;
    const unsigned long long ga = (unsigned long long)(out + (threadIdx.x & 63) * 4);
    const unsigned galo = (unsigned)ga, gahi = (unsigned)(ga >> 32);
    __syncthreads();
    const long long t0 = clock64();
    if (MODE == 20) { LOOP(V3STEP(DSR) V3STEP(DSR)) }
    if (MODE == 21) { LOOP(V3STEP(GLD) V3STEP(GLD)) }
    if (MODE == 22) { LOOP(V3STEP(DSR WAITL) V3STEP(DSR WAITL)) }
    if (MODE == 23) { LOOP(V3STEP(GLD WAITV) V3STEP(GLD WAITV)) }
    if (MODE == 24) { LOOP(V3STEP(FMA2 EXP1 DSR) V3STEP(FMA2 EXP1 GLD)) }
    if (MODE == 25) { LOOP(V2STEP(DSR)) }
    if (MODE == 30) { LOOP(KSTEP_MEM KSTEP_MEM) }
    if (MODE == 31) { LOOP(KSTEP_ALL KSTEP_ALL) }
    if (MODE == 32) { LOOP(KSTEP_VALU KSTEP_VALU) }
    if (MODE == 33) { LOOP(KSTEP_GLD KSTEP_GLD) }
    if (MODE == 0) { LOOP(TRIP(NOFILL)) }
    if (MODE == 1) { LOOP(TRIP(FMA2)) }
    if (MODE == 2) { LOOP(TRIP(FMA4)) }
    if (MODE == 3) { LOOP(TRIP(FMA4 FMA2)) }
    if (MODE == 4) { LOOP(TRIP(FMA4 FMA4)) }
    if (MODE == 5) { LOOP(TRIP(FMA4 FMA4 FMA4)) }
    if (MODE == 6) { LOOP(TRIP(EXP1)) }
    if (MODE == 7) { LOOP(TRIP(EXP2)) }
    if (MODE == 8) { LOOP(TRIP(MIX7)) }
    if (MODE == 9) { LOOP(TRIP(FMA4 FMA4 FMA4 FMA4)) }
    if (MODE == 10) { LOOP(TRIP(MIX7 MIX7)) }
    if (MODE == 13) { LOOP(V3STEP(NOFILL) V3STEP(NOFILL)) }
    if (MODE == 14) { LOOP(V3RSTEP(NOFILL) V3RSTEP(NOFILL)) }
    if (MODE == 15) { LOOP(V2STEP(NOFILL)) }
    if (MODE == 16) { LOOP(V3STEP(FMA2 EXP1) V3STEP(FMA2 EXP1)) }
    if (MODE == 17) { LOOP(V2STEP(FMA2 EXP1)) }
    if (MODE == 11) { LOOP(NOMF(FMA4 FMA4)) }
    if (MODE == 12) { LOOP(NOMF(MIX7)) }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(t1 - t0);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* name, int fillers, bool mfma, float* out, long long* cyc, int per_trip = 8) {
    const int iters = 2000;
    for (int threads = 256; threads <= 512; threads += 256) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        (void)hipDeviceSynchronize();
        static long long h[256 * 8];
        (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double t = 0; const int nw = threads / 64;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) t += (double)h[b * 8 + w];
        t /= 256.0 * nw * iters * per_trip;   // per [MFMA + fillers] group of one wave
        if (mfma) printf("%-34s %d wave/SIMD: %6.1f cycles per (MFMA + %2d fillers) of a wave = %5.1f per MFMA of the SIMD\n", name, threads / 256, t, fillers, t / (threads / 256));
        else printf("%-34s %d wave/SIMD: %6.2f cycles per filler instruction of a wave\n", name, threads / 256, t / fillers);
    }
}

int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    (void)hipMemset(cyc, 0, 256 * 8 * 8);
    run<0>("MFMA only", 0, true, out, cyc);
    run<1>("MFMA + 2 v_fma", 2, true, out, cyc);
    run<2>("MFMA + 4 v_fma", 4, true, out, cyc);
    run<3>("MFMA + 6 v_fma", 6, true, out, cyc);
    run<4>("MFMA + 8 v_fma", 8, true, out, cyc);
    run<5>("MFMA + 12 v_fma", 12, true, out, cyc);
    run<9>("MFMA + 16 v_fma", 16, true, out, cyc);
    run<6>("MFMA + 1 v_exp", 1, true, out, cyc);
    run<7>("MFMA + 2 v_exp", 2, true, out, cyc);
    run<8>("MFMA + mix7 (5 plain, 2 trans)", 7, true, out, cyc);
    run<10>("MFMA + 2 x mix7", 14, true, out, cyc);
    printf("dependency patterns of the real K loops (12 MFMAs per trip), memory instructions as fillers:\n");
    run<13>("v3 order (4 acc: h0 h1 l0 l1 l0 l1)", 0, true, out, cyc, 12);
    run<14>("v3 reordered (l0 h0 l1 h1 l0 l1)", 0, true, out, cyc, 12);
    run<15>("v2 order (8 acc)", 0, true, out, cyc, 12);
    run<16>("v3 order + 3 fillers", 3, true, out, cyc, 12);
    run<17>("v2 order + 3 fillers", 3, true, out, cyc, 12);
    run<20>("v3 order + ds_read_b128 per MFMA", 1, true, out, cyc, 12);
    run<21>("v3 order + global_load x4 per MFMA", 1, true, out, cyc, 12);
    run<22>("v3 order + ds_read + waitcnt", 2, true, out, cyc, 12);
    run<23>("v3 order + global_load + waitcnt", 2, true, out, cyc, 12);
    run<24>("v3 order + 3 valu + 1 mem", 4, true, out, cyc, 12);
    run<25>("v2 order + ds_read per MFMA", 1, true, out, cyc, 12);
    printf("the K loops' instruction mix (per 6 MFMAs: 2 x 1 KiB loads, 4 ds_read_b128, 18 vector instructions):\n");
    run<33>("MFMA + the 2 weight loads only", 0, true, out, cyc, 12);
    run<30>("MFMA + loads + LDS reads", 1, true, out, cyc, 12);
    run<32>("MFMA + 3 vector per MFMA", 3, true, out, cyc, 12);
    run<31>("MFMA + loads + LDS reads + 3 vector", 4, true, out, cyc, 12);
    run<11>("8 v_fma alone", 8, false, out, cyc);
    run<12>("mix7 alone", 7, false, out, cyc);
    return 0;
}
