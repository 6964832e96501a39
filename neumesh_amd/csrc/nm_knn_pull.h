// nm_knn_pull.h -- pull form of the K-NN kernels (several ray chunks in flight, nm_render_cfg.overlap).  Included by nm_knn.h after
// nm_distance_body.  Device-only.
#pragma once

// ------------------------------------------------------------------------------ pull kernels (K-NN beside the MLP kernels of another ray chunk)
// The K-NN kernels are bound by vector-instruction issue, the MLP kernels by the matrix pipe, and the two pipes of a SIMD run side by side
// (MI355X_MICROARCH.md, wave scheduling).  When a call is rendered as several ray chunks on several streams, one chunk's K-NN kernels can therefore
// run UNDER another chunk's MLP kernels -- if both are resident on the same SIMDs: two MLP workgroups per CU leave 512 - 2 x 192 = 128 registers per
// SIMD lane and 16 KB (geometry) / 8 KB (colour) of LDS, room for exactly one K-NN wave per SIMD.  A grid-mapped K-NN launch never leaves that
// room: its own pending workgroups refill every slot its waves free, and an MLP workgroup (192 registers on all four SIMDs of ONE CU + 72 KB of LDS at
// once) starves until the K-NN grid is exhausted.  The pull form:
//   * workgroup = ONE wave (64 threads: its registers and its 1.25 / 3 KB of LDS are freed the moment it exits); the launch has at most
//     (SIMDs of the chip) x (waves per SIMD the kernel is compiled for) of them, and every wave draws packet indices from a counter until none are left;
//   * NmYield (one per device, shared by all streams): `wanted` = MLP launches queued or running (raised / lowered by one-thread kernels around
//     them), occ[simd] = pull waves resident on that SIMD (from HW_REG_HW_ID / HW_REG_XCC_ID);
//   * while wanted > 0 a SIMD keeps at most `cap` pull waves: the others exit before their next packet, a wave that arrives on a full SIMD exits at
//     once -- so the MLP workgroups find room within one packet's time (~0.1 ms), wherever the launch order put them; a launch never gives up
//     its last `min_alive` waves (the SIMD counts are shared by every K-NN launch in flight);
//   * with wanted == 0 the launch fills the chip like the grid-mapped form.
// Which wave evaluates which packet changes no result bit (every packet's outputs depend on its own queries only).
// MEASURED (round 5, profiles/r05_overlap_sweep.txt, 70 variants in one call, identical pixels throughout): yielding makes the frame 5-15 % SLOWER --
// one K-NN wave per SIMD beside two MLP workgroups runs at ~20 % of the kernels' full rate (its dependent chain issues an instruction every ~11 cycles
// even alone) while the MLP kernels lose issue slots to it -- so the mode is opt-in (nm_render_cfg.overlap) and the default launches are the
// grid-mapped one-wave workgroups of nm_knn.h.  Two K-NN waves per SIMD beside the MLP would need <= 64-register traversal kernels.
struct NmYield {
    int wanted;          // MLP launches that want room (queued or running)
    int pad[15];
    int occ[2048 * 4];   // pull waves per SIMD, index = nm_simd_key()
};
struct NmPull {
    unsigned long long* next;   // next[0]: packet counter of THIS launch, next[1]: its waves still at work (both zeroed by the host, stream-ordered)
    long long npackets;
    NmYield* y;                 // nullptr: never yield
    int cap;                    // pull waves a SIMD keeps while MLP launches want room
    int min_alive;              // a wave only leaves while at least this many waves of its launch stay at work: the SIMD counts are shared by every K-NN
                                // launch in flight, and a launch whose waves all sat beside another launch's must not be left without workers
};
// (xcc, se, sh, cu, simd) of the calling wave -> [0, 8192)
__device__ __forceinline__ int nm_simd_key() {
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_REG_HW_ID: simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13]
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11));   // HW_REG_XCC_ID [3:0]
    return (int)((((xcc & 7u) << 8 | ((hw >> 13) & 7u) << 5 | ((hw >> 12) & 1u) << 4 | ((hw >> 8) & 15u)) << 2) | ((hw >> 4) & 3u));
}
// true: this wave leaves (its occ entry and its share of the launch's alive count are already given back)
__device__ __forceinline__ bool nm_pull_should_leave(const NmPull& pl, int key) {
    if (!pl.y) return false;
    int leave = 0;
    if ((threadIdx.x & 63) == 0 && __hip_atomic_load(&pl.y->wanted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0 &&
        __hip_atomic_load(&pl.y->occ[key], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > pl.cap) {
        unsigned long long* alive = pl.next + 1;
        if ((long long)atomicAdd(alive, ~0ull) - 1 >= (long long)pl.min_alive) {   // (atomicAdd of -1)
            if (atomicSub(&pl.y->occ[key], 1) > pl.cap) leave = 1;
            else atomicAdd(&pl.y->occ[key], 1);
        }
        if (!leave) atomicAdd(alive, 1ull);
    }
    return __builtin_amdgcn_readfirstlane(leave) != 0;
}
__device__ __forceinline__ long long nm_pull_next(const NmPull& pl) {
    unsigned long long w = 0;
    if ((threadIdx.x & 63) == 0) w = atomicAdd(pl.next, 1ull);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(w & 0xffffffffull)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(w >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
#define NM_PULL_LOOP(BODY)                                                                           \
    const int nm_key_ = pl.y ? nm_simd_key() : 0;                                                     \
    if (pl.y && (threadIdx.x & 63) == 0) {                                                            \
        atomicAdd(&pl.y->occ[nm_key_], 1);                                                            \
        atomicAdd(pl.next + 1, 1ull);                                                                 \
    }                                                                                                 \
    for (;;) {                                                                                        \
        if (nm_pull_should_leave(pl, nm_key_)) return;                                                \
        const long long wave = nm_pull_next(pl);                                                      \
        if (wave >= pl.npackets) break;                                                               \
        BODY;                                                                                         \
    }                                                                                                 \
    if (pl.y && (threadIdx.x & 63) == 0) {                                                            \
        atomicSub(&pl.y->occ[nm_key_], 1);                                                            \
        atomicAdd(pl.next + 1, ~0ull);                                                                \
    }

template <bool CHAIN>
__global__ __launch_bounds__(64, CHAIN ? NM_KNN_WAVES_CHAIN : NM_KNN_WAVES) void nm_distance_pull_kernel(NmGridView g, NmPointSrc src, long long Q, NmPull pl,
                                                          const float* __restrict__ verts,
                                                          const float* __restrict__ indicator, float w1,
                                                          float* __restrict__ ds_out, int* __restrict__ idx32_out,
                                                          long long* __restrict__ idx64_out,
                                                          float* __restrict__ w_out, float* __restrict__ grad_out,
                                                          float* __restrict__ radius_out,
                                                          const float* __restrict__ geo_table, int gdim, float* __restrict__ fg_out,
                                                          const float* __restrict__ col_table, int cdim, float* __restrict__ ft_out) {
    NM_PULL_LOOP((nm_distance_body<CHAIN, false, 64>(g, src, Q, wave, verts, indicator, w1, ds_out, idx32_out, idx64_out, w_out, grad_out, radius_out,
                                                     geo_table, gdim, fg_out, col_table, cdim, ft_out)))
}
__global__ void nm_yield_add_kernel(NmYield* y, int delta) { atomicAdd(&y->wanted, delta); }

#ifdef NM_TESTING
// Test / measurement library only (tools/coresidency_probe.py): the PLAIN K-NN traversal -- no projection, no gathers: 62 registers, so that
// TWO of its waves fit beside two MLP workgroups on a SIMD -- in the pull form, to measure what a <= 64-register traversal kernel would get
// under the MLP kernels before anyone builds the split traversal / epilogue pair.
__device__ __forceinline__ void nm_knn_plain_body(const NmGridView& g, const NmPointSrc& src, long long Q, long long wave, long long* __restrict__ idx_out,
                                                  float* __restrict__ d2_out) {
    long long q, r;
    int p;
    const bool active = nm_lane_query(src, Q, q, r, p, 0, wave);
    float x = 0.f, y = 0.f, z = 0.f, dep = 0.f;
    if (active) nm_fetch_point(src, r, p, x, y, z, dep);
    unsigned long long kk[8];
    nm_knn_wave<8, false, 64>(g, x, y, z, active, kk);
    if (!active) return;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        idx_out[q * 8 + k] = (long long)nm_key_idx(kk[k]);
        d2_out[q * 8 + k] = nm_key_d2(kk[k]);
    }
}
__global__ __launch_bounds__(64, 8) void nm_knn_pull_kernel(NmGridView g, NmPointSrc src, long long Q, NmPull pl, long long* __restrict__ idx_out,
                                                            float* __restrict__ d2_out) {
    NM_PULL_LOOP((nm_knn_plain_body(g, src, Q, wave, idx_out, d2_out)))
}
#endif
