// nm_mlp_r.h -- fused embed + MLP kernels on the f16 matrix pipe, REGISTER-RESIDENT activations
// (split-half operands exactly as nm_mlp_h2.h: a = h1 + h2 * 2^-11, three f16 MFMAs per fp32 product).
//
// What bounded the LDS-tile kernels of nm_mlp_h2.h (round-2 measurements): every wave streamed its
// 64 output columns of every layer from L1/L2 into VGPRs (341 B of weight fragments per MFMA and wave -- the
// L1 fill rate co-limits the K loops), every layer ended in a workgroup barrier + an LDS round trip of the
// activations, and the vector epilogues only overlapped matrix work through a second co-resident workgroup.
// Here the roles are re-cut around the MFMA operand layouts themselves:
//
//  * A wave owns 32 POINTS (the N dimension of v_mfma_f32_32x32x16_f16: D = W_tile[32 out x 16 k] * X[16 k x 32 points])
//    and ALL 256 columns of every layer.  The weight rows of a 32-column tile are packed so that the 16 results
//    a lane receives (one point, MFMA rows (r&3) + 8(r>>2) + 4h) are exactly the 2 x 8 input features that lane
//    must hold as B operand of the NEXT layer's k-steps 2t and 2t+1 (features 32t + 8h + r and 32t + 16 + 8h + r):
//    an activation goes accumulator -> softplus/ReLU -> split halves -> B operand register of the next layer
//    without leaving the lane.  No activation tile in LDS, no barrier between layers, no bank conflicts.
//  * The weights of a layer are the same for all waves: they are streamed ONCE per workgroup through a ring of
//    3 x 32 KB LDS slots by LDS-DMA (global_load_lds_dwordx4, 8 KB per wave and chunk) and every wave reads its
//    A fragments from there (ds_read_b128, 2 KB per 3 MFMAs and wave = a third of the LDS read rate).  A chunk =
//    one 32-column tile, all k-steps, both planes; one s_barrier per chunk hands the slot over.
//  * One wave per SIMD (the kernel lives in the 512-register file: 128 B-operand registers in, 128 out, two
//    accumulator sets of 32); the epilogue of a column tile is issued inside the K loop of the NEXT tile (its
//    vector instructions sit in the shadow of that tile's MFMAs; the last tile of a layer runs beside the first
//    tile of the next layer).
//  * Layer 0: every lane builds its own B operands (8 features per k-step) from the K-NN record of its point:
//    the physical input order is [code 32 | sin | cos | sin 2x | cos 2x (32 each) | (sin, cos) of ds 2^b | ds ...],
//    folded into the packed weights.
//  * Value + tangent (nabla) kernel: the 32 columns of a wave are 16 points and their 16 tangent columns
//    t = d h / d ds; the activation derivative crosses from lane n to lane n + 16 with v_permlane16_swap.
//
// Only the reference configuration is built this way (32-d codes, 2 / 8 / 4 embedding bands, W = 256, 3 geometry
// and 4 colour layers: configs/neumesh_dtu_scan63.yaml); anything else runs the kernels of nm_mlp_h2.h.
// Numerics: the same split-half arithmetic and k order as nm_mlp_h2.h (value rows bit-identical per layer); the
// output projection sums in another order (per-lane partial sums + one cross-lane add).
// Reference semantics: models/frameworks/neumesh/neumesh.py:204-260, models/base.py:52-70.
#pragma once

#include "nm_mlp_h2.h"

#define NM_R_THREADS 256
#define NM_R_SLOT 32768          // bytes per ring slot = the largest chunk (16 k-steps x 2 planes x 1 KiB)
#define NM_R_SLOTS 3
#ifndef NM_R_VALU_PER_MFMA
#define NM_R_VALU_PER_MFMA 5     // vector instructions scheduled behind each MFMA of the K loops
#endif
#ifndef NM_R_EXP
#define NM_R_EXP 0   // timing knock-outs (wrong results): 1 no activations, 2 no DMA / barriers, 4 no fragment loads, 8 no layers
#endif
#ifndef NM_R_PF
#define NM_R_PF 2                // A-fragment prefetch distance in k-steps (NM_R_PF + 1 fragment register sets)
#endif
#define NM_R_KSTEP_BYTES 2048    // one k-step of a chunk: [plane 2][lane 64][8 halves]

struct NmLayerR {
    const _Float16* W;  // [tile 8][k-step KS][plane 2][lane 64][8 halves]; lane = MFMA row i | (k-half << 5)
    const float* b;     // [tile 8][h 2][r 16] fp32: bias of the feature register r of lane-half h holds after tile t
};
struct NmGeoParamsR {
    NmLayerR layer[3];
    const float* wd;    // density weights in register order [tile 8][h 2][r 16], x 1/S (log2 units, nm_mlp_h2.h)
    float bd;
};
struct NmColParamsR {
    NmLayerR layer[4];
    const float* wrgb;  // [3][tile 8][h 2][r 16]
    float brgb[3];
};

// output feature held in register r (0..15) of lane-half h after column tile t: the B-operand slot it becomes
__host__ __device__ __forceinline__ int nm_r_feature(int t, int h, int r) { return 32 * t + (r < 8 ? 8 * h + r : 16 + 8 * h + (r - 8)); }

// weights: fp32 [256][in_dim] (PyTorch layout, logical input columns) -> split halves in the chunk order above.
// perm (device, [16 * KS] ints or nullptr): physical input column -> logical column (-1: zero padding).
__global__ void nm_pack_weight_r_kernel(const float* __restrict__ src, int in_dim, int KS, const int* __restrict__ perm, float scale,
                                        _Float16* __restrict__ dst) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (output feature n, physical column kp)
    const int Kpad = 16 * KS;
    if (e >= NM_W * Kpad) return;
    const int n = e / Kpad, kp = e - n * Kpad;
    const int kl = perm ? perm[kp] : kp;
    const float w = (kl >= 0 && kl < in_dim) ? src[(size_t)n * in_dim + kl] * scale : 0.f;
    _Float16 h1, h2;
    nm_split_half(w, &h1, &h2);
    const int t = n >> 5, m = n & 31;
    const int h = m < 16 ? (m >> 3) : ((m - 16) >> 3);
    const int r = m < 16 ? (m & 7) : 8 + ((m - 16) & 7);
    const int i = (r & 3) + 8 * (r >> 2) + 4 * h;                  // MFMA row of the 32-column tile
    const int lane = i | (((kp >> 3) & 1) << 5), ks = kp >> 4, el = kp & 7;
    const size_t o = ((size_t)t * KS + ks) * 2 * 64 * 8;
    dst[o + (size_t)lane * 8 + el] = h1;
    dst[o + 64 * 8 + (size_t)lane * 8 + el] = h2;
}
// fp32 [256] (per output feature) -> register order [tile 8][h 2][r 16]
__global__ void nm_pack_vec_r_kernel(const float* __restrict__ src, float scale, float* __restrict__ dst) {
    const int e = threadIdx.x;  // 256 threads
    const int t = e >> 5, h = (e >> 4) & 1, r = e & 15;
    dst[e] = src[nm_r_feature(t, h, r)] * scale;
}

// ------------------------------------------------------------------------------------------ chunk schedule
// Chunk s of a kernel = column tile s % 8 of layer s / 8: KS_l k-steps of 2 KiB (KS even: the four waves copy KS / 2 KiB each).
template <int KS0, int D>
struct NmRSched {
    static constexpr int S = 8 * D;
    __host__ __device__ static constexpr int ks_of(int l) { return l == 0 ? KS0 : 16; }
    __host__ __device__ static constexpr int nks(int s) { return ks_of(s / 8); }
    __host__ __device__ static constexpr int dma_ops(int s) { return s < S ? nks(s) / 2 : 0; }   // 1 KiB LDS-DMA pieces per wave
    __host__ __device__ static constexpr size_t offset_halves(int s) { return (size_t)(s % 8) * ks_of(s / 8) * (NM_R_KSTEP_BYTES / 2); }
    __host__ __device__ static constexpr int ksteps_before(int s) { return s < 8 ? s * KS0 : 8 * KS0 + (s - 8) * 16; }
};

typedef __attribute__((address_space(1))) const void* nm_gptr;
typedef __attribute__((address_space(3))) void* nm_lptr;

// this wave's quarter of a chunk: NPC pieces of 1 KiB, global -> LDS slot (lane-linear image = the order the fragments are read in)
template <int NPC>
__device__ __forceinline__ void nm_r_dma(const _Float16* chunk, char* slot, int wave, int lane) {
    const char* g = reinterpret_cast<const char*>(chunk) + wave * (NPC * 1024) + lane * 16;
    const unsigned l = (unsigned)(size_t)(nm_lptr)slot + wave * (NPC * 1024);   // LDS byte address of this wave's share (uniform)
    // Issued as inline assembly: through the builtin the compiler knows the instruction writes LDS and puts s_waitcnt vmcnt(0)
    // in front of the next ds_read -- every wave would sit out its own copy right after issuing it.  The hand-over protocol
    // of nm_r_layer (counted vmcnt + s_barrier before a slot is read) orders the copy against its readers instead.
#pragma unroll
    for (int i = 0; i < NPC; ++i)
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g + i * 1024), "s"(l + i * 1024) : "memory");
}
// piece i of this wave's share (`pieces` x 1 KiB) of a chunk
__device__ __forceinline__ void nm_r_dma_piece(const _Float16* chunk, char* slot, int pieces, int i, int wave, int lane) {
    const char* g = reinterpret_cast<const char*>(chunk) + (wave * pieces + i) * 1024 + lane * 16;
    const unsigned l = (unsigned)(size_t)(nm_lptr)slot + (wave * pieces + i) * 1024;
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l) : "memory");
}
template <int N>
__device__ __forceinline__ void nm_r_wait_vm() {  // all but the N youngest vector-memory operations of this wave have completed, and every LDS operation
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void nm_r_wait_dma() {  // every LDS-DMA piece this wave has issued has landed (LDS reads in flight stay in flight)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

struct NmRFrag {
    nm_h8 a, b;  // plane h1, plane h2 of one column tile's k-step
};
// lane_off: byte offset of the slot + 16 * lane (one register per ring slot, made opaque by the caller: the k-step / plane
// offsets then fit the 16-bit immediate of ds_read_b128 instead of costing an address register each)
__device__ __forceinline__ NmRFrag nm_r_ld_frag(const char* ring, unsigned lane_off, int j) {
    NmRFrag f;
    const char* p = ring + lane_off + j * NM_R_KSTEP_BYTES;
    f.a = *reinterpret_cast<const nm_h8*>(p);
    if (NM_R_EXP & 16) f.b = f.a;   // timing knock-out: half the fragment loads
    else f.b = *reinterpret_cast<const nm_h8*>(p + 1024);
    if (NM_R_EXP & 32) {            // timing knock-out: the loads happen, nothing waits for them
        asm volatile("" ::"v"(f.a), "v"(f.b));
        f.a = f.b = nm_h8{1, 1, 1, 1, 1, 1, 1, 1};
    }
    return f;
}

#ifndef NM_R_LO2
#define NM_R_LO2 0   // 1: the two residual products of a k-step go to two scaled accumulators (no MFMA ever follows one on its own accumulator)
#endif
struct NmRAcc {
    nm_f32x16 hi, lo;  // one column tile: main and 2^11-scaled accumulators
#if NM_R_LO2
    nm_f32x16 lo2;
#endif
};
struct NmROps {
    uint4 v[16][2];  // B operands of a layer: [k-step][plane], 8 halves each; this lane: features 16 ks + 8 h + (0..7) of its point
};
__device__ __forceinline__ unsigned& nm_r_word(uint4& q, int i) { return i == 0 ? q.x : i == 1 ? q.y : i == 2 ? q.z : q.w; }

// 8 fp32 values of one lane -> the lane's 8 halves of a B operand, both planes
__device__ __forceinline__ void nm_r_pack8(const float (&y)[8], uint4& a, uint4& b, float& mx) {
    nm_h2_split2(y[0], y[1], a.x, b.x);
    nm_h2_split2(y[2], y[3], a.y, b.y);
    nm_h2_split2(y[4], y[5], a.z, b.z);
    nm_h2_split2(y[6], y[7], a.w, b.w);
#pragma unroll
    for (int e = 0; e < 8; e += 2) mx = fmaxf(fmaxf(mx, fabsf(y[e])), fabsf(y[e + 1]));
    asm volatile("" : "+v"(mx));
}

// value of `x` in the lane 16 below (tangent lanes n >= 16 of a 32-lane half read their value lane n - 16)
__device__ __forceinline__ float nm_r_from_value_lane(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);  // result[0]: odd rows of 16 lanes <- even rows of the source
    return __uint_as_float(r[0]);
}

// ---------------------------------------------------------------------- one activation of a tile's epilogue
// Register r (0..15) of column tile t, whose accumulators are c -> one activation, in THREE stages issued in consecutive
// ticks (S1 of activation a beside S2 of a - 1 and S3 of a - 2).  A wave is alone on its SIMD and issues in order: a
// dependent chain z -> exp -> log -> max written in one piece holds every later instruction -- the next MFMA included --
// behind each quarter-rate result; staged, every instruction finds its operands long finished.
//   S1: z = hi + lo 2^-11; e = 2^min(z, 21 log2 e)        S2: l = log2(1 + e) (and 1 / (1 + e) with TANGENT)
//   S3: y = max(z, l) [value] / z g(partner) [tangent column]; odd r completes a pair = one 32-bit word (two halves) of each
//       plane of the next layer's B operand, k-step 2 t + (r >> 3), word (r & 7) >> 1; LAST: y into the NOUT head sums.
// ACT 0: softplus in log2 units (nm_softplus_l2), 1: ReLU (S1, S2 do nothing beyond z).
struct NmRActPipe {
    float z[3], e[3], l[3], rc[3];
    float carry;
};
template <int ACT, bool TANGENT>
__device__ __forceinline__ void nm_r_act_s1(const NmRAcc& c, int r, NmRActPipe& ps) {
    const float sc = 1.0f / 2048.0f;
#if NM_R_LO2
    const float z = fmaf(c.lo[r] + c.lo2[r], sc, c.hi[r]);
#else
    const float z = fmaf(c.lo[r], sc, c.hi[r]);  // (bias: in the accumulator)
#endif
    ps.z[r % 3] = z;
    if (ACT == 0) ps.e[r % 3] = __builtin_amdgcn_exp2f(fminf(z, 30.2965958f));
}
template <int ACT, bool TANGENT>
__device__ __forceinline__ void nm_r_act_s2(int r, NmRActPipe& ps) {
    if (ACT == 0) {
        const float u = 1.0f + ps.e[r % 3];
        ps.l[r % 3] = __builtin_amdgcn_logf(u);
        if (TANGENT) ps.rc[r % 3] = __builtin_amdgcn_rcpf(u);
    }
}
template <int ACT, bool TANGENT, bool LAST, int NOUT>
__device__ __forceinline__ void nm_r_act_s3(int t, int r, NmROps& out, const float* head_w, float (&so)[NOUT], bool tangent_lane, int h, NmRActPipe& ps,
                                            float& mx) {
    const float z = ps.z[r % 3];
    float y = ACT == 0 ? fmaxf(z, ps.l[r % 3]) : fmaxf(z, 0.f);
    if (TANGENT) {
        const float g = ACT == 0 ? ps.e[r % 3] * ps.rc[r % 3] : (z > 0.f ? 1.f : 0.f);
        const float gp = nm_r_from_value_lane(g);
        y = tangent_lane ? z * gp : y;
    }
    if (!LAST) {
        if ((r & 1) == 0) {
            ps.carry = y;
        } else {
            const int ks = 2 * t + (r >> 3), w = (r & 7) >> 1;
            nm_h2_split2(ps.carry, y, nm_r_word(out.v[ks][0], w), nm_r_word(out.v[ks][1], w));
            mx = fmaxf(fmaxf(mx, fabsf(ps.carry)), fabsf(y));
            asm volatile("" : "+v"(mx));   // (taken now: left to the scheduler, the running maximum is formed at the very end and every activation spilled for it)
        }
    } else {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) so[o] = fmaf(y, head_w[o * NM_W + (t * 2 + h) * 16 + r], so[o]);
    }
}
// tick k (0..17) of a tile's epilogue: S1 of activation k, S2 of k - 1, S3 of k - 2
template <int ACT, bool TANGENT, bool LAST, int NOUT>
__device__ __forceinline__ void nm_r_act_tick(const NmRAcc& c, int t, int k, NmROps& out, const float* head_w, float (&so)[NOUT], bool tangent_lane, int h,
                                              NmRActPipe& ps, float& mx) {
    if (k >= 2 && k - 2 < 16) nm_r_act_s3<ACT, TANGENT, LAST, NOUT>(t, k - 2, out, head_w, so, tangent_lane, h, ps, mx);
    if (k >= 1 && k - 1 < 16) nm_r_act_s2<ACT, TANGENT>(k - 1, ps);
    if (k < 16) nm_r_act_s1<ACT, TANGENT>(c, k, ps);
}

// main accumulator of a tile <- bias (value columns) / 0 (tangent columns); scaled accumulator <- 0
template <bool TANGENT>
__device__ __forceinline__ void nm_r_init_acc(NmRAcc& c, const float* bias_lds, int t, int h, bool tangent_lane) {
    const float* bp = bias_lds + (t * 2 + h) * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 b4 = *reinterpret_cast<const float4*>(bp + 4 * q);
        c.hi[4 * q + 0] = (TANGENT && tangent_lane) ? 0.f : b4.x;
        c.hi[4 * q + 1] = (TANGENT && tangent_lane) ? 0.f : b4.y;
        c.hi[4 * q + 2] = (TANGENT && tangent_lane) ? 0.f : b4.z;
        c.hi[4 * q + 3] = (TANGENT && tangent_lane) ? 0.f : b4.w;
    }
    c.lo = nm_f32x16{0};
#if NM_R_LO2
    c.lo2 = nm_f32x16{0};
#endif
}

// Kernel-wide description: D layers (the first with KS0 k-steps), activation, head width, tangent columns.
template <int KS0_, int D_, int ACT_, int NOUT_, bool TANGENT_>
struct NmRCfg {
    static constexpr int KS0 = KS0_, D = D_, ACT = ACT_, NOUT = NOUT_;
    static constexpr bool TANGENT = TANGENT_;
    typedef NmRSched<KS0_, D_> SCH;
};

__device__ __forceinline__ void nm_r_dma_n(const _Float16* src, char* slot, int pieces, int wave, int lane) {
    switch (pieces) {
        case 6: nm_r_dma<6>(src, slot, wave, lane); break;
        case 7: nm_r_dma<7>(src, slot, wave, lane); break;
        default: nm_r_dma<8>(src, slot, wave, lane); break;
    }
}
__device__ __forceinline__ void nm_r_wait_n(int younger_ops) {
    switch (younger_ops) {
        case 0: nm_r_wait_vm<0>(); break;
        case 6: nm_r_wait_vm<6>(); break;
        case 7: nm_r_wait_vm<7>(); break;
        default: nm_r_wait_vm<8>(); break;
    }
}

// ------------------------------------------------------------------------------------------------ the layers
// All D layers of a kernel as one unrolled sequence of column TILES (global tile index gt = 8 l + t = chunk index,
// accumulator set gt & 1).  While tile gt's MFMAs run, the 8 epilogue units of tile gt - 1 are issued between them
// (slots = the tile's k-steps; a tile that opens a layer must have the previous layer's last features -- B-operand
// k-steps 14, 15 -- complete before its own k-step 14), and in the tile's last k-step the accumulators of tile gt + 1
// are initialised.  Hand-over of the weight ring: in the MIDDLE of chunk s every wave waits for its share of chunk
// s + 1, crosses the barrier (=> chunk s + 1 is complete, and every wave has finished chunk s - 1) and issues its share
// of chunk s + 2 into the slot of chunk s - 1; the first fragment of chunk s + 1 is then fetched during the last
// k-step of chunk s, so a chunk boundary costs no LDS latency.
// bufs[l & 1] holds the B operands of layer l.  Returns with the head sums so[] complete.
// (one call per layer: a single loop over all 8 D tiles exceeds the compiler's full-unroll budget for D = 4)
template <class CFG, int L>
__device__ __forceinline__ void nm_r_layer(const _Float16* const (&Wl)[CFG::D], char* ring, const float* cst, const float* head_w,
                                           NmROps (&bufs)[2], NmRAcc (&acc)[2], NmRFrag (&f)[NM_R_PF + 1], float (&so)[CFG::NOUT], bool tangent_lane, float& mx) {
    typedef typename CFG::SCH SCH;
    constexpr int D = CFG::D, NT = 8 * D;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;
    NmRActPipe ps;
    unsigned lane_off[NM_R_SLOTS];
#pragma unroll
    for (int k = 0; k < NM_R_SLOTS; ++k) {
        lane_off[k] = k * NM_R_SLOT + lane * 16;
        asm volatile("" : "+v"(lane_off[k]));
    }
#pragma unroll
    for (int gt = 8 * L; gt < 8 * L + 8; ++gt) {
        const int l = gt >> 3, t = gt & 7, s = gt;
        const int KS = SCH::ks_of(l);
        NmRAcc& c = acc[gt & 1];
        const NmROps& in = bufs[l & 1];
        // where the previous tile's epilogue goes: same layer -> this layer's output operands (or the head sums);
        // previous layer's last tile -> THIS layer's input operands 14, 15
        const bool prev_cross = t == 0;
        const int n_slots = gt == 0 ? 0 : (prev_cross ? (KS - 1 < 14 ? KS - 1 : 14) : KS - 1);
        const int g0 = SCH::ksteps_before(s);
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const int fb = (g0 + j) % (NM_R_PF + 1), fn = (g0 + j + NM_R_PF) % (NM_R_PF + 1);
            if (!(NM_R_EXP & 2))
            if (j == KS / 2 && s + 1 < SCH::S) {  // ring hand-over (see above); chunk s + 1 is the youngest copy in flight
                nm_r_wait_dma();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            // this wave's share of chunk s + 2, one 1 KiB piece per k-step from the hand-over on (a piece costs its issuer tens of
            // cycles of issue: eight in a row would idle the matrix pipe)
            if (!(NM_R_EXP & 2))
            if (j >= KS / 2 && j - KS / 2 < SCH::dma_ops(s + 2) && s + 2 < SCH::S)
                nm_r_dma_piece(Wl[(s + 2) / 8] + SCH::offset_halves(s + 2), ring + ((s + 2) % NM_R_SLOTS) * NM_R_SLOT, SCH::dma_ops(s + 2), j - KS / 2, wave, lane);
            if (!(NM_R_EXP & 4))
            if (j + NM_R_PF < KS) f[fn] = nm_r_ld_frag(ring, lane_off[s % NM_R_SLOTS], j + NM_R_PF);
            else if (s + 1 < SCH::S && !(NM_R_EXP & 4)) f[fn] = nm_r_ld_frag(ring, lane_off[(s + 1) % NM_R_SLOTS], j + NM_R_PF - KS);
            const NmRFrag& F = f[fb];
            const nm_h8 x0 = __builtin_bit_cast(nm_h8, in.v[j][0]), x1 = __builtin_bit_cast(nm_h8, in.v[j][1]);
            c.lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.b, x0, c.lo, 0, 0, 0);
            c.hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a, x0, c.hi, 0, 0, 0);
#if NM_R_LO2
            c.lo2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a, x1, c.lo2, 0, 0, 0);
#else
            c.lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a, x1, c.lo, 0, 0, 0);
#endif
            // epilogue ticks of the previous tile that belong to this slot (18 ticks: 16 activations through 3 stages)
            if (j < n_slots) {
                const int k0 = 18 * j / n_slots, k1 = 18 * (j + 1) / n_slots;
#pragma unroll
                for (int k = k0; k < k1; ++k) {
                    if (prev_cross) {   // previous layer's tile 7 (never a LAST layer) -> this layer's in operands
                        float dummy[CFG::NOUT];
                        nm_r_act_tick<CFG::ACT, CFG::TANGENT, false, CFG::NOUT>(acc[(gt - 1) & 1], 7, k, bufs[l & 1], nullptr, dummy, tangent_lane, h, ps, mx);
                    } else if (l == D - 1) {
                        nm_r_act_tick<CFG::ACT, CFG::TANGENT, true, CFG::NOUT>(acc[(gt - 1) & 1], t - 1, k, bufs[(l + 1) & 1], head_w, so, tangent_lane, h, ps, mx);
                    } else {
                        nm_r_act_tick<CFG::ACT, CFG::TANGENT, false, CFG::NOUT>(acc[(gt - 1) & 1], t - 1, k, bufs[(l + 1) & 1], nullptr, so, tangent_lane, h, ps, mx);
                    }
                }
            }
            // accumulators of the next tile (its set was drained by the units above: they end before the last k-step)
            if (j == KS - 1 && gt + 1 < NT)
                nm_r_init_acc<CFG::TANGENT>(acc[(gt + 1) & 1], cst + ((gt + 1) >> 3) * NM_W, (gt + 1) & 7, h, tangent_lane);
            // issue order inside the k-step: every MFMA followed by a few of the vector / LDS instructions above
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, NM_R_VALU_PER_MFMA, 0);   // a few vector instructions
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // up to 1 LDS read
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
template <class CFG>
__device__ __forceinline__ void nm_r_layers(const _Float16* const (&Wl)[CFG::D], char* ring, const float* cst, const float* head_w,
                                            NmROps (&bufs)[2], float (&so)[CFG::NOUT], bool tangent_lane, float& mx) {
    typedef typename CFG::SCH SCH;
    constexpr int NT = 8 * CFG::D;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    NmRAcc acc[2];
    NmRFrag f[NM_R_PF + 1];
    // chunk 0 has landed: this wave's share (all but the DMA of chunk 1 complete), then everyone's; also publishes cst
    nm_r_wait_n(SCH::dma_ops(1));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < NM_R_PF; ++i) f[i] = nm_r_ld_frag(ring, lane * 16, i);
    nm_r_init_acc<CFG::TANGENT>(acc[0], cst, 0, h, tangent_lane);
    nm_phase_stamp(1);
    if (!(NM_R_EXP & 8))
    nm_r_layer<CFG, 0>(Wl, ring, cst, head_w, bufs, acc, f, so, tangent_lane, mx);
    nm_phase_stamp(2);
    if (CFG::D > 1 && !(NM_R_EXP & 8)) nm_r_layer<CFG, (CFG::D > 1 ? 1 : 0)>(Wl, ring, cst, head_w, bufs, acc, f, so, tangent_lane, mx);
    nm_phase_stamp(3);
    if (CFG::D > 2 && !(NM_R_EXP & 8)) nm_r_layer<CFG, (CFG::D > 2 ? 2 : 0)>(Wl, ring, cst, head_w, bufs, acc, f, so, tangent_lane, mx);
    nm_phase_stamp(4);
    if (CFG::D > 3 && !(NM_R_EXP & 8)) nm_r_layer<CFG, (CFG::D > 3 ? 3 : 0)>(Wl, ring, cst, head_w, bufs, acc, f, so, tangent_lane, mx);
    nm_phase_stamp(5);
    // the last tile of the last layer: its epilogue has nothing left to hide behind
    NmRActPipe ps;
#pragma unroll
    for (int k = 0; k < 18; ++k)
        nm_r_act_tick<CFG::ACT, CFG::TANGENT, true, CFG::NOUT>(acc[(NT - 1) & 1], 7, k, bufs[0], head_w, so, tangent_lane, h, ps, mx);
}

// first two chunks of a kernel into slots 0 and 1
template <class SCH>
__device__ __forceinline__ void nm_r_prime(const _Float16* W0, char* ring) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    nm_r_dma_n(W0 + SCH::offset_halves(0), ring, SCH::dma_ops(0), wave, lane);
    nm_r_dma_n(W0 + SCH::offset_halves(1), ring + NM_R_SLOT, SCH::dma_ops(1), wave, lane);
}

// x[8] and its sin / cos / sin 2x / cos 2x blocks (two embedding bands) as B operands: k-steps base, base + 2, + 4, + 6, + 8
__device__ __forceinline__ void nm_r_embed_code8(NmROps& in, int base, const float (&x)[8], float& mx) {
    float s[8], c[8], s2[8], c2[8];
    float top = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) top = fmaxf(top, fabsf(x[e]));
    if (top <= NM_SINCOS_FAST_MAX) {
#pragma unroll
        for (int e = 0; e < 8; ++e) nm_sincos_fast(x[e], &s[e], &c[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) nm_sincos(x[e], &s[e], &c[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {  // second band from the first by the double-angle identities (as nm_h2_embed_chunk_t)
        s2[e] = 2.0f * s[e] * c[e];
        c2[e] = (c[e] - s[e]) * (c[e] + s[e]);
    }
    nm_r_pack8(x, in.v[base][0], in.v[base][1], mx);
    nm_r_pack8(s, in.v[base + 2][0], in.v[base + 2][1], mx);
    nm_r_pack8(c, in.v[base + 4][0], in.v[base + 4][1], mx);
    nm_r_pack8(s2, in.v[base + 6][0], in.v[base + 6][1], mx);
    nm_r_pack8(c2, in.v[base + 8][0], in.v[base + 8][1], mx);
}

// (sin, cos) pairs of ds * 2^b for the four bands b0 .. b0 + 3: the 8 features of one lane-half of the ds k-step
__device__ __forceinline__ void nm_r_embed_ds4(float dsv, int b0, float (&y)[8], float (&dy)[8]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const float f = (float)(1 << (b0 + b));
        float s, co;
        nm_sincos(dsv * f, &s, &co);
        y[2 * b] = s;
        y[2 * b + 1] = co;
        dy[2 * b] = (NM_H2_TANGENT_SCALE * f) * co;   // d/d ds, scaled as the tangent rows of nm_mlp_h2.h
        dy[2 * b + 1] = -(NM_H2_TANGENT_SCALE * f) * s;
    }
}

// ------------------------------------------------------------------------------------------- geometry MLP
// Same contract as nm_geo_mlp_h2_kernel<NABLA, true, 3>.  Physical layer-0 columns (12 k-steps):
//   ks 0,1: code[0..15], code[16..31] | 2,3: sin | 4,5: cos | 6,7: sin 2x | 8,9: cos 2x | 10: (sin, cos)(ds 2^b) b = 0..7 | 11: ds, 0 ...
typedef NmRSched<12, 3> NmGeoSched;
template <bool NABLA>
__global__ __launch_bounds__(NM_R_THREADS, 1) void nm_geo_mlp_r_kernel(
    NmGeoParamsR prm, const float* __restrict__ fg_rec, const float* __restrict__ ds, const float* __restrict__ grad, NmRecMap rmap,
    long long npts, float* __restrict__ sdf_out, int P, int stride, int off, float* __restrict__ nabla_out, int nabla_slotted,
    NmSlotMap smap, int* __restrict__ overflow) {
    __shared__ __attribute__((aligned(1024))) char ring[NM_R_SLOTS * NM_R_SLOT];
    __shared__ __attribute__((aligned(16))) float cst[4 * NM_W];  // biases of the 3 layers | density weights (register order)
    constexpr int PTS_W = NABLA ? 16 : 32, PTS = 4 * PTS_W;       // points per wave / workgroup
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, h = lane >> 5;
    const bool tangent_lane = NABLA && (n >= 16);
    const long long base = (long long)blockIdx.x * PTS;
    if (smap.order) {  // no point in this workgroup's positions (valid entries lead each 64-position block)
        bool any = false;
        for (int b = 0; b < PTS; b += 64) any = any || smap.order[base + b] != 0xffffu;
        if (!any) return;
    }
    nm_phase_stamp(0);
    nm_r_prime<NmGeoSched>(prm.layer[0].W, ring);
    // ---- this lane's point
    const int p_local = wave * PTS_W + (NABLA ? (n & 15) : n);
    const long long q = base + p_local;
    const bool by_list = rmap.by_list && smap.order;
    const bool ok = q < npts && nm_slot_valid(smap, q);
    long long rq = 0, oidx = 0;
    if (ok) {
        if (by_list) {
            const long long ray0 = (q / smap.E) * smap.G;
            long long ray;
            int sp;
            nm_slot_ray(smap, q, ray0, ray, sp);
            rq = ray * rmap.stride + (rmap.slot ? (long long)rmap.slot[ray * rmap.stride + rmap.off + sp] : rmap.off + sp);
            oidx = ray * stride + off + sp;
        } else {
            rq = nm_rec_index(rmap, q);
            const long long orow = q / P;
            oidx = orow * stride + off + (q - orow * P);
        }
    }
    float dsv = 0.f;
    float x0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, x1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ok) {
        dsv = ds[rq];
        if (!tangent_lane) {
            const float4 a0 = *reinterpret_cast<const float4*>(fg_rec + rq * 32 + 8 * h), a1 = *reinterpret_cast<const float4*>(fg_rec + rq * 32 + 8 * h + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(fg_rec + rq * 32 + 16 + 8 * h), b1 = *reinterpret_cast<const float4*>(fg_rec + rq * 32 + 16 + 8 * h + 4);
            x0[0] = a0.x; x0[1] = a0.y; x0[2] = a0.z; x0[3] = a0.w; x0[4] = a1.x; x0[5] = a1.y; x0[6] = a1.z; x0[7] = a1.w;
            x1[0] = b0.x; x1[1] = b0.y; x1[2] = b0.z; x1[3] = b0.w; x1[4] = b1.x; x1[5] = b1.y; x1[6] = b1.z; x1[7] = b1.w;
        }
    }
    // constants -> LDS (register order, packed once by nm_field_pack)
    {
        const float v0 = prm.layer[0].b[threadIdx.x], v1 = prm.layer[1].b[threadIdx.x], v2 = prm.layer[2].b[threadIdx.x], v3 = prm.wd[threadIdx.x];
        cst[threadIdx.x] = v0;
        cst[NM_W + threadIdx.x] = v1;
        cst[2 * NM_W + threadIdx.x] = v2;
        cst[3 * NM_W + threadIdx.x] = v3;
    }
    float mx = 0.f;
    NmROps bufs[2];
    NmROps& A = bufs[0];
    // ---- layer-0 B operands of this lane
    if (!tangent_lane) {
        nm_r_embed_code8(A, 0, x0, mx);
        nm_r_embed_code8(A, 1, x1, mx);
    } else {
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) A.v[ks][0] = A.v[ks][1] = make_uint4(0u, 0u, 0u, 0u);
    }
    {
        float y[8], dy[8];
        nm_r_embed_ds4(dsv, 4 * h, y, dy);
        float last[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dlast[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        last[0] = h == 0 ? dsv : 0.f;
        dlast[0] = h == 0 ? NM_H2_TANGENT_SCALE : 0.f;
        if (!ok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = dy[e] = last[e] = dlast[e] = 0.f;
        }
        if (tangent_lane) {
            nm_r_pack8(dy, A.v[10][0], A.v[10][1], mx);
            nm_r_pack8(dlast, A.v[11][0], A.v[11][1], mx);
        } else {
            nm_r_pack8(y, A.v[10][0], A.v[10][1], mx);
            nm_r_pack8(last, A.v[11][0], A.v[11][1], mx);
        }
    }
    const _Float16* const Wl[3] = {prm.layer[0].W, prm.layer[1].W, prm.layer[2].W};
    float so[1] = {0.f};
    nm_r_layers<NmRCfg<12, 3, 0, 1, NABLA>>(Wl, ring, cst, cst + 3 * NM_W, bufs, so, tangent_lane, mx);
    // ---- head: this lane summed its 128 features; the other 128 sit in the lane 32 above / below
    const float tot = so[0] + __shfl_xor(so[0], 32);
    if (NABLA) {
        const float dsdf = __shfl(tot, (lane & 32) | (n & 15) | 16) * (1.0f / NM_H2_TANGENT_SCALE);  // the tangent column of this point
        if (ok && h == 0 && n < 16) {
            if (sdf_out) sdf_out[oidx] = tot + prm.bd;
            if (nabla_out) {
                const long long no = nabla_slotted ? oidx : q;
                nabla_out[no * 3 + 0] = dsdf * grad[rq * 3 + 0];
                nabla_out[no * 3 + 1] = dsdf * grad[rq * 3 + 1];
                nabla_out[no * 3 + 2] = dsdf * grad[rq * 3 + 2];
            }
        }
    } else {
        if (ok && h == 0 && sdf_out) sdf_out[oidx] = tot + prm.bd;
    }
    nm_h2_raise(overflow, mx);
    nm_phase_stamp(15);
}

// --------------------------------------------------------------------------------------------- colour MLP
// Same contract as nm_col_mlp_h2_kernel<true, 3>.  Physical layer-0 columns (13 k-steps + one of zeros: chunks are copied in 4 equal shares):
//   ks 0..9: colour-code embedding as above | 10: (sin, cos)(ds 2^b) | 11: h = 0: ds, view (3), nabla (3), 0; h = 1: view-band values 0..7 |
//   12: h = 0: view-band values 8..15, h = 1: 16..23;  band value e = 6 b + (dim: sin) / 6 b + 3 + dim (cos), b = 0..3
typedef NmRSched<14, 4> NmColSched;
__device__ __forceinline__ float nm_r_view_band(const float (&dv)[3], int e) {
    const int b = e / 6, w = e - 6 * b, dim = w % 3;
    float s, co;
    nm_sincos((dim == 0 ? dv[0] : dim == 1 ? dv[1] : dv[2]) * (float)(1 << b), &s, &co);
    return w < 3 ? s : co;
}
__global__ __launch_bounds__(NM_R_THREADS, 1) void nm_col_mlp_r_kernel(
    NmColParamsR prm, const float* __restrict__ ft_rec, const float* __restrict__ ds, const float* __restrict__ nabla,
    const float* __restrict__ dirs, int dir_div, long long npts, float* __restrict__ rgb_out, NmSlotMap smap, int* __restrict__ overflow) {
    __shared__ __attribute__((aligned(1024))) char ring[NM_R_SLOTS * NM_R_SLOT];
    __shared__ __attribute__((aligned(16))) float cst[7 * NM_W];  // biases of the 4 layers | rgb weights [3][256] (register order)
    constexpr int PTS = 128;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, h = lane >> 5;
    const long long base = (long long)blockIdx.x * PTS;
    if (smap.order) {
        bool any = false;
        for (int b = 0; b < PTS; b += 64) any = any || smap.order[base + b] != 0xffffu;
        if (!any) return;
    }
    nm_phase_stamp(0);
    nm_r_prime<NmColSched>(prm.layer[0].W, ring);
    const long long q = base + wave * 32 + n;
    const bool ok = q < npts && nm_slot_valid(smap, q);
    float dsv = 0.f, dv[3] = {0.f, 0.f, 0.f}, nb[3] = {0.f, 0.f, 0.f};
    float x0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, x1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long oq = q;
    if (ok) {
        long long ray;
        int sp;
        if (smap.order) {
            nm_slot_ray(smap, q, (q / smap.E) * smap.G, ray, sp);
            oq = ray * smap.P + sp;   // ordered lists: the colour goes back to its (ray, sample) position
        } else {
            ray = q / dir_div;
        }
        dsv = ds[q];
        dv[0] = dirs[ray * 3 + 0]; dv[1] = dirs[ray * 3 + 1]; dv[2] = dirs[ray * 3 + 2];
        if (h == 0) { nb[0] = nabla[q * 3 + 0]; nb[1] = nabla[q * 3 + 1]; nb[2] = nabla[q * 3 + 2]; }
        const float4 a0 = *reinterpret_cast<const float4*>(ft_rec + q * 32 + 8 * h), a1 = *reinterpret_cast<const float4*>(ft_rec + q * 32 + 8 * h + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(ft_rec + q * 32 + 16 + 8 * h), b1 = *reinterpret_cast<const float4*>(ft_rec + q * 32 + 16 + 8 * h + 4);
        x0[0] = a0.x; x0[1] = a0.y; x0[2] = a0.z; x0[3] = a0.w; x0[4] = a1.x; x0[5] = a1.y; x0[6] = a1.z; x0[7] = a1.w;
        x1[0] = b0.x; x1[1] = b0.y; x1[2] = b0.z; x1[3] = b0.w; x1[4] = b1.x; x1[5] = b1.y; x1[6] = b1.z; x1[7] = b1.w;
    }
    {
        float v[7];
#pragma unroll
        for (int l = 0; l < 4; ++l) v[l] = prm.layer[l].b[threadIdx.x];
#pragma unroll
        for (int o = 0; o < 3; ++o) v[4 + o] = prm.wrgb[o * NM_W + threadIdx.x];
#pragma unroll
        for (int l = 0; l < 7; ++l) cst[l * NM_W + threadIdx.x] = v[l];
    }
    float mx = 0.f;
    NmROps bufs[2];
    NmROps& A = bufs[0];
    nm_r_embed_code8(A, 0, x0, mx);
    nm_r_embed_code8(A, 1, x1, mx);
    {
        float y[8], dy[8];
        nm_r_embed_ds4(dsv, 4 * h, y, dy);
        float k11[8], k12[8];
        if (h == 0) {
            k11[0] = dsv; k11[1] = dv[0]; k11[2] = dv[1]; k11[3] = dv[2]; k11[4] = nb[0]; k11[5] = nb[1]; k11[6] = nb[2]; k11[7] = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) k12[e] = nm_r_view_band(dv, 8 + e);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                k11[e] = nm_r_view_band(dv, e);
                k12[e] = nm_r_view_band(dv, 16 + e);
            }
        }
        if (!ok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = k11[e] = k12[e] = 0.f;
        }
        nm_r_pack8(y, A.v[10][0], A.v[10][1], mx);
        nm_r_pack8(k11, A.v[11][0], A.v[11][1], mx);
        nm_r_pack8(k12, A.v[12][0], A.v[12][1], mx);
        A.v[13][0] = A.v[13][1] = make_uint4(0u, 0u, 0u, 0u);
    }
    const _Float16* const Wl[4] = {prm.layer[0].W, prm.layer[1].W, prm.layer[2].W, prm.layer[3].W};
    float so[3] = {0.f, 0.f, 0.f};
    nm_r_layers<NmRCfg<14, 4, 1, 3, false>>(Wl, ring, cst, cst + 4 * NM_W, bufs, so, false, mx);
    float z[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) z[c] = so[c] + __shfl_xor(so[c], 32) + prm.brgb[c];
    if (ok && h == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb_out[oq * 3 + c] = __fdiv_rn(1.0f, 1.0f + expf(-z[c]));
    }
    nm_h2_raise(overflow, mx);
    nm_phase_stamp(15);
}
