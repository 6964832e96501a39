// nm_api.hip -- C ABI of libneumesh_hip.so (see include/neumesh_hip.h) and the host-side launch
// sequence of the render hot path.  Built for gfx950 only:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC nm_api.hip
// -ffp-contract=off keeps the declared K-NN arithmetic (no FMA contraction); fused multiply-adds
// are written explicitly (fmaf / MFMA) where they are wanted.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>  // library primitive for the one plain sort of the path (ray order)
#include <rocprim/device/device_select.hpp>      // stable compaction of the walking-ray list (nm_surface_hits)

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/neumesh_hip.h"
#include "nm_grid_build.h"
#include "nm_grid_build_dev.h"
#include "nm_kernels.h"
#include "nm_mlp.h"
#include "nm_mlp_h2.h"
#include "nm_surface.h"
#include "nm_edit.h"
#include "nm_train.h"

#define NM_PROBE_STEP 8  // probes per ray and step of nm_probe_bounds_kernel: 8 = 8 rays per wave (measured: K-NN per frame 99.9 ms with 4, 97.2 with 8, 101.7 with 16)

// ------------------------------------------------------------------------------ error state
static thread_local std::string g_err;
static int nm_fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}
#define NM_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) return nm_fail("%s failed: %s", #call, hipGetErrorString(e_));  \
    } while (0)
#define NM_LAUNCH_CHECK()                                                                      \
    do {                                                                                       \
        hipError_t e_ = hipGetLastError();                                                     \
        if (e_ != hipSuccess) return nm_fail("kernel launch failed: %s", hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------- in-stream kernel timing
// When enabled (nm_profile_enable), every launch of the four hot kernels is bracketed by a pair
// of HIP events recorded on the SAME stream the kernel is launched on; nm_profile_read sums the
// elapsed times per kernel kind.  Event records do not serialise anything; disabled by default.
// The log is the library's only process-wide state: one mutex guards it (the hot path takes it only
// while profiling is on -- `on` is an atomic flag read first).
enum { NM_K_DISTANCE = 0, NM_K_GEO = 1, NM_K_GEO_NABLA = 2, NM_K_COLOR = 3, NM_K_KINDS = 4 };
// Launches that process a data-dependent number of points add it to a device counter instead of
// `units`: counter 0 = mid-points kept by the zero-weight skip (the order kernel counts them; the
// K-NN, geometry and colour launches of the mid-point pass each process exactly those), counter 1 =
// probes actually searched by nm_probe_bounds_kernel.
enum { NM_CNT_NONE = -1, NM_CNT_MID = 0, NM_CNT_PROBE = 1, NM_CNT_N = 2 };
struct NmProfRec { hipEvent_t a, b; int kind; long long units; int counter; };
struct NmProfState {
    std::atomic<bool> on{false};
    std::mutex mu;
    std::vector<NmProfRec> recs;
    unsigned long long* counters = nullptr;  // device, NM_CNT_N entries (on the device profiling was enabled on)
};
static NmProfState g_prof;
static unsigned long long* nm_prof_counter(int which) {
    return (g_prof.on.load(std::memory_order_relaxed) && g_prof.counters) ? g_prof.counters + which : nullptr;
}
struct NmProfScope {
    NmProfRec r;
    hipStream_t s;
    bool on;
    NmProfScope(int kind, long long units, hipStream_t stream, int counter = NM_CNT_NONE) : s(stream), on(g_prof.on.load(std::memory_order_relaxed)) {
        if (!on) return;
        r.kind = kind;
        r.units = units;
        r.counter = counter;
        if (hipEventCreate(&r.a) != hipSuccess) { on = false; return; }
        if (hipEventCreate(&r.b) != hipSuccess) { (void)hipEventDestroy(r.a); on = false; return; }
        (void)hipEventRecord(r.a, s);
    }
    ~NmProfScope() {
        if (!on) return;
        (void)hipEventRecord(r.b, s);
        std::lock_guard<std::mutex> lk(g_prof.mu);
        g_prof.recs.push_back(r);
    }
};

static inline unsigned nm_blocks(long long n, int per) { return (unsigned)((n + per - 1) / per); }
static inline size_t nm_align(size_t x) { return (x + 255) & ~(size_t)255; }

// ---------------------------------------------------------------------------------- handles
struct nm_grid_s {
    NmGridView view;        // device pointers
    float* verts = nullptr;  // device copy of the vertices in ORIGINAL order [V,3]
    void* mem[3] = {nullptr, nullptr, nullptr};  // owned allocations (host build: one blob; device build: nodes, sverts, verts)
    size_t bytes = 0;
    size_t n_nodes = 0;
    int occupied = 0;
    float origin[3] = {0, 0, 0};
    float root_size = 0;
    // scratch of the deferred queries of small launches (nm_launch_distance), one block per stream that used this handle: stream-ordered
    // reuse is safe, two streams never share a block
    std::mutex defer_mu;
    std::vector<std::pair<hipStream_t, void*>> defer_scratch;
    std::atomic<int> defer_budget{-1};   // work units after which a wave of a small launch hands its queries on; -1 = the build's default (nm_grid_set_option)
};

struct nm_field_s {
    nm_field_desc desc;  // copy (pointers inside are NOT retained)
    NmGeoParams geo;
    NmColParams col;
    float* blob = nullptr;  // packed weights
    size_t blob_floats = 0;
    int precision = 0;       // 0 fp32 MFMA (nm_mlp.h), 2 f16 MFMA (nm_mlp_h2.h); mlp_precision 4 / 5 / 6 / 7 = 2 with other product counts
    bool single = false;     // f16 MFMA with ONE product per fp32 product in BOTH networks (plain fp16 operands): error-quantified mode, never the default
    int geo_np = 3, col_np = 3;  // NP template argument of the f16 kernels per network: 3 = three products / two accumulators, 6 = three products /
                                 // one accumulator (unscaled residual halves), 1 = one product
    NmGeoParamsH2 geo_h2;
    NmColParamsH2 col_h2;
    bool geo_fixed = false, col_fixed = false;  // reference configuration: kernels with constant embedding trip counts
    _Float16* blob_h = nullptr;  // split-half weights in fragment order (layout of the active mode)
    size_t blob_h_halves = 0;
    int* overflow = nullptr;     // device flag raised by the split-half kernels when a value leaves the fp16 range
};

extern "C" {

int nm_abi_version(void) { return NM_ABI_VERSION; }
const char* nm_last_error(void) { return g_err.c_str(); }
int nm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ================================================================================ grid
// Device-side build (nm_grid_build_dev.h): all O(V) work on the GPU, a few scalars through the host.
int nm_grid_create(const float* verts_device, int64_t V, int leaf_level, nm_stream_t stream_, nm_grid_t* out) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!out) return nm_fail("nm_grid_create: out is NULL");
    if (V < 1 || V > 0x7ffffff0LL) return nm_fail("nm_grid_create: V=%lld", (long long)V);
    if (leaf_level < 0 || leaf_level > NM_MAX_LEVEL) return nm_fail("nm_grid_create: leaf_level %d out of [0,%d]", leaf_level, NM_MAX_LEVEL);
    if (!verts_device) return nm_fail("nm_grid_create: verts is NULL");
    NmDevGrid dg;
    bool bad = false;
    const hipError_t e = nm_build_device_grid(verts_device, V, leaf_level, stream, dg, &bad);
    if (bad) return nm_fail("nm_grid_create: non-finite vertex coordinates");
    if (e != hipSuccess) return nm_fail("nm_grid_create: device build failed: %s", hipGetErrorString(e));
    nm_grid_s* g = new nm_grid_s();
    float* vcopy = nullptr;
    hipError_t e2 = hipMalloc((void**)&vcopy, (size_t)V * 12);
    if (e2 == hipSuccess) e2 = hipMemcpyAsync(vcopy, verts_device, (size_t)V * 12, hipMemcpyDeviceToDevice, stream);
    if (e2 == hipSuccess) e2 = hipStreamSynchronize(stream);
    if (e2 != hipSuccess) {
        hipFree(dg.nodes); hipFree(dg.sverts);
        if (vcopy) hipFree(vcopy);
        delete g;
        return nm_fail("nm_grid_create: vertex copy failed: %s", hipGetErrorString(e2));
    }
    g->mem[0] = dg.nodes; g->mem[1] = dg.sverts; g->mem[2] = vcopy;
    g->bytes = dg.n_nodes * sizeof(NmNode) + ((size_t)V + 4) * sizeof(float4) + (size_t)V * 12;
    g->n_nodes = dg.n_nodes;
    g->view.L = dg.L;
    g->view.V = (int)V;
    g->view.coop_extent = 0.75f * dg.root.root_size;   // (as nm_host_view)
    g->view.n_nodes = (int)dg.n_nodes;
    g->view.nodes = dg.nodes;
    g->view.sverts = dg.sverts;
    g->verts = vcopy;
    g->origin[0] = dg.root.ox; g->origin[1] = dg.root.oy; g->origin[2] = dg.root.oz;
    g->root_size = dg.root.root_size;
    g->occupied = dg.occupied_leaves;
    *out = g;
    return 0;
}

#ifdef NM_TESTING   // ---- test hooks: only in the separate test / measurement library (neumesh_amd/build.py: build_testing)
// Host build (nm_grid_build.h), the reference implementation the device build is checked against: device->host copy,
// CPU sort, upload.
int nm_grid_create_host(const float* verts_device, int64_t V, int leaf_level, nm_stream_t stream_, nm_grid_t* out) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!out) return nm_fail("nm_grid_create: out is NULL");
    if (V < 1) return nm_fail("nm_grid_create: V=%lld", (long long)V);
    if (leaf_level < 0 || leaf_level > NM_MAX_LEVEL) return nm_fail("nm_grid_create: leaf_level %d out of [0,%d]", leaf_level, NM_MAX_LEVEL);
    std::vector<float> hv((size_t)V * 3);
    NM_HIP(hipMemcpyAsync(hv.data(), verts_device, hv.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
    NM_HIP(hipStreamSynchronize(stream));
    NmHostGrid hg;
    if (!nm_build_host_grid(hv.data(), V, leaf_level, hg)) return nm_fail("nm_grid_create: non-finite vertex coordinates");
    nm_grid_s* g = new nm_grid_s();
    const size_t b_nodes = nm_align(hg.nodes.size() * sizeof(NmNode));
    const size_t b_sv = nm_align(hg.sverts.size() * sizeof(float4));
    const size_t b_v = nm_align(hv.size() * sizeof(float));
    g->bytes = b_nodes + b_sv + b_v;
    if (hipMalloc(&g->mem[0], g->bytes) != hipSuccess) {
        delete g;
        return nm_fail("nm_grid_create: hipMalloc(%zu) failed", b_nodes + b_sv + b_v);
    }
    char* base = (char*)g->mem[0];
    hipError_t e = hipMemcpyAsync(base, hg.nodes.data(), hg.nodes.size() * sizeof(NmNode), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(base + b_nodes, hg.sverts.data(), hg.sverts.size() * sizeof(float4), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(base + b_nodes + b_sv, hv.data(), hv.size() * sizeof(float), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
        hipFree(g->mem[0]);
        delete g;
        return nm_fail("nm_grid_create: upload failed: %s", hipGetErrorString(e));
    }
    g->view = nm_host_view(hg);
    g->view.nodes = (const NmNode*)base;
    g->view.sverts = (const float4*)(base + b_nodes);
    g->verts = (float*)(base + b_nodes + b_sv);
    g->n_nodes = hg.nodes.size();
    g->origin[0] = hg.ox; g->origin[1] = hg.oy; g->origin[2] = hg.oz;
    g->root_size = hg.root_size;
    g->occupied = hg.occupied_leaves;
    *out = g;
    return 0;
}

// Test hook: the node records and sorted vertices of a handle, copied to host buffers (sizes from nm_grid_get_info).
int nm_grid_debug_export(nm_grid_t g, void* nodes_host, int64_t nodes_bytes, void* sverts_host, int64_t sverts_bytes) {
    if (!g || !nodes_host || !sverts_host) return nm_fail("nm_grid_debug_export: NULL argument");
    if (nodes_bytes != (int64_t)(g->n_nodes * sizeof(NmNode)) || sverts_bytes != (int64_t)(((size_t)g->view.V + 4) * sizeof(float4)))
        return nm_fail("nm_grid_debug_export: buffer sizes %lld / %lld do not match the handle", (long long)nodes_bytes, (long long)sverts_bytes);
    NM_HIP(hipMemcpy(nodes_host, g->view.nodes, (size_t)nodes_bytes, hipMemcpyDeviceToHost));
    NM_HIP(hipMemcpy(sverts_host, g->view.sverts, (size_t)sverts_bytes, hipMemcpyDeviceToHost));
    return 0;
}
#endif  // NM_TESTING

int nm_grid_destroy(nm_grid_t g) {
    if (!g) return 0;
    for (void* m : g->mem)
        if (m) hipFree(m);
    for (auto& e : g->defer_scratch)
        if (e.second) hipFree(e.second);
    delete g;
    return 0;
}

int nm_grid_set_option(nm_grid_t g, int option, int64_t value) {
    if (!g) return nm_fail("nm_grid_set_option: NULL handle");
    if (option == NM_GRID_DEFER_BUDGET) {
        if (value < -1 || value > (1ll << 30)) return nm_fail("nm_grid_set_option: budget %lld out of range", (long long)value);
        g->defer_budget.store((int)value, std::memory_order_relaxed);
        return 0;
    }
    if (option == NM_GRID_TRIM) {   // give the deferral scratch back (33.7 MB per stream that used the index); the caller vouches that no launch on this index is in flight
        std::lock_guard<std::mutex> lk(g->defer_mu);
        for (auto& e : g->defer_scratch) (void)hipFree(e.second);
        g->defer_scratch.clear();
        return 0;
    }
    return nm_fail("nm_grid_set_option: unknown option %d", option);
}

int nm_grid_get_info(nm_grid_t g, nm_grid_info* out) {
    if (!g || !out) return nm_fail("nm_grid_get_info: NULL argument");
    out->num_vertices = g->view.V;
    out->leaf_level = g->view.L;
    out->occupied_leaves = g->occupied;
    out->origin[0] = g->origin[0]; out->origin[1] = g->origin[1]; out->origin[2] = g->origin[2];
    out->root_size = g->root_size;
    out->device_bytes = (int64_t)g->bytes;
    out->num_nodes = (int64_t)g->n_nodes;
    return 0;
}

// Q: number of points of the call (0 = unknown: 64 queries per wave).  Below ~2^18 points the waves get fewer queries each
// (NmPointSrc.lanes): the launch is bound by one wave's serial traversal, not by throughput.
static NmPointSrc nm_src_xyz(const float* xyz, long long Q = 0) {
    NmPointSrc s;
    memset(&s, 0, sizeof(s));
    s.mode = 0;
    s.P = 1;
    s.xyz = xyz;
    s.lanes = 64;
    // Small launches live as long as their slowest wave, and a wave's traversal is the UNION of its queries' traversals executed at the
    // ~11 cycles per instruction of a wave that has its SIMD to itself: fewer queries per wave = more, shorter waves that overlap.  Aim at
    // >= 8192 waves, down to ONE query per wave (measured on a training step's eight K-NN launches, tools/train_trace.sh: target 4096 waves
    // with >= 8 queries each 6.9 ms, 8192 / >= 4: 6.5, 8192 / >= 1: 5.6 -- the 8 k-point launches 0.40-0.69 -> 0.26-0.33 ms --, 16384 / >= 1: 5.8).
    long long target = 8192;
#ifdef NM_TESTING
    static const int target_env = getenv("NEUMESH_KNN_WAVE_TARGET") ? atoi(getenv("NEUMESH_KNN_WAVE_TARGET")) : 0;   // A/B of the wave count aimed at
    if (target_env > 0) target = target_env;
#endif
    if (Q > 0)
        while (s.lanes > 1 && Q / s.lanes < target) s.lanes >>= 1;
#ifdef NM_TESTING
    static const int lanes_env = getenv("NEUMESH_KNN_LANES") ? atoi(getenv("NEUMESH_KNN_LANES")) : 0;   // tools/knn_small.py: queries per wave A/B
    if (lanes_env > 0) s.lanes = lanes_env;
#endif
    return s;
}

int nm_knn(nm_grid_t g, const float* q, int64_t Q, int K, int64_t* idx, float* d2, nm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (K < 1 || K > NM_MAX_K) return nm_fail("nm_knn: K=%d out of [1,%d]", K, NM_MAX_K);
    if (Q <= 0) return Q == 0 ? 0 : nm_fail("nm_knn: Q<0");
    if (!g || !idx || !d2 || !q) return nm_fail("nm_knn: NULL argument");
    const NmPointSrc src = nm_src_xyz(q, Q);
    const dim3 grid(nm_query_blocks(src, Q)), block(256);
    long long* idx_ll = reinterpret_cast<long long*>(idx);
    if (K <= 8) hipLaunchKernelGGL(nm_knn_kernel<8>, grid, block, 0, stream, g->view, src, (long long)Q, K, idx_ll, d2);
    else if (K <= 16) hipLaunchKernelGGL(nm_knn_kernel<16>, grid, block, 0, stream, g->view, src, (long long)Q, K, idx_ll, d2);
    else hipLaunchKernelGGL(nm_knn_kernel<32>, grid, block, 0, stream, g->view, src, (long long)Q, K, idx_ll, d2);
    NM_LAUNCH_CHECK();
    return 0;
}

static const NmRecMap NM_COMPACT = {1, 0, 0, nullptr, 0};

struct NmGather {  // optional gather-interpolation outputs of the distance kernel
    const float* geo_table; int gdim; float* fg;
    const float* col_table; int cdim; float* ft;
};
static const NmGather NM_NO_GATHER = {nullptr, 0, nullptr, nullptr, 0, nullptr};

// Small launches: waves that exceed a work budget hand their queries on, each to a wave of its own (nm_kernels.h, "the deferred queries of a
// small launch").  NM_DEFER_MAX_Q: above it a launch is throughput-bound and its tail does not matter.  Budget in work units (24 per node
// test, 7 per staged vertex).  NEUMESH_KNN_BUDGET overrides it (0 = never defer; 1 = defer everything the list has room for: the tests'
// way to run queries through the second path).
#define NM_DEFER_MAX_Q (1ll << 18)
#define NM_DEFER_CAP 8192
#ifndef NM_DEFER_BUDGET
#define NM_DEFER_BUDGET 30000
#endif
static size_t nm_defer_bytes() { return 256 + (size_t)NM_DEFER_CAP * (8 + 4 + 4 + 64 * 8 * sizeof(unsigned long long)); }   // count | masks | list | bounds | keys
static void* nm_defer_block(nm_grid_t g, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g->defer_mu);
    for (auto& e : g->defer_scratch)
        if (e.first == stream) return e.second;
    if (g->defer_scratch.size() >= 8) return nullptr;   // an application that keeps making streams: later ones simply do not defer
    void* p = nullptr;
    if (hipMalloc(&p, nm_defer_bytes()) != hipSuccess) {
        (void)hipGetLastError();                          // (not an error of the call: the launch runs without the second phase)
        return nullptr;
    }
    g->defer_scratch.emplace_back(stream, p);
    return p;
}

// ---- several chunks in flight (nm_render_cfg.overlap): the device-wide yield state of the pull kernels (nm_kernels.h) and what a call
// needs to launch them.  One NmYield per device, allocated on first use and never freed (33 KB); every stream of the process shares it.
struct NmOverlap {
    NmYield* y = nullptr;                 // nullptr: overlap mode off for this call
    unsigned long long* counters = nullptr;   // packet counters of this call's K-NN launches (in the caller's workspace, zeroed at the start of the call)
    int used = 0;                         // counters handed out so far
    int cap = 1, simds = 1024, prio = 0;
};
#define NM_PULL_COUNTERS 16
static int nm_yield_state(NmYield** out, int* simds) {
    static std::mutex mu;
    static NmYield* per_dev[64] = {nullptr};
    static int simd_count[64] = {0};
    int dev = 0;
    NM_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return nm_fail("overlap mode: device index %d out of range", dev);
    std::lock_guard<std::mutex> lk(mu);
    if (!per_dev[dev]) {
        NmYield* y = nullptr;
        NM_HIP(hipMalloc((void**)&y, sizeof(NmYield)));
        NM_HIP(hipMemset(y, 0, sizeof(NmYield)));
        hipDeviceProp_t prop;
        NM_HIP(hipGetDeviceProperties(&prop, dev));
        simd_count[dev] = 4 * (prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
        per_dev[dev] = y;
    }
    *out = per_dev[dev];
    *simds = simd_count[dev];
    return 0;
}
static NmPull nm_pull_for(NmOverlap* ov, long long npackets) {
    NmPull pl;
    pl.next = ov->counters + 2 * ov->used++;   // (packet counter, waves at work); callers check nm_pull_ok() first
    pl.npackets = npackets;
    pl.y = ov->y;
    pl.cap = ov->cap;
    pl.min_alive = ov->simds / 4;   // one wave per CU
    return pl;
}
static inline bool nm_pull_ok(const NmOverlap* ov) { return ov && ov->y && ov->used < NM_PULL_COUNTERS; }   // (a call with more K-NN launches than counters: plain launches for the rest)
static inline unsigned nm_pull_grid(const NmOverlap* ov, long long npackets, int waves_per_simd) {
    const long long full = (long long)ov->simds * waves_per_simd;
    return (unsigned)(npackets < full ? npackets : full);
}
// an MLP launch of a call in overlap mode: announce it (the pull waves of the other chunks make room), launch, withdraw
struct NmWantRoom {
    NmYield* y;
    hipStream_t s;
    NmWantRoom(const NmOverlap* ov, hipStream_t stream) : y(ov ? ov->y : nullptr), s(stream) {
        if (y) hipLaunchKernelGGL(nm_yield_add_kernel, dim3(1), dim3(1), 0, s, y, 1);
    }
    ~NmWantRoom() {
        if (y) hipLaunchKernelGGL(nm_yield_add_kernel, dim3(1), dim3(1), 0, s, y, -1);
    }
};

// ---- a second stream inside one call (nm_render_rays: the mid-point search beside the sample points' nabla launch).  One side stream + two
// events per (device, caller stream) that asked for one, created on first use, kept for the life of the process (at most 64; later callers
// simply run in order).  The fork / join is stream-ordered: no host synchronisation.
struct NmSide { int dev; hipStream_t main, side; hipEvent_t fork, join; };
static bool nm_side_for(hipStream_t main, NmSide* out) {
    static std::mutex mu;
    static std::vector<NmSide> pool;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lk(mu);
    for (auto& e : pool)
        if (e.dev == dev && e.main == main) { *out = e; return true; }
    if (pool.size() >= 64) return false;
    NmSide e;
    e.dev = dev;
    e.main = main;
    if (hipStreamCreateWithFlags(&e.side, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipEventCreateWithFlags(&e.fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e.join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    pool.push_back(e);
    *out = e;
    return true;
}

// work budget of small launches: the build's constant unless the index was given its own (nm_grid_set_option)
static int nm_defer_budget(nm_grid_t g) {
    const int b = g->defer_budget.load(std::memory_order_relaxed);
    return b >= 0 ? b : NM_DEFER_BUDGET;
}

static int nm_launch_distance(nm_grid_t g, const NmPointSrc& src_in, long long Q, const float* indicator, float w1,
                              float* ds, int* idx32, long long* idx64, float* w, float* grad, hipStream_t stream,
                              float* radius = nullptr, NmGather ga = NM_NO_GATHER, bool counted = false, NmOverlap* ov = nullptr) {
    if (Q <= 0) return 0;
    NmProfScope prof(NM_K_DISTANCE, counted ? 0 : Q, stream, counted ? NM_CNT_MID : NM_CNT_NONE);
    NmPointSrc src = src_in;
    src.budget = 0;
    if (nm_pull_ok(ov) && src.mode != 0) {   // pull form: one-wave workgroups draw the launch's packets from a counter
        const long long packets = (long long)nm_query_blocks(src, Q) * 4;
        const NmPull pl = nm_pull_for(ov, packets);
        if (nm_chain_len(src) > 1)
            hipLaunchKernelGGL(nm_distance_pull_kernel<true>, dim3(nm_pull_grid(ov, packets, NM_KNN_WAVES_CHAIN)), dim3(64), 0, stream, g->view, src, Q, pl, g->verts,
                               indicator, w1, ds, idx32, idx64, w, grad, radius, ga.geo_table, ga.gdim, ga.fg, ga.col_table, ga.cdim, ga.ft);
        else
            hipLaunchKernelGGL(nm_distance_pull_kernel<false>, dim3(nm_pull_grid(ov, packets, NM_KNN_WAVES)), dim3(64), 0, stream, g->view, src, Q, pl, g->verts,
                               indicator, w1, ds, idx32, idx64, w, grad, radius, ga.geo_table, ga.gdim, ga.fg, ga.col_table, ga.cdim, ga.ft);
        NM_LAUNCH_CHECK();
        return 0;
    }
    if (Q <= NM_DEFER_MAX_Q && nm_chain_len(src) <= 1 && !src.order) {
        const int budget = nm_defer_budget(g);
        char* blk = budget > 0 ? (char*)nm_defer_block(g, stream) : nullptr;
        if (blk) {
            src.budget = budget;
            src.defer_cap = NM_DEFER_CAP;
            src.defer_count = (int*)blk;
            unsigned long long* masks = (unsigned long long*)(blk + 256);
            src.defer_list = (int*)(blk + 256 + (size_t)NM_DEFER_CAP * 8);
            src.defer_bound2 = (float*)(blk + 256 + (size_t)NM_DEFER_CAP * 12);
            unsigned long long* keys = (unsigned long long*)(blk + 256 + (size_t)NM_DEFER_CAP * 16);
            NM_HIP(hipMemsetAsync(blk, 0, 256 + (size_t)NM_DEFER_CAP * 8, stream));   // the counter and the masks
            hipLaunchKernelGGL((nm_distance_kernel<false, true>), dim3(nm_query_blocks(src, Q)), dim3(256), 0, stream, g->view, src, Q, g->verts,
                               indicator, w1, ds, idx32, idx64, w, grad, radius, ga.geo_table, ga.gdim, ga.fg, ga.col_table, ga.cdim, ga.ft);
            hipLaunchKernelGGL(nm_knn_subtree_kernel, dim3(1536), dim3(256), 0, stream, g->view, src, src.defer_bound2, keys, masks);
            hipLaunchKernelGGL(nm_distance_deferred_kernel, dim3(NM_DEFER_CAP / 256), dim3(256), 0, stream, g->view, src, keys, masks, g->verts, indicator, w1,
                               ds, idx32, idx64, w, grad, radius, ga.geo_table, ga.gdim, ga.fg, ga.col_table, ga.cdim, ga.ft);
            NM_LAUNCH_CHECK();
            return 0;
        }
    }
    if (nm_chain_len(src) > 1)
        hipLaunchKernelGGL(nm_distance_kernel<true>, dim3(nm_query_blocks(src, Q)), dim3(256), 0, stream, g->view, src, Q, g->verts,
                           indicator, w1, ds, idx32, idx64, w, grad, radius, ga.geo_table, ga.gdim, ga.fg, ga.col_table, ga.cdim, ga.ft);
    else
        hipLaunchKernelGGL(nm_distance_kernel<false>, dim3(nm_query_blocks(src, Q)), dim3(256), 0, stream, g->view, src, Q, g->verts,
                           indicator, w1, ds, idx32, idx64, w, grad, radius, ga.geo_table, ga.gdim, ga.fg, ga.col_table, ga.cdim, ga.ft);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_compute_distance(nm_grid_t g, const float* q, int64_t Q, const float* indicator, float w1, int K, float* ds,
                        int64_t* idx, float* w, float* dds_dx, nm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (Q == 0) return 0;
    if (!g || !q) return nm_fail("nm_compute_distance: NULL argument");
    if (K != 8) return nm_fail("nm_compute_distance: K=%d unsupported (the fused kernel is built for K=8, the value the reference uses: models/mesh_grid.py:77)", K);
    if (!indicator) return nm_fail("nm_compute_distance: indicator is NULL");
    if (g->view.V < 8) return nm_fail("nm_compute_distance: mesh has %d < 8 vertices", g->view.V);
    if (Q < 0) return nm_fail("nm_compute_distance: Q<0");
    return nm_launch_distance(g, nm_src_xyz(q, Q), Q, indicator, w1, ds, nullptr, reinterpret_cast<long long*>(idx), w, dds_dx, stream);
}

int nm_distance_interpolate(nm_grid_t g, const float* q, int64_t Q, const float* indicator, float w1, const float* table,
                            int dim, float* ds, int64_t* idx, float* w, float* feat, nm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (Q == 0) return 0;
    if (!g || !q || !indicator || !table || !feat) return nm_fail("nm_distance_interpolate: NULL argument");
    if (Q < 0) return nm_fail("nm_distance_interpolate: Q<0");
    if (dim < 4 || dim % 4) return nm_fail("nm_distance_interpolate: dim=%d must be a positive multiple of 4", dim);
    if (g->view.V < 8) return nm_fail("nm_distance_interpolate: mesh has %d < 8 vertices", g->view.V);
    const NmGather ga = {table, dim, feat, nullptr, 0, nullptr};
    return nm_launch_distance(g, nm_src_xyz(q, Q), Q, indicator, w1, ds, nullptr, reinterpret_cast<long long*>(idx), w, nullptr, stream, nullptr, ga);
}

// =============================================================================== field
static int nm_round16(int k) { return (k + 15) & ~15; }

static int nm_field_validate(const nm_field_desc* d) {
    if (!d) return nm_fail("nm_field: desc is NULL");
    if (d->W != NM_W) return nm_fail("nm_field: W=%d unsupported (kernels are tiled for W=256, the reference's value: models/frameworks/neumesh/__init__.py:26)", d->W);
    if (d->D_density < 1 || d->D_density > NM_MAX_LAYERS || d->D_color < 1 || d->D_color > NM_MAX_LAYERS) return nm_fail("nm_field: layer counts out of range");
    if (d->geometry_dim < 4 || d->geometry_dim > 64 || d->geometry_dim % 4) return nm_fail("nm_field: geometry_dim=%d must be a multiple of 4 in [4,64]", d->geometry_dim);
    if (d->color_dim < 4 || d->color_dim > 64 || d->color_dim % 4) return nm_fail("nm_field: color_dim=%d must be a multiple of 4 in [4,64]", d->color_dim);
    if (d->multires_d < 0 || d->multires_fg < 0 || d->multires_ft < 0 || d->multires_view < 0) return nm_fail("nm_field: negative multires (identity embedders) unsupported");
    if (d->multires_d > 16 || d->multires_view > 16) return nm_fail("nm_field: multires too large");
    if (!d->use_view_dirs) return nm_fail("nm_field: use_view_dirs=0 unsupported");
    if (d->mlp_precision != 0 && d->mlp_precision != 2 && (d->mlp_precision < 4 || d->mlp_precision > 7))
        return nm_fail("nm_field: mlp_precision=%d (0 = fp32 MFMA, 2 = split-half f16 MFMA, 4 = single-product f16 MFMA, 5 = split-half geometry + "
                       "single-product colour, 6 = split-half with one accumulator, 7 = 6 with single-product colour)", d->mlp_precision);
    const int in_geo = 1 + 2 * d->multires_d + d->geometry_dim * (1 + 2 * d->multires_fg);
    const int in_col = (d->enable_nablas_input ? 3 : 0) + 1 + 2 * d->multires_d + 3 * (1 + 2 * d->multires_view) + d->color_dim * (1 + 2 * d->multires_ft);
    if (in_geo > 256 || in_col > 256) return nm_fail("nm_field: MLP input width %d/%d exceeds the 256-column LDS tile", in_geo, in_col);
    for (int l = 0; l < d->D_density; ++l) if (!d->geo_weight[l] || !d->geo_bias[l]) return nm_fail("nm_field: geo layer %d NULL", l);
    for (int l = 0; l < d->D_color; ++l) if (!d->col_weight[l] || !d->col_bias[l]) return nm_fail("nm_field: col layer %d NULL", l);
    if (!d->density_weight || !d->density_bias || !d->rgb_weight || !d->rgb_bias) return nm_fail("nm_field: output layer NULL");
    return 0;
}

static int nm_field_pack(nm_field_s* f, const nm_field_desc* d, hipStream_t stream) {
    const int in_geo = 1 + 2 * d->multires_d + d->geometry_dim * (1 + 2 * d->multires_fg);
    const int in_col = (d->enable_nablas_input ? 3 : 0) + 1 + 2 * d->multires_d + 3 * (1 + 2 * d->multires_view) + d->color_dim * (1 + 2 * d->multires_ft);
    size_t need = 0;
    for (int l = 0; l < d->D_density; ++l) need += (size_t)NM_W * nm_round16(l == 0 ? in_geo : NM_W) + NM_W;
    for (int l = 0; l < d->D_color; ++l) need += (size_t)NM_W * nm_round16(l == 0 ? in_col : NM_W) + NM_W;
    need += NM_W + 3 * NM_W + 64;
    need += (size_t)(d->D_density + 1) * NM_W;  // precision 2: geometry biases and density weights in log2 units (nm_mlp_h2.h)
    if (need > f->blob_floats) {
        if (f->blob) hipFree(f->blob);
        f->blob = nullptr;
        f->blob_floats = 0;
        NM_HIP(hipMalloc((void**)&f->blob, need * sizeof(float)));
        f->blob_floats = need;
    }
    float* p = f->blob;
    auto pack = [&](const float* src, int in_dim, NmLayer& L, const float* bias) -> int {
        L.Kpad = nm_round16(in_dim);
        L.W = p;
        hipLaunchKernelGGL(nm_pack_weight_kernel, dim3(nm_blocks((long long)NM_W * L.Kpad, 256)), dim3(256), 0, stream, src, NM_W, in_dim, L.Kpad, p);
        p += (size_t)NM_W * L.Kpad;
        L.b = p;
        if (hipMemcpyAsync(p, bias, NM_W * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
        p += NM_W;
        return 0;
    };
    memset(&f->geo, 0, sizeof(f->geo));
    memset(&f->col, 0, sizeof(f->col));
    for (int l = 0; l < d->D_density; ++l)
        if (pack(d->geo_weight[l], l == 0 ? in_geo : NM_W, f->geo.layer[l], d->geo_bias[l])) return nm_fail("nm_field: pack failed");
    for (int l = 0; l < d->D_color; ++l)
        if (pack(d->col_weight[l], l == 0 ? in_col : NM_W, f->col.layer[l], d->col_bias[l])) return nm_fail("nm_field: pack failed");
    NM_LAUNCH_CHECK();
    NM_HIP(hipMemcpyAsync(p, d->density_weight, NM_W * sizeof(float), hipMemcpyDeviceToDevice, stream));
    f->geo.wd = p;
    p += NM_W;
    NM_HIP(hipMemcpyAsync(p, d->rgb_weight, 3 * NM_W * sizeof(float), hipMemcpyDeviceToDevice, stream));
    f->col.wrgb = p;
    p += 3 * NM_W;
    float hb[4] = {0, 0, 0, 0};
    NM_HIP(hipMemcpyAsync(&hb[0], d->density_bias, sizeof(float), hipMemcpyDeviceToHost, stream));
    NM_HIP(hipMemcpyAsync(&hb[1], d->rgb_bias, 3 * sizeof(float), hipMemcpyDeviceToHost, stream));
    NM_HIP(hipStreamSynchronize(stream));
    f->geo.D = d->D_density;
    f->geo.bd = hb[0];
    f->geo.multires_d = d->multires_d;
    f->geo.multires_fg = d->multires_fg;
    f->geo.gdim = d->geometry_dim;
    f->geo.d_emb = 1 + 2 * d->multires_d;
    f->geo.in_dim = in_geo;
    f->col.D = d->D_color;
    f->col.brgb[0] = hb[1]; f->col.brgb[1] = hb[2]; f->col.brgb[2] = hb[3];
    f->col.multires_d = d->multires_d;
    f->col.multires_ft = d->multires_ft;
    f->col.multires_view = d->multires_view;
    f->col.cdim = d->color_dim;
    f->col.use_nabla = d->enable_nablas_input ? 1 : 0;
    f->col.d_emb = 1 + 2 * d->multires_d;
    f->col.in_dim = in_col;
    f->desc = *d;
    f->precision = d->mlp_precision >= 4 ? 2 : d->mlp_precision;
    f->single = d->mlp_precision == 4;
    f->geo_np = d->mlp_precision == 4 ? 1 : (d->mlp_precision >= 6 ? 6 : 3);
    f->col_np = (d->mlp_precision == 4 || d->mlp_precision == 5 || d->mlp_precision == 7) ? 1 : (d->mlp_precision == 6 ? 6 : 3);
    if (d->mlp_precision >= 1) {
        size_t need_h = 0;
        for (int l = 0; l < d->D_density; ++l) need_h += (size_t)NM_W * nm_round16(l == 0 ? in_geo : NM_W) * 2;
        for (int l = 0; l < d->D_color; ++l) need_h += (size_t)NM_W * nm_round16(l == 0 ? in_col : NM_W) * 2;
        if (need_h > f->blob_h_halves) {
            if (f->blob_h) hipFree(f->blob_h);
            f->blob_h = nullptr;
            f->blob_h_halves = 0;
            NM_HIP(hipMalloc((void**)&f->blob_h, need_h * sizeof(_Float16)));
            f->blob_h_halves = need_h;
        }
        if (!f->overflow) {
            NM_HIP(hipMalloc((void**)&f->overflow, sizeof(int)));
            NM_HIP(hipMemsetAsync(f->overflow, 0, sizeof(int), stream));
        }
    }
    if (f->precision == 2) {
        _Float16* ph = f->blob_h;
        memset(&f->geo_h2, 0, sizeof(f->geo_h2));
        memset(&f->col_h2, 0, sizeof(f->col_h2));
        auto pack_h2 = [&](const float* src, int in_dim, const NmColSeg& seg, NmLayerH& L, const float* packed_bias, float scale, int np) {
            L.Kpad = nm_round16(in_dim);
            L.W = ph;
            L.b = packed_bias;
            hipLaunchKernelGGL(nm_pack_weight_h2_kernel, dim3(nm_blocks((long long)NM_W * L.Kpad, 256)), dim3(256), 0, stream, src, in_dim, L.Kpad, seg, scale, np == 6 ? 1.0f : 2048.0f, ph);
            ph += (size_t)NM_W * L.Kpad * 2;
        };
        // geometry MLP in log2 units: biases x S, density weights x 1/S (fp32 copies behind the unscaled ones), layer-0 weights x S
        float* ps = p;
        auto scaled = [&](const float* src, float scale) -> const float* {
            float* dst = ps;
            hipLaunchKernelGGL(nm_scale_copy_kernel, dim3(1), dim3(NM_W), 0, stream, src, scale, NM_W, dst);
            ps += NM_W;
            return dst;
        };
        NmColSeg ident;
        memset(&ident, 0, sizeof(ident));
        ident.n = 1; ident.len[0] = NM_W; ident.src[0] = 0;
        // layer-0 column orders (nm_mlp_h2.h): logical = the reference's torch.cat order
        //   geometry  logical [ds | (sin,cos) x md | code embedding]         -> physical [code embedding | (sin,cos) x md | ds]
        //   colour    logical [nabla | ds | (sin,cos) x md | view | view bands | code embedding]
        //                                                                      -> physical [code embedding | (sin,cos) x md | view bands | view | nabla | ds]
        const int md2 = 2 * d->multires_d, d_emb = 1 + md2;
        const int FG = d->geometry_dim * (1 + 2 * d->multires_fg), FT = d->color_dim * (1 + 2 * d->multires_ft);
        NmColSeg sg;
        memset(&sg, 0, sizeof(sg));
        sg.n = 3;
        sg.len[0] = FG;  sg.src[0] = d_emb;
        sg.len[1] = md2; sg.src[1] = 1;
        sg.len[2] = 1;   sg.src[2] = 0;
        const int nb = d->enable_nablas_input ? 3 : 0, vb = 6 * d->multires_view;
        const int lo_d = nb, lo_v = nb + d_emb, lo_f = lo_v + 3 + vb;
        NmColSeg sc;
        memset(&sc, 0, sizeof(sc));
        sc.n = 6;
        sc.len[0] = FT;  sc.src[0] = lo_f;
        sc.len[1] = md2; sc.src[1] = lo_d + 1;
        sc.len[2] = vb;  sc.src[2] = lo_v + 3;
        sc.len[3] = 3;   sc.src[3] = lo_v;
        sc.len[4] = nb;  sc.src[4] = 0;
        sc.len[5] = 1;   sc.src[5] = lo_d;
        for (int l = 0; l < d->D_density; ++l)
            pack_h2(d->geo_weight[l], l == 0 ? in_geo : NM_W, l == 0 ? sg : ident, f->geo_h2.layer[l], scaled(f->geo.layer[l].b, NM_H2_S), l == 0 ? NM_H2_S : 1.0f, f->geo_np);
        for (int l = 0; l < d->D_color; ++l) pack_h2(d->col_weight[l], l == 0 ? in_col : NM_W, l == 0 ? sc : ident, f->col_h2.layer[l], f->col.layer[l].b, 1.0f, f->col_np);
        f->geo_h2.wd = scaled(f->geo.wd, 1.0f / NM_H2_S);
        NM_LAUNCH_CHECK();
        NM_HIP(hipStreamSynchronize(stream));
        f->geo_h2.D = f->geo.D; f->geo_h2.bd = f->geo.bd;
        f->geo_h2.multires_d = f->geo.multires_d; f->geo_h2.multires_fg = f->geo.multires_fg; f->geo_h2.gdim = f->geo.gdim;
        f->geo_h2.fg_w = FG; f->geo_h2.in_dim = f->geo.in_dim;
        f->col_h2.D = f->col.D; f->col_h2.wrgb = f->col.wrgb;
        f->col_h2.brgb[0] = f->col.brgb[0]; f->col_h2.brgb[1] = f->col.brgb[1]; f->col_h2.brgb[2] = f->col.brgb[2];
        f->col_h2.multires_d = f->col.multires_d; f->col_h2.multires_ft = f->col.multires_ft; f->col_h2.multires_view = f->col.multires_view;
        f->col_h2.cdim = f->col.cdim; f->col_h2.use_nabla = f->col.use_nabla; f->col_h2.ft_w = FT; f->col_h2.in_dim = f->col.in_dim;
        f->geo_fixed = d->geometry_dim == 32 && d->multires_fg == 2 && d->multires_d == 8;
        f->col_fixed = d->color_dim == 32 && d->multires_ft == 2 && d->multires_d == 8 && d->multires_view == 4 && d->enable_nablas_input;
    }
    return 0;
}

int nm_field_create(const nm_field_desc* desc, nm_stream_t stream, nm_field_t* out) {
    if (!out) return nm_fail("nm_field_create: out is NULL");
    if (nm_field_validate(desc)) return 1;
    nm_field_s* f = new nm_field_s();
    if (nm_field_pack(f, desc, (hipStream_t)stream)) {
        nm_field_destroy(f);  // frees both weight blobs
        return 1;
    }
    *out = f;
    return 0;
}

int nm_field_update(nm_field_t f, const nm_field_desc* desc, nm_stream_t stream) {
    if (!f) return nm_fail("nm_field_update: NULL handle");
    if (nm_field_validate(desc)) return 1;
    return nm_field_pack(f, desc, (hipStream_t)stream);
}

int nm_field_destroy(nm_field_t f) {
    if (!f) return 0;
    if (f->blob) hipFree(f->blob);
    if (f->blob_h) hipFree(f->blob_h);
    if (f->overflow) hipFree(f->overflow);
    delete f;
    return 0;
}

int nm_field_overflow(nm_field_t f, int* flag, nm_stream_t stream_) {
    if (!f || !flag) return nm_fail("nm_field_overflow: NULL argument");
    *flag = 0;
    if (!f->overflow) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    NM_HIP(hipMemcpyAsync(flag, f->overflow, sizeof(int), hipMemcpyDeviceToHost, stream));
    NM_HIP(hipStreamSynchronize(stream));
    if (*flag) NM_HIP(hipMemsetAsync(f->overflow, 0, sizeof(int), stream));
    return 0;
}

// scratch layout for P points: ds | idx32[8] | w[8] | grad[3] | nabla[3] | fg[gdim] | ft[cdim]   (parts a caller does not use: 0 bytes)
struct NmScratch {
    float* ds;
    int* idx;
    float* w;
    float* grad;
    float* nabla;
    float* fg;  // interpolated geometry codes [P][geometry_dim]
    float* ft;  // interpolated colour codes   [P][color_dim]
    size_t bytes;
};
static NmScratch nm_carve(void* base, long long P, bool with_idx_w = true, int gdim = 64, int cdim = 64, bool with_nabla = true) {
    NmScratch s;
    char* p = (char*)base;
    size_t o = 0;
    s.ds = (float*)(p + o);    o += nm_align((size_t)P * 4);
    s.idx = (int*)(p + o);     o += nm_align(with_idx_w ? (size_t)P * 32 : 0);
    s.w = (float*)(p + o);     o += nm_align(with_idx_w ? (size_t)P * 32 : 0);
    s.grad = (float*)(p + o);  o += nm_align((size_t)P * 12);
    s.nabla = (float*)(p + o); o += nm_align(with_nabla ? (size_t)P * 12 : 0);
    s.fg = (float*)(p + o);    o += nm_align((size_t)P * gdim * 4);
    s.ft = (float*)(p + o);    o += nm_align((size_t)P * cdim * 4);
    s.bytes = o;
    return s;
}
int64_t nm_field_scratch_bytes(int64_t P) { return (int64_t)nm_carve(nullptr, P < 1 ? 1 : P).bytes; }

static int nm_check_field_args(nm_field_t f, nm_grid_t g, const nm_field_tables* t, const char* who) {
    if (!f || !g || !t) return nm_fail("%s: NULL handle/tables", who);
    if (!t->geometry_features || !t->color_features || !t->indicator_vector) return nm_fail("%s: NULL table", who);
    if (g->view.V < 8) return nm_fail("%s: mesh has %d < 8 vertices", who, g->view.V);
    return 0;
}

static const NmSlotMap NM_NO_SLOTS = {nullptr, 0, 0, 0};

static int nm_launch_geo(nm_field_t f, const float* fg, const float* ds, const float* grad, long long P, bool nabla,
                         float* sdf, int Pper, int stride, int off, float* nabla_out, hipStream_t stream,
                         NmRecMap rmap = NM_COMPACT, int nabla_slotted = 0, NmSlotMap smap = NM_NO_SLOTS, bool counted = false, const NmOverlap* ov = nullptr) {
    if (P <= 0) return 0;
    NmProfScope prof(nabla ? NM_K_GEO_NABLA : NM_K_GEO, counted ? 0 : P, stream, counted ? NM_CNT_MID : NM_CNT_NONE);
    NmWantRoom room(ov, stream);
    const int prio = ov ? ov->prio : 0;
    if (f->precision == 2) {
        const dim3 gr(nm_blocks(P, nabla ? 32 : 64)), bl(NM_H_THREADS);
#define NM_GEO_H2(NB, FX, NP) hipLaunchKernelGGL((nm_geo_mlp_h2_kernel<NB, FX, NP>), gr, bl, 0, stream, f->geo_h2, fg, ds, grad, rmap, P, sdf, Pper, stride, off, nabla_out, nabla_slotted, smap, f->overflow, prio)
        if (f->geo_np == 3) {
            if (nabla && f->geo_fixed) NM_GEO_H2(true, true, 3);
            else if (nabla) NM_GEO_H2(true, false, 3);
            else if (f->geo_fixed) NM_GEO_H2(false, true, 3);
            else NM_GEO_H2(false, false, 3);
        } else if (f->geo_np == 6) {
            if (nabla && f->geo_fixed) NM_GEO_H2(true, true, 6);
            else if (nabla) NM_GEO_H2(true, false, 6);
            else if (f->geo_fixed) NM_GEO_H2(false, true, 6);
            else NM_GEO_H2(false, false, 6);
        } else {
            if (nabla && f->geo_fixed) NM_GEO_H2(true, true, 1);
            else if (nabla) NM_GEO_H2(true, false, 1);
            else if (f->geo_fixed) NM_GEO_H2(false, true, 1);
            else NM_GEO_H2(false, false, 1);
        }
#undef NM_GEO_H2
        NM_LAUNCH_CHECK();
        return 0;
    }
    if (nabla) {
        hipLaunchKernelGGL((nm_geo_mlp_kernel<true, false>), dim3(nm_blocks(P, 32)), dim3(256), 0, stream, f->geo, fg, ds,
                           grad, rmap, P, sdf, Pper, stride, off, nabla_out, (float*)nullptr, nabla_slotted, smap);
    } else {
        hipLaunchKernelGGL((nm_geo_mlp_kernel<false, false>), dim3(nm_blocks(P, 64)), dim3(256), 0, stream, f->geo, fg, ds,
                           grad, rmap, P, sdf, Pper, stride, off, nabla_out, (float*)nullptr, nabla_slotted, smap);
    }
    NM_LAUNCH_CHECK();
    return 0;
}

static int nm_launch_col(nm_field_t f, const float* ft, const float* ds, const float* nabla, const float* dirs, int dir_div,
                         long long P, float* rgb, hipStream_t stream, NmSlotMap smap = NM_NO_SLOTS, bool counted = false, const NmOverlap* ov = nullptr) {
    if (P <= 0) return 0;
    NmProfScope prof(NM_K_COLOR, counted ? 0 : P, stream, counted ? NM_CNT_MID : NM_CNT_NONE);
    NmWantRoom room(ov, stream);
    const int prio = ov ? ov->prio : 0;
    if (f->precision == 2) {
#define NM_COL_H2(FX, NP) hipLaunchKernelGGL((nm_col_mlp_h2_kernel<FX, NP>), dim3(nm_blocks(P, 64)), dim3(NM_H_THREADS), 0, stream, f->col_h2, ft, ds, nabla, dirs, dir_div, P, rgb, smap, f->overflow, prio)
        if (f->col_np == 3) {
            if (f->col_fixed) NM_COL_H2(true, 3);
            else NM_COL_H2(false, 3);
        } else if (f->col_np == 6) {
            if (f->col_fixed) NM_COL_H2(true, 6);
            else NM_COL_H2(false, 6);
        } else {
            if (f->col_fixed) NM_COL_H2(true, 1);
            else NM_COL_H2(false, 1);
        }
#undef NM_COL_H2
        NM_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL((nm_col_mlp_kernel<false>), dim3(nm_blocks(P, 64)), dim3(256), 0, stream, f->col, ft, ds, nabla, dirs,
                       dir_div, P, rgb, (float*)nullptr, smap);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_field_density(nm_field_t f, nm_grid_t g, const nm_field_tables* t, const float* xyz, int64_t P, float* sdf,
                     float* nabla, void* scratch, nm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (nm_check_field_args(f, g, t, "nm_field_density")) return 1;
    if (P < 0 || (P > 0 && (!xyz || !sdf || !scratch))) return nm_fail("nm_field_density: bad arguments");
    if (P == 0) return 0;
    const NmScratch s = nm_carve(scratch, P);
    const NmGather ga = {t->geometry_features, f->geo.gdim, s.fg, nullptr, 0, nullptr};
    if (nm_launch_distance(g, nm_src_xyz(xyz, P), P, t->indicator_vector, t->indicator_weight, s.ds, nullptr, nullptr, nullptr,
                           nabla ? s.grad : nullptr, stream, nullptr, ga)) return 1;
    return nm_launch_geo(f, s.fg, s.ds, s.grad, P, nabla != nullptr, sdf, 1, 1, 0, nabla, stream);
}

int nm_field_forward(nm_field_t f, nm_grid_t g, const nm_field_tables* t, const float* xyz, const float* view_dirs,
                     int64_t P, float* sdf, float* rgb, float* nabla, float* ds, int64_t* idx, float* w, void* scratch,
                     nm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (nm_check_field_args(f, g, t, "nm_field_forward")) return 1;
    if (P < 0 || (P > 0 && (!xyz || !view_dirs || !sdf || !rgb || !scratch))) return nm_fail("nm_field_forward: bad arguments");
    if (P == 0) return 0;
    const NmScratch s = nm_carve(scratch, P);
    const NmGather ga = {t->geometry_features, f->geo.gdim, s.fg, t->color_features, f->col.cdim, s.ft};
    if (nm_launch_distance(g, nm_src_xyz(xyz, P), P, t->indicator_vector, t->indicator_weight, s.ds, nullptr,
                           reinterpret_cast<long long*>(idx), w, s.grad, stream, nullptr, ga)) return 1;
    float* nab = nabla ? nabla : s.nabla;
    if (nm_launch_geo(f, s.fg, s.ds, s.grad, P, true, sdf, 1, 1, 0, nab, stream)) return 1;
    if (nm_launch_col(f, s.ft, s.ds, nab, view_dirs, 1, P, rgb, stream)) return 1;
    if (ds) NM_HIP(hipMemcpyAsync(ds, s.ds, (size_t)P * 4, hipMemcpyDeviceToDevice, stream));
    return 0;
}

int nm_field_color(nm_field_t f, const float* color_features, const float* ds, const float* view_dirs, const int64_t* idx,
                   const float* w, const float* nabla, int64_t P, float* rgb, void* scratch, nm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!f) return nm_fail("nm_field_color: NULL handle");
    if (P < 0 || (P > 0 && (!color_features || !ds || !view_dirs || !idx || !w || !rgb || !scratch))) return nm_fail("nm_field_color: bad arguments");
    if (f->col.use_nabla && !nabla && P > 0) return nm_fail("nm_field_color: nabla required (enable_nablas_input)");
    if (P == 0) return 0;
    const NmScratch s = nm_carve(scratch, P);
    hipLaunchKernelGGL(nm_interp_kernel, dim3(nm_blocks(P, 256)), dim3(256), 0, stream, color_features, f->col.cdim,
                       reinterpret_cast<const long long*>(idx), (const int*)nullptr, w, (long long)P, s.ft);
    NM_LAUNCH_CHECK();
    return nm_launch_col(f, s.ft, ds, nabla, view_dirs, 1, P, rgb, stream);
}

// =============================================================================== training form of the field (nm_train.h)
static int nm_train_validate(const nm_field_desc* d, const char* who) {
    if (!d) return nm_fail("%s: NULL descriptor", who);
    if (d->W < 16 || d->W % 16) return nm_fail("%s: W=%d must be a positive multiple of 16", who, d->W);
    if (d->D_density < 1 || d->D_density > 8 || d->D_color < 1 || d->D_color > 8) return nm_fail("%s: depths out of [1,8]", who);
    if (d->geometry_dim < 4 || d->geometry_dim % 4 || d->color_dim < 4 || d->color_dim % 4) return nm_fail("%s: code widths must be multiples of 4", who);
    if (d->multires_d > 15 || d->multires_fg > 15 || d->multires_ft > 15 || d->multires_view > 15) return nm_fail("%s: more than 15 embedder bands", who);
    for (int l = 0; l < d->D_density; ++l)
        if (!d->geo_weight[l] || !d->geo_bias[l]) return nm_fail("%s: NULL geometry weight", who);
    for (int l = 0; l < d->D_color; ++l)
        if (!d->col_weight[l] || !d->col_bias[l]) return nm_fail("%s: NULL colour weight", who);
    if (!d->density_weight || !d->density_bias || !d->rgb_weight || !d->rgb_bias) return nm_fail("%s: NULL head weight", who);
    return 0;
}

int64_t nm_train_workspace_bytes(const nm_field_desc* d, int64_t P) {
    if (!d || P < 0) return -1;
    return (int64_t)nm_train_carve(nullptr, P, nm_train_dims(d)).bytes;
}

static int nm_t_split(long long M, long long N, long long K) {   // K chunks so that a weight-gradient product fills the chip
    const long long tiles = ((M + NM_G_BM - 1) / NM_G_BM) * ((N + NM_G_BN - 1) / NM_G_BN);
    long long chunks = (1024 + tiles - 1) / tiles;
    const long long most = (K + 511) / 512;
    if (chunks > most) chunks = most;
    return (int)(chunks < 1 ? 1 : chunks);
}

static NmGemm nm_t_gemm(const float* A, long long lda, int a_kc, const float* B, long long ldb, int b_kc, float* C, long long ldc,
                        long long M, long long N, long long K, int accumulate = 0) {
    NmGemm g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.lda = lda; g.a_kc = a_kc; g.B = B; g.ldb = ldb; g.b_kc = b_kc; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.atomic = accumulate;      // C += A . B (weight gradients: added to what the caller's buffer holds, whatever the split)
    return g;
}

#define NM_T_GEMM(g, split)                                                            \
    do {                                                                               \
        if (nm_gemm_launch((g), (split), stream, d->mlp_precision != 0)) return nm_fail("nm_train: GEMM launch failed"); \
    } while (0)

int nm_train_forward(const nm_field_desc* d, nm_grid_t g, const nm_field_tables* t, const float* xyz, const float* view_dirs,
                     int64_t P, int with_nabla, float* sdf, float* nabla, float* rgb, void* workspace, nm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (nm_train_validate(d, "nm_train_forward")) return 1;
    if (!g || !t || !t->geometry_features || !t->color_features || !t->indicator_vector) return nm_fail("nm_train_forward: NULL handle/tables");
    if (g->view.V < 8) return nm_fail("nm_train_forward: mesh has %d < 8 vertices", g->view.V);
    if (P < 0 || (P > 0 && (!xyz || !sdf || !workspace))) return nm_fail("nm_train_forward: bad arguments");
    if (P == 0) return 0;
    const bool color = view_dirs != nullptr;
    if (color && !rgb) return nm_fail("nm_train_forward: rgb is NULL");
    const NmTrainDims td = nm_train_dims(d);
    const int tangent = (with_nabla || (color && td.use_nabla)) ? 1 : 0;
    if (with_nabla && !nabla) return nm_fail("nm_train_forward: nabla is NULL");
    NmTrainWs s = nm_train_carve(workspace, P, td);
    const long long W = td.W, rows = tangent ? 2 * P : P, toff = P * W;
    const NmGather ga = {t->geometry_features, td.G, s.fg, color ? t->color_features : nullptr, color ? td.Cd : 0, color ? s.ft : nullptr};
    if (nm_launch_distance(g, nm_src_xyz(xyz, P), P, t->indicator_vector, t->indicator_weight, s.ds, s.idx, nullptr, s.w, s.gds, stream,
                           nullptr, ga)) return 1;
    hipLaunchKernelGGL(nm_t_pad_kernel, dim3((unsigned)((W * td.K0p + 255) / 256)), dim3(256), 0, stream, d->geo_weight[0], s.W0p, (int)W, td.K0, td.K0p, 0);
    if (color)
        hipLaunchKernelGGL(nm_t_pad_kernel, dim3((unsigned)((W * td.Kc0p + 255) / 256)), dim3(256), 0, stream, d->col_weight[0], s.Wc0p, (int)W, td.Kc0, td.Kc0p, 0);
    hipLaunchKernelGGL(nm_t_embed_kernel, dim3((unsigned)((P + 7) / 8)), dim3(256), 0, stream, td, (long long)P, s.ds, s.fg, color ? s.ft : nullptr, view_dirs,
                       s.X0, s.T0, color ? s.C0 : nullptr, xyz, s.xyz);
    NM_LAUNCH_CHECK();
    // geometry MLP on (value | tangent) rows
    {
        NmGemm m = nm_t_gemm(s.X0, td.K0p, 1, s.W0p, td.K0p, 1, s.ZU[0], W, P, W, td.K0p);
        m.bias = d->geo_bias[0]; m.bias_rows = P;
        NM_T_GEMM(m, 1);
        if (tangent) NM_T_GEMM(nm_t_gemm(s.T0, td.Kt, 1, s.W0p, td.K0p, 1, s.ZU[0] + toff, W, P, W, td.Kt), 1);
    }
    const unsigned act_blocks = (unsigned)((P * W / 4 + 255) / 256);
    for (int l = 0; l < td.Dg; ++l) {
        if (l > 0) {
            NmGemm m = nm_t_gemm(s.HT[l - 1], W, 1, d->geo_weight[l], W, 1, s.ZU[l], W, rows, W, W);
            m.bias = d->geo_bias[l]; m.bias_rows = P;
            NM_T_GEMM(m, 1);
        }
        hipLaunchKernelGGL(nm_t_softplus_kernel, dim3(act_blocks), dim3(256), 0, stream, s.ZU[l], s.HT[l], P * W, toff, tangent);
    }
    hipLaunchKernelGGL(nm_t_geo_head_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, stream, td, (long long)P, s.HT[td.Dg - 1], d->density_weight,
                       d->density_bias, s.gds, tangent, s.sdf, s.alpha, s.nabla, color ? s.C0 : nullptr, sdf, nabla);
    NM_LAUNCH_CHECK();
    if (!color) return 0;
    for (int l = 0; l < td.Dc; ++l) {
        NmGemm m = l == 0 ? nm_t_gemm(s.C0, td.Kc0p, 1, s.Wc0p, td.Kc0p, 1, s.HC[0], W, P, W, td.Kc0p)
                          : nm_t_gemm(s.HC[l - 1], W, 1, d->col_weight[l], W, 1, s.HC[l], W, P, W, W);
        m.bias = d->col_bias[l]; m.bias_rows = P; m.relu = 1;
        NM_T_GEMM(m, 1);
    }
    hipLaunchKernelGGL(nm_t_col_head_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, stream, td, (long long)P, s.HC[td.Dc - 1], d->rgb_weight, d->rgb_bias, s.rgb, rgb);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_train_composite_forward(const float* sdf, const float* s, const float* d_mid, int d_mid_stride, const float* radiance,
                               const float* nablas, int64_t R, int N, int white_bkgd, float* rgb, float* depth, float* acc, float* normals,
                               float* cdf, float* alpha, float* weights, float* transmittance, nm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (R < 0 || N < 2 || d_mid_stride < N - 1) return nm_fail("nm_train_composite_forward: bad sizes");
    if (R == 0) return 0;
    if (!sdf || !s || !d_mid || !rgb || !depth || !acc || !cdf || !alpha || !weights || !transmittance) return nm_fail("nm_train_composite_forward: NULL argument");
    if (nablas && !normals) return nm_fail("nm_train_composite_forward: normals is NULL");
    hipLaunchKernelGGL(nm_t_composite_fwd_kernel, dim3((unsigned)((R + 63) / 64)), dim3(64), 0, stream, (long long)R, N, sdf, s, d_mid, d_mid_stride, radiance,
                       nablas, white_bkgd, rgb, depth, acc, nablas ? normals : nullptr, cdf, alpha, weights, transmittance);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_train_composite_backward(const float* sdf, const float* s, const float* d_mid, int d_mid_stride, const float* radiance,
                                const float* nablas, int64_t R, int N, int white_bkgd, const float* cdf, const float* alpha,
                                const float* weights, const float* transmittance, const float* acc, const float* depth,
                                const float* g_rgb, const float* g_depth, const float* g_acc, const float* g_normals,
                                float* g_sdf, float* g_radiance, float* g_nablas, float* g_s, nm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (R < 0 || N < 2 || d_mid_stride < N - 1) return nm_fail("nm_train_composite_backward: bad sizes");
    if (R == 0) return 0;
    if (!sdf || !s || !d_mid || !cdf || !alpha || !weights || !transmittance || !acc || !depth || !g_sdf) return nm_fail("nm_train_composite_backward: NULL argument");
    hipLaunchKernelGGL(nm_t_composite_bwd_kernel, dim3((unsigned)((R + 63) / 64)), dim3(64), 0, stream, (long long)R, N, sdf, s, d_mid, d_mid_stride, radiance,
                       nablas, white_bkgd, cdf, alpha, weights, transmittance, acc, depth, g_rgb, g_depth, g_acc, nablas ? g_normals : nullptr, g_sdf,
                       g_radiance, nablas ? g_nablas : nullptr, g_s);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_train_backward(const nm_field_desc* d, nm_grid_t g, const nm_field_tables* t, int64_t P, int with_nabla, int with_color,
                      const float* g_sdf, const float* g_nabla, const float* g_rgb, void* workspace, const nm_train_grads* out,
                      nm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (nm_train_validate(d, "nm_train_backward")) return 1;
    if (!g || !t || !t->indicator_vector || !out) return nm_fail("nm_train_backward: NULL handle/tables/grads");
    if (P < 0 || (P > 0 && !workspace)) return nm_fail("nm_train_backward: bad arguments");
    if (P == 0) return 0;
    const NmTrainDims td = nm_train_dims(d);
    const bool color = with_color != 0;
    const int tangent = (with_nabla || (color && td.use_nabla)) ? 1 : 0;
    NmTrainWs s = nm_train_carve(workspace, P, td);
    const long long W = td.W, rows = tangent ? 2 * P : P, toff = P * W;
    const unsigned strips = (unsigned)(P / 64 < 64 ? 64 : P / 64 > 2048 ? 2048 : P / 64);   // workgroups of the strip-reducing head kernels (4 waves each, >= 16 points per wave)
    NM_HIP(hipMemsetAsync(s.dds, 0, (size_t)P * 4, stream));
    NM_HIP(hipMemsetAsync(s.dnab, 0, (size_t)P * 12, stream));
    float *cur = s.DA, *oth = s.DB;
    if (color) {
        // rgb head + ReLU mask of the last hidden layer; gradients of a NULL member go to the (unused) padded scratch
        float* dWr = out->rgb_weight ? out->rgb_weight : s.dWc0p;
        float* dbr = out->rgb_bias ? out->rgb_bias : s.dWc0p;
        hipLaunchKernelGGL(nm_t_col_head_bwd_kernel, dim3(strips), dim3(256), 0, stream, td, (long long)P, g_rgb, s.rgb, s.HC[td.Dc - 1], d->rgb_weight, cur, dWr, dbr);
        NM_LAUNCH_CHECK();
        NM_HIP(hipMemsetAsync(s.dWc0p, 0, (size_t)W * td.Kc0p * 4, stream));
        for (int l = td.Dc - 1; l >= 0; --l) {
            if (out->col_bias[l]) hipLaunchKernelGGL(nm_t_colsum_kernel, dim3((unsigned)((P + NM_T_COLSUM_ROWS - 1) / NM_T_COLSUM_ROWS)), dim3(256), 0, stream, cur, (long long)P, (int)W, out->col_bias[l]);
            if (l > 0) {
                if (out->col_weight[l]) NM_T_GEMM(nm_t_gemm(cur, W, 0, s.HC[l - 1], W, 0, out->col_weight[l], W, W, W, P, 1), nm_t_split(W, W, P));
                NmGemm m = nm_t_gemm(cur, W, 1, d->col_weight[l], W, 0, oth, W, P, W, W);
                m.mask = s.HC[l - 1]; m.ldmask = W;
                NM_T_GEMM(m, 1);
                float* x = cur; cur = oth; oth = x;
            } else {
                if (out->col_weight[0]) NM_T_GEMM(nm_t_gemm(cur, W, 0, s.C0, td.Kc0p, 0, s.dWc0p, td.Kc0p, W, td.Kc0p, P, 1), nm_t_split(W, td.Kc0p, P));
                NM_T_GEMM(nm_t_gemm(cur, W, 1, s.Wc0p, td.Kc0p, 0, s.DC0, td.Kc0p, P, td.Kc0p, W), 1);
            }
        }
        if (out->col_weight[0])
            hipLaunchKernelGGL(nm_t_pad_kernel, dim3((unsigned)((W * td.Kc0p + 255) / 256)), dim3(256), 0, stream, s.dWc0p, out->col_weight[0], (int)W, td.Kc0, td.Kc0p, 1);
        hipLaunchKernelGGL(nm_t_col_input_bwd_kernel, dim3((unsigned)((P + 7) / 8)), dim3(256), 0, stream, td, (long long)P, s.DC0, s.ds, s.ft, s.idx, s.w,
                           s.dnab, s.dds, out->color_features);
        NM_LAUNCH_CHECK();
    }
    // density head + last layer's activation backward
    {
        float* dwd = out->density_weight ? out->density_weight : s.dW0p;
        float* dbd = out->density_bias ? out->density_bias : s.dW0p;
        hipLaunchKernelGGL(nm_t_geo_head_bwd_kernel, dim3(strips), dim3(256), 0, stream, td, (long long)P, g_sdf, g_nabla, color ? s.dnab : nullptr, s.gds, s.alpha,
                           d->density_weight, s.ZU[td.Dg - 1], s.HT[td.Dg - 1], tangent, cur, s.gvec, dwd, dbd);
        NM_LAUNCH_CHECK();
        NM_HIP(hipMemsetAsync(s.dW0p, 0, (size_t)W * td.K0p * 4, stream));
    }
    const unsigned act_blocks = (unsigned)((P * W / 4 + 255) / 256);
    for (int l = td.Dg - 1; l >= 0; --l) {
        if (out->geo_bias[l]) hipLaunchKernelGGL(nm_t_colsum_kernel, dim3((unsigned)((P + NM_T_COLSUM_ROWS - 1) / NM_T_COLSUM_ROWS)), dim3(256), 0, stream, cur, (long long)P, (int)W, out->geo_bias[l]);
        if (l > 0) {
            if (out->geo_weight[l]) NM_T_GEMM(nm_t_gemm(cur, W, 0, s.HT[l - 1], W, 0, out->geo_weight[l], W, W, W, rows, 1), nm_t_split(W, W, rows));
            NM_T_GEMM(nm_t_gemm(cur, W, 1, d->geo_weight[l], W, 0, oth, W, rows, W, W), 1);
            hipLaunchKernelGGL(nm_t_softplus_bwd_kernel, dim3(act_blocks), dim3(256), 0, stream, oth, s.ZU[l - 1], oth, P * W, toff, tangent);
            float* x = cur; cur = oth; oth = x;
        } else {
            if (out->geo_weight[0]) {
                NM_T_GEMM(nm_t_gemm(cur, W, 0, s.X0, td.K0p, 0, s.dW0p, td.K0p, W, td.K0p, P, 1), nm_t_split(W, td.K0p, P));
                if (tangent) NM_T_GEMM(nm_t_gemm(cur + toff, W, 0, s.T0, td.Kt, 0, s.dW0p, td.K0p, W, td.Kt, P, 1), nm_t_split(W, td.Kt, P));
            }
            NM_T_GEMM(nm_t_gemm(cur, W, 1, s.W0p, td.K0p, 0, s.DX0, td.K0p, P, td.K0p, W), 1);
            if (tangent) NM_T_GEMM(nm_t_gemm(cur + toff, W, 1, s.W0p, td.K0p, 0, s.DT0, td.Kt, P, td.Kt, W), 1);
        }
    }
    if (out->geo_weight[0])
        hipLaunchKernelGGL(nm_t_pad_kernel, dim3((unsigned)((W * td.K0p + 255) / 256)), dim3(256), 0, stream, s.dW0p, out->geo_weight[0], (int)W, td.K0, td.K0p, 1);
    hipLaunchKernelGGL(nm_t_geo_input_bwd_kernel, dim3((unsigned)((P + 7) / 8)), dim3(256), 0, stream, td, (long long)P, s.DX0, tangent ? s.DT0 : nullptr, s.ds, s.fg,
                       s.idx, s.w, s.dds, out->geometry_features);
    if (out->indicator_vector || out->indicator_weight)
        hipLaunchKernelGGL(nm_t_distance_bwd_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, (long long)P, s.xyz, s.idx, s.w, g->verts,
                           t->indicator_vector, t->indicator_weight, s.dds, tangent ? s.gvec : nullptr, out->indicator_vector, out->indicator_weight);
    NM_LAUNCH_CHECK();
    return 0;
}

// ============================================================================== renderer
// 4-sample tiles a wave chains along its rays in the regular-grid passes (probes, coarse samples):
// as long as possible (measured on the 800x800 frame: 1 -> 414 ms of K-NN, 8 -> 374, 32 -> 361), but
// never so long that the launch has fewer than ~4 waves per SIMD of the whole chip.
static int nm_chain_tiles(const nm_render_cfg* c, long long R, int P) {
    int vmax = c->chain_tiles > 0 ? c->chain_tiles : 32;  // upper limit (tests compare 1 against the default)
    if (vmax > 64) vmax = 64;
    const long long tiles_p = (P + NM_TILE_SAMPLES - 1) / NM_TILE_SAMPLES, total = ((R + NM_TILE_RAYS - 1) / NM_TILE_RAYS) * tiles_p;
    long long ch = total / 16384;
    if (ch > vmax) ch = vmax;
    if (ch > tiles_p) ch = tiles_p;
    return ch < 1 ? 1 : (int)ch;
}

// Rays per depth-bucket group of an importance-sample pass (n_new samples per ray, <= 8192 keys per sort;
// Morton-ordered rays: 64 -> 170 ms of K-NN per frame, 128..512 -> 166 ms)
static int nm_fine_group_rays(const nm_render_cfg* c, int n_new) {
    int g = c->fine_group_rays;
    if (g != 64 && g != 128 && g != 256 && g != 512) g = 128;
    while (g >= 64 && g * n_new > 8192) g >>= 1;
    return g >= 64 ? g : 0;
}
// Rays per depth-bucket group of the mid-point pass: 64 (8128 samples sorted in 64 KiB of LDS; with the
// zero-weight samples dropped ~3500 of them remain, i.e. the sample density of a 28-ray group; measured
// K-NN time per frame: 16 rays 240 ms, 32 rays 233 ms, 64 rays 210 ms), fewer when the per-ray sample
// count is larger; 0 = no ordering (lists longer than the 8192-key sort).
static int nm_mid_group_rays(const nm_render_cfg* c, int N) {
    int g = c->mid_group_rays;
    if (g != 16 && g != 32 && g != 64) g = 64;
    while (g >= 16 && g * (N - 1) > 8192) g >>= 1;
    return g >= 16 ? g : 0;
}

// the per-ray kernels keep 64 rays' rows in dynamic LDS (nm_ray_lds_bytes(cap) > 64 KiB for cap >= 110)
static int nm_ray_lds_prepare(int cap, size_t* bytes) {
    *bytes = nm_ray_lds_bytes(cap);
    if (cap > NM_MAX_SAMPLES) return nm_fail("per-ray stages: %d samples per ray (limit %d)", cap, NM_MAX_SAMPLES);
    if (*bytes > 160 * 1024) return nm_fail("per-ray stages: %d samples per ray need %zu bytes of LDS (limit 160 KiB)", cap, *bytes);
    // the dynamic-LDS limit is a per-device function attribute: remember what each device was granted
    static std::mutex mu;
    static size_t granted[64] = {0};
    int dev = 0;
    NM_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 0 || dev >= 64 || *bytes > granted[dev]) {
        NM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nm_rays_upsample_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)*bytes));
        NM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nm_rays_finalize_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)*bytes));
        NM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nm_rays_composite_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)*bytes));
        if (dev >= 0 && dev < 64) granted[dev] = *bytes;
    }
    return 0;
}

struct NmWorkspace {
    float *dirn, *nf0, *nf, *d, *sdf, *dmid, *probe;
    float *rays_o_s, *rays_d_s;   // rays in spatial processing order
    unsigned *key_in, *key_out;   // Morton keys before / after the sort
    int *perm_in, *perm;          // perm[i] = caller's index of the i-th ray in processing order
    void* sort_tmp;
    size_t sort_tmp_bytes;
    float *rgb_mid, *nab_pts, *nab_mid;
    int* slot;                    // [R][N] generation position of the sample at each sorted position
    float *radius, *bound, *bound_mid;  // [R][N] K-th-neighbour distance per slot; warm-start bounds
    unsigned short* order;        // depth-bucket lane assignment of one up-sampling pass
    NmScratch slots;  // per-ray slot records (coarse + up-sampling passes), reused by the final pass
    NmScratch pts;    // compact records of the mid-point pass
    float *nab_rot, *dirn_rot;             // texture editing with a rotated reference frame: nablas [pos][3], directions [R][3]
    float *rgb_ref, *edit_w, *edit_share;  // texture editing: reference colours [R][N][3], renormalised painted weights [pos][8], (rest, paint) shares [pos][2]
    unsigned long long* pull_counters;     // (packet counter, waves at work) of the call's pull-form K-NN launches (nm_render_cfg.overlap), NM_PULL_COUNTERS pairs
    size_t bytes;
};
static NmWorkspace nm_carve_ws(void* base, const nm_render_cfg* c, long long R) {
    NmWorkspace w;
    const int N = c->N_samples + c->N_importance;
    char* p = (char*)base;
    size_t o = 0;
    auto take = [&](size_t bytes) { char* r = p + o; o += nm_align(bytes); return r; };
    w.pull_counters = (unsigned long long*)take(2 * NM_PULL_COUNTERS * sizeof(unsigned long long));
    w.rays_o_s = (float*)take((size_t)R * 12);
    w.rays_d_s = (float*)take((size_t)R * 12);
    w.key_in = (unsigned*)take((size_t)R * 4);
    w.key_out = (unsigned*)take((size_t)R * 4);
    w.perm_in = (int*)take((size_t)R * 4);
    w.perm = (int*)take((size_t)R * 4);
    w.sort_tmp_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, w.sort_tmp_bytes, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr, (int*)nullptr, (size_t)R, 0, 30);
    w.sort_tmp = (void*)take(w.sort_tmp_bytes);
    w.dirn = (float*)take((size_t)R * 12);
    w.nf0 = (float*)take((size_t)R * 8);
    w.nf = (float*)take((size_t)R * 8);
    w.d = (float*)take((size_t)R * N * 4);
    w.sdf = (float*)take((size_t)R * N * 4);
    w.dmid = (float*)take((size_t)R * N * 4);
    w.probe = (float*)take((size_t)R * (c->bounded_near_far ? c->probe_grid : 1) * 4);
    w.rgb_mid = (float*)take((size_t)R * N * 12);
    w.nab_pts = (float*)take((size_t)R * N * 12);
    // mid-point list positions: an upper bound over the group sizes nm_mid_group_rays can pick
    const long long slots16 = ((R + 15) / 16) * (((long long)16 * (N - 1) + 63) & ~63LL), slots32 = ((R + 31) / 32) * (((long long)32 * (N - 1) + 63) & ~63LL);
    const long long slots64 = ((R + 63) / 64) * (((long long)64 * (N - 1) + 63) & ~63LL);
    const long long mid_slots = slots16 > slots32 ? (slots16 > slots64 ? slots16 : slots64) : (slots32 > slots64 ? slots32 : slots64);
    w.nab_mid = (float*)take((size_t)(mid_slots > R * N ? mid_slots : R * N) * 12);
    w.slot = (int*)take((size_t)R * N * 4);
    w.radius = (float*)take((size_t)R * N * 4);
    w.bound = (float*)take((size_t)R * N * 4);
    w.bound_mid = (float*)take((size_t)R * N * 4);
    {   // lane assignments of one pass: importance samples (64-ray groups) or mid-points (16-ray groups)
        const size_t n_new = c->N_importance > 0 ? c->N_importance / c->N_upsample_iters : 1;
        const size_t e_fine = (size_t)(R + 512) * n_new + 64;  // any group size: ceil(R/g)*g*n_new <= (R+g)*n_new (g*n_new is a multiple of 64)
        const size_t e16 = (size_t)((R + 15) / 16) * ((16 * (size_t)(N - 1) + 63) & ~(size_t)63), e32 = (size_t)((R + 31) / 32) * ((32 * (size_t)(N - 1) + 63) & ~(size_t)63);
        const size_t e64 = (size_t)((R + 63) / 64) * ((64 * (size_t)(N - 1) + 63) & ~(size_t)63);
        const size_t e_mid = e16 > e32 ? (e16 > e64 ? e16 : e64) : (e32 > e64 ? e32 : e64);
        w.order = (unsigned short*)take((e_fine > e_mid ? e_fine : e_mid) * 2);
    }
    // K-NN records sized by the field's code widths (nm_render_cfg.code_dims; 0 = not given: the maximum, 64 + 64): the sample slots
    // hold ds, grad, the geometry code (144 B at 32 dims), the mid-point records the colour code as well (272 B) -- 63 KB per ray at
    // 128 samples instead of the 148 KB of maximum-width records (a texture-edited call gathers the edited colour codes into the
    // mid-points' geometry slot: sized by the wider of the two)
    int gdim = c->code_dims & 0xffff, cdim = (c->code_dims >> 16) & 0xffff;
    if (gdim <= 0 || gdim > 64) gdim = 64;
    if (cdim <= 0 || cdim > 64) cdim = 64;
    w.slots = nm_carve(p + o, R * N, false, gdim, 0, false);
    o += w.slots.bytes;
    const long long pts_n = mid_slots > R * N ? mid_slots : R * N;
    w.pts = nm_carve(p + o, pts_n, c->n_edit > 0, (c->n_edit > 0 && cdim > gdim) ? cdim : gdim, cdim, false);   // (texture editing needs the neighbour lists)
    o += w.pts.bytes;
    w.rgb_ref = (float*)take(c->n_edit > 0 ? (size_t)R * N * 12 : 0);
    w.edit_w = (float*)take(c->n_edit > 0 ? (size_t)pts_n * 32 : 0);
    w.edit_share = (float*)take(c->n_edit > 0 ? (size_t)pts_n * 8 : 0);
    bool any_rot = false;
    for (int i = 0; i < c->n_edit; ++i) any_rot = any_rot || c->edit_use_rot[i] != 0;
    w.nab_rot = (float*)take(any_rot ? (size_t)pts_n * 12 : 0);
    w.dirn_rot = (float*)take(any_rot ? (size_t)R * 12 : 0);
    w.bytes = o;
    return w;
}

static int nm_check_cfg(const nm_render_cfg* c) {
    if (!c) return nm_fail("nm_render: cfg is NULL");
    if (c->N_samples < 2 || c->N_importance < 0 || c->N_samples + c->N_importance > NM_MAX_SAMPLES) return nm_fail("nm_render: N_samples=%d N_importance=%d unsupported (sum <= %d)", c->N_samples, c->N_importance, NM_MAX_SAMPLES);
    if (c->N_importance > 0 && (c->N_upsample_iters < 1 || c->N_importance % c->N_upsample_iters)) return nm_fail("nm_render: N_importance %% N_upsample_iters != 0");
    if (c->bounded_near_far && (c->probe_grid < 2 || c->probe_grid > 4096)) return nm_fail("nm_render: probe_grid=%d", c->probe_grid);
    if (c->n_edit < 0 || c->n_edit > NM_MAX_EDIT) return nm_fail("nm_render: n_edit=%d (0..%d)", c->n_edit, NM_MAX_EDIT);
    if (c->overlap < 0 || c->overlap > 1 || c->knn_keep < 0 || c->knn_keep > 8 || c->mlp_prio < 0 || c->mlp_prio > 3)
        return nm_fail("nm_render: overlap=%d knn_keep=%d mlp_prio=%d (0..1, 0..8, 0..3)", c->overlap, c->knn_keep, c->mlp_prio);
    for (int i = 0; i < c->n_edit; ++i)
        if (!c->edit_field[i] || !c->edit_mask[i] || !c->edit_color_features) return nm_fail("nm_render: texture editing: NULL reference field / mask / colour table");
    return 0;
}

int64_t nm_render_workspace_bytes(const nm_render_cfg* cfg, int64_t R) {
    if (nm_check_cfg(cfg) || R < 1) return -1;
    return (int64_t)nm_carve_ws(nullptr, cfg, R).bytes;
}

int nm_render_rays(nm_field_t f, nm_grid_t g, const nm_field_tables* t, const float* rays_o, const float* rays_d, int64_t R,
                   const nm_render_cfg* c, float* rgb, float* depth, float* acc, float* normals, const nm_render_debug* dbg,
                   void* workspace, nm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (nm_check_field_args(f, g, t, "nm_render_rays") || nm_check_cfg(c)) return 1;
    const bool sample_only = (c->flags & NM_RENDER_SAMPLE_ONLY) != 0;
    if (R < 0 || (R > 0 && (!rays_o || !rays_d || !workspace))) return nm_fail("nm_render_rays: bad arguments");
    if (sample_only) {
        if (R > 0 && (!dbg || !dbg->d_all)) return nm_fail("nm_render_rays: NM_RENDER_SAMPLE_ONLY needs dbg->d_all");
    } else {
        if (R > 0 && (!rgb || !depth || !acc)) return nm_fail("nm_render_rays: bad arguments");
        if (c->calc_normal && !normals) return nm_fail("nm_render_rays: calc_normal set but normals is NULL");
    }
    if (R == 0) return 0;
    if (c->code_dims) {   // the workspace was sized for these code widths: the fields rendered through it must fit
        const int gd = c->code_dims & 0xffff, cd = (c->code_dims >> 16) & 0xffff;
        if (f->geo.gdim > gd || f->col.cdim > cd) return nm_fail("nm_render_rays: cfg.code_dims = (%d, %d) but the field has code widths (%d, %d)", gd, cd, f->geo.gdim, f->col.cdim);
        for (int e = 0; e < c->n_edit; ++e)
            if (c->edit_field[e]->col.cdim > cd) return nm_fail("nm_render_rays: reference field %d has a colour code of %d > cfg.code_dims %d", e, c->edit_field[e]->col.cdim, cd);
    }
    const NmWorkspace ws = nm_carve_ws(workspace, c, R);
    const int N = c->N_samples + c->N_importance, cap = N;
    const dim3 rgrid(nm_blocks(R, 64)), rblock(64);
    // several chunks in flight (cfg.overlap): pull-form K-NN launches that yield to the other chunks' MLP launches (nm_kernels.h)
    NmOverlap ov_state, *ov = nullptr;
    if (c->overlap) {
        if (nm_yield_state(&ov_state.y, &ov_state.simds)) return 1;
        ov_state.counters = ws.pull_counters;
        ov_state.cap = c->knn_keep > 0 ? c->knn_keep : 1;
        ov_state.prio = c->mlp_prio;
        NM_HIP(hipMemsetAsync(ws.pull_counters, 0, 2 * NM_PULL_COUNTERS * sizeof(unsigned long long), stream));
        ov = &ov_state;
    }
    const dim3 rblock_io(NM_RAY_IO_THREADS);   // upsample / finalize: 64 rays per workgroup, every thread moves rows between HBM and LDS

    // processing order: rays sorted by the Morton code of their closest approach to the scene centre (see
    // nm_ray_keys_kernel); per-ray outputs (pixels, debug arrays) go back to the caller's order through perm
    const int* perm = nullptr;
    if (R >= 256 && !(c->flags & NM_RENDER_NO_RAY_SORT)) {
        hipLaunchKernelGGL(nm_ray_keys_kernel, dim3(nm_blocks(R, 256)), dim3(256), 0, stream, rays_o, rays_d, (long long)R,
                           1.0f / fmaxf(c->obj_bounding_radius, 1e-6f), ws.key_in, ws.perm_in);
        NM_LAUNCH_CHECK();
        size_t tmp = ws.sort_tmp_bytes;
        NM_HIP(rocprim::radix_sort_pairs(ws.sort_tmp, tmp, (const unsigned*)ws.key_in, ws.key_out, (const int*)ws.perm_in, ws.perm, (size_t)R, 0, 30, stream));
        hipLaunchKernelGGL(nm_ray_gather_kernel, dim3(nm_blocks(R, 256)), dim3(256), 0, stream, rays_o, rays_d, ws.perm, (long long)R, ws.rays_o_s, ws.rays_d_s);
        NM_LAUNCH_CHECK();
        rays_o = ws.rays_o_s;
        rays_d = ws.rays_d_s;
        perm = ws.perm;
    }
    // rays: normalise directions, sphere near/far (renderer.py:153, rend_util.py:179-199)
    hipLaunchKernelGGL(nm_rays_setup_kernel, rgrid, rblock, 0, stream, rays_o, rays_d, (long long)R, c->obj_bounding_radius, ws.dirn, ws.nf0);
    NM_LAUNCH_CHECK();
    NmPointSrc src;
    memset(&src, 0, sizeof(src));
    src.rays_o = rays_o;
    src.dirn = ws.dirn;
    src.dstride = cap;
    const float* nf = ws.nf0;
    if (c->bounded_near_far) {  // renderer.py:66-102
        if (!(c->flags & NM_RENDER_FULL_PROBES)) {  // first / last hit only (nm_probe_bounds_kernel)
            NmProfScope prof(NM_K_DISTANCE, 0, stream, NM_CNT_PROBE);  // units = probes actually searched (device counter)
            static_assert(NM_PROBE_STEP == 8, "launch geometry below is for 8 probes per ray and step");
            if (nm_pull_ok(ov)) {
                const long long packets = (R + 7) / 8;
                hipLaunchKernelGGL(nm_probe_bounds_pull_kernel<8>, dim3(nm_pull_grid(ov, packets, NM_KNN_WAVES_PROBE)), dim3(64), 0, stream, g->view, nm_pull_for(ov, packets),
                                   rays_o, ws.dirn, ws.nf0, (long long)R, c->probe_grid, c->probe_thresh, g->verts, t->indicator_vector, t->indicator_weight, ws.nf,
                                   nm_prof_counter(NM_CNT_PROBE));
            } else
            hipLaunchKernelGGL(nm_probe_bounds_kernel<8>, dim3(nm_blocks((R + 7) / 8, 4)), dim3(256), 0, stream, g->view, rays_o, ws.dirn, ws.nf0,
                                   (long long)R, c->probe_grid, c->probe_thresh, g->verts, t->indicator_vector, t->indicator_weight, ws.nf,
                                   nm_prof_counter(NM_CNT_PROBE));
            NM_LAUNCH_CHECK();
        } else {  // every probe, then the reduction (the staged API's form; kept for A/B measurements)
            src.mode = 2;
            src.chain = nm_chain_tiles(c, R, c->probe_grid);
            src.P = c->probe_grid;
            src.nearfar = ws.nf0;
            src.depth_out = nullptr;
            src.bound = nullptr;
            src.out_stride = 0;
            src.out_off = 0;
            if (nm_launch_distance(g, src, (long long)R * c->probe_grid, t->indicator_vector, t->indicator_weight, ws.probe, nullptr, nullptr, nullptr, nullptr, stream, nullptr, NM_NO_GATHER, false, ov)) return 1;
            hipLaunchKernelGGL(nm_rays_bounds_kernel, rgrid, rblock, 0, stream, ws.probe, (long long)R, c->probe_grid, c->probe_thresh, ws.nf0, ws.nf);
            NM_LAUNCH_CHECK();
        }
        nf = ws.nf;
    }
    if (c->near_bypass >= 0.f || c->far_bypass >= 0.f) {  // renderer.py:172-175
        if (nf == ws.nf0) {
            NM_HIP(hipMemcpyAsync(ws.nf, ws.nf0, (size_t)R * 8, hipMemcpyDeviceToDevice, stream));
            nf = ws.nf;
        }
        hipLaunchKernelGGL(nm_rays_bypass_kernel, rgrid, rblock, 0, stream, (long long)R, c->near_bypass, c->far_bypass, ws.nf);
        NM_LAUNCH_CHECK();
    }
    // coarse samples + SDF (renderer.py:193-207).  The K-NN records of the coarse and up-sampling
    // passes are written to per-ray SLOTS (slot = position at which the sample was generated):
    // the final pass over all N samples visits exactly these points again, so it reuses the
    // records through the sort permutation instead of searching a second time (same input, same
    // deterministic kernel => bit-identical record), and every later search is warm-started with
    // the cached K-th-neighbour radius of the neighbouring sample on its ray.
    const bool want_grad = c->calc_normal != 0 && !sample_only;   // (the sample placement itself never needs a nabla)
    // (decided further down; needed here already) zero-weight skip active => the nablas of the N sample points are
    // evaluated AFTER the sampling passes, and only where the visibility weight is not zero (see below)
    const bool lazy_nabla_possible = want_grad && f->precision == 2 && !(dbg && (dbg->nablas_all || dbg->radiance)) &&
                                     nm_mid_group_rays(c, c->N_samples + c->N_importance) > 0 &&
                                     !(c->flags & (NM_RENDER_NO_MID_ORDER | NM_RENDER_NO_ZERO_SKIP | NM_RENDER_EAGER_NABLAS));
    const NmGather ga_slots = {t->geometry_features, f->geo.gdim, ws.slots.fg, nullptr, 0, nullptr};
    src.mode = 2;
    src.chain = nm_chain_tiles(c, R, c->N_samples);
    src.P = c->N_samples;
    src.nearfar = nf;
    src.depth_out = ws.d;
    src.doff = 0;
    src.bound = nullptr;
    src.out_stride = cap;
    src.out_off = 0;
    if (nm_launch_distance(g, src, (long long)R * c->N_samples, t->indicator_vector, t->indicator_weight, ws.slots.ds, nullptr, nullptr, nullptr, want_grad ? ws.slots.grad : nullptr, stream, ws.radius, ga_slots, false, ov)) return 1;
    // With normals requested the sampling passes already run the tangent form of the geometry MLP:
    // forward_with_nablas(pts) (renderer.py:271-276) is evaluated at exactly these points, and the
    // value rows of the tangent kernel are bit-identical to the forward-only kernel, so the nablas
    // are written per slot now (into the buffer the mid-point pass overwrites later) and merely
    // permuted at the end -- instead of a second pass of N evaluations per ray.
    const bool eager_nabla = want_grad && !lazy_nabla_possible;   // tangent rows in the sampling passes themselves
    float* nab_slot = eager_nabla ? ws.nab_mid : nullptr;
    {
        const NmRecMap rm = {c->N_samples, cap, 0, nullptr, 0};
        if (nm_launch_geo(f, ws.slots.fg, ws.slots.ds, want_grad ? ws.slots.grad : nullptr, (long long)R * c->N_samples, eager_nabla, ws.sdf, c->N_samples, cap, 0, nab_slot, stream, rm, 1, NM_NO_SLOTS, false, ov)) return 1;
    }
    if (dbg && dbg->sdf_coarse) {
        hipLaunchKernelGGL(nm_rows_out_kernel, dim3(nm_blocks(R * c->N_samples, 256)), dim3(256), 0, stream, ws.sdf, (long long)R, c->N_samples, cap, perm, dbg->sdf_coarse);
        NM_LAUNCH_CHECK();
    }
    // hierarchical up-sampling (renderer.py:208-258)
    int n = c->N_samples, pending = 0;
    size_t ray_lds = 0;
    if (nm_ray_lds_prepare(cap, &ray_lds)) return 1;
    const int mid_g = nm_mid_group_rays(c, N);  // rays per depth-bucket group of the mid-point pass
    const bool use_order = mid_g > 0 && !(c->flags & NM_RENDER_NO_MID_ORDER);
    const bool skip_zero = use_order && !(dbg && dbg->radiance) && !(c->flags & NM_RENDER_NO_ZERO_SKIP);  // all radiances requested => evaluate all
    if (c->N_importance > 0) {
        const int n_new = c->N_importance / c->N_upsample_iters;
        for (int it = 0; it < c->N_upsample_iters; ++it) {
            hipLaunchKernelGGL(nm_rays_upsample_kernel, rgrid, rblock_io, ray_lds, stream, ws.d, ws.sdf, ws.slot, ws.radius, ws.bound, (long long)R, cap, n, pending, it, n_new,
                               c->u_rand ? c->u_rand + (size_t)it * R * n_new : (const float*)nullptr, perm);
            NM_LAUNCH_CHECK();
            src.mode = 1;
            src.P = n_new;
            src.depth = ws.d;
            src.doff = n;
            src.depth_out = nullptr;
            src.bound = ws.bound;
            src.out_stride = cap;
            src.out_off = n;
            src.order = nullptr;
            const int fine_g = nm_fine_group_rays(c, n_new);
            if (fine_g > 0) {
                hipLaunchKernelGGL(nm_rays_order_kernel, dim3((unsigned)((R + fine_g - 1) / fine_g)), dim3(256), nm_order_lds_bytes(fine_g * n_new), stream, ws.d, (long long)R, cap, n, n_new, fine_g, ws.order, (const float*)nullptr, (unsigned long long*)nullptr);
                NM_LAUNCH_CHECK();
                src.order = ws.order;
                src.order_rays = fine_g;
            }
            if (nm_launch_distance(g, src, (long long)R * n_new, t->indicator_vector, t->indicator_weight, ws.slots.ds, nullptr, nullptr, nullptr, want_grad ? ws.slots.grad : nullptr, stream, ws.radius, ga_slots, false, ov)) return 1;
            const NmRecMap rm = {n_new, cap, n, nullptr, 0};
            if (nm_launch_geo(f, ws.slots.fg, ws.slots.ds, want_grad ? ws.slots.grad : nullptr, (long long)R * n_new, eager_nabla, ws.sdf, n_new, cap, n, nab_slot, stream, rm, 1, NM_NO_SLOTS, false, ov)) return 1;
            n += n_new;
            pending = n_new;
        }
    }
    hipLaunchKernelGGL(nm_rays_finalize_kernel, rgrid, rblock_io, ray_lds, stream, ws.d, ws.sdf, ws.slot, ws.radius, (long long)R, cap, n, pending, ws.dmid, ws.bound_mid, t->s, skip_zero ? ws.bound : (float*)nullptr, c->weight_eps > 0.f ? c->weight_eps : 0.f);
    NM_LAUNCH_CHECK();
    if (sample_only) {   // the caller continues from the sorted depths (training: field queries with autograd)
        auto rows_out = [&](const float* src_, int n_, int src_stride, float* dst) {
            hipLaunchKernelGGL(nm_rows_out_kernel, dim3(nm_blocks(R * n_, 256)), dim3(256), 0, stream, src_, (long long)R, n_, src_stride, perm, dst);
        };
        if (dbg->near_far) rows_out(nf, 2, 2, dbg->near_far);
        rows_out(ws.d, N, cap, dbg->d_all);
        if (dbg->sdf_all) rows_out(ws.sdf, N, cap, dbg->sdf_all);
        NM_LAUNCH_CHECK();
        return 0;
    }
    // SDF (+ nablas) at all N sample points (renderer.py:264, 271-276): no new search and no new
    // MLP pass -- the SDF values merged above ARE forward_with_nablas(pts)[0] (same points, same
    // arithmetic), the nablas are brought into sorted order through the slot permutation.
    if (eager_nabla) {
        hipLaunchKernelGGL(nm_permute_rows3_kernel, dim3(nm_blocks(R * N, 256)), dim3(256), 0, stream, nab_slot, ws.slot, (long long)R, cap, N, ws.nab_pts);
        NM_LAUNCH_CHECK();
    }
    // SDF + nabla + radiance at the N-1 mid-points (renderer.py:266-267, 279-282).
    // The mid-points are handed to the waves by depth buckets over 16 adjacent rays (ws.order), and
    // those whose visibility weight is EXACTLY zero are dropped from the list: their colour would be
    // multiplied by 0 in the compositing sum (nm_ray_composite), so neither their K-NN search nor their
    // geometry / colour MLPs can change a bit of the result (alpha = 0 wherever the SDF does not
    // decrease along the ray: more than half of the mid-points on the benchmark scene).  With the list
    // in use, records and nablas are stored at list positions and the colours scattered back.
    src.order = nullptr;
    src.out_by_slot = 0;
    NmSlotMap smap = NM_NO_SLOTS;
    long long mid_pts = (long long)R * (N - 1);
    if (use_order) {
        hipLaunchKernelGGL(nm_rays_order_kernel, dim3((unsigned)((R + mid_g - 1) / mid_g)), dim3(256), nm_order_lds_bytes(mid_g * (N - 1)), stream, ws.dmid, (long long)R, cap, 0, N - 1, mid_g, ws.order,
                           skip_zero ? (const float*)ws.bound : (const float*)nullptr, skip_zero ? nm_prof_counter(NM_CNT_MID) : (unsigned long long*)nullptr);
        NM_LAUNCH_CHECK();
        src.order = ws.order;
        src.order_rays = mid_g;
        src.out_by_slot = 1;
        smap.order = ws.order;
        smap.G = mid_g;
        smap.P = N - 1;
        smap.E = (mid_g * (N - 1) + 63) & ~63;
        mid_pts = ((long long)(R + mid_g - 1) / mid_g) * smap.E;  // list positions (incl. padding)
    }
    // Nablas of the N sample points (renderer.py:271-276) where they can reach the normals at all: sample j enters
    // normals_volume with visibility weight w_j (renderer.py:336-341), the same weight the mid-point list was cut by, so
    // the list's (ray, j) entries are exactly the sample points that need a nabla.  Their K-NN records sit in the slot
    // arrays of the sampling passes (which therefore ran the forward-only MLP: 64 instead of 32 points per tile, no
    // tangent rows); the tangent kernel reads them through the slot permutation and writes nab_pts[ray][j].
    // The sample points' nabla launch (matrix pipe) and the mid-points' search (vector issue) are independent: the search goes to a side stream
    // of this call and runs in what the MLP workgroups leave free (cfg.flags & NM_RENDER_NO_FORK: in order, as before; same results either way).
    NmSide sd;
    const bool fork = want_grad && !eager_nabla && !ov && !(c->flags & NM_RENDER_NO_FORK) && nm_side_for(stream, &sd);
    if (want_grad && !eager_nabla) {
        if (!(use_order && skip_zero)) return nm_fail("nm_render_rays: internal: lazy nablas need the zero-weight list");
        if (fork) NM_HIP(hipEventRecord(sd.fork, stream));
        const NmRecMap rm = {N - 1, cap, 0, ws.slot, 1};
        if (nm_launch_geo(f, ws.slots.fg, ws.slots.ds, ws.slots.grad, mid_pts, true, nullptr, 1, N, 0, ws.nab_pts, stream, rm, 1, smap, true, ov)) return 1;
    }
    hipStream_t knn_stream = stream;
    if (fork) {
        NM_HIP(hipStreamWaitEvent(sd.side, sd.fork, 0));
        knn_stream = sd.side;
    }
    src.mode = 1;
    src.P = N - 1;
    src.depth = ws.dmid;
    src.doff = 0;
    src.depth_out = nullptr;
    src.bound = ws.bound_mid;
    src.out_stride = 0;
    src.out_off = 0;
    {
        const NmGather ga_mid = {t->geometry_features, f->geo.gdim, ws.pts.fg, t->color_features, f->col.cdim, ws.pts.ft};
        if (nm_launch_distance(g, src, (long long)R * (N - 1), t->indicator_vector, t->indicator_weight, ws.pts.ds, c->n_edit > 0 ? ws.pts.idx : nullptr, nullptr,
                               c->n_edit > 0 ? ws.pts.w : nullptr, ws.pts.grad, knn_stream, nullptr, ga_mid, skip_zero, ov)) return 1;
    }
    if (fork) {
        NM_HIP(hipEventRecord(sd.join, sd.side));
        NM_HIP(hipStreamWaitEvent(stream, sd.join, 0));
    }
    if (nm_launch_geo(f, ws.pts.fg, ws.pts.ds, ws.pts.grad, mid_pts, true, nullptr, 1, 1, 0, ws.nab_mid, stream, NM_COMPACT, 0, smap, skip_zero, ov)) return 1;
    if (nm_launch_col(f, ws.pts.ft, ws.pts.ds, ws.nab_mid, ws.dirn, N - 1, mid_pts, ws.rgb_mid, stream, smap, skip_zero, ov)) return 1;
    // Texture editing (texture_neumesh.py:79-121): per reference model, the painted share of every mid-point's interpolation
    // weight, the reference colour from the edited colour table under the painted neighbours' renormalised weights, the blend.
    for (int e = 0; e < c->n_edit; ++e) {
        nm_field_t rf = c->edit_field[e];
        if (rf->col.cdim != f->col.cdim || rf->col.in_dim != f->col.in_dim) return nm_fail("nm_render_rays: texture editing: reference model %d has another colour configuration", e);
        const bool rot = c->edit_use_rot[e] != 0;
        NmRot3 rm;
        for (int i = 0; i < 9; ++i) rm.m[i] = c->edit_rot[e][i];
        hipLaunchKernelGGL(nm_edit_prepare_kernel, dim3(nm_blocks(mid_pts, 256)), dim3(256), 0, stream, mid_pts, smap, ws.pts.idx, ws.pts.w, c->edit_mask[e],
                           c->edit_color_features, f->col.cdim, ws.edit_w, ws.edit_share, ws.pts.fg, rm, rot ? ws.nab_mid : (const float*)nullptr, ws.nab_rot);
        NM_LAUNCH_CHECK();
        if (rot) {
            hipLaunchKernelGGL(nm_rotate_rows3_kernel, dim3(nm_blocks(R, 256)), dim3(256), 0, stream, (long long)R, rm, ws.dirn, ws.dirn_rot);
            NM_LAUNCH_CHECK();
        }
        if (nm_launch_col(rf, ws.pts.fg, ws.pts.ds, rot ? ws.nab_rot : ws.nab_mid, rot ? ws.dirn_rot : ws.dirn, N - 1, mid_pts, ws.rgb_ref, stream, smap, skip_zero, ov)) return 1;
        hipLaunchKernelGGL(nm_edit_blend_kernel, dim3(nm_blocks(mid_pts, 256)), dim3(256), 0, stream, mid_pts, smap, N - 1, ws.edit_share, ws.rgb_ref, ws.rgb_mid);
        NM_LAUNCH_CHECK();
    }
    // alpha + compositing (renderer.py:278, 302-333)
    hipLaunchKernelGGL(nm_rays_composite_kernel, rgrid, rblock_io, ray_lds, stream, ws.sdf, ws.d, (long long)R, cap, N, t->s, ws.rgb_mid,
                       c->calc_normal ? ws.nab_pts : (const float*)nullptr, c->white_bkgd, rgb, depth, acc, c->calc_normal ? normals : (float*)nullptr,
                       skip_zero ? (const float*)ws.bound : (const float*)nullptr, perm);
    NM_LAUNCH_CHECK();
    if (dbg) {  // per-ray debug rows, back in the caller's ray order
        auto rows_out = [&](const float* src, int n, int src_stride, float* dst) {
            hipLaunchKernelGGL(nm_rows_out_kernel, dim3(nm_blocks(R * n, 256)), dim3(256), 0, stream, src, (long long)R, n, src_stride, perm, dst);
        };
        if (dbg->near_far) rows_out(nf, 2, 2, dbg->near_far);
        if (dbg->d_all) rows_out(ws.d, N, cap, dbg->d_all);
        if (dbg->sdf_all) rows_out(ws.sdf, N, cap, dbg->sdf_all);
        if (dbg->nablas_all && c->calc_normal) rows_out(ws.nab_pts, 3 * N, 3 * N, dbg->nablas_all);
        if (dbg->radiance) rows_out(ws.rgb_mid, 3 * (N - 1), 3 * (N - 1), dbg->radiance);
        NM_LAUNCH_CHECK();
    }
    return 0;
}

// ============================================================================ first-hit surface points
// models/ray_casting.py:45-200 for a NeuMesh field (nm_surface.h).  The host loop below sizes every launch by the number of rays
// still walking, which it reads back after each block of NM_SURF_BLOCK proposals: this entry point synchronises the stream
// (N_steps / 16 times + once per chunk); everything between is stream-ordered.
#define NM_SURF_CHUNK (1 << 18)   // rays per internal chunk (records of a walk step: 136 B per ray and proposal)
struct NmSurfWs {
    NmSurfState st;
    int *ids_a, *ids_b, *count, *perm_in, *perm;
    unsigned *key_in, *key_out;
    unsigned char *keep, *hit;
    float *nearfar, *ds, *fg, *val, *d_pred, *xyz;
    void *sel_tmp, *sort_tmp;
    size_t sel_tmp_bytes, sort_tmp_bytes, bytes;
};
static NmSurfWs nm_surf_carve(void* base, long long Rc, int gdim) {
    NmSurfWs w;
    memset(&w, 0, sizeof(w));
    char* p = (char*)base;
    size_t o = 0;
    auto take = [&](size_t n) { char* q = p ? p + o : nullptr; o += nm_align(n); return (void*)q; };
    const size_t R = (size_t)Rc;
    w.st.idx = (int*)take(R * 4);
    w.st.f_high = (float*)take(R * 4); w.st.f_low = (float*)take(R * 4); w.st.d_high = (float*)take(R * 4); w.st.d_low = (float*)take(R * 4);
    w.st.val0 = (float*)take(R * 4); w.st.prev = (float*)take(R * 4);
    w.ids_a = (int*)take(R * 4); w.ids_b = (int*)take(R * 4); w.count = (int*)take(256);
    w.perm_in = (int*)take(R * 4); w.perm = (int*)take(R * 4); w.key_in = (unsigned*)take(R * 4); w.key_out = (unsigned*)take(R * 4);
    w.keep = (unsigned char*)take(R); w.hit = (unsigned char*)take(R);
    w.nearfar = (float*)take(R * 8);
    w.ds = (float*)take(R * NM_SURF_BLOCK * 4);
    w.fg = (float*)take(R * NM_SURF_BLOCK * (size_t)gdim * 4);
    w.val = (float*)take(R * NM_SURF_BLOCK * 4);
    w.d_pred = (float*)take(R * 4); w.xyz = (float*)take(R * 12);
    (void)rocprim::select(nullptr, w.sel_tmp_bytes, (const int*)nullptr, (const unsigned char*)nullptr, (int*)nullptr, (int*)nullptr, R);
    (void)rocprim::radix_sort_pairs(nullptr, w.sort_tmp_bytes, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr, (int*)nullptr, R, 0, 30);
    w.sel_tmp = take(w.sel_tmp_bytes);
    w.sort_tmp = take(w.sort_tmp_bytes);
    w.bytes = o;
    return w;
}
__global__ void nm_surf_fill_nearfar_kernel(long long R, const float* __restrict__ src, float near_s, float far_s, float* __restrict__ dst) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    dst[2 * r] = src ? src[2 * r] : near_s;
    dst[2 * r + 1] = src ? src[2 * r + 1] : far_s;
}
__global__ void nm_surf_hit_by_pos_kernel(const int* __restrict__ ids, int n, const unsigned char* __restrict__ hit_by_ray, unsigned char* __restrict__ flag) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a < n) flag[a] = hit_by_ray[ids[a]];
}

static int nm_check_surface_cfg(const nm_surface_cfg* c) {
    if (!c) return nm_fail("nm_surface: cfg is NULL");
    if (c->N_steps < 2 || c->N_steps > 65536) return nm_fail("nm_surface: N_steps=%d out of [2,65536]", c->N_steps);
    if (c->n_secant_steps < -1 || c->n_secant_steps > 64) return nm_fail("nm_surface: n_secant_steps=%d out of [-1,64]", c->n_secant_steps);
    return 0;
}
int64_t nm_surface_workspace_bytes(nm_field_t f, int64_t R) {
    if (!f || R < 1) return -1;
    return (int64_t)nm_surf_carve(nullptr, std::min<long long>(R, NM_SURF_CHUNK), f->geo.gdim).bytes;
}

int nm_surface_hits(nm_field_t f, nm_grid_t g, const nm_field_tables* t, const float* rays_o, const float* rays_d, int64_t R,
                    const float* near_far, const nm_surface_cfg* c, float* d_out, float* pt_out, uint8_t* mask, uint8_t* mask_sign_change,
                    void* workspace, nm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (nm_check_field_args(f, g, t, "nm_surface_hits") || nm_check_surface_cfg(c)) return 1;
    if (R < 0 || (R > 0 && (!rays_o || !rays_d || !d_out || !pt_out || !mask || !mask_sign_change || !workspace))) return nm_fail("nm_surface_hits: bad arguments");
    const int N = c->N_steps;
    const NmGather ga_none = NM_NO_GATHER;
    (void)ga_none;
    for (int64_t c0 = 0; c0 < R; c0 += NM_SURF_CHUNK) {
        const long long Rc = std::min<long long>(R - c0, NM_SURF_CHUNK);
        const NmSurfWs ws = nm_surf_carve(workspace, std::min<long long>(R, NM_SURF_CHUNK), f->geo.gdim);
        const float* ro = rays_o + 3 * c0;
        const float* rd = rays_d + 3 * c0;
        const dim3 rg(nm_blocks(Rc, 256)), rb(256);
        hipLaunchKernelGGL(nm_surf_fill_nearfar_kernel, rg, rb, 0, stream, Rc, near_far ? near_far + 2 * c0 : (const float*)nullptr, c->near, c->far, ws.nearfar);
        hipLaunchKernelGGL(nm_surf_init_kernel, rg, rb, 0, stream, Rc, ws.st, (int*)nullptr);
        // walking order: rays sorted by the Morton code of their closest approach to the scene centre (as nm_render_rays), so that the
        // 16 consecutive rays of a distance-kernel tile are neighbours in space; the stable compaction below keeps that order
        hipLaunchKernelGGL(nm_ray_keys_kernel, rg, rb, 0, stream, ro, rd, Rc, 1.0f / fmaxf(c->scene_radius, 1e-6f), ws.key_in, ws.perm_in);
        NM_LAUNCH_CHECK();
        {
            size_t tmp = ws.sort_tmp_bytes;
            NM_HIP(rocprim::radix_sort_pairs(ws.sort_tmp, tmp, (const unsigned*)ws.key_in, ws.key_out, (const int*)ws.perm_in, ws.perm, (size_t)Rc, 0, 30, stream));
        }
        NM_HIP(hipMemcpyAsync(ws.ids_a, ws.perm, (size_t)Rc * 4, hipMemcpyDeviceToDevice, stream));
        int* ids = ws.ids_a;
        int* ids_next = ws.ids_b;
        int nA = (int)Rc;
        for (int k0 = 0; k0 < N && nA > 0; k0 += NM_SURF_BLOCK) {
            const int n = std::min(NM_SURF_BLOCK, N - k0);
            NmPointSrc src;
            memset(&src, 0, sizeof(src));
            src.mode = 2;
            src.P = n;
            src.rays_o = ro;
            src.dirn = rd;
            src.nearfar = ws.nearfar;
            src.ray_index = ids;
            src.p_off = k0;
            src.p_total = N;
            src.chain = 1;   // (chained, warm-started tiles measured slower here: 231 / 205 / 183 ms per frame with 4 / 2 / 1 tiles per wave --
                             //  a block is only 16 proposals long, and four times as many short waves fill the chip better)
            const NmGather ga = {t->geometry_features, f->geo.gdim, ws.fg, nullptr, 0, nullptr};
            if (nm_launch_distance(g, src, (long long)nA * n, t->indicator_vector, t->indicator_weight, ws.ds, nullptr, nullptr, nullptr, nullptr, stream, nullptr, ga)) return 1;
            if (nm_launch_geo(f, ws.fg, ws.ds, nullptr, (long long)nA * n, false, ws.val, 1, 1, 0, nullptr, stream)) return 1;
            hipLaunchKernelGGL(nm_surf_scan_kernel, dim3(nm_blocks(nA, 256)), dim3(256), 0, stream, (const int*)ids, nA, (const float*)ws.val, n, k0, N, c->logit_tau,
                               (const float*)ws.nearfar, 0.f, 0.f, ws.st, ws.keep);
            NM_LAUNCH_CHECK();
            if (k0 + n >= N) break;
            size_t tmp = ws.sel_tmp_bytes;
            NM_HIP(rocprim::select(ws.sel_tmp, tmp, (const int*)ids, (const unsigned char*)ws.keep, ids_next, ws.count, (size_t)nA, stream));
            int cnt = 0;
            NM_HIP(hipMemcpyAsync(&cnt, ws.count, sizeof(int), hipMemcpyDeviceToHost, stream));
            NM_HIP(hipStreamSynchronize(stream));
            nA = cnt;
            std::swap(ids, ids_next);
        }
        // ---- the hits, in walking order; secant refinement on them
        hipLaunchKernelGGL(nm_surf_hit_flags_kernel, rg, rb, 0, stream, Rc, ws.st, ws.hit, ws.perm_in);
        hipLaunchKernelGGL(nm_surf_hit_by_pos_kernel, rg, rb, 0, stream, (const int*)ws.perm, (int)Rc, (const unsigned char*)ws.hit, ws.keep);
        NM_LAUNCH_CHECK();
        int nH = 0;
        if (c->n_secant_steps > 0) {
            size_t tmp = ws.sel_tmp_bytes;
            NM_HIP(rocprim::select(ws.sel_tmp, tmp, (const int*)ws.perm, (const unsigned char*)ws.keep, ws.ids_a, ws.count, (size_t)Rc, stream));
            NM_HIP(hipMemcpyAsync(&nH, ws.count, sizeof(int), hipMemcpyDeviceToHost, stream));
            NM_HIP(hipStreamSynchronize(stream));
        }
        for (int it = 0; it < c->n_secant_steps && nH > 0; ++it) {
            hipLaunchKernelGGL(nm_surf_secant_points_kernel, dim3(nm_blocks(nH, 256)), dim3(256), 0, stream, (const int*)ws.ids_a, nH, ws.st, ro, rd, ws.d_pred, ws.xyz);
            NM_LAUNCH_CHECK();
            const NmGather ga = {t->geometry_features, f->geo.gdim, ws.fg, nullptr, 0, nullptr};
            if (nm_launch_distance(g, nm_src_xyz(ws.xyz, nH), nH, t->indicator_vector, t->indicator_weight, ws.ds, nullptr, nullptr, nullptr, nullptr, stream, nullptr, ga)) return 1;
            if (nm_launch_geo(f, ws.fg, ws.ds, nullptr, nH, false, ws.val, 1, 1, 0, nullptr, stream)) return 1;
            hipLaunchKernelGGL(nm_surf_secant_update_kernel, dim3(nm_blocks(nH, 256)), dim3(256), 0, stream, (const int*)ws.ids_a, nH, ws.st, (const float*)ws.d_pred,
                               (const float*)ws.val, c->logit_tau);
            NM_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(nm_surf_finish_kernel, rg, rb, 0, stream, Rc, ws.st, (const unsigned char*)ws.hit, ro, rd, (const float*)ws.nearfar, 0.f,
                           c->n_secant_steps >= 0 ? 1 : 0, c->fill_inf, d_out + c0, pt_out + 3 * c0, mask + c0, mask_sign_change + c0);
        NM_LAUNCH_CHECK();
        if (c0 + NM_SURF_CHUNK < R) NM_HIP(hipStreamSynchronize(stream));   // the next chunk reuses the workspace
    }
    return 0;
}

// ==================================================================== per-ray stages (staged API)
int nm_rays_setup(const float* rays_o, const float* rays_d, int64_t R, float radius, float* dirn, float* near_far, nm_stream_t stream_) {
    if (R < 0 || (R > 0 && (!rays_o || !rays_d || !dirn || !near_far))) return nm_fail("nm_rays_setup: bad arguments");
    if (R == 0) return 0;
    hipLaunchKernelGGL(nm_rays_setup_kernel, dim3(nm_blocks(R, 64)), dim3(64), 0, (hipStream_t)stream_, rays_o, rays_d, (long long)R, radius, dirn, near_far);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_rays_points(const float* rays_o, const float* dirn, int64_t R, int P, int mode, const float* near_far, const float* depth,
                   int cap, int off, float* depth_out, float* xyz, nm_stream_t stream_) {
    if (R < 0 || P < 1 || (mode != 1 && mode != 2)) return nm_fail("nm_rays_points: bad arguments");
    if (R == 0) return 0;
    if (!rays_o || !dirn || !xyz || (mode == 2 && !near_far) || (mode == 1 && !depth)) return nm_fail("nm_rays_points: NULL argument");
    NmPointSrc src;
    memset(&src, 0, sizeof(src));
    src.mode = mode;
    src.P = P;
    src.rays_o = rays_o;
    src.dirn = dirn;
    src.depth = depth;
    src.nearfar = near_far;
    src.depth_out = depth_out;
    src.dstride = cap;
    src.doff = off;
    hipLaunchKernelGGL(nm_rays_points_kernel, dim3(nm_blocks(R * P, 256)), dim3(256), 0, (hipStream_t)stream_, src, (long long)R * P, xyz);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_rays_bounds(const float* ds_probe, int64_t R, int G, float thresh, const float* near_far_in, float* near_far_out, nm_stream_t stream_) {
    if (R < 0 || G < 2 || (R > 0 && (!ds_probe || !near_far_in || !near_far_out))) return nm_fail("nm_rays_bounds: bad arguments");
    if (R == 0) return 0;
    hipLaunchKernelGGL(nm_rays_bounds_kernel, dim3(nm_blocks(R, 64)), dim3(64), 0, (hipStream_t)stream_, ds_probe, (long long)R, G, thresh, near_far_in, near_far_out);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_rays_upsample(float* d, float* sdf, int64_t R, int cap, int n, int m, int it, int n_new, const float* u, nm_stream_t stream_) {
    if (R < 0 || n < 2 || m < 0 || m > n || n + n_new > cap || cap > NM_MAX_SAMPLES || it < 0 || it > 20 || (R > 0 && (!d || !sdf))) return nm_fail("nm_rays_upsample: bad arguments");
    if (R == 0) return 0;
    size_t ray_lds = 0;
    if (nm_ray_lds_prepare(cap, &ray_lds)) return 1;
    hipLaunchKernelGGL(nm_rays_upsample_kernel, dim3(nm_blocks(R, 64)), dim3(NM_RAY_IO_THREADS), ray_lds, (hipStream_t)stream_, d, sdf, (int*)nullptr, (const float*)nullptr, (float*)nullptr, (long long)R, cap, n, m, it, n_new, u);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_rays_finalize(float* d, float* sdf, int64_t R, int cap, int n, int m, float* d_mid, nm_stream_t stream_) {
    if (R < 0 || n < 2 || n > cap || m < 0 || m > n || (R > 0 && (!d || !sdf || !d_mid))) return nm_fail("nm_rays_finalize: bad arguments");
    if (R == 0) return 0;
    size_t ray_lds = 0;
    if (nm_ray_lds_prepare(cap, &ray_lds)) return 1;
    hipLaunchKernelGGL(nm_rays_finalize_kernel, dim3(nm_blocks(R, 64)), dim3(NM_RAY_IO_THREADS), ray_lds, (hipStream_t)stream_, d, sdf, (int*)nullptr, (const float*)nullptr, (long long)R, cap, n, m, d_mid, (float*)nullptr, 0.f, (float*)nullptr, 0.f);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_rays_composite(const float* sdf, const float* d, int64_t R, int cap, int N, float s, const float* rgb_mid, const float* nablas,
                      int white_bkgd, float* rgb, float* depth, float* acc, float* normals, nm_stream_t stream_) {
    if (R < 0 || N < 2 || N > cap || N > NM_MAX_SAMPLES || (R > 0 && (!sdf || !d || !rgb_mid || !rgb || !depth || !acc))) return nm_fail("nm_rays_composite: bad arguments");
    if (R == 0) return 0;
    size_t ray_lds = 0;
    if (nm_ray_lds_prepare(cap, &ray_lds)) return 1;
    hipLaunchKernelGGL(nm_rays_composite_kernel, dim3(nm_blocks(R, 64)), dim3(NM_RAY_IO_THREADS), ray_lds, (hipStream_t)stream_, sdf, d, (long long)R, cap, N, s, rgb_mid, nablas, white_bkgd, rgb, depth, acc, normals, (const float*)nullptr, (const int*)nullptr);
    NM_LAUNCH_CHECK();
    return 0;
}

// ============================================================================== ray set-up
int nm_assemble_frame(const float* rgb, const float* depth, const float* normals, int64_t count, int bgr, uint8_t* rgb8,
                      uint8_t* depth8, uint8_t* normal8, float* depth_max_scratch, nm_stream_t stream_) {
    if (count < 0) return nm_fail("nm_assemble_frame: count < 0");
    if (count == 0) return 0;
    if ((rgb8 && !rgb) || (depth8 && !depth) || (normal8 && !normals)) return nm_fail("nm_assemble_frame: output without its input");
    if (depth8 && !depth_max_scratch) return nm_fail("nm_assemble_frame: depth8 needs depth_max_scratch");
    hipStream_t stream = (hipStream_t)stream_;
    if (depth8) {
        NM_HIP(hipMemsetAsync(depth_max_scratch, 0, sizeof(float), stream));
        const unsigned blocks = (unsigned)std::min<long long>(nm_blocks(count, 256), 2048);
        hipLaunchKernelGGL(nm_depth_max_kernel, dim3(blocks), dim3(256), 0, stream, depth, (long long)count, reinterpret_cast<unsigned*>(depth_max_scratch));
    }
    hipLaunchKernelGGL(nm_assemble_kernel, dim3(nm_blocks(count, 256)), dim3(256), 0, stream, rgb, depth, normals, (long long)count, bgr,
                       rgb8, depth8, normal8, (const float*)depth_max_scratch);
    NM_LAUNCH_CHECK();
    return 0;
}

static int nm_camera_convert(const nm_camera* cam, NmCamera& c, const char* who) {
    if (!cam) return nm_fail("%s: cam is NULL", who);
    if (cam->H < 1 || cam->W < 1) return nm_fail("%s: image size %dx%d", who, cam->H, cam->W);
    for (int i = 0; i < 12; ++i) c.r[i] = cam->c2w[i];
    c.fx = cam->fx; c.fy = cam->fy; c.cx = cam->cx; c.cy = cam->cy; c.sk = cam->sk;
    c.H = cam->H; c.W = cam->W;
    return 0;
}

int nm_make_rays(const nm_camera* cam, int64_t first_pixel, int64_t count, float* rays_o, float* rays_d, nm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NmCamera c;
    if (int rc = nm_camera_convert(cam, c, "nm_make_rays")) return rc;
    if (first_pixel < 0 || count < 0 || first_pixel + count > (int64_t)cam->H * cam->W)
        return nm_fail("nm_make_rays: pixel range [%lld,+%lld) outside %dx%d", (long long)first_pixel, (long long)count, cam->H, cam->W);
    if (count == 0) return 0;
    if (!rays_o || !rays_d) return nm_fail("nm_make_rays: NULL output");
    hipLaunchKernelGGL(nm_make_rays_kernel, dim3(nm_blocks(count, 256)), dim3(256), 0, stream, c, (long long)first_pixel, (long long)count, rays_o, rays_d);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_make_rays_indexed(const nm_camera* cam, const int64_t* pixels, int64_t count, float* rays_o, float* rays_d, nm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NmCamera c;
    if (int rc = nm_camera_convert(cam, c, "nm_make_rays_indexed")) return rc;
    if (count < 0) return nm_fail("nm_make_rays_indexed: count %lld", (long long)count);
    if (count == 0) return 0;
    if (!pixels || !rays_o || !rays_d) return nm_fail("nm_make_rays_indexed: NULL pointer");
    hipLaunchKernelGGL(nm_make_rays_indexed_kernel, dim3(nm_blocks(count, 256)), dim3(256), 0, stream, c, (const long long*)pixels, (long long)count, rays_o, rays_d);
    NM_LAUNCH_CHECK();
    return 0;
}

// ======================================================================== instrumentation
int nm_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.on.store(false);
    for (auto& r : g_prof.recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.recs.clear();
    if (g_prof.counters) { (void)hipFree(g_prof.counters); g_prof.counters = nullptr; }
    if (on) {  // counters live on the CURRENT device: profile one device at a time
        NM_HIP(hipMalloc((void**)&g_prof.counters, NM_CNT_N * sizeof(unsigned long long)));
        NM_HIP(hipMemset(g_prof.counters, 0, NM_CNT_N * sizeof(unsigned long long)));
        g_prof.on.store(true);
    }
    return 0;
}

// Shader clock right now: one wave counts its own clock (s_memtime) against the constant 100 MHz counter (s_memrealtime) for `micros`
// microseconds.  Launched on a stream of its own beside a running workload it reads the clock the chip holds UNDER THAT LOAD -- the
// figure a roofline fraction priced at the nominal 2.4 GHz needs beside it.  Synchronises `stream`.
__global__ void nm_clock_probe_kernel(unsigned long long* out, int ticks100) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
    unsigned long long r1 = r0;
    while ((long long)(r1 - r0) < (long long)ticks100) {
        __builtin_amdgcn_s_sleep(16);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    out[0] = c1 - c0;
    out[1] = r1 - r0;
}
int nm_profile_clock(int micros, float* mhz, nm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!mhz || micros < 1 || micros > 100000) return nm_fail("nm_profile_clock: bad arguments");
    unsigned long long* d = nullptr;
    unsigned long long h[2] = {0, 0};
    NM_HIP(hipMalloc((void**)&d, sizeof(h)));
    hipLaunchKernelGGL(nm_clock_probe_kernel, dim3(1), dim3(64), 0, stream, d, micros * 100);
    hipError_t e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d);
    if (e != hipSuccess) return nm_fail("nm_profile_clock: %s", hipGetErrorString(e));
    *mhz = h[1] ? (float)((double)h[0] / ((double)h[1] / 100.0)) : 0.f;   // shader ticks per microsecond
    return 0;
}

int nm_profile_read(int kind, double* total_ms, int64_t* launches, int64_t* units) {
    if (kind < 0 || kind >= NM_K_KINDS || !total_ms || !launches || !units) return nm_fail("nm_profile_read: bad arguments");
    std::lock_guard<std::mutex> lk(g_prof.mu);
    double ms = 0.0;
    int64_t n = 0, u = 0;
    int used[NM_CNT_N] = {0, 0};   // launches of this kind that processed "counter i" points
    for (auto& r : g_prof.recs) {
        if (r.kind != kind) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) return nm_fail("nm_profile_read: event sync failed");
        float e = 0.f;
        if (hipEventElapsedTime(&e, r.a, r.b) != hipSuccess) return nm_fail("nm_profile_read: elapsed failed");
        ms += e;
        n += 1;
        u += r.units;
        if (r.counter >= 0 && r.counter < NM_CNT_N) used[r.counter] += 1;
    }
    if (g_prof.counters && (used[0] || used[1])) {  // every such launch processed the counted points once per frame
        unsigned long long c[NM_CNT_N] = {0, 0};
        NM_HIP(hipMemcpy(c, g_prof.counters, sizeof(c), hipMemcpyDeviceToHost));
        // The device counters accumulate over all calls since nm_profile_enable (one increment per call: the mid-point
        // list / the probe walk); a kind with k counted launches per call therefore processed k * counter points.  The
        // number of calls = the colour kernel's counted launches (exactly one per call).
        int calls = 0;
        for (auto& r : g_prof.recs)
            if (r.kind == NM_K_COLOR && r.counter == NM_CNT_MID) ++calls;
        for (int i = 0; i < NM_CNT_N; ++i)
            if (used[i]) {
                const int k = (i == NM_CNT_MID && calls > 0) ? (used[i] + calls - 1) / calls : 1;
                u += (int64_t)c[i] * k;
            }
    }
    *total_ms = ms;
    *launches = n;
    *units = u;
    return 0;
}

#ifdef NM_TESTING
// test library only: install (or clear, with NULL) the device buffer the MLP kernels write phase timestamps to
int nm_debug_wave_log(void* device_buf_i64) {   // [1 + 3 * 2^20] int64 (or NULL): per-wave start / end / wave index of the distance kernels
    long long* p = (long long*)device_buf_i64;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_nm_wave_log), &p, sizeof(p)) != hipSuccess) return nm_fail("nm_debug_wave_log: hipMemcpyToSymbol failed");
    return 0;
}
int nm_debug_phase_log(void* device_buf_32x16_i64) {
    long long* p = (long long*)device_buf_32x16_i64;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_nm_phase_log), &p, sizeof(p)) != hipSuccess) return nm_fail("nm_debug_phase_log: hipMemcpyToSymbol failed");
    return 0;
}
#endif

int nm_time_kernel(nm_field_t f, nm_grid_t g, const nm_field_tables* t, int which, const float* xyz, const float* view_dirs,
                   int64_t P, void* scratch, int iters, float* avg_ms, nm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (nm_check_field_args(f, g, t, "nm_time_kernel")) return 1;
    if (!xyz || !scratch || !avg_ms || P < 1 || iters < 1) return nm_fail("nm_time_kernel: bad arguments");
    if (which == 3 && !view_dirs) return nm_fail("nm_time_kernel: view_dirs required for the colour kernel");
    const NmScratch s = nm_carve(scratch, P);
    const NmGather ga = {t->geometry_features, f->geo.gdim, s.fg, t->color_features, f->col.cdim, s.ft};
    // inputs of the MLP kernels come from one K-NN pass
    if (nm_launch_distance(g, nm_src_xyz(xyz, P), P, t->indicator_vector, t->indicator_weight, s.ds, nullptr, nullptr, nullptr, s.grad, stream, nullptr, ga)) return 1;
    if (which == 3 && nm_launch_geo(f, s.fg, s.ds, s.grad, P, true, nullptr, 1, 1, 0, s.nabla, stream)) return 1;
    hipEvent_t e0, e1;
    NM_HIP(hipEventCreate(&e0));
    NM_HIP(hipEventCreate(&e1));
    int rc = 0;
    // the sdf/rgb outputs of the timed kernels go to the (otherwise unused here) nabla/grad slots
    float* sink = s.nabla;
    for (int i = -1; i < iters && !rc; ++i) {  // i = -1: untimed warm-up
        if (i == 0) hipEventRecord(e0, stream);
        switch (which) {
            case 0: rc = nm_launch_distance(g, nm_src_xyz(xyz, P), P, t->indicator_vector, t->indicator_weight, s.ds, nullptr, nullptr, nullptr, s.grad, stream, nullptr, ga); break;
            case 1: rc = nm_launch_geo(f, s.fg, s.ds, s.grad, P, false, sink, 1, 1, 0, nullptr, stream); break;
            case 2: rc = nm_launch_geo(f, s.fg, s.ds, s.grad, P, true, nullptr, 1, 1, 0, sink, stream); break;
            case 3: rc = nm_launch_col(f, s.ft, s.ds, s.nabla, view_dirs, 1, P, s.grad, stream); break;
            default: rc = nm_fail("nm_time_kernel: which=%d", which);
        }
    }
    if (!rc) {
        hipEventRecord(e1, stream);
        if (hipEventSynchronize(e1) != hipSuccess) rc = nm_fail("nm_time_kernel: event sync failed");
        float ms = 0.f;
        if (!rc && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = nm_fail("nm_time_kernel: elapsed failed");
        *avg_ms = ms / (float)iters;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return rc;
}

#ifdef NM_TESTING
// Device self-check hook (test library only): the geometry / colour MLP computed with the scalar-ALU
// reference layer instead of the MFMA tile code, same inputs, same outputs.  valu_tmp: device
// buffer of ceil(P/32) * 64*256 floats.
int nm_selfcheck_field(nm_field_t f, nm_grid_t g, const nm_field_tables* t, const float* xyz, const float* view_dirs,
                       int64_t P, float* sdf, float* nabla, float* rgb, void* scratch, float* valu_tmp, nm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (nm_check_field_args(f, g, t, "nm_selfcheck_field")) return 1;
    if (!xyz || !view_dirs || !sdf || !nabla || !rgb || !scratch || !valu_tmp || P < 1) return nm_fail("nm_selfcheck_field: bad arguments");
    const NmScratch s = nm_carve(scratch, P);
    const NmGather ga = {t->geometry_features, f->geo.gdim, s.fg, t->color_features, f->col.cdim, s.ft};
    if (nm_launch_distance(g, nm_src_xyz(xyz, P), P, t->indicator_vector, t->indicator_weight, s.ds, nullptr, nullptr, nullptr, s.grad, stream, nullptr, ga)) return 1;
    hipLaunchKernelGGL((nm_geo_mlp_kernel<true, true>), dim3(nm_blocks(P, 32)), dim3(256), 0, stream, f->geo, s.fg, s.ds, s.grad,
                       NM_COMPACT, (long long)P, sdf, 1, 1, 0, nabla, valu_tmp, 0, NM_NO_SLOTS);
    NM_LAUNCH_CHECK();
    hipLaunchKernelGGL((nm_col_mlp_kernel<true>), dim3(nm_blocks(P, 64)), dim3(256), 0, stream, f->col, s.ft, s.ds, nabla,
                       view_dirs, 1, (long long)P, rgb, valu_tmp, NM_NO_SLOTS);
    NM_LAUNCH_CHECK();
    return 0;
}

// number of queries the last small launch on (g, stream) handed to the exhaustive kernels (synchronises the stream)
int nm_debug_last_deferred(nm_grid_t g, int* count, nm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!g || !count) return nm_fail("nm_debug_last_deferred: NULL argument");
    *count = -1;
    void* blk = nullptr;
    {
        std::lock_guard<std::mutex> lk(g->defer_mu);
        for (auto& e : g->defer_scratch)
            if (e.first == stream) blk = e.second;
    }
    if (!blk) return 0;
    NM_HIP(hipMemcpyAsync(count, blk, sizeof(int), hipMemcpyDeviceToHost, stream));
    NM_HIP(hipStreamSynchronize(stream));
    return 0;
}

// nm_simd_key() of `n` one-wave workgroups (each holds its SIMD for a few microseconds so that the launch spreads over the chip): the
// pull kernels count their resident waves per SIMD by this key, so a full launch must show 4 x CUs distinct values
__global__ __launch_bounds__(64) void nm_debug_simd_key_kernel(int* __restrict__ out) {
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < 2000) __builtin_amdgcn_s_sleep(8);   // 100 MHz counter: 20 us
    if (threadIdx.x == 0) out[blockIdx.x] = nm_simd_key();
}
int nm_debug_simd_keys(int* out_device, int n, nm_stream_t stream_) {
    if (!out_device || n < 1) return nm_fail("nm_debug_simd_keys: bad arguments");
    hipLaunchKernelGGL(nm_debug_simd_key_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream_, out_device);
    NM_LAUNCH_CHECK();
    return 0;
}

// The training path's GEMM alone (nm_gemm.h): C[M,N] = A . B (+ bias / relu as in NmGemm), operand layouts as there.  mode 0 = fp32 pipe,
// 1 = bf16 x 3.  iters > 0: launched that many times, *avg_ms = mean launch time (events on `stream`).
int nm_debug_gemm(const float* A, int64_t lda, int a_kc, const float* B, int64_t ldb, int b_kc, float* C, int64_t ldc, int64_t M, int64_t N,
                  int64_t K, const float* bias, int relu, int split_k, int accumulate, int mode, int iters, float* avg_ms, nm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NmGemm g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.lda = lda; g.a_kc = a_kc; g.B = B; g.ldb = ldb; g.b_kc = b_kc; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.bias_rows = bias ? M : 0; g.relu = relu; g.atomic = accumulate;
    hipEvent_t e0, e1;
    NM_HIP(hipEventCreate(&e0));
    NM_HIP(hipEventCreate(&e1));
    int rc = 0;
    for (int i = (iters > 1 ? -1 : 0); i < (iters > 0 ? iters : 1) && !rc; ++i) {
        if (i == 0) hipEventRecord(e0, stream);
        rc = nm_gemm_launch(g, split_k, stream, mode != 0);
    }
    hipEventRecord(e1, stream);
    if (!rc && avg_ms) {
        float ms = 0.f;
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = nm_fail("nm_debug_gemm: events failed");
        *avg_ms = ms / (float)(iters > 0 ? iters : 1);
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return rc ? nm_fail("nm_debug_gemm: launch failed") : 0;
}
#endif  // NM_TESTING

}  // extern "C"
