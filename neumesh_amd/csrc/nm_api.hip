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

#ifndef NM_PROBE_STEP
#define NM_PROBE_STEP 8  // probes per ray and step of nm_probe_bounds_kernel: 8 = 8 rays per wave (measured: K-NN per frame 99.9 ms with 4, 97.2 with 8, 101.7 with 16)
#endif

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

#include "nm_api_grid.inc"
#include "nm_api_field.inc"
#include "nm_api_train.inc"
#include "nm_api_render.inc"
#include "nm_api_surface.inc"
#include "nm_api_stages.inc"
#include "nm_api_instrumentation.inc"

}  // extern "C"
