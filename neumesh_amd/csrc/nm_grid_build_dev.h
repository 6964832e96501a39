// nm_grid_build_dev.h -- DEVICE-side construction of the sparse-octree index (nm_grid.h), bit-identical to
// the host build of nm_grid_build.h (which stays as the reference implementation: tests/hostcheck compiles it
// with g++, and the GPU test compares the two node arrays byte for byte).
//
// Why: the editing tools re-build the index for every deformed mesh (reference: deform_model /
// update_mesh_grid, editing/render_geometry_editing.py:37-67 -> MeshGrid.__init__ -> FRNN grid build,
// models/mesh_grid.py:64-74); a host build costs a device->host copy of the vertices, ~100 ms of CPU sorting for
// 1.4e5 vertices and the upload.  Here all O(V) work runs on the GPU:
//   1. bounding box (block reduction + atomics on order-preserving keys)            -> 6 floats to the host
//   2. level-8 Morton codes, one keys-only radix sort, occupied cells per level     -> 7 counts to the host,
//      which picks the leaf level L exactly as the host build does (<= NM_LEAF_TARGET vertices per occupied leaf)
//   3. radix sort of (level-L code, vertex index) pairs (stable: ties by index), gather of the sorted vertices
//   4. per level, leaves first: segment heads by adjacent-difference + exclusive scan, tight boxes by one thread
//      per node over its vertices / children                                         -> 1 count per level to the host
//   5. one kernel per level writes the 64-byte node records (box slack, centre keys, ordered child masks)
// The scalars that cross to the host are a few words per call; nm_grid_create is documented as synchronising.
#pragma once

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "nm_grid_build.h"

struct NmDevRoot {
    float ox, oy, oz, root_size, slack;
};

__device__ __forceinline__ uint32_t nm_leaf_code_dev(const NmDevRoot g, int L, float x, float y, float z) {
    const int n = 1 << L;
    const float inv = (float)n / g.root_size;
    int ix = (int)floorf((x - g.ox) * inv), iy = (int)floorf((y - g.oy) * inv), iz = (int)floorf((z - g.oz) * inv);
    ix = min(max(ix, 0), n - 1);
    iy = min(max(iy, 0), n - 1);
    iz = min(max(iz, 0), n - 1);
    return nm_morton((uint32_t)ix, (uint32_t)iy, (uint32_t)iz);
}

// out[0..2] = keys of the minima, out[3..5] = keys of the maxima (initialised to ~0 / 0), out[6] = non-finite flag
__global__ void nm_bbox_kernel(const float* __restrict__ v, long long V, uint32_t* __restrict__ out) {
    __shared__ uint32_t s[7];
    if (threadIdx.x < 7) s[threadIdx.x] = threadIdx.x < 3 ? 0xffffffffu : 0u;
    __syncthreads();
    uint32_t lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u}, bad = 0u;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (long long)gridDim.x * blockDim.x)
        for (int a = 0; a < 3; ++a) {
            const float c = v[3 * i + a];
            if (!isfinite(c)) bad = 1u;
            const uint32_t k = nm_float_key(c);
            lo[a] = min(lo[a], k);
            hi[a] = max(hi[a], k);
        }
    for (int a = 0; a < 3; ++a) {
        atomicMin(&s[a], lo[a]);
        atomicMax(&s[3 + a], hi[a]);
    }
    if (bad) atomicOr(&s[6], 1u);
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&out[threadIdx.x], s[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&out[threadIdx.x], s[threadIdx.x]);
    else if (threadIdx.x == 6 && s[6]) atomicOr(&out[6], 1u);
}

__global__ void nm_codes_kernel(const float* __restrict__ v, long long V, NmDevRoot g, int L, uint32_t* __restrict__ code,
                                uint32_t* __restrict__ index) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    code[i] = nm_leaf_code_dev(g, L, v[3 * i], v[3 * i + 1], v[3 * i + 2]);
    if (index) index[i] = (uint32_t)i;
}

// occupied cells at levels 1..7 from the SORTED level-8 codes: counts[L] += #{p : p == 0 or prefix_L differs from p-1}
__global__ void nm_level_occupancy_kernel(const uint32_t* __restrict__ code8, long long V, unsigned* __restrict__ counts) {
    __shared__ unsigned s[8];
    if (threadIdx.x < 8) s[threadIdx.x] = 0u;
    __syncthreads();
    unsigned c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < V; p += (long long)gridDim.x * blockDim.x) {
        const uint32_t k = code8[p], kp = p ? code8[p - 1] : 0u;
        for (int L = 1; L <= 7; ++L) {
            const int sh = 3 * (8 - L);
            if (p == 0 || (k >> sh) != (kp >> sh)) ++c[L];
        }
    }
    for (int L = 1; L <= 7; ++L)
        if (c[L]) atomicAdd(&s[L], c[L]);
    __syncthreads();
    if (threadIdx.x >= 1 && threadIdx.x <= 7 && s[threadIdx.x]) atomicAdd(&counts[threadIdx.x], s[threadIdx.x]);
}

__global__ void nm_sverts_kernel(const float* __restrict__ v, const uint32_t* __restrict__ order, long long V, float4* __restrict__ sverts) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= V + 4) return;
    if (p >= V) {
        sverts[p] = make_float4(NM_INF_F, NM_INF_F, NM_INF_F, nm_as_float(0x7fffffff));
        return;
    }
    const uint32_t i = order[p];
    sverts[p] = make_float4(v[3 * (size_t)i], v[3 * (size_t)i + 1], v[3 * (size_t)i + 2], nm_as_float((int)i));
}

// head flags of the segments of equal (code >> shift) in a sorted code array
__global__ void nm_head_flags_kernel(const uint32_t* __restrict__ code, long long n, int shift, unsigned* __restrict__ flag) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    flag[p] = (p == 0 || (code[p] >> shift) != (code[p - 1] >> shift)) ? 1u : 0u;
}

// per segment head: node code and first element; per element: its segment (= parent) index.  scan = exclusive scan of flag.
__global__ void nm_heads_scatter_kernel(const uint32_t* __restrict__ code, const unsigned* __restrict__ flag, const unsigned* __restrict__ scan,
                                        long long n, int shift, uint32_t* __restrict__ node_code, uint32_t* __restrict__ node_first,
                                        uint32_t* __restrict__ elem_seg, unsigned* __restrict__ count) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const unsigned id = scan[p] + flag[p] - 1u;  // inclusive - 1 = index of the segment p belongs to
    if (elem_seg) elem_seg[p] = id;
    if (flag[p]) {
        node_code[id] = code[p] >> shift;
        node_first[id] = (uint32_t)p;
    }
    if (p == n - 1) *count = id + 1u;
}

// tight boxes: leaves over their vertices ...
__global__ void nm_leaf_boxes_kernel(const float4* __restrict__ sverts, const uint32_t* __restrict__ first, unsigned n, long long V,
                                     float* __restrict__ lo, float* __restrict__ hi) {
    const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t a = first[j], b = (j + 1 < n) ? first[j + 1] : (uint32_t)V;
    float l[3] = {NM_INF_F, NM_INF_F, NM_INF_F}, h[3] = {-NM_INF_F, -NM_INF_F, -NM_INF_F};
    for (uint32_t p = a; p < b; ++p) {
        const float4 v = sverts[p];
        l[0] = fminf(l[0], v.x); l[1] = fminf(l[1], v.y); l[2] = fminf(l[2], v.z);
        h[0] = fmaxf(h[0], v.x); h[1] = fmaxf(h[1], v.y); h[2] = fmaxf(h[2], v.z);
    }
    for (int c = 0; c < 3; ++c) { lo[3 * j + c] = l[c]; hi[3 * j + c] = h[c]; }
}
// ... internal nodes over their children's boxes
__global__ void nm_inner_boxes_kernel(const float* __restrict__ clo, const float* __restrict__ chi, const uint32_t* __restrict__ first,
                                      unsigned n, unsigned n_child, float* __restrict__ lo, float* __restrict__ hi) {
    const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t a = first[j], b = (j + 1 < n) ? first[j + 1] : n_child;
    float l[3] = {NM_INF_F, NM_INF_F, NM_INF_F}, h[3] = {-NM_INF_F, -NM_INF_F, -NM_INF_F};
    for (uint32_t c = a; c < b; ++c)
        for (int k = 0; k < 3; ++k) {
            l[k] = fminf(l[k], clo[3 * c + k]);
            h[k] = fmaxf(h[k], chi[3 * c + k]);
        }
    for (int k = 0; k < 3; ++k) { lo[3 * j + k] = l[k]; hi[3 * j + k] = h[k]; }
}

// the node records of one level (same expressions as the record loop of nm_build_host_grid)
__global__ void nm_node_records_kernel(const uint32_t* __restrict__ code, const uint32_t* __restrict__ first, const float* __restrict__ lo,
                                       const float* __restrict__ hi, const uint32_t* __restrict__ parent_local,
                                       const uint32_t* __restrict__ child_code, unsigned n, unsigned n_child, int is_leaf, int is_root_level,
                                       uint32_t off_this, uint32_t off_parent, uint32_t off_child, long long V, float slack,
                                       NmNode* __restrict__ nodes) {
    const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    NmNode r;
    uint32_t mask = 0;
    if (is_leaf) {
        r.first = first[j];
        r.end = (j + 1 < n) ? first[j + 1] : (uint32_t)V;
    } else {
        r.first = off_child + first[j];
        r.end = 0;
        const uint32_t b = (j + 1 < n) ? first[j + 1] : n_child;
        for (uint32_t c = first[j]; c < b; ++c) mask |= 1u << (child_code[c] & 7u);
    }
    r.parent = is_root_level ? 0u : off_parent + parent_local[j];
    r.info = mask | ((code[j] & 7u) << 8);
    float blo[3], bhi[3];
    for (int a = 0; a < 3; ++a) {
        const float e = slack + 1e-6f * fmaxf(fabsf(lo[3 * j + a]), fabsf(hi[3 * j + a]));
        blo[a] = lo[3 * j + a] - e;
        bhi[a] = hi[3 * j + a] + e;
    }
    r.lox = blo[0]; r.loy = blo[1]; r.loz = blo[2];
    r.hix = bhi[0]; r.hiy = bhi[1]; r.hiz = bhi[2];
    r.ckx = nm_float_key(0.5f * (blo[0] + bhi[0]));
    r.cky = nm_float_key(0.5f * (blo[1] + bhi[1]));
    r.ckz = nm_float_key(0.5f * (blo[2] + bhi[2]));
    r.om_lo = r.om_hi = 0;
    for (int fo = 0; fo < 8; ++fo) {
        const uint32_t om = nm_ordered_mask(mask, fo);
        if (fo < 4) r.om_lo |= om << (8 * fo);
        else r.om_hi |= om << (8 * (fo - 4));
    }
    r.pad = 0;
    nodes[(size_t)off_this + j] = r;
}

// ---------------------------------------------------------------------------------------------- host driver
struct NmDevGrid {   // result: device arrays owned by the caller afterwards
    NmNode* nodes = nullptr;
    float4* sverts = nullptr;
    size_t n_nodes = 0;
    int L = 0, occupied_leaves = 0;
    NmDevRoot root{};
};

#define NM_DG(call)                                  \
    do {                                             \
        hipError_t e_ = (call);                      \
        if (e_ != hipSuccess) { err = e_; goto fail; } \
    } while (0)

// verts: device [V,3].  On success fills `out` (two hipMalloc'ed arrays) and returns hipSuccess; *bad_input is set when the
// vertices hold NaN / Inf (nothing allocated then).
static hipError_t nm_build_device_grid(const float* verts, long long V, int leaf_level, hipStream_t stream, NmDevGrid& out, bool* bad_input) {
    hipError_t err = hipSuccess;
    *bad_input = false;
    const unsigned T = 256;
    auto blocks = [&](long long n) { return dim3((unsigned)((n + T - 1) / T)); };
    // one scratch allocation: codes / order (double-buffered for the sorts), flags, scans, per-level arrays
    const size_t nV = (size_t)V;
    uint32_t *code_a = nullptr, *code_b = nullptr, *idx_a = nullptr, *idx_b = nullptr;
    unsigned *flag = nullptr, *scan = nullptr, *small = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    uint32_t* lvl_code[NM_MAX_LEVEL + 1] = {nullptr};
    uint32_t* lvl_first[NM_MAX_LEVEL + 1] = {nullptr};
    uint32_t* lvl_parent[NM_MAX_LEVEL + 1] = {nullptr};
    float* lvl_lo[NM_MAX_LEVEL + 1] = {nullptr};
    float* lvl_hi[NM_MAX_LEVEL + 1] = {nullptr};
    unsigned lvl_n[NM_MAX_LEVEL + 2] = {0};
    uint32_t off[NM_MAX_LEVEL + 2] = {0};
    uint32_t hbox[7];
    unsigned hcounts[8];
    int L = leaf_level;
    float lo[3], hi[3], ext, amax = 0.f;
    NmDevRoot g;
    {
        size_t a = 0, b = 0, c = 0;
        (void)rocprim::radix_sort_keys(nullptr, a, (uint32_t*)nullptr, (uint32_t*)nullptr, nV, 0, 24, stream);
        (void)rocprim::radix_sort_pairs(nullptr, b, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, nV, 0, 24, stream);
        (void)rocprim::exclusive_scan(nullptr, c, (unsigned*)nullptr, (unsigned*)nullptr, 0u, nV, rocprim::plus<unsigned>(), stream);
        tmp_bytes = a > b ? (a > c ? a : c) : (b > c ? b : c);
    }
    NM_DG(hipMalloc((void**)&code_a, nV * 4));
    NM_DG(hipMalloc((void**)&code_b, nV * 4));
    NM_DG(hipMalloc((void**)&idx_a, nV * 4));
    NM_DG(hipMalloc((void**)&idx_b, nV * 4));
    NM_DG(hipMalloc((void**)&flag, nV * 4));
    NM_DG(hipMalloc((void**)&scan, nV * 4));
    NM_DG(hipMalloc((void**)&small, 64 * 4));
    NM_DG(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    // ---- 1. bounding box
    {
        const uint32_t init[7] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u};
        NM_DG(hipMemcpyAsync(small, init, sizeof(init), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(nm_bbox_kernel, dim3((unsigned)std::min<long long>((V + T - 1) / T, 1024)), dim3(T), 0, stream, verts, V, (uint32_t*)small);
        NM_DG(hipMemcpyAsync(hbox, small, sizeof(hbox), hipMemcpyDeviceToHost, stream));
        NM_DG(hipStreamSynchronize(stream));
        if (hbox[6]) { *bad_input = true; err = hipSuccess; goto fail; }
        auto unkey = [](uint32_t k) { const uint32_t u = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k; return nm_as_float((int)u); };
        for (int a = 0; a < 3; ++a) { lo[a] = unkey(hbox[a]); hi[a] = unkey(hbox[3 + a]); }
        // the very expressions of nm_build_host_grid
        ext = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
        for (int a = 0; a < 3; ++a) amax = std::max(amax, std::max(std::fabs(lo[a]), std::fabs(hi[a])));
        if (!(ext > 0.f)) ext = std::max(1e-3f, 1e-3f * amax);
        g.root_size = ext * 1.001f + 1e-6f * std::max(amax, 1.0f);
        g.ox = 0.5f * (lo[0] + hi[0]) - 0.5f * g.root_size;
        g.oy = 0.5f * (lo[1] + hi[1]) - 0.5f * g.root_size;
        g.oz = 0.5f * (lo[2] + hi[2]) - 0.5f * g.root_size;
        g.slack = 4e-6f * (amax + g.root_size);
    }
    // ---- 2. leaf level: smallest depth with <= NM_LEAF_TARGET vertices per occupied leaf on average (as the host build)
    if (L <= 0) {
        hipLaunchKernelGGL(nm_codes_kernel, blocks(V), dim3(T), 0, stream, verts, V, g, NM_MAX_LEVEL, code_a, (uint32_t*)nullptr);
        size_t tb = tmp_bytes;
        NM_DG(rocprim::radix_sort_keys(tmp, tb, code_a, code_b, nV, 0, 24, stream));
        NM_DG(hipMemsetAsync(small, 0, 8 * 4, stream));
        hipLaunchKernelGGL(nm_level_occupancy_kernel, dim3((unsigned)std::min<long long>((V + T - 1) / T, 1024)), dim3(T), 0, stream, code_b, V, small);
        NM_DG(hipMemcpyAsync(hcounts, small, sizeof(hcounts), hipMemcpyDeviceToHost, stream));
        NM_DG(hipStreamSynchronize(stream));
        for (L = 1; L < NM_MAX_LEVEL; ++L)
            if ((double)V / (double)hcounts[L] <= NM_LEAF_TARGET) break;
    }
    L = std::min(std::max(L, 1), NM_MAX_LEVEL);
    // ---- 3. vertices in (leaf code, index) order
    {
        hipLaunchKernelGGL(nm_codes_kernel, blocks(V), dim3(T), 0, stream, verts, V, g, L, code_a, idx_a);
        size_t tb = tmp_bytes;
        NM_DG(rocprim::radix_sort_pairs(tmp, tb, code_a, code_b, idx_a, idx_b, nV, 0, 3 * L, stream));   // stable: ties keep index order
        NM_DG(hipMalloc((void**)&out.sverts, (nV + 4) * sizeof(float4)));
        hipLaunchKernelGGL(nm_sverts_kernel, blocks(V + 4), dim3(T), 0, stream, verts, idx_b, V, out.sverts);
    }
    // ---- 4. levels, leaves first.  Level l's codes are (sorted leaf codes) >> 3*(L-l), segment heads give its nodes.
    {
        const uint32_t* elem_code = code_b;   // sorted codes of the elements being grouped (vertices, then child nodes)
        long long n_elem = V;
        for (int l = L; l >= 0; --l) {
            const int shift = (l == L) ? 0 : 3;
            hipLaunchKernelGGL(nm_head_flags_kernel, blocks(n_elem), dim3(T), 0, stream, elem_code, n_elem, shift, flag);
            size_t tb = tmp_bytes;
            NM_DG(rocprim::exclusive_scan(tmp, tb, flag, scan, 0u, (size_t)n_elem, rocprim::plus<unsigned>(), stream));
            // (upper bound of this level's node count: the element count)
            NM_DG(hipMalloc((void**)&lvl_code[l], (size_t)n_elem * 4));
            NM_DG(hipMalloc((void**)&lvl_first[l], (size_t)n_elem * 4));
            if (l < L) NM_DG(hipMalloc((void**)&lvl_parent[l + 1], (size_t)n_elem * 4));   // parent (local) of every child node
            hipLaunchKernelGGL(nm_heads_scatter_kernel, blocks(n_elem), dim3(T), 0, stream, elem_code, flag, scan, n_elem, shift, lvl_code[l],
                               lvl_first[l], l < L ? lvl_parent[l + 1] : (uint32_t*)nullptr, small);
            NM_DG(hipMemcpyAsync(&lvl_n[l], small, 4, hipMemcpyDeviceToHost, stream));
            NM_DG(hipStreamSynchronize(stream));
            const unsigned n = lvl_n[l];
            NM_DG(hipMalloc((void**)&lvl_lo[l], (size_t)n * 12));
            NM_DG(hipMalloc((void**)&lvl_hi[l], (size_t)n * 12));
            if (l == L) hipLaunchKernelGGL(nm_leaf_boxes_kernel, blocks(n), dim3(T), 0, stream, out.sverts, lvl_first[l], n, V, lvl_lo[l], lvl_hi[l]);
            else hipLaunchKernelGGL(nm_inner_boxes_kernel, blocks(n), dim3(T), 0, stream, lvl_lo[l + 1], lvl_hi[l + 1], lvl_first[l], n, lvl_n[l + 1], lvl_lo[l], lvl_hi[l]);
            elem_code = lvl_code[l];
            n_elem = n;
        }
    }
    // ---- 5. records
    for (int l = 0; l <= L; ++l) off[l + 1] = off[l] + lvl_n[l];
    out.n_nodes = off[L + 1];
    NM_DG(hipMalloc((void**)&out.nodes, out.n_nodes * sizeof(NmNode)));
    for (int l = 0; l <= L; ++l)
        hipLaunchKernelGGL(nm_node_records_kernel, blocks(lvl_n[l]), dim3(T), 0, stream, lvl_code[l], lvl_first[l], lvl_lo[l], lvl_hi[l],
                           l > 0 ? lvl_parent[l] : (const uint32_t*)nullptr, l < L ? lvl_code[l + 1] : (const uint32_t*)nullptr, lvl_n[l],
                           l < L ? lvl_n[l + 1] : 0u, l == L ? 1 : 0, l == 0 ? 1 : 0, off[l], l > 0 ? off[l - 1] : 0u, off[l + 1], V, g.slack, out.nodes);
    err = hipGetLastError();
    if (err != hipSuccess) goto fail;
    NM_DG(hipStreamSynchronize(stream));
    out.L = L;
    out.occupied_leaves = (int)lvl_n[L];
    out.root = g;
fail:
    for (int l = 0; l <= NM_MAX_LEVEL; ++l) {
        if (lvl_code[l]) hipFree(lvl_code[l]);
        if (lvl_first[l]) hipFree(lvl_first[l]);
        if (lvl_parent[l]) hipFree(lvl_parent[l]);
        if (lvl_lo[l]) hipFree(lvl_lo[l]);
        if (lvl_hi[l]) hipFree(lvl_hi[l]);
    }
    if (code_a) hipFree(code_a);
    if (code_b) hipFree(code_b);
    if (idx_a) hipFree(idx_a);
    if (idx_b) hipFree(idx_b);
    if (flag) hipFree(flag);
    if (scan) hipFree(scan);
    if (small) hipFree(small);
    if (tmp) hipFree(tmp);
    if (err != hipSuccess || *bad_input) {
        if (out.nodes) hipFree(out.nodes);
        if (out.sverts) hipFree(out.sverts);
        out.nodes = nullptr;
        out.sverts = nullptr;
    }
    return err;
}
#undef NM_DG
