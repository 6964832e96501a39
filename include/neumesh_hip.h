/*
 * neumesh_hip.h -- C ABI of libneumesh_hip.so (gfx950 / MI355X), the drop-in boundary of the
 * NeuMesh volumetric-render hot path.
 *
 * Conventions
 *   - every pointer marked "device" is a HIP device pointer owned by the CALLER (PyTorch
 *     allocates); the library borrows it for the duration of the call's stream work and never
 *     frees it.  The library owns only what its handles hold (nm_grid_t, nm_field_t).
 *   - every entry point that launches work takes an explicit stream (hipStream_t passed as
 *     void*), enqueues asynchronously and does NOT synchronise (exception: nm_grid_create and
 *     nm_field_create/update, one-off setup calls, which synchronise the stream).
 *   - the library reads no environment variables: every switch is an argument (nm_render_cfg.flags
 *     and the tuning fields next to it).  Calls on different handles / streams may come from
 *     different threads; the only process-wide state is the profiling log (nm_profile_*), which
 *     is mutex-protected and meant for one measuring thread.
 *   - return value: 0 = ok, non-zero = error; nm_last_error() gives a thread-local message.
 *   - all floating point is IEEE fp32; K-NN indices are int64 at this boundary because the
 *     reference indexes tensors with them (models/mesh_grid.py:126,134-136;
 *     editing/texture_neumesh/texture_neumesh.py:85).
 *
 * Reference interfaces replaced (file:line relative to the NeuMesh reference tree):
 *   nm_grid_create        frnn.frnn_grid_points(..., grid=None)  models/mesh_grid.py:64-74
 *   nm_knn                frnn.frnn_grid_points(..., grid=g)     models/mesh_grid.py:109-119
 *   nm_compute_distance   MeshGrid.compute_distance_frnn         models/mesh_grid.py:88-144
 *   nm_distance_interpolate  compute_distance_frnn + interpolation(features, indices, weights)
 *                                                 mesh_grid.py:88-144 + neumesh.py:11-13
 *   nm_field_density      NeuMesh.forward_density_only / forward_with_nablas
 *                                                 models/frameworks/neumesh/neumesh.py:140-154
 *   nm_field_forward      NeuMesh.forward                        neumesh.py:113-138
 *   nm_field_color        NeuMesh.forward_color                  neumesh.py:156-168
 *   nm_render_rays        volume_render -> render_rayschunk      models/renderer.py:105-368
 */
#ifndef NEUMESH_HIP_H
#define NEUMESH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NM_ABI_VERSION 11
#define NM_MAX_K 32

typedef struct nm_grid_s* nm_grid_t;    /* spatial index over the mesh vertices */
typedef struct nm_field_s* nm_field_t;  /* packed MLP weights + borrowed code tables */
typedef void* nm_stream_t;              /* hipStream_t */

int nm_abi_version(void);
const char* nm_last_error(void);
/* number of HIP devices visible to the library (0 => every compute entry point fails) */
int nm_device_count(void);

/* ----------------------------------------------------------------------------- spatial index
 * Builds the search structure over `V` vertices (device, [V,3] fp32, row-major).
 * leaf_level = 0 picks the octree depth automatically (the smallest depth with <= 40 vertices
 * per occupied leaf on average).  The vertex data is COPIED (sorted copy lives in the handle); `verts` may be
 * freed afterwards.  The build runs on the device (Morton codes, radix sort, per-level node
 * kernels; a deformed mesh is re-indexed in about a millisecond); a few scalars -- bounding box,
 * per-level node counts -- pass through the host, so the call synchronises `stream`. */
int nm_grid_create(const float* verts_device, int64_t V, int leaf_level, nm_stream_t stream,
                   nm_grid_t* out);
int nm_grid_destroy(nm_grid_t g);
/* Per-index options.
 *   NM_GRID_DEFER_BUDGET  point-wise launches of <= 2^18 queries: work units (24 per node test, 7 per scanned vertex) after which a
 *                         wave hands its unfinished queries to the whole chip (exact either way).  0 = never, -1 = the build's
 *                         default (30000).  The hand-over needs 33.7 MB of scratch per (index, stream) that used it, allocated with
 *                         hipMalloc on the first such launch of a stream (at most 8 streams per index; later ones do not defer).
 *   NM_GRID_TRIM          frees that scratch (value ignored).  The caller vouches that no launch on this index is in flight. */
#define NM_GRID_DEFER_BUDGET 1
#define NM_GRID_TRIM 2
int nm_grid_set_option(nm_grid_t g, int option, int64_t value);

typedef struct nm_grid_info {
    int64_t num_vertices;
    int32_t leaf_level;         /* octree depth L: 2^L cells per axis */
    int32_t occupied_leaves;
    float origin[3];            /* min corner of the root cube */
    float root_size;            /* edge of the root cube */
    int64_t device_bytes;       /* memory held by the handle */
    int64_t num_nodes;          /* octree node records (64 bytes each) */
} nm_grid_info;
int nm_grid_get_info(nm_grid_t g, nm_grid_info* out);

/* Exact K nearest vertices of each query (device [Q,3]) under the declared arithmetic
 * (fp32, dx=q-v, d2=(dx*dx+dy*dy)+dz*dz, no FMA, ascending (d2, index)).
 * idx: device [Q,K] int64, d2: device [Q,K] fp32 (squared distances, like FRNN).
 * If V < K the missing slots are idx=-1, d2=-1 (FRNN's padding). 1 <= K <= NM_MAX_K. */
int nm_knn(nm_grid_t g, const float* q_device, int64_t Q, int K, int64_t* idx_device,
           float* d2_device, nm_stream_t stream);

/* Fused K-NN + inverse-distance weights + indicator-blended projected signed distance.
 * indicator: device [V,3]; w1: indicator weight (0.1 or sigmoid(indicator_weight_raw)).
 * Outputs (each may be NULL to skip): ds [Q] fp32, idx [Q,K] int64, w [Q,K] fp32,
 * dds_dx [Q,3] fp32 = closed-form d ds / d xyz with idx/w held constant (they are detached in
 * the reference, mesh_grid.py:121-122).  K must be 8 when ds/dds_dx are requested. */
int nm_compute_distance(nm_grid_t g, const float* q_device, int64_t Q,
                        const float* indicator_device, float w1, int K, float* ds_device,
                        int64_t* idx_device, float* w_device, float* dds_dx_device,
                        nm_stream_t stream);

/* nm_compute_distance followed by interpolation(table, idx, w) = sum_k table[idx_k] * w_k
 * (models/frameworks/neumesh/neumesh.py:11-13) in the same kernel: the wave that found the
 * neighbours gathers their rows.  table: device [V,dim] fp32, dim a multiple of 4 (any width: the
 * 256-d stress table of BASELINE config 5 included); feat: device [Q,dim].  ds / idx / w may be NULL. */
int nm_distance_interpolate(nm_grid_t g, const float* q_device, int64_t Q,
                            const float* indicator_device, float w1, const float* table_device,
                            int dim, float* ds_device, int64_t* idx_device, float* w_device,
                            float* feat_device, nm_stream_t stream);

/* ----------------------------------------------------------------------------------- field
 * Dense layers are given as PyTorch stores nn.Linear: weight [out,in] row-major, bias [out]
 * (weight-norm already folded: W = g * v / ||v||_row).  The library re-packs them. */
typedef struct nm_field_desc {
    int32_t W;                 /* hidden width; must be 256 */
    int32_t D_density;         /* geometry MLP hidden layers (1..8) */
    int32_t D_color;           /* colour MLP hidden layers (1..8) */
    int32_t geometry_dim;      /* multiple of 4, <= 64 */
    int32_t color_dim;         /* multiple of 4, <= 64 */
    int32_t multires_d, multires_fg, multires_ft, multires_view; /* embedder bands, >= 0 */
    int32_t enable_nablas_input;
    int32_t use_view_dirs;     /* must be 1 */
    int32_t mlp_precision;     /* 0: fp32 MFMA (reference numerics); 2: split-half f16 MFMA -- every
                                  value carried as two fp16 halves (22 bits), 3 f16 MFMAs per product,
                                  fp32 accumulation; |activations| must stay < 65504 (nm_field_overflow
                                  reports a violation); 4: the same kernels with ONE f16 product per fp32
                                  product (11-bit operands: reduced precision, never a default);
                                  5: 2 for the geometry network, 4 for the colour network; 6: as 2 with the residual
                                  halves unscaled and ONE accumulator for the three products (absolute operand
                                  precision 2^-25); 7: 6 for the geometry network, 4 for the colour network */
    const float* geo_weight[8];   /* device; layer 0: [W, in_geo], others [W,W] */
    const float* geo_bias[8];     /* device [W] */
    const float* density_weight;  /* device [1,W] */
    const float* density_bias;    /* device [1] */
    const float* col_weight[8];   /* device; layer 0: [W, in_col], others [W,W] */
    const float* col_bias[8];
    const float* rgb_weight;      /* device [3,W] */
    const float* rgb_bias;        /* device [3] */
} nm_field_desc;

int nm_field_create(const nm_field_desc* desc, nm_stream_t stream, nm_field_t* out);
int nm_field_update(nm_field_t f, const nm_field_desc* desc, nm_stream_t stream); /* re-pack */
int nm_field_destroy(nm_field_t f);
/* Split-half modes: *flag = 1 if any kernel launched on this handle since the last call saw a value
 * outside the fp16 range (|v| >= 65504: the outputs of those launches are unusable, re-run with
 * mlp_precision 0), else 0.  Synchronises `stream`; resets the flag. */
int nm_field_overflow(nm_field_t f, int* flag, nm_stream_t stream);
/* ABI v11 -- the same flag with NO synchronisation ("no hidden sync", SURVEY 8b threading row; the reference's renderer returns
 * unsynchronised tensors, models/renderer.py:353-368): queues a copy of the flag into *host_flag on `stream` and returns.  host_flag must be
 * PINNED host memory that stays valid until the stream has passed this point; the caller reads it after an event of its own.  Nothing is
 * reset: on reading 1 call nm_field_overflow (resets) and re-run with mlp_precision 0. */
int nm_field_overflow_post(nm_field_t f, int* host_flag, nm_stream_t stream);

/* Tables + scalars that change without re-packing (borrowed device pointers, [V,dim]). */
typedef struct nm_field_tables {
    const float* geometry_features;  /* [V, geometry_dim] */
    const float* color_features;     /* [V, color_dim] */
    const float* indicator_vector;   /* [V, 3] */
    float indicator_weight;          /* w1 */
    float s;                         /* forward_s() = exp(ln_s * speed_factor) */
} nm_field_tables;

/* sdf (and nabla = d sdf / d xyz if nabla != NULL) at P points (device [P,3]).
 * scratch: device buffer of nm_field_scratch_bytes(P) bytes. */
int64_t nm_field_scratch_bytes(int64_t P);
int nm_field_density(nm_field_t f, nm_grid_t g, const nm_field_tables* t, const float* xyz,
                     int64_t P, float* sdf, float* nabla, void* scratch, nm_stream_t stream);
/* NeuMesh.forward(need_nablas=True): sdf [P], rgb [P,3]; optional nabla [P,3], ds [P],
 * idx [P,8] int64, w [P,8] (return_ds=True path used by the editing wrappers). */
int nm_field_forward(nm_field_t f, nm_grid_t g, const nm_field_tables* t, const float* xyz,
                     const float* view_dirs, int64_t P, float* sdf, float* rgb, float* nabla,
                     float* ds, int64_t* idx, float* w, void* scratch, nm_stream_t stream);
/* NeuMesh.forward_color(d, view_dirs, color_features, indices, weights, nabla). */
int nm_field_color(nm_field_t f, const float* color_features, const float* ds,
                   const float* view_dirs, const int64_t* idx, const float* w,
                   const float* nabla, int64_t P, float* rgb, void* scratch,
                   nm_stream_t stream);

/* -------------------------------------------------------------------------------- renderer */
typedef struct nm_render_cfg {
    float obj_bounding_radius;   /* 1.0 */
    int32_t N_samples;           /* 64 */
    int32_t N_importance;        /* 64 */
    int32_t N_upsample_iters;    /* 4; N_importance % N_upsample_iters == 0 */
    int32_t bounded_near_far;    /* 1 */
    int32_t calc_normal;         /* 0/1 */
    int32_t white_bkgd;          /* 0/1 */
    int32_t probe_grid;          /* 256 (compute_bounded_near_far sample_grid) */
    float probe_thresh;          /* 0.1 (distance_thresh) */
    float near_bypass, far_bypass; /* < 0 => unset */
    uint32_t flags;              /* NM_RENDER_* bits, 0 = defaults */
    int32_t chain_tiles;         /* regular-grid passes: max 4-sample tiles a wave chains; 0 = default (8) */
    int32_t fine_group_rays;     /* rays per depth-bucket group of an importance pass: 64/128/256/512; 0 = default (512) */
    int32_t mid_group_rays;      /* rays per depth-bucket group of the mid-point pass: 16/32/64; 0 = default (64) */
    float weight_eps;            /* 0 = exact (default).  > 0: a mid-point (and the nabla of a sample) whose visibility
                                    weight is below weight_eps is treated like one of weight 0 in the radiance / normal
                                    sums, i.e. never evaluated: rgb and normals move by less than (N-1) * weight_eps,
                                    depth and acc not at all (their weights come from the sample SDFs).  The one
                                    setting that is NOT bit-identical to evaluating everything. */
    /* Texture editing (editing/texture_neumesh/texture_neumesh.py:53-122: TextureEditableNeuMesh.forward), optional.
       n_edit reference models: where a mid-point's interpolation weight sits on vertices painted from reference i
       (edit_mask[i][v] != 0), the colour is blended, in the order i = 0, 1, ..:
           paint = sum_k w_k [painted_k], rest = sum_k w_k [not painted_k]; where paint > 0:
           colour = colour * (rest / (paint + rest)) + colour_i * (paint / (paint + rest)),
       colour_i = reference i's colour MLP on edit_color_features interpolated with the painted neighbours' renormalised
       weights (same ds; view direction and nabla rotated into the reference's frame when edit_use_rot[i]).  Geometry, depth, acc and
       normals are the main model's. */
    int32_t n_edit;              /* 0 = plain model; at most NM_MAX_EDIT */
    int32_t code_dims;           /* geometry_dim | color_dim << 16 of the field the call renders: sizes the K-NN records of the workspace
                                    (nm_render_workspace_bytes; 0 = not given: records of the maximum width, 64 + 64 floats).  nm_render_rays
                                    refuses a field whose code widths exceed the ones given here */
    nm_field_t edit_field[4];    /* reference models: their colour MLPs (same color_dim / embedders as the main model) */
    const uint8_t* edit_mask[4]; /* device [V] */
    const float* edit_color_features; /* device [V, color_dim]  (main_editing_colorfeats) */
    int32_t edit_use_rot[4];     /* != 0: reference i lives in another frame (T_r_m_list): its colour call takes the view    */
    float edit_rot[4][9];        /* direction and the nabla rotated by this row-major 3x3 matrix (transform_direction)      */
    const float* u_rand;         /* NULL: deterministic importance samples (sample_pdf(det=True), perturb=False).  Else device
                                    [N_upsample_iters][R][N_importance / N_upsample_iters] uniform numbers in [0,1): the stratum
                                    positions of sample_pdf(det=False) (rend_util.py:300-302), rows in the CALLER's ray order */
    int32_t mid_passes;          /* ABI v11: the mid-point stage (K-NN records, value + tangent MLP, colour MLP, texture blend) runs in this many
                                    sub-passes over contiguous ray ranges that share one record region: the workspace shrinks from 63 to
                                    (28 + 35 / mid_passes) KB per ray at 32 + 32-d codes.  0 = default (3); 1 = one pass (the round-5 layout);
                                    at most 16; fewer are used while a sub-pass would hold fewer than 32 768 rays.  No result bit changes. */
    /* (ABI v11: the v10 fields overlap / knn_keep / mlp_prio -- K-NN kernels yielding to other chunks' MLP kernels -- are gone: the mode was
       measured 5-15 % slower in every variant, profiles/r05_overlap_sweep.txt) */
} nm_render_cfg;
#define NM_MAX_EDIT 4

/* nm_render_cfg.flags.  None of them changes a result bit (tests compare the variants); they select
 * the evaluation strategy, mostly for A/B measurements:
 *   FULL_PROBES   evaluate all probe_grid probes of compute_bounded_near_far (renderer.py:79-102)
 *                 instead of only those before the first / after the last one below the threshold
 *   NO_ZERO_SKIP  evaluate the mid-points whose visibility weight is exactly 0 as well
 *   NO_RAY_SORT   process the rays in the caller's order (default: Morton order of closest approach)
 *   NO_MID_ORDER  hand the mid-points to waves as (16 rays x 4 samples) tiles, not by depth buckets
 *   EAGER_NABLAS  (calc_normal) evaluate the nabla of every sample point inside the sampling passes instead of
 *                 afterwards and only where the sample's visibility weight is not zero */
#define NM_RENDER_FULL_PROBES 1u
#define NM_RENDER_NO_ZERO_SKIP 2u
#define NM_RENDER_NO_RAY_SORT 4u
#define NM_RENDER_NO_MID_ORDER 8u
#define NM_RENDER_EAGER_NABLAS 16u
/*   SAMPLE_ONLY   stop after the sample placement (renderer.py:162-259, the part the reference runs under no_grad): only
 *                 dbg->d_all (required), dbg->near_far and dbg->sdf_all are written; rgb / depth / acc / normals may be NULL.
 *                 The training renderer places its samples with this call and queries the field with autograd afterwards. */
#define NM_RENDER_SAMPLE_ONLY 32u

int64_t nm_render_workspace_bytes(const nm_render_cfg* cfg, int64_t R);

/* Optional per-ray / per-sample debug outputs (device pointers, NULL to skip). */
typedef struct nm_render_debug {
    float* near_far;    /* [R,2] */
    float* d_all;       /* [R, N_samples+N_importance] sorted sample depths */
    float* sdf_all;     /* [R, N_samples+N_importance] */
    float* nablas_all;  /* [R, N_samples+N_importance, 3] (calc_normal) */
    float* radiance;    /* [R, N-1, 3] */
    float* sdf_coarse;  /* [R, N_samples] */
} nm_render_debug;

/* One chunk of R rays (device [R,3] each; rays_d need not be normalised, renderer.py:153).
 * Outputs: rgb [R,3], depth [R], acc [R], normals [R,3] (NULL unless calc_normal). */
int nm_render_rays(nm_field_t f, nm_grid_t g, const nm_field_tables* t, const float* rays_o,
                   const float* rays_d, int64_t R, const nm_render_cfg* cfg, float* rgb,
                   float* depth, float* acc, float* normals, const nm_render_debug* dbg,
                   void* workspace, nm_stream_t stream);

/* ------------------------------------------------------------------ per-ray stages, one by one
 * The stages nm_render_rays chains internally, exposed separately so that a caller whose FIELD is
 * not a plain NeuMesh (the editing tools wrap it: editing/texture_neumesh/texture_neumesh.py:41-122)
 * can run the reference's render_rayschunk (models/renderer.py:162-350) with its own model methods
 * between them.  All arrays are device, fp32, ray-major; `cap` = row length of d / sdf / d_mid.
 *   nm_rays_setup     renderer.py:153 + rend_util.py:179-199 -> dirn [R,3], near_far [R,2]
 *   nm_rays_points    depth mode 2: near*(1-t)+far*t, t = linspace(0,1,P) (also stored to
 *                     depth_out[r*cap + off + p] if non-NULL); mode 1: depth[r*cap + off + p];
 *                     -> xyz [R,P,3] = o + d * dirn
 *   nm_rays_bounds    renderer.py:88-102 from the probes' projected distances ds [R,G]
 *   nm_rays_upsample  merge the m samples appended last time, then draw n_new new depths into
 *                     d[:, n:n+n_new] (renderer.py:209-245,255-258; rend_util.py:276-319);
 *                     perturb=True: the caller passes its torch.rand as u (rend_util.py:300-302)
 *   nm_rays_finalize  last merge + mid-point depths d_mid[:, :n-1] (renderer.py:266)
 *   nm_rays_composite renderer.py:278,302-333 */
int nm_rays_setup(const float* rays_o, const float* rays_d, int64_t R, float radius, float* dirn,
                  float* near_far, nm_stream_t stream);
int nm_rays_points(const float* rays_o, const float* dirn, int64_t R, int P, int mode,
                   const float* near_far, const float* depth, int cap, int off, float* depth_out,
                   float* xyz, nm_stream_t stream);
int nm_rays_bounds(const float* ds_probe, int64_t R, int G, float thresh, const float* near_far_in,
                   float* near_far_out, nm_stream_t stream);
int nm_rays_upsample(float* d, float* sdf, int64_t R, int cap, int n, int m, int it, int n_new,
                     const float* u /* [R,n_new] uniform randoms = sample_pdf(det=False), or NULL = det=True */,
                     nm_stream_t stream);
int nm_rays_finalize(float* d, float* sdf, int64_t R, int cap, int n, int m, float* d_mid,
                     nm_stream_t stream);
int nm_rays_composite(const float* sdf, const float* d, int64_t R, int cap, int N, float s,
                      const float* rgb_mid, const float* nablas, int white_bkgd, float* rgb,
                      float* depth, float* acc, float* normals, nm_stream_t stream);

/* ----------------------------------------------------------------------------- ray set-up
 * rend_util.get_rays (utils/rend_util.py:123-176, pose-matrix branch, N_rays=-1) for the pixels
 * [first_pixel, first_pixel+count) of an H x W image in row-major order: pixel (x = p % W,
 * y = p / W) is lifted to z = 1 with the pin-hole intrinsics (utils/rend_util.py:95-118),
 * normalised and rotated by c2w[:3,:3]; the origin is c2w[:3,3].  Each rank of a ray-sharded render
 * generates its own pixel block on the device: no host->device ray traffic. */
typedef struct nm_camera {
    float c2w[12];            /* rows 0..2 of the camera-to-world matrix, row-major [3][4] */
    float fx, fy, cx, cy, sk; /* intrinsics[0,0], [1,1], [0,2], [1,2], [0,1] */
    int32_t H, W;
} nm_camera;
int nm_make_rays(const nm_camera* cam, int64_t first_pixel, int64_t count, float* rays_o,
                 float* rays_d, nm_stream_t stream);
/* The same for a LIST of row-major pixel indices (device pointer, int64; entries outside the frame are clamped):
 * the pixels of a rank's interleaved tiles in a ray-sharded frame (neumesh_amd/sharded.py), or the random pixel
 * selection of a training step (rend_util.get_rays with N_rays > 0, utils/rend_util.py:150-160). */
int nm_make_rays_indexed(const nm_camera* cam, const int64_t* pixels, int64_t count, float* rays_o,
                         float* rays_d, nm_stream_t stream);

/* --------------------------------------------------------------------- first-hit surface points
 * root_finding_surface_points of models/ray_casting.py:45-200 (with run_secant_method :12-38) for a NeuMesh field: N_steps
 * proposals d_j = near (1 - t_j) + far t_j, t = linspace(0, 1, N_steps), val_j = sdf(o + d_j dir) - logit_tau; a ray's hit is
 * its FIRST sign change of val, provided it goes from outside (> 0) to inside and val_0 > 0; the root is refined by
 * n_secant_steps >= 0 regula-falsi steps (0 = the first secant estimate; -1 = the reference's behaviour for method != "secant":
 * depth 1 at the hits).
 *   rays_d: unit directions (surface_render normalises, :262).  near_far: [R][2] per-ray bounds, or NULL for cfg.near / cfg.far.
 *   d_out [R]: hit depth; no hit: +inf (fill_inf) or the ray's far; 0 where the first proposal is already inside (:189-192).
 *   pt_out [R][3]: o + d dir at the hits, (1,1,1) elsewhere (:177-180).  mask / mask_sign_change [R]: uint8.
 * Same values as the reference (fp32, its operation order); a ray's proposals are only evaluated up to the block that holds
 * its first sign change (nothing the routine returns depends on later ones).  The call reads the number of rays still walking
 * back from the device after each block of 16 proposals: it SYNCHRONISES the stream (unlike every other entry point).
 * workspace: nm_surface_workspace_bytes(field, R) bytes (rays are processed in internal chunks of 2^18). */
typedef struct {
    float near, far;          /* used when near_far == NULL */
    int32_t N_steps;          /* 256 in the reference */
    float logit_tau;
    int32_t n_secant_steps;   /* 8 in the reference; -1: no secant method */
    int32_t fill_inf;
    float scene_radius;       /* only orders the rays in space (any positive value gives the same results) */
} nm_surface_cfg;
int64_t nm_surface_workspace_bytes(nm_field_t field, int64_t R);
int nm_surface_hits(nm_field_t field, nm_grid_t grid, const nm_field_tables* tables, const float* rays_o, const float* rays_d,
                    int64_t R, const float* near_far, const nm_surface_cfg* cfg, float* d_out, float* pt_out, uint8_t* mask,
                    uint8_t* mask_sign_change, void* workspace, nm_stream_t stream);

/* ----------------------------------------------------------------------------- training form of the field
 * What one optimisation step differentiates (models/trainer.py:75-81,186-209): NeuMesh.forward_with_nablas
 * (neumesh.py:146-153) at the samples and NeuMesh.forward (neumesh.py:113-136) at the mid-points -- sdf, nabla = d sdf / d xyz
 * and rgb -- with the gradients of a loss on those outputs with respect to every parameter the reference trains: both MLPs,
 * the geometry / colour code tables, the indicator vectors and the indicator weight.  The reference gets them from an
 * autograd graph of ~60 torch ops per query and a create_graph=True gradient for nabla (neumesh.py:223-232); here
 * nm_train_forward keeps every intermediate in `workspace` and nm_train_backward is the closed-form reverse pass (nabla is
 * the kernel's forward-mode tangent, so a cotangent on it -- the eikonal loss -- needs no second-order machinery).
 *
 * `desc` carries the fp32 weights in PyTorch layout ([out, in], weight-norm already folded: the caller's autograd maps the
 * returned gradient of a folded weight to its g / v parameters).  Operands and results are fp32 either way; mlp_precision selects the
 * matrix pipe of the layer products: 0 = v_mfma_f32_32x32x2_f32, anything else = the bf16 pipe with every fp32 operand cut into three
 * exact bf16 pieces and six piece products per product (same error against float64, 1.5 x faster; csrc/nm_gemm.h).
 * view_dirs == NULL: geometry only (rgb is not computed).  with_nabla == 0: sdf only (forward_density_only under autograd).
 * nm_train_backward must follow an nm_train_forward on the same workspace, P and flags; g_sdf [P], g_nabla [P,3], g_rgb [P,3]
 * are the cotangents (NULL = zero).  Gradients are ADDED into the non-NULL members of `out` (device pointers, shapes of the
 * parameters; the caller zeroes them) with atomic adds: the summation order is not fixed from run to run. */
typedef struct nm_train_grads {
    float* geo_weight[8];      /* [W, in_geo] / [W, W] */
    float* geo_bias[8];        /* [W] */
    float* density_weight;     /* [1, W] */
    float* density_bias;       /* [1] */
    float* col_weight[8];
    float* col_bias[8];
    float* rgb_weight;         /* [3, W] */
    float* rgb_bias;           /* [3] */
    float* geometry_features;  /* [V, geometry_dim] */
    float* color_features;     /* [V, color_dim] */
    float* indicator_vector;   /* [V, 3] */
    float* indicator_weight;   /* [1]: d / d w1 (the caller chains through its sigmoid) */
} nm_train_grads;
int64_t nm_train_workspace_bytes(const nm_field_desc* desc, int64_t P);
int nm_train_forward(const nm_field_desc* desc, nm_grid_t g, const nm_field_tables* t, const float* xyz, const float* view_dirs,
                     int64_t P, int with_nabla, float* sdf, float* nabla, float* rgb, void* workspace, nm_stream_t stream);
int nm_train_backward(const nm_field_desc* desc, nm_grid_t g, const nm_field_tables* t, int64_t P, int with_nabla, int with_color,
                      const float* g_sdf, const float* g_nabla, const float* g_rgb, void* workspace,
                      const nm_train_grads* out, nm_stream_t stream);

/* Differentiable compositing of the training step (renderer.py:264-333: sdf_to_alpha, alpha_to_w, the weighted sums) on given sample SDFs
 * sdf [R,N] at sorted depths, mid-point depths d_mid (row stride d_mid_stride >= N-1), radiance [R,N-1,3] (or NULL), nablas [R,N,3] (or NULL:
 * no normals), s = forward_s() read from DEVICE memory (one float).  Forward writes rgb [R,3], depth [R], acc [R], normals [R,3] and the
 * per-sample cdf [R,N], alpha / weights / transmittance [R,N-1] (the last three are also what backward reads back).  Backward: cotangents of
 * rgb / depth / acc / normals (NULL = zero) -> g_sdf [R,N], g_radiance [R,N-1,3], g_nablas [R,N,3] (either may be NULL), and g_s[0] += d/ds. */
int nm_train_composite_forward(const float* sdf, const float* s, const float* d_mid, int d_mid_stride, const float* radiance,
                               const float* nablas, int64_t R, int N, int white_bkgd, float* rgb, float* depth, float* acc, float* normals,
                               float* cdf, float* alpha, float* weights, float* transmittance, nm_stream_t stream);
int nm_train_composite_backward(const float* sdf, const float* s, const float* d_mid, int d_mid_stride, const float* radiance,
                                const float* nablas, int64_t R, int N, int white_bkgd, const float* cdf, const float* alpha,
                                const float* weights, const float* transmittance, const float* acc, const float* depth,
                                const float* g_rgb, const float* g_depth, const float* g_acc, const float* g_normals,
                                float* g_sdf, float* g_radiance, float* g_nablas, float* g_s, nm_stream_t stream);

/* ----------------------------------------------------------------------------- image assembly
 * What render.py:219-249 does on the host with the three outputs of a frame, per pixel, on the device
 * (the frame then leaves the GPU as 7 bytes per pixel instead of 28):
 *   rgb8[p][c]    = (uint8)(rgb[p][c] * 255.0f)                 integerify, render.py:183-184 (fp32 product, truncation)
 *   depth8[p]     = (uint8)((depth[p] / max_q depth[q]) * 255)  render.py:221-222, 253
 *   normal8[p][c] = (uint8)((normals[p][c] / 2 + 0.5) * 255)    render.py:233, 246-248
 * bgr != 0 writes rgb8 in B,G,R order (the channel swap before cv2.imwrite, render.py:236).  depth / normals and
 * their outputs may be NULL.  depth_max_scratch: one float of device scratch (the maximum is taken over THIS call's
 * count pixels: call it on whole frames).  Values outside [0, 256/255) are clamped to 0 / 255 (numpy's cast of
 * such values is undefined). */
int nm_assemble_frame(const float* rgb, const float* depth, const float* normals, int64_t count, int bgr,
                      uint8_t* rgb8, uint8_t* depth8, uint8_t* normal8, float* depth_max_scratch, nm_stream_t stream);

/* -------------------------------------------------------------------------- instrumentation
 * In-stream timing of the hot kernels inside ordinary calls (nm_render_rays, nm_field_*):
 * nm_profile_enable(1) clears the log and starts bracketing every launch of the K-NN/distance,
 * geometry-MLP (without / with tangent) and colour-MLP kernels with HIP events recorded on the
 * launch stream; nm_profile_read(kind) waits for those events and returns the summed kernel
 * time, the number of launches and the number of points processed.
 * kind: 0 = K-NN+distance (units = points actually SEARCHED: probes skipped by the first/last-hit
 *           walk and zero-weight mid-points are not counted),
 *       1 = geometry MLP (density only), 2 = geometry MLP (+tangent), 3 = colour MLP.
 * Process-wide log, mutex-protected; intended for one measuring thread (bench.py). */
int nm_profile_enable(int on);
int nm_profile_read(int kind, double* total_ms, int64_t* launches, int64_t* units);
/* Shader clock in MHz as one wave measures it over `micros` microseconds (its cycle counter against the constant 100 MHz counter).  Called on
 * a stream of its own while a workload runs on another, it returns the clock the chip holds under that load.  Synchronises `stream`. */
int nm_profile_clock(int micros, float* mhz, nm_stream_t stream);

/* Launches `iters` back-to-back passes of one internal kernel on `stream`, bracketed by HIP
 * events recorded on that same stream; returns the average duration in milliseconds.
 * which: as `kind` above (the MLP kernels get their inputs from one untimed K-NN pass). */
int nm_time_kernel(nm_field_t f, nm_grid_t g, const nm_field_tables* t, int which,
                   const float* xyz, const float* view_dirs, int64_t P, void* scratch,
                   int iters, float* avg_ms, nm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NEUMESH_HIP_H */
