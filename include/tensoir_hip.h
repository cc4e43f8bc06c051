/*
 * tensoir_hip.h -- C ABI of libtensoir_hip.so (MI355X / gfx950).
 *
 * The reference (Haian-Jin/TensoIR) has no FFI of its own: its hot path is a chain of ATen ops
 * behind a Python call surface.  This header is the boundary a maintainer
 * binds instead of those op chains; every entry point names the reference code it replaces
 * (paths relative to the TensoIR repository root).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes, no framework types.  Every pointer is a DEVICE pointer unless it is
 *     a `const Tir*` descriptor struct, which lives in HOST memory and is copied at launch.
 *   - the caller owns every buffer (inputs, outputs, workspaces); the library allocates nothing
 *     and keeps no state.  Calls are asynchronous on `stream` (a hipStream_t passed as void*);
 *     no hipDeviceSynchronize inside; safe for hipGraph capture.
 *   - fp32 I/O, int32 indices, int64 sizes.  Return 0 on success, a negative TIR_ERR_* /
 *     -(hipError_t) on failure; nothing throws across the ABI.
 *   - re-entrant; no internal threads.
 */
#ifndef TENSOIR_HIP_H
#define TENSOIR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TIR_VERSION 100            /* 0.1.0 */

#define TIR_OK               0
#define TIR_ERR_ARG         -1001  /* null pointer / negative size / inconsistent descriptor      */
#define TIR_ERR_UNSUPPORTED -1002  /* shape outside what the gfx950 kernels were built for        */
#define TIR_ERR_NO_DEVICE   -1003  /* no HIP device / kernel image not loadable on this device    */

/* ---------------------------------------------------------------------------------------------
 * Packed VM field (device-resident shadow of TensorVMSplit's parameters).
 * Reference layout: density_plane[i] [1,C,H_i,W_i], density_line[i] [1,C,R_i,1]
 * (models/tensoRF_rotated_lights.py:19-29); here channel-last so one bilinear tap is one
 * contiguous C*4-byte run (64 B for C=16, 192 B for C=48).
 *   plane i: H_i = grid[matMode[i][1]], W_i = grid[matMode[i][0]]; line i: R_i = grid[vecMode[i]]
 *   matMode = {{0,1},{0,2},{1,2}}, vecMode = {2,1,0}   (models/tensorBase_rotated_lights.py:398-399)
 * ------------------------------------------------------------------------------------------- */
typedef struct TirField {
    float aabb_min[3];
    float aabb_max[3];
    float inv_aabb[3];       /* 2/(max-min), fp32 as update_stepSize computes it (:611-612)        */
    int32_t grid[3];         /* gridSize x,y,z                                                      */
    float step_size;         /* stepSize (:615)                                                     */
    float distance_scale;    /* 25  (opt.py:77)                                                     */
    float density_shift;     /* -10 (opt.py:79)                                                     */
    float weight_thres;      /* rayMarch_weight_thres 1e-4 (:354)                                   */
    float near_, far_;       /* near_far (:385)                                                     */
    int32_t act;             /* 0 = softplus, 1 = relu  (feature2density :813-817)                  */
    int32_t n_dcomp;         /* density channels per plane (16), multiple of 4                      */
    int32_t n_acomp;         /* appearance channels per plane (48), multiple of 4                   */
    int32_t app_dim;         /* 27                                                                  */
    int32_t n_lights;
    const float* dplane[3];  /* [H_i][W_i][n_dcomp]                                                 */
    const float* dline[3];   /* [R_i][n_dcomp]                                                      */
    const float* aplane[3];  /* [H_i][W_i][n_acomp]                                                 */
    const float* aline[3];   /* [R_i][n_acomp]                                                      */
    const float* basis_t;    /* [3*n_acomp][32]: basis_mat^T, app_dim padded to 32 with zeros       */
    const float* light_line; /* [n_lights][3*n_acomp]  (light_line.weight)                          */
    const float* light_mean; /* [3*n_acomp] mean over lights (tensoRF_rotated_lights.py:160-161)    */
    const uint8_t* occ_nbr;  /* AlphaGridMask as 2x2x2 neighbourhood bytes (tir_pack_occupancy); NULL = no mask */
    int32_t occ_dim[3];      /* W,H,D of the mask volume                                            */
    float occ_aabb_min[3];   /* the mask's own aabb (:105) ...                                      */
    float occ_inv[3];        /* ... and (1/size)*2 (:107)                                           */
    float occ_lo[3];         /* world-space box that contains every point the mask can report as occupied (the occupied   */
    float occ_hi[3];         /* voxels' extent + one cell): the secondary march skips 32-sample steps that lie outside.  */
                             /* occ_lo >= occ_hi on any axis (e.g. all zeros) = not given, nothing is skipped.            */
    /* Launch options (performance only, results are identical either way).  They travel WITH the descriptor: the library
     * keeps no settable state and reads no environment variable, so two embedders in one process cannot disturb each other. */
    int32_t tune_lds_lines;  /* secondary march with the density lines staged in LDS: 0 = default (on), 1 = on, 2 = off */
    int32_t tune_xcd_order;  /* contiguous per-XCD work ranges in the gathers / secondary march: 0 = default (off), 1 = on, 2 = off */
} TirField;

/* fp16 shadow of the appearance planes / lines (same channel-last element order as TirField::aplane / aline; 16-byte aligned),
 * produced by tir_pack_half and read by tir_vm_app_fwd_h16. */
typedef struct TirFieldHalf {
    const void* aplane[3];   /* [H_i][W_i][n_acomp] fp16 */
    const void* aline[3];    /* [R_i][n_acomp]      fp16 */
} TirFieldHalf;

/* (internal launch records of tir_pack_half) */
#define TIR_HALF_MAX_JOBS 8
typedef struct TirHalfJob { const float* src; void* dst; int64_t n; unsigned* absmax; } TirHalfJob;
typedef struct TirHalfJobs { TirHalfJob job[TIR_HALF_MAX_JOBS]; } TirHalfJobs;

/* One 3-layer decoder (in -> hidden ReLU -> hidden ReLU -> out, then activation):
 * MLPRender_Fea / MLPBRDF_PEandFeature (models/tensorBase_rotated_lights.py:122-146, :182-208).
 * `packed` is produced by tir_pack_mlp. */
typedef struct TirMlp {
    const float* packed;
    int32_t feat_dim;        /* 27                                                                  */
    int32_t pe;              /* fea_pe == view_pe == pos_pe (2)                                     */
    int32_t hidden;          /* 128                                                                 */
    int32_t out_dim;         /* 3 (rgb, normal) or 4 (albedo+roughness)                             */
    int32_t act;             /* 0 = sigmoid, 1 = tanh                                               */
    int32_t tune_grid;       /* launch option: persistent workgroups of a decoder launch, 0 = default (256 = one per CU) */
} TirMlp;

/* Environment light as spherical Gaussians + per-light z-rotation
 * (models/tensorBase_rotated_lights.py:461-488, :577-606). */
typedef struct TirEnvSG {
    const float* sgs;        /* [n_sg][7]  lobe xyz, lambda, mu rgb (raw parameters)                */
    const float* rot;        /* [n_lights][9] row-major light_rotation_matrix                       */
    int32_t n_sg;
    int32_t n_lights;
} TirEnvSG;

int  tir_version(void);
const char* tir_error_string(int code);
/* 0 when a gfx950 device is present and the code object loads. */
int  tir_device_check(void);

/* ---- packing (build the shadow copies; re-run after any parameter update / upsample / shrink:
 *      train_tensoIR.py:385-422) -------------------------------------------------------------- */
/* [C,H,W] -> [H,W,C]   (also lines with W=1) */
int tir_pack_plane(const float* src, float* dst, int32_t C, int32_t H, int32_t W, void* stream);
/* float volume [D][H][W] -> (D+1)*(H+1)*(W+1) neighbourhood bytes: byte (z0+1,y0+1,x0+1), bit dx+2dy+4dz =
 * voxel (x0+dx,y0+dy,z0+dz) inside the grid and > 0.5.  The trilinear "sample_alpha(...) > 0" test of a
 * point then needs one byte. */
int tir_pack_occupancy(const float* vol, uint8_t* nbr, int32_t W, int32_t H, int32_t D, void* stream);
/* basis_mat.weight [app_dim][n_in] -> [n_in][32] */
int tir_pack_basis(const float* w, float* dst, int32_t app_dim, int32_t n_in, void* stream);
/* light_line.weight [L][n] -> mean over L [n] */
int tir_light_mean(const float* light_line, float* mean, int32_t L, int32_t n, void* stream);
int64_t tir_mlp_packed_floats(int32_t feat_dim, int32_t pe, int32_t hidden, int32_t out_dim);
/* nn.Linear weights ([out][in] row-major) + biases of mlp.{0,2,4} -> packed blob */
int tir_pack_mlp(const float* w0, const float* b0, const float* w1, const float* b1,
                 const float* w2, const float* b2, int32_t feat_dim, int32_t pe, int32_t hidden,
                 int32_t out_dim, float* packed, void* stream);

/* ---- K2: compute_densityfeature + feature2density (models/tensoRF_rotated_lights.py:95-110,
 *      models/tensorBase_rotated_lights.py:813-817).  xyz normalised to [-1,1]^3.
 *      feat / sigma may each be NULL. */
int tir_vm_density_fwd(const TirField* f, const float* xyz, float* feat, float* sigma,
                       int64_t n, void* stream);

/* ---- a2: AlphaGridMask.sample_alpha(xyz) > 0 (models/tensorBase_rotated_lights.py:112-119) at
 *      world-space points; hit[n] = 1/0.  Needs f->occ_nbr. */
int tir_occupancy_query(const TirField* f, const float* xyz, uint8_t* hit, int64_t n, void* stream);

/* ---- occupancy-grid maintenance (models/tensorBase_rotated_lights.py:737-811, :819-837).
 *      tir_dense_alpha: getDenseAlpha + compute_alpha on a gx*gy*gz lattice spanning the aabb; lin_* are the
 *      caller's linspace(0,1,g) tables (device), alpha [gx][gy][gz] = 1 - exp(-sigma * length), sigma = 0 where the
 *      current mask (f->occ_nbr, may be NULL) culls the point.
 *      tir_alpha_pool: updateAlphaMask's clamp -> transpose -> 3x3x3 max-pool -> threshold; vol [gz][gy][gx] float 0/1,
 *      bbox (optional, 6 ints pre-set to {INT_MAX x3, -1 x3}) receives the index bounding box of the occupied voxels.
 *      tir_filter_rays: filtering_rays; mask[i] = ray i may hit something (bbox_only: slab test only). */
int tir_dense_alpha(const TirField* f, const float* lin_x, const float* lin_y, const float* lin_z,
                    int32_t gx, int32_t gy, int32_t gz, float length, float* alpha, void* stream);
int tir_alpha_pool(const float* alpha, int32_t gx, int32_t gy, int32_t gz, float thres, float* vol,
                   int32_t* bbox, void* stream);
int tir_filter_rays(const TirField* f, const float* rays, int64_t n, int32_t n_samples, int32_t bbox_only,
                    uint8_t* mask, void* stream);

/* ---- K6: analytic d sigma/d xyz and derived normal -normalize(grad, eps=1e-6)
 *      (compute_derived_normals models/tensorBase_rotated_lights.py:839-856 ->
 *       compute_densityfeature_with_xyz_grad models/tensoRF_rotated_lights.py:113-129 ->
 *       models/relight_utils.py:57-107).  sigma/grad/normal may each be NULL.  n_dev: as tir_vm_app_fwd. */
int tir_density_grad_fwd(const TirField* f, const float* xyz, float* sigma, float* grad,
                         float* normal, int64_t n, const int32_t* n_dev, void* stream);

/* ---- K4: compute_appfeature / compute_intrinfeature / compute_bothfeature
 *      (models/tensoRF_rotated_lights.py:132-224).  light_idx (per point, or per `idx_map` entry when
 *      idx_map != NULL: light_idx[idx_map[p]]; with idx_div > 1 the selector is divided by idx_div first, e.g.
 *      pair id -> surface point) may be NULL when rad_feat is NULL.
 *      rad_feat / int_feat [n][out_stride] (app_dim <= out_stride <= 32; columns >= app_dim are written
 *      as 0; 32 gives aligned 128-byte rows), either may be NULL.
 *      n_dev (optional device pointer): the kernels process min(n, *n_dev) points, so a caller that sized its
 *      buffers for n can launch before the producing kernel's count has reached the host. */
int tir_vm_app_fwd(const TirField* f, const float* xyz, const int32_t* light_idx,
                   const int32_t* idx_map, float* rad_feat, float* int_feat, int32_t out_stride,
                   int32_t idx_div, int64_t n, const int32_t* n_dev, void* stream);

/* Intrinsic feature of JITTERED points xyz + scale * N(0,1) (the smoothness pass of
 * models/tensorBase_rotated_lights.py:937-938) with the noise drawn inside the gather kernel: Philox4x32-10 keyed by
 * (seed, offset) -- the caller framework's generator state -- with the point index as counter, Box-Muller.  rng_state
 * (device int64 {seed, offset}, optional) overrides the by-value pair (HIP-graph replays).  xyz_out [n][3] receives the
 * jittered points (aux input of the BRDF decoder), int_feat [n][out_stride] their intrinsic features. */
int tir_vm_app_jitter_fwd(const TirField* f, const float* xyz, int64_t n, const int32_t* n_dev, float scale,
                          uint64_t seed, uint64_t offset, const int64_t* rng_state, float* xyz_out,
                          float* int_feat, int32_t out_stride, void* stream);

/* tir_vm_app_fwd (radiance + intrinsic features of xyz) and tir_vm_app_jitter_fwd (intrinsic features of the jittered xyz)
 * in ONE launch -- the two appearance gathers of the primary stage (n_acomp == 48).  Arguments as in the two calls. */
int tir_vm_app_primary_fwd(const TirField* f, const float* xyz, const int32_t* light_idx, const int32_t* idx_map,
                           float* rad_feat, float* int_feat, int32_t out_stride, int64_t n, const int32_t* n_dev,
                           float scale, uint64_t seed, uint64_t offset, const int64_t* rng_state, float* xyz_out,
                           float* int_feat_jit, void* stream);
/* the same launch with basis_mat's contraction (models/tensoRF_rotated_lights.py:159-165, :191-195) on v_mfma_f32_16x16x32_f16 and
 * both operands split x = hi + lo in fp16 (three products, fp32 accumulate: ~2^-21 relative per product -- the features agree with
 * the exact-fp32 contraction of tir_vm_app_primary_fwd to ~2e-6 of their scale): a quarter of the matrix-pipe time of the exact-fp32
 * instruction, which runs at the vector rate.  Opt-in (TENSOIR_APP_CONTRACTION=x3): fp16 residues of products below 0.125 are
 * subnormal, so the features are good to ~1e-6 of their scale, which a trained BRDF decoder amplifies to 1e-4 on the albedo map. */
int tir_vm_app_primary_x3_fwd(const TirField* f, const float* xyz, const int32_t* light_idx, const int32_t* idx_map,
                              float* rad_feat, float* int_feat, int32_t out_stride, int64_t n, const int32_t* n_dev,
                              float scale, uint64_t seed, uint64_t offset, const int64_t* rng_state, float* xyz_out,
                              float* int_feat_jit, void* stream);
/* tir_vm_app_fwd with the contraction on v_mfma_f32_16x16x32_f16 and both operands split x = hi + lo in fp16 (three products,
 * ~2^-21 relative per product; see tir_vm_app_primary_x3_fwd): n_acomp 16 / 24 / 48. */
int tir_vm_app_fwd_x3(const TirField* f, const float* xyz, const int32_t* light_idx,
                      const int32_t* idx_map, float* rad_feat, float* int_feat, int32_t out_stride,
                      int32_t idx_div, int64_t n, const int32_t* n_dev, void* stream);
/* same contract with the 144 x 27 contraction on v_mfma_f32_16x16x32_bf16 and every operand split x = hi + lo in bf16
 * (three products, fp32 accumulate: parity grade, features agree with the exact kernel to ~1e-6); n_acomp <= 64. */
int tir_vm_app_fwd_bf16x3(const TirField* f, const float* xyz, const int32_t* light_idx,
                          const int32_t* idx_map, float* rad_feat, float* int_feat, int32_t out_stride,
                          int32_t idx_div, int64_t n, const int32_t* n_dev, void* stream);
/* same contract, one-sample-per-lane VALU kernel (cross-check of the matrix-core kernel) */
int tir_vm_app_fwd_valu(const TirField* f, const float* xyz, const int32_t* light_idx,
                        const int32_t* idx_map, float* rad_feat, float* int_feat, int32_t out_stride,
                        int32_t idx_div, int64_t n, const int32_t* n_dev, void* stream);

/* ---- K5: positional_encoding + 3-layer MLP + activation
 *      (models/tensorBase_rotated_lights.py:12-17, :136-146, :198-208).
 *      input row = [feat, aux, PE(feat), PE(aux)]; aux row p is aux[aux_map ? aux_map[p] : p];
 *      feat rows are feat_stride floats apart (>= feat_dim).  aux_mod > 0: the aux row index is taken modulo
 *      aux_mod (pair id -> direction).  n_dev: as tir_vm_app_fwd. */
int tir_mlp_fwd(const TirMlp* m, const float* feat, int32_t feat_stride, const float* aux,
                const int32_t* aux_map, int32_t aux_mod, float* out, int64_t n, const int32_t* n_dev,
        void* stream);
/* same contract on v_mfma_f32_32x32x16_bf16 with every operand split x = hi + lo in bf16 and the three
 * products hi*hi + hi*lo + lo*hi accumulated in fp32 (~16 mantissa bits; measured <= 2e-6 abs on the
 * decoder outputs): the parity-grade fast path. */
int tir_mlp_fwd_bf16x3(const TirMlp* m, const float* feat, int32_t feat_stride, const float* aux,
                       const int32_t* aux_map, int32_t aux_mod, float* out, int64_t n, const int32_t* n_dev,
        void* stream);

/* ---- precision policy for INDIRECT light (the radiance of the secondary-ray records, models/relight_utils.py:818-832).
 *      tir_pack_half: fp32 -> fp16 copies (round to nearest even, SATURATING: |x| > 65504 -> +-65504, never inf) of up to
 *      TIR_HALF_MAX_JOBS tables in one launch; srcs / dsts / counts are HOST arrays, tables 16-byte aligned.
 *      tir_pack_half_checked: the same, and absmax[i] (DEVICE floats, zeroed by the caller before the call) receives max |x| of
 *      table i (a NaN anywhere in the table is reported as NaN); dsts[i] == NULL scans table i without writing a copy (light
 *      rows, basis_mat).  RANGE CONTRACT of the fp16 gather below: interpolation and the products (plane * line) * light-row run
 *      on the packed fp16 pipe WITHOUT a saturation test in the hot loop, so the caller must establish
 *      max|plane_i| * max|line_i| * max(1, max|light row|) < 65504 for i = 0..2 and max|basis_mat| < 65504 from these maxima
 *      (bilinear taps are convex combinations, so the bound is rigorous) and use the fp32 gather (tir_vm_app_fwd) otherwise --
 *      the host mirror does exactly that (tensoir_amd/relight.py: _indirect_mode, ops.HalfRange).
 *      tir_vm_app_fwd_h16 = tir_vm_app_fwd(rad_feat only) for n_acomp == 48 on the fp16 shadow `fh` of f->aplane / f->aline:
 *      half the bytes through the vector L1 (the bound of the fp32 gather), interpolation and light-row product on the packed
 *      fp16 pipe (v_pk_fma_f16: two channels per instruction, rounded after every step), the basis_mat contraction on
 *      v_mfma_f32_32x32x16_f16 (fp32 accumulate).  ~5e-4 relative on a
 *      feature: NOT parity grade on its own -- the product path uses it only for the secondary-ray records, whose radiance is
 *      averaged over a ray's records and the light directions before it reaches rgb_with_brdf_map.  Other arguments as
 *      tir_vm_app_fwd. */
int tir_pack_half(const float* const* srcs, void* const* dsts, const int64_t* counts, int32_t n_tables, void* stream);
int tir_pack_half_checked(const float* const* srcs, void* const* dsts, const int64_t* counts, int32_t n_tables, float* absmax,
                          void* stream);
int tir_vm_app_fwd_h16(const TirField* f, const TirFieldHalf* fh, const float* xyz, const int32_t* light_idx,
                       const int32_t* idx_map, float* rad_feat, int32_t out_stride, int32_t idx_div, int64_t n,
                       const int32_t* n_dev, void* stream);

/* The two launches above fused (north_star: gathers and decoder "in one pass"): for every record s < min(n, *n_dev) the radiance
 * features of xyz[s] (light row light_idx[rec_map[s] / idx_div]) are gathered from the fp16 shadow, contracted with basis_mat and
 * decoded by `m` (the radiance decoder; view-direction columns from `table` row rec_map[s] % aux_mod, tir_mlp_aux_table) without
 * the feature rows ever reaching HBM.  out [n][m->out_dim].  n_acomp == 48, app_dim == 27.  Same precision class as the pair
 * tir_vm_app_fwd_h16 + tir_mlp_fwd_auxtab_f16 (results agree to fp32 summation order). */
int tir_indirect_fused_fwd(const TirField* f, const TirFieldHalf* fh, const TirMlp* m, const float* xyz,
                           const int32_t* light_idx, const int32_t* rec_map, int32_t idx_div, int32_t aux_mod,
                           const float* table, float* out, int64_t n, const int32_t* n_dev, void* stream);

/* The same stage (models/relight_utils.py:818-829: compute_appfeature -> renderModule on the secondary-ray records) in one
 * launch at the precision a TRAINED checkpoint needs (the auto policy's fallback when its self-check rejects the fp16 kernel;
 * before round 6 that fallback was tir_vm_app_fwd + tir_mlp_fwd_auxtab_bf16x3 with the feature rows through HBM): fp32 taps from
 * the parameters themselves (no shadow), fp32 interpolation, basis_mat contraction on fp16 hi + lo operands (three products),
 * decoder with fp16 activations and weights as fp16 + fp8 residue (tir_pack_mlp's OFF_F8 image, multiplied on the block-scaled
 * fp8 matrix instruction with the scale 2^-17 into the same fp32 accumulators; layer 3 exact).  n_lights <= 8.  Arguments as tir_indirect_fused_fwd without the shadow.  Deviation from the exact decoder on rgb_with_brdf_map:
 * measured per checkpoint by the policy's self-check (profiles/r06_*). */
int tir_indirect_fused_hp_fwd(const TirField* f, const TirMlp* m, const float* xyz, const int32_t* light_idx,
                              const int32_t* rec_map, int32_t idx_div, int32_t aux_mod, const float* table, float* out,
                              int64_t n, const int32_t* n_dev, void* stream);

/* Up to four decoders over the SAME n rows in one launch (split-bf16 matrix cores): the primary stage evaluates the
 * radiance, BRDF, jittered-BRDF and normal decoders (models/tensorBase_rotated_lights.py:927-955) on the same records.
 * mlps / feats / auxs / aux_maps / outs are HOST arrays of n_jobs entries (aux_maps or its entries may be NULL); feature
 * rows must be 16-byte aligned with a stride that is a multiple of 4 floats; outs[i] is [n][mlps[i]->out_dim]. */
int tir_mlp_fwd_multi_bf16x3(const TirMlp* const* mlps, const float* const* feats, int32_t feat_stride,
                             const float* const* auxs, const int32_t* const* aux_maps, float* const* outs,
                             int32_t n_jobs, int64_t n, const int32_t* n_dev, void* stream);

/* ---- aux-table variant of the split-bf16 decoder.  The 3 aux inputs of MLPRender_Fea are the view direction
 *      (models/tensorBase_rotated_lights.py:137-142): one value per ray in the primary stage, one per light direction for the
 *      secondary records -- few distinct rows for many decoder rows.  tir_mlp_aux_table evaluates, per aux row a, the part of
 *      layer 1 that depends on it alone: table[a][unit] = b0[unit] + sum over the 15 columns (aux, sin / cos PE(aux)) of
 *      W0[unit][col] x_col  (exact fp32; table is [n_aux][hidden], 16-byte aligned).  tir_mlp_fwd_auxtab_bf16x3 is
 *      tir_mlp_fwd_bf16x3 with that table in place of `aux` (row aux_map[s] -- or s when NULL -- modulo aux_mod when > 0): the
 *      layer-1 accumulators start from the table row, the matrix product runs over the other 135 inputs (9 instead of 10
 *      k-blocks).  Same results to fp32 rounding.  tir_mlp_fwd_multi_auxtab_bf16x3: tables[i] != NULL selects the variant for
 *      job i (auxs[i] is then unused and may be NULL). */
int tir_mlp_aux_table(const TirMlp* m, const float* aux, int64_t n_aux, float* table, void* stream);
int tir_mlp_fwd_auxtab_bf16x3(const TirMlp* m, const float* feat, int32_t feat_stride, const float* table,
                              const int32_t* aux_map, int32_t aux_mod, float* out, int64_t n, const int32_t* n_dev,
                              void* stream);
/* The aux-table decoder with ONE fp16 product per tile (v_mfma_f32_32x32x16_f16; operands rounded to 11 bits, fp32 accumulate,
 * layer 3 exact fp32): a third of the matrix work of tir_mlp_fwd_auxtab_bf16x3 at ~1e-4 absolute error per decoder output.
 * NOT parity grade by itself: meant for the radiance of the secondary-ray records (models/relight_utils.py:818-832), whose
 * error is averaged over a ray's records and the light directions before it reaches rgb_with_brdf_map.  Same arguments. */
int tir_mlp_fwd_auxtab_f16(const TirMlp* m, const float* feat, int32_t feat_stride, const float* table,
                           const int32_t* aux_map, int32_t aux_mod, float* out, int64_t n, const int32_t* n_dev,
                           void* stream);
/* The same launch for the training forward (h1, h2 [n][128] post-ReLU, as tir_mlp_train_fwd_bf16x3).  With aux_map == NULL and
 * aux_mod == 0 the table has one row PER DECODER ROW: that is how normals_kind == 'residue_prediction' is evaluated -- its decoder
 * MLPNormal_normal_and_PExyz (models/tensorBase_rotated_lights.py:236-262) feeds the derived normal as three more inputs of layer 1,
 * which the caller adds to the table rows (table += n W0[:, 3:6]^T) before the launch. */
int tir_mlp_train_fwd_auxtab_bf16x3(const TirMlp* m, const float* feat, int32_t feat_stride, const float* table,
                                    const int32_t* aux_map, int32_t aux_mod, float* out, float* h1, float* h2, int64_t n,
                                    const int32_t* n_dev, void* stream);
int tir_mlp_fwd_multi_auxtab_bf16x3(const TirMlp* const* mlps, const float* const* feats, int32_t feat_stride,
                                    const float* const* auxs, const int32_t* const* aux_maps,
                                    const float* const* tables, float* const* outs, int32_t n_jobs, int64_t n,
                                    const int32_t* n_dev, void* stream);
/* The same launch for the training forward: every job also writes its post-ReLU hidden activations h1s[i], h2s[i]
 * [n][128] (what tir_mlp_train_fwd_bf16x3 returns for one decoder). */
int tir_mlp_train_fwd_multi_bf16x3(const TirMlp* const* mlps, const float* const* feats, int32_t feat_stride,
                                   const float* const* auxs, const int32_t* const* aux_maps, float* const* outs,
                                   float* const* h1s, float* const* h2s, int32_t n_jobs, int64_t n,
                                   const int32_t* n_dev, void* stream);
/* single bf16 product (8 mantissa bits): reduced-precision mode, NOT parity grade (normals ~5e-3).  Everything that enters
 * a matrix product is rounded to bf16 once -- inputs, weights AND layer 1's bias, which rides as the weight of a constant-1
 * input: a bias of magnitude b carries an error of up to 2^-9 b (the split-bf16 entry keeps ~16 bits of it, the exact entry
 * all); tests/test_gpu_parity.py::test_decoder_large_biases states the resulting bounds. */
int tir_mlp_fwd_bf16(const TirMlp* m, const float* feat, int32_t feat_stride, const float* aux,
                     const int32_t* aux_map, int32_t aux_mod, float* out, int64_t n, const int32_t* n_dev,
        void* stream);
/* same contract, plain VALU kernel (any hidden size); used to cross-check the MFMA kernel */
int tir_mlp_fwd_valu(const TirMlp* m, const float* feat, int32_t feat_stride, const float* aux,
                     const int32_t* aux_map, int32_t aux_mod, float* out, int64_t n, const int32_t* n_dev,
        void* stream);

/* ---- K1+K2+K3 primary march: sample_ray + alpha-mask cull + density + raw2alpha
 *      (models/tensorBase_rotated_lights.py:705-724, :892-921, :21-28).
 *      rays [B][6]; ray_jitter [B] (is_train stratification, :717) or NULL.
 *      Outputs: weight [B][S]; acc[B] = sum w; depth[B] = sum w*z; t_end[B] = prod(1-a+1e-10);
 *      app_count[B] = #{w > weight_thres}.  t_stop: a ray stops marching once its running
 *      transmittance drops below t_stop (remaining weights are written as 0; the error in acc is
 *      < t_stop); pass 0 for the exact full march.
 *      stats (optional, NULL to skip): *stats += number of samples whose density was gathered
 *      (in bbox and not culled by the occupancy mask) -- the unit of the roofline accounting. */
int tir_march_primary_fwd(const TirField* f, const float* rays, const float* ray_jitter,
                          int32_t B, int32_t S, float t_stop, float* weight, float* acc,
                          float* depth, float* t_end, int32_t* app_count,
                          unsigned long long* stats, void* stream);

/* tir_march_primary_fwd plus the by-products that used to be separate launches of a step (each optional, NULL/0 to skip):
 *   viewdirs [B][3]     rays[:, 3:6] as a contiguous table = aux input of the radiance decoder
 *                       (the `viewdirs` of models/tensorBase_rotated_lights.py:872);
 *   zero_words[n_zero]  int32 counters of LATER kernels of the same pass, re-armed here (n_zero <= 256);
 *   ticket/offsets/cap/total  tir_exclusive_scan_capped(app_count) -> offsets[B+1], *total, done by the workgroup that
 *                       finishes last; `ticket` is one device int32 that must be zero before the first use (it re-arms
 *                       itself).  Replaces the boolean-mask bookkeeping of :924-926 without a second launch. */
int tir_march_primary_fused_fwd(const TirField* f, const float* rays, const float* ray_jitter,
                                int32_t B, int32_t S, float t_stop, float* weight, float* acc,
                                float* depth, float* t_end, int32_t* app_count, unsigned long long* stats,
                                float* viewdirs, int32_t* zero_words, int32_t n_zero, int32_t* ticket,
                                int32_t* offsets, int32_t cap, int32_t* total, void* stream);

/* exclusive scan of counts[n] -> offsets[n+1] (offsets[n] = total) */
int tir_exclusive_scan(const int32_t* counts, int32_t* offsets, int32_t n, void* stream);
/* same, with every offset clamped to `cap` (a record capacity chosen before the counts are known on the host):
 * consumers that index records through the offsets then stay inside buffers of `cap` records; *total (optional)
 * receives the unclamped total so the caller can detect the overflow afterwards and redo the pass. */
int tir_exclusive_scan_capped(const int32_t* counts, int32_t* offsets, int32_t n, int32_t cap,
                              int32_t* total, void* stream);

/* Record-capacity check of a captured (HIP-graph) step.  `state` (device, int64[5], zeroed by the caller, never cleared
 * by the call): running maximum of each of up to 4 device-side record counters [0..3] and a sticky overflow flag [4]
 * (set when counters[i] > caps[i]).  `host_out` (pinned host memory the device can write, int64[9]) receives this
 * launch's counters [0..3], the running maxima [4..7] and the flag [8].  `counters` / `caps` are HOST arrays. */
int tir_record_check(const int32_t* const* counters, const int64_t* caps, int32_t n, int64_t* state,
                     int64_t* host_out, void* stream);

/* Compact the samples with weight > thres into records ordered by (ray, sample) -- the order of
 * the reference's boolean-mask indexing xyz_sampled[app_mask] (:924-926).
 * rec_ray [A], rec_k [A], rec_w [A], rec_xyz [A][3] (normalised coords, :916). */
int tir_compact_primary(const TirField* f, const float* rays, const float* ray_jitter,
                        const float* weight, const int32_t* offsets, int32_t B, int32_t S,
                        int32_t* rec_ray, int32_t* rec_k, float* rec_w, float* rec_xyz,
                        void* stream);

/* ---- compositing + tone mapping of the primary pass (models/tensorBase_rotated_lights.py:973-1031).
 *      Per-record decoder outputs are summed per ray in sample order.  is_relight == 0 follows
 *      the early-return branch (:978-986).  Any per-record pointer may be NULL (treated as 0).
 *      out_maps [B][20]: rgb3 depth1 normal3 albedo3 rough1 fresnel3 acc1 ndiff1 norient1
 *                        albcost1 rghcost1 (pad1). */
#define TIR_MAP_STRIDE 20
int tir_composite_primary(const float* rays, const int32_t* offsets, const float* rec_w,
                          const float* rgb, const float* brdf, const float* brdf_jit,
                          const float* pred_normal, const float* derived_normal,
                          const float* acc, const float* depth, int32_t B, int32_t white_bg,
                          int32_t is_relight, float fixed_fresnel, float* out_maps, void* stream);

/* tir_composite_primary plus (a) the two smoothness losses = means over all B rays of map columns 17 / 18
 * (models/tensorBase_rotated_lights.py:999-1000) written to smooth_out[2] by the workgroup that finishes last
 * (`ticket`: one device int32, zero before the first use, re-arms itself; both or neither), and (b) the per-pass
 * advance of a device-side jitter-noise state rng_state = int64 {seed, offset}: offset += rng_step (NULL to skip). */
int tir_composite_primary_fused(const float* rays, const int32_t* offsets, const float* rec_w,
                                const float* rgb, const float* brdf, const float* brdf_jit,
                                const float* pred_normal, const float* derived_normal,
                                const float* acc, const float* depth, int32_t B, int32_t white_bg,
                                int32_t is_relight, float fixed_fresnel, float* out_maps, int32_t* ticket,
                                float* smooth_out, int64_t* rng_state, int64_t rng_step, void* stream);


/* ---- K7 secondary march: sample_ray_equally + cull + density + raw2alpha
 *      (models/relight_utils.py:707-722, :657-705, :777-834).
 *      Ray p starts at origins[org_map ? org_map[p] : p] along dirs[dir_map ? dir_map[p] : p];
 *      (with both maps NULL and n_dirs > 0: origin p / n_dirs, direction p % n_dirs -- the dense
 *      [surface point][direction] pair grid);  active[p] == 0 skips the ray (outputs 0).  n_sample <= 256.
 *      z_vals [n_sample] (device) are the sample distances near*(1-t)+far*t, t = linspace(0,1,n)
 *      (:716-717) -- passed in so that they are bit-identical to the caller framework's linspace.
 *      vis[p] = T_end, one_minus_acc[p] = 1 - sum w (either may be NULL).
 *      When rec_counter != NULL (TWO int32, zeroed by the caller) the samples with w > weight_thres are appended
 *      (contiguously per ray, in sample order; ray segments in arbitrary order) to rec_* (capacity rec_cap records;
 *      overflowing rays are dropped; rec_counter[0] still counts them = the capacity a re-run needs, rec_counter[1] =
 *      length of the record prefix that was actually written -- the row count consumers may process) and
 *      ray_rec_off[p] / ray_rec_cnt[p] locate ray p's segment.  stats: as tir_march_primary_fwd. */
int tir_march_secondary_fwd(const TirField* f, const float* origins, const int32_t* org_map,
                            const float* dirs, const int32_t* dir_map, const uint8_t* active,
                            int64_t n_rays, int32_t n_dirs, int32_t n_sample, const float* z_vals,
                            float t_stop, float* vis, float* one_minus_acc,
                            int32_t* rec_counter, int64_t rec_cap, int32_t* rec_ray,
                            float* rec_w, float* rec_xyz, int32_t* ray_rec_off,
                            int32_t* ray_rec_cnt, unsigned long long* stats, void* stream);

/* tir_march_secondary_fwd over a LIST of rays: ray_ids[0 .. min(n_rays, *n_ids_dev)) are pair ids (as produced by
 * tir_shade_setup_compact); origin / direction / active / every output (vis, one_minus_acc, ray_rec_off, ray_rec_cnt,
 * rec_ray entries) are addressed by the pair id, so the per-pair arrays keep their dense [M*D] shape while no half-wave
 * is spent on a masked pair.  ray_ids == n_ids_dev == NULL: identical to tir_march_secondary_fwd. */
int tir_march_secondary_ids_fwd(const TirField* f, const float* origins, const int32_t* org_map,
                                const float* dirs, const int32_t* dir_map, const uint8_t* active,
                                int64_t n_rays, int32_t n_dirs, int32_t n_sample, const float* z_vals,
                                float t_stop, float* vis, float* one_minus_acc,
                                int32_t* rec_counter, int64_t rec_cap, int32_t* rec_ray,
                                float* rec_w, float* rec_xyz, int32_t* ray_rec_off,
                                int32_t* ray_rec_cnt, unsigned long long* stats, const int32_t* ray_ids,
                                const int32_t* n_ids_dev, void* stream);

/* indirect[p] = sum over ray p's records of w * rgb  (models/relight_utils.py:832) */
int tir_accumulate_records(const int32_t* ray_rec_off, const int32_t* ray_rec_cnt,
                           const float* rec_w, const float* rec_rgb, int64_t n_rays,
                           float* indirect, void* stream);

/* ---- a15: get_light_rgbs for spherical Gaussians (models/tensorBase_rotated_lights.py:577-588,
 *      :70-86).  dirs [D][3] -> out [n_lights][D][3]. */
int tir_env_sg_fwd(const TirEnvSG* e, const float* dirs, int32_t D, float* out, void* stream);

/* ---- a15, light_kind == 'pixel' (models/tensorBase_rotated_lights.py:585-605): the environment map is a learnable image
 *      light_rgbs [H][W][3] (`_light_rgbs`, :459-460) behind softplus(beta = 5); env[l][d][:] = bilinear lookup
 *      (F.grid_sample, align_corners=False, zero padding) at the equirectangular position of dirs[d] . rot[l]
 *      (rot [n_lights][9] row-major light_rotation_matrix).  env [n_lights][n_dirs][3].
 *      softplus = 0: the image is used as it is -- light_kind == 'gt', the data set's own probe (`dataset.lights_probes`, :592-593).
 *      _bwd: g_light [H][W][3] += d loss / d light_rgbs given g_env (atomics; zero-fill first; the softplus form only:
 *      the 'gt' probe is not trained). */
int tir_env_pixel_fwd(const float* light_rgbs, int32_t H, int32_t W, const float* rot, const float* dirs,
                      int32_t n_lights, int64_t n_dirs, int32_t softplus, float* env, void* stream);
int tir_env_pixel_bwd(const float* light_rgbs, int32_t H, int32_t W, const float* rot, const float* dirs,
                      int32_t n_lights, int64_t n_dirs, const float* g_env, float* g_light, void* stream);

/* ---- geometry of render_with_BRDF (models/relight_utils.py:417-435): surface point, view
 *      vector and the cosine mask.  maps = [M][TIR_MAP_STRIDE] rows of the selected rays,
 *      rays [M][6], dirs [D][3].  surf [M][3], active [M][D] = cosine > 1e-6 and acc > acc_thres
 *      (acc_mask of :1031 / renderer.py:86; pass a large negative value to shade every row). */
int tir_shade_setup(const float* maps, const float* rays, const float* dirs, int32_t M,
                    int32_t D, float acc_thres, float* surf, uint8_t* active, void* stream);

/* tir_shade_setup that also emits the compacted list of active pairs -- the reference's boolean-mask indexing
 * surf2l[cosine_mask] (models/relight_utils.py:440-441): pair_ids[0 .. *n_active) = m * D + d of the pairs with
 * active != 0 (order of wave-sized groups arbitrary), *n_active += their number (caller zeroes it, or lets
 * tir_shade_integrate_records re-arm it); vis[pair] = 0 and ray_rec_cnt[pair] = 0 (either may be NULL) for the masked
 * pairs, which then need no secondary ray at all (tir_march_secondary_ids_fwd).  pair_order (launch option, same results):
 * 0 = default = 1 = direction-major list (a march workgroup walks a bundle of parallel rays from neighbouring points),
 * 2 = point-major. */
int tir_shade_setup_compact(const float* maps, const float* rays, const float* dirs, int32_t M,
                            int32_t D, float acc_thres, float* surf, uint8_t* active, int32_t* pair_ids,
                            int32_t* n_active, float* vis, int32_t* ray_rec_cnt, int32_t pair_order, void* stream);

/* ---- K8: GGX_specular + rendering-equation sum + tone map
 *      (models/relight_utils.py:17-50, :452-480, :489-515).
 *      vis [M][D], indirect [M][D][3] (NULL = no indirect), env [n_lights][D][3],
 *      weight_d [D] = light_area_weight (or NULL with equal_area != 0: mean * 4pi, :470-471).
 *      out_rgb [M][3]; rows with acc <= acc_thres get the white background (renderer.py:105-106). */
int tir_shade_integrate(const float* maps, const float* rays, const float* dirs,
                        const int32_t* light_idx, const float* vis, const float* indirect,
                        const float* env, const float* weight_d, int32_t M, int32_t D,
                        int32_t n_lights, int32_t equal_area, int32_t use_srgb, float acc_thres,
                        float* out_rgb, void* stream);

/* Same as tir_shade_integrate with the indirect radiance of pair (m, d) summed inside the kernel from that ray's
 * records (ray_rec_off / ray_rec_cnt of tir_march_secondary_fwd, rec_rgb [A][3] = decoder output per record), i.e.
 * tir_accumulate_records fused into the integration: the [M][D][3] indirect buffer is never written.
 * reset_counter (optional): a device int the kernel sets to 0 -- the pair counter of tir_shade_setup_compact, re-armed by
 * the last consumer of the step so that no separate fill launch is needed (also under HIP-graph replay). */
int tir_shade_integrate_records(const float* maps, const float* rays, const float* dirs,
                                const int32_t* light_idx, const float* vis, const int32_t* ray_rec_off,
                                const int32_t* ray_rec_cnt, const float* rec_w, const float* rec_rgb,
                                const float* env, const float* weight_d, int32_t M, int32_t D,
                                int32_t n_lights, int32_t equal_area, int32_t use_srgb, float acc_thres,
                                float* out_rgb, int32_t* reset_counter, void* stream);

/* ---- K9: importance-sampled HDR relighting, loop body of scripts/relight_importance.py:119-170.
 *      Per surface point m and sample s: light_dir/rgb [M][Ns][3], pdf [M][Ns], vis [M][Ns].
 *      albedo [M][3], rough [M], fresnel [M][3], normal [M][3], rays_d [M][3].  out [M][3]. */
int tir_relight_importance(const float* normal, const float* albedo, const float* rough,
                           const float* fresnel, const float* rays_d, const float* light_dir,
                           const float* light_rgb, const float* light_pdf, const float* vis,
                           int32_t M, int32_t Ns, float* out_rgb, void* stream);

/* ---- K9 on the device (models/relight_utils.py:150-205, scripts/relight_importance.py:119-171).
 * tir_env_sample_setup: Environment_Light.sample_light + the cosine mask.  For every (point m, sample s) one cell of the
 *   H x W map is drawn from pdf_sample ~ (R+G+B) sin(theta) by inverse-CDF search -- row_cdf [H] = inclusive prefix sum of
 *   the row marginals (last = 1), col_cdf [H][W] = inclusive prefix sums of every row normalised to 1 -- with Philox
 *   uniforms keyed by (seed, offset), counter = m*Ns + s (the reference: torch.multinomial on the same pdf).
 *   cell [M][Ns] = flat cell index; active [M][Ns] = (env_dir[cell] . normal[m] > 1e-6) (:125-127).
 * tir_relight_importance_cells: tir_relight_importance with light_dir / rgb / pdf looked up by cell in the map's tables
 *   env_dir [H*W][3], env_rgb [H*W][3], env_pdf [H*W] (= hdr_pdf_return).
 * tir_env_lookup: Environment_Light.get_light, bilinear background lookup (align_corners=True) at n directions. */
int tir_env_sample_setup(const float* row_cdf, const float* col_cdf, int32_t H, int32_t W, const float* env_dir,
                         const float* normal, int32_t M, int32_t Ns, uint64_t seed, uint64_t offset,
                         int32_t* cell, uint8_t* active, void* stream);
/* tir_env_sample_setup_list: the same draws and mask (same Philox counters: identical cells), plus what the visibility
 *   query of :128-160 needs -- the UNMASKED pairs only, as a list.  pair_ids [M*Ns] receives the ids m*Ns + s of the pairs
 *   that pass the cosine mask, n_active [1] (zero on entry) their count; vis [M][Ns] is set to 0 for the masked pairs (the
 *   march -- tir_march_secondary_ids_fwd with ray_ids = pair_ids, n_ids_dev = n_active -- fills the rest).  The list is
 *   grouped: blocks of `block_pairs` consecutive pairs (256 ... 32768), inside a block by direction bin (bins_r x bins_c
 *   equal cells of the map, <= 255 bins; 1 x 1 = plain compaction), so that neighbouring list entries are nearly parallel
 *   rays from neighbouring surface points.  The order of the list enters no result.
 *   row_guide [guide_rows + 1] / col_guide [H][guide_cols + 2] (both or neither; sizes powers of two; a column row holds
 *   guide_cols + 1 entries and one pad): guide[k] = the search result for u = k / G (last entry = n - 1); the inverse-CDF
 *   search then starts inside [guide[k], guide[k+1]], k = floor(u G), and returns the same cell as the full search.
 *   dir_stride: floats between the directions of consecutive cells in env_dir (3 = the [H*W][3] table; 8 = the packed records
 *   of tir_relight_importance_cells_packed, whose first three floats are the direction). */
int tir_env_sample_setup_list(const float* row_cdf, const float* col_cdf, int32_t H, int32_t W, const float* env_dir,
                                int32_t dir_stride, const float* normal, int32_t M, int32_t Ns, uint64_t seed, uint64_t offset,
                                int32_t bins_r, int32_t bins_c, int32_t block_pairs, const int32_t* row_guide,
                                const uint16_t* col_guide, int32_t guide_rows, int32_t guide_cols, int32_t* cell,
                                float* vis, int32_t* pair_ids, int32_t* n_active, void* stream);
int tir_relight_importance_cells(const float* normal, const float* albedo, const float* rough,
                                 const float* fresnel, const float* rays_d, const int32_t* cell,
                                 const float* env_dir, const float* env_rgb, const float* env_pdf,
                                 const float* vis, int32_t M, int32_t Ns, float* out_rgb, void* stream);
/* the same with the three per-cell tables interleaved: env_cell [H*W][8] = {dir.x, dir.y, dir.z, pdf_return, r, g, b, 0}
 * (16-byte aligned): one 32-byte record per sample instead of three reads at unrelated addresses; same arithmetic. */
int tir_relight_importance_cells_packed(const float* normal, const float* albedo, const float* rough,
                                        const float* fresnel, const float* rays_d, const int32_t* cell,
                                        const float* env_cell, const float* vis, int32_t M, int32_t Ns,
                                        float* out_rgb, void* stream);
int tir_env_lookup(const float* env_rgb, int32_t H, int32_t W, const float* dirs, int64_t n, float* out, void* stream);

/* A relight chunk without a host round trip (scripts/relight_importance.py:99-113 selects the acc > 0.5 rows with boolean
 * masks -- a device-to-host synchronisation per chunk -- and :166-171 puts the relit colours back with index_put_):
 * tir_surface_compact: the rows of a chunk's primary maps [B][20] (tir_composite_primary layout) with acc > acc_thres as
 *   compacted surface-point arrays in ascending row order -- surf = o + depth d (:104), normal, albedo, rough [.], fresnel,
 *   rays_d (capacity B rows each; rows >= the count are left untouched) --, slot [B] = compacted index of a row or -1, and
 *   n_hit [1] = the count, all on the device.
 * tir_env_sample_setup_list_n / tir_relight_importance_cells_packed_n: the entries above with the number of surface points
 *   read from device memory (m_dev [1], clamped to M = the arrays' capacity; NULL = M): what tir_surface_compact produced.
 *   Same Philox counters per (point, sample) as the host-compacted call, hence the same cells and colours.
 * tir_env_compose: out[i] (rows of out_stride floats) = fg_rgb[slot[i]] where slot[i] >= 0, else the background lookup of
 *   tir_env_lookup at dirs[i] (rows of dir_stride floats: 6 with dirs = rays + 3). */
int tir_surface_compact(const float* maps, const float* rays, int32_t B, float acc_thres, float* surf, float* normal,
                        float* albedo, float* rough, float* fresnel, float* rays_d, int32_t* slot, int32_t* n_hit,
                        void* stream);
int tir_env_sample_setup_list_n(const float* row_cdf, const float* col_cdf, int32_t H, int32_t W, const float* env_dir,
                                int32_t dir_stride, const float* normal, int32_t M, int32_t Ns, uint64_t seed, uint64_t offset,
                                int32_t bins_r, int32_t bins_c, int32_t block_pairs, const int32_t* row_guide,
                                const uint16_t* col_guide, int32_t guide_rows, int32_t guide_cols, int32_t* cell,
                                float* vis, int32_t* pair_ids, int32_t* n_active, const int32_t* m_dev, void* stream);
int tir_relight_importance_cells_packed_n(const float* normal, const float* albedo, const float* rough,
                                          const float* fresnel, const float* rays_d, const int32_t* cell,
                                          const float* env_cell, const float* vis, int32_t M, int32_t Ns,
                                          float* out_rgb, const int32_t* m_dev, void* stream);
int tir_env_compose(const float* env_rgb, int32_t H, int32_t W, const float* dirs, int32_t dir_stride, int64_t n,
                    const int32_t* slot, const float* fg_rgb, float* out, int32_t out_stride, void* stream);

/* GGX_specular alone (models/relight_utils.py:17-50): normal/v [M][3], l [M][D][3],
 * rough/fresnel [M][3] -> spec [M][D][3]. */
int tir_ggx_specular(const float* normal, const float* v, const float* l, const float* rough,
                     const float* fresnel, int32_t M, int32_t D, float* spec, void* stream);

/* =============================================================================================
 * Training (backward) entry points -- SURVEY.md section 8(f)-1.  The reference obtains these from
 * torch.autograd over its op chain (train_tensoIR.py:315-317 `total_loss.backward()`); here every
 * chain has a hand-written backward kernel.  Gradient buffers are caller-allocated and ACCUMULATED
 * into (zero them first); field gradients are in the packed channel-last layout of TirField.
 * Scatter-adds use hardware fp32 atomics (like ATen's grid_sampler backward), so the summation order --
 * not the set of addends -- varies from run to run.
 * ============================================================================================= */
typedef struct TirFieldGrad {
    float* dplane[3];        /* [H_i][W_i][n_dcomp]   d loss / d density_plane (channel-last)       */
    float* dline[3];         /* [R_i][n_dcomp]                                                      */
    float* aplane[3];        /* [H_i][W_i][n_acomp]                                                 */
    float* aline[3];         /* [R_i][n_acomp]                                                      */
    float* light_line;       /* [n_lights][3*n_acomp]  d loss / d light_line.weight                 */
    float* light_mean;       /* [3*n_acomp]  d loss / d mean-over-lights row (caller adds /L to every light) */
} TirFieldGrad;

/* tir_march_primary_fwd that additionally stores the per-sample density sigma [B][S] (0 for culled
 * samples) -- the only extra state the backward needs. */
int tir_march_primary_train_fwd(const TirField* f, const float* rays, const float* ray_jitter,
                                int32_t B, int32_t S, float t_stop, float* weight, float* sigma,
                                float* acc, float* depth, float* t_end, int32_t* app_count, void* stream);

/* Backward of tir_composite_primary (models/tensorBase_rotated_lights.py:973-1031).
 * g_maps [B][20]: d loss / d out_maps.  Outputs (written, not accumulated): per-record gradients g_rgb [A][3],
 * g_brdf [A][4] (w.r.t. the raw sigmoid outputs), g_brdf_jit [A][4], g_pred [A][3], g_der [A][3] (NULL where the
 * forward input was NULL); g_weight [B][S] dense, MUST be zero-filled by the caller -- receives the record
 * part of d loss / d weight at (ray, rec_k); g_acc [B], g_depth [B]: gradients of the dense sums
 * acc = sum w, depth = sum w z. */
int tir_composite_primary_bwd(const float* rays, const int32_t* offsets, const int32_t* rec_k,
                              const float* rec_w, const float* rgb, const float* brdf,
                              const float* brdf_jit, const float* pred_normal,
                              const float* derived_normal, const float* acc, const float* depth,
                              int32_t B, int32_t S, int32_t white_bg, int32_t is_relight,
                              float fixed_fresnel, const float* g_maps, float* g_rgb, float* g_brdf,
                              float* g_brdf_jit, float* g_pred, float* g_der, float* g_weight,
                              float* g_acc, float* g_depth, void* stream);

/* Backward of raw2alpha + feature2density + compute_densityfeature along the primary rays
 * (models/tensorBase_rotated_lights.py:21-28, :813-817; models/tensoRF_rotated_lights.py:95-110):
 * d loss / d weight[b][k] = g_weight[b][k] + g_acc[b] + z_k g_depth[b]  ->  d alpha -> d sigma -> d feature,
 * scatter-added into g->dplane / g->dline.  g_feature (optional, NULL to skip): [B][S] receives d loss / d
 * density feature per sample (what is scattered). */
int tir_march_primary_bwd(const TirField* f, const TirFieldGrad* g, const float* rays,
                          const float* ray_jitter, int32_t B, int32_t S, const float* sigma,
                          const float* weight, const float* g_weight, const float* g_acc,
                          const float* g_depth, float* g_feature, void* stream);

/* Backward of tir_density_grad_fwd's derived normal w.r.t. the density planes/lines (the second-order
 * path of compute_derived_normals, models/tensorBase_rotated_lights.py:839-856: create_graph=True). */
int tir_density_grad_bwd(const TirField* f, const TirFieldGrad* g, const float* xyz,
                         const float* g_normal, int64_t n, void* stream);

/* Backward of tir_vm_app_fwd.  g_rad / g_int [n][stride] (either may be NULL).  Accumulates into g->aplane,
 * g->aline, g->light_line, g->light_mean; writes y_rad / y_int [n][3*n_acomp] = (plane*line) (.) light row,
 * the left operand of d basis_mat = g_feat^T y (tir_gemm_tn).  y_rad (y_int) is REQUIRED whenever g_rad (g_int) is
 * given: the buffer first holds dY = g_feat . basis_mat (an MFMA GEMM) and is then overwritten in place with y.
 * Samples should arrive in (ray, sample-along-ray) order -- tir_compact_primary's order: consecutive samples that
 * share a plane cell are summed in registers before they are scattered. */
int tir_vm_app_bwd(const TirField* f, const TirFieldGrad* g, const float* xyz,
                   const int32_t* light_idx, const int32_t* idx_map, const float* g_rad,
                   const float* g_int, int32_t stride, int64_t n, float* y_rad, float* y_int,
                   void* stream);

/* Decoder forward that also stores the post-ReLU hidden activations h1, h2 [n][hidden] (exact fp32 MFMA). */
int tir_mlp_train_fwd(const TirMlp* m, const float* feat, int32_t feat_stride, const float* aux,
                      const int32_t* aux_map, int32_t aux_mod, float* out, float* h1, float* h2,
                      int64_t n, const int32_t* n_dev, void* stream);
/* same on the split-bf16 matrix-core kernel (the product's default decoder precision; ~4x faster than exact fp32) */
int tir_mlp_train_fwd_bf16x3(const TirMlp* m, const float* feat, int32_t feat_stride, const float* aux,
                             const int32_t* aux_map, int32_t aux_mod, float* out, float* h1, float* h2,
                             int64_t n, const int32_t* n_dev, void* stream);
/* decoder input rows [n][160] = [feat, aux, PE(feat), PE(aux), 0-pad]  (right operand of d W0) */
int tir_mlp_inputs(const TirMlp* m, const float* feat, int32_t feat_stride, const float* aux,
                   const int32_t* aux_map, int32_t aux_mod, float* x, int64_t n, void* stream);
int64_t tir_mlp_bwd_packed_floats(int32_t feat_dim, int32_t pe, int32_t hidden, int32_t out_dim);
int tir_pack_mlp_bwd(const float* w0, const float* w1, const float* w2, int32_t feat_dim, int32_t pe,
                     int32_t hidden, int32_t out_dim, float* packed, void* stream);
/* Backward-data of one decoder: out / g_out [n][out_dim] (post-activation values and their gradients) ->
 * g_feat [n][32] (through the positional encoding; columns >= feat_dim are 0), and the pre-activation
 * gradients dz1, dz2 [n][hidden], dz3 [n][4] whose products with (x, h1, h2) are the weight gradients
 * (tir_gemm_tn).  `packed_bwd` from tir_pack_mlp_bwd. */
int tir_mlp_bwd(const TirMlp* m, const float* packed_bwd, const float* feat, int32_t feat_stride,
                const float* out, const float* g_out, const float* h1, const float* h2, int64_t n,
                float* g_feat, float* dz1, float* dz2, float* dz3, void* stream);
/* Same on the bf16 matrix pipe with split operands (3 products, fp32 accumulation; the blob of tir_pack_mlp_bwd
 * carries both images). */
int tir_mlp_bwd_bf16x3(const TirMlp* m, const float* packed_bwd, const float* feat, int32_t feat_stride,
                const float* out, const float* g_out, const float* h1, const float* h2, int64_t n,
                float* g_feat, float* dz1, float* dz2, float* dz3, void* stream);
/* tir_mlp_bwd_bf16x3 for up to four decoder invocations over the same n rows in one launch (the grid is split between
 * them; host arrays of n_jobs entries, arguments as in the single call). */
int tir_mlp_bwd_multi_bf16x3(const TirMlp* const* mlps, const float* const* packed_bwds, const float* const* feats,
                             int32_t feat_stride, const float* const* outs, const float* const* g_outs,
                             const float* const* h1s, const float* const* h2s, int32_t n_jobs, int64_t n,
                             float* const* g_feats, float* const* dz1s, float* const* dz2s, float* const* dz3s, void* stream);

/* C[M][ldc] += sum_j A_j^T B_j for up to three operand pairs over the same n rows, M <= 32, N <= 160: the gradient of
 * `basis_mat` (nn.Linear(144 -> 27, bias=False), models/tensoRF_rotated_lights.py:17) = g_feat^T y summed over the stage's
 * appearance gathers, in one launch.  A_j [n][lda] (first M columns), B_j [n][ldb] (first N columns).  Split-bf16 matrix
 * cores, fp32 accumulation; results are ADDED to C (atomics). */
int tir_gemm_tn_small_bf16x3(const float* const* As, int32_t lda, int32_t M, const float* const* Bs, int32_t ldb,
                             int32_t N, int32_t n_jobs, int64_t n, float* C, int32_t ldc, void* stream);

/* Weight gradients of up to four decoder invocations over the same n rows in ONE launch -- the `loss.backward()` leaves of
 * nn.Linear in MLPRender_Fea / MLPBRDF_PEandFeature (models/tensorBase_rotated_lights.py:122-146, :182-208;
 * train_tensoIR.py:315):   dW0 [128][150] += dz1^T x,  dW1 [128][128] += dz2^T h1,  dW2 [4][128] += dz3^T h2,
 * db0 [128] / db1 [128] / db2 [4] += the column sums of dz1 / dz2 / dz3.  dz1, dz2 [n][128], dz3 [n][4] from
 * tir_mlp_bwd*; h1, h2 [n][128] from the training forward; x is rebuilt in registers from feat [n][feat_stride] and
 * aux [.][3] (aux_maps[i] may be NULL: row s uses aux row s), so no input-row buffer exists.  Split-bf16 matrix cores,
 * fp32 accumulation; results are ADDED (atomics): zero-fill the outputs first.  Two jobs may share outputs (the BRDF
 * decoder's two invocations).  Host arrays of n_jobs entries.  max_workgroups (launch option, 0 = 256 = one per CU): a
 * workgroup occupies a whole CU, so a caller that runs this beside other kernels passes fewer. */
int tir_mlp_wgrad_multi(const float* const* dz1s, const float* const* dz2s, const float* const* dz3s,
                        const float* const* h1s, const float* const* h2s, const float* const* feats,
                        int32_t feat_stride, const float* const* auxs, const int32_t* const* aux_maps,
                        float* const* dW0s, float* const* db0s, float* const* dW1s, float* const* db1s,
                        float* const* dW2s, float* const* db2s, int32_t n_jobs, int64_t n, int32_t max_workgroups,
                        void* stream);

/* C[M][ldc] += A^T B (+ column N = A^T 1 when ones_col != 0: the bias gradient); A [n][lda] (first M columns),
 * B [n][ldb] (first N columns); M <= 128, N + ones_col <= 160.  fp32 MFMA, split over n.  bias_out (may be NULL):
 * A^T 1 is added to bias_out[M] instead of column N of C, so that C can be the exact [M][N] weight gradient. */
int tir_gemm_tn(const float* A, int32_t lda, int32_t M, const float* B, int32_t ldb, int32_t N,
                int32_t ones_col, int64_t n, float* C, int32_t ldc, float* bias_out, void* stream);
/* Same product on the bf16 matrix pipe with every operand split x = hi + lo (3 products, fp32 accumulation:
 * ~2^-16 relative per product, the decoders' split-bf16 scheme). */
int tir_gemm_tn_bf16x3(const float* A, int32_t lda, int32_t M, const float* B, int32_t ldb, int32_t N,
                int32_t ones_col, int64_t n, float* C, int32_t ldc, float* bias_out, void* stream);

/* optimizer.step() of the training loop (train_tensoIR.py:317; torch.optim.Adam(grad_vars, betas=(0.9, 0.99)), :197) for a
 * list of tensors in one launch: torch's default Adam update (no amsgrad, no weight decay), element for element,
 *   m += (g - m)(1 - beta1);  v = v beta2 + (1 - beta2) g g;  p -= lr / bc1 * m / (sqrt(v) / sqrt(bc2) + eps).
 * Host arrays of n_tensors entries: device pointers p / g / m / v (dense storage in the same element order), element
 * counts, and per tensor lr, bias_correction1 = 1 - beta1^step, bias_correction2 = 1 - beta2^step. */
int tir_adam_step(int32_t n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v,
                  const int64_t* count, const float* lr, const float* bias_correction1,
                  const float* bias_correction2, float beta1, float beta2, float eps, void* stream);

/* Backward of tir_shade_integrate w.r.t. the map rows (normal 4:7, albedo 7:10, roughness 10, fresnel 11:14)
 * and the environment radiance.  g_out [M][3] -> g_maps [M][20] (written), g_env [n_lights][D][3] (accumulated).
 * Visibility and indirect light are constants (compute_secondary_shading_effects is @torch.no_grad,
 * models/relight_utils.py:344). */
int tir_shade_integrate_bwd(const float* maps, const float* rays, const float* dirs,
                            const int32_t* light_idx, const float* vis, const float* indirect,
                            const float* env, const float* weight_d, int32_t M, int32_t D,
                            int32_t n_lights, int32_t equal_area, int32_t use_srgb, float acc_thres,
                            const float* g_out, float* g_maps, float* g_env, void* stream);

/* Backward of tir_env_sg_fwd: g_env [n_lights][D][3] -> g_sgs [n_sg][7] (accumulated). */
int tir_env_sg_bwd(const TirEnvSG* e, const float* dirs, int32_t D, const float* g_env, float* g_sgs,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TENSOIR_HIP_H */
