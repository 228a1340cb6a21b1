/*
 * ngp_hip.h -- C ABI of libngp_hip.so, the MI355X (gfx950) implementation of raw_ngp's
 * data-parallel hot path: multiresolution hash-grid encoder, spherical-harmonics
 * encoder, density-grid ray marcher / volumetric compositor (+ the fused extensions
 * the trainer uses).
 *
 * This is the drop-in boundary: every `ngp_<name>` below replaces the function of the
 * same name that the reference exports from its pybind11 modules `_gridencoder`,
 * `_shencoder`, `_raymarching_mob`, `_freqencoder` (declarations cited per function).
 * The argument ORDER and MEANING are the reference's; at::Tensor arguments become raw
 * device pointers, at::optional<Tensor> becomes a nullable pointer, and one trailing
 * `stream` (a hipStream_t, NULL = default stream) is appended.
 *
 * Conventions
 *   - all pointers are DEVICE pointers to contiguous buffers the CALLER allocated;
 *     float = IEEE binary32, index arrays int32, bitfields uint8
 *   - functions only enqueue work on `stream`; they never allocate, never synchronise,
 *     keep no pointer after returning and are safe to capture in a hipGraph
 *   - return 0 on success, a negative NGP_E* code otherwise; ngp_last_error() gives the
 *     message of the calling thread's last failure (the Python shim raises RuntimeError,
 *     which is what the reference's TORCH_CHECK / std::runtime_error surface as)
 *   - outputs the reference requires the caller to zero-initialise stay the caller's
 *     job (grad_embeddings, grad_inputs, weights, xyzs/dirs/ts of march_rays, outputs /
 *     dy_dx when max_level < L)
 */
#ifndef NGP_HIP_H
#define NGP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NGP_ABI_VERSION 1

#define NGP_OK 0
#define NGP_EINVAL (-1)  /* unsupported D / C / degree, null pointer, bad size */
#define NGP_ELAUNCH (-2) /* hipGetLastError() after a launch */

typedef void *ngp_stream_t; /* hipStream_t */

int ngp_abi_version(void);
const char *ngp_last_error(void);

/* ------------------------------------------------------------------------------------
 * grid encoder -- replaces gridencoder/src/gridencoder.h:12-16 (bindings.cpp:5-9)
 *   inputs      [B, D]  in [0,1]            embeddings [offsets[L], C]
 *   offsets     [L+1]   int32 (device)      outputs    [L, B, C]
 *   dy_dx       [B, L*D*C] or NULL          D in {2,3,4,5}, C in {1,2,4,8,16,32}
 *   gridtype 0 = hash, 1 = tiled; interp 0 = linear, 1 = smoothstep
 * ---------------------------------------------------------------------------------- */
int ngp_grid_encode_forward(const float *inputs, const float *embeddings, const int32_t *offsets,
                            float *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                            uint32_t max_level, float S, uint32_t H, float *dy_dx, uint32_t gridtype,
                            int align_corners, uint32_t interp, ngp_stream_t stream);

/* grad [L, B, C]; grad_embeddings (pre-zeroed) += scatter; grad_inputs [B, D] written when
 * dy_dx != NULL */
int ngp_grid_encode_backward(const float *grad, const float *inputs, const float *embeddings,
                             const int32_t *offsets, float *grad_embeddings, uint32_t B, uint32_t D,
                             uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H,
                             const float *dy_dx, float *grad_inputs, uint32_t gridtype,
                             int align_corners, uint32_t interp, ngp_stream_t stream);

int ngp_grad_total_variation(const float *inputs, const float *embeddings, float *grad,
                             const int32_t *offsets, float weight, uint32_t B, uint32_t D, uint32_t C,
                             uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                             ngp_stream_t stream);

int ngp_grad_weight_decay(const float *embeddings, float *grad, const int32_t *offsets, float weight,
                          uint32_t B, uint32_t C, uint32_t L, ngp_stream_t stream);

/* ------------------------------------------------------------------------------------
 * SH encoder -- replaces shencoder/src/shencoder.h:8-9.  D must be 3, 1 <= degree <= 8.
 *   inputs [B,3] (unit vectors), outputs [B, degree^2], dy_dx [B, 3*degree^2] or NULL
 * ---------------------------------------------------------------------------------- */
int ngp_sh_encode_forward(const float *inputs, float *outputs, uint32_t B, uint32_t D, uint32_t degree,
                          float *dy_dx, ngp_stream_t stream);
/* grad_inputs [B,3] += J^T grad (caller zero-initialises, sphere_harmonics.py:50) */
int ngp_sh_encode_backward(const float *grad, const float *inputs, uint32_t B, uint32_t D,
                           uint32_t degree, const float *dy_dx, float *grad_inputs, ngp_stream_t stream);

/* ------------------------------------------------------------------------------------
 * frequency encoder -- replaces freqencoder/src/freqencoder.h (freq_encode_forward/backward)
 *   outputs [B, C], C = D + 2*D*deg
 * ---------------------------------------------------------------------------------- */
int ngp_freq_encode_forward(const float *inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                            float *outputs, ngp_stream_t stream);
int ngp_freq_encode_backward(const float *grad, const float *outputs, uint32_t B, uint32_t D, uint32_t deg,
                             uint32_t C, float *grad_inputs, ngp_stream_t stream);

/* ------------------------------------------------------------------------------------
 * ray marching -- replaces raymarching/src/raymarching.h:7-19
 * ---------------------------------------------------------------------------------- */
int ngp_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N,
                           float min_near, float *nears, float *fars, ngp_stream_t stream);
int ngp_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords,
                     ngp_stream_t stream);
int ngp_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, ngp_stream_t stream);
int ngp_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, ngp_stream_t stream);
/* grid [N*8] floats -> bitfield [N] bytes, bit i = grid[8n+i] > density_thresh */
int ngp_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield,
                 ngp_stream_t stream);
int ngp_flatten_rays(const int32_t *rays, uint32_t N, uint32_t M, int32_t *res, ngp_stream_t stream);

/* Two-call protocol of raymarching.py:301-311.
 *   call 1 (xyzs == NULL): rays[n] = (offset, count), counter[0] += total.  Offsets are the
 *     exclusive prefix sum of the counts in ray order starting at the incoming counter[0]
 *     (the reference's atomicAdd hands them out in scheduling order; ray order is the
 *     reproducible member of that family and the one raymarching.py:325-328 assumes).
 *   call 2 (xyzs != NULL): writes xyzs/dirs [M,3], ts [M,2] (, ldirs [M,3]) for those rays.
 * rays_ldir / ldirs may be NULL.  C = cascades, H = grid size. */
int ngp_march_rays_train(const float *rays_o, const float *rays_d, const float *rays_ldir,
                         const uint8_t *grid, float bound, int contract, float dt_gamma,
                         uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, const float *nears,
                         const float *fars, float *xyzs, float *dirs, float *ts, float *ldirs,
                         int32_t *rays, int32_t *counter, const float *noises, ngp_stream_t stream);

int ngp_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *ts,
                                     const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                                     float *weights, float *weights_sum, float *depth, float *image,
                                     ngp_stream_t stream);

int ngp_composite_rays_train_backward(const float *grad_weights, const float *grad_weights_sum,
                                      const float *grad_depth, const float *grad_image,
                                      const float *sigmas, const float *rgbs, const float *ts,
                                      const int32_t *rays, const float *weights_sum, const float *depth,
                                      const float *image, uint32_t M, uint32_t N, float T_thresh,
                                      float *grad_sigmas, float *grad_rgbs, ngp_stream_t stream);

int ngp_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                   const float *rays_o, const float *rays_d, float bound, int contract, float dt_gamma,
                   uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid, const float *nears,
                   const float *fars, float *xyzs, float *dirs, float *ts, const float *noises,
                   ngp_stream_t stream);

int ngp_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive,
                       float *rays_t, const float *sigmas, const float *rgbs, const float *ts,
                       float *weights_sum, float *depth, float *image, ngp_stream_t stream);

/* ====================================================================================
 * Extensions (ngp_x_*): no counterpart in the reference's bindings; they fuse steps the
 * reference does in Python/torch between two `_backend` calls so that a training step
 * needs no host synchronisation and no layout copies.  Each names what it replaces.
 * ================================================================================== */

/* Replaces torch_scatter.segment_csr in _march_rays_train.backward (raymarching.py:319-329):
 *   grad_rays_o[n] = sum_i grad_xyzs[i];  grad_rays_d[n] = sum_i grad_xyzs[i]*ts[i,0] + grad_dirs[i]
 * over the samples of ray n; grad_dirs may be NULL. */
int ngp_x_march_rays_train_backward(const float *grad_xyzs, const float *grad_dirs, const float *ts,
                                    const int32_t *rays, uint32_t N, uint32_t M, float *grad_rays_o,
                                    float *grad_rays_d, ngp_stream_t stream);

/* One-call training march into a pre-sized arena (replaces raymarching.py:292-311 including the
 * `.item()` host sync): counts, ray-ordered prefix sum, and a sample-parallel coalesced write.
 *   t_scratch   [N * max_steps] floats of scratch (sample start times, ray-major)
 *   M_cap       capacity of xyzs/dirs/ts/ldirs in samples; rays whose samples would not fit get
 *               count 0 (dropped like the reference's overflow rule raymarching.cu:540)
 *   counter     [2] int32: counter[0] = M actually written (overwritten, not accumulated),
 *               counter[1] = M the batch would have needed
 *   ray_idx     [M_cap] int32 or NULL: sample -> ray id (what flatten_rays would give) */
int ngp_x_march_rays_train_arena(const float *rays_o, const float *rays_d, const float *rays_ldir,
                                 const uint8_t *grid, float bound, int contract, float dt_gamma,
                                 uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                 const float *nears, const float *fars, const float *noises,
                                 float *t_scratch, uint32_t M_cap, float *xyzs, float *dirs, float *ts,
                                 float *ldirs, int32_t *rays, int32_t *counter, int32_t *ray_idx,
                                 const uint32_t *occ_index, float *chain, uint16_t *chain_code, int32_t *chain_len,
                                 uint32_t chain_cap, ngp_stream_t stream);
/* chain != NULL selects the chain-parallel first pass (same result, no serial per-ray loop over the grid): chain
 * (chain_cap * N f32) and chain_code (chain_cap * N u16) are scratch (stored ray-major), chain_len [N] i32; chain_cap must cover
 * (far - near) / dt_min + 1 candidate parameters per ray (max_steps * ceil(bound) + 2 always does).  counter then
 * needs 4 ints: counter[2] becomes non-zero (and stays so) if a ray's chain did not fit.
 * With dt_gamma == 0 (constant step) the chain is not stored at all: the candidate parameters have a closed form per
 * binade and ONE kernel computes, classifies and walks them (raymarching.hip: march_const_step_kernel; same samples bit
 * for bit; the chain buffers stay untouched, chain_cap still bounds the candidates per ray, near must be >= 0 -- a
 * negative parameter is reported through counter[2]).  occ_index (ngp_x_build_occupancy_index) then lets that kernel
 * probe an LDS copy of the bitfield; NULL: probes go to `grid`. */

/* The same march in two halves (chain-parallel variant only, chain != NULL).  stage 0: everything (= the call above);
 * 1: the first kernel alone -- every ray's candidate parameters, from rays, near/far and noise: it does not read the
 * occupancy grid; 2: everything after it.  A caller whose grid is still being rebuilt (density-grid refresh) can run stage 1
 * early and stage 2 once the grid is final; together they are stage 0 bit for bit.  (dt_gamma == 0: stage 1 has nothing to
 * do, stage 2 is the whole march.) */
int ngp_x_march_rays_train_arena_stage(const float *rays_o, const float *rays_d, const float *rays_ldir,
                                       const uint8_t *grid, float bound, int contract, float dt_gamma,
                                       uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                       const float *nears, const float *fars, const float *noises,
                                       float *t_scratch, uint32_t M_cap, float *xyzs, float *dirs, float *ts,
                                       float *ldirs, int32_t *rays, int32_t *counter, int32_t *ray_idx,
                                       const uint32_t *occ_index, float *chain, uint16_t *chain_code, int32_t *chain_len,
                                       uint32_t chain_cap, int stage, ngp_stream_t stream);

/* Compressed copy of the occupancy bitfield that the arena march can keep in LDS (occ_index above; NULL = probe
 * the bitfield in global memory).  The bitfield is Morton-ordered (raymarching.cu:56-81), so 64 consecutive bits
 * are one 4x4x4 block; the index stores which blocks are non-zero plus the non-zero blocks themselves.  Rebuild it
 * after every ngp_packbits.  Needs C*H^3 % 2048 == 0; index buffer of ngp_x_occupancy_index_bytes(C, H) bytes,
 * 8-byte aligned.  The march result does not depend on whether the index is used. */
size_t ngp_x_occupancy_index_bytes(uint32_t C, uint32_t H);
int ngp_x_build_occupancy_index(const uint8_t *grid, uint32_t C, uint32_t H, uint32_t *index, ngp_stream_t stream);

/* Hash-grid table gradient without global float atomics (D = 3, C = 2 only): same result as
 * ngp_grid_encode_backward's scatter (grad_embeddings += ...; different summation order), computed as
 * bin -> LDS reduce (csrc/grid_backward_binned.hip).  Needs caller-owned scratch:
 *   workspace_bytes >= ngp_x_grid_backward_workspace_bytes(B, L, n_rows_total), 16-byte aligned,
 *   n_rows_total = rows of `embeddings` (= offsets[L], known to the host from the tensor shape);
 *   max_level_rows = rows of the largest level (max_l offsets[l+1] - offsets[l]) or 0 if the host does
 *   not know it (it sizes the per-workgroup LDS histograms; a correct value is faster, never required...
 *   but a value SMALLER than the truth is an error the kernels cannot detect).
 * grad is [L][grad_stride][2] (grad_stride = B for the reference layout); B_dev: optional device int32 with
 * the number of live points (clamped to B; launch geometry and workspace are sized by B).
 * Does not compute grad_inputs (ngp_x_grid_input_backward does).
 * Two record layouts exist behind the same calls.  Tile-local (the default; up to 2048 x 512 samples and 128 chunks per
 * level): every fill workgroup writes its records, sorted by chunk, into a region of its own and leaves a directory; nothing
 * is counted, scanned or reserved across workgroups.  Global bins (NGP_BINNED_LOCAL=0, or beyond those limits): one record
 * stream per chunk, sized by a counting pass + scan, filled through per-chunk cursors.  ngp_x_grid_backward_binned_counts
 * tells which one a call of this shape takes: 1 = the apply half needs the counts (prepare stage 0, or stage 1 + a
 * counting forward + stage 2), 0 = it only needs the header reset of prepare stage 1 (counting and scanning are wasted). */
size_t ngp_x_grid_backward_workspace_bytes(uint32_t B, uint32_t L, uint32_t n_rows_total);
int ngp_x_grid_backward_binned_counts(uint32_t B, uint32_t L, uint32_t n_rows_total, uint32_t max_level_rows);
/* the workspace's geometry, for tools and tests that look inside it (host call, no GPU): out[0] = table rows per chunk,
 * out[1] = samples per fill tile, out[2] = record slots per (level, tile) region of the tile-local layout, out[3] = tiles
 * the tile-local reduce indexes per batch */
int ngp_x_grid_backward_binned_geometry(uint32_t *out);
int ngp_x_grid_encode_backward_binned(const float *grad, const float *inputs, const int32_t *offsets,
                                      float *grad_embeddings, const int32_t *B_dev, uint32_t B,
                                      uint32_t grad_stride, uint32_t L, uint32_t max_level,
                                      float S, uint32_t H, uint32_t gridtype, int align_corners,
                                      uint32_t interp, uint32_t n_rows_total, uint32_t max_level_rows,
                                      void *workspace, size_t workspace_bytes, ngp_stream_t stream);

/* The same call in two halves, so the positions-only half can run early (e.g. on another stream, while the
 * forward pass is still running): `prepare` = plan + count + scan, reads only the sample positions -- in [0,1] when
 * in_bound = 0, world positions in [-in_bound, in_bound] (mapped like gridencoder/grid.py:161) otherwise;
 * `apply` = fill + reduce on the prepared workspace with the [0,1] inputs and the same B_dev value. */
int ngp_x_grid_backward_binned_prepare(const float *inputs, float in_bound, const int32_t *offsets, const int32_t *B_dev,
                                       uint32_t B, uint32_t L, uint32_t max_level, float S, uint32_t H,
                                       uint32_t gridtype, int align_corners, uint32_t interp, uint32_t n_rows_total,
                                       uint32_t max_level_rows, int single_segment, uint32_t merge_max_res, int stage,
                                       void *workspace, size_t workspace_bytes, ngp_stream_t stream);
/* stage: 0 = plan + count + scan; 1 = plan only, 2 = scan only -- for callers that let
 * ngp_x_grid_encode_forward_slab(..., binned_workspace) count the records between the two (it has every corner's row
 * in registers and idle issue slots: the backward then needs no counting pass). */
/* merge_max_res: finest level resolution at which runs of same-cell samples are merged before binning (0 = 1024,
 * the limit of the cell key); choose ~0.7 / (sample spacing in [0,1] units) for ray-ordered samples. */
int ngp_x_grid_backward_binned_apply(const float *grad, const float *inputs, const int32_t *offsets,
                                     float *grad_embeddings, const int32_t *B_dev, uint32_t B, uint32_t grad_stride,
                                     uint32_t L, uint32_t max_level, float S, uint32_t H, uint32_t gridtype,
                                     int align_corners, uint32_t interp, uint32_t n_rows_total, uint32_t max_level_rows,
                                     void *workspace, size_t workspace_bytes, float *adam_param, float *adam_exp_avg,
                                     float *adam_exp_avg_sq, const float *adam_hyper, float beta1, float beta2,
                                     float eps, int overwrite, float *loss_scaler, ngp_stream_t stream);
/* ngp_x_grid_backward_binned_apply over a LIST of samples (see ngp_x_grid_backward_binned_apply_mlp_list) */
int ngp_x_grid_backward_binned_apply_list(const float *grad, const float *inputs, const int32_t *sample_index,
                                          const int32_t *offsets, float *grad_embeddings, const int32_t *B_dev, uint32_t B,
                                          uint32_t grad_stride, uint32_t L, uint32_t max_level, float S, uint32_t H,
                                          uint32_t gridtype, int align_corners, uint32_t interp, uint32_t n_rows_total,
                                          uint32_t max_level_rows, void *workspace, size_t workspace_bytes,
                                          float *adam_param, float *adam_exp_avg, float *adam_exp_avg_sq,
                                          const float *adam_hyper, float beta1, float beta2, float eps, int overwrite,
                                          float *loss_scaler, ngp_stream_t stream);
/* ngp_x_grid_backward_binned_apply with ngp_x_mlp_reduce_dw riding along (its arguments, mlp_ prefix, same meaning and
 * checks): the weight-gradient reduction of the fused MLP runs as extra workgroups of the fill kernel instead of as a
 * kernel of its own -- one launch and one dependent-launch gap fewer on the fused step's critical path.  Nothing in the
 * table backward depends on it or vice versa. */
int ngp_x_grid_backward_binned_apply_mlp(
    const float *grad, const float *inputs, const int32_t *offsets, float *grad_embeddings, const int32_t *B_dev, uint32_t B,
    uint32_t grad_stride, uint32_t L, uint32_t max_level, float S, uint32_t H, uint32_t gridtype, int align_corners,
    uint32_t interp, uint32_t n_rows_total, uint32_t max_level_rows, void *workspace, size_t workspace_bytes, float *adam_param,
    float *adam_exp_avg, float *adam_exp_avg_sq, const float *adam_hyper, float beta1, float beta2, float eps, int overwrite,
    uint32_t mlp_M, float mlp_loss_scale, float *dw1, float *dw2, float *dw3, float *dw4, float *dw5, float *dw6,
    const void *mlp_workspace, size_t mlp_workspace_bytes, float *mlp_adam_param, const float *mlp_adam_grad,
    float *mlp_adam_exp_avg, float *mlp_adam_exp_avg_sq, uint32_t mlp_adam_n, const float *mlp_adam_hyper, float mlp_beta1,
    float mlp_beta2, float mlp_eps, void *mlp_adam_image, float *loss_scaler, ngp_stream_t stream);
/* ngp_x_grid_backward_binned_apply_mlp over a LIST of samples (tile-local layout only): entry b of the call is sample
 * sample_index[b] -- `inputs` is addressed by sample, the gradient slab `grad` is in list order (as ngp_x_mlp_backward_list
 * writes it), *B_dev entries are used.  NULL: samples 0 .. B - 1. */
int ngp_x_grid_backward_binned_apply_mlp_list(
    const float *grad, const float *inputs, const int32_t *sample_index, const int32_t *offsets, float *grad_embeddings,
    const int32_t *B_dev, uint32_t B, uint32_t grad_stride, uint32_t L, uint32_t max_level, float S, uint32_t H,
    uint32_t gridtype, int align_corners, uint32_t interp, uint32_t n_rows_total, uint32_t max_level_rows, void *workspace,
    size_t workspace_bytes, float *adam_param, float *adam_exp_avg, float *adam_exp_avg_sq, const float *adam_hyper,
    float beta1, float beta2, float eps, int overwrite, uint32_t mlp_M, float mlp_loss_scale, float *dw1, float *dw2,
    float *dw3, float *dw4, float *dw5, float *dw6, const void *mlp_workspace, size_t mlp_workspace_bytes,
    float *mlp_adam_param, const float *mlp_adam_grad, float *mlp_adam_exp_avg, float *mlp_adam_exp_avg_sq,
    uint32_t mlp_adam_n, const float *mlp_adam_hyper, float mlp_beta1, float mlp_beta2, float mlp_eps, void *mlp_adam_image,
    float *loss_scaler, ngp_stream_t stream);
/* grad_embeddings == NULL without adam_param: FILL ONLY (records + directory, and the _mlp variants' passengers); the caller
 * reduces afterwards, in chunk ranges of its own: */
int ngp_x_grid_backward_binned_reduce_range(const int32_t *offsets, float *grad_embeddings, const int32_t *B_dev, uint32_t B,
                                            uint32_t L, float S, uint32_t H, uint32_t n_rows_total, uint32_t max_level_rows,
                                            void *workspace, size_t workspace_bytes, int overwrite, uint32_t chunk_lo,
                                            uint32_t chunk_hi, float *loss_scaler, ngp_stream_t stream);
/* ... the reduce half over the chunks [chunk_lo, chunk_hi) (level-major numbering: ceil(rows of level / chunk rows) chunks
 * per level, ngp_x_grid_backward_binned_geometry) of a workspace a fill-only call has filled -- tile-local layout only.  The
 * data-parallel step reduces the levels in two ranges so that the exchange of the first overlaps the second's reduction. */
/* overwrite != 0 (workspace prepared with single_segment, max_level == L): grad_embeddings = sums for EVERY row of every
 * level (zeros where nothing landed) instead of +=, so the caller neither zeroes the gradient nor pays its read.
 * overwrite == 2: the same, stored as bfloat16 (round to nearest even) -- grad_embeddings then points to
 * n_rows_total * C 16-bit values, the wire format of the data-parallel gradient exchange. */
/* loss_scaler (NULL: none): the dynamic loss scale of the fused MLP backward (see "Dynamic loss scale" below).  The reduce
 * launch settles the step: a non-finite feature gradient anywhere in the batch raises the scaler's overflow word; with the
 * fused Adam the table is left alone when that word is set (by this launch or by the weight-gradient reduction); in the
 * _mlp variants with mlp_adam_param the MLP weights' Adam step (and their operand-image entries) moves from the fill
 * launch's passengers to passengers of the reduce launch, so that it falls under the same verdict. */
/* adam_param != NULL (single GPU, no weight decay / TV on the table): the gradient of a chunk never leaves LDS -- the
 * reduce kernel applies torch.optim.Adam to the chunk's rows of `adam_param` directly (hyper as in
 * ngp_x_adam_step_dev) and grad_embeddings is neither read nor written.  Requires a workspace prepared with
 * single_segment != 0 (one workgroup per 4096-row chunk). */

/* grad_inputs[b, d] = sum_{l, ch} grad[l, b, ch] * dy_dx[b, l, d, ch] -- the second half of
 * ngp_grid_encode_backward (gridencoder.cu:352-378) on its own. */
int ngp_x_grid_input_backward(const float *grad, const float *dy_dx, float *grad_inputs, uint32_t B, uint32_t D,
                              uint32_t C, uint32_t L, int level_major, ngp_stream_t stream);
/* ngp_grid_encode_forward with the Jacobian stored level-major, dy_dx[level, b, d, ch], when level_major != 0 (pass the
 * same flag to ngp_x_grid_input_backward): with one (sample, level) per lane the reference layout [b, level, d, ch] makes
 * every lane write D*C floats 384 bytes apart (L = 16, D = 3, C = 2); level-major the wave writes one contiguous run. */
int ngp_x_grid_encode_forward_jac(const float *inputs, const float *embeddings, const int32_t *offsets, float *outputs,
                                  uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H,
                                  float *dy_dx, uint32_t gridtype, int align_corners, uint32_t interp, int level_major,
                                  ngp_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Fused tiny-MLP field (csrc/fused_mlp*.hip): replaces the six nn.Linear GEMMs + slicing / cat /
 * activation kernels of nerf/network.py:27-35,74-143 for the default field configuration
 * (grid_mlp 32-64-64-16, view_mlp 31-64-64-3, bias-free, ReLU, density trunc_exp, colour clamped_exp,
 * no rfield).  f16 MFMA operands with f32 accumulation = the precision of the reference's `--fp16`
 * autocast path; weights stay fp32 masters in torch layout [out][in].
 *   image  scratch of ngp_x_mlp_image_bytes() bytes, 16-byte aligned: f16 operand fragments, rebuilt by
 *          ngp_x_mlp_prepare whenever the weights change (once per optimiser step)
 *   enc    hash-grid features in the level-major slab layout the grid kernel writes: [16][stride][2]
 *   dirs   [M,3] view directions (any length; normalised in-kernel)
 *   M_dev  optional device int32: number of valid samples (clamped to M); NULL = use M
 * ---------------------------------------------------------------------------------- */
size_t ngp_x_mlp_image_bytes(void);
int ngp_x_mlp_prepare(const float *w1, const float *w2, const float *w3, const float *w4, const float *w5,
                      const float *w6, void *image, ngp_stream_t stream);
/* sigma [M], rgb [M,3]; rgb == NULL (then dirs may be NULL too) evaluates the density only */
int ngp_x_mlp_forward(const float *enc, uint32_t stride, const float *dirs, const int32_t *M_dev, uint32_t M,
                      const void *image, float *sigma, float *rgb, ngp_stream_t stream);
/* ... with the field's other activations (nerf/network.py:31-34,111-135): color_act 0 = clamp(exp(x - 5), max 5) (the
 * default above), 1 = exp(x - 5), 2 = sigmoid; density_act 0 = trunc_exp (default), 1 = softplus(beta, threshold 20);
 * internal_act 0 = ReLU hidden layers (default), 1 = softplus(beta, threshold 20) (opt.internal_activation). */
int ngp_x_mlp_forward_act(const float *enc, uint32_t stride, const float *dirs, const int32_t *M_dev, uint32_t M,
                          const void *image, float *sigma, float *rgb, uint32_t color_act, uint32_t density_act,
                          uint32_t internal_act, float beta, ngp_stream_t stream);
/* Density only, scattered (the density-grid refresh, nerf/renderer.py:874-880: evaluate the drawn cells, write them
 * into the scratch grid; where a cell was drawn twice the larger density stays): row i's density goes to tmp_cas[cells[i]] by atomic max, cells[i] < 0 is dropped -- ngp_x_mlp_forward with rgb == NULL
 * and ngp_x_density_grid_scatter as ONE launch, without the sigma array between them.  tmp_cas: the cascade's scratch grid. */
int ngp_x_mlp_density_scatter(const float *enc, uint32_t stride, uint32_t M, const void *image, const int32_t *cells,
                              float *tmp_cas, uint32_t density_act, uint32_t internal_act, float beta, ngp_stream_t stream);

/* Backward of the fused field.  dsigma [M], drgb [M,3] = dL/d(sigma, rgb); outputs d(enc) in the slab
 * layout of `enc` (rows >= M untouched) and the six weight gradients (fp32, torch layout, OVERWRITTEN).
 * Activations are recomputed from enc / dirs; nothing from the forward call is needed.  `loss_scale`
 * multiplies the incoming deltas before they become f16 operands and is divided out of every output
 * (the role GradScaler plays in the reference's --fp16 path, train_utils.py:404,897); deltas saturate at +-65504.
 *
 * Dynamic loss scale (`loss_scaler`, the _list / reduce / apply entry points; NULL: the static `loss_scale` above): eight
 * 32-bit device words owned by the caller -- [0] f32 scale, [1] f32 1 / scale, [2] u32 overflow seen in the current step,
 * [3] u32 clean steps since the scale last changed, [4] u32 optimiser steps taken, [5] u32 steps skipped, [6] u32 a step
 * has run, [7] reserved; initialise to {S, 1 / S, 0, 0, 0, 0, 0, 0}.  The backward kernels read [0] / [1] and let an
 * overflowing delta become inf (no saturation): it makes the weight gradient of its layer non-finite, the reduction of the
 * weight gradients raises [2], every optimiser kernel that is handed the words (or &[2] as `skip`) leaves its parameters
 * alone, and ngp_x_step_begin of the NEXT step halves the scale (or doubles it after growth_interval clean steps) and
 * advances Adam's t only for steps that were taken: torch.cuda.amp.GradScaler's scale / step / update
 * (train_utils.py:404,897-904) without a host read.
 * workspace: ngp_x_mlp_backward_workspace_bytes(M) bytes, 16-byte aligned. */
size_t ngp_x_mlp_backward_workspace_bytes(uint32_t M);
int ngp_x_mlp_backward(const float *enc, uint32_t stride, const float *dirs, const float *dsigma,
                       const float *drgb, const int32_t *M_dev, uint32_t M, const void *image, float loss_scale,
                       float *denc, float *dw1, float *dw2, float *dw3, float *dw4, float *dw5, float *dw6,
                       void *workspace, size_t workspace_bytes, ngp_stream_t stream);
/* dw1..dw6 all NULL: the per-workgroup partial sums stay in `workspace` and ngp_x_mlp_reduce_dw (same M, loss_scale,
 * workspace) produces the six gradients later -- e.g. on another stream, off the critical path. */
/* ... and d loss / d (un-normalised view direction) in ddirs [M,3] (NULL: not wanted): the SH Jacobian applied to the gradient
 * of the view MLP's SH inputs, then the tangent projection of d / |d| (renderer.py:541, sphere_harmonics.py:81) -- what pose
 * refinement needs from the field besides d enc. */
int ngp_x_mlp_backward_dirs(const float *enc, uint32_t stride, const float *dirs, const float *dsigma,
                            const float *drgb, const int32_t *M_dev, uint32_t M, const void *image, float loss_scale,
                            float *denc, float *ddirs, float *dw1, float *dw2, float *dw3, float *dw4, float *dw5,
                            float *dw6, void *workspace, size_t workspace_bytes, ngp_stream_t stream);
/* ngp_x_mlp_backward_dirs over a LIST of samples: sample_index [M] int32 (device; NULL = samples 0 .. M - 1), *M_dev entries
 * of it are used.  enc / dirs / dsigma / drgb / ddirs are addressed by sample, `denc` (and the delta-3 scratch) in LIST order:
 * denc[level][c] belongs to sample sample_index[c].  The fused step runs the backward over the samples in front of the
 * compositor's early stop only (ngp_x_composite_mse_train_idx) -- the others have exactly zero gradients. */
int ngp_x_mlp_backward_list(const float *enc, uint32_t stride, const float *dirs, const float *dsigma,
                            const float *drgb, const int32_t *M_dev, uint32_t M, const int32_t *sample_index,
                            const void *image, float loss_scale, float *denc, float *ddirs, float *dw1, float *dw2,
                            float *dw3, float *dw4, float *dw5, float *dw6, void *workspace, size_t workspace_bytes,
                            float *loss_scaler, ngp_stream_t stream);
/* ... with the activations of ngp_x_mlp_forward_act: their derivatives enter the deltas (hidden softplus: sigmoid(beta x)
 * recovered from the recomputed activation as 1 - exp(-beta softplus(x))) */
int ngp_x_mlp_backward_act(const float *enc, uint32_t stride, const float *dirs, const float *dsigma, const float *drgb,
                           const int32_t *M_dev, uint32_t M, const int32_t *sample_index, const void *image, float loss_scale,
                           float *denc, float *ddirs, float *dw1, float *dw2, float *dw3, float *dw4, float *dw5, float *dw6,
                           void *workspace, size_t workspace_bytes, float *loss_scaler, uint32_t color_act,
                           uint32_t density_act, uint32_t internal_act, float beta, ngp_stream_t stream);
/* d h0 / d enc for samples 0 .. M - 1 (level-major slab like `enc`), h0 = the density network's first output, sigma =
 * trunc_exp(h0) (nerf/network.py:111-118): the MLP's part of torch.autograd.grad(sigma, pos) in the orientation term
 * (nerf/renderer.py:558-566).  The f16 chain of the backward's density kernel with delta = e_0; no weight gradients. */
int ngp_x_mlp_density_gradient(const float *enc, uint32_t stride, const int32_t *M_dev, uint32_t M, const void *image,
                               float *denc, ngp_stream_t stream);
/* ... for the light-conditioned field (`image` from ngp_x_mlp_rf_prepare; the density network is the same); level_w
 * (optional, [16]): the level window its kernels apply to the features (as ngp_x_mlp_rf_forward) -- d enc is then the
 * gradient with respect to the un-windowed features */
int ngp_x_mlp_rf_density_gradient(const float *enc, uint32_t stride, const float *level_w, const int32_t *M_dev, uint32_t M,
                                  const void *image, float *denc, ngp_stream_t stream);
int ngp_x_mlp_reduce_dw(uint32_t M, float loss_scale, float *dw1, float *dw2, float *dw3, float *dw4, float *dw5,
                        float *dw6, const void *workspace, size_t workspace_bytes, float *adam_param,
                        const float *adam_grad, float *adam_exp_avg, float *adam_exp_avg_sq, uint32_t adam_n,
                        const float *adam_hyper, float beta1, float beta2, float eps, void *adam_image,
                        float *loss_scaler, ngp_stream_t stream);
/* adam_param != NULL: dw1..dw6 are views of the flat buffer adam_grad (adam_n floats) and every element is also pushed
 * through ngp_x_adam_step_dev's update of adam_param / exp_avg / exp_avg_sq as it comes out of the reduction.
 * adam_image != NULL (an image ngp_x_mlp_prepare has filled once): each updated weight is also written, as f16, to its
 * two places in the operand image, so the next forward needs no prepare pass.
 * loss_scaler != NULL: 1 / scale comes from it, non-finite sums raise its overflow word, and Adam (if asked for) runs as a
 * second launch that is skipped when the word is set. */

/* ------------------------------------------------------------------------------------
 * The same fused field for the light-conditioned configuration (`--rfield`, nerf/network.py:55-56,111-143:
 * view_mlp = MLP(15 + 16 + 16, 3, 64 + 16, 3): colour from [features, SH16(view dir), SH16(light dir)] through
 * 47 -> 80 -> 80 -> 3), csrc/fused_mlp_rf.hip.  w4 [80,47], w5 [80,80], w6 [3,80]; w1..w3 as above.
 *   ldirs    [M,3] light directions (normalised in-kernel, like the view directions)
 *   level_w  optional device float[16]: per-level weights multiplied onto the encoder features before the density MLP
 *            (the BARF window of network.py:99-109); the backward scales d(enc) by the same weights.  NULL = ones
 *   ddirs    optional [M,3]: d loss / d (un-normalised view direction), through the SH basis' Jacobian
 *            (shencoder.cu:126-350) and d / |d| (renderer.py:541): what raymarching.py:319-329 sums per ray
 * Hidden layers are ReLU; the output activations default to trunc_exp density and clamped_exp colour, the *_act entry points
 * take the reference's others (as ngp_x_mlp_forward_act).
 * ---------------------------------------------------------------------------------- */
size_t ngp_x_mlp_rf_image_bytes(void);
int ngp_x_mlp_rf_prepare(const float *w1, const float *w2, const float *w3, const float *w4, const float *w5,
                         const float *w6, void *image, ngp_stream_t stream);
/* rgb == NULL (then dirs / ldirs may be NULL): density only */
int ngp_x_mlp_rf_forward(const float *enc, uint32_t stride, const float *dirs, const float *ldirs, const float *level_w,
                         const int32_t *M_dev, uint32_t M, const void *image, float *sigma, float *rgb,
                         ngp_stream_t stream);
/* ... with the field's other OUTPUT activations (nerf/network.py:115,131-135): color_act 0 = clamp(exp(x - 5), max 5),
 * 1 = exp(x - 5), 2 = sigmoid; density_act 0 = trunc_exp, 1 = softplus(beta, threshold 20) */
int ngp_x_mlp_rf_forward_act(const float *enc, uint32_t stride, const float *dirs, const float *ldirs, const float *level_w,
                             const int32_t *M_dev, uint32_t M, const void *image, float *sigma, float *rgb,
                             uint32_t color_act, uint32_t density_act, float beta, ngp_stream_t stream);
size_t ngp_x_mlp_rf_backward_workspace_bytes(uint32_t M);
/* outputs: d(enc) (slab layout, rows >= M untouched), ddirs (optional), six weight gradients (OVERWRITTEN) */
int ngp_x_mlp_rf_backward(const float *enc, uint32_t stride, const float *dirs, const float *ldirs, const float *level_w,
                          const float *dsigma, const float *drgb, const int32_t *M_dev, uint32_t M, const void *image,
                          float loss_scale, float *denc, float *ddirs, float *dw1, float *dw2, float *dw3, float *dw4,
                          float *dw5, float *dw6, void *workspace, size_t workspace_bytes, ngp_stream_t stream);
/* ... over a LIST of samples (as ngp_x_mlp_backward_list): inputs and ddirs by sample, denc in list order */
int ngp_x_mlp_rf_backward_list(const float *enc, uint32_t stride, const float *dirs, const float *ldirs, const float *level_w,
                               const float *dsigma, const float *drgb, const int32_t *M_dev, uint32_t M,
                               const int32_t *sample_index, const void *image, float loss_scale, float *denc, float *ddirs,
                               float *dw1, float *dw2, float *dw3, float *dw4, float *dw5, float *dw6, void *workspace,
                               size_t workspace_bytes, float *loss_scaler, ngp_stream_t stream);
/* ... with the output activations of ngp_x_mlp_rf_forward_act: their derivatives enter the output deltas */
int ngp_x_mlp_rf_backward_act(const float *enc, uint32_t stride, const float *dirs, const float *ldirs, const float *level_w,
                              const float *dsigma, const float *drgb, const int32_t *M_dev, uint32_t M,
                              const int32_t *sample_index, const void *image, float loss_scale, float *denc, float *ddirs,
                              float *dw1, float *dw2, float *dw3, float *dw4, float *dw5, float *dw6, void *workspace,
                              size_t workspace_bytes, float *loss_scaler, uint32_t color_act, uint32_t density_act,
                              float beta, ngp_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Kernels of the fused training step (csrc/engine_kernels.hip; host side raw_ngp_amd/nerf/engine.py).
 * ---------------------------------------------------------------------------------- */

/* ngp_grid_encode_forward for D = 3, C = 2 on WORLD-space points: applies GridEncoder.forward's map
 * x01 = (x + bound) / (2 bound) (gridencoder/grid.py:161) on load, writes out[L][stride][2] and, if
 * inputs01 != NULL, the mapped points [B,3] (what the backward kernels take as `inputs`).
 * B_dev: optional device int32 with the number of live points (clamped to B_cap). */
int ngp_x_grid_encode_forward_slab(const float *xyzs, float bound, const float *embeddings, const int32_t *offsets,
                                   float *out, float *inputs01, const int32_t *B_dev, uint32_t B_cap, uint32_t stride,
                                   uint32_t L, uint32_t max_level, float S, uint32_t H, uint32_t gridtype,
                                   int align_corners, uint32_t interp, void *binned_workspace, uint32_t n_rows_total,
                                   ngp_stream_t stream);
/* the same with the Jacobian d out / d x01 (gridencoder.cu:205-247) as a level-major slab dydx[L][stride][3][2]
 * (dydx == NULL: as above).  Contracted with d(enc) per ray by ngp_x_ray_gradients. */
int ngp_x_grid_encode_forward_slab_jac(const float *xyzs, float bound, const float *embeddings, const int32_t *offsets,
                                       float *out, float *inputs01, const int32_t *B_dev, uint32_t B_cap, uint32_t stride,
                                       uint32_t L, uint32_t max_level, float S, uint32_t H, uint32_t gridtype,
                                       int align_corners, uint32_t interp, void *binned_workspace, uint32_t n_rows_total,
                                       float *dydx, ngp_stream_t stream);
/* levels level_lo .. level_hi - 1 only (the other levels' slab rows stay untouched; inputs01 is written with level 0; dydx
 * optional): the data-parallel step encodes the levels whose parameters have arrived while the all-gather of the others is
 * still on the wire.  Same arithmetic per level, so two calls that cover [0, L) equal one call of the whole. */
int ngp_x_grid_encode_forward_slab_levels(const float *xyzs, float bound, const float *embeddings, const int32_t *offsets,
                                          float *out, float *inputs01, const int32_t *B_dev, uint32_t B_cap, uint32_t stride,
                                          uint32_t L, uint32_t level_lo, uint32_t level_hi, float S, uint32_t H,
                                          uint32_t gridtype, int align_corners, uint32_t interp, float *dydx,
                                          ngp_stream_t stream);
/* the same with the caller's level -> XCD placement: level_cost (HOST pointer, max_level floats > 0, or NULL) is the
 * relative cost of one 256-point tile of each level for the caller's points (ray-ordered samples: growing with the
 * level; scattered points: flat over the hashed levels).  The levels are dealt to the 8 XCDs in runs of equal cost
 * instead of the fixed pairing (level k with 15 - k) the other two entries use.  Placement only; results identical. */
int ngp_x_grid_encode_forward_slab_placed(const float *xyzs, float bound, const float *embeddings, const int32_t *offsets,
                                          float *out, float *inputs01, const int32_t *B_dev, uint32_t B_cap,
                                          uint32_t stride, uint32_t L, uint32_t max_level, float S, uint32_t H,
                                          uint32_t gridtype, int align_corners, uint32_t interp, void *binned_workspace,
                                          uint32_t n_rows_total, float *dydx, const float *level_cost,
                                          ngp_stream_t stream);
/* binned_workspace != NULL: a workspace of ngp_x_grid_backward_binned_prepare(stage 1) for the same samples; the
 * kernel also counts the records per 4096-row chunk (n_rows_total = rows of the whole table). */

/* ngp_composite_rays_train_forward / _backward with one wave per ray (prefix product / scans instead of the
 * serial walk).  Same arguments and results, except that EVERY sample of a live ray is written (zeros after
 * the early stop), so weights / grad_sigmas / grad_rgbs need no zero-initialisation. */
int ngp_x_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *ts,
                                       const int32_t *rays, uint32_t M, uint32_t N, float T_thresh, float *weights,
                                       float *weights_sum, float *depth, float *image, ngp_stream_t stream);
int ngp_x_composite_rays_train_backward(const float *grad_weights, const float *grad_weights_sum,
                                        const float *grad_depth, const float *grad_image, const float *sigmas,
                                        const float *rgbs, const float *ts, const int32_t *rays,
                                        const float *weights_sum, const float *depth, const float *image, uint32_t M,
                                        uint32_t N, float T_thresh, float *grad_sigmas, float *grad_rgbs,
                                        ngp_stream_t stream);

/* Backward of the harness loss through the compositor in one kernel:
 *   pred = image + (1 - weights_sum) * bg            (nerf/renderer.py:672)
 *   gt   = rgb * a + bg * (1 - a)                    (nerf/train_utils.py:503-506), gt_rgba [N,4]
 *   loss = mean over rays and channels of (pred - gt)^2   (:540-541; added to loss_out[0])
 * bg_rgb [N,3] or NULL (then bg_const for all channels).  Writes d loss / d sigma, d loss / d rgb. */
int ngp_x_composite_mse_backward(const float *gt_rgba, const float *bg_rgb, float bg_const, const float *sigmas,
                                 const float *rgbs, const float *ts, const int32_t *rays, const float *weights_sum,
                                 const float *depth, const float *image, uint32_t M, uint32_t N, float T_thresh,
                                 float *grad_sigmas, float *grad_rgbs, float *loss_out, ngp_stream_t stream);
/* ngp_x_composite_rays_train_forward + ngp_x_composite_mse_backward in one launch (what a training step needs): each
 * wave composites its ray, writes weights_sum / depth / image (no per-sample weights) and runs the loss backward with
 * the totals it still holds.  Same bits as the two calls. */
int ngp_x_composite_mse_train(const float *gt_rgba, const float *bg_rgb, float bg_const, const float *sigmas,
                              const float *rgbs, const float *ts, const int32_t *rays, uint32_t M, uint32_t N,
                              float T_thresh, float *weights_sum, float *depth, float *image, float *grad_sigmas,
                              float *grad_rgbs, float *loss_out, ngp_stream_t stream);
/* ngp_x_composite_mse_train that also lists the samples in front of the early stop -- the only ones whose output gradients
 * can be non-zero: live_n [N] = their number per ray, live_idx [M] = their indices in ray order, live_count [1] = how many.
 * The backward kernels of the fused step run over that list (ngp_x_mlp_backward_list, ..._binned_apply_mlp_list). */
int ngp_x_composite_mse_train_idx(const float *gt_rgba, const float *bg_rgb, float bg_const, const float *sigmas,
                                  const float *rgbs, const float *ts, const int32_t *rays, uint32_t M, uint32_t N,
                                  float T_thresh, float *weights_sum, float *depth, float *image, float *grad_sigmas,
                                  float *grad_rgbs, float *loss_out, int32_t *live_n, int32_t *live_idx,
                                  int32_t *live_count, ngp_stream_t stream);
/* The same launch with the HDR loss of nerf/train_utils.py:512-536 (`--image_mode HDR`, the RawNeRF loss) in place of the
 * MSE:  clip = min(1, pred * exposure[n]);  loss = sum((clip - gt)^2 / (1e-3 + sg(clip))^2 * weight) * inv_norm, no
 * gradient where pred * exposure >= 1.  exposure [N]; weight [N,3] = lossmult * loss_weight or NULL (ones);
 * inv_norm = 1 / sum(lossmult) (= 1 / (3 N) without a Bayer mask). */
int ngp_x_composite_hdr_train(const float *gt_rgba, const float *bg_rgb, float bg_const, const float *exposure,
                              const float *weight, float inv_norm, const float *sigmas, const float *rgbs,
                              const float *ts, const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                              float *weights_sum, float *depth, float *image, float *grad_sigmas, float *grad_rgbs,
                              float *loss_out, ngp_stream_t stream);

/* One torch.optim.Adam step (no amsgrad, no weight decay; main.py:245 uses eps 1e-15) over a flat fp32
 * tensor in a single pass; `step` counts from 1; zero_grad != 0 clears `grad` afterwards. */
int ngp_x_adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, uint64_t n, float lr, double beta1,
                    double beta2, float eps, uint32_t step, int zero_grad, ngp_stream_t stream);

/* Same update with {lr, 1 - beta1^t, 1/sqrt(1 - beta2^t)} read from `hyper` (device, written by
 * ngp_x_schedule_step earlier on the stream): no host scalar changes between steps, so the launch can be replayed
 * from a captured graph. */
int ngp_x_adam_step_dev(float *param, float *grad, float *exp_avg, float *exp_avg_sq, uint64_t n, const float *hyper,
                        float beta1, float beta2, float eps, int zero_grad, const uint32_t *skip, ngp_stream_t stream);
/* skip (NULL: never): a device word; non-zero = leave everything alone (the step saw a non-finite gradient: word [2] of the
 * dynamic loss scale -- GradScaler.step, train_utils.py:897) */
/* Two parameter tensors in one launch (the hash table and the flat MLP weights).  grad_a_bf16 != 0: grad_a points
 * to n_a bfloat16 values (the data-parallel wire format written by ngp_x_grid_backward_binned_apply with
 * overwrite = 2 and averaged over the ranks in place); it is then never zeroed. */
int ngp_x_adam_step_dev2(float *param_a, float *grad_a, float *exp_avg_a, float *exp_avg_sq_a, uint64_t n_a,
                         int zero_grad_a, float *param_b, float *grad_b, float *exp_avg_b, float *exp_avg_sq_b,
                         uint64_t n_b, int zero_grad_b, const float *hyper, float beta1, float beta2, float eps,
                         int grad_a_bf16, const uint32_t *skip, ngp_stream_t stream);

/* Device-side scheduler: t = step_counter[0] steps are done; writes hyper = {lr0 * 0.1^min(t/decay_steps, 1)
 * (the LambdaLR of main.py:261), 1 - beta1^(t+1), 1/sqrt(1 - beta2^(t+1))} and increments the counter. */
int ngp_x_schedule_step(uint32_t *step_counter, float *hyper, double lr0, double decay_steps, double beta1,
                        double beta2, ngp_stream_t stream);
/* ... plus the other per-step scalars of the training step in the same launch: loss_out[0] = 0 (the compositor's
 * backward accumulates into it) and samples_seen[0] += sample_counter[0]; each pair may be NULL. */
int ngp_x_step_begin(uint32_t *step_counter, float *hyper, double lr0, double decay_steps, double beta1, double beta2,
                     float *loss_out, int64_t *samples_seen, const int32_t *sample_counter, void *binned_workspace,
                     uint32_t L, uint32_t n_rows_total, int single_segment, float *loss_scaler, double growth, double backoff,
                     uint32_t growth_interval, ngp_stream_t stream);
/* loss_scaler != NULL (the eight words of the dynamic loss scale): the PREVIOUS step is settled first -- overflow: scale *=
 * backoff, tracker = 0, skipped + 1; else taken + 1 and, after growth_interval clean steps, scale *= growth (GradScaler's
 * defaults: 2, 0.5, 2000) -- and the bias corrections follow the optimiser steps taken instead of the step counter (the
 * learning rate follows the step counter either way: lr_scheduler.step() is unconditional, train_utils.py:906-907). */
/* binned_workspace != NULL: the same launch also does ngp_x_grid_backward_binned_prepare(stage 2) on that workspace
 * (L levels, n_rows_total table rows) -- call it after the encoder's counting forward pass. */

/* ngp_x_mlp_forward with ngp_x_step_begin (same arguments, same checks) as one more workgroup of the same launch: the
 * step's one-workgroup bookkeeping leaves the critical path (nothing reads its results before the compositor). */
int ngp_x_mlp_forward_step_begin(const float *enc, uint32_t stride, const float *dirs, const int32_t *M_dev, uint32_t M,
                                 const void *image, float *sigma, float *rgb, uint32_t *step_counter, float *hyper,
                                 double lr0, double decay_steps, double beta1, double beta2, float *loss_out,
                                 int64_t *samples_seen, const int32_t *sample_counter, void *binned_workspace, uint32_t L,
                                 uint32_t n_rows_total, int single_segment, float *loss_scaler, double growth,
                                 double backoff, uint32_t growth_interval, ngp_stream_t stream);

/* counter[0] += delta, stream-ordered. */
int ngp_x_counter_add(uint32_t *counter, uint32_t delta, ngp_stream_t stream);

/* Training-ray batch in one kernel: what the harness does with torch.randint + get_rays + an index gather
 * (nerf/provider.py collate, nerf/train_utils.py:96-172, :492-506).  Ray n draws (view, pixel) with Philox4x32-10,
 * counter (n, draw, 0, 0), key = seed: view = mulhi(r0, V), pixel = mulhi(r1, H*W); noise = 24 high bits of r2;
 * bg_rgb from counter (n, draw, 1, 0).  `draw` is read from draw_dev[0] when draw_dev is not NULL.
 * images [V,H,W,C] uint8 (C = 3 or 4, straight alpha), poses [V,4,4] camera-to-world; outputs rays_o/rays_d [N,3],
 * gt_rgba [N,4] in [0,1] (alpha 1 when C = 3), optional noises [N], bg_rgb [N,3], index [N,2] = (view, pixel). */
int ngp_x_sample_rays(const uint8_t *images, uint32_t V, uint32_t H, uint32_t W, uint32_t C, const float *poses, float fx,
                      float fy, float cx, float cy, uint32_t N, uint64_t seed, const uint32_t *draw_dev, uint32_t draw,
                      float *rays_o, float *rays_d, float *gt_rgba, float *noises, float *bg_rgb, int32_t *index,
                      ngp_stream_t stream);
/* ... and per-ray light directions rays_ldir [N,3] = view_ldirs[view] (colmap_provider.py:619-620; both NULL: as above) */
int ngp_x_sample_rays_lit(const uint8_t *images, uint32_t V, uint32_t H, uint32_t W, uint32_t C, const float *poses,
                          float fx, float fy, float cx, float cy, uint32_t N, uint64_t seed, const uint32_t *draw_dev,
                          uint32_t draw, float *rays_o, float *rays_d, float *gt_rgba, float *noises, float *bg_rgb,
                          int32_t *index, const float *view_ldirs, float *rays_ldir, ngp_stream_t stream);

/* ... and adaptive batch sizes (`--adaptive_num_rays`, train_utils.py:563-564) decided on the device: the batch gets
 * live[0] = clamp(round(num_points / prev_samples[0] * prev_live[0]), 1, N) rays (prev_samples / prev_live NULL: N), written
 * by the kernel; ray slots >= live[0] are parked outside the volume (no samples; index -1; exposure 1).  prev_live may be
 * the same word as live (one ray slot): the count is then formed by a one-thread launch in front of the sampler.
 * view_exposure [V] + exposure [N] (both or neither): exposure[n] = view_exposure[view] (colmap_provider.py:605-606). */
int ngp_x_sample_rays_adaptive(const uint8_t *images, uint32_t V, uint32_t H, uint32_t W, uint32_t C, const float *poses,
                               float fx, float fy, float cx, float cy, uint32_t N, uint64_t seed, const uint32_t *draw_dev,
                               uint32_t draw, float *rays_o, float *rays_d, float *gt_rgba, float *noises, float *bg_rgb,
                               int32_t *index, const float *view_ldirs, float *rays_ldir, const int32_t *prev_samples,
                               const int32_t *prev_live, int32_t *live, uint32_t num_points, const float *view_exposure,
                               float *exposure, ngp_stream_t stream);
/* ngp_x_composite_mse_train (exposure == NULL) / ngp_x_composite_hdr_train with the loss taken over the first n_live[0]
 * of the N ray slots only (mean over those rays; n_live == NULL: all N), plus the entropy term of train_utils.py:554-557
 * when lambda_entropy > 0: loss += lambda_entropy * mean_rays(H(clamp(weights_sum, 1e-5, 1 - 1e-5))),
 * H(w) = -w log2 w - (1 - w) log2(1 - w), with its gradient through weights_sum. */
int ngp_x_composite_train_live(const float *gt_rgba, const float *bg_rgb, float bg_const, const float *exposure,
                               const float *weight, float inv_norm, const int32_t *n_live, float lambda_entropy,
                               const float *sigmas, const float *rgbs, const float *ts, const int32_t *rays, uint32_t M,
                               uint32_t N, float T_thresh, float *weights_sum, float *depth, float *image,
                               float *grad_sigmas, float *grad_rgbs, float *loss_out, ngp_stream_t stream);
/* ... that also lists the samples in front of the early stop (as ngp_x_composite_mse_train_idx); live_off [N] (optional):
 * where each ray's entries start in the list.  live_n == NULL: no list. */
int ngp_x_composite_train_live_idx(const float *gt_rgba, const float *bg_rgb, float bg_const, const float *exposure,
                                   const float *weight, float inv_norm, const int32_t *n_live, float lambda_entropy,
                                   const float *sigmas, const float *rgbs, const float *ts, const int32_t *rays, uint32_t M,
                                   uint32_t N, float T_thresh, float *weights_sum, float *depth, float *image,
                                   float *grad_sigmas, float *grad_rgbs, float *loss_out, int32_t *live_n,
                                   int32_t *live_idx, int32_t *live_count, int32_t *live_off, ngp_stream_t stream);
/* ... plus a term over the samples' compositing weights (nerf/renderer.py:571, train_utils.py:546-548 with
 * sample_term = ngp_x_orientation_term's output): loss += lambda_sample * sum_i weights[i] * sample_term[i] -- a SUM over
 * all samples (the reference takes torch.mean of a scalar).  The gradient is the one the reference's compositor backward
 * gives grad_weights (raymarching.cu:694: added to grad_weights_sum at the sample itself); sample_term itself is a
 * constant (the reference's autograd.grad runs without create_graph).  sample_term == NULL: none. */
int ngp_x_composite_train_terms(const float *gt_rgba, const float *bg_rgb, float bg_const, const float *exposure,
                                const float *weight, float inv_norm, const int32_t *n_live, float lambda_entropy,
                                const float *sample_term, float lambda_sample, const float *sigmas, const float *rgbs,
                                const float *ts, const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                                float *weights_sum, float *depth, float *image, float *grad_sigmas, float *grad_rgbs,
                                float *loss_out, int32_t *live_n, int32_t *live_idx, int32_t *live_count, int32_t *live_off,
                                float *term_weight, ngp_stream_t stream);
/* term_weight (optional, [M]) <- lambda_sample * weights[i]: what a consumer of the term's other inputs multiplies its own
 * derivative with (pose refinement: the orientation term also depends on the view direction, ngp_x_ray_gradients_terms) */
/* The per-sample factor of the orientation term (nerf/renderer.py:558-571): g = d sigma / d xyz = clamp(sigma, e^-80, e^80)
 * (trunc_exp's backward, activation.py:20) * sum_l dh_denc_l . dydx_l / (2 bound) from the level-major slabs of
 * ngp_x_mlp_density_gradient and ngp_x_grid_encode_forward_slab_jac; normal = (-g / max(|g|, 1e-12) + 1) / 2;
 * term[i] = min(0, sum_d normal_d * -(dirs / |dirs|)_d)^2.  One thread per sample, samples 0 .. min(*M_dev, M) - 1. */
int ngp_x_orientation_term(const float *dh_denc, const float *dydx, uint32_t stride, uint32_t L, float bound,
                           const float *sigmas, const float *dirs, const int32_t *M_dev, uint32_t M, float *term,
                           float *dterm_ddirs, ngp_stream_t stream);
/* ... for a softplus density (nerf/network.py:115; density_act 1, beta as in ngp_x_mlp_forward_act): d sigma / d h0 =
 * sigmoid(beta h0) = 1 - exp(-beta sigma) */
int ngp_x_orientation_term_act(const float *dh_denc, const float *dydx, uint32_t stride, uint32_t L, float bound,
                           const float *sigmas, const float *dirs, const int32_t *M_dev, uint32_t M, float *term,
                           float *dterm_ddirs, uint32_t density_act, float beta, ngp_stream_t stream);
/* dterm_ddirs (optional, [M,3]) <- d term[i] / d dirs[i] (un-normalised direction; the normal is a constant, as in the
 * reference, whose autograd.grad runs without create_graph) */

/* ---- pose refinement around the fused step (csrc/pose_kernels.hip) -----------------------------------------------
 * ngp_x_step_window   annealing = float16((step_counter[0] + step_offset) / iters) (train_utils.py:488) -> the BARF level
 *                     window of network.py:99-109 for L levels (float16 alpha, float32 cosine, level 0 forced to 1) in
 *                     level_w[L]; flags (optional, int32[2]) = {annealing < end_annealing, step}
 * ngp_x_ray_gradients raymarching.py:319-329 + gridencoder.cu:352-378 in one pass: per ray the sums over its samples of
 *                     d xyz and of ts[:,0] * d xyz + d dirs, with d xyz = sum_l d enc_l . dydx_l / (2 bound) formed on the
 *                     fly from the level-major slabs (denc [L][stride][2], dydx [L][stride][3][2]); ddirs may be NULL
 * ngp_x_pose_gradient d loss / d (camera-to-world 3 x 4) per camera [V][12] from the ray gradients and the batch's
 *                     (view, pixel) list (the adjoint of get_rays, train_utils.py:150-160); fixed summation order
 * ngp_x_pose_update   xi [V,6] se(3) corrections, base [V,12] the poses they refine: refined[V,16] =
 *                     compose(exp(xi), base) (barf/camera.py:47-63,91-102).  grad_pose != NULL: first one
 *                     torch.optim.Adam step on xi (lr = lr0 * gamma^step, bias corrections from step + 1, step = flags[1])
 *                     when flags[0] != 0, with d loss / d xi by forward-mode differentiation of the exponential map;
 *                     grad_xi (optional) receives that gradient; loss_scaler (optional, the eight words of the dynamic
 *                     loss scale): no step when its overflow word is set, Adam's t counts the steps taken
 *                     (train_utils.py:898-899: the pose optimiser goes through the same GradScaler) */
int ngp_x_step_window(const uint32_t *step_counter, uint32_t step_offset, double iters, float start_annealing,
                      float end_annealing, uint32_t L, float *level_w, int32_t *flags, ngp_stream_t stream);
/* ... the BAA-NGP window (network.py:77-97): level 0 always counts, level j >= 1 ramps in like the (j - 1)-th of L - 1 levels;
 * ngp_x_slab_window applies it to the level-major encoder slab [L][stride][2] in place: with c = the finest level whose
 * weight is > 0, f'_l = w_l f_l + (1 - w_l) f_c (backward != 0: the adjoint, in place on the slab of feature gradients). */
int ngp_x_step_window_baa(const uint32_t *step_counter, uint32_t step_offset, double iters, float start_annealing,
                          float end_annealing, uint32_t L, float *level_w, int32_t *flags, ngp_stream_t stream);
int ngp_x_slab_window(float *slab, uint32_t stride, uint32_t L, const float *level_w, const int32_t *M_dev, uint32_t M,
                      int backward, ngp_stream_t stream);
/* backward == 2: no blend, f'_l = w_l f_l in place (the BARF window, network.py:99-109, for field kernels that do not apply
 * it themselves; it is its own adjoint) */
int ngp_x_ray_gradients(const float *denc, const float *dydx, uint32_t stride, uint32_t L, float bound, const float *ddirs,
                        const float *ts, const int32_t *rays, uint32_t N, uint32_t M, float *grad_rays_o,
                        float *grad_rays_d, ngp_stream_t stream);
/* ... plus, per sample, term_weight[i] * dterm_ddirs[i] added to the direction gradient (both NULL: none) -- the orientation
 * term's path to the cameras through the view direction; live_n / live_off as ngp_x_ray_gradients_list or both NULL */
int ngp_x_ray_gradients_terms(const float *denc, const float *dydx, uint32_t stride, uint32_t L, float bound,
                              const float *ddirs, const float *ts, const int32_t *rays, const int32_t *live_n,
                              const int32_t *live_off, const float *term_weight, const float *dterm_ddirs, uint32_t N,
                              uint32_t M, float *grad_rays_o, float *grad_rays_d, ngp_stream_t stream);
/* ... when the backward ran over the list of live samples: denc in list order -- ray n's entries start at live_off[n], the
 * first live_n[n] samples of the ray have gradients (dydx, ts, ddirs stay in sample order) */
int ngp_x_ray_gradients_list(const float *denc, const float *dydx, uint32_t stride, uint32_t L, float bound,
                             const float *ddirs, const float *ts, const int32_t *rays, const int32_t *live_n,
                             const int32_t *live_off, uint32_t N, uint32_t M, float *grad_rays_o, float *grad_rays_d,
                             ngp_stream_t stream);
int ngp_x_pose_gradient(const int32_t *index, const float *grad_rays_o, const float *grad_rays_d, uint32_t N, uint32_t V,
                        uint32_t W, float fx, float fy, float cx, float cy, float *grad_pose, ngp_stream_t stream);
int ngp_x_pose_update(float *xi, const float *base, const float *grad_pose, uint32_t V, const int32_t *flags,
                      float *exp_avg, float *exp_avg_sq, float lr0, float gamma, float beta1, float beta2, float eps,
                      float *refined, float *grad_xi, const float *loss_scaler, ngp_stream_t stream);

/* The slab test NeRFRenderer.run_cuda actually uses (the torch function, nerf/renderer.py:139-158, not the
 * CUDA kernel): divides by (d + 1e-15), marks a miss with near = far = 1e9. */
int ngp_x_near_far_from_aabb_v2(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N,
                                float min_near, float *nears, float *fars, ngp_stream_t stream);

/* ---- density-grid refresh on the device (NeRFRenderer.update_extra_state, nerf/renderer.py:811-897) -------------
 * One cascade at a time:
 *   ngp_x_density_grid_sample   cells to re-evaluate: n_uniform uniformly drawn cells (full != 0: every cell once,
 *                               n_uniform = H^3) then n_occupied cells drawn uniformly among those with density > 0
 *                               (index -1 when none is); xyz = (2 c/(H-1) - 1) * span + (2u - 1) * half with
 *                               span = bound_cas - half (renderer.py:868-872).  Philox4x32-10, counters (i, draw, 2|3, 0).
 *                               Random draws (full == 0; H a power of two, H^3 >= 4096) are GENERATED half by half in
 *                               Morton order of their cells, bin by bin (4096 bins per half) -- the encoder that
 *                               evaluates them runs faster on neighbouring points.  n independent uniform draws are a
 *                               multinomial count per bin and, inside every bin, that many independent uniform draws:
 *                               the counts are those of the keys of the stream (i, draw, 2, 0); output slot j, in the
 *                               bin whose slot range holds it, takes its position inside the bin from (j, draw, 4, 0)
 *                               and its jitter from (j, draw, 5, 0).  Same law as the independent draws (what other
 *                               sizes and NGP_REFRESH_SORT=0 deliver, in draw order), reproducible slot by slot
 *                               (oracle.density_grid_sample(..., binned=True)).
 *   (caller evaluates the density at xyzs)
 *   ngp_x_density_grid_scatter  tmp[index] = max(tmp[index], sigma); tmp holds -1 where nothing was evaluated
 * then over all cascades:
 *   ngp_x_density_grid_update   grid = max(grid * decay, tmp) where grid >= 0 and tmp >= 0; tmp is reset to -1;
 *                               stats[4 .. 4 + 1024) = partial sums of clamp(grid, 0) (stats: 1028 floats)
 *   ngp_x_packbits_mean         ngp_packbits with thresh = min(mean, density_thresh), mean = the partials added in a
 *                               fixed order / cells (bitwise reproducible); N = bytes of the bitfield; writes
 *                               stats[0] = sum, stats[1] = mean density, stats[2] = thresh
 * Nothing is read back to the host. */
size_t ngp_x_density_grid_workspace_bytes(uint32_t H);
int ngp_x_density_grid_sample(const float *grid_cas, uint32_t H, float span, float half, uint32_t n_uniform,
                              uint32_t n_occupied, int full, uint64_t seed, const uint32_t *draw_dev, uint32_t draw,
                              void *workspace, size_t workspace_bytes, int32_t *indices, float *xyzs,
                              ngp_stream_t stream);
int ngp_x_density_grid_scatter(const int32_t *indices, const float *sigmas, uint32_t n, float *tmp_cas,
                               ngp_stream_t stream);
int ngp_x_density_grid_update(float *grid, float *tmp, uint32_t n_cells, float decay, float *stats,
                              ngp_stream_t stream);
int ngp_x_packbits_mean(const float *grid, uint32_t N, float *stats, float density_thresh, uint8_t *bitfield,
                        ngp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NGP_HIP_H */
