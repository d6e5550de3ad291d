/*
 * g4s_rasterizer.h -- C ABI of the MI355X (gfx950) surfel rasterizer, libg4s_hip.so.
 *
 * This is the drop-in boundary for G4Splat's `gaussian_renderer.render()` hot path: the
 * entry points below are what the reference's torch glue
 * (dsr/rasterize_points.cu, knn/spatial.cu) binds underneath, with the C++ classes
 * `CudaRasterizer::Rasterizer` (dsr/cuda_rasterizer/rasterizer.h:24-86) and `SimpleKNN`
 * (knn/simple_knn.h:15-19) flattened to `extern "C"`:
 *
 *   - plain device pointers and sizes, no torch / STL types;
 *   - the three `std::function<char*(size_t)>` scratch callbacks of
 *     Rasterizer::forward (rasterizer.h:32-34) become `char* (*)(void* ctx, size_t)`;
 *   - every launch goes to the caller's `hipStream_t` (passed as `void*`; NULL = the
 *     null stream) instead of the legacy default stream;
 *   - errors are an int status + g4s_last_error() instead of C++ exceptions.
 *
 * All pointers are device pointers unless stated.  Inputs are borrowed and read-only;
 * outputs and the scratch chunks are owned by the caller.  "Absent" optional inputs are
 * NULL (the reference passes empty tensors whose data pointer is null,
 * dsr/cuda_rasterizer/rasterizer_impl.cu:322-323).  Re-entrant across host threads and
 * devices (no global mutable state besides a thread-local pinned word).
 *
 * dsr/ = 2d-gaussian-splatting/submodules/diff-surfel-rasterization/
 * knn/ = 2d-gaussian-splatting/submodules/simple-knn/
 */
#ifndef G4S_RASTERIZER_H_INCLUDED
#define G4S_RASTERIZER_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define G4S_OK 0
#define G4S_ERR_INVALID_ARGUMENT (-1) /* bad shape / NULL required pointer / limit exceeded */
#define G4S_ERR_HIP (-2)              /* a HIP runtime call or kernel launch failed */
#define G4S_ERR_ALLOC (-3)            /* a resize callback returned NULL */
#define G4S_ERR_UNSUPPORTED (-4)      /* e.g. NUM_CHANNELS != 3 without precomputed colours */

/* Scratch-chunk resize callback: must return a device pointer to at least `nbytes`
 * bytes (any alignment; the library aligns sub-allocations to 256 B itself) that stays
 * valid until the matching backward call.  Replaces std::function<char*(size_t)>
 * (dsr/cuda_rasterizer/rasterizer.h:32-34, dsr/rasterize_points.cu:31-37). */
typedef char* (*g4s_resize_fn)(void* ctx, size_t nbytes);

/* Message of the last error raised on the calling thread ("" if none). */
const char* g4s_last_error(void);

/* Version / build identification: "g4s-hip <semver> gfx950 build <id>"; <id> = first 12 hex digits of the SHA-256 of the
 * library's sources (csrc/Makefile), so that profiles and bench lines can name the binary they were taken on. */
const char* g4s_version(void);

/*
 * Forward rasterisation.  Replaces CudaRasterizer::Rasterizer::forward
 * (dsr/cuda_rasterizer/rasterizer.h:30-56, rasterizer_impl.cu:198-342).
 *
 *   P, D, M          #Gaussians, active SH degree (0..3), SH coefficients per Gaussian in memory
 *   background[3], width, height
 *   means3D[P,3], shs[P,M,3] or NULL, colors_precomp[P,3] or NULL, opacities[P],
 *   scales[P,2] or NULL, scale_modifier, rotations[P,4] (w,x,y,z) or NULL,
 *   transMat_precomp[P,9] or NULL, viewmatrix[16], projmatrix[16], cam_pos[3],
 *   tan_fovx, tan_fovy, prefiltered (accepted, ignored: the reference only traps on it)
 *   out_color[3,H,W], out_others[7,H,W] (channel map dsr/cuda_rasterizer/auxiliary.h:23-27),
 *   radii[P] int32 or NULL
 *   debug != 0: synchronise and check after every launch (CHECK_CUDA, auxiliary.h:295-302)
 *
 * Outputs need no pre-initialisation (the library writes every element).
 * Returns num_rendered (>= 0; the number of (Gaussian, tile) instances, identical to the
 * reference's) or a negative G4S_ERR_*.  Contains ONE host synchronisation (reading
 * num_rendered to size the binning chunk), exactly like the reference
 * (rasterizer_impl.cu:281-282).
 * Limit: at most 65535 tiles (1 048 560 pixels) across and 32767 tiles (524 272 pixels) down.
 */
int g4s_rasterizer_forward(
    g4s_resize_fn geometry_buffer, void* geometry_ctx,
    g4s_resize_fn binning_buffer, void* binning_ctx,
    g4s_resize_fn image_buffer, void* image_ctx,
    int P, int D, int M,
    const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* transMat_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
    float tan_fovx, float tan_fovy, int prefiltered,
    float* out_color, float* out_others, int* radii, int debug, void* stream);

/* Bytes of transient workspace g4s_rasterizer_backward needs for a forward that
 * returned R (per-instance gradient records, 80 B + one validity byte each, + a 256 KB tile list). */
size_t g4s_rasterizer_backward_workspace(int P, int R);

/*
 * Backward.  Replaces CudaRasterizer::Rasterizer::backward
 * (dsr/cuda_rasterizer/rasterizer.h:58-85, rasterizer_impl.cu:346-448).
 *
 *   geom_buffer / binning_buffer / image_buffer : the chunks the forward call filled
 *   R : the forward's return value
 *   dL_dpix[3,H,W], dL_depths[7,H,W] : cotangents of out_color / out_others
 *   workspace : >= g4s_rasterizer_backward_workspace(P,R) bytes, contents undefined
 *   outputs (all fully written, no pre-zeroing needed; rows of invisible Gaussians = 0):
 *     dL_dmean2D[P,3]  (densification surrogate, backward.cu:637-640; .z = 0)
 *     dL_dnormal[P,3] (view-space normal gradient; may be NULL, the reference's binding never
 *     returns it), dL_dopacity[P], dL_dcolor[P,3], dL_dmean3D[P,3],
 *     dL_dtransMat[P,9] (may be NULL: an intermediate nobody reads unless T matrices were handed in precomputed),
 *     dL_dsh[P,M,3] (if M > 0), dL_dscale[P,2], dL_drot[P,4]
 * Deterministic (no floating-point atomics), unlike the reference.
 * Returns G4S_OK or a negative G4S_ERR_*.
 */
int g4s_rasterizer_backward(
    int P, int D, int M, int R,
    const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp,
    const float* scales, float scale_modifier, const float* rotations, const float* transMat_precomp,
    const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy, const int* radii,
    char* geom_buffer, char* binning_buffer, char* image_buffer,
    const float* dL_dpix, const float* dL_depths,
    float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
    float* dL_dtransMat, float* dL_dsh, float* dL_dscale, float* dL_drot,
    char* workspace, size_t workspace_bytes, int debug, void* stream);

/*
 * Split-SH variants (extension; same semantics and results as the two calls above).  The reference's
 * GaussianModel stores the SH coefficients as two parameters, _features_dc [P,1,3] and _features_rest [P,M-1,3]
 * (2dgs/scene/gaussian_model.py: get_features = cat(_features_dc, _features_rest)), and concatenates them before
 * every render; the backward then slices dL_dsh apart again.  These entry points read the two tensors where they
 * are and write the two gradients separately, which removes 4 x P x M x 12 bytes of copy traffic per training
 * iteration (1.15 GB at P = 1.5 M, M = 16).  M counts all coefficients (dc + rest); colours always come from SH.
 */
int g4s_rasterizer_forward_split_sh(
    g4s_resize_fn geometry_buffer, void* geometry_ctx,
    g4s_resize_fn binning_buffer, void* binning_ctx,
    g4s_resize_fn image_buffer, void* image_ctx,
    int P, int D, int M,
    const float* background, int width, int height,
    const float* means3D, const float* sh_dc, const float* sh_rest, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* transMat_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
    float tan_fovx, float tan_fovy, int prefiltered,
    float* out_color, float* out_others, int* radii, int debug, void* stream);

int g4s_rasterizer_backward_split_sh(
    int P, int D, int M, int R,
    const float* background, int width, int height,
    const float* means3D, const float* sh_dc, const float* sh_rest,
    const float* scales, float scale_modifier, const float* rotations, const float* transMat_precomp,
    const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy, const int* radii,
    char* geom_buffer, char* binning_buffer, char* image_buffer,
    const float* dL_dpix, const float* dL_depths,
    float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
    float* dL_dtransMat, float* dL_dsh_dc, float* dL_dsh_rest, float* dL_dscale, float* dL_drot,
    char* workspace, size_t workspace_bytes, int debug, void* stream);

/*
 * Backward that ACCUMULATES (extension; the reference has no counterpart -- it trains one view per optimiser step,
 * train_with_refine_depth.py:373-378, and leaves the summing of several views to autograd's dense `grad += g`).
 * For multi-view batches (gradient accumulation; SURVEY.md 8(e)'s 8 views over fewer than 8 GPUs).  With
 * first_view == 0 the five PARAMETER gradients -- dL_dmean3D, dL_dopacity, dL_dsh (or dL_dsh_dc / dL_dsh_rest),
 * dL_dscale, dL_drot -- are ADDED to what the tensors hold, rows of Gaussians the view does not see are not touched, and
 * nothing is zero-filled: per view that is V x 232 B of read-modify-write instead of a P x 232 B write plus autograd's
 * 3 x P x 232 B dense add.  With first_view != 0 the call writes every row exactly like
 * g4s_rasterizer_backward[_split_sh] (the first view of a batch starts the sums).  The per-view outputs (dL_dmean2D,
 * dL_dcolor, dL_dtransMat) are overwritten either way.
 *
 *   sh_dc, sh_rest      sh_rest == NULL: sh_dc is the packed [P,M,3] tensor and dL_dsh_dc the packed [P,M,3] gradient;
 *                       otherwise the split layout of g4s_rasterizer_backward_split_sh
 *   view_stats          [P,2] or NULL: the densification statistics of the batch, (sum over the views of THIS VIEW's
 *                       ||dL_dmean2D.xy|| -- the reference accumulates a norm per view, gaussian_model.py:649-651, not the
 *                       norm of a sum --, number of views that saw the Gaussian); written when first_view, added otherwise
 *   after_event         hipEvent_t or NULL.  The sums are ordered by the caller: views of a batch may be in flight on
 *                       different streams (their blend kernels overlap), but the accumulating per-Gaussian kernel of
 *                       view j must run after that of view j-1.  If non-NULL, `stream` waits for this event -- recorded
 *                       by the caller behind the previous view's backward -- between the blend backward and the
 *                       per-Gaussian kernel, so only the latter is serialised.  Sums in a fixed order are bit-reproducible.
 * Everything else as g4s_rasterizer_backward_split_sh.  Colours come from SH (no colors_precomp).
 */
int g4s_rasterizer_backward_accumulate(
    int P, int D, int M, int R,
    const float* background, int width, int height,
    const float* means3D, const float* sh_dc, const float* sh_rest,
    const float* scales, float scale_modifier, const float* rotations, const float* transMat_precomp,
    const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy, const int* radii,
    char* geom_buffer, char* binning_buffer, char* image_buffer,
    const float* dL_dpix, const float* dL_depths,
    float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
    float* dL_dtransMat, float* dL_dsh_dc, float* dL_dsh_rest, float* dL_dscale, float* dL_drot,
    float* view_stats, int first_view,
    char* workspace, size_t workspace_bytes, void* after_event, int debug, void* stream);

/*
 * g4s_rasterizer_backward_accumulate that ALSO leaves the parameter gradients of the view PACKED (extension, for the
 * multi-GPU owner exchange of SURVEY.md 8(e): the rows a rank sends to the owners of its visible Gaussians).  The
 * per-Gaussian kernel writes, next to the tensors, one row per Gaussian with radii > 0, in index order:
 *     [dL_dmean3D 3 | dL_dsh 3 M (all coefficients, dc first) | dL_dopacity 1 | dL_dscale 2 | dL_drot 4 |
 *      view_stats 2 (||dL_dmean2D.xy||, 1) | bits(Gaussian index) 1]  =  3 M + 13 floats
 * -- the buffer layout of g4s_pack_rows(mode bits 1 | 3) over those tensors, so that the exchange no longer needs its pack
 * launch (0.08 ms at 1.5 M surfels, on the critical path between the backward and the all_to_all).
 *   packed->rows        [capacity, 3 M + 13] floats; rows beyond `capacity` are dropped (the caller compares the count it
 *                       knows with its capacity)
 *   packed->block_offs  DEVICE uint32 per block of 256 consecutive Gaussians: the number of Gaussians with radii > 0 in
 *                       the blocks before it (the caller has the visible set since the forward; it is read when the
 *                       per-Gaussian kernel runs, i.e. it may still be in flight on `stream` when the call is issued)
 * Only for first_view != 0 (a later view of a batch updates the sums of rows it sees, not of every row); packed == NULL
 * or packed->rows == NULL: exactly g4s_rasterizer_backward_accumulate.
 */
typedef struct g4s_packed_rows {
    float* rows;
    const uint32_t* block_offs;
    long long capacity;
} g4s_packed_rows;
int g4s_rasterizer_backward_accumulate_packed(
    int P, int D, int M, int R,
    const float* background, int width, int height,
    const float* means3D, const float* sh_dc, const float* sh_rest,
    const float* scales, float scale_modifier, const float* rotations, const float* transMat_precomp,
    const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy, const int* radii,
    char* geom_buffer, char* binning_buffer, char* image_buffer,
    const float* dL_dpix, const float* dL_depths,
    float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
    float* dL_dtransMat, float* dL_dsh_dc, float* dL_dsh_rest, float* dL_dscale, float* dL_drot,
    float* view_stats, int first_view, const g4s_packed_rows* packed,
    char* workspace, size_t workspace_bytes, void* after_event, int debug, void* stream);

/*
 * The forward WITHOUT its host synchronisation (extension; same results as g4s_rasterizer_forward[_split_sh]).
 * The reference -- and the two entry points above -- read num_rendered back to size the binning chunk
 * (rasterizer_impl.cu:281-282): the host stalls until the GPU has drained everything queued before the call, every
 * frame.  Here the caller fixes the chunks up front:
 *
 *   instance_capacity : the largest number of binned (Gaussian, tile) instances the call may produce
 *   geom / binning / image buffers : at least the geom_bytes / binning_bytes / image_bytes that
 *                       g4s_rasterizer_layout(P, instance_capacity, width, height) reports
 *   status_dev[4]     : DEVICE words written by the call: [0] num_rendered (the reference's count), [1] instances
 *                       binned, [2] Gaussians that emit instances, [3] 1 if [1] > instance_capacity -- the frame
 *                       is then incomplete (memory-safe, the surplus instances are dropped) and the caller must
 *                       repeat it with a larger capacity.  Read them whenever convenient (e.g. once per N frames).
 *   sh_rest           : NULL = `shs` is the packed [P,M,3] tensor; otherwise shs = [P,1,3], sh_rest = [P,M-1,3]
 *
 * Nothing is read back and no launch depends on a host-side count: the call returns as soon as its ~20 launches are
 * queued.  The matching backward is g4s_rasterizer_backward[_split_sh] with R = instance_capacity (R only sizes the
 * layout of the chunks and the workspace).  Returns G4S_OK or a negative G4S_ERR_*.
 */
int g4s_rasterizer_forward_presized(
    char* geom_buffer, size_t geom_bytes, char* binning_buffer, size_t binning_bytes, char* image_buffer, size_t image_bytes,
    int instance_capacity, uint32_t* status_dev,
    int P, int D, int M,
    const float* background, int width, int height,
    const float* means3D, const float* shs, const float* sh_rest, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* transMat_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
    float tan_fovx, float tan_fovy,
    float* out_color, float* out_others, int* radii, int debug, void* stream);

/* Near-plane visibility.  Replaces CudaRasterizer::Rasterizer::markVisible
 * (dsr/cuda_rasterizer/rasterizer.h:24-29, rasterizer_impl.cu:54-66,141-153).
 * present[P] is one byte per Gaussian (bool). */
int g4s_rasterizer_mark_visible(int P, const float* means3D, const float* viewmatrix,
                                const float* projmatrix, uint8_t* present, void* stream);

/* Bytes of scratch g4s_knn_mean_dist needs for P points. */
size_t g4s_knn_workspace(int P);

/* Mean squared distance to the three nearest other points.  Replaces SimpleKNN::knn
 * (knn/simple_knn.h:18, knn/simple_knn.cu:185-221); unlike the reference it allocates
 * nothing itself: `workspace` must hold g4s_knn_workspace(P) bytes.
 * points[P,3], meanDists[P].  No host synchronisation (the bounding box stays on the device). */
int g4s_knn_mean_dist(int P, const float* points, float* meanDists, char* workspace,
                      size_t workspace_bytes, void* stream);

/* ---- diagnostics (used by the parity tests to compare stage by stage) ---------------- */

/* Describes where the forward call put its private state inside the three chunks so a
 * test can read it back.  Offsets are relative to the chunk base returned by the resize
 * callbacks.  Not part of the drop-in surface. */
typedef struct g4s_layout {
    /* geometry chunk */
    size_t rec;          /* P x 32 floats: xy, inst_off(u32), count(u32) | bit 31 REC_AFFINE, normal, opacity, Tu,Tv,Tw -- or, for
                            REC_AFFINE splats, the affine ray-splat intersection A', B', Dc' --, rgb, box, Tw.z, cutoff ellipse
                            (csrc/g4s_internal.h) */
    size_t clamped;      /* P x u8 (bit c = channel c clamped) */
    size_t depth_sorted; /* u32 indices of the Gaussians that emit instances, in (depth, index) order */
    size_t tiles_touched;/* P x u32 */
    size_t geom_bytes;
    /* binning chunk */
    size_t entries;      /* R x u64 sorted instances: tile<<32 | idx */
    size_t qhit;         /* one byte per sorted instance: bit q = quadrant q of its tile blended it */
    size_t binning_bytes;
    /* image chunk */
    size_t ranges;       /* tiles x (u32 start, u32 end) */
    size_t final_T;      /* 3N floats: T, M1, M2 */
    size_t n_contrib;    /* 2N u32: last contributor, median contributor */
    size_t tile_order;   /* tiles x u32: workgroup id -> tile, longest instance list first */
    size_t image_bytes;
    size_t hot_count;    /* (image chunk) u32: tiles the last backward over this state handed to the four-wave kernel */
} g4s_layout;

int g4s_rasterizer_layout(int P, int R, int width, int height, g4s_layout* out);

/* Diagnostic switches of the parity tests -- process-wide integers held by the library, default 0 / unset.  They are
 * set through this call only: the library never reads the environment on a call path.  None of them changes a result
 * beyond rounding (the tests assert exactly that):
 *   "box_only"           forward skips quadrants by the bounding box only, not by the exact cutoff region
 *   "no_fastpath"        no splat is certified REC_AFFINE: every (pixel, splat) pair is evaluated with the reference's
 *                        own arithmetic (results equal the default's to rounding, not bit for bit)
 *   "bwd_fwd_order"      the blend backward walks the tiles in the forward's order
 *   "bwd_hot_threshold"  list depth above which a tile goes to the four-wave backward (G4S_OPTION_UNSET = automatic)
 *   "no_side_zero"       the blend backward does not clear dL_dsh on the side (K8 clears the rows it skips)
 * Returns G4S_OK or G4S_ERR_INVALID_ARGUMENT for an unknown name. */
#define G4S_OPTION_UNSET (-2147483647 - 1)
int g4s_set_option(const char* name, int value);
int g4s_get_option(const char* name, int* value);

/*
 * Row packing for the multi-GPU gradient exchange (new functionality; no reference
 * counterpart): gathers the rows `row_index[0..n)` of up to 8 row-major float segments
 * (segment s = [P, widths[s]]) into one contiguous buffer, or scatters such a buffer back.  `mode`:
 *   bit 0   0 = pack (rows -> buffer), 1 = unpack (buffer -> rows)
 *   bit 1   buffer layout: 0 = segment after segment (packed = [n*widths[0] | n*widths[1] | ...]),
 *           1 = row-major [n, sum(widths)] (row ranges of the buffer are contiguous: all_to_all splits)
 *   bit 2   unpack adds to the rows instead of overwriting them (indices must be distinct)
 *   bit 3   (row-major only) a buffer row is sum(widths) + 1 floats: the last one carries the row's index as int32
 *           bits -- pack writes it, unpack reads it INSTEAD of `row_index` (which may be NULL then): rows and their
 *           indices travel in one all_to_all
 * `segments` / `widths` are HOST arrays of device pointers / ints; `row_index` is a device int64 array.
 */
int g4s_pack_rows(int nseg, float* const* segments, const int* widths, const long long* row_index, int n,
                  float* packed, int mode, void* stream);

/*
 * The owner's side of that exchange in ONE launch (new functionality): `packed` holds, back to back, the row-major
 * rows with index column (g4s_pack_rows mode bits 1 | 3 layout) received from `nsrc` sources -- source i's rows are
 * [src_offsets[i], src_offsets[i] + src_counts[i]), every index in [row_lo, row_hi) (the owner's shard), ASCENDING inside
 * a source and distinct inside a source.  Adds them to the segments' rows, source after source: element for element the
 * same sequence of additions -- the same bits -- as one g4s_pack_rows(mode 15) call per source in that order, with one
 * read and one write of the shard instead of one per source.  `src_offsets` / `src_counts` are HOST arrays.
 */
int g4s_accumulate_rows(int nseg, float* const* segments, const int* widths, int nsrc, const int* src_offsets,
                        const int* src_counts, const float* packed, int row_lo, int row_hi, void* stream);

/*
 * The same with the owner's own contribution (what the segments' rows hold on entry) taking position `own_position` in the
 * order of additions instead of the first: every element becomes ((0 + s_0 + ... + s_{k-1}) + own) + s_k + ... + s_{n-1},
 * k = own_position (0 <= k <= nsrc <= 8; k = 0 is g4s_accumulate_rows).  With the sources in rank order and k = the owner's
 * rank, a row is summed in RANK ORDER whichever rank owns it -- the order in which one process accumulating the same views one
 * after the other sums them, so an N-rank step of the exchange reproduces single-process gradient accumulation bit for bit
 * (g4splat_amd/parallel.py: OwnerReduce.rank_order).  Same traffic as g4s_accumulate_rows.
 */
int g4s_accumulate_rows_ordered(int nseg, float* const* segments, const int* widths, int nsrc, const int* src_offsets,
                                const int* src_counts, const float* packed, int row_lo, int row_hi, int own_position,
                                void* stream);

/* Optional per-kernel timing with HIP events recorded on the launch stream (used by bench.py
 * for the roofline figure; off by default, process-wide).  Kernel groups 0..g4s_profile_kernels()-1
 * are named by g4s_profile_name().  g4s_profile_read() synchronises on the recorded events and
 * returns the summed duration in milliseconds and the number of recordings. */
void g4s_profile_enable(int on);
int g4s_profile_kernels(void);
const char* g4s_profile_name(int id);
int g4s_profile_read(int id, double* total_ms, int* count);
void g4s_profile_reset(void);

#ifdef __cplusplus
}
#endif
#endif /* G4S_RASTERIZER_H_INCLUDED */
