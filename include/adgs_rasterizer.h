/*
 * adgs_rasterizer.h -- C ABI of the MI355X-native AD-GS hot path (libadgs_hip.so).
 *
 * Drop-in boundary for the reference's native extension entry points.  Every
 * function takes plain device pointers and sizes (no torch types) and enqueues
 * its work on the HIP stream passed as `stream` (a hipStream_t, 0 = default
 * stream).  "RAST/" = submodules/depth-diff-gaussian-rasterization, "KNN/" =
 * submodules/simple-knn of the reference.
 *
 * Return value convention: >= 0 success (adgs_raster_forward returns
 * num_rendered), < 0 failure; adgs_last_error() then returns a thread-local,
 * NUL-terminated description.  The reference throws std::runtime_error instead
 * (RAST/cuda_rasterizer/auxiliary.h:166-173, rasterizer_impl.cu:249-252).
 *
 * All float data is fp32.  Optional inputs are NULL (the reference passes the
 * data pointer of an empty tensor, RAST/diff_gaussian_rasterization/__init__.py:220-236).
 */
#ifndef ADGS_RASTERIZER_H
#define ADGS_RASTERIZER_H

#include <stddef.h>
#include <stdint.h>
#include "adgs_deform.h"   /* adgs_func_eval */

#ifdef __cplusplus
extern "C" {
#endif

/* Replaces std::function<char*(size_t)> (RAST/cuda_rasterizer/rasterizer.h:26-28,
 * RAST/rasterize_points.cu:27-33): must return a device buffer of at least
 * `bytes` bytes that stays valid until the matching backward call. */
typedef char* (*adgs_alloc_fn)(void* user, size_t bytes);

const char* adgs_last_error(void);

/* Library / device probe; returns 0 when a gfx950 device is usable. */
int adgs_device_check(void);

/* CudaRasterizer::Rasterizer::forward  (RAST/cuda_rasterizer/rasterizer.h:31-62,
 * rasterizer_impl.cu:198-352).  Outputs must be zero-initialised by the caller
 * (RAST/rasterize_points.cu:82-87).  Returns num_rendered. */
int adgs_raster_forward(
	adgs_alloc_fn geometryBuffer, void* geometryUser,
	adgs_alloc_fn binningBuffer, void* binningUser,
	adgs_alloc_fn imageBuffer, void* imageUser,
	int P, int D, int M, int D_S,
	const float* background,
	int width, int height,
	const float* means3D,
	const float* shs,
	const float* colors_precomp,
	const float* flow_points,
	const float* semantic,
	const float* opacities,
	const float* scales,
	float scale_modifier,
	const float* rotations,
	const float* cov3D_precomp,
	const float* viewmatrix,
	const float* projmatrix,
	const float* cam_pos,
	float tan_fovx, float tan_fovy,
	int prefiltered,
	float* out_color,
	float* out_depth,
	float* img_opacity,
	float* img_flow,
	float* img_semantic,
	int inv_depth,
	int* radii,
	int debug,
	void* stream);

/* The forward-only render: adgs_raster_forward's arguments, images and radii -- bit for bit -- with nothing kept for a backward: no replay
 * lists, no per-pixel contributor counts, no accumulator lines, no tile order; the three buffers are scratch.  What the reference runs under
 * torch.no_grad() for evaluation and reports as its render FPS (render.py:52-55,86,156) -- there through the same forward() as training.
 * Frames with more than one semantic channel publish their lists all the same (channels 1.. are blended by a replay of them). */
int adgs_raster_render(
	adgs_alloc_fn geometryBuffer, void* geometryUser,
	adgs_alloc_fn binningBuffer, void* binningUser,
	adgs_alloc_fn imageBuffer, void* imageUser,
	int P, int D, int M, int D_S,
	const float* background,
	int width, int height,
	const float* means3D,
	const float* shs,
	const float* colors_precomp,
	const float* flow_points,
	const float* semantic,
	const float* opacities,
	const float* scales,
	float scale_modifier,
	const float* rotations,
	const float* cov3D_precomp,
	const float* viewmatrix,
	const float* projmatrix,
	const float* cam_pos,
	float tan_fovx, float tan_fovy,
	int prefiltered,
	float* out_color,
	float* out_depth,
	float* img_opacity,
	float* img_flow,
	float* img_semantic,
	int inv_depth,
	int* radii,
	int debug,
	void* stream);

/* CudaRasterizer::Rasterizer::backward  (RAST/cuda_rasterizer/rasterizer.h:64-103,
 * rasterizer_impl.cu:356-476).  All dL_* outputs must be zero-initialised by the
 * caller (RAST/rasterize_points.cu:195-206) unless adgs_raster_needs_zero_init() says
 * otherwise.  In that (default, v2) pipeline dL_dconic, dL_dcolor, dL_ddepth and dL_dcov3D
 * -- intermediates of the reference that never leave its C++ layer except as the
 * gradients of colors_precomp / cov3D_precomp -- may be NULL (not written).  Returns 0. */
int adgs_raster_backward(
	int P, int D, int M, int R, int D_S,
	const float* background,
	int width, int height,
	const float* means3D,
	const float* shs,
	const float* colors_precomp,
	const float* flow_points,
	const float* semantic,
	const float* scales,
	float scale_modifier,
	const float* rotations,
	const float* cov3D_precomp,
	const float* viewmatrix,
	const float* projmatrix,
	const float* campos,
	float tan_fovx, float tan_fovy,
	const int* radii,
	char* geom_buffer,
	char* binning_buffer,
	char* img_buffer,
	const float* dL_dpix,
	const float* dL_dpix_depth,
	const float* dL_dpix_flow,
	const float* dL_dpix_semantic,
	float* dL_dmean2D,
	float* dL_dconic,
	float* dL_dopacity,
	float* dL_dcolor,
	float* dL_ddepth,
	float* dL_dmean3D,
	float* dL_dcov3D,
	float* dL_dsh,
	float* dL_dscale,
	float* dL_drot,
	float* dL_dflow,
	float* dL_dsemantic,
	const float* grad_img_opacity,
	const float* img_opacity,
	int inv_depth,
	int debug,
	void* stream);

/* ---- "raw SH" fast path (no counterpart in the reference) ----------------
 * Same operators, but the SH coefficients are read straight from the reference GaussianModel's raw
 * tensors -- scene || object, coefficient 0 = dc + f_shs(t), coefficients 1.. = rest
 * (scene/gaussian_model.py:198-205) -- instead of a materialised [P, M, 3] tensor, and the SH
 * gradients are written straight into the layout of those raw tensors (every element written).
 * Needs the default pipeline (not ADGS_RASTER_MODE=classic); scales/rotations only. */
typedef struct adgs_sh_source {
	int32_t Ns;                               /* scene Gaussians come first; No = P - Ns */
	const float *scene_dc, *obj_dc;           /* [Ns,1,3] [No,1,3] */
	const float *scene_rest, *obj_rest;       /* [Ns,M-1,3] [No,M-1,3] */
	const float *scene_deform, *obj_deform;   /* shs_deform_param_{scene,obj} [.,3,C] or NULL */
	adgs_func_eval f;                         /* f_shs evaluated at the camera time (n_params = C) */
	/* Raw scene geometry (all four or none; NULL = off): the Gaussians idx < Ns take their position, log-scale, raw rotation and
	 * opacity logit from these RAW tensors (_scene_xyz [Ns,3], _scene_scaling [Ns,3], _scene_rotation [Ns,4], _scene_opacity
	 * [Ns,1]) and the preprocess applies exp / normalize / sigmoid itself (scene/gaussian_model.py:89-152); rows idx < Ns of
	 * means3D / scales / rotations / opacities / flow_points are then never read (their flow point is the position itself), and
	 * the backward writes the gradients of the raw tensors (adgs_sh_grads) instead of rows idx < Ns of dL_dmean3D / dL_dscale /
	 * dL_drot / dL_dopacity / dL_dflow. */
	const float *scene_xyz, *scene_scaling, *scene_rotation, *scene_opacity;
	/* Per-pixel background [3,H,W] (NULL = the constant `background` colour): the blend epilogue writes out_color = C + T * bg_image,
	 * i.e. the reference's composite `render = foreground + (1 - img_opacity) * background` of the environment map
	 * (gaussian_renderer/__init__.py:93-94, scene/env.py:43-76) without a pass of its own; the backward then uses the per-pixel
	 * background in its `bg . dL/dC` term and writes dL/dbg_image = T_final * dL/dC (adgs_sh_grads.bg_image). */
	const float *bg_image;
} adgs_sh_source;
struct adgs_sh_adam;      /* adgs_optim.h */
typedef struct adgs_sh_grads {
	uint64_t struct_bytes;   /* sizeof(adgs_sh_grads) as the CALLER was compiled: members beyond it are taken as NULL, so a caller built against an
	                            older header (before `adam`) keeps working; 0 or less than the first seven pointers is an error */
	float *scene_dc, *obj_dc, *scene_rest, *obj_rest, *scene_deform, *obj_deform;   /* NULL = not wanted */
	float *rgb_factor;   /* [P,3] or NULL: the clamp-masked colour gradient dL/dRGB * (1 - clamped) (backward.cu:20-139), 0 for
	                        Gaussians with radii == 0 -- the one per-camera vector every SH gradient row above is a multiple of
	                        (adgs_exchange.h: data-parallel ranks exchange this instead of the expanded rows) */
	float *scene_xyz, *scene_scaling, *scene_rotation, *scene_opacity;   /* raw scene geometry gradients (required when the
	                        source carries raw scene geometry): every row written */
	float *bg_image;     /* [3,H,W] gradient of the per-pixel background or NULL (not wanted) */
	/* NULL, or the Adam step to apply in place of storing gradients (adgs_optim.h: adgs_sh_adam).  For every slot that is on, the
	 * gradient destination of the same name above must be NULL (the gradient is never materialised); the deformation slots need the
	 * dc destinations (their rows are multiples of dL/d(dc)).  Refused while the stream is being captured into a graph (the bias
	 * corrections of step t are launch arguments). */
	const struct adgs_sh_adam* adam;
} adgs_sh_grads;

int adgs_raster_forward_rawsh(
	adgs_alloc_fn geometryBuffer, void* geometryUser, adgs_alloc_fn binningBuffer, void* binningUser,
	adgs_alloc_fn imageBuffer, void* imageUser,
	int P, int D, int M, int D_S, const float* background, int width, int height,
	const float* means3D, const adgs_sh_source* sh, const float* flow_points, const float* semantic,
	const float* opacities, const float* scales, float scale_modifier, const float* rotations,
	const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
	float* out_color, float* out_depth, float* img_opacity, float* img_flow, float* img_semantic,
	int inv_depth, int* radii, int debug, void* stream);

/* ... and its forward-only form (see adgs_raster_render) */
int adgs_raster_render_rawsh(
	adgs_alloc_fn geometryBuffer, void* geometryUser, adgs_alloc_fn binningBuffer, void* binningUser,
	adgs_alloc_fn imageBuffer, void* imageUser,
	int P, int D, int M, int D_S, const float* background, int width, int height,
	const float* means3D, const adgs_sh_source* sh, const float* flow_points, const float* semantic,
	const float* opacities, const float* scales, float scale_modifier, const float* rotations,
	const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
	float* out_color, float* out_depth, float* img_opacity, float* img_flow, float* img_semantic,
	int inv_depth, int* radii, int debug, void* stream);

int adgs_raster_backward_rawsh(
	int P, int D, int M, int R, int D_S, const float* background, int width, int height,
	const float* means3D, const adgs_sh_source* sh, const float* flow_points, const float* semantic,
	const float* scales, float scale_modifier, const float* rotations,
	const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
	const int* radii, char* geom_buffer, char* binning_buffer, char* img_buffer,
	const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dpix_flow, const float* dL_dpix_semantic,
	float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D,
	float* dL_dcov3D, const adgs_sh_grads* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dflow, float* dL_dsemantic,
	const float* grad_img_opacity, const float* img_opacity, int inv_depth, int debug, void* stream);

/* The reference requires the caller to zero-fill every output of forward and backward
 * (rasterize_points.cu:82-87,195-206).  Zero-filled outputs are always accepted; this returns 0
 * when the pipeline selected for D_S semantic channels writes every element itself (the default
 * coarse-binned pipeline), so a caller may skip the fills, and 1 otherwise.  Exceptions that
 * must stay zero-filled when the corresponding input is absent: out_color (no shs/colours),
 * img_flow / dL_dflow (no flow_points), dL_dscale / dL_drot (cov3D_precomp given). */
int adgs_raster_needs_zero_init(int D_S);
/* ... and for the backward of a given forward (its state buffers, shape and point count): what THAT forward's pipeline needs, from the
 * library's frame table -- the backward never consults the environment.  Unknown state: 1 (zero-fill is always safe). */
int adgs_raster_backward_needs_zero_init(const char* geom_buffer, const char* img_buffer, int width, int height, int P);

/* CudaRasterizer::Rasterizer::markVisible (RAST/cuda_rasterizer/rasterizer.h:24-29,
 * rasterizer_impl.cu:141-153).  `present` is a bool (1 byte) array of length P. */
int adgs_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
	uint8_t* present, void* stream);

/* SimpleKNN::knn (KNN/simple_knn.h:17-20, simple_knn.cu:185-221): mean squared
 * distance to the 3 nearest other points.  `workspace` must hold
 * adgs_knn_workspace_bytes(P) bytes of device memory.  After the call, the 64-bit word at byte 24 of the workspace
 * holds the number of candidate boxes (1024 Morton-sorted points each, KNN/simple_knn.cu:12) the search scanned, summed over
 * the ceil(P / 1024) query boxes (a statistic for bench.py; device memory). */
size_t adgs_knn_workspace_bytes(int P);
int adgs_knn_dist2(int P, const float* points, float* meanDists, char* workspace, void* stream);

/* Statistics of the most recent adgs_raster_forward in this process
 * (P_visible, num_rendered, max tile-list length); host-side, for bench.py. */
typedef struct adgs_frame_stats {
	int64_t num_rendered;
	int32_t tiles;
	int32_t sort_bits;
	int32_t sort_passes;
	int32_t reserved;
	int64_t fine_pairs;       /* v2: sum over Gaussians of the fine tiles of their shrunk rectangle (bound of the blended pairs) */
} adgs_frame_stats;
void adgs_get_frame_stats(adgs_frame_stats* out);

/* Capacity status of the calling thread's most recent default-pipeline forward on the current device.
 * The reference sizes its binning buffer after a blocking device->host copy of the pair count in the middle of the forward
 * (RAST/cuda_rasterizer/rasterizer_impl.cu:288).  This library enqueues the whole forward against a capacity (previous
 * frames' counts + 25 %), compares the exact totals on the device, and only afterwards reads the totals the device published
 * to a host mailbox: a frame that did not fit blended nothing and is enqueued again with exact sizes before
 * adgs_raster_forward returns (`eager_reruns` counts these).  Under stream capture (HIP graphs) nothing can be read back
 * or re-enqueued: after a replay and a stream synchronisation the caller must check `overflow` (this frame) or
 * `unrepaired_overflow_count` (replayed frames that did not fit, since the mailbox exists; `overflow_count` also counts eager
 * frames, which repaired themselves); on overflow the captured frame's outputs are a defined EMPTY render (background, opacity 0,
 * no gradients), and an eager frame (which raises the capacity hints) followed by a new capture repairs it.  Statistics, capacity hints and the mailbox are kept per
 * (host thread, device): threads (or devices) rendering at the same time do not disturb each other; only the stage profiler below is process-wide. */
typedef struct adgs_frame_status {
	int64_t pairs;                 /* (cell, Gaussian) pairs of the frame */
	int64_t fine_pairs;            /* bound of the blended (tile, Gaussian) entries */
	int64_t capacity_pairs;        /* what the frame's launches were enqueued against (a replay: what its capture was enqueued against) */
	int64_t capacity_fine_pairs;
	int64_t overflow_count;        /* frames that did not fit, since the mailbox exists */
	int64_t eager_reruns;          /* eager forwards of this thread on this device that were enqueued twice */
	int32_t overflow;              /* 1: this frame did not fit its capacity */
	int32_t order_hint;            /* 1: the forward was handed the longest-first tile order an earlier forward of the same camera (same matrix
	                                  addresses, image shape) on the same stream left behind; 0: bottom-up (a camera's first render) */
	int64_t unrepaired_overflow_count;      /* overflow_count minus the eager frames this library re-enqueued itself: overflows of graph replays */
	int64_t order_hint_lookups;    /* forwards of this thread on this device that looked for a tile-order hint ... */
	int64_t order_hint_hits;       /* ... and found one written by an earlier forward: the hit rate tells whether the caller's cameras keep their addresses */
	int64_t fullest_slab_units;    /* bucket binning: entries of the frame's fullest depth slab in units of 4096 (1: every slab was sorted in one piece;
	                                  more: that slab was bisected and its cell streamed again -- the bounds did not fit this camera) */
} adgs_frame_status;
int adgs_get_frame_status(adgs_frame_status* out);

/* Optional per-stage timing with HIP events recorded on the launch stream (used by bench.py
 * for the roofline figure).  Process-wide.  adgs_profile_collect() must be called after the
 * stream has been synchronised; it ADDS into total_ms[] / counts[] (adgs_profile_num_stages()
 * entries each). */
void adgs_profile_enable(int stage_mask);   /* bit i set: time stage i (two HIP events per launch group, ~10 us of
                                               queue bubble each); 0 = off, -1 = every stage */
int adgs_profile_reserve(int n_events);     /* pre-create event objects (returns how many); a timed loop then never calls hipEventCreate */
int adgs_profile_num_stages(void);
const char* adgs_profile_stage_name(int stage);
int adgs_profile_collect(double* total_ms, int64_t* counts);

#ifdef __cplusplus
}
#endif
#endif /* ADGS_RASTERIZER_H */
