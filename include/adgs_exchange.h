/*
 * adgs_exchange.h -- C ABI of the factored SH-gradient exchange for camera-parallel training
 * (libadgs_hip.so).  No counterpart in the reference, which is single-GPU / one camera per
 * iteration (train.py:55-61,74); SURVEY.md section 8(e) defines the multi-GPU path this serves.
 *
 * Every SH-related parameter gradient of one camera's backward is a multiple of ONE 3-vector per
 * Gaussian, the clamp-masked colour gradient g = dL/dRGB * (1 - clamped)
 * (computeColorFromSH backward, RAST/cuda_rasterizer/backward.cu:20-139):
 *     dL/d shs_dc[m, 0, :]        = SH_C0 * g
 *     dL/d shs_rest[m, k-1, :]    = b_k(dir) * g,   dir = normalize(mean_m(t) - campos),  k = 1 .. (deg+1)^2 - 1
 *     dL/d shs_deform[m, ch, j]   = w_j(t) * SH_C0 * g[ch]     (f_shs is linear in its parameters,
 *                                                               utils/func_utils.py:121-156; shs = cat(dc + f_shs(t), rest),
 *                                                               scene/gaussian_model.py:198-205)
 * i.e. 84 floats per Gaussian (M = 16, C = 12) that are functions of 3.  Data-parallel ranks therefore exchange g
 * (12 B per Gaussian and camera, adgs_sh_grads.rgb_factor of adgs_raster_backward_rawsh) with an all-gather instead of
 * all-reducing the expanded rows (336 B per Gaussian), and every rank expands and sums all cameras locally with
 * adgs_sh_grad_expand -- in camera order, so the result is bit-identical on every rank.
 *
 * All pointers are device pointers to fp32 data unless stated; NULL = absent.
 */
#ifndef ADGS_EXCHANGE_H
#define ADGS_EXCHANGE_H
#include <stddef.h>
#include <stdint.h>
#include "adgs_rasterizer.h"
#ifdef __cplusplus
extern "C" {
#endif

#define ADGS_EXPAND_MAX_CAMS 32

/* One camera's contribution. */
typedef struct adgs_sh_expand_cam {
	const float* rgb;        /* [P,3] clamp-masked colour gradient of this camera (all-zero rows: not visible / no contribution) */
	const float* xyz_tail;   /* [P-row0,3] this camera's means3D for Gaussians >= row0 (the time-dependent ones); NULL iff row0 == P */
	float campos[3];         /* camera centre (GaussianRasterizationSettings.campos) */
	float reserved;
} adgs_sh_expand_cam;

/* Sum over n_cams cameras of the expanded SH gradients, written (not accumulated) to `out` in the raw tensors' layout:
 *   out->scene_dc [Ns,1,3], out->obj_dc [P-Ns,1,3], out->scene_rest [Ns,M-1,3], out->obj_rest [P-Ns,M-1,3],
 *   out->scene_deform [Ns,3,C], out->obj_deform [P-Ns,3,C]   (NULL = not wanted; out->rgb_factor is ignored).
 * cams      host array [n_cams], n_cams <= ADGS_EXPAND_MAX_CAMS
 * W         device [n_cams, C]: W[c][j] = d f_shs(t_c) / d param[..., j] (dense; 0 for columns outside the active B-spline window);
 *           NULL iff C == 0 or no deform gradient is wanted
 * xyz_head  [row0,3] means3D of the Gaussians < row0, identical for every camera (the static scene Gaussians when there is no
 *           background deformation); row0 == 0: every camera supplies all means in xyz_tail
 * D         active SH degree (0..3), M = (max degree + 1)^2 coefficients per Gaussian in the raw tensors. */
int adgs_sh_grad_expand(int n_cams, const adgs_sh_expand_cam* cams, const float* W, int C,
	int P, int Ns, int row0, const float* xyz_head, int D, int M, const adgs_sh_grads* out, void* stream);

/* The same for a parameter tensor [count, 3, C] that enters its output linearly (f_xyz: utils/func_utils.py:121-156,
 * scene/gaussian_model.py:173-185): out[m, d, j] = scale * sum_e W[e][j] * g[e][m, d] over n_terms <= ADGS_EXPAND_MAX_CAMS factors.
 * In the factored exchange: xyz_deform_param's gradient from every camera's two upstream position gradients of the object range
 * (camera time and flow time), g = HOST array of n_terms device pointers to [count,3], W device [n_terms, C] (dense basis rows). */
int adgs_lin_grad_expand(int n_terms, const float* const* g, const float* W, int C, int count, float scale, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
