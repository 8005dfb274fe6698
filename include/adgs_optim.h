/*
 * adgs_optim.h -- C ABI of the fused optimizer step (libadgs_hip.so).  SURVEY.md section 8(f) row 1.
 *
 * Replaces the torch.optim.Adam(l, lr=0.0, eps=1e-15) step over the 18 parameter groups of
 * GaussianModel.training_setup (scene/gaussian_model.py:346-372, stepped at train.py:163-167):
 * ONE launch updates every group (the reference's multi-kernel foreach implementation makes
 * several passes over p, g, m, v), optionally zeroing the gradients in the same pass.
 *
 * Per element, identical to torch.optim.Adam without weight decay / amsgrad / maximize:
 *   m = m + (1 - beta1) (g - m);   v = beta2 v + (1 - beta2) g g
 *   p = p - (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 * All pointers are device pointers to fp32; `step` is t (1-based) and may differ per group.
 */
#ifndef ADGS_OPTIM_H
#define ADGS_OPTIM_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ADGS_ADAM_MAX_GROUPS 32

typedef struct adgs_adam_group {
	float* param;
	float* grad;            /* read; zeroed when zero_grad != 0 */
	float* exp_avg;
	float* exp_avg_sq;
	int64_t numel;
	float lr;
	int32_t step;           /* t of this parameter after the increment (>= 1) */
	/* Optional (NULL = off): one byte per tile of ADGS_ADAM_TILE consecutive elements, 0 = "every element of the tile has had
	 * g = m = v = 0 in every step so far".  For such a tile the Adam update is the identity, bit for bit (m and v stay 0, the
	 * parameter moves by step_size * 0 / (0 + eps) = 0): the kernel reads only the tile's gradient, and leaves everything else
	 * untouched while that is still all zero; the first non-zero gradient sets the byte to 1 for good.  Caller-owned, zero-
	 * initialised when the moments are created as zeros, ALL ONES if the moments come from anywhere else.  Pays for parameters
	 * most of which never receive a gradient: the 8192^2 environment map (scene/env.py) under a few dozen cameras.  (A tile is 1 KiB of a
	 * map row: 4096-element tiles -- half a row of that map, 180 degrees of azimuth -- left 5x more of it active than the cameras see.) */
	uint8_t* tile_active;
	/* Bit 0 (ADGS_ADAM_TILES_MARKED): whoever produced this gradient set tile_active[tile] = 1 for every tile it wrote a non-zero
	 * value into (adgs_envmap_backward_marked does), so a tile whose byte is 0 is not even read: an 8192^2 x 3 map costs 0.8 GB of
	 * gradient reads per step otherwise.  Only valid if NOTHING else has written into the gradient.
	 * Bit 1 (ADGS_ADAM_ZERO_GRAD): zero the gradient of the tiles that are updated, whatever adgs_adam_step's zero_grad says --
	 * with bit 0 the buffer is then all zero again and can take the next backward without a fill pass. */
	int32_t flags;
	int32_t reserved;
} adgs_adam_group;
#define ADGS_ADAM_TILES_MARKED 1
#define ADGS_ADAM_ZERO_GRAD 2
#define ADGS_ADAM_TILE 256

/* n_groups <= ADGS_ADAM_MAX_GROUPS per call (call again for more).  Returns 0, or a negative
 * code with adgs_last_error() set. */
int adgs_adam_step(const adgs_adam_group* groups, int n_groups, float beta1, float beta2, float eps, int zero_grad, void* stream);

/* The same step applied INSIDE the rasterizer's backward (adgs_sh_grads.adam, adgs_rasterizer.h), for the single-camera iteration
 * of train.py:47-167 (one backward, then optimizer.step() -- :163-167).  The kernels that produce the gradient of a raw SH tensor
 * hold each element of it in a register or in LDS exactly once: they read (p, m, v), apply the update above and write (p, m, v)
 * back instead of storing the gradient -- 24 instead of 4 + 28 bytes per parameter, no gradient buffer.  Per slot the arithmetic
 * is the arithmetic of adgs_adam_step on the gradient that would have been stored (bit for bit: tests/test_gpu_optim.py).
 * A slot with param == NULL is off (that tensor's gradient is written as usual). */
typedef struct adgs_adam_slot {
	float* param;           /* must be the tensor the frame reads (the same pointer as in adgs_sh_source) */
	float* exp_avg;
	float* exp_avg_sq;
	float lr;
	int32_t step;           /* t of this parameter after the increment (>= 1) */
} adgs_adam_slot;
typedef struct adgs_sh_adam {
	adgs_adam_slot scene_rest, obj_rest;       /* updated by the preprocess backward (needs M == 16 and its LDS row staging) */
	adgs_adam_slot scene_deform, obj_deform;   /* updated by the pass that expands dL/d(dc) into the deformation rows */
	float beta1, beta2, eps;
	int32_t reserved;
} adgs_sh_adam;

/* Per-iteration densification statistics (train.py:148-150):
 *   gaussians.max_radii2D[vis] = max(gaussians.max_radii2D[vis], radii[vis])
 *   GaussianModel.add_densification_stats (scene/gaussian_model.py:863-867):
 *     xyz_gradient_accum[vis] += ||viewspace_points.grad[vis, :2]||,  denom[vis] += 1
 * with vis = radii > 0, as ONE elementwise kernel (the reference's boolean-mask indexing costs a nonzero() with a host
 * synchronisation plus ~10 kernels).  viewspace_grad is [N,3] (the rasterizer's dL/dmeans2D), the three accumulators
 * are fp32 [N] / [N,1]. */
int adgs_densification_stats(int N, const int32_t* radii, const float* viewspace_grad, float* xyz_gradient_accum, float* denom,
	float* max_radii2D, void* stream);

#ifdef __cplusplus
}
#endif
#endif
