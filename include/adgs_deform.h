/*
 * adgs_deform.h -- C ABI of the fused per-frame deformation kernels (libadgs_hip.so).
 *
 * Replaces, for the hot path, the chain of small PyTorch kernels behind
 *   utils/func_utils.py:121-173        get_func_result(v, param, order_args)
 *   scene/gaussian_model.py:173-231    GaussianModel.get_deformed_xyz / _rotation / _shs /
 *                                      get_time_masked_opacity / get_deformed_pkg
 *   scene/gaussian_model.py:88-152     get_scaling / get_opacity / ... (cat + activation)
 * of the reference.  The time-dependent basis values (B-spline window weights,
 * v^i, sin/cos(f*pi*v), cumulative quaternion basis) are the same for every Gaussian;
 * the host evaluates them once per frame exactly like the reference does (float32) and
 * passes them in an adgs_func_eval; the kernels do the per-Gaussian streaming work.
 *
 * All pointers are device pointers to fp32 data unless stated; NULL = absent.
 */
#ifndef ADGS_DEFORM_H
#define ADGS_DEFORM_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ADGS_FUNC_MAX_TERMS 48
#define ADGS_FUNC_MAX_QUAT 8

/* One get_func_result(v, param[..., D, n_params], order_args) at a fixed v. */
typedef struct adgs_func_eval {
	int32_t n_params;                       /* last dimension of `param` */
	int32_t n_terms[3];                     /* #terms of the B-spline window, polynomial, Fourier parts (in this order) */
	int32_t index[ADGS_FUNC_MAX_TERMS];     /* column of param for each term (all three parts, concatenated) */
	float weight[ADGS_FUNC_MAX_TERMS];      /* basis value for each term */
	int32_t quat_start;                     /* first control-quaternion column (offset + segment start); -1: no quaternion spline */
	int32_t quat_k;                         /* spline order kq: kq+1 control quaternions are used */
	float quat_cum[ADGS_FUNC_MAX_QUAT];     /* cumulative basis B~_1..B~_kq (func_utils.py:162-163) */
} adgs_func_eval;

/* Generic get_func_result: out[n, d] for n < N, d < D (D = 3, or 4 for quaternion splines). */
int adgs_func_eval_forward(int N, int D, const float* param, const adgs_func_eval* f, float* out, void* stream);
/* dL_dparam must be zero-initialised [N, D, n_params]. */
int adgs_func_eval_backward(int N, int D, const float* param, const adgs_func_eval* f, const float* dL_dout, float* dL_dparam, void* stream);

/* Raw parameters of the reference GaussianModel (scene || object split, scene first). */
#define ADGS_DEFORM_TIME_MASK 1
#define ADGS_DEFORM_SKIP_SCENE 2
typedef struct adgs_deform_params {
	int32_t Ns, No;                         /* number of scene / object Gaussians */
	int32_t sh_coeffs;                      /* (max_sh_degree+1)^2 */
	int32_t use_time_mask;                  /* bit 0: time-masked object opacity (scene/gaussian_model.py:207-214); bit 1
	                                           (ADGS_DEFORM_SKIP_SCENE): the scene range [0, Ns) is not processed -- no output row
	                                           written, no scene gradient produced (its activations are applied by the rasterizer's
	                                           preprocess from the raw tensors, adgs_sh_source.scene_xyz ...) */
	float t;                                /* camera time in [0,1] */
	const float *scene_xyz, *obj_xyz;                   /* [Ns,3] [No,3] */
	const float *scene_rotation, *obj_rotation;         /* [Ns,4] [No,4] */
	const float *scene_shs_dc, *obj_shs_dc;             /* [Ns,1,3] [No,1,3] */
	const float *scene_shs_rest, *obj_shs_rest;         /* [Ns,sh_coeffs-1,3] ... */
	const float *scene_opacity, *obj_opacity;           /* [Ns,1] [No,1] (logits) */
	const float *scene_scaling, *obj_scaling;           /* [Ns,3] [No,3] (log) */
	const float *xyz_deform_param;                      /* [No,3,Cx] */
	const float *rotation_deform_param;                 /* [No,4,Cr] */
	const float *shs_deform_param_scene, *shs_deform_param_obj;   /* [Ns,3,Cs] [No,3,Cs] */
	const float *background_deform_param;               /* [1,3,Cb] */
	const float *gs_time, *gs_time_sigma;               /* [No,1] [No,2] */
} adgs_deform_params;

/* Outputs [N = Ns+No, ...]; any of them may be NULL (e.g. only xyz for the flow points). */
typedef struct adgs_deform_outputs {
	float *xyz;        /* [N,3] */
	float *rotation;   /* [N,4] normalised */
	float *shs;        /* [N,sh_coeffs,3] */
	float *opacity;    /* [N,1] */
	float *scales;     /* [N,3] */
} adgs_deform_outputs;

/* Gradients w.r.t. every raw parameter; same shapes as adgs_deform_params; NULL = not wanted.
 * background_deform_param's gradient is accumulated with atomics: zero-initialise it; all other
 * outputs are fully written. */
typedef struct adgs_deform_grads {
	float *scene_xyz, *obj_xyz, *scene_rotation, *obj_rotation, *scene_shs_dc, *obj_shs_dc, *scene_shs_rest, *obj_shs_rest,
	      *scene_opacity, *obj_opacity, *scene_scaling, *obj_scaling, *xyz_deform_param, *rotation_deform_param,
	      *shs_deform_param_scene, *shs_deform_param_obj, *background_deform_param, *gs_time_sigma;
} adgs_deform_grads;

int adgs_deform_forward(const adgs_deform_params* p, const adgs_func_eval* f_xyz, const adgs_func_eval* f_rotation,
	const adgs_func_eval* f_shs, const adgs_func_eval* f_background, const adgs_deform_outputs* out, void* stream);

/* dL_d* are the upstream gradients of the outputs (NULL = zero). */
int adgs_deform_backward(const adgs_deform_params* p, const adgs_func_eval* f_xyz, const adgs_func_eval* f_rotation,
	const adgs_func_eval* f_shs, const adgs_func_eval* f_background,
	const float* dL_dxyz, const float* dL_drotation, const float* dL_dshs, const float* dL_dopacity, const float* dL_dscales,
	const adgs_deform_grads* grads, void* stream);

/* The same with the flow points fused in: gaussian_renderer/__init__.py:57 evaluates get_deformed_xyz a second
 * time at flow_time; here both evaluations share one pass over the xyz deformation rows.  f_xyz_flow /
 * f_background_flow are the basis values at the flow time, flow_xyz [N,3] the second output (NULL: absent),
 * dL_dflow_xyz its upstream gradient; the xyz / xyz_deform_param / background gradients are the sums over both. */
int adgs_deform_forward_flow(const adgs_deform_params* p, const adgs_func_eval* f_xyz, const adgs_func_eval* f_rotation,
	const adgs_func_eval* f_shs, const adgs_func_eval* f_background, const adgs_func_eval* f_xyz_flow, const adgs_func_eval* f_background_flow,
	const adgs_deform_outputs* out, float* flow_xyz, void* stream);
int adgs_deform_backward_flow(const adgs_deform_params* p, const adgs_func_eval* f_xyz, const adgs_func_eval* f_rotation,
	const adgs_func_eval* f_shs, const adgs_func_eval* f_background, const adgs_func_eval* f_xyz_flow, const adgs_func_eval* f_background_flow,
	const float* dL_dxyz, const float* dL_drotation, const float* dL_dshs, const float* dL_dopacity, const float* dL_dscales, const float* dL_dflow_xyz,
	const adgs_deform_grads* grads, void* stream);

#ifdef __cplusplus
}
#endif
#endif
