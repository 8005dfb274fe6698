/*
 * adgs_loss.h -- C ABI of the fused photometric loss (libadgs_hip.so).  SURVEY.md section 8(f) row 2.
 *
 * Replaces, for train.py:79-80,112, the pair
 *   utils/loss_utils.py:20-21   l1_loss(image, gt)          = mean |image - gt|
 *   utils/loss_utils.py:37-68   ssim(image, gt)             = mean of the SSIM map, 11x11 Gaussian window
 *                                                             (sigma 1.5), zero padding, C1 = 0.01^2, C2 = 0.03^2
 * (five depthwise conv2d + ~15 elementwise kernels forward, as many again in autograd's backward)
 * by one forward and one backward kernel.  Images are `planes` x H x W fp32 (planes = batch x channels).
 * The forward also stores the three per-pixel partial derivatives of the SSIM map w.r.t. the window
 * means (mu1, E[x1^2], E[x1 x2]) that the backward convolves; only d/d(image) exists (gt is constant).
 */
#ifndef ADGS_LOSS_H
#define ADGS_LOSS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Work buffers of this header (`sums`, `work`): their SLOT rows must be zero when a forward is enqueued; the kernel that consumes the
 * rows (adgs_l1_ssim_means, the finish kernels inside the other forwards) leaves them zero again, so ONE zero-initialised buffer per
 * concurrent loss term serves every iteration without a fill (the scalars behind the rows -- totals the backward reads -- are
 * overwritten by every forward: a buffer may be reused once its backward has been enqueued, stream order).
 *
 * sums: ADGS_LOSS_SLOTS x 2 device doubles, zero on entry.  Workgroup b adds its partial sums
 * (sum |image - gt|, sum ssim_map) into row b % ADGS_LOSS_SLOTS -- atomics on ONE address would serialise in the L2;
 * adgs_l1_ssim_means (or the caller) adds the rows up and divides by planes*H*W for the two means.
 * d_mu1 / d_e11 / d_e12: planes*H*W floats each, or all NULL for a forward without backward. */
#define ADGS_LOSS_SLOTS 256
int adgs_l1_ssim_forward(int planes, int H, int W, const float* image, const float* gt, double* sums,
	float* d_mu1, float* d_e11, float* d_e12, void* stream);
/* out2[0] = mean |image - gt| (utils/loss_utils.py:20-21), out2[1] = mean ssim_map (:37-68), n = planes*H*W; consumes `sums` (zero afterwards). */
int adgs_l1_ssim_means(double* sums, long long n, float* out2, void* stream);

/* dL_dimage = g_l1[0] * sign(image - gt) / n + g_ssim[0] * d(mean ssim)/d(image),  n = planes*H*W.
 * g_l1 / g_ssim are DEVICE scalars (the upstream gradients of the two means; NULL = 0): no host round trip. */
int adgs_l1_ssim_backward(int planes, int H, int W, const float* image, const float* gt,
	const float* d_mu1, const float* d_e11, const float* d_e12, const float* g_l1, const float* g_ssim, float* dL_dimage, void* stream);

/*
 * Scale-and-shift-invariant depth loss: utils/loss_utils.py:70-75 get_depth_loss over
 * utils/depth_utils.py:3-45 (closed-form least squares for scale s and shift t of the prediction, then
 * sum(|s p + t - g| m) / sum(m)), differentiable through s and t like the reference's autograd, with the
 * reference's `det == 0 -> (0, 0)` case decided on the device (the reference synchronises for it).
 * work: ADGS_DEPTH_WORK_DOUBLES device doubles (256 x 8 slot rows, zero on entry and on return; 16 scalars), kept for the backward.
 * mask may be NULL (all ones).  loss: one device float.
 */
#define ADGS_DEPTH_WORK_DOUBLES (256 * 8 + 16)
int adgs_depth_loss_forward(int n, const float* prediction, const float* target, const float* mask, double* work, float* loss, void* stream);
/* dL_dprediction[i] for the upstream gradient g_loss (DEVICE scalar) */
int adgs_depth_loss_backward(int n, const float* prediction, const float* target, const float* mask, const double* work, const float* g_loss,
	float* dL_dprediction, void* stream);

/*
 * Flow re-projection loss: utils/loss_utils.py:86-106 get_flow_loss over utils/flow_utils.py:5-10 flow_points_project.
 * img_flow [3,H,W] are the rendered 3-D flow points, flow [2,H,W] the target pixel coordinates, flow_vis [H,W] the visibility
 * score (selected iff > 0.5 and the target lies inside the image), img_opacity [H,W] an optional weight (NULL = 1).
 * K, R, T: HOST pointers to the 3x3 / 3x3 / 3 camera of the flow target (row-major).  Per selected pixel
 *   p = K (R f + T); (u, v) = p.xy / max(p.z, dist); loss += (|u - flow_x| / W + |v - flow_y| / H) * opacity * [p.z > dist]
 * and loss = sum / #selected, 0 when nothing is selected -- decided on the device (the reference's nonzero() synchronises).
 * work: ADGS_AUX_WORK_DOUBLES device doubles (256 x 2 slot rows, zero on entry and on return; 2 scalars), kept for the backward.
 */
#define ADGS_AUX_WORK_DOUBLES (256 * 2 + 2)
int adgs_flow_loss_forward(int H, int W, const float* img_flow, const float* flow, const float* flow_vis, const float* img_opacity,
	const float* K, const float* R, const float* T, float dist, double* work, float* loss, void* stream);
/* dL_dimg_flow [3,H,W] (fully written), dL_dimg_opacity [H,W] or NULL; g_loss is a DEVICE scalar. */
int adgs_flow_loss_backward(int H, int W, const float* img_flow, const float* flow, const float* flow_vis, const float* img_opacity,
	const float* K, const float* R, const float* T, float dist, const double* work, const float* g_loss, float* dL_dimg_flow, float* dL_dimg_opacity,
	void* stream);
/* The same two entry points with K, R, T as DEVICE pointers (train.py:68-71 moves the flow package to the GPU every iteration:
 * `flow_pkg = [a.cuda() ...]`): the kernels form K R and K T themselves, in the host's operation order (bit-identical results);
 * nothing is read back and no host-side copy of the camera exists that could go stale. */
int adgs_flow_loss_forward_devcam(int H, int W, const float* img_flow, const float* flow, const float* flow_vis, const float* img_opacity,
	const float* K, const float* R, const float* T, float dist, double* work, float* loss, void* stream);
int adgs_flow_loss_backward_devcam(int H, int W, const float* img_flow, const float* flow, const float* flow_vis, const float* img_opacity,
	const float* K, const float* R, const float* T, float dist, const double* work, const float* g_loss, float* dL_dimg_flow, float* dL_dimg_opacity,
	void* stream);

/*
 * Clipped binary cross entropy (train.py:95-103): mean BCE of q against t, q = clip(pred, lo, hi) or, with invert,
 * 1 - clip(pred, lo, hi) (the sky term); t = target or, with positive_target, (target > 0) (the object-mask term).
 * Same work buffer convention as above.
 */
int adgs_bce_clip_forward(int n, const float* pred, const float* target, float lo, float hi, int invert, int positive_target,
	double* work, float* loss, void* stream);
int adgs_bce_clip_backward(int n, const float* pred, const float* target, float lo, float hi, int invert, int positive_target,
	const float* g_loss, float* dL_dpred, void* stream);

/*
 * Neighbourhood regularisers of the training loop (train.py:104-113), over GaussianModel.obj_near_idx [G, K] (int64 rows of
 * the object range, scene/gaussian_model.py:825-833):
 *   reg_loss       = mean(sum(var(xyz_deform_param[obj_near_idx], dim=1), dim=-1))   x = xyz_deform_param [N,3,C]: D = 3 C, inner = C
 *   reg_sigma_loss = mean(sum(var(gs_time_sigma[obj_near_idx], dim=1), dim=-1))      x = gs_time_sigma [N,2]:      D = 2,   inner = 2
 * i.e. loss = sum_g sum_d var_k(x[idx[g,k], d]) / (G D / inner) with torch.var's unbiased estimator (2 <= K <= 32).
 * work: ADGS_AUX_WORK_DOUBLES device doubles, zero-initialised by the caller; loss: one device float.
 * The backward ADDS into dL_dx [N, D] (zero-initialised by the caller: a Gaussian can sit in several neighbourhoods).
 */
int adgs_group_var_forward(int N, int G, int K, int D, int inner, const float* x, const int64_t* idx, double* work, float* loss, void* stream);
int adgs_group_var_backward(int N, int G, int K, int D, int inner, const float* x, const int64_t* idx, const float* g_loss, float* dL_dx, void* stream);
/* sigma_loss = mean(|frame_gap / mean(exp(gs_time_sigma), dim=-1)|) over gs_time_sigma [N,2] (train.py:108-110); every row of
 * dL_dlog_sigma is written. */
int adgs_sigma_loss_forward(int N, const float* log_sigma, float frame_gap, double* work, float* loss, void* stream);
int adgs_sigma_loss_backward(int N, const float* log_sigma, float frame_gap, const float* g_loss, float* dL_dlog_sigma, void* stream);

#ifdef __cplusplus
}
#endif
#endif
