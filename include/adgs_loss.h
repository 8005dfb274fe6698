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

/* sums: ADGS_LOSS_SLOTS x 2 device doubles, zero-initialised by the caller.  Workgroup b adds its partial sums
 * (sum |image - gt|, sum ssim_map) into row b % ADGS_LOSS_SLOTS -- atomics on ONE address would serialise in the L2;
 * the caller adds the rows up and divides by planes*H*W for the two means.
 * d_mu1 / d_e11 / d_e12: planes*H*W floats each, or all NULL for a forward without backward. */
#define ADGS_LOSS_SLOTS 256
int adgs_l1_ssim_forward(int planes, int H, int W, const float* image, const float* gt, double* sums,
	float* d_mu1, float* d_e11, float* d_e12, void* stream);

/* dL_dimage = g_l1[0] * sign(image - gt) / n + g_ssim[0] * d(mean ssim)/d(image),  n = planes*H*W.
 * g_l1 / g_ssim are DEVICE scalars (the upstream gradients of the two means; NULL = 0): no host round trip. */
int adgs_l1_ssim_backward(int planes, int H, int W, const float* image, const float* gt,
	const float* d_mu1, const float* d_e11, const float* d_e12, const float* g_l1, const float* g_ssim, float* dL_dimage, void* stream);

/*
 * Scale-and-shift-invariant depth loss: utils/loss_utils.py:70-75 get_depth_loss over
 * utils/depth_utils.py:3-45 (closed-form least squares for scale s and shift t of the prediction, then
 * sum(|s p + t - g| m) / sum(m)), differentiable through s and t like the reference's autograd, with the
 * reference's `det == 0 -> (0, 0)` case decided on the device (the reference synchronises for it).
 * work: ADGS_DEPTH_WORK_DOUBLES device doubles, zero-initialised by the caller, kept for the backward.
 * mask may be NULL (all ones).  loss: one device float.
 */
#define ADGS_DEPTH_WORK_DOUBLES (256 * 8 + 16)
int adgs_depth_loss_forward(int n, const float* prediction, const float* target, const float* mask, double* work, float* loss, void* stream);
/* dL_dprediction[i] for the upstream gradient g_loss (DEVICE scalar) */
int adgs_depth_loss_backward(int n, const float* prediction, const float* target, const float* mask, const double* work, const float* g_loss,
	float* dL_dprediction, void* stream);

#ifdef __cplusplus
}
#endif
#endif
