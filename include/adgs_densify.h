/*
 * adgs_densify.h -- C ABI of the densify / prune compaction (libadgs_hip.so).  SURVEY.md section 8(f) row 4.
 *
 * Replaces, per side (scene or object Gaussians), the tensor surgery of
 *   GaussianModel.densify_and_prune / densify_and_clone / densify_and_split / prune_points
 *   (scene/gaussian_model.py:581-611, 715-823, 835-861) and of cat_tensors_to_optimizer / _prune_optimizer (:561-635):
 * the reference concatenates and boolean-masks every one of the 17 parameter tensors and both Adam moments three times
 * (clone, split + parent prune, final prune; ~150 ATen kernels and a dozen host synchronisations).  Here the composition
 * of the three steps is evaluated as ONE row map
 *     result = [ surviving originals | surviving clones | surviving split children (copy 1, then copy 2) ]
 * (exactly the reference's row order) and every tensor is moved once by a gather.
 *
 *   1. adgs_densify_select   clone / split decisions of this side, as ordered index lists + their counts
 *   2. (host)                reads the counts; draws torch.normal(0, scaling[split].repeat(2,1)) like the reference
 *   3. adgs_densify_plan     prune decisions on originals, clones and children; row map of the result + its length
 *   4. adgs_densify_gather_rows / adgs_densify_split_rows     one pass per tensor
 *
 * All pointers are device pointers; fp32 data unless stated.  Returns 0 or a negative code (adgs_last_error()).
 */
#ifndef ADGS_DENSIFY_H
#define ADGS_DENSIFY_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ADGS_ROW_KEEP  0xffffffffu    /* row_aux: a surviving original (keeps its Adam moments) */
#define ADGS_ROW_CLONE 0xfffffffeu    /* row_aux: a clone (moments start at zero); other values: index into the split samples */

typedef struct adgs_densify_side {
	int32_t N;                 /* Gaussians of this side before densification */
	const float* grad_accum;   /* [N] this side's rows of xyz_gradient_accum */
	const float* denom;        /* [N] this side's rows of denom */
	const float* scaling;      /* [N,3] raw (log) scales */
	const float* opacity;      /* [N,1] raw (logit) opacities */
	float grad_threshold;      /* max_scene_grad / max_obj_grad: selected iff |accum / denom| (NaN -> 0) >= threshold (:836-843) */
	float dense_extent;        /* extent * percent_dense: clone iff max exp(scaling) <= it, split iff > (:717-718, :771-772) */
	float min_opacity;         /* prune iff sigmoid(opacity) < min_opacity (:851-852) */
	float big_extent;          /* scene_extent * 0.05 / object_extent * 0.1 (:854-855) */
	int32_t prune_big;         /* prune_big_points */
} adgs_densify_side;

size_t adgs_densify_workspace_bytes(int n_rows);     /* for select: n_rows = N; for plan: n_rows = N + n_clone + 2 * n_split */

/* clone_index / split_index: [N] u32, the selected originals in ascending order; counts: device u32[2] = {n_clone, n_split}. */
int adgs_densify_select(const adgs_densify_side* side, uint32_t* clone_index, uint32_t* split_index, uint32_t* counts, char* workspace, void* stream);

/* Row map of the result.  Candidates, in the reference's order: originals 0..N-1 (dropped when split, :765-767), clones,
 * children copy 1, children copy 2 (child j of 2*n_split has parent split_index[j % n_split], scaling log(exp(s) / 1.6)).
 * row_src / row_aux: [N + n_clone + 2*n_split] u32, filled for the first *count rows; count: device u32[1]. */
int adgs_densify_plan(const adgs_densify_side* side, const uint32_t* clone_index, int n_clone, const uint32_t* split_index, int n_split,
	uint32_t* row_src, uint32_t* row_aux, uint32_t* count, char* workspace, void* stream);

/* dst[i, :] = src[row_src[i], :] for i < n_out; rows of row_floats floats.  is_state != 0 (Adam exp_avg / exp_avg_sq): rows whose
 * row_aux is not ADGS_ROW_KEEP are zero (cat_tensors_to_optimizer :623-624). */
int adgs_densify_gather_rows(const float* src, float* dst, int row_floats, int n_out, const uint32_t* row_src, const uint32_t* row_aux,
	int is_state, void* stream);

/* Overwrites the rows of split children (row_aux < ADGS_ROW_CLONE) in xyz_dst / scaling_dst [n_out,3]:
 *   xyz = build_rotation(rotation[src]) @ samples[row_aux] + xyz[src]   (utils/general_utils.py:79-95, :721-722)
 *   scaling = log(exp(scaling[src]) / 1.6)                              (:723)
 * samples: [2*n_split,3] = torch.normal(0, exp(scaling)[split].repeat(2,1)) drawn by the caller. */
int adgs_densify_split_rows(const float* xyz_src, const float* scaling_src, const float* rotation_src, const float* samples,
	int n_out, const uint32_t* row_src, const uint32_t* row_aux, float* xyz_dst, float* scaling_dst, void* stream);

/* GaussianModel.reset_opacity (:465-469): opacity = inverse_sigmoid(min(sigmoid(opacity), 0.01)), in place, [N]. */
int adgs_reset_opacity(int N, float* opacity, void* stream);

#ifdef __cplusplus
}
#endif
#endif
