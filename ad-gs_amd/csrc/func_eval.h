// Device helpers shared by the deformation kernels and the raw-SH path of the preprocess kernels.
#pragma once
#include "common.h"
#include "../../include/adgs_deform.h"

namespace adgs {

// linear families of get_func_result: out = ((0 + bspline) + poly) + fft, each part summed in order
__device__ __forceinline__ float lin_eval(const float* __restrict__ row, const adgs_func_eval& f) {
	float result = 0.f;
	int i = 0;
#pragma unroll
	for (int part = 0; part < 3; part++) {
		const int cnt = f.n_terms[part];
		if (cnt > 0) {
			float s = 0.f;
			for (int k = 0; k < cnt; k++, i++) s += row[f.index[i]] * f.weight[i];
			result = result + s;
		}
	}
	return result;
}
__device__ __forceinline__ void lin_bwd(float* __restrict__ grow, const adgs_func_eval& f, float g) {
	const int total = f.n_terms[0] + f.n_terms[1] + f.n_terms[2];
	for (int i = 0; i < total; i++) grow[f.index[i]] = f.weight[i] * g;
}
__device__ __forceinline__ bool has_lin(const adgs_func_eval& f) { return (f.n_terms[0] + f.n_terms[1] + f.n_terms[2]) > 0; }

// SH coefficients assembled on the fly from the reference GaussianModel's raw tensors
// (scene || object, dc + f_shs(t) || rest; scene/gaussian_model.py:198-205) instead of a
// materialised [P, M, 3] tensor.  scene_dc == nullptr means "not used".
struct ShSource {
	int Ns;
	const float *scene_dc, *obj_dc, *scene_rest, *obj_rest, *scene_sp, *obj_sp;
	adgs_func_eval f;
};
struct ShGradDst {
	float *scene_dc, *obj_dc, *scene_rest, *obj_rest, *scene_sp, *obj_sp;
};

} // namespace adgs
