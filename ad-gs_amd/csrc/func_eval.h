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
__device__ __forceinline__ void lin_bwd_add(float* __restrict__ grow, const adgs_func_eval& f, float g) {
	const int total = f.n_terms[0] + f.n_terms[1] + f.n_terms[2];
	for (int i = 0; i < total; i++) grow[f.index[i]] += f.weight[i] * g;
}
__device__ __forceinline__ bool has_lin(const adgs_func_eval& f) { return (f.n_terms[0] + f.n_terms[1] + f.n_terms[2]) > 0; }

// Cooperative, fully coalesced transfer of the per-Gaussian parameter rows (L floats each) of the
// `count` consecutive Gaussians gi0.. between global memory and LDS rows of `stride` floats (odd
// stride: the later one-row-per-thread accesses are bank-conflict free).  Scene members (gi < Ns)
// live in `scene`, object members in `obj`; a nullptr side is skipped.  One thread per float,
// consecutive threads on consecutive addresses; the (row, column) pair advances without divisions.
template <bool TO_LDS, typename PtrT>
__device__ __forceinline__ void stage_rows(float* __restrict__ s, int stride, int L, int gi0, int count, int Ns, PtrT scene, PtrT obj,
	int tid, int nthreads) {
	constexpr int U = 8;                      // transfers in flight per thread: all loads of a batch are issued before the first store
	int g = tid / L, c = tid - g * L;
	const int dq = nthreads / L, dr = nthreads - dq * L;
	const int total = count * L;
	for (int e = tid; e < total; e += nthreads * U) {
		PtrT p[U]; int so[U]; float v[U];
#pragma unroll
		for (int u = 0; u < U; u++) {
			const int gi = gi0 + g;
			const bool in = e + u * nthreads < total;
			PtrT q = (gi >= Ns) ? (obj ? obj + (size_t)(gi - Ns) * L : nullptr) : (scene ? scene + (size_t)gi * L : nullptr);
			p[u] = (in && q) ? q + c : nullptr;
			so[u] = g * stride + c;
			c += dr; g += dq;
			if (c >= L) { c -= L; g++; }
		}
		if (TO_LDS) {
#pragma unroll
			for (int u = 0; u < U; u++) v[u] = p[u] ? *p[u] : 0.f;
#pragma unroll
			for (int u = 0; u < U; u++) if (p[u]) s[so[u]] = v[u];
		} else {
#pragma unroll
			for (int u = 0; u < U; u++) v[u] = p[u] ? s[so[u]] : 0.f;
#pragma unroll
			for (int u = 0; u < U; u++) if (p[u]) *const_cast<float*>(p[u]) = v[u];
		}
	}
}

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
