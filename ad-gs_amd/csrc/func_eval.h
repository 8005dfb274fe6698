// Device helpers shared by the deformation kernels and the raw-SH path of the preprocess kernels.
#pragma once
#include "common.h"
#include "stream_access.h"
#include "adam_update.h"
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
// U4: 16-byte transfers in flight per thread on the contiguous path -- a batch is one memory round trip for the whole workgroup, so a
// caller whose registers are idle while it stages (the preprocess kernels: 12 quads per thread for 45-float SH rows) asks for all of them at once.
// `between`: called exactly once, after the loads of the first batch have been issued and before their LDS stores -- the place for a
// caller's own loads that should share that round trip.
struct StageNoOp { __device__ __forceinline__ void operator()() const {} };
#ifndef ADGS_STAGE_U4_DEFAULT
#define ADGS_STAGE_U4_DEFAULT 4
#endif
template <bool TO_LDS, int U4 = ADGS_STAGE_U4_DEFAULT, typename PtrT, typename Between = StageNoOp>
__device__ __forceinline__ void stage_rows(float* __restrict__ s, int stride, int L, int gi0, int count, int Ns, PtrT scene, PtrT obj,
	int tid, int nthreads, Between between = Between()) {
	constexpr int U = 8;                      // transfers in flight per thread: all loads of a batch are issued before the first store
	bool called = false;
	const int total = count * L;
	int e_begin = 0;
	// fast path, block entirely on one side of the scene|object boundary: its rows are one contiguous slab,
	// moved with 16-byte global accesses (four LDS words each; a quad may straddle two rows)
	const bool all_scene = gi0 + count <= Ns, all_obj = gi0 >= Ns;
	if (all_scene || all_obj) {
		PtrT slab = all_scene ? (scene ? scene + (size_t)gi0 * L : nullptr) : (obj ? obj + (size_t)(gi0 - Ns) * L : nullptr);
		if (!slab) { between(); return; }
		if ((reinterpret_cast<uintptr_t>(slab) & 15) == 0 && L >= 4) {
			const int total4 = total >> 2;
			int e = 4 * tid;
			int g = e / L, c = e - g * L;
			const int step = 4 * nthreads, dq = step / L, dr = step - dq * L;
			for (int q = tid; q < total4; q += nthreads * U4) {
				float4 v[U4]; int so[U4], cc[U4]; bool in[U4];
#pragma unroll
				for (int u = 0; u < U4; u++) {
					in[u] = q + u * nthreads < total4;
					so[u] = g * stride + c; cc[u] = c;
					// UNCONDITIONAL load at a clamped index: a load under `in[u] ? ... : ...` sits in its own exec-masked block, whose end
					// waits for it (s_waitcnt vmcnt(0)) -- the U4 round trips of a batch then run one after the other
					if (TO_LDS) v[u] = ld_stream4(reinterpret_cast<const float4*>(slab) + min(q + u * nthreads, total4 - 1));
					c += dr; g += dq;
					if (c >= L) { c -= L; g++; }
				}
				if (!called) { between(); called = true; }
#pragma unroll
				for (int u = 0; u < U4; u++) {
					if (!in[u]) continue;
					// element i of the quad sits at column cc + i, wrapping into the next row (one pad word further)
					const int pad = stride - L;
					const int o0 = so[u], o1 = so[u] + 1 + (cc[u] + 1 >= L ? pad : 0), o2 = so[u] + 2 + (cc[u] + 2 >= L ? pad : 0), o3 = so[u] + 3 + (cc[u] + 3 >= L ? pad : 0);
					if (TO_LDS) { s[o0] = v[u].x; s[o1] = v[u].y; s[o2] = v[u].z; s[o3] = v[u].w; }
					else st_stream4(reinterpret_cast<float4*>(const_cast<float*>(slab)) + (q + u * nthreads), make_float4(s[o0], s[o1], s[o2], s[o3]));
				}
			}
			e_begin = total4 << 2;            // at most three tail elements go through the generic loop
		}
	}
	if (!called) { between(); called = true; }
	if (!scene && !obj) return;
	int g = (e_begin + tid) / L, c = (e_begin + tid) - g * L;
	const int dq = nthreads / L, dr = nthreads - dq * L;
	for (int e = e_begin + tid; e < total; e += nthreads * U) {
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
			// unconditional loads (see above): an absent side / an element past the end reads a valid dummy address -- one from the side
			// that HAS rows in this block (a non-null pointer to a zero-row tensor must never be dereferenced: foreign C callers)
			const bool scene_rows = scene && gi0 < Ns, obj_rows = obj && gi0 + count > Ns;
			PtrT dummy = scene_rows ? scene + (size_t)gi0 * L : (obj_rows ? obj + (size_t)(max(gi0, Ns) - Ns) * L : (scene ? scene : obj));
#pragma unroll
			for (int u = 0; u < U; u++) v[u] = ld_stream(p[u] ? p[u] : dummy);
#pragma unroll
			for (int u = 0; u < U; u++) if (p[u]) s[so[u]] = v[u];
		} else {
#pragma unroll
			for (int u = 0; u < U; u++) v[u] = p[u] ? s[so[u]] : 0.f;
#pragma unroll
			for (int u = 0; u < U; u++) if (p[u]) st_stream(const_cast<float*>(p[u]), v[u]);
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
	// raw scene geometry (scene_xyz != nullptr): Gaussians idx < Ns take position / log-scale / raw rotation / opacity logit from
	// these raw tensors and the preprocess applies exp / normalize / sigmoid itself; rows idx < Ns of the activated inputs are
	// then never read (scene/gaussian_model.py:89-152: the reference materialises the activations with torch ops)
	const float *scene_xyz, *scene_scaling, *scene_rotation, *scene_opacity;
	const float *bg_image;    // [3,H,W] per-pixel background composited in the blend epilogue (nullptr: the constant background colour)
};
struct ShGradDst {
	float *scene_dc, *obj_dc, *scene_rest, *obj_rest, *scene_sp, *obj_sp;
	float *rgb_factor;   // [P,3] clamp-masked colour gradient: the per-camera factor every SH gradient row is a multiple of
	float *scene_xyz, *scene_scaling, *scene_rotation, *scene_opacity;   // raw scene geometry gradients ([Ns,3] [Ns,3] [Ns,4] [Ns,1])
	float *bg_image;          // [3,H,W] gradient of the per-pixel background (every pixel written) or nullptr
	// slots with p != nullptr: the Adam step in place of that tensor's gradient store (include/adgs_optim.h: adgs_sh_adam)
	AdamEpilogue adam = { { nullptr, nullptr, nullptr, 0.f, 0.f }, { nullptr, nullptr, nullptr, 0.f, 0.f }, { nullptr, nullptr, nullptr, 0.f, 0.f },
	                      { nullptr, nullptr, nullptr, 0.f, 0.f }, 0.f, 0.f, 0.f };
};

} // namespace adgs
