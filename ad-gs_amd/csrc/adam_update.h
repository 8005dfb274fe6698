// The Adam update as ONE device function, shared by the optimizer's own kernel (optim.hip) and by the backward kernels that apply
// the step in place of storing a gradient (include/adgs_optim.h: adgs_sh_adam): same expression, contraction off, so that a
// parameter takes the same bits whichever kernel updates it.
#pragma once
#include "common.h"
#include "stream_access.h"

namespace adgs {

struct AdamSlot { float *p, *m, *v; float step_size, inv_bc2_sqrt; };         // p == nullptr: off
struct AdamEpilogue { AdamSlot scene_rest, obj_rest, scene_sp, obj_sp; float beta1, beta2, eps; };

__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float g, float beta1, float beta2, float eps, float step_size, float inv_bc2_sqrt) {
#pragma clang fp contract(off)
	m = m + (1.f - beta1) * (g - m);                     // exp_avg.lerp_(grad, 1 - beta1)
	v = beta2 * v + (1.f - beta2) * g * g;               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
	const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
	p = p - step_size * (m / denom);
}
__device__ __forceinline__ void adam_update4(float4& p, float4& m, float4& v, const float4 g, float beta1, float beta2, float eps, float step_size, float inv_bc2_sqrt) {
	adam_update(p.x, m.x, v.x, g.x, beta1, beta2, eps, step_size, inv_bc2_sqrt);
	adam_update(p.y, m.y, v.y, g.y, beta1, beta2, eps, step_size, inv_bc2_sqrt);
	adam_update(p.z, m.z, v.z, g.z, beta1, beta2, eps, step_size, inv_bc2_sqrt);
	adam_update(p.w, m.w, v.w, g.w, beta1, beta2, eps, step_size, inv_bc2_sqrt);
}

// The per-Gaussian gradient rows of `count` consecutive Gaussians gi0.. sit in LDS (rows of `stride` floats, L used: what
// stage_rows<false> would stream out to the gradient tensor): apply the step to the parameter rows instead.  Scene members (gi < Ns)
// belong to slot `scene`, object members to `obj`.  Same access pattern as stage_rows: a block on one side of the boundary moves
// 16-byte quads of p / m / v, fully coalesced; a block that straddles it goes element by element.
template <int U4>
__device__ __forceinline__ void adam_rows(const float* __restrict__ s, int stride, int L, int gi0, int count, int Ns, const AdamSlot& scene, const AdamSlot& obj,
	float beta1, float beta2, float eps, int tid, int nthreads) {
	const int total = count * L;
	int e_begin = 0;
	const bool all_scene = gi0 + count <= Ns, all_obj = gi0 >= Ns;
	if (all_scene || all_obj) {
		// the slot's fields as values (a reference chosen at run time would move the by-value kernel argument into scratch memory)
		float* const sp = all_scene ? scene.p : obj.p;
		if (!sp) return;
		const float step_size = all_scene ? scene.step_size : obj.step_size, ibc2 = all_scene ? scene.inv_bc2_sqrt : obj.inv_bc2_sqrt;
		const size_t off = (size_t)(all_scene ? gi0 : gi0 - Ns) * L;
		float *P = sp + off, *M = (all_scene ? scene.m : obj.m) + off, *V = (all_scene ? scene.v : obj.v) + off;
		if (((reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(M) | reinterpret_cast<uintptr_t>(V)) & 15) == 0 && L >= 4) {
			const int total4 = total >> 2;
			int e = 4 * tid;
			int g = e / L, c = e - g * L;
			const int step = 4 * nthreads, dq = step / L, dr = step - dq * L;
			const int pad = stride - L;
			for (int q = tid; q < total4; q += nthreads * U4) {
				float4 p4[U4], m4[U4], v4[U4]; int so[U4], cc[U4];
#pragma unroll
				for (int u = 0; u < U4; u++) {
					so[u] = g * stride + c; cc[u] = c;
					const int qq = min(q + u * nthreads, total4 - 1);         // unconditional loads at a clamped index (stage_rows)
					p4[u] = ld_stream4(reinterpret_cast<const float4*>(P) + qq); m4[u] = ld_stream4(reinterpret_cast<const float4*>(M) + qq); v4[u] = ld_stream4(reinterpret_cast<const float4*>(V) + qq);
					c += dr; g += dq;
					if (c >= L) { c -= L; g++; }
				}
#pragma unroll
				for (int u = 0; u < U4; u++) {
					if (q + u * nthreads >= total4) continue;
					const int o0 = so[u], o1 = so[u] + 1 + (cc[u] + 1 >= L ? pad : 0), o2 = so[u] + 2 + (cc[u] + 2 >= L ? pad : 0), o3 = so[u] + 3 + (cc[u] + 3 >= L ? pad : 0);
					adam_update4(p4[u], m4[u], v4[u], make_float4(s[o0], s[o1], s[o2], s[o3]), beta1, beta2, eps, step_size, ibc2);
					st_stream4(reinterpret_cast<float4*>(P) + (q + u * nthreads), p4[u]); st_stream4(reinterpret_cast<float4*>(M) + (q + u * nthreads), m4[u]); st_stream4(reinterpret_cast<float4*>(V) + (q + u * nthreads), v4[u]);
				}
			}
			e_begin = total4 << 2;
		}
	}
	int g = (e_begin + tid) / L, c = (e_begin + tid) - g * L;
	const int dq = nthreads / L, dr = nthreads - dq * L;
	for (int e = e_begin + tid; e < total; e += nthreads) {
		const int gi = gi0 + g;
		const bool o = gi >= Ns;
		float* const sp = o ? obj.p : scene.p;
		if (sp) {
			float* const sm = o ? obj.m : scene.m; float* const sv = o ? obj.v : scene.v;
			const size_t i = (size_t)(o ? gi - Ns : gi) * L + c;
			float p = sp[i], m = sm[i], v = sv[i];
			adam_update(p, m, v, s[g * stride + c], beta1, beta2, eps, o ? obj.step_size : scene.step_size, o ? obj.inv_bc2_sqrt : scene.inv_bc2_sqrt);
			sp[i] = p; sm[i] = m; sv[i] = v;
		}
		c += dr; g += dq;
		if (c >= L) { c -= L; g++; }
	}
}

} // namespace adgs
