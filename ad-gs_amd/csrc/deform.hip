// Fused per-frame deformation for gfx950: B-spline / polynomial / Fourier trajectories,
// cumulative quaternion B-spline, scene||object concatenation and the activations
// (exp, sigmoid x time mask, normalize) in ONE streaming pass, with a hand-written backward.
//
// Reference semantics: utils/func_utils.py:121-173 (get_func_result) and
// scene/gaussian_model.py:88-231 (getters / get_deformed_pkg); the roma quaternion helpers
// (unitquat_to_rotvec / rotvec_to_unitquat / quat_product, XYZW) are restated from the
// library's published algorithm (see oracle/deform_oracle.py for the parity status).
//
// Roofline: HBM streaming.  Per Gaussian the forward reads the raw parameters once
// (59 floats + deformation rows) and writes the 59 activated floats the rasterizer reads;
// the backward reads the 59 upstream gradients and writes every parameter gradient.
#include "common.h"
#include <initializer_list>
#include <cstdlib>
#include "func_eval.h"
#include <cstring>
#include <algorithm>

namespace adgs {
namespace {

constexpr int MAXQ = ADGS_FUNC_MAX_QUAT;     // control quaternions per evaluation: quat_k + 1 <= MAXQ

struct Q { float w, x, y, z; };
__device__ __forceinline__ Q qmul(const Q& a, const Q& b) {
	Q c;
	c.x = a.w * b.x + b.w * a.x + (a.y * b.z - a.z * b.y);
	c.y = a.w * b.y + b.w * a.y + (a.z * b.x - a.x * b.z);
	c.z = a.w * b.z + b.w * a.z + (a.x * b.y - a.y * b.x);
	c.w = a.w * b.w - (a.x * b.x + a.y * b.y + a.z * b.z);
	return c;
}
__device__ __forceinline__ Q qconj(const Q& a) { return { a.w, -a.x, -a.y, -a.z }; }
__device__ __forceinline__ Q qadd(const Q& a, const Q& b) { return { a.w + b.w, a.x + b.x, a.y + b.y, a.z + b.z }; }
__device__ __forceinline__ float qdot(const Q& a, const Q& b) { return a.w * b.w + a.x * b.x + a.y * b.y + a.z * b.z; }

// ---- cumulative quaternion B-spline (func_utils.py:156-171), wxyz in / wxyz out ----
// param block of one Gaussian: [4][n_params]; control quaternion j = normalize(param[:, c0 + j] + (1,0,0,0)).
//   out = q_0 * prod_{j>=1} exp(B~_j * log(conj(q_{j-1}) q_j))
// log = roma.unitquat_to_rotvec (shortest arc: flip when w < 0; angle a = 2 atan2(|v|, w); factor a / sin(a/2),
// Taylor 2 + a^2/12 + 7a^4/2880 for a <= 1e-3), exp = roma.rotvec_to_unitquat (scale sin(n/2)/n, Taylor
// 0.5 - n^2/48 + n^4/3840 for n <= 1e-3).  Per segment this costs ONE atan2f and ONE sincosf: sin(a/2) and
// cos(a/2) of the log map are |v|/r and w/r with r = sqrt(|v|^2 + w^2) (the sine / cosine of atan2(|v|, w)), and
// the backward reuses the saved forward values instead of re-evaluating any transcendental.
// NQ = number of control quaternions used (quat_k + 1), a template parameter so that the per-segment state
// lives in registers sized for the actual spline order.
template <int NQ> struct QuatSave {
	Q q[NQ]; float inv_nrm[NQ];                 // normalised control quaternions, 1 / |ctrl|
	Q r[NQ];                                    // r[j] = exp(B~_j log(conj(q[j-1]) q[j])), j >= 1
	float vx[NQ], vy[NQ], vz[NQ], wq[NQ];       // delta_j = conj(q[j-1]) q[j] after the shortest-arc flip
	float a[NQ], n[NQ], she[NQ];                // log angle; |rotvec| of the exp and sin(n/2)
	bool flip[NQ];
};
__device__ __forceinline__ float log_factor(float a, float m, float rr) {
	if (a <= 1e-3f) { const float a2 = a * a; return 2.f + a2 / 12.f + 7.f * a2 * a2 / 2880.f; }
	return a * rr / m;                          // a / sin(a/2)
}
__device__ __forceinline__ float exp_scale(float n, float she) {
	if (n <= 1e-3f) { const float n2 = n * n; return 0.5f - n2 / 48.f + n2 * n2 / 3840.f; }
	return she / n;
}
template <int NQ, bool SAVE>
__device__ __forceinline__ Q quat_spline_eval(const float* __restrict__ p, const adgs_func_eval& f, QuatSave<NQ>* s) {
	const int np = f.n_params, c0 = f.quat_start;
	Q prev = { 1.f, 0.f, 0.f, 0.f }, acc = prev;
#pragma unroll
	for (int j = 0; j < NQ; j++) {
		const Q c = { p[0 * np + c0 + j] + 1.0f, p[1 * np + c0 + j], p[2 * np + c0 + j], p[3 * np + c0 + j] };
		const float inv = 1.f / fmaxf(sqrtf(qdot(c, c)), 1e-12f);
		const Q q = { c.w * inv, c.x * inv, c.y * inv, c.z * inv };
		if (SAVE) { s->q[j] = q; s->inv_nrm[j] = inv; }
		if (j == 0) acc = q;
		else {
			Q dq = qmul(qconj(prev), q);
			const bool flip = dq.w < 0.f;
			if (flip) { dq.w = -dq.w; dq.x = -dq.x; dq.y = -dq.y; dq.z = -dq.z; }
			const float m2 = dq.x * dq.x + dq.y * dq.y + dq.z * dq.z;
			const float m = sqrtf(m2);
			const float a = 2.f * atan2f(m, dq.w);
			const float fl = log_factor(a, m, sqrtf(m2 + dq.w * dq.w));
			const float B = f.quat_cum[j - 1];
			const float rv0 = fl * dq.x * B, rv1 = fl * dq.y * B, rv2 = fl * dq.z * B;
			const float n = sqrtf(rv0 * rv0 + rv1 * rv1 + rv2 * rv2);
			float she, che;
			sincosf(n * 0.5f, &she, &che);
			const float sc = exp_scale(n, she);
			const Q r = { che, sc * rv0, sc * rv1, sc * rv2 };
			acc = qmul(acc, r);
			if (SAVE) {
				s->r[j] = r; s->vx[j] = dq.x; s->vy[j] = dq.y; s->vz[j] = dq.z; s->wq[j] = dq.w;
				s->a[j] = a; s->n[j] = n; s->she[j] = she; s->flip[j] = flip;
			}
		}
		prev = q;
	}
	return acc;
}

// backward: writes d/dparam of (g_out . out) into gp [4][n_params] (columns c0..c0+NQ-1), ASSIGNING
template <int NQ>
__device__ __forceinline__ void quat_spline_bwd(const QuatSave<NQ>& s, const Q& out, const adgs_func_eval& f, const Q& g_out, float* __restrict__ gp) {
	const int np = f.n_params, c0 = f.quat_start;
	Q gq[NQ];
#pragma unroll
	for (int j = 0; j < NQ; j++) gq[j] = { 0.f, 0.f, 0.f, 0.f };
	Q G = g_out;                 // gradient w.r.t. part[j] = q0 r1 ... rj, starting at j = NQ-1
	Q P = out;                   // part[j], walked back with part[j-1] = part[j] conj(r_j)
#pragma unroll
	for (int j = NQ - 1; j >= 1; j--) {
		const Q rc = qconj(s.r[j]);
		const Q Pm = qmul(P, rc);
		const Q g_r = qmul(qconj(Pm), G);              // d(part[j-1] * r_j) / d r_j
		G = qmul(G, rc);                               // -> gradient w.r.t. part[j-1]
		P = Pm;
		// exp map backward
		const float vx = s.vx[j], vy = s.vy[j], vz = s.vz[j], w = s.wq[j], a = s.a[j];
		const float m2 = vx * vx + vy * vy + vz * vz, m = sqrtf(m2), rr2 = m2 + w * w, rr = sqrtf(rr2);
		const float fl = log_factor(a, m, rr);
		const float B = f.quat_cum[j - 1];
		const float rv0 = fl * vx * B, rv1 = fl * vy * B, rv2 = fl * vz * B;
		const float n = s.n[j], she = s.she[j], che = s.r[j].w;
		const float sc = exp_scale(n, she);
		float dsc_over_n, sinc;
		if (n <= 1e-3f) { dsc_over_n = -1.f / 24.f + n * n / 960.f; sinc = (n > 0.f) ? she / n : 0.5f; }
		else { dsc_over_n = ((0.5f * che * n - she) / (n * n)) / n; sinc = she / n; }
		const float t = rv0 * g_r.x + rv1 * g_r.y + rv2 * g_r.z;
		const float k = dsc_over_n * t - 0.5f * sinc * g_r.w;
		const float go0 = (sc * g_r.x + rv0 * k) * B, go1 = (sc * g_r.y + rv1 * k) * B, go2 = (sc * g_r.z + rv2 * k) * B;
		// log map backward: sin(a/2) = m / rr, cos(a/2) = w / rr
		float fp;
		if (a <= 1e-3f) fp = a / 6.f + 7.f * a * a * a / 720.f;
		else { const float sh = m / rr, ch = w / rr; fp = (sh - 0.5f * a * ch) / (sh * sh); }
		const float sdot = vx * go0 + vy * go1 + vz * go2;
		const float da_dm = 2.f * w / rr2, da_dw = -2.f * m / rr2;
		const float kv = (m > 0.f) ? fp * sdot * da_dm / m : 0.f;
		Q g_d = { fp * sdot * da_dw, fl * go0 + kv * vx, fl * go1 + kv * vy, fl * go2 + kv * vz };
		if (s.flip[j]) { g_d.w = -g_d.w; g_d.x = -g_d.x; g_d.y = -g_d.y; g_d.z = -g_d.z; }
		gq[j] = qadd(gq[j], qmul(s.q[j - 1], g_d));                       // L(conj(q_{j-1}))^T g = q_{j-1} * g
		gq[j - 1] = qadd(gq[j - 1], qconj(qmul(g_d, qconj(s.q[j]))));     // through the conjugation
	}
	gq[0] = qadd(gq[0], G);
#pragma unroll
	for (int j = 0; j < NQ; j++) {
		const float dd = qdot(s.q[j], gq[j]);
		const float inv = s.inv_nrm[j];
		gp[0 * np + c0 + j] = (gq[j].w - s.q[j].w * dd) * inv;
		gp[1 * np + c0 + j] = (gq[j].x - s.q[j].x * dd) * inv;
		gp[2 * np + c0 + j] = (gq[j].y - s.q[j].y * dd) * inv;
		gp[3 * np + c0 + j] = (gq[j].z - s.q[j].z * dd) * inv;
	}
}

// run CALL(NQ) with the compile-time number of control quaternions of `f` (0: no quaternion spline)
#define ADGS_NQ_SWITCH(f, CALL) \
	switch ((f).quat_start >= 0 ? (f).quat_k + 1 : 0) { \
		case 0: CALL(0); break; case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break; \
		case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break; default: CALL(8); break; }

// ------------------------------------------------------------------ generic get_func_result
template <int D, int NQ>
__global__ void __launch_bounds__(256) func_eval_fwd_kernel(int N, const float* __restrict__ param, adgs_func_eval f, float* __restrict__ out) {
	const int n = blockIdx.x * blockDim.x + threadIdx.x;
	if (n >= N) return;
	const float* p = param + (size_t)n * D * f.n_params;
	float res[D];
#pragma unroll
	for (int d = 0; d < D; d++) res[d] = lin_eval(p + d * f.n_params, f);
	if (D == 4 && NQ > 0) {
		const Q qv = quat_spline_eval<(NQ > 0 ? NQ : 1), false>(p, f, nullptr);
		res[0] = res[0] + qv.w; res[1] = res[1] + qv.x; res[2] = res[2] + qv.y; res[3 % D] = res[3 % D] + qv.z;
	}
#pragma unroll
	for (int d = 0; d < D; d++) out[(size_t)n * D + d] = res[d];
}
template <int D, int NQ>
__global__ void __launch_bounds__(256) func_eval_bwd_kernel(int N, const float* __restrict__ param, adgs_func_eval f,
	const float* __restrict__ dL_dout, float* __restrict__ dL_dparam) {
	const int n = blockIdx.x * blockDim.x + threadIdx.x;
	if (n >= N) return;
	const float* p = param + (size_t)n * D * f.n_params;
	float* gp = dL_dparam + (size_t)n * D * f.n_params;
	float g[D];
#pragma unroll
	for (int d = 0; d < D; d++) { g[d] = dL_dout[(size_t)n * D + d]; lin_bwd(gp + d * f.n_params, f, g[d]); }
	if (D == 4 && NQ > 0) {
		QuatSave<(NQ > 0 ? NQ : 1)> st;
		const Q out = quat_spline_eval<(NQ > 0 ? NQ : 1), true>(p, f, &st);
		quat_spline_bwd<(NQ > 0 ? NQ : 1)>(st, out, f, { g[0], g[1], g[2], g[3 % D] }, gp);
	}
}

// ------------------------------------------------------------------ fused get_deformed_pkg
// One thread per Gaussian of [n_begin, n_end).  The object Gaussians' deformation rows ([3][Cx] for xyz,
// [4][Cr] for the rotation) are staged through LDS with coalesced loads (func_eval.h: stage_rows), one
// parameter family after the other in the same buffer; the host launches the scene range without LDS.
// xyz is evaluated at the camera time and, when flow_xyz is given, at the flow time from the same staged
// rows (gaussian_renderer/__init__.py:57 calls get_deformed_xyz a second time for it).
struct DeformArgs {
	adgs_deform_params p; adgs_func_eval fx, fr, fb, fx2, fb2; adgs_deform_outputs o; float* flow_xyz;
	int n_begin, n_end, stride_x, stride_r;
	int scene4;          // scene range as groups of four Gaussians with 16-byte accesses (every pointer 16-byte aligned)
	int xyz_rows;        // object xyz rows read directly (deform_fwd_xyz_rows) instead of staged through LDS
};

// ---- scene range, four Gaussians per thread: every tensor of the scene Gaussians is a plain elementwise map
// (copy + background offset, normalise, sigmoid, exp), and [N,3] rows read or written one Gaussian per thread are 12-byte
// strided accesses that reach ~1.9 TB/s (measured: 41.6 us for 800 k Gaussians, 80 MB).  Twelve floats = three float4:
// a thread that owns four consecutive Gaussians only ever issues aligned 16-byte loads and stores.
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add3(float4 v, float a, float b, float c, float d) { return make_float4(v.x + a, v.y + b, v.z + c, v.w + d); }

__device__ __forceinline__ void deform_fwd_scene_one(const DeformArgs& a, int n, const float* bg, const float* bg2) {
	// one scene Gaussian, scalar accesses (the < 4 Gaussians behind the last full group)
	for (int d = 0; d < 3; d++) {
		const float v = a.p.scene_xyz[3 * (size_t)n + d];
		if (a.o.xyz) a.o.xyz[3 * (size_t)n + d] = v + bg[d];
		if (a.flow_xyz) a.flow_xyz[3 * (size_t)n + d] = v + bg2[d];
	}
	if (a.o.rotation) {
		const float4 q = ld4(a.p.scene_rotation + 4 * (size_t)n);
		const float inv = 1.f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
		st4(a.o.rotation + 4 * (size_t)n, make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv));
	}
	if (a.o.opacity) a.o.opacity[n] = 1.f / (1.f + expf(-a.p.scene_opacity[n]));
	if (a.o.scales) for (int d = 0; d < 3; d++) a.o.scales[3 * (size_t)n + d] = expf(a.p.scene_scaling[3 * (size_t)n + d]);
}
__device__ __forceinline__ void deform_fwd_scene4(const DeformArgs& a, int blk) {
	const int g = blk * blockDim.x + threadIdx.x;          // group of Gaussians 4g .. 4g+3
	const int Ns = a.p.Ns;
	if (4 * (size_t)g >= (size_t)Ns) return;
	float bg[3] = { 0.f, 0.f, 0.f }, bg2[3] = { 0.f, 0.f, 0.f };
	if (a.p.background_deform_param) {
#pragma unroll
		for (int d = 0; d < 3; d++) {
			if (a.o.xyz && has_lin(a.fb)) bg[d] = lin_eval(a.p.background_deform_param + d * a.fb.n_params, a.fb);
			if (a.flow_xyz && has_lin(a.fb2)) bg2[d] = lin_eval(a.p.background_deform_param + d * a.fb2.n_params, a.fb2);
		}
	}
	if (4 * g + 4 > Ns) { for (int n = 4 * g; n < Ns; n++) deform_fwd_scene_one(a, n, bg, bg2); return; }
	const size_t o3 = 12 * (size_t)g, o4 = 16 * (size_t)g;
	if (a.o.xyz || a.flow_xyz) {
		const float4 v0 = ld4(a.p.scene_xyz + o3), v1 = ld4(a.p.scene_xyz + o3 + 4), v2 = ld4(a.p.scene_xyz + o3 + 8);
		// component pattern of twelve consecutive floats: x y z x | y z x y | z x y z   (v + 0.0 keeps v bit for bit)
		if (a.o.xyz) {
			st4(a.o.xyz + o3, add3(v0, bg[0], bg[1], bg[2], bg[0])); st4(a.o.xyz + o3 + 4, add3(v1, bg[1], bg[2], bg[0], bg[1]));
			st4(a.o.xyz + o3 + 8, add3(v2, bg[2], bg[0], bg[1], bg[2]));
		}
		if (a.flow_xyz) {
			st4(a.flow_xyz + o3, add3(v0, bg2[0], bg2[1], bg2[2], bg2[0])); st4(a.flow_xyz + o3 + 4, add3(v1, bg2[1], bg2[2], bg2[0], bg2[1]));
			st4(a.flow_xyz + o3 + 8, add3(v2, bg2[2], bg2[0], bg2[1], bg2[2]));
		}
	}
	if (a.o.rotation) {
		float4 q[4];
#pragma unroll
		for (int k = 0; k < 4; k++) q[k] = ld4(a.p.scene_rotation + o4 + 4 * k);
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const float inv = 1.f / fmaxf(sqrtf(q[k].x * q[k].x + q[k].y * q[k].y + q[k].z * q[k].z + q[k].w * q[k].w), 1e-12f);
			st4(a.o.rotation + o4 + 4 * k, make_float4(q[k].x * inv, q[k].y * inv, q[k].z * inv, q[k].w * inv));
		}
	}
	if (a.o.opacity) {
		const float4 x = ld4(a.p.scene_opacity + 4 * (size_t)g);
		st4(a.o.opacity + 4 * (size_t)g, make_float4(1.f / (1.f + expf(-x.x)), 1.f / (1.f + expf(-x.y)), 1.f / (1.f + expf(-x.z)), 1.f / (1.f + expf(-x.w))));
	}
	if (a.o.scales) {
#pragma unroll
		for (int k = 0; k < 3; k++) {
			const float4 v = ld4(a.p.scene_scaling + o3 + 4 * k);
			st4(a.o.scales + o3 + 4 * k, make_float4(expf(v.x), expf(v.y), expf(v.z), expf(v.w)));
		}
	}
}

// PARTS: bit 0 = xyz / flow, bit 1 = rotation, bit 2 = opacity + scales (the object range is covered twice, by an xyz block set and a
// rotation block set, so that the two LDS-staged phases run side by side instead of one after the other in every block)
constexpr int DP_XYZ = 1, DP_ROT = 2, DP_REST = 4;
template <int PARTS, bool OBJ, int NQ>
__device__ __forceinline__ void deform_fwd_body(const DeformArgs& a, int blk, int n_begin, int n_end, float* __restrict__ s_rows) {
	const int tid = threadIdx.x, B = blockDim.x, base = n_begin + blk * B;
	const int Ns = a.p.Ns;
	const int count = min(B, n_end - base);
	const int n = base + tid;
	const bool valid = tid < count;
	const bool is_obj = OBJ && valid && n >= Ns;
	const int m = is_obj ? n - Ns : n;
	const bool blk_obj = OBJ && base + count > Ns;   // block-uniform: some member is an object Gaussian
	// ---- xyz (gaussian_model.py:173-185), at t and at the flow time
	if ((PARTS & DP_XYZ) && (a.o.xyz || a.flow_xyz)) {
		const int np = a.fx.n_params;
		const bool lin1 = a.o.xyz && a.p.xyz_deform_param && has_lin(a.fx);
		const bool lin2 = a.flow_xyz && a.p.xyz_deform_param && has_lin(a.fx2);
		const bool staged = blk_obj && (lin1 || lin2);
		if (staged) {
			stage_rows<true>(s_rows, a.stride_x, 3 * np, base, count, Ns, (const float*)nullptr, a.p.xyz_deform_param, tid, B);
			__syncthreads();
		}
		if (valid) {
			float bg[3] = { 0.f, 0.f, 0.f }, bg2[3] = { 0.f, 0.f, 0.f };
			if (a.p.background_deform_param) {
#pragma unroll
				for (int d = 0; d < 3; d++) {
					if (a.o.xyz && has_lin(a.fb)) bg[d] = lin_eval(a.p.background_deform_param + d * a.fb.n_params, a.fb);
					if (a.flow_xyz && has_lin(a.fb2)) bg2[d] = lin_eval(a.p.background_deform_param + d * a.fb2.n_params, a.fb2);
				}
			}
			const float* bp = is_obj ? a.p.obj_xyz + 3 * (size_t)m : a.p.scene_xyz + 3 * (size_t)m;
			const float* row = s_rows + tid * a.stride_x;
#pragma unroll
			for (int d = 0; d < 3; d++) {
				const float v = bp[d];
				if (a.o.xyz) {
					float v1 = v;
					if (is_obj && lin1) v1 = v1 + lin_eval(row + d * np, a.fx);
					a.o.xyz[3 * (size_t)n + d] = v1 + bg[d];
				}
				if (a.flow_xyz) {
					float v2 = v;
					if (is_obj && lin2) v2 = v2 + lin_eval(row + d * np, a.fx2);
					a.flow_xyz[3 * (size_t)n + d] = v2 + bg2[d];
				}
			}
		}
		if (staged) __syncthreads();                  // the buffer is reused below
	}
	// ---- rotation (gaussian_model.py:187-196): normalize(cat(scene_rot, obj_rot))
	if ((PARTS & DP_ROT) && a.o.rotation) {
		const int np = a.fr.n_params;
		const bool staged = blk_obj && a.p.rotation_deform_param != nullptr;
		if (staged) {
			stage_rows<true>(s_rows, a.stride_r, 4 * np, base, count, Ns, (const float*)nullptr, a.p.rotation_deform_param, tid, B);
			__syncthreads();
		}
		if (valid) {
			float u[4];
			if (!OBJ || !is_obj) {
				const float4 q = *reinterpret_cast<const float4*>(a.p.scene_rotation + 4 * (size_t)m);
				u[0] = q.x; u[1] = q.y; u[2] = q.z; u[3] = q.w;
			} else {
				const float* rp = a.p.rotation_deform_param ? s_rows + tid * a.stride_r : nullptr;
				float fv[4] = { 0.f, 0.f, 0.f, 0.f };
				if (rp) {
#pragma unroll
					for (int d = 0; d < 4; d++) fv[d] = lin_eval(rp + d * np, a.fr);
					if (NQ > 0) { const Q qv = quat_spline_eval<(NQ > 0 ? NQ : 1), false>(rp, a.fr, nullptr); fv[0] += qv.w; fv[1] += qv.x; fv[2] += qv.y; fv[3] += qv.z; }
				}
				if (a.fr.quat_start >= 0) {
#pragma unroll
					for (int d = 0; d < 4; d++) u[d] = fv[d];          // quaternion spline replaces _obj_rotation (:189-190)
				} else {
#pragma unroll
					for (int d = 0; d < 4; d++) u[d] = a.p.obj_rotation[4 * (size_t)m + d] + fv[d];
				}
			}
			const float inv = 1.f / fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2] + u[3] * u[3]), 1e-12f);
			*reinterpret_cast<float4*>(a.o.rotation + 4 * (size_t)n) = make_float4(u[0] * inv, u[1] * inv, u[2] * inv, u[3] * inv);
		}
	}
	if (!valid || !(PARTS & DP_REST)) return;
	// ---- opacity (gaussian_model.py:207-214)
	if (a.o.opacity) {
		const float x = is_obj ? a.p.obj_opacity[m] : a.p.scene_opacity[m];
		float o = 1.f / (1.f + expf(-x));
		if (is_obj && (a.p.use_time_mask & ADGS_DEFORM_TIME_MASK)) {
			const float dt = a.p.t - a.p.gs_time[m];
			const float sig = expf(dt < 0.f ? a.p.gs_time_sigma[2 * (size_t)m] : a.p.gs_time_sigma[2 * (size_t)m + 1]);
			const float r = dt / sig;
			o = o * expf(-0.5f * (r * r));
		}
		a.o.opacity[n] = o;
	}
	// ---- scales (gaussian_model.py:89-91)
	if (a.o.scales) {
		const float* s = is_obj ? a.p.obj_scaling + 3 * (size_t)m : a.p.scene_scaling + 3 * (size_t)m;
#pragma unroll
		for (int d = 0; d < 3; d++) a.o.scales[3 * (size_t)n + d] = expf(s[d]);
	}
}


// Object xyz / flow rows without staging: one thread per (object Gaussian, component) ROW of xyz_deform_param, read as 8-byte
// words straight from global memory (consecutive threads own consecutive rows: every fetched line is fully used) and dotted
// with the dense basis vectors of the two time stamps; the [No,3] inputs and outputs are then plain coalesced 4-byte accesses.
__device__ __forceinline__ void deform_fwd_xyz_rows(const DeformArgs a, int blk, float* __restrict__ s_w) {   // by value: with a reference the compiler spills the whole argument struct to scratch (2.4 KB per lane)
	const int tid = threadIdx.x, B = blockDim.x, np = a.fx.n_params;
	const bool lin1 = a.o.xyz && has_lin(a.fx), lin2 = a.flow_xyz && has_lin(a.fx2);
	for (int k = tid; k < 2 * np; k += B) s_w[k] = 0.f;
	__syncthreads();
	{
		const int t1 = lin1 ? a.fx.n_terms[0] + a.fx.n_terms[1] + a.fx.n_terms[2] : 0;
		const int t2 = lin2 ? a.fx2.n_terms[0] + a.fx2.n_terms[1] + a.fx2.n_terms[2] : 0;
		// uniform loop index: a per-lane index into the by-value argument struct would force the whole struct into scratch memory
		for (int i = 0; i < t1; i++) if (tid == 0) s_w[a.fx.index[i]] = a.fx.weight[i];
		for (int i = 0; i < t2; i++) if (tid == 0) s_w[np + a.fx2.index[i]] = a.fx2.weight[i];
	}
	__syncthreads();
	const size_t r = (size_t)blk * B + tid;
	if (r >= 3 * (size_t)a.p.No) return;
	const int d = (int)(r % 3);
	const float2* row = reinterpret_cast<const float2*>(a.p.xyz_deform_param + r * np);
	float acc1 = 0.f, acc2 = 0.f;
#pragma unroll 3
	for (int k = 0; k < np / 2; k++) {
		const float2 v = row[k];
		// explicit fused multiply-adds in one fixed order: the camera-time and the flow-time sums must round identically, so that
		// evaluating a time stamp in either slot gives the same bits
		acc1 = fmaf(v.y, s_w[2 * k + 1], fmaf(v.x, s_w[2 * k], acc1));
		acc2 = fmaf(v.y, s_w[np + 2 * k + 1], fmaf(v.x, s_w[np + 2 * k], acc2));
	}
	float bg = 0.f, bg2 = 0.f;
	if (a.p.background_deform_param) {               // all three components with compile-time row offsets (as above: no per-lane struct access)
		float b1[3] = { 0.f, 0.f, 0.f }, b2[3] = { 0.f, 0.f, 0.f };
#pragma unroll
		for (int c = 0; c < 3; c++) {
			if (a.o.xyz && has_lin(a.fb)) b1[c] = lin_eval(a.p.background_deform_param + c * a.fb.n_params, a.fb);
			if (a.flow_xyz && has_lin(a.fb2)) b2[c] = lin_eval(a.p.background_deform_param + c * a.fb2.n_params, a.fb2);
		}
		bg = d == 0 ? b1[0] : (d == 1 ? b1[1] : b1[2]);
		bg2 = d == 0 ? b2[0] : (d == 1 ? b2[1] : b2[2]);
	}
	const float v = a.p.obj_xyz[r];
	const size_t o = 3 * (size_t)a.p.Ns + r;
	if (a.o.xyz) a.o.xyz[o] = (lin1 ? v + acc1 : v) + bg;
	if (a.flow_xyz) a.flow_xyz[o] = (lin2 ? v + acc2 : v) + bg2;
}

// ---- SH coefficients as flat, fully coalesced streams (the per-Gaussian 192-byte rows are the
// bulk of the deformation traffic; one thread per Gaussian would touch 64 cache lines per access)
struct ShsFwdArgs {
	int Ns, N, M;
	const float *scene_dc, *obj_dc, *scene_rest, *obj_rest, *sp_scene, *sp_obj;
	adgs_func_eval fs;
	float* out;
};
__global__ void __launch_bounds__(256) deform_shs_fwd_kernel(ShsFwdArgs a) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // one float4 of the output
	const int row = a.M * 3;                                             // floats per Gaussian (multiple of 4 for M = 16; generic path below)
	const size_t total = (size_t)a.N * row;
	const size_t e0 = i * 4;
	if (e0 >= total) return;
	float v[4];
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const size_t e = e0 + k;
		if (e >= total) { v[k] = 0.f; continue; }
		const int n = (int)(e / row), c = (int)(e % row);
		const bool is_obj = n >= a.Ns;
		const int m = is_obj ? n - a.Ns : n;
		if (c < 3) {
			float x = (is_obj ? a.obj_dc : a.scene_dc)[3 * (size_t)m + c];
			const float* sp = is_obj ? a.sp_obj : a.sp_scene;
			if (sp && has_lin(a.fs)) x = x + lin_eval(sp + ((size_t)m * 3 + c) * a.fs.n_params, a.fs);
			v[k] = x;
		} else {
			v[k] = (is_obj ? a.obj_rest : a.scene_rest)[(size_t)m * (row - 3) + (c - 3)];
		}
	}
	if (e0 + 4 <= total && (row & 3) == 0) *reinterpret_cast<float4*>(a.out + e0) = make_float4(v[0], v[1], v[2], v[3]);
	else { for (int k = 0; k < 4; k++) if (e0 + k < total) a.out[e0 + k] = v[k]; }
}

// M = 16 (the reference's max_sh_degree 3): coefficient 0 is written by the sh0 kernel with a 48-float stride; this kernel moves the 45 floats of
// `rest` per Gaussian -- one thread per 16-byte word of the output, constant divisions, consecutive lanes on consecutive addresses on both
// sides.  (The generic kernel above computes e / row and e % row with run-time divisors per ELEMENT and evaluates f_shs in every 12th lane:
// 328 us at C3 for 372 MB.)
__global__ void __launch_bounds__(256) shs_rest_interleave_kernel(int Ns, int N, const float* __restrict__ scene_rest, const float* __restrict__ obj_rest, float* __restrict__ out) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (size_t)N * 12) return;
	const int n = (int)(i / 12), q = (int)(i - (size_t)n * 12);
	const bool is_obj = n >= Ns;
	const float* r = (is_obj ? obj_rest : scene_rest) + (size_t)(is_obj ? n - Ns : n) * 45;
	float* o = out + (size_t)n * 48 + 4 * q;
	if (q == 0) { o[3] = r[0]; return; }
	const float a = r[4 * q - 3], b = r[4 * q - 2], c = r[4 * q - 1], d = r[4 * q];
	*reinterpret_cast<float4*>(o) = make_float4(a, b, c, d);
}
// the inverse for the gradient [N,16,3] -> dc / rest gradients
__global__ void __launch_bounds__(256) shs_grad_split_kernel(int Ns, int N, const float* __restrict__ g, float* __restrict__ g_scene_dc, float* __restrict__ g_obj_dc,
	float* __restrict__ g_scene_rest, float* __restrict__ g_obj_rest) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (size_t)N * 12) return;
	const int n = (int)(i / 12), q = (int)(i - (size_t)n * 12);
	const bool is_obj = n >= Ns;
	const size_t m = is_obj ? n - Ns : n;
	const float4 v = *reinterpret_cast<const float4*>(g + (size_t)n * 48 + 4 * q);
	float* rest = is_obj ? g_obj_rest : g_scene_rest;
	if (q == 0) {
		float* dc = is_obj ? g_obj_dc : g_scene_dc;
		if (dc) { dc[3 * m] = v.x; dc[3 * m + 1] = v.y; dc[3 * m + 2] = v.z; }
		if (rest) rest[m * 45] = v.w;
		return;
	}
	if (rest) { float* r = rest + m * 45 + 4 * q - 3; r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w; }
}

struct ShsBwdArgs {
	int Ns, N, M;
	const float* g;                                   // [N, M, 3]
	float *g_scene_dc, *g_obj_dc, *g_scene_rest, *g_obj_rest;
};
__global__ void __launch_bounds__(256) deform_shs_bwd_copy_kernel(ShsBwdArgs a) {
	const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int row = a.M * 3;
	if (e >= (size_t)a.N * row) return;
	const int n = (int)(e / row), c = (int)(e % row);
	const bool is_obj = n >= a.Ns;
	const int m = is_obj ? n - a.Ns : n;
	const float v = a.g[e];
	if (c < 3) { float* d = is_obj ? a.g_obj_dc : a.g_scene_dc; if (d) d[3 * (size_t)m + c] = v; }
	else { float* d = is_obj ? a.g_obj_rest : a.g_scene_rest; if (d) d[(size_t)m * (row - 3) + (c - 3)] = v; }
}
// d/d(param[n, c, k]) = w[k] * g[n, 0, c]: every column is written (zeros outside the active terms)
struct ParamGradArgs {
	int n0, count, D, gstride;                        // Gaussians [n0, n0+count) of g; D rows per Gaussian
	const float* g;                                   // g[(n0 + m) * gstride + d]
	float* out;                                       // [count, D, n_params]
	adgs_func_eval f;
	// optional second segment in the same launch (blocks >= nb0): the object side after the scene side
	int nb0, n0_b, count_b; const float* g_b; float* out_b;
	// ADAM instantiation: a segment whose slot is on (p != nullptr) applies the step to its parameter rows -- which have the layout
	// of `out` -- instead of storing the gradient (include/adgs_optim.h: adgs_sh_adam)
	AdamSlot adam = { nullptr, nullptr, nullptr, 0.f, 0.f }, adam_b = { nullptr, nullptr, nullptr, 0.f, 0.f };
	float beta1 = 0.f, beta2 = 0.f, eps = 0.f;
};
constexpr int PG_ITEMS = 8;                          // outputs per thread: two 16-byte stores, 256 float4 apart
// Every store instruction of a wave covers 1 KiB of consecutive bytes (lane i writes float4 number base + i; the thread's second
// float4 lies one block width further): 16 full 64-byte lines per instruction instead of 64 half-covered 32-byte pieces.
template <bool ADAM>
__global__ void __launch_bounds__(256) deform_lin_param_grad_kernel(ParamGradArgs a) {
	extern __shared__ float s_w[];
	const int np = a.f.n_params;
	// the segment's fields go into locals: writing to the by-value argument struct would move all of it into scratch memory
	const bool second = a.nb0 > 0 && (int)blockIdx.x >= a.nb0;
	const int seg_n0 = second ? a.n0_b : a.n0, seg_count = second ? a.count_b : a.count;
	const float* __restrict__ seg_g = second ? a.g_b : a.g;
	float* __restrict__ seg_out = second ? a.out_b : a.out;
	const size_t tot = (size_t)seg_count * a.D * np;
	const size_t blk0 = (size_t)(blockIdx.x - (second ? a.nb0 : 0)) * 256 * PG_ITEMS;
	// (Gaussian m, channel d, column k) of each half's first output and its upstream value: requested before the basis vector is
	// assembled in LDS, so that the load's round trip runs under the two barriers
	constexpr int HALVES = PG_ITEMS / 4;
	size_t e0[HALVES], m0[HALVES]; int k0[HALVES], d0[HALVES]; float gv0[HALVES];
#pragma unroll
	for (int half = 0; half < HALVES; half++) {
		e0[half] = blk0 + ((size_t)half * 256 + threadIdx.x) * 4;
		const size_t e = min(e0[half], tot - 1);
		size_t row;
		if (tot <= 0xffffffffull) { const uint32_t r32 = (uint32_t)e / (uint32_t)np; row = r32; k0[half] = (int)((uint32_t)e - r32 * (uint32_t)np); }
		else { row = e / np; k0[half] = (int)(e - row * np); }
		m0[half] = row / a.D; d0[half] = (int)(row - m0[half] * a.D);
		gv0[half] = seg_g[(seg_n0 + m0[half]) * (size_t)a.gstride + d0[half]];
	}
	// ADAM: the parameter and its moments, requested in the same round trip (clamped quad index: unconditional loads)
	float* const ad_p = ADAM ? (second ? a.adam_b.p : a.adam.p) : nullptr;
	float* const ad_m = ADAM ? (second ? a.adam_b.m : a.adam.m) : nullptr;
	float* const ad_v = ADAM ? (second ? a.adam_b.v : a.adam.v) : nullptr;
	const float ad_step = ADAM ? (second ? a.adam_b.step_size : a.adam.step_size) : 0.f, ad_ibc2 = ADAM ? (second ? a.adam_b.inv_bc2_sqrt : a.adam.inv_bc2_sqrt) : 0.f;
	float4 p4[HALVES], m4[HALVES], v4[HALVES];
	if (ADAM && ad_p) {
		const size_t last4 = tot >= 4 ? (tot - 4) & ~(size_t)3 : 0;
#pragma unroll
		for (int half = 0; half < HALVES; half++) {
			const size_t e = tot >= 4 ? min(e0[half], last4) : 0;
			if (tot >= 4) { p4[half] = ld_stream4(reinterpret_cast<const float4*>(ad_p + e)); m4[half] = ld_stream4(reinterpret_cast<const float4*>(ad_m + e)); v4[half] = ld_stream4(reinterpret_cast<const float4*>(ad_v + e)); }
		}
	}
	for (int k = threadIdx.x; k < np; k += blockDim.x) s_w[k] = 0.f;
	__syncthreads();
	const int total = a.f.n_terms[0] + a.f.n_terms[1] + a.f.n_terms[2];
	for (int i = threadIdx.x; i < total; i += blockDim.x) s_w[a.f.index[i]] = a.f.weight[i];
	__syncthreads();
#pragma unroll
	for (int half = 0; half < HALVES; half++) {
		if (e0[half] >= tot) return;
		size_t m = m0[half]; int k = k0[half], d = d0[half];
		float gv = gv0[half];
		float v[4];
#pragma unroll
		for (int it = 0; it < 4; it++) {
			v[it] = __fmul_rn(s_w[k], gv);       // rounded as the stored gradient is: the ADAM instantiation must not contract it into the update
			if (++k == np) {
				k = 0;
				if (++d == a.D) { d = 0; m++; }
				if (e0[half] + it + 1 < tot) gv = seg_g[(seg_n0 + m) * (size_t)a.gstride + d];
			}
		}
		if (ADAM && ad_p) {
			if (e0[half] + 4 <= tot) {
				adam_update4(p4[half], m4[half], v4[half], make_float4(v[0], v[1], v[2], v[3]), a.beta1, a.beta2, a.eps, ad_step, ad_ibc2);
				st_stream4(reinterpret_cast<float4*>(ad_p + e0[half]), p4[half]); st_stream4(reinterpret_cast<float4*>(ad_m + e0[half]), m4[half]); st_stream4(reinterpret_cast<float4*>(ad_v + e0[half]), v4[half]);
			} else for (int it = 0; it < 4 && e0[half] + it < tot; it++) {
				const size_t e = e0[half] + it;
				float pp = ad_p[e], mm = ad_m[e], vv = ad_v[e];
				adam_update(pp, mm, vv, v[it], a.beta1, a.beta2, a.eps, ad_step, ad_ibc2);
				ad_p[e] = pp; ad_m[e] = mm; ad_v[e] = vv;
			}
			continue;
		}
		if (e0[half] + 4 <= tot) st_stream4(reinterpret_cast<float4*>(seg_out + e0[half]), make_float4(v[0], v[1], v[2], v[3]));
		else for (int it = 0; it < 4 && e0[half] + it < tot; it++) seg_out[e0[half] + it] = v[it];
	}
}

struct DeformBwdArgs {
	adgs_deform_params p; adgs_func_eval fx, fr, fb, fx2, fb2;
	const float *g_xyz, *g_flow, *g_rot, *g_op, *g_sc;
	adgs_deform_grads g;
	int n_begin, n_end, stride_x, stride_r;
	int scene4;          // as in DeformArgs
};

// Backward of deform_fwd_kernel.  The object Gaussians' parameter-gradient rows are assembled in LDS (every
// column written: zeros outside the active terms, so no caller zero-fill) and stored coalesced; the rotation
// rows need the parameters (staged in) and a second LDS region for the gradients.
// PARTS: bit 0 = xyz / background, bit 1 = rotation, bit 2 = opacity + scales.  OBJ = false is the scene
// range (no object member: none of the staging / spline code is instantiated, so the kernel stays light).
template <int PARTS, bool OBJ, int NQ>
__device__ __forceinline__ void deform_bwd_body(const DeformBwdArgs& a, int blk, int n_begin, int n_end, float* __restrict__ s_rows, float (*s_bg)[256 / WAVE]) {
	const int tid = threadIdx.x, B = blockDim.x, base = n_begin + blk * B;
	const int Ns = a.p.Ns;
	const int count = min(B, n_end - base);
	const int n = base + tid;
	const bool valid = tid < count;
	const bool is_obj = OBJ && valid && n >= Ns;
	const int m = is_obj ? n - Ns : n;
	const bool blk_obj = OBJ && base + count > Ns;
	// ---- xyz: upstream of the camera-time points and of the flow-time points
	if ((PARTS & DP_XYZ) && (a.g_xyz || a.g_flow)) {
		float gx[3] = { 0.f, 0.f, 0.f }, gf[3] = { 0.f, 0.f, 0.f };
		if (valid) {
#pragma unroll
			for (int d = 0; d < 3; d++) {
				if (a.g_xyz) gx[d] = a.g_xyz[3 * (size_t)n + d];
				if (a.g_flow) gf[d] = a.g_flow[3 * (size_t)n + d];
			}
			float* dst = is_obj ? (a.g.obj_xyz ? a.g.obj_xyz + 3 * (size_t)m : nullptr) : (a.g.scene_xyz ? a.g.scene_xyz + 3 * (size_t)m : nullptr);
			if (dst) { dst[0] = gx[0] + gf[0]; dst[1] = gx[1] + gf[1]; dst[2] = gx[2] + gf[2]; }
		}
		const int np = a.fx.n_params;
		if (blk_obj && a.g.xyz_deform_param) {
			// dense basis rows of the two time stamps behind the staging rows: d/dparam[n,d,k] = w1[k] g1[d] + w2[k] g2[d]
			float* s_w = s_rows + B * a.stride_x;
			for (int k = tid; k < 2 * np; k += B) s_w[k] = 0.f;
			__syncthreads();
			{
				const int t1 = a.g_xyz ? a.fx.n_terms[0] + a.fx.n_terms[1] + a.fx.n_terms[2] : 0;
				const int t2 = a.g_flow ? a.fx2.n_terms[0] + a.fx2.n_terms[1] + a.fx2.n_terms[2] : 0;
				for (int i = tid; i < t1; i += B) s_w[a.fx.index[i]] = a.fx.weight[i];
				for (int i = tid; i < t2; i += B) s_w[np + a.fx2.index[i]] = a.fx2.weight[i];
			}
			__syncthreads();
			if (is_obj) {
				float* row = s_rows + tid * a.stride_x;
				for (int k = 0; k < np; k++) {
					const float w1 = s_w[k], w2 = s_w[np + k];
#pragma unroll
					for (int d = 0; d < 3; d++) row[d * np + k] = w1 * gx[d] + w2 * gf[d];
				}
			}
			__syncthreads();
			stage_rows<false>(s_rows, a.stride_x, 3 * np, base, count, Ns, (float*)nullptr, a.g.xyz_deform_param, tid, B);
			__syncthreads();
		}
		// background: the same [1,3,Cb] row is added to every Gaussian -> reduce g over all n
		if (a.g.background_deform_param && (has_lin(a.fb) || has_lin(a.fb2))) {
#pragma unroll
			for (int d = 0; d < 6; d++) {
				float v = d < 3 ? gx[d % 3] : gf[d % 3];
#pragma unroll
				for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
				if ((tid & (WAVE - 1)) == 0) s_bg[d][tid / WAVE] = v;
			}
			__syncthreads();
			if (tid < 6) {
				float v = 0.f;
				for (int w = 0; w < (B + WAVE - 1) / WAVE; w++) v += s_bg[tid][w];
				const adgs_func_eval& f = tid < 3 ? a.fb : a.fb2;
				const bool on = tid < 3 ? (a.g_xyz != nullptr) : (a.g_flow != nullptr);
				const int total = f.n_terms[0] + f.n_terms[1] + f.n_terms[2];
				if (on) for (int i = 0; i < total; i++)
					atomicAdd(a.g.background_deform_param + (tid % 3) * f.n_params + f.index[i], f.weight[i] * v);
			}
			__syncthreads();
		}
	}
	// ---- rotation: r = u / |u|
	if ((PARTS & DP_ROT) && a.g_rot) {
		const int np = a.fr.n_params;
		const bool haver = a.p.rotation_deform_param != nullptr;
		const bool staged = blk_obj && haver;
		float* s_out = s_rows;           // gradient rows, in place: a thread is done with its parameter row (spline state saved in registers) before it writes its gradient row, and two sets of rows halved the waves a CU can hold
		if (staged) {
			stage_rows<true>(s_rows, a.stride_r, 4 * np, base, count, Ns, (const float*)nullptr, a.p.rotation_deform_param, tid, B);
			__syncthreads();
		}
		if (valid) {
			float u[4];
			const float* rp = (is_obj && haver) ? s_rows + tid * a.stride_r : nullptr;
			QuatSave<(NQ > 0 ? NQ : 1)> qs;
			Q qout = { 0.f, 0.f, 0.f, 0.f };
			if (!OBJ || !is_obj) {
				const float4 q = *reinterpret_cast<const float4*>(a.p.scene_rotation + 4 * (size_t)m);
				u[0] = q.x; u[1] = q.y; u[2] = q.z; u[3] = q.w;
			} else {
				float fv[4] = { 0.f, 0.f, 0.f, 0.f };
				if (rp) {
#pragma unroll
					for (int d = 0; d < 4; d++) fv[d] = lin_eval(rp + d * np, a.fr);
					if (NQ > 0) { qout = quat_spline_eval<(NQ > 0 ? NQ : 1), true>(rp, a.fr, &qs); fv[0] += qout.w; fv[1] += qout.x; fv[2] += qout.y; fv[3] += qout.z; }
				}
#pragma unroll
				for (int d = 0; d < 4; d++) u[d] = (a.fr.quat_start >= 0) ? fv[d] : a.p.obj_rotation[4 * (size_t)m + d] + fv[d];
			}
			const float nr = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2] + u[3] * u[3]);
			const float inv = 1.f / fmaxf(nr, 1e-12f);
			float g[4], r[4], dot = 0.f;
			const float4 g4 = *reinterpret_cast<const float4*>(a.g_rot + 4 * (size_t)n);
			g[0] = g4.x; g[1] = g4.y; g[2] = g4.z; g[3] = g4.w;
#pragma unroll
			for (int d = 0; d < 4; d++) { r[d] = u[d] * inv; dot += r[d] * g[d]; }
			float gu[4];
#pragma unroll
			for (int d = 0; d < 4; d++) gu[d] = (nr > 1e-12f) ? (g[d] - r[d] * dot) * inv : g[d] * inv;
			if (!OBJ || !is_obj) {
				if (a.g.scene_rotation) *reinterpret_cast<float4*>(a.g.scene_rotation + 4 * (size_t)m) = make_float4(gu[0], gu[1], gu[2], gu[3]);
			} else {
				if (a.g.obj_rotation) {
					const float k = (a.fr.quat_start >= 0) ? 0.f : 1.f;
					*reinterpret_cast<float4*>(a.g.obj_rotation + 4 * (size_t)m) = make_float4(k * gu[0], k * gu[1], k * gu[2], k * gu[3]);
				}
				if (a.g.rotation_deform_param && rp) {
					float* gp = s_out + tid * a.stride_r;
					for (int k = 0; k < 4 * np; k++) gp[k] = 0.f;
#pragma unroll
					for (int d = 0; d < 4; d++) lin_bwd(gp + d * np, a.fr, gu[d]);
					if (NQ > 0) quat_spline_bwd<(NQ > 0 ? NQ : 1)>(qs, qout, a.fr, { gu[0], gu[1], gu[2], gu[3] }, gp);
				}
			}
		}
		if (staged && a.g.rotation_deform_param) {
			__syncthreads();
			stage_rows<false>(s_out, a.stride_r, 4 * np, base, count, Ns, (float*)nullptr, a.g.rotation_deform_param, tid, B);
		}
	}
	if (!valid || !(PARTS & DP_REST)) return;
	// ---- opacity
	if (a.g_op) {
		const float g = a.g_op[n];
		const float x = is_obj ? a.p.obj_opacity[m] : a.p.scene_opacity[m];
		const float sg = 1.f / (1.f + expf(-x));
		float mask = 1.f;
		if (is_obj && (a.p.use_time_mask & ADGS_DEFORM_TIME_MASK)) {
			const float dt = a.p.t - a.p.gs_time[m];
			const bool neg = dt < 0.f;
			const float sig = expf(neg ? a.p.gs_time_sigma[2 * (size_t)m] : a.p.gs_time_sigma[2 * (size_t)m + 1]);
			const float r = dt / sig;
			mask = expf(-0.5f * (r * r));
			if (a.g.gs_time_sigma) {
				const float gsel = g * sg * mask * (r * r);      // d mask / d log-sigma = mask * (dt/sigma)^2
				a.g.gs_time_sigma[2 * (size_t)m] = neg ? gsel : 0.f;
				a.g.gs_time_sigma[2 * (size_t)m + 1] = neg ? 0.f : gsel;
			}
		} else if (is_obj && a.g.gs_time_sigma) {
			a.g.gs_time_sigma[2 * (size_t)m] = 0.f; a.g.gs_time_sigma[2 * (size_t)m + 1] = 0.f;
		}
		float* dst = is_obj ? a.g.obj_opacity : a.g.scene_opacity;
		if (dst) dst[m] = g * mask * sg * (1.f - sg);
	}
	// ---- scales
	if (a.g_sc) {
		const float* s = is_obj ? a.p.obj_scaling + 3 * (size_t)m : a.p.scene_scaling + 3 * (size_t)m;
		float* dst = is_obj ? (a.g.obj_scaling ? a.g.obj_scaling + 3 * (size_t)m : nullptr) : (a.g.scene_scaling ? a.g.scene_scaling + 3 * (size_t)m : nullptr);
		if (dst) {
#pragma unroll
			for (int d = 0; d < 3; d++) dst[d] = a.g_sc[3 * (size_t)n + d] * expf(s[d]);
		}
	}
}

// Backward of deform_fwd_scene4: four scene Gaussians per thread, 16-byte accesses only.
__device__ __forceinline__ float4 rot_norm_bwd(float4 q, float4 g) {
	const float nr = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
	const float inv = 1.f / fmaxf(nr, 1e-12f);
	const float r0 = q.x * inv, r1 = q.y * inv, r2 = q.z * inv, r3 = q.w * inv;
	const float dot = ((r0 * g.x + r1 * g.y) + r2 * g.z) + r3 * g.w;
	if (nr > 1e-12f) return make_float4((g.x - r0 * dot) * inv, (g.y - r1 * dot) * inv, (g.z - r2 * dot) * inv, (g.w - r3 * dot) * inv);
	return make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
}
__device__ __forceinline__ float sig_bwd(float g, float x) { const float sg = 1.f / (1.f + expf(-x)); return g * 1.f * sg * (1.f - sg); }
__device__ __forceinline__ void deform_bwd_scene4(const DeformBwdArgs& a, int blk, float (*s_bg)[256 / WAVE]) {
	const int tid = threadIdx.x, B = blockDim.x;
	const int g = blk * B + tid;
	const int Ns = a.p.Ns;
	const bool in_range = 4 * (size_t)g < (size_t)Ns;
	const bool full = in_range && 4 * g + 4 <= Ns;
	const int n0 = 4 * g, n1 = in_range ? min(Ns, n0 + 4) : n0;
	const size_t o3 = 12 * (size_t)g;
	float sx[3] = { 0.f, 0.f, 0.f }, sf[3] = { 0.f, 0.f, 0.f };      // per-component sums of the thread's upstream position gradients
	if (a.g_xyz || a.g_flow) {
		if (full) {
			const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
			const float4 x0 = a.g_xyz ? ld4(a.g_xyz + o3) : z, x1 = a.g_xyz ? ld4(a.g_xyz + o3 + 4) : z, x2 = a.g_xyz ? ld4(a.g_xyz + o3 + 8) : z;
			const float4 f0 = a.g_flow ? ld4(a.g_flow + o3) : z, f1 = a.g_flow ? ld4(a.g_flow + o3 + 4) : z, f2 = a.g_flow ? ld4(a.g_flow + o3 + 8) : z;
			if (a.g.scene_xyz) {
				st4(a.g.scene_xyz + o3, make_float4(x0.x + f0.x, x0.y + f0.y, x0.z + f0.z, x0.w + f0.w));
				st4(a.g.scene_xyz + o3 + 4, make_float4(x1.x + f1.x, x1.y + f1.y, x1.z + f1.z, x1.w + f1.w));
				st4(a.g.scene_xyz + o3 + 8, make_float4(x2.x + f2.x, x2.y + f2.y, x2.z + f2.z, x2.w + f2.w));
			}
			// x y z x | y z x y | z x y z
			sx[0] = ((x0.x + x0.w) + x1.z) + x2.y; sx[1] = ((x0.y + x1.x) + x1.w) + x2.z; sx[2] = ((x0.z + x1.y) + x2.x) + x2.w;
			sf[0] = ((f0.x + f0.w) + f1.z) + f2.y; sf[1] = ((f0.y + f1.x) + f1.w) + f2.z; sf[2] = ((f0.z + f1.y) + f2.x) + f2.w;
		} else {
			for (int n = n0; n < n1; n++)
				for (int d = 0; d < 3; d++) {
					const float gx = a.g_xyz ? a.g_xyz[3 * (size_t)n + d] : 0.f, gf = a.g_flow ? a.g_flow[3 * (size_t)n + d] : 0.f;
					if (a.g.scene_xyz) a.g.scene_xyz[3 * (size_t)n + d] = gx + gf;
					sx[d] += gx; sf[d] += gf;
				}
		}
		// background: the same [1,3,Cb] row is added to every Gaussian -> reduce g over all n
		if (a.g.background_deform_param && (has_lin(a.fb) || has_lin(a.fb2))) {
#pragma unroll
			for (int d = 0; d < 6; d++) {
				float v = d < 3 ? sx[d % 3] : sf[d % 3];
#pragma unroll
				for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
				if ((tid & (WAVE - 1)) == 0) s_bg[d][tid / WAVE] = v;
			}
			__syncthreads();
			if (tid < 6) {
				float v = 0.f;
				for (int w = 0; w < (B + WAVE - 1) / WAVE; w++) v += s_bg[tid][w];
				const adgs_func_eval& f = tid < 3 ? a.fb : a.fb2;
				const bool on = tid < 3 ? (a.g_xyz != nullptr) : (a.g_flow != nullptr);
				const int total = f.n_terms[0] + f.n_terms[1] + f.n_terms[2];
				if (on) for (int i = 0; i < total; i++)
					atomicAdd(a.g.background_deform_param + (tid % 3) * f.n_params + f.index[i], f.weight[i] * v);
			}
		}
	}
	if (!in_range) return;
	if (a.g_rot && a.g.scene_rotation) {
		for (int n = n0; n < n1; n++)
			st4(a.g.scene_rotation + 4 * (size_t)n, rot_norm_bwd(ld4(a.p.scene_rotation + 4 * (size_t)n), ld4(a.g_rot + 4 * (size_t)n)));
	}
	if (a.g_op && a.g.scene_opacity) {
		if (full) {
			const float4 gq = ld4(a.g_op + 4 * (size_t)g), x = ld4(a.p.scene_opacity + 4 * (size_t)g);
			st4(a.g.scene_opacity + 4 * (size_t)g, make_float4(sig_bwd(gq.x, x.x), sig_bwd(gq.y, x.y), sig_bwd(gq.z, x.z), sig_bwd(gq.w, x.w)));
		} else for (int n = n0; n < n1; n++) a.g.scene_opacity[n] = sig_bwd(a.g_op[n], a.p.scene_opacity[n]);
	}
	if (a.g_sc && a.g.scene_scaling) {
		if (full) {
#pragma unroll
			for (int k = 0; k < 3; k++) {
				const float4 gq = ld4(a.g_sc + o3 + 4 * k), v = ld4(a.p.scene_scaling + o3 + 4 * k);
				st4(a.g.scene_scaling + o3 + 4 * k, make_float4(gq.x * expf(v.x), gq.y * expf(v.y), gq.z * expf(v.z), gq.w * expf(v.w)));
			}
		} else for (int n = n0; n < n1; n++) for (int d = 0; d < 3; d++) a.g.scene_scaling[3 * (size_t)n + d] = a.g_sc[3 * (size_t)n + d] * expf(a.p.scene_scaling[3 * (size_t)n + d]);
	}
}

// One launch per direction: the object Gaussians' blocks (splines: long dependent chains, few waves) come FIRST in the
// grid and the streaming scene blocks fill the rest of the chip around them -- separate launches would run the
// latency-bound object kernels on a mostly idle GPU.
template <int NQ>
__global__ void __launch_bounds__(256) deform_fwd_kernel(DeformArgs a, int nb_rot, int nb_xyz) {
	extern __shared__ float s_rows[];
	const int b = blockIdx.x;
	if (b < nb_rot) deform_fwd_body<DP_ROT | DP_REST, true, NQ>(a, b, a.p.Ns, a.p.Ns + a.p.No, s_rows);
	else if (b < nb_rot + nb_xyz) {
		if (a.xyz_rows) deform_fwd_xyz_rows(a, b - nb_rot, s_rows);
		else deform_fwd_body<DP_XYZ, true, 0>(a, b - nb_rot, a.p.Ns, a.p.Ns + a.p.No, s_rows);
	}
	else if (a.scene4) deform_fwd_scene4(a, b - nb_rot - nb_xyz);
	else deform_fwd_body<DP_XYZ | DP_ROT | DP_REST, false, 0>(a, b - nb_rot - nb_xyz, 0, a.p.Ns, s_rows);
}
template <int NQ>
__global__ void __launch_bounds__(256) deform_bwd_kernel(DeformBwdArgs a, int nb_rot, int nb_xyz) {
	extern __shared__ float s_rows[];
	__shared__ float s_bg[6][256 / WAVE];
	const int b = blockIdx.x;
	if (b < nb_rot) deform_bwd_body<DP_ROT, true, NQ>(a, b, a.p.Ns, a.p.Ns + a.p.No, s_rows, s_bg);
	else if (b < nb_rot + nb_xyz) deform_bwd_body<DP_XYZ | DP_REST, true, 0>(a, b - nb_rot, a.p.Ns, a.p.Ns + a.p.No, s_rows, s_bg);
	else if (a.scene4) deform_bwd_scene4(a, b - nb_rot - nb_xyz, s_bg);
	else deform_bwd_body<DP_XYZ | DP_ROT | DP_REST, false, 0>(a, b - nb_rot - nb_xyz, 0, a.p.Ns, s_rows, s_bg);
}

// the four-Gaussians-per-thread scene path needs every (non-NULL) scene pointer 16-byte aligned
static int scene4_ok(std::initializer_list<const void*> ptrs) {
	for (const void* q : ptrs) if (reinterpret_cast<uintptr_t>(q) & 15) return 0;
	return 1;
}
static adgs_func_eval empty_func() { adgs_func_eval f; memset(&f, 0, sizeof(f)); f.quat_start = -1; return f; }
static int check_func(const adgs_func_eval* f, const char* what) {
	if (!f) return 0;
	const int total = f->n_terms[0] + f->n_terms[1] + f->n_terms[2];
	if (total < 0 || total > ADGS_FUNC_MAX_TERMS) { set_error(std::string(what) + ": too many basis terms"); return -1; }
	for (int i = 0; i < total; i++) if (f->index[i] < 0 || f->index[i] >= f->n_params) { set_error(std::string(what) + ": term index out of range"); return -1; }
	if (f->quat_start >= 0 && (f->quat_k + 1 > MAXQ || f->quat_k < 0 || f->quat_start + f->quat_k >= f->n_params)) {
		set_error(std::string(what) + ": quaternion spline window out of range"); return -1;
	}
	return 0;
}

} // namespace
} // namespace adgs

namespace adgs {
namespace {
// coefficient 0 of the raw-SH path: sh0[n, c] = dc[n, c] + f_shs(t)(shs_deform_param[n, c, :]).
// Fast kernel (n_params a multiple of 4, at most 32): one thread per (Gaussian, channel) ROW, which it reads as 16-byte
// words -- consecutive threads own consecutive rows, so the loads are perfectly coalesced without any staging -- and dots
// with the dense basis vector (zero where the family has no term).  General kernel: rows staged through LDS.
template <int NV4>
__global__ void __launch_bounds__(256) sh0_rows_kernel(int N, ShSource s, float* __restrict__ out, FramePrologue pro, int ostride) {
	__shared__ float s_w[NV4 * 4];
	run_frame_prologue(pro);      // counters the binning kernels accumulate into, the frame's snapshot of the slab bounds (saves a launch)
	__shared__ float s_part[256 * NV4 + 4];
	// The block owns 256 consecutive rows = 256 * NV4 consecutive 16-byte words of one segment (scene or object side).  Lane i of a
	// load instruction reads word base + i -- 1 KiB of consecutive bytes per wave and instruction instead of 64 words 16 * NV4 bytes
	// apart --, dots it with its quarter of the basis vector, and the NV4 partial sums of a row meet in LDS.  A block that straddles
	// the scene / object boundary takes the row-per-thread path.  The loads are issued BEFORE the basis vector is assembled in LDS
	// (two barriers): their round trip runs under that prologue.
	const int e0 = blockIdx.x * 256;                       // first row (row = n * 3 + c)
	const int rows = min(256, N * 3 - e0);
	const int split = 3 * s.Ns;                            // rows [0, split): scene side
	const bool one_side = (e0 >= split) || (e0 + rows <= split);
	const bool ob = e0 >= split;
	const float* sp = ob ? s.obj_sp : s.scene_sp;
	const bool fast = rows > 0 && one_side && sp != nullptr;            // block-uniform
	const size_t r0 = ob ? (size_t)e0 - split : (size_t)e0;
	float4 q[NV4];
	float dc = 0.f;
	if (fast) {
		const float4* src = reinterpret_cast<const float4*>(sp + r0 * (NV4 * 4));
		const int words = rows * NV4;
#pragma unroll
		for (int k = 0; k < NV4; k++) { const int wd = k * 256 + threadIdx.x; q[k] = ld_stream4(src + min(wd, words - 1)); }
		dc = (ob ? s.obj_dc : s.scene_dc)[r0 + min((int)threadIdx.x, rows - 1)];
	}
	for (int k = threadIdx.x; k < NV4 * 4; k += 256) s_w[k] = 0.f;
	__syncthreads();
	const int total = s.f.n_terms[0] + s.f.n_terms[1] + s.f.n_terms[2];
	for (int i = threadIdx.x; i < total; i += 256) s_w[s.f.index[i]] = s.f.weight[i];
	__syncthreads();
	if (rows <= 0) return;
	if (fast) {
#pragma unroll
		for (int k = 0; k < NV4; k++) {
			const int wd = k * 256 + threadIdx.x, part = wd % NV4;
			s_part[wd] = q[k].x * s_w[4 * part] + q[k].y * s_w[4 * part + 1] + q[k].z * s_w[4 * part + 2] + q[k].w * s_w[4 * part + 3];
		}
		__syncthreads();
		if ((int)threadIdx.x < rows) {
			float acc = 0.f;
#pragma unroll
			for (int k = 0; k < NV4; k++) acc += s_part[threadIdx.x * NV4 + k];
			const int e = e0 + threadIdx.x, n = e / 3;
			out[(size_t)n * ostride + (e - 3 * n)] = dc + acc;      // ostride = 3: the compact [N,3] array; 3 M: coefficient 0 inside [N,M,3]
		}
		return;
	}
	const int e = e0 + threadIdx.x;
	if (e >= N * 3) return;
	const int n = e / 3;
	const bool tob = n >= s.Ns;
	const size_t r = tob ? (size_t)e - 3 * (size_t)s.Ns : (size_t)e;
	float v = (tob ? s.obj_dc : s.scene_dc)[r];
	const float* tsp = tob ? s.obj_sp : s.scene_sp;
	if (tsp) {
		const float4* row = reinterpret_cast<const float4*>(tsp + r * (NV4 * 4));
		float4 q[NV4];
#pragma unroll
		for (int k = 0; k < NV4; k++) q[k] = row[k];
		float acc = 0.f;
#pragma unroll
		for (int k = 0; k < NV4; k++) acc += q[k].x * s_w[4 * k] + q[k].y * s_w[4 * k + 1] + q[k].z * s_w[4 * k + 2] + q[k].w * s_w[4 * k + 3];
		v = v + acc;
	}
	out[(size_t)n * ostride + (e - 3 * n)] = v;
}
__global__ void __launch_bounds__(256) sh0_kernel(int N, ShSource s, float* __restrict__ out, FramePrologue pro, int ostride) {
	extern __shared__ float s_rows[];
	run_frame_prologue(pro);
	const int tid = threadIdx.x, B = blockDim.x, base = blockIdx.x * B, count = min(B, N - base);
	const int np = s.f.n_params, L = 3 * np, stride = L | 1;
	const bool lin = (s.scene_sp || s.obj_sp) && has_lin(s.f);
	if (lin) {
		stage_rows<true>(s_rows, stride, L, base, count, s.Ns, s.scene_sp, s.obj_sp, tid, B);
		__syncthreads();
	}
	if (tid >= count) return;
	const int n = base + tid;
	const bool ob = n >= s.Ns;
	const size_t m = ob ? n - s.Ns : n;
	const float* dc = (ob ? s.obj_dc : s.scene_dc) + 3 * m;
	const bool has_row = lin && (ob ? s.obj_sp : s.scene_sp) != nullptr;
#pragma unroll
	for (int c = 0; c < 3; c++) {
		float v = dc[c];
		if (has_row) v = v + lin_eval(s_rows + tid * stride + c * np, s.f);
		out[(size_t)ostride * n + c] = v;
	}
}
} // namespace
constexpr size_t MAX_STAGING_LDS = 156 * 1024;
// block size / LDS of the staged geometry kernels: the largest block whose rows fit 48 KiB; one-wave blocks may take up to
// MAX_STAGING_LDS of the CU's 160 KiB (very long parameter rows: B-spline + polynomial + Fourier + quaternion parts together)
static int pick_block(int row_floats, size_t* lds) {
	for (int B = 256; B >= 64; B >>= 1) {
		const size_t bytes = (size_t)B * row_floats * sizeof(float);
		if (bytes <= 48 * 1024 || B == 64) { *lds = bytes; return B; }
	}
	return 64;
}

int launch_sh0(int N, const ShSource& s, float* out, hipStream_t stream, const FramePrologue* prologue, int ostride) {
	if (N <= 0) return 0;
	const FramePrologue pro = prologue ? *prologue : FramePrologue{ nullptr, 0, nullptr, nullptr, 0 };
	const int np = s.f.n_params;
	const bool lin = (s.scene_sp || s.obj_sp) && (s.f.n_terms[0] + s.f.n_terms[1] + s.f.n_terms[2]) > 0 && np > 0;
	const bool aligned = ((reinterpret_cast<uintptr_t>(s.scene_sp) | reinterpret_cast<uintptr_t>(s.obj_sp)) & 15) == 0;
	if (lin && np % 4 == 0 && np <= 32 && aligned) {
		const unsigned blocks = (unsigned)(((size_t)N * 3 + 255) / 256);
		switch (np / 4) {
			case 1: hipLaunchKernelGGL(sh0_rows_kernel<1>, dim3(blocks), dim3(256), 0, stream, N, s, out, pro, ostride); break;
			case 2: hipLaunchKernelGGL(sh0_rows_kernel<2>, dim3(blocks), dim3(256), 0, stream, N, s, out, pro, ostride); break;
			case 3: hipLaunchKernelGGL(sh0_rows_kernel<3>, dim3(blocks), dim3(256), 0, stream, N, s, out, pro, ostride); break;
			case 4: hipLaunchKernelGGL(sh0_rows_kernel<4>, dim3(blocks), dim3(256), 0, stream, N, s, out, pro, ostride); break;
			case 5: hipLaunchKernelGGL(sh0_rows_kernel<5>, dim3(blocks), dim3(256), 0, stream, N, s, out, pro, ostride); break;
			case 6: hipLaunchKernelGGL(sh0_rows_kernel<6>, dim3(blocks), dim3(256), 0, stream, N, s, out, pro, ostride); break;
			case 7: hipLaunchKernelGGL(sh0_rows_kernel<7>, dim3(blocks), dim3(256), 0, stream, N, s, out, pro, ostride); break;
			default: hipLaunchKernelGGL(sh0_rows_kernel<8>, dim3(blocks), dim3(256), 0, stream, N, s, out, pro, ostride); break;
		}
		ADGS_HIP_CHECK(hipGetLastError());
		return 0;
	}
	size_t lds = 0;
	const int B = pick_block((3 * np) | 1, &lds);
	if (lds > MAX_STAGING_LDS) { set_error("launch_sh0: SH deformation rows too large for the LDS staging buffer (more than 207 parameters per channel)"); return -1; }
	if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sh0_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	hipLaunchKernelGGL(sh0_kernel, dim3((unsigned)((N + B - 1) / B)), dim3(B), lds, stream, N, s, out, pro, ostride);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
int launch_lin_param_grad(int count, int D, const float* g, int gstride, float* out, const adgs_func_eval& f, hipStream_t stream);
// two segments (scene side, object side) in ONE launch: a kernel boundary costs ~5 us, as much as a quarter of one side's work
int launch_lin_param_grad2(int count_a, const float* g_a, float* out_a, int count_b, const float* g_b, float* out_b, int D, int gstride,
	const adgs_func_eval& f, hipStream_t stream, const AdamSlot* adam_a, const AdamSlot* adam_b, float beta1, float beta2, float eps) {
	if (f.n_params <= 0) return 0;
	const AdamSlot off = { nullptr, nullptr, nullptr, 0.f, 0.f };
	const AdamSlot sa = adam_a ? *adam_a : off, sb = adam_b ? *adam_b : off;
	const bool on_a = (out_a || sa.p) && count_a > 0, on_b = (out_b || sb.p) && count_b > 0;
	if (!on_a && !on_b) return 0;
	if ((on_a && !g_a) || (on_b && !g_b)) { set_error("launch_lin_param_grad2: NULL gradient buffer"); return -1; }
	if ((on_a && out_a && sa.p) || (on_b && out_b && sb.p)) { set_error("launch_lin_param_grad2: a side takes the gradient store OR the Adam step"); return -1; }
	ParamGradArgs pg;
	pg.D = D; pg.gstride = gstride; pg.f = f; pg.beta1 = beta1; pg.beta2 = beta2; pg.eps = eps;
	const size_t per_block = (size_t)256 * PG_ITEMS;
	const size_t nb_a = on_a ? ((size_t)count_a * D * f.n_params + per_block - 1) / per_block : 0, nb_b = on_b ? ((size_t)count_b * D * f.n_params + per_block - 1) / per_block : 0;
	if (on_a) {
		pg.n0 = 0; pg.count = count_a; pg.g = g_a; pg.out = out_a; pg.adam = sa;
		pg.nb0 = on_b ? (int)nb_a : 0; pg.n0_b = 0; pg.count_b = on_b ? count_b : 0; pg.g_b = on_b ? g_b : nullptr; pg.out_b = on_b ? out_b : nullptr; pg.adam_b = on_b ? sb : off;
	} else {
		pg.n0 = 0; pg.count = count_b; pg.g = g_b; pg.out = out_b; pg.adam = sb;
		pg.nb0 = 0; pg.n0_b = 0; pg.count_b = 0; pg.g_b = nullptr; pg.out_b = nullptr; pg.adam_b = off;
	}
	if (sa.p || sb.p) hipLaunchKernelGGL(deform_lin_param_grad_kernel<true>, dim3((unsigned)(nb_a + nb_b)), dim3(256), f.n_params * sizeof(float), stream, pg);
	else hipLaunchKernelGGL(deform_lin_param_grad_kernel<false>, dim3((unsigned)(nb_a + nb_b)), dim3(256), f.n_params * sizeof(float), stream, pg);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
int launch_lin_param_grad(int count, int D, const float* g, int gstride, float* out, const adgs_func_eval& f, hipStream_t stream) {
	if (count <= 0 || f.n_params <= 0) return 0;
	if (!g || !out) { set_error("launch_lin_param_grad: NULL gradient buffer"); return -1; }
	return launch_lin_param_grad2(count, g, out, 0, nullptr, nullptr, D, gstride, f, stream, nullptr, nullptr, 0.f, 0.f, 0.f);
}
} // namespace adgs

using namespace adgs;

extern "C" int adgs_func_eval_forward(int N, int D, const float* param, const adgs_func_eval* f, float* out, void* stream_) {
	if (N <= 0) return 0;
	if (!param || !f || !out || (D != 3 && D != 4)) { set_error("adgs_func_eval_forward: bad arguments"); return -1; }
	if (check_func(f, "adgs_func_eval_forward") != 0) return -1;
	if (D == 3 && f->quat_start >= 0) { set_error("quaternion spline needs D == 4"); return -1; }
	hipStream_t stream = (hipStream_t)stream_;
	if (D == 3) hipLaunchKernelGGL((func_eval_fwd_kernel<3, 0>), dim3((N + 255) / 256), dim3(256), 0, stream, N, param, *f, out);
	else {
#define ADGS_CALL(NQ) hipLaunchKernelGGL((func_eval_fwd_kernel<4, NQ>), dim3((N + 255) / 256), dim3(256), 0, stream, N, param, *f, out)
		ADGS_NQ_SWITCH(*f, ADGS_CALL)
#undef ADGS_CALL
	}
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

extern "C" int adgs_func_eval_backward(int N, int D, const float* param, const adgs_func_eval* f, const float* dL_dout, float* dL_dparam, void* stream_) {
	if (N <= 0) return 0;
	if (!param || !f || !dL_dout || !dL_dparam || (D != 3 && D != 4)) { set_error("adgs_func_eval_backward: bad arguments"); return -1; }
	if (check_func(f, "adgs_func_eval_backward") != 0) return -1;
	hipStream_t stream = (hipStream_t)stream_;
	if (D == 3) hipLaunchKernelGGL((func_eval_bwd_kernel<3, 0>), dim3((N + 255) / 256), dim3(256), 0, stream, N, param, *f, dL_dout, dL_dparam);
	else {
#define ADGS_CALL(NQ) hipLaunchKernelGGL((func_eval_bwd_kernel<4, NQ>), dim3((N + 255) / 256), dim3(256), 0, stream, N, param, *f, dL_dout, dL_dparam)
		ADGS_NQ_SWITCH(*f, ADGS_CALL)
#undef ADGS_CALL
	}
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

extern "C" int adgs_deform_forward_flow(const adgs_deform_params* p, const adgs_func_eval* f_xyz, const adgs_func_eval* f_rotation,
	const adgs_func_eval* f_shs, const adgs_func_eval* f_background, const adgs_func_eval* f_xyz_flow, const adgs_func_eval* f_background_flow,
	const adgs_deform_outputs* out, float* flow_xyz, void* stream_) {
	if (!p || !out) { set_error("adgs_deform_forward: NULL params/outputs"); return -1; }
	const int N = p->Ns + p->No;
	if (N <= 0) return 0;
	if (check_func(f_xyz, "f_xyz") || check_func(f_rotation, "f_rotation") || check_func(f_shs, "f_shs") || check_func(f_background, "f_background") ||
	    check_func(f_xyz_flow, "f_xyz_flow") || check_func(f_background_flow, "f_background_flow")) return -1;
	DeformArgs a;
	a.p = *p; a.o = *out; a.flow_xyz = flow_xyz;
	a.fx = f_xyz ? *f_xyz : empty_func(); a.fr = f_rotation ? *f_rotation : empty_func();
	a.fb = f_background ? *f_background : empty_func();
	a.fx2 = f_xyz_flow ? *f_xyz_flow : empty_func(); a.fb2 = f_background_flow ? *f_background_flow : empty_func();
	if (flow_xyz && f_xyz && f_xyz_flow && f_xyz->n_params != f_xyz_flow->n_params) { set_error("f_xyz and f_xyz_flow describe different parameter tensors"); return -1; }
	if (flow_xyz && !f_xyz_flow) a.fx2.n_params = a.fx.n_params;
	const adgs_func_eval fs = f_shs ? *f_shs : empty_func();
	hipStream_t stream = (hipStream_t)stream_;
	StageTimer timer(ST_DEFORM_FWD, stream);
	float* shs_out = a.o.shs;
	a.o.shs = nullptr;                       // SH rows go through the flat coalesced kernel below
	if (a.o.xyz || a.flow_xyz || a.o.rotation || a.o.opacity || a.o.scales) {
		const int np_x = std::max(a.fx.n_params, a.fx2.n_params);
		a.fx.n_params = a.fx2.n_params = np_x;
		a.stride_x = (3 * np_x) | 1; a.stride_r = (4 * a.fr.n_params) | 1;
		{
			size_t lds = 0;
			// object xyz rows: direct 8-byte reads when the row length is even (rows are then 8-byte aligned), else staged through LDS
			a.xyz_rows = p->No > 0 && np_x > 0 && np_x % 2 == 0 && p->xyz_deform_param && (reinterpret_cast<uintptr_t>(p->xyz_deform_param) & 7) == 0 &&
				p->obj_xyz;
			const int B = p->No > 0 ? pick_block(a.xyz_rows ? a.stride_r : std::max(a.stride_x, a.stride_r), &lds) : 256;
			if (a.xyz_rows) lds = std::max(lds, (size_t)2 * np_x * sizeof(float));
			if (lds > MAX_STAGING_LDS) { set_error("adgs_deform_forward: deformation rows too large for the LDS staging buffer"); return -1; }
			a.scene4 = scene4_ok({ p->scene_xyz, p->scene_rotation, p->scene_opacity, p->scene_scaling, a.o.xyz, a.flow_xyz, a.o.rotation, a.o.opacity, a.o.scales });
			const int nb_o = (p->No + B - 1) / B, nb_scene = (p->use_time_mask & ADGS_DEFORM_SKIP_SCENE) ? 0 : (a.scene4 ? ((p->Ns + 3) / 4 + B - 1) / B : (p->Ns + B - 1) / B);
			const int nb_rot = (a.o.rotation || a.o.opacity || a.o.scales) ? nb_o : 0;
			const int nb_xyz = (a.o.xyz || a.flow_xyz) ? (a.xyz_rows ? (int)((3 * (size_t)p->No + B - 1) / B) : nb_o) : 0;
			a.n_begin = 0; a.n_end = N;
#define ADGS_CALL(NQ) do { if (nb_rot + nb_xyz + nb_scene == 0) break;      /* a model without objects on the raw-scene path: nothing to deform */ \
				if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&deform_fwd_kernel<NQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
				hipLaunchKernelGGL((deform_fwd_kernel<NQ>), dim3(nb_rot + nb_xyz + nb_scene), dim3(B), lds, stream, a, nb_rot, nb_xyz); } while (0)
			ADGS_NQ_SWITCH(a.fr, ADGS_CALL)
#undef ADGS_CALL
			ADGS_HIP_CHECK(hipGetLastError());
		}
	}
	if (shs_out) {
		ShsFwdArgs sa;
		sa.Ns = p->Ns; sa.N = N; sa.M = p->sh_coeffs;
		sa.scene_dc = p->scene_shs_dc; sa.obj_dc = p->obj_shs_dc; sa.scene_rest = p->scene_shs_rest; sa.obj_rest = p->obj_shs_rest;
		sa.sp_scene = p->shs_deform_param_scene; sa.sp_obj = p->shs_deform_param_obj; sa.fs = fs; sa.out = shs_out;
		const size_t quads = ((size_t)N * p->sh_coeffs * 3 + 3) / 4;
		if (p->sh_coeffs == 16 && (reinterpret_cast<uintptr_t>(shs_out) & 15) == 0 && (sa.scene_dc || sa.obj_dc)) {
			ShSource src;
			memset(&src, 0, sizeof(src));
			src.Ns = sa.Ns; src.scene_dc = sa.scene_dc ? sa.scene_dc : sa.obj_dc; src.obj_dc = sa.obj_dc ? sa.obj_dc : sa.scene_dc;
			src.scene_sp = sa.sp_scene; src.obj_sp = sa.sp_obj; src.f = fs;
			if (launch_sh0(N, src, shs_out, stream, nullptr, 48) != 0) return -1;
			hipLaunchKernelGGL(shs_rest_interleave_kernel, dim3((unsigned)(((size_t)N * 12 + 255) / 256)), dim3(256), 0, stream, sa.Ns, N, sa.scene_rest, sa.obj_rest, shs_out);
		} else {
			hipLaunchKernelGGL(deform_shs_fwd_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, stream, sa);
		}
		ADGS_HIP_CHECK(hipGetLastError());
	}
	return 0;
}
extern "C" int adgs_deform_forward(const adgs_deform_params* p, const adgs_func_eval* f_xyz, const adgs_func_eval* f_rotation,
	const adgs_func_eval* f_shs, const adgs_func_eval* f_background, const adgs_deform_outputs* out, void* stream_) {
	return adgs_deform_forward_flow(p, f_xyz, f_rotation, f_shs, f_background, nullptr, nullptr, out, nullptr, stream_);
}

extern "C" int adgs_deform_backward_flow(const adgs_deform_params* p, const adgs_func_eval* f_xyz, const adgs_func_eval* f_rotation,
	const adgs_func_eval* f_shs, const adgs_func_eval* f_background, const adgs_func_eval* f_xyz_flow, const adgs_func_eval* f_background_flow,
	const float* dL_dxyz, const float* dL_drotation, const float* dL_dshs, const float* dL_dopacity, const float* dL_dscales, const float* dL_dflow_xyz,
	const adgs_deform_grads* grads, void* stream_) {
	if (!p || !grads) { set_error("adgs_deform_backward: NULL params/grads"); return -1; }
	const int N = p->Ns + p->No;
	if (N <= 0) return 0;
	if (check_func(f_xyz, "f_xyz") || check_func(f_rotation, "f_rotation") || check_func(f_shs, "f_shs") || check_func(f_background, "f_background") ||
	    check_func(f_xyz_flow, "f_xyz_flow") || check_func(f_background_flow, "f_background_flow")) return -1;
	DeformBwdArgs a;
	a.p = *p; a.g = *grads;
	a.fx = f_xyz ? *f_xyz : empty_func(); a.fr = f_rotation ? *f_rotation : empty_func();
	a.fb = f_background ? *f_background : empty_func();
	a.fx2 = f_xyz_flow ? *f_xyz_flow : empty_func(); a.fb2 = f_background_flow ? *f_background_flow : empty_func();
	const adgs_func_eval fs = f_shs ? *f_shs : empty_func();
	hipStream_t stream = (hipStream_t)stream_;
	StageTimer timer(ST_DEFORM_BWD, stream);
	a.g_xyz = dL_dxyz; a.g_flow = dL_dflow_xyz; a.g_rot = dL_drotation; a.g_op = dL_dopacity; a.g_sc = dL_dscales;
	if (dL_dxyz || dL_dflow_xyz || dL_drotation || dL_dopacity || dL_dscales) {
		const int np_x = std::max(a.fx.n_params, a.fx2.n_params);
		a.fx.n_params = a.fx2.n_params = np_x;
		a.stride_x = (3 * np_x) | 1; a.stride_r = (4 * a.fr.n_params) | 1;
		{
			const bool want_rest = dL_dxyz || dL_dflow_xyz || dL_dopacity || dL_dscales;
			size_t lds = 0;
			int B = 256;
			if (p->No > 0) {
				B = pick_block(std::max(a.stride_x, a.stride_r), &lds);
				lds = std::max(lds, (size_t)B * a.stride_x * sizeof(float) + 2 * (size_t)np_x * sizeof(float));     // + dense basis rows of the two time stamps
			}
			if (lds > MAX_STAGING_LDS) { set_error("adgs_deform_backward: deformation rows too large for the LDS staging buffer"); return -1; }
			const int nb_o = (p->No + B - 1) / B;
			a.scene4 = scene4_ok({ p->scene_xyz, p->scene_rotation, p->scene_opacity, p->scene_scaling, dL_dxyz, dL_dflow_xyz, dL_drotation, dL_dopacity, dL_dscales,
				grads->scene_xyz, grads->scene_rotation, grads->scene_opacity, grads->scene_scaling });
			const int nb_rot = dL_drotation ? nb_o : 0, nb_xyz = want_rest ? nb_o : 0,
				nb_scene = (p->use_time_mask & ADGS_DEFORM_SKIP_SCENE) ? 0 : (a.scene4 ? ((p->Ns + 3) / 4 + B - 1) / B : (p->Ns + B - 1) / B);
			a.n_begin = 0; a.n_end = N;
#define ADGS_CALL(NQ) do { if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&deform_bwd_kernel<NQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
				if (nb_rot + nb_xyz + nb_scene != 0) hipLaunchKernelGGL((deform_bwd_kernel<NQ>), dim3(nb_rot + nb_xyz + nb_scene), dim3(B), lds, stream, a, nb_rot, nb_xyz); } while (0)
			ADGS_NQ_SWITCH(a.fr, ADGS_CALL)
#undef ADGS_CALL
			ADGS_HIP_CHECK(hipGetLastError());
		}
	}
	if (dL_dshs) {
		const int M = p->sh_coeffs;
		ShsBwdArgs sb;
		sb.Ns = p->Ns; sb.N = N; sb.M = M; sb.g = dL_dshs;
		sb.g_scene_dc = grads->scene_shs_dc; sb.g_obj_dc = grads->obj_shs_dc; sb.g_scene_rest = grads->scene_shs_rest; sb.g_obj_rest = grads->obj_shs_rest;
		const size_t tot = (size_t)N * M * 3;
		if (M == 16 && (reinterpret_cast<uintptr_t>(dL_dshs) & 15) == 0)
			hipLaunchKernelGGL(shs_grad_split_kernel, dim3((unsigned)(((size_t)N * 12 + 255) / 256)), dim3(256), 0, stream, sb.Ns, N, dL_dshs, sb.g_scene_dc, sb.g_obj_dc, sb.g_scene_rest, sb.g_obj_rest);
		else
			hipLaunchKernelGGL(deform_shs_bwd_copy_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, sb);
		ADGS_HIP_CHECK(hipGetLastError());
		for (int part = 0; part < 2; part++) {
			float* out = part == 0 ? grads->shs_deform_param_scene : grads->shs_deform_param_obj;
			const int count = part == 0 ? p->Ns : p->No;
			if (!out || count == 0 || fs.n_params == 0) continue;
			ParamGradArgs pg;
			pg.n0 = part == 0 ? 0 : p->Ns; pg.count = count; pg.D = 3; pg.gstride = M * 3; pg.g = dL_dshs; pg.out = out; pg.f = fs;
			pg.nb0 = 0; pg.n0_b = 0; pg.count_b = 0; pg.g_b = nullptr; pg.out_b = nullptr;
			const size_t t2 = (size_t)count * 3 * fs.n_params;
			hipLaunchKernelGGL(deform_lin_param_grad_kernel<false>, dim3((unsigned)((t2 + 256 * PG_ITEMS - 1) / (256 * PG_ITEMS))), dim3(256), fs.n_params * sizeof(float), stream, pg);
			ADGS_HIP_CHECK(hipGetLastError());
		}
	}
	return 0;
}
extern "C" int adgs_deform_backward(const adgs_deform_params* p, const adgs_func_eval* f_xyz, const adgs_func_eval* f_rotation,
	const adgs_func_eval* f_shs, const adgs_func_eval* f_background,
	const float* dL_dxyz, const float* dL_drotation, const float* dL_dshs, const float* dL_dopacity, const float* dL_dscales,
	const adgs_deform_grads* grads, void* stream_) {
	return adgs_deform_backward_flow(p, f_xyz, f_rotation, f_shs, f_background, nullptr, nullptr,
		dL_dxyz, dL_drotation, dL_dshs, dL_dopacity, dL_dscales, nullptr, grads, stream_);
}
