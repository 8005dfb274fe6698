// Front-to-back alpha compositing (forward) and its back-to-front replay
// (backward) for gfx950 / wave64.
//
// Reference semantics: RAST/cuda_rasterizer/forward.cu:261-402 and
// backward.cu:417-646, including the 1-T
// storage, the exclusive T<1e-4 stop and the "opacity-T quirk"
// (backward.cu:612-614).
//
// Layout: one workgroup = one 16x16 tile = 4 wave64; a wave owns 4 rows of 16
// pixels.  Per batch of 256 list entries every thread gathers ONE 64-byte Splat
// line (4 x 16-B loads) into LDS, so colour/depth/flow/semantic payloads are
// staged together with the conic (the reference re-reads payloads from global
// memory per pixel).  The backward reduces every per-Gaussian partial sum over
// the 64 lanes of a wave before touching memory: one atomic per (wave, Gaussian,
// component) instead of one per (pixel, Gaussian, component).
#include "common.h"
#include "kernels.h"

namespace adgs {
namespace {

constexpr float ALPHA_MAX = 0.99f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_STOP = 0.0001f;

template <bool MULTI_SEM>
__global__ void __launch_bounds__(TILE_PIX) render_fwd_kernel(RenderFwdArgs a) {
	__shared__ float4 s_splat[TILE_PIX * 4];
	__shared__ uint32_t s_id[TILE_PIX];
	const int tid = threadIdx.y * TILE_X + threadIdx.x;
	const uint32_t pixx = blockIdx.x * TILE_X + threadIdx.x, pixy = blockIdx.y * TILE_Y + threadIdx.y;
	const bool inside = pixx < (uint32_t)a.W && pixy < (uint32_t)a.H;
	const size_t pix_id = (size_t)a.W * pixy + pixx;
	const float pxf = (float)pixx, pyf = (float)pixy;
	const uint2 range = a.ranges[blockIdx.y * a.gx + blockIdx.x];
	const int rounds = (int)((range.y - range.x + TILE_PIX - 1) / TILE_PIX);
	int toDo = (int)(range.y - range.x);
	bool done = !inside;

	float T = 1.0f;
	uint32_t contributor = 0, last_contributor = 0;
	float C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, F0 = 0.f, F1 = 0.f, F2 = 0.f, S0 = 0.f;
	float S[MULTI_SEM ? MAX_SEMANTIC : 1];
	if (MULTI_SEM) {
#pragma unroll
		for (int k = 0; k < MAX_SEMANTIC; k++) S[k] = 0.f;
	}

	for (int i = 0; i < rounds; i++, toDo -= TILE_PIX) {
		const int num_done = __syncthreads_count(done);
		if (num_done == TILE_PIX) break;
		const uint32_t progress = (uint32_t)i * TILE_PIX + tid;
		if (range.x + progress < range.y) {
			const uint32_t id = a.point_list[range.x + progress];
			const float4* src = reinterpret_cast<const float4*>(a.splats + id);
			s_splat[tid * 4 + 0] = src[0];
			s_splat[tid * 4 + 1] = src[1];
			s_splat[tid * 4 + 2] = src[2];
			s_splat[tid * 4 + 3] = src[3];
			if (MULTI_SEM) s_id[tid] = id;
		}
		__syncthreads();
		const int lim = min(TILE_PIX, toDo);
		for (int j = 0; !done && j < lim; j++) {
			contributor++;
			const float4 q0 = s_splat[j * 4 + 0];      // x y ca cb
			const float4 q1 = s_splat[j * 4 + 1];      // cc op r g
			const float dx = q0.x - pxf, dy = q0.y - pyf;
			const float power = -0.5f * (q0.z * dx * dx + q1.x * dy * dy) - q0.w * dx * dy;
			if (power > 0.0f) continue;
			const float alpha = fminf(ALPHA_MAX, q1.y * expf(power));
			if (alpha < ALPHA_MIN) continue;
			const float test_T = T * (1 - alpha);
			if (test_T < T_STOP) { done = true; continue; }
			const float4 q2 = s_splat[j * 4 + 2];      // b dval fx fy
			const float4 q3 = s_splat[j * 4 + 3];      // fz sem0 zview pad
			const float w = alpha * T;
			C0 += q1.z * w; C1 += q1.w * w; C2 += q2.x * w;
			F0 += q2.z * w; F1 += q2.w * w; F2 += q3.x * w;
			Dp += q2.y * w;
			if (MULTI_SEM) {
				const float* sem = a.semantic + (size_t)s_id[j] * a.D_S;
				for (int ch = 0; ch < a.D_S; ch++) S[ch] += sem[ch] * w;
			} else {
				S0 += q3.y * w;
			}
			T = test_T;
			last_contributor = contributor;
		}
	}
	if (inside) {
		const size_t HW = (size_t)a.H * a.W;
		a.final_T[pix_id] = (float)(1.0 - (double)T);
		a.n_contrib[pix_id] = last_contributor;
		if (a.has_color) {
			a.out_color[0 * HW + pix_id] = C0 + T * a.bg[0];
			a.out_color[1 * HW + pix_id] = C1 + T * a.bg[1];
			a.out_color[2 * HW + pix_id] = C2 + T * a.bg[2];
		}
		if (a.has_flow) {
			a.out_flow[0 * HW + pix_id] = F0; a.out_flow[1 * HW + pix_id] = F1; a.out_flow[2 * HW + pix_id] = F2;
		}
		if (a.has_sem) {
			if (MULTI_SEM) { for (int ch = 0; ch < a.D_S; ch++) a.out_semantic[ch * HW + pix_id] = S[ch]; }
			else a.out_semantic[pix_id] = S0;
		}
		a.out_depth[pix_id] = Dp;
	}
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
	for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
	return v;
}

template <bool MULTI_SEM>
__global__ void __launch_bounds__(TILE_PIX) render_bwd_kernel(RenderBwdArgs a) {
	__shared__ float4 s_splat[TILE_PIX * 4];
	__shared__ uint32_t s_id[TILE_PIX];
	const int tid = threadIdx.y * TILE_X + threadIdx.x;
	const int lane = tid & (WAVE - 1);
	const uint32_t pixx = blockIdx.x * TILE_X + threadIdx.x, pixy = blockIdx.y * TILE_Y + threadIdx.y;
	const bool inside = pixx < (uint32_t)a.W && pixy < (uint32_t)a.H;
	const size_t pix_id = (size_t)a.W * pixy + pixx;
	const size_t HW = (size_t)a.H * a.W;
	const float pxf = (float)pixx, pyf = (float)pixy;
	const uint2 range = a.ranges[blockIdx.y * a.gx + blockIdx.x];
	const int rounds = (int)((range.y - range.x + TILE_PIX - 1) / TILE_PIX);
	int toDo = (int)(range.y - range.x);

	const float T_final = inside ? (float)(1.0 - (double)a.final_T[pix_id]) : 0.f;
	float T = T_final;
	uint32_t contributor = (uint32_t)toDo;
	const int last_contributor = inside ? (int)a.n_contrib[pix_id] : 0;

	float acc_c0 = 0.f, acc_c1 = 0.f, acc_c2 = 0.f, acc_f0 = 0.f, acc_f1 = 0.f, acc_f2 = 0.f, acc_d = 0.f, acc_s0 = 0.f;
	float gC0 = 0.f, gC1 = 0.f, gC2 = 0.f, gF0 = 0.f, gF1 = 0.f, gF2 = 0.f, gD = 0.f, gO = 0.f, gS0 = 0.f;
	float acc_s[MULTI_SEM ? MAX_SEMANTIC : 1], gS[MULTI_SEM ? MAX_SEMANTIC : 1], last_s[MULTI_SEM ? MAX_SEMANTIC : 1];
	if (MULTI_SEM) {
#pragma unroll
		for (int k = 0; k < MAX_SEMANTIC; k++) { acc_s[k] = 0.f; gS[k] = 0.f; last_s[k] = 0.f; }
	}
	if (inside) {
		if (a.do_color) { gC0 = a.dL_dpix[0 * HW + pix_id]; gC1 = a.dL_dpix[1 * HW + pix_id]; gC2 = a.dL_dpix[2 * HW + pix_id]; }
		if (a.do_flow) { gF0 = a.dL_dpix_flow[0 * HW + pix_id]; gF1 = a.dL_dpix_flow[1 * HW + pix_id]; gF2 = a.dL_dpix_flow[2 * HW + pix_id]; }
		if (a.do_sem) {
			if (MULTI_SEM) { for (int ch = 0; ch < a.D_S; ch++) gS[ch] = a.dL_dpix_sem[ch * HW + pix_id]; }
			else gS0 = a.dL_dpix_sem[pix_id];
		}
		if (a.do_depth) gD = a.dL_dpix_depth[pix_id];
		if (a.do_opacity) gO = a.dL_dpix_opacity[pix_id];
	}
	float last_alpha = 0.f, last_c0 = 0.f, last_c1 = 0.f, last_c2 = 0.f, last_d = 0.f, last_f0 = 0.f, last_f1 = 0.f, last_f2 = 0.f, last_s0 = 0.f;
	const float ddelx_dx = (float)(0.5 * a.W), ddely_dy = (float)(0.5 * a.H);
	// dL/dpixel . bg is loop invariant (backward.cu:620-623)
	float bg_dot_dpixel = 0.f;
	bg_dot_dpixel += a.bg[0] * gC0; bg_dot_dpixel += a.bg[1] * gC1; bg_dot_dpixel += a.bg[2] * gC2;

	for (int i = 0; i < rounds; i++, toDo -= TILE_PIX) {
		__syncthreads();
		const uint32_t progress = (uint32_t)i * TILE_PIX + tid;
		if (range.x + progress < range.y) {
			const uint32_t id = a.point_list[range.y - progress - 1];
			const float4* src = reinterpret_cast<const float4*>(a.splats + id);
			s_splat[tid * 4 + 0] = src[0];
			s_splat[tid * 4 + 1] = src[1];
			s_splat[tid * 4 + 2] = src[2];
			s_splat[tid * 4 + 3] = src[3];
			s_id[tid] = id;
		}
		__syncthreads();
		const int lim = min(TILE_PIX, toDo);
		for (int j = 0; j < lim; j++) {
			// wave-uniform loop: every lane walks the batch so the wave reductions stay convergent
			contributor--;
			bool active = inside && (int)contributor < last_contributor;
			const float4 q0 = s_splat[j * 4 + 0];
			const float4 q1 = s_splat[j * 4 + 1];
			const float dx = q0.x - pxf, dy = q0.y - pyf;
			const float power = -0.5f * (q0.z * dx * dx + q1.x * dy * dy) - q0.w * dx * dy;
			active = active && !(power > 0.0f);
			const float G = expf(power);
			const float alpha = fminf(ALPHA_MAX, q1.y * G);
			active = active && !(alpha < ALPHA_MIN);
			if (!__any(active)) continue;
			const float4 q2 = s_splat[j * 4 + 2];
			const float4 q3 = s_splat[j * 4 + 3];
			const uint32_t gid = s_id[j];

			float v_c0 = 0.f, v_c1 = 0.f, v_c2 = 0.f, v_f0 = 0.f, v_f1 = 0.f, v_f2 = 0.f, v_s0 = 0.f, v_d = 0.f;
			float v_mx = 0.f, v_my = 0.f, v_ca = 0.f, v_cb = 0.f, v_cc = 0.f, v_op = 0.f;
			float w_act = 0.f;
			if (active) {
				T = T / (1.f - alpha);
				const float dchannel_dcolor = alpha * T;
				w_act = dchannel_dcolor;
				float dL_dalpha = 0.0f;
				if (a.do_color) {
					acc_c0 = last_alpha * last_c0 + (1.f - last_alpha) * acc_c0; last_c0 = q1.z;
					dL_dalpha += (q1.z - acc_c0) * gC0; v_c0 = dchannel_dcolor * gC0;
					acc_c1 = last_alpha * last_c1 + (1.f - last_alpha) * acc_c1; last_c1 = q1.w;
					dL_dalpha += (q1.w - acc_c1) * gC1; v_c1 = dchannel_dcolor * gC1;
					acc_c2 = last_alpha * last_c2 + (1.f - last_alpha) * acc_c2; last_c2 = q2.x;
					dL_dalpha += (q2.x - acc_c2) * gC2; v_c2 = dchannel_dcolor * gC2;
				}
				if (a.do_flow) {
					acc_f0 = last_alpha * last_f0 + (1.f - last_alpha) * acc_f0; last_f0 = q2.z;
					dL_dalpha += (q2.z - acc_f0) * gF0; v_f0 = dchannel_dcolor * gF0;
					acc_f1 = last_alpha * last_f1 + (1.f - last_alpha) * acc_f1; last_f1 = q2.w;
					dL_dalpha += (q2.w - acc_f1) * gF1; v_f1 = dchannel_dcolor * gF1;
					acc_f2 = last_alpha * last_f2 + (1.f - last_alpha) * acc_f2; last_f2 = q3.x;
					dL_dalpha += (q3.x - acc_f2) * gF2; v_f2 = dchannel_dcolor * gF2;
				}
				if (a.do_sem) {
					if (MULTI_SEM) {
						const float* sem = a.semantic + (size_t)gid * a.D_S;
						for (int ch = 0; ch < a.D_S; ch++) {
							const float s = sem[ch];
							acc_s[ch] = last_alpha * last_s[ch] + (1.f - last_alpha) * acc_s[ch]; last_s[ch] = s;
							dL_dalpha += (s - acc_s[ch]) * gS[ch];
						}
					} else {
						acc_s0 = last_alpha * last_s0 + (1.f - last_alpha) * acc_s0; last_s0 = q3.y;
						dL_dalpha += (q3.y - acc_s0) * gS0; v_s0 = dchannel_dcolor * gS0;
					}
				}
				if (a.do_depth) {
					acc_d = last_alpha * last_d + (1.f - last_alpha) * acc_d; last_d = q2.y;
					dL_dalpha += (q2.y - acc_d) * gD; v_d = dchannel_dcolor * gD;
				}
				if (a.do_opacity) dL_dalpha += gO * T_final / (1.f - alpha);   // before the *= T: reference quirk
				dL_dalpha *= T;
				last_alpha = alpha;
				dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
				const float dL_dG = q1.y * dL_dalpha;
				const float gdx = G * dx, gdy = G * dy;
				const float dG_ddelx = -gdx * q0.z - gdy * q0.w;
				const float dG_ddely = -gdy * q1.x - gdx * q0.w;
				v_mx = dL_dG * dG_ddelx * ddelx_dx;
				v_my = dL_dG * dG_ddely * ddely_dy;
				v_ca = -0.5f * gdx * dx * dL_dG;
				v_cb = -0.5f * gdx * dy * dL_dG;
				v_cc = -0.5f * gdy * dy * dL_dG;
				v_op = G * dL_dalpha;
			}
			// wave64 reductions, then one atomic per component from lane 0
			v_mx = wave_sum(v_mx); v_my = wave_sum(v_my);
			v_ca = wave_sum(v_ca); v_cb = wave_sum(v_cb); v_cc = wave_sum(v_cc); v_op = wave_sum(v_op);
			if (a.do_color) { v_c0 = wave_sum(v_c0); v_c1 = wave_sum(v_c1); v_c2 = wave_sum(v_c2); }
			if (a.do_flow) { v_f0 = wave_sum(v_f0); v_f1 = wave_sum(v_f1); v_f2 = wave_sum(v_f2); }
			if (a.do_depth) v_d = wave_sum(v_d);
			if (a.do_sem && !MULTI_SEM) v_s0 = wave_sum(v_s0);
			if (lane == 0) {
				atomicAdd(&a.dL_dmean2D[3 * (size_t)gid + 0], v_mx);
				atomicAdd(&a.dL_dmean2D[3 * (size_t)gid + 1], v_my);
				atomicAdd(&a.dL_dconic[4 * (size_t)gid + 0], v_ca);
				atomicAdd(&a.dL_dconic[4 * (size_t)gid + 1], v_cb);
				atomicAdd(&a.dL_dconic[4 * (size_t)gid + 3], v_cc);
				atomicAdd(&a.dL_dopacity[gid], v_op);
				if (a.do_color) {
					atomicAdd(&a.dL_dcolor[3 * (size_t)gid + 0], v_c0);
					atomicAdd(&a.dL_dcolor[3 * (size_t)gid + 1], v_c1);
					atomicAdd(&a.dL_dcolor[3 * (size_t)gid + 2], v_c2);
				}
				if (a.do_flow) {
					atomicAdd(&a.dL_dflow[3 * (size_t)gid + 0], v_f0);
					atomicAdd(&a.dL_dflow[3 * (size_t)gid + 1], v_f1);
					atomicAdd(&a.dL_dflow[3 * (size_t)gid + 2], v_f2);
				}
				if (a.do_depth) atomicAdd(&a.dL_ddepth[gid], v_d);
				if (a.do_sem && !MULTI_SEM) atomicAdd(&a.dL_dsem[gid], v_s0);
			}
			if (MULTI_SEM && a.do_sem) {
				for (int ch = 0; ch < a.D_S; ch++) {
					float v = wave_sum(w_act * gS[ch]);
					if (lane == 0) atomicAdd(&a.dL_dsem[(size_t)gid * a.D_S + ch], v);
				}
			}
		}
	}
}

} // namespace

int launch_render_fwd(const RenderFwdArgs& a, hipStream_t stream) {
	dim3 grid(a.gx, a.gy), block(TILE_X, TILE_Y);
	if (a.has_sem && a.D_S > 1) hipLaunchKernelGGL(render_fwd_kernel<true>, grid, block, 0, stream, a);
	else hipLaunchKernelGGL(render_fwd_kernel<false>, grid, block, 0, stream, a);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

int launch_render_bwd(const RenderBwdArgs& a, hipStream_t stream) {
	dim3 grid(a.gx, a.gy), block(TILE_X, TILE_Y);
	if (a.do_sem && a.D_S > 1) hipLaunchKernelGGL(render_bwd_kernel<true>, grid, block, 0, stream, a);
	else hipLaunchKernelGGL(render_bwd_kernel<false>, grid, block, 0, stream, a);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

} // namespace adgs
