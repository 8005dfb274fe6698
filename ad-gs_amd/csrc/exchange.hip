// Factored SH-gradient exchange (include/adgs_exchange.h): every rank expands and sums the per-camera colour-gradient
// factors it gathered into the SH parameter gradients of all cameras.
//
// Roofline: HBM streaming -- per Gaussian 12 B read per camera (+ 12 B of position) and 12*M + 12*C B written
// (336 B at M = 16, C = 12); the arithmetic is ~60 flops per (Gaussian, camera).
#include "common.h"
#include "kernels.h"
#include "geom.h"
#include "func_eval.h"
#include "../../include/adgs_exchange.h"

namespace adgs {
namespace {

constexpr int EX_THREADS = 256;
constexpr int EX_REST = 45;          // (16 - 1) coefficient rows x 3 channels

struct ExpandArgs {
	int n, P, Ns, row0, D, M, C;
	const float* xyz_head; const float* W;
	ShGradDst out;
	const float* rgb[ADGS_EXPAND_MAX_CAMS];
	const float* xyz_tail[ADGS_EXPAND_MAX_CAMS];
	float campos[ADGS_EXPAND_MAX_CAMS][3];
};

// dc and rest rows: one thread per Gaussian accumulates over the cameras in registers; the rest rows leave through LDS
// (coalesced 16-byte stores of the block's contiguous slab instead of 64 scattered 180-byte rows per wave).
__global__ void __launch_bounds__(EX_THREADS) sh_expand_rows_kernel(ExpandArgs a) {
	extern __shared__ float s_rows[];
	const int tid = threadIdx.x;
	const int base = blockIdx.x * EX_THREADS;
	const int idx = base + tid;
	const int nvalid = min(EX_THREADS, a.P - base);
	const int L = (a.M - 1) * 3;
	float dc[3] = { 0.f, 0.f, 0.f };
	float acc[EX_REST];
#pragma unroll
	for (int i = 0; i < EX_REST; i++) acc[i] = 0.f;
	if (idx < a.P) {
		float hx = 0.f, hy = 0.f, hz = 0.f;
		const bool head = idx < a.row0;
		if (head) { hx = a.xyz_head[3 * (size_t)idx]; hy = a.xyz_head[3 * (size_t)idx + 1]; hz = a.xyz_head[3 * (size_t)idx + 2]; }
		for (int c = 0; c < a.n; c++) {
			const float* rp = a.rgb[c] + 3 * (size_t)idx;
			const float g0 = rp[0], g1 = rp[1], g2 = rp[2];
			if (g0 == 0.f && g1 == 0.f && g2 == 0.f) continue;      // not visible from this camera (or fully clamped): no contribution
			float mx = hx, my = hy, mz = hz;
			if (!head) { const float* tp = a.xyz_tail[c] + 3 * (size_t)(idx - a.row0); mx = tp[0]; my = tp[1]; mz = tp[2]; }
			const float ox = mx - a.campos[c][0], oy = my - a.campos[c][1], oz = mz - a.campos[c][2];
			const float len = sqrtf(ox * ox + oy * oy + oz * oz);
			float coef[16];
			sh_coef_factors(a.D, ox / len, oy / len, oz / len, coef);
			dc[0] += coef[0] * g0; dc[1] += coef[0] * g1; dc[2] += coef[0] * g2;
#pragma unroll
			for (int k = 1; k < 16; k++) { acc[(k - 1) * 3] += coef[k] * g0; acc[(k - 1) * 3 + 1] += coef[k] * g1; acc[(k - 1) * 3 + 2] += coef[k] * g2; }
		}
		const bool is_obj = idx >= a.Ns;
		const size_t m = is_obj ? idx - a.Ns : idx;
		float* gdc = is_obj ? a.out.obj_dc : a.out.scene_dc;
		if (gdc) { gdc[3 * m] = dc[0]; gdc[3 * m + 1] = dc[1]; gdc[3 * m + 2] = dc[2]; }
	}
	if (L > 0 && (a.out.scene_rest || a.out.obj_rest)) {
		const int stride = L | 1;
#pragma unroll
		for (int i = 0; i < EX_REST; i++) if (i < L) s_rows[tid * stride + i] = acc[i];
		__syncthreads();
		stage_rows<false>(s_rows, stride, L, base, nvalid, a.Ns, a.out.scene_rest, a.out.obj_rest, tid, EX_THREADS);
	}
}

// deform rows: flat over the output elements of one side (scene or object), PG consecutive floats per thread:
// out[m, ch, j] = sum_c W[c][j] * SH_C0 * rgb_c[n0 + m, ch]
constexpr int EX_ITEMS = 8;
struct ExpandDeformArgs {
	int n, C, count, n0;
	float scale;                 // SH_C0 for the SH deformation rows (the factor is dL/dRGB), 1 for plain linear families
	const float* W; float* out;
	const float* rgb[ADGS_EXPAND_MAX_CAMS];
};
__global__ void __launch_bounds__(EX_THREADS) sh_expand_deform_kernel(ExpandDeformArgs a) {
	extern __shared__ float s_w[];
	const int np = a.C;
	for (int i = threadIdx.x; i < a.n * np; i += blockDim.x) s_w[i] = a.W[i];
	__syncthreads();
	const size_t tot = (size_t)a.count * 3 * np;
	const size_t e0 = ((size_t)blockIdx.x * EX_THREADS + threadIdx.x) * EX_ITEMS;
	if (e0 >= tot) return;
	const float C0 = a.scale;
	const size_t row0 = e0 / np;                     // (Gaussian, channel) row of the first output
	const int k0 = (int)(e0 - row0 * np);
	float v[EX_ITEMS];
#pragma unroll
	for (int it = 0; it < EX_ITEMS; it++) v[it] = 0.f;
	for (int c = 0; c < a.n; c++) {
		const float* g = a.rgb[c] + (size_t)a.n0 * 3;   // rows are (m, ch) pairs: rgb is [.,3] contiguous, so row r is element r
		const float* w = s_w + c * np;
		size_t row = row0; int k = k0;
		float gv = C0 * g[row];
#pragma unroll
		for (int it = 0; it < EX_ITEMS; it++) {
			v[it] += w[k] * gv;
			if (++k == np) {
				k = 0; row++;
				if (e0 + it + 1 < tot) gv = C0 * g[row];
			}
		}
	}
	if (e0 + EX_ITEMS <= tot) {
		float4* o = reinterpret_cast<float4*>(a.out + e0);
		o[0] = make_float4(v[0], v[1], v[2], v[3]); o[1] = make_float4(v[4], v[5], v[6], v[7]);
	} else {
		for (int it = 0; it < EX_ITEMS && e0 + it < tot; it++) a.out[e0 + it] = v[it];
	}
}

} // namespace
} // namespace adgs

using namespace adgs;

extern "C" int adgs_sh_grad_expand(int n_cams, const adgs_sh_expand_cam* cams, const float* W, int C,
	int P, int Ns, int row0, const float* xyz_head, int D, int M, const adgs_sh_grads* out, void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	StageTimer timer(ST_EXPAND, stream);
	if (P <= 0) return 0;
	if (!cams || !out || n_cams < 1 || n_cams > ADGS_EXPAND_MAX_CAMS) { set_error("adgs_sh_grad_expand: need 1.." + std::to_string(ADGS_EXPAND_MAX_CAMS) + " cameras"); return -1; }
	if (Ns < 0 || Ns > P || row0 < 0 || row0 > P || D < 0 || D > 3 || M < 1 || M > 16 || (D + 1) * (D + 1) > M || C < 0) {
		set_error("adgs_sh_grad_expand: inconsistent sizes"); return -1;
	}
	if (row0 > 0 && !xyz_head) { set_error("adgs_sh_grad_expand: xyz_head is NULL"); return -1; }
	ExpandArgs a;
	a.n = n_cams; a.P = P; a.Ns = Ns; a.row0 = row0; a.D = D; a.M = M; a.C = C; a.xyz_head = xyz_head; a.W = W;
	a.out.scene_dc = out->scene_dc; a.out.obj_dc = out->obj_dc; a.out.scene_rest = out->scene_rest; a.out.obj_rest = out->obj_rest;
	a.out.scene_sp = out->scene_deform; a.out.obj_sp = out->obj_deform; a.out.rgb_factor = nullptr;
	for (int c = 0; c < n_cams; c++) {
		if (!cams[c].rgb || (row0 < P && !cams[c].xyz_tail)) { set_error("adgs_sh_grad_expand: camera " + std::to_string(c) + " has a NULL pointer"); return -1; }
		a.rgb[c] = cams[c].rgb; a.xyz_tail[c] = cams[c].xyz_tail;
		a.campos[c][0] = cams[c].campos[0]; a.campos[c][1] = cams[c].campos[1]; a.campos[c][2] = cams[c].campos[2];
	}
	const bool want_rows = a.out.scene_dc || a.out.obj_dc || a.out.scene_rest || a.out.obj_rest;
	if (want_rows) {
		const unsigned grid = (unsigned)((P + EX_THREADS - 1) / EX_THREADS);
		const size_t lds = (size_t)EX_THREADS * (size_t)(((M - 1) * 3) | 1) * sizeof(float);
		hipLaunchKernelGGL(sh_expand_rows_kernel, dim3(grid), dim3(EX_THREADS), lds, stream, a);
		ADGS_HIP_CHECK(hipGetLastError());
	}
	if (C > 0 && (a.out.scene_sp || a.out.obj_sp)) {
		if (!W) { set_error("adgs_sh_grad_expand: W is NULL"); return -1; }
		if ((size_t)n_cams * C * sizeof(float) > 48 * 1024) { set_error("adgs_sh_grad_expand: n_cams * C too large"); return -1; }
		ExpandDeformArgs d;
		d.n = n_cams; d.C = C; d.W = W; d.scale = 0.28209479177387814f;
		for (int c = 0; c < n_cams; c++) d.rgb[c] = cams[c].rgb;
		for (int side = 0; side < 2; side++) {
			d.out = side == 0 ? a.out.scene_sp : a.out.obj_sp;
			d.count = side == 0 ? Ns : P - Ns; d.n0 = side == 0 ? 0 : Ns;
			if (!d.out || d.count <= 0) continue;
			const size_t tot = (size_t)d.count * 3 * C;
			const size_t per_block = (size_t)EX_THREADS * EX_ITEMS;
			hipLaunchKernelGGL(sh_expand_deform_kernel, dim3((unsigned)((tot + per_block - 1) / per_block)), dim3(EX_THREADS),
				(size_t)n_cams * C * sizeof(float), stream, d);
			ADGS_HIP_CHECK(hipGetLastError());
		}
	}
	return 0;
}

// out[m, d, j] = scale * sum_e W[e][j] * g_e[m, d]: the gradient of a parameter tensor [count, 3, C] that enters linearly
// (utils/func_utils.py:121-156) summed over n_terms (camera, time stamp) factors -- xyz_deform_param in the factored exchange
extern "C" int adgs_lin_grad_expand(int n_terms, const float* const* g, const float* W, int C, int count, float scale, float* out, void* stream_) {
	if (count <= 0 || C <= 0) return 0;
	if (!g || !W || !out || n_terms < 1 || n_terms > ADGS_EXPAND_MAX_CAMS) { set_error("adgs_lin_grad_expand: need 1.." + std::to_string(ADGS_EXPAND_MAX_CAMS) + " factors"); return -1; }
	if ((size_t)n_terms * C * sizeof(float) > 48 * 1024) { set_error("adgs_lin_grad_expand: n_terms * C too large"); return -1; }
	ExpandDeformArgs d;
	d.n = n_terms; d.C = C; d.count = count; d.n0 = 0; d.scale = scale; d.W = W; d.out = out;
	for (int e = 0; e < n_terms; e++) { if (!g[e]) { set_error("adgs_lin_grad_expand: NULL factor"); return -1; } d.rgb[e] = g[e]; }
	const size_t tot = (size_t)count * 3 * C, per_block = (size_t)EX_THREADS * EX_ITEMS;
	StageTimer timer(ST_EXPAND, (hipStream_t)stream_);
	hipLaunchKernelGGL(sh_expand_deform_kernel, dim3((unsigned)((tot + per_block - 1) / per_block)), dim3(EX_THREADS), (size_t)n_terms * C * sizeof(float),
		(hipStream_t)stream_, d);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
