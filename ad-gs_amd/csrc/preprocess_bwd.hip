// Per-Gaussian backward of the preprocess stage, fused into ONE pass over P
// (the reference runs two kernels: computeCov2DCUDA then preprocessCUDA,
// RAST/cuda_rasterizer/backward.cu:144-274, 346-414, driver :648-716).
//
// conic grads -> cov2D -> cov3D -> (scale, rotation) and -> mean;
// mean2D grads -> mean through the projection; depth grads -> mean;
// colour grads -> SH coefficients and (view direction) -> mean.
// The reference's non-derivative details are kept: 1/(det^2+1e-7), clamp masks on
// t.x/t.y only, un-normalised quaternion (no normalisation Jacobian).
//
// Roofline: HBM streaming; the 12*M-byte dL_dsh row per visible Gaussian
// dominates the writes.
#include "common.h"
#include "kernels.h"
#include "geom.h"
#include "func_eval.h"
#include <cstdlib>

#ifndef ADGS_PREB_STAGE_U4
#define ADGS_PREB_STAGE_U4 12
#endif
namespace adgs {
namespace {

struct M3 { float v[3][3]; };
__device__ __forceinline__ M3 m3mul(const M3& a, const M3& b) {
	M3 r;
#pragma unroll
	for (int c = 0; c < 3; c++)
#pragma unroll
		for (int rr = 0; rr < 3; rr++)
			r.v[c][rr] = a.v[0][rr] * b.v[c][0] + a.v[1][rr] * b.v[c][1] + a.v[2][rr] * b.v[c][2];
	return r;
}
__device__ __forceinline__ M3 m3t(const M3& a) {
	M3 r;
#pragma unroll
	for (int c = 0; c < 3; c++)
#pragma unroll
		for (int rr = 0; rr < 3; rr++) r.v[c][rr] = a.v[rr][c];
	return r;
}

// STAGED (M == 16 only): the block's SH rows go through LDS -- coalesced loads in, per-thread
// compute on the own row (odd row stride: conflict-free), coalesced stores of the gradient rows out --
// instead of 64 scattered 192-byte rows per wave access.
constexpr int BW_THREADS = 256;
constexpr int SH_ROW_FULL = 48, SH_ROW_FULL_LDS = 49, SH_ROW_REST = 45;

// Per-Gaussian inputs of the staged v2 kernel, requested with the SH rows (preprocess.hip: PreIn -- the same reasoning: radii -> (visible?)
// -> position / scale / rotation / opacity / accumulator line / Splat line -> clamp bits are dependent round trips behind the staging
// barrier otherwise).  Loaded unconditionally at a clamped index; the staged v2 paths that recompute cov3D (gacc != nullptr, no cov3D) use them.
struct PreBIn { alignas(16) float q[4]; float4 u0, u1, u2, u3, s0, s1; float p[3], s[3], op; int radius; uint8_t clamped; float dd[9]; };
__device__ __forceinline__ PreBIn load_preb_in(const PreprocessBwdArgs& a, const int idx) {
	PreBIn in;
	const size_t i = (size_t)min(idx, a.P - 1);
	const bool rs = a.sh_src.scene_xyz != nullptr && (int)i < a.sh_src.Ns;
	in.radius = a.radii[i];
	const float* pos = rs ? a.sh_src.scene_xyz : a.means3D;
	in.p[0] = pos[3 * i]; in.p[1] = pos[3 * i + 1]; in.p[2] = pos[3 * i + 2];
	const float* sc = rs ? a.sh_src.scene_scaling : a.scales;
	const float* rt = rs ? a.sh_src.scene_rotation : a.rotations;
	in.s[0] = sc[3 * i]; in.s[1] = sc[3 * i + 1]; in.s[2] = sc[3 * i + 2];
	const float4 r = *reinterpret_cast<const float4*>(rt + 4 * i);
	in.q[0] = r.x; in.q[1] = r.y; in.q[2] = r.z; in.q[3] = r.w;
	in.op = rs ? a.sh_src.scene_opacity[i] : 0.f;
	const float4* ga = reinterpret_cast<const float4*>(a.gacc + i * GACC_STRIDE);
	in.u0 = ld_stream4(ga); in.u1 = ld_stream4(ga + 1); in.u2 = ld_stream4(ga + 2); in.u3 = ld_stream4(ga + 3);
	const float4* sp = reinterpret_cast<const float4*>(a.splats + i);
	in.s0 = ld_stream4(sp); in.s1 = ld_stream4(sp + 1);
	in.clamped = a.clamped[i];
	if (a.ddir) {
#pragma unroll
		for (int k = 0; k < 9; k++) in.dd[k] = ld_stream(a.ddir + (size_t)k * a.P + i);      // (garbage for a culled Gaussian: never used)
	}
	return in;
}

template <bool STAGED>
__global__ void __launch_bounds__(BW_THREADS) preprocess_bwd_kernel(PreprocessBwdArgs a) {
	extern __shared__ float s_sh[];
	const int tid = threadIdx.x;
	const int base = blockIdx.x * BW_THREADS;
	const int idx = base + tid;
	const bool raw = a.sh_src.scene_dc != nullptr;
	const int nvalid = min(BW_THREADS, a.P - base);
	// prefetched inputs: the staged v2 path that recomputes cov3D from scales / rotations (what the raw-SH frames run)
	const bool pf = STAGED && a.gacc != nullptr && a.cov3D == nullptr && a.scales != nullptr && a.rotations != nullptr;      // kernel-uniform
	// the Adam step in place of the store of the `rest` gradient rows (adgs_sh_adam): the rows are assembled in LDS either way
	const bool adam_rest = STAGED && raw && (a.sh_dst.adam.scene_rest.p != nullptr || a.sh_dst.adam.obj_rest.p != nullptr);      // kernel-uniform
	PreBIn in;
	// the forward left d colour / d direction behind (PreprocessArgs.ddir): no second pass over the `rest` rows -- the LDS rows only carry the gradients out
	const bool have_ddir = STAGED && raw && pf && a.ddir != nullptr && a.D == 3;      // kernel-uniform
	if (STAGED) {
		if (raw && have_ddir) {
			in = load_preb_in(a, idx);
		} else if (raw) {
			stage_rows<true, ADGS_PREB_STAGE_U4>(s_sh, SH_ROW_REST, SH_ROW_REST, base, nvalid, a.sh_src.Ns, a.sh_src.scene_rest, a.sh_src.obj_rest, tid, BW_THREADS,
				[&]() { if (pf) in = load_preb_in(a, idx); });
		} else {
			// materialised [P,16,3] SH tensor: all 12 quads of a thread requested before the first LDS store (preprocess.hip)
			const float4* src = reinterpret_cast<const float4*>(a.shs + (size_t)base * SH_ROW_FULL);
			constexpr int NQ4 = SH_ROW_FULL / 4;
			const int total4 = nvalid * NQ4;
			float4 v[NQ4];
#pragma unroll
			for (int u = 0; u < NQ4; u++) v[u] = ld_stream4(src + min(tid + u * BW_THREADS, total4 - 1));
			if (pf) in = load_preb_in(a, idx);
#pragma unroll
			for (int u = 0; u < NQ4; u++) {
				const int q = tid + u * BW_THREADS;
				if (q < total4) {
					const int g = q / NQ4, c = (q - g * NQ4) * 4;
					float* d = s_sh + g * SH_ROW_FULL_LDS + c;
					d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
				}
			}
		}
		if (!have_ddir) __syncthreads();      // (with ddir nothing was staged: every thread only ever touches its own LDS row until the closing barrier)
	}
	const bool valid = idx < a.P;
	const bool vis = valid && ((pf ? in.radius : a.radii[idx]) > 0);
	// raw scene geometry (preprocess.hip): position / log-scale / raw rotation / opacity logit of Gaussians idx < Ns come from the raw
	// tensors, and their gradients -- chain rule through exp / normalize / sigmoid included -- go to the raw tensors' gradients
	const bool rs = a.sh_src.scene_xyz != nullptr && idx < a.sh_src.Ns;
	if (valid && !vis) {
		// v2: every output row is written here, so the caller does not have to zero-fill them
		if (a.gacc) {
			a.out_mean2D[3 * (size_t)idx] = 0.f; a.out_mean2D[3 * (size_t)idx + 1] = 0.f; a.out_mean2D[3 * (size_t)idx + 2] = 0.f;
			// out_conic / out_color / out_depth / dL_dcov3D are intermediates of the reference's ABI: NULL = not wanted
			if (a.out_conic) *reinterpret_cast<float4*>(a.out_conic + 4 * (size_t)idx) = make_float4(0.f, 0.f, 0.f, 0.f);
			if (!rs) a.out_opacity[idx] = 0.f;
			if (a.out_depth) a.out_depth[idx] = 0.f;
			if (a.out_color) { a.out_color[3 * (size_t)idx] = 0.f; a.out_color[3 * (size_t)idx + 1] = 0.f; a.out_color[3 * (size_t)idx + 2] = 0.f; }
			if (a.out_flow && !rs) { a.out_flow[3 * (size_t)idx] = 0.f; a.out_flow[3 * (size_t)idx + 1] = 0.f; a.out_flow[3 * (size_t)idx + 2] = 0.f; }
			if (a.out_sem && a.D_S >= 1) a.out_sem[(size_t)idx * a.D_S] = 0.f;      // channels >= 1: api.hip (extra replays)
			if (rs) {      // raw scene row: the gradients of the raw tensors are what the caller reads
				float* gx = a.sh_dst.scene_xyz + 3 * (size_t)idx; gx[0] = 0.f; gx[1] = 0.f; gx[2] = 0.f;
				float* gs = a.sh_dst.scene_scaling + 3 * (size_t)idx; gs[0] = 0.f; gs[1] = 0.f; gs[2] = 0.f;
				*reinterpret_cast<float4*>(a.sh_dst.scene_rotation + 4 * (size_t)idx) = make_float4(0.f, 0.f, 0.f, 0.f);
				a.sh_dst.scene_opacity[idx] = 0.f;
			} else {
				a.dL_dmean3D[3 * (size_t)idx] = 0.f; a.dL_dmean3D[3 * (size_t)idx + 1] = 0.f; a.dL_dmean3D[3 * (size_t)idx + 2] = 0.f;
			}
			if (a.dL_dcov3D) for (int i = 0; i < 6; i++) a.dL_dcov3D[6 * (size_t)idx + i] = 0.f;
			if (raw) {
				const bool is_obj = idx >= a.sh_src.Ns;
				const size_t m = is_obj ? idx - a.sh_src.Ns : idx;
				float* gdc = (is_obj ? a.sh_dst.obj_dc : a.sh_dst.scene_dc);
				float* gre = (is_obj ? a.sh_dst.obj_rest : a.sh_dst.scene_rest);
				if (gdc) { gdc[3 * m] = 0.f; gdc[3 * m + 1] = 0.f; gdc[3 * m + 2] = 0.f; }
				if (a.sh_dst.rgb_factor) { float* rf = a.sh_dst.rgb_factor + 3 * (size_t)idx; rf[0] = 0.f; rf[1] = 0.f; rf[2] = 0.f; }
				if (STAGED) { if (gre || adam_rest) for (int i = 0; i < SH_ROW_REST; i++) s_sh[tid * SH_ROW_REST + i] = 0.f; }
				else if (gre) for (int i = 0; i < (a.M - 1) * 3; i++) gre[m * (size_t)(a.M - 1) * 3 + i] = 0.f;
			} else if (a.shs) {
				if (STAGED) { for (int i = 0; i < SH_ROW_FULL; i++) s_sh[tid * SH_ROW_FULL_LDS + i] = 0.f; }
				else { float* dsh = a.dL_dsh + (size_t)idx * a.M * 3; for (int i = 0; i < a.M * 3; i++) dsh[i] = 0.f; }
			}
			if (a.scales && !rs) {
				a.dL_dscale[3 * (size_t)idx] = 0.f; a.dL_dscale[3 * (size_t)idx + 1] = 0.f; a.dL_dscale[3 * (size_t)idx + 2] = 0.f;
				*reinterpret_cast<float4*>(a.dL_drot + 4 * (size_t)idx) = make_float4(0.f, 0.f, 0.f, 0.f);
			}
		}
	}
	if (vis) {
	const float* V = a.view; const float* PJ = a.proj;
	const float* pos = rs ? a.sh_src.scene_xyz : a.means3D;
	const float mx = pf ? in.p[0] : pos[3 * (size_t)idx], my = pf ? in.p[1] : pos[3 * (size_t)idx + 1], mz = pf ? in.p[2] : pos[3 * (size_t)idx + 2];
	SceneAct act;
	if (rs) act = pf ? scene_activations(in.s, in.q, &in.op, (size_t)0)
	                 : scene_activations(a.sh_src.scene_scaling, a.sh_src.scene_rotation, a.sh_src.scene_opacity, (size_t)idx);
	float gflow[3] = { 0.f, 0.f, 0.f };      // raw scene row: the flow point IS the position, its gradient joins the position's

	// per-Gaussian sums of the blend backward: classic = separate ABI arrays filled by atomics,
	// v2 = one packed 64-byte line per Gaussian, unpacked here into the ABI outputs
	float dcon_x, dcon_y, dcon_z, g2x, g2y, gd, gcol[3];
	if (a.gacc) {
		const float4* ga = reinterpret_cast<const float4*>(a.gacc + (size_t)idx * GACC_STRIDE);
		const float4 u0 = pf ? in.u0 : ga[0], u1 = pf ? in.u1 : ga[1], u2 = pf ? in.u2 : ga[2], u3 = pf ? in.u3 : ga[3];
		// (the line stays as it is: api.hip zeroes the accumulator again before a second backward over the same forward)
		// u0 = (S0, Sx, Sy, Sxx), u1 = (Sxy, Syy, c0, c1): raw moment sums of L = G*dL/dalpha (render_v2.hip);
		// the per-Gaussian factors of backward.cu:626-643 are applied here, once per Gaussian
		const float4* sp = reinterpret_cast<const float4*>(a.splats + idx);
		const float4 s0 = pf ? in.s0 : sp[0], s1 = pf ? in.s1 : sp[1];               // x y ca cb | cc op r g
		const float op = s1.y, qa = s0.z, qb = s0.w, qc = s1.x;
		g2x = -op * (qa * u0.y + qb * u0.z) * (float)(0.5 * a.W);
		g2y = -op * (qc * u0.z + qb * u0.y) * (float)(0.5 * a.H);
		dcon_x = -0.5f * op * u0.w; dcon_y = -0.5f * op * u1.x; dcon_z = -0.5f * op * u1.y;
		gcol[0] = u1.z; gcol[1] = u1.w; gcol[2] = u2.x; gd = u2.y;
		st_stream(a.out_mean2D + 3 * (size_t)idx, g2x); st_stream(a.out_mean2D + 3 * (size_t)idx + 1, g2y); st_stream(a.out_mean2D + 3 * (size_t)idx + 2, 0.f);
		if (a.out_conic) *reinterpret_cast<float4*>(a.out_conic + 4 * (size_t)idx) = make_float4(dcon_x, dcon_y, 0.f, dcon_z);
		if (rs) st_stream(a.sh_dst.scene_opacity + idx, u0.x * act.op * (1.f - act.op));      // d sigmoid
		else st_stream(a.out_opacity + idx, u0.x);
		if (a.out_color) { a.out_color[3 * (size_t)idx] = gcol[0]; a.out_color[3 * (size_t)idx + 1] = gcol[1]; a.out_color[3 * (size_t)idx + 2] = gcol[2]; }
		if (a.out_depth) a.out_depth[idx] = gd;
		if (a.out_flow && rs) { gflow[0] = u2.z; gflow[1] = u2.w; gflow[2] = u3.x; }
		else if (a.out_flow) { a.out_flow[3 * (size_t)idx] = u2.z; a.out_flow[3 * (size_t)idx + 1] = u2.w; a.out_flow[3 * (size_t)idx + 2] = u3.x; }
		if (a.out_sem && a.D_S >= 1) a.out_sem[(size_t)idx * a.D_S] = u3.y;
	} else {
		dcon_x = a.dL_dconic[4 * (size_t)idx]; dcon_y = a.dL_dconic[4 * (size_t)idx + 1]; dcon_z = a.dL_dconic[4 * (size_t)idx + 3];
		g2x = a.dL_dmean2D[3 * (size_t)idx]; g2y = a.dL_dmean2D[3 * (size_t)idx + 1];
		gd = a.dL_ddepth[idx];
		gcol[0] = a.dL_dcolor[3 * (size_t)idx]; gcol[1] = a.dL_dcolor[3 * (size_t)idx + 1]; gcol[2] = a.dL_dcolor[3 * (size_t)idx + 2];
	}
	// ---------------- cov2D backward (backward.cu:144-274)
	// in registers either way: a pointer that may refer to a local array would put that array into scratch memory
	float c3[6];
	if (a.cov3D) {
#pragma unroll
		for (int i = 0; i < 6; i++) c3[i] = a.cov3D[6 * (size_t)idx + i];
	} else if (rs) {
		cov3d_from_values(act.s[0], act.s[1], act.s[2], a.scale_modifier, act.q[0], act.q[1], act.q[2], act.q[3], c3);
	} else if (pf) {
		cov3d_from_scale_rot(in.s, a.scale_modifier, in.q, c3);
	} else {                  // v2 without cov3D_precomp: recomputed instead of stored by the forward (24 B written + read per Gaussian)
		cov3d_from_scale_rot(a.scales + 3 * (size_t)idx, a.scale_modifier, a.rotations + 4 * (size_t)idx, c3);
	}
	float tx = V[0] * mx + V[4] * my + V[8] * mz + V[12];
	float ty = V[1] * mx + V[5] * my + V[9] * mz + V[13];
	const float tz = V[2] * mx + V[6] * my + V[10] * mz + V[14];
	const float h_x = a.focal_x, h_y = a.focal_y;
	const float limx = 1.3f * a.tan_fovx, limy = 1.3f * a.tan_fovy;
	const float txtz = tx / tz, tytz = ty / tz;
	tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
	ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
	const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
	const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
	M3 J = { { { h_x / tz, 0.0f, -(h_x * tx) / (tz * tz) }, { 0.0f, h_y / tz, -(h_y * ty) / (tz * tz) }, { 0.f, 0.f, 0.f } } };
	M3 Wm = { { { V[0], V[4], V[8] }, { V[1], V[5], V[9] }, { V[2], V[6], V[10] } } };
	M3 Vrk = { { { c3[0], c3[1], c3[2] }, { c3[1], c3[3], c3[4] }, { c3[2], c3[4], c3[5] } } };
	M3 T = m3mul(Wm, J);
	M3 cov2D = m3mul(m3mul(m3t(T), m3t(Vrk)), T);
	const float ca = cov2D.v[0][0] + 0.3f, cb = cov2D.v[0][1], cc = cov2D.v[1][1] + 0.3f;
	const float denom = ca * cc - cb * cb;
	float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
	const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
	float dcov[6];
	if (denom2inv != 0) {
		dL_da = denom2inv * (-cc * cc * dcon_x + 2 * cb * cc * dcon_y + (denom - ca * cc) * dcon_z);
		dL_dc = denom2inv * (-ca * ca * dcon_z + 2 * ca * cb * dcon_y + (denom - ca * cc) * dcon_x);
		dL_db = denom2inv * 2 * (cb * cc * dcon_x - (denom + 2 * cb * cb) * dcon_y + ca * cb * dcon_z);
		dcov[0] = (T.v[0][0] * T.v[0][0] * dL_da + T.v[0][0] * T.v[1][0] * dL_db + T.v[1][0] * T.v[1][0] * dL_dc);
		dcov[3] = (T.v[0][1] * T.v[0][1] * dL_da + T.v[0][1] * T.v[1][1] * dL_db + T.v[1][1] * T.v[1][1] * dL_dc);
		dcov[5] = (T.v[0][2] * T.v[0][2] * dL_da + T.v[0][2] * T.v[1][2] * dL_db + T.v[1][2] * T.v[1][2] * dL_dc);
		dcov[1] = 2 * T.v[0][0] * T.v[0][1] * dL_da + (T.v[0][0] * T.v[1][1] + T.v[0][1] * T.v[1][0]) * dL_db + 2 * T.v[1][0] * T.v[1][1] * dL_dc;
		dcov[2] = 2 * T.v[0][0] * T.v[0][2] * dL_da + (T.v[0][0] * T.v[1][2] + T.v[0][2] * T.v[1][0]) * dL_db + 2 * T.v[1][0] * T.v[1][2] * dL_dc;
		dcov[4] = 2 * T.v[0][2] * T.v[0][1] * dL_da + (T.v[0][1] * T.v[1][2] + T.v[0][2] * T.v[1][1]) * dL_db + 2 * T.v[1][1] * T.v[1][2] * dL_dc;
	} else {
#pragma unroll
		for (int i = 0; i < 6; i++) dcov[i] = 0.f;
	}
	if (a.dL_dcov3D) {
#pragma unroll
		for (int i = 0; i < 6; i++) a.dL_dcov3D[6 * (size_t)idx + i] = dcov[i];
	}

	const float tv0 = T.v[0][0] * Vrk.v[0][0] + T.v[0][1] * Vrk.v[0][1] + T.v[0][2] * Vrk.v[0][2];
	const float tv1 = T.v[0][0] * Vrk.v[1][0] + T.v[0][1] * Vrk.v[1][1] + T.v[0][2] * Vrk.v[1][2];
	const float tv2 = T.v[0][0] * Vrk.v[2][0] + T.v[0][1] * Vrk.v[2][1] + T.v[0][2] * Vrk.v[2][2];
	const float uv0 = T.v[1][0] * Vrk.v[0][0] + T.v[1][1] * Vrk.v[0][1] + T.v[1][2] * Vrk.v[0][2];
	const float uv1 = T.v[1][0] * Vrk.v[1][0] + T.v[1][1] * Vrk.v[1][1] + T.v[1][2] * Vrk.v[1][2];
	const float uv2 = T.v[1][0] * Vrk.v[2][0] + T.v[1][1] * Vrk.v[2][1] + T.v[1][2] * Vrk.v[2][2];
	const float dL_dT00 = 2 * tv0 * dL_da + uv0 * dL_db;
	const float dL_dT01 = 2 * tv1 * dL_da + uv1 * dL_db;
	const float dL_dT02 = 2 * tv2 * dL_da + uv2 * dL_db;
	const float dL_dT10 = 2 * uv0 * dL_dc + tv0 * dL_db;
	const float dL_dT11 = 2 * uv1 * dL_dc + tv1 * dL_db;
	const float dL_dT12 = 2 * uv2 * dL_dc + tv2 * dL_db;
	const float dL_dJ00 = Wm.v[0][0] * dL_dT00 + Wm.v[0][1] * dL_dT01 + Wm.v[0][2] * dL_dT02;
	const float dL_dJ02 = Wm.v[2][0] * dL_dT00 + Wm.v[2][1] * dL_dT01 + Wm.v[2][2] * dL_dT02;
	const float dL_dJ11 = Wm.v[1][0] * dL_dT10 + Wm.v[1][1] * dL_dT11 + Wm.v[1][2] * dL_dT12;
	const float dL_dJ12 = Wm.v[2][0] * dL_dT10 + Wm.v[2][1] * dL_dT11 + Wm.v[2][2] * dL_dT12;
	const float itz = 1.f / tz, itz2 = itz * itz, itz3 = itz2 * itz;
	const float dL_dtx = x_grad_mul * -h_x * itz2 * dL_dJ02;
	const float dL_dty = y_grad_mul * -h_y * itz2 * dL_dJ12;
	const float dL_dtz = -h_x * itz2 * dL_dJ00 - h_y * itz2 * dL_dJ11 + (2 * h_x * tx) * itz3 * dL_dJ02 + (2 * h_y * ty) * itz3 * dL_dJ12;
	// transformVec4x3Transpose (auxiliary.h:89-97); this ASSIGNS (backward.cu:273)
	float gmx = V[0] * dL_dtx + V[1] * dL_dty + V[2] * dL_dtz;
	float gmy = V[4] * dL_dtx + V[5] * dL_dty + V[6] * dL_dtz;
	float gmz = V[8] * dL_dtx + V[9] * dL_dty + V[10] * dL_dtz;

	// ---------------- projection path (backward.cu:371-390)
	{
		const float hw = PJ[3] * mx + PJ[7] * my + PJ[11] * mz + PJ[15];
		const float m_w = 1.0f / (hw + 0.0000001f);
		const float mul1 = (PJ[0] * mx + PJ[4] * my + PJ[8] * mz + PJ[12]) * m_w * m_w;
		const float mul2 = (PJ[1] * mx + PJ[5] * my + PJ[9] * mz + PJ[13]) * m_w * m_w;
		gmx += (PJ[0] * m_w - PJ[3] * mul1) * g2x + (PJ[1] * m_w - PJ[3] * mul2) * g2y;
		gmy += (PJ[4] * m_w - PJ[7] * mul1) * g2x + (PJ[5] * m_w - PJ[7] * mul2) * g2y;
		gmz += (PJ[8] * m_w - PJ[11] * mul1) * g2x + (PJ[9] * m_w - PJ[11] * mul2) * g2y;
	}
	// ---------------- depth path (backward.cu:392-405)
	{
		const float mul3 = V[2] * mx + V[6] * my + V[10] * mz + V[14];
		const float demon = a.inv_depth ? (-1.0f / (mul3 * mul3 + 0.0000001f)) : 1.0f;
		gmx += (V[2] - V[3] * mul3) * gd * demon;
		gmy += (V[6] - V[7] * mul3) * gd * demon;
		gmz += (V[10] - V[11] * mul3) * gd * demon;
	}
	// ---------------- SH path (backward.cu:20-139)
	if (a.shs || a.sh_src.scene_dc) {
		const float ox = mx - a.campos[0], oy = my - a.campos[1], oz = mz - a.campos[2];
		const float len = sqrtf(ox * ox + oy * oy + oz * oz);
		const float x = ox / len, y = oy / len, z = oz / len;
		// `sh[k*3+c]` / `dsh[k*3+c]` are used for k >= 1 only; coefficient 0 goes through dsh0.
		// STAGED: sh and dsh are the SAME LDS row, so every read of sh happens before any write of dsh.
		const float* sh; float* dsh; float* dsh0;
		bool want_rows = true;       // raw path: false when the caller did not ask for the gradient of the `rest` coefficients
		if (raw) {
			const bool is_obj = idx >= a.sh_src.Ns;
			const size_t m = is_obj ? idx - a.sh_src.Ns : idx;
			float* gdc = is_obj ? a.sh_dst.obj_dc : a.sh_dst.scene_dc;
			float* gre = is_obj ? a.sh_dst.obj_rest : a.sh_dst.scene_rest;
			dsh0 = gdc ? gdc + 3 * m : nullptr;
			want_rows = gre != nullptr || adam_rest;
			if (STAGED) { dsh = s_sh + tid * SH_ROW_REST - 3; sh = dsh; }
			else {
				sh = (is_obj ? a.sh_src.obj_rest : a.sh_src.scene_rest) + m * (size_t)(a.M - 1) * 3 - 3;
				dsh = gre + m * (size_t)(a.M - 1) * 3 - 3;
			}
		} else if (STAGED) {
			dsh = s_sh + tid * SH_ROW_FULL_LDS; sh = dsh; dsh0 = dsh;
		} else {
			sh = a.shs + (size_t)idx * a.M * 3;
			dsh = a.dL_dsh + (size_t)idx * a.M * 3;
			dsh0 = dsh;
		}
		const uint8_t cl = pf ? in.clamped : a.clamped[idx];
		float g[3];
#pragma unroll
		for (int c = 0; c < 3; c++) g[c] = ((cl >> c) & 1) ? 0.f : gcol[c];
		float dx3[3] = { 0.f, 0.f, 0.f }, dy3[3] = { 0.f, 0.f, 0.f }, dz3[3] = { 0.f, 0.f, 0.f };
		const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
		const float C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f };
		const float C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
			-0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f };
		const int deg = a.D;
		// ---- pass A: everything that READS the coefficients (d colour / d direction)
		float xx = 0.f, yy = 0.f, zz = 0.f, xy = 0.f, yz = 0.f, xz = 0.f;
		if (have_ddir) {
#pragma unroll
			for (int c = 0; c < 3; c++) { dx3[c] = in.dd[3 * c]; dy3[c] = in.dd[3 * c + 1]; dz3[c] = in.dd[3 * c + 2]; }
		} else if (deg > 0) {
#pragma unroll
			for (int c = 0; c < 3; c++) { dx3[c] = -C1 * sh[3 * 3 + c]; dy3[c] = -C1 * sh[1 * 3 + c]; dz3[c] = C1 * sh[2 * 3 + c]; }
			if (deg > 1) {
				xx = x * x; yy = y * y; zz = z * z; xy = x * y; yz = y * z; xz = x * z;
#pragma unroll
				for (int c = 0; c < 3; c++) {
					dx3[c] += C2[0] * y * sh[4 * 3 + c] + C2[2] * 2.f * -x * sh[6 * 3 + c] + C2[3] * z * sh[7 * 3 + c] + C2[4] * 2.f * x * sh[8 * 3 + c];
					dy3[c] += C2[0] * x * sh[4 * 3 + c] + C2[1] * z * sh[5 * 3 + c] + C2[2] * 2.f * -y * sh[6 * 3 + c] + C2[4] * 2.f * -y * sh[8 * 3 + c];
					dz3[c] += C2[1] * y * sh[5 * 3 + c] + C2[2] * 2.f * 2.f * z * sh[6 * 3 + c] + C2[3] * x * sh[7 * 3 + c];
				}
				if (deg > 2) {
#pragma unroll
					for (int c = 0; c < 3; c++) {
						dx3[c] += (C3[0] * sh[9 * 3 + c] * 3.f * 2.f * xy + C3[1] * sh[10 * 3 + c] * yz + C3[2] * sh[11 * 3 + c] * -2.f * xy +
							C3[3] * sh[12 * 3 + c] * -3.f * 2.f * xz + C3[4] * sh[13 * 3 + c] * (-3.f * xx + 4.f * zz - yy) +
							C3[5] * sh[14 * 3 + c] * 2.f * xz + C3[6] * sh[15 * 3 + c] * 3.f * (xx - yy));
						dy3[c] += (C3[0] * sh[9 * 3 + c] * 3.f * (xx - yy) + C3[1] * sh[10 * 3 + c] * xz + C3[2] * sh[11 * 3 + c] * (-3.f * yy + 4.f * zz - xx) +
							C3[3] * sh[12 * 3 + c] * -3.f * 2.f * yz + C3[4] * sh[13 * 3 + c] * -2.f * xy + C3[5] * sh[14 * 3 + c] * -2.f * yz +
							C3[6] * sh[15 * 3 + c] * -3.f * 2.f * xy);
						dz3[c] += (C3[1] * sh[10 * 3 + c] * xy + C3[2] * sh[11 * 3 + c] * 4.f * 2.f * yz + C3[3] * sh[12 * 3 + c] * 3.f * (2.f * zz - xx - yy) +
							C3[4] * sh[13 * 3 + c] * 4.f * 2.f * xz + C3[5] * sh[14 * 3 + c] * (xx - yy));
					}
				}
			}
		}
		// ---- pass B: gradients w.r.t. the coefficients (WRITES; may alias the row read above)
		if (dsh0) { dsh0[0] = C0 * g[0]; dsh0[1] = C0 * g[1]; dsh0[2] = C0 * g[2]; }
		if (raw && a.sh_dst.rgb_factor) { float* rf = a.sh_dst.rgb_factor + 3 * (size_t)idx; rf[0] = g[0]; rf[1] = g[1]; rf[2] = g[2]; }
		if (want_rows) {
			float coef[16];
			sh_coef_factors(deg, x, y, z, coef);
			// coefficients above the active degree receive no gradient (their factor is 0)
#pragma unroll
			for (int k = 1; k < 16; k++)
				if (k < a.M) { dsh[k * 3] = coef[k] * g[0]; dsh[k * 3 + 1] = coef[k] * g[1]; dsh[k * 3 + 2] = coef[k] * g[2]; }
		}
		const float ddx = dx3[0] * g[0] + dx3[1] * g[1] + dx3[2] * g[2];
		const float ddy = dy3[0] * g[0] + dy3[1] * g[1] + dy3[2] * g[2];
		const float ddz = dz3[0] * g[0] + dz3[1] * g[1] + dz3[2] * g[2];
		// dnormvdv (auxiliary.h:107-117)
		const float sum2 = ox * ox + oy * oy + oz * oz;
		const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
		gmx += ((+sum2 - ox * ox) * ddx - oy * ox * ddy - oz * ox * ddz) * invsum32;
		gmy += (-ox * oy * ddx + (sum2 - oy * oy) * ddy - oz * oy * ddz) * invsum32;
		gmz += (-ox * oz * ddx - oy * oz * ddy + (sum2 - oz * oz) * ddz) * invsum32;
	}
	if (rs) { float* gx = a.sh_dst.scene_xyz + 3 * (size_t)idx; st_stream(gx, gmx + gflow[0]); st_stream(gx + 1, gmy + gflow[1]); st_stream(gx + 2, gmz + gflow[2]); }
	else { st_stream(a.dL_dmean3D + 3 * (size_t)idx, gmx); st_stream(a.dL_dmean3D + 3 * (size_t)idx + 1, gmy); st_stream(a.dL_dmean3D + 3 * (size_t)idx + 2, gmz); }

	// ---------------- cov3D -> scale / rotation (backward.cu:278-341)
	if (a.scales) {
		float r, x, y, z;
		if (rs) { r = act.q[0]; x = act.q[1]; y = act.q[2]; z = act.q[3]; }
		else if (pf) { r = in.q[0]; x = in.q[1]; y = in.q[2]; z = in.q[3]; }
		else { const float* q = a.rotations + 4 * (size_t)idx; r = q[0]; x = q[1]; y = q[2]; z = q[3]; }
		const float sc0 = rs ? act.s[0] : (pf ? in.s[0] : a.scales[3 * (size_t)idx]), sc1 = rs ? act.s[1] : (pf ? in.s[1] : a.scales[3 * (size_t)idx + 1]),
			sc2 = rs ? act.s[2] : (pf ? in.s[2] : a.scales[3 * (size_t)idx + 2]);
		M3 R = { { { 1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y) },
		           { 2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x) },
		           { 2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y) } } };
		const float s0 = a.scale_modifier * sc0, s1 = a.scale_modifier * sc1, s2 = a.scale_modifier * sc2;
		M3 S = { { { s0, 0.f, 0.f }, { 0.f, s1, 0.f }, { 0.f, 0.f, s2 } } };
		M3 Mm = m3mul(S, R);
		M3 dSig = { { { dcov[0], 0.5f * dcov[1], 0.5f * dcov[2] }, { 0.5f * dcov[1], dcov[3], 0.5f * dcov[4] }, { 0.5f * dcov[2], 0.5f * dcov[4], dcov[5] } } };
		M3 twoM;
#pragma unroll
		for (int c = 0; c < 3; c++)
#pragma unroll
			for (int rr = 0; rr < 3; rr++) twoM.v[c][rr] = Mm.v[c][rr] * 2.0f;
		M3 dL_dM = m3mul(twoM, dSig);
		M3 Rt = m3t(R);
		M3 dMt = m3t(dL_dM);
		const float ds0 = Rt.v[0][0] * dMt.v[0][0] + Rt.v[0][1] * dMt.v[0][1] + Rt.v[0][2] * dMt.v[0][2];
		const float ds1 = Rt.v[1][0] * dMt.v[1][0] + Rt.v[1][1] * dMt.v[1][1] + Rt.v[1][2] * dMt.v[1][2];
		const float ds2 = Rt.v[2][0] * dMt.v[2][0] + Rt.v[2][1] * dMt.v[2][1] + Rt.v[2][2] * dMt.v[2][2];
		if (rs) { float* gs = a.sh_dst.scene_scaling + 3 * (size_t)idx; st_stream(gs, ds0 * sc0); st_stream(gs + 1, ds1 * sc1); st_stream(gs + 2, ds2 * sc2); }      // d exp
		else { st_stream(a.dL_dscale + 3 * (size_t)idx + 0, ds0); st_stream(a.dL_dscale + 3 * (size_t)idx + 1, ds1); st_stream(a.dL_dscale + 3 * (size_t)idx + 2, ds2); }
#pragma unroll
		for (int k = 0; k < 3; k++) { dMt.v[0][k] *= s0; dMt.v[1][k] *= s1; dMt.v[2][k] *= s2; }
#define MT(i, j) dMt.v[i][j]
		float4 dq;
		dq.x = 2 * z * (MT(0, 1) - MT(1, 0)) + 2 * y * (MT(2, 0) - MT(0, 2)) + 2 * x * (MT(1, 2) - MT(2, 1));
		dq.y = 2 * y * (MT(1, 0) + MT(0, 1)) + 2 * z * (MT(2, 0) + MT(0, 2)) + 2 * r * (MT(1, 2) - MT(2, 1)) - 4 * x * (MT(2, 2) + MT(1, 1));
		dq.z = 2 * x * (MT(1, 0) + MT(0, 1)) + 2 * r * (MT(2, 0) - MT(0, 2)) + 2 * z * (MT(1, 2) + MT(2, 1)) - 4 * y * (MT(2, 2) + MT(0, 0));
		dq.w = 2 * r * (MT(0, 1) - MT(1, 0)) + 2 * x * (MT(2, 0) + MT(0, 2)) + 2 * y * (MT(1, 2) + MT(2, 1)) - 4 * z * (MT(1, 1) + MT(0, 0));
#undef MT
		if (rs) {      // the Python-side F.normalize of the reference (scene/gaussian_model.py:44): (g - q (q . g)) / |raw|
			const float qg = r * dq.x + x * dq.y + y * dq.z + z * dq.w;
			st_stream4(reinterpret_cast<float4*>(a.sh_dst.scene_rotation + 4 * (size_t)idx),
				make_float4((dq.x - r * qg) * act.inv_norm, (dq.y - x * qg) * act.inv_norm, (dq.z - y * qg) * act.inv_norm, (dq.w - z * qg) * act.inv_norm));
		} else st_stream4(reinterpret_cast<float4*>(a.dL_drot + 4 * (size_t)idx), dq);     // no normalisation Jacobian (backward.cu:340)
	}
	}   // if (vis)
	if (STAGED) {
		// the LDS rows now hold the SH gradients: stream them out fully coalesced
		__syncthreads();
		if (raw && adam_rest) {
			// a Gaussian outside the frustum has a zero gradient row, not no row: Adam moves it by its moments (torch.optim.Adam is dense)
			adam_rows<4>(s_sh, SH_ROW_REST, SH_ROW_REST, base, nvalid, a.sh_src.Ns, a.sh_dst.adam.scene_rest, a.sh_dst.adam.obj_rest, a.sh_dst.adam.beta1, a.sh_dst.adam.beta2, a.sh_dst.adam.eps,
				tid, BW_THREADS);
		} else if (raw) {
			stage_rows<false>(s_sh, SH_ROW_REST, SH_ROW_REST, base, nvalid, a.sh_src.Ns, a.sh_dst.scene_rest, a.sh_dst.obj_rest, tid, BW_THREADS);
		} else {
			float4* dst = reinterpret_cast<float4*>(a.dL_dsh + (size_t)base * SH_ROW_FULL);
			for (int q = tid; q < nvalid * (SH_ROW_FULL / 4); q += BW_THREADS) {
				const int g = q / (SH_ROW_FULL / 4), c = (q - g * (SH_ROW_FULL / 4)) * 4;
				const float* s = s_sh + g * SH_ROW_FULL_LDS + c;
				st_stream4(dst + q, make_float4(s[0], s[1], s[2], s[3]));
			}
		}
	}
}

} // namespace

int launch_preprocess_bwd(const PreprocessBwdArgs& a, hipStream_t stream) {
	if (a.P == 0) return 0;
	const bool raw = a.sh_src.scene_dc != nullptr;
	// the staged kernel needs 16 SH coefficients and, in the v2 (gacc) path, writes every row
	const bool staged = a.M == 16 && (raw || a.shs) && a.gacc != nullptr && a.sh_staging != 0;
	const unsigned grid = (unsigned)((a.P + BW_THREADS - 1) / BW_THREADS);
	const bool adam_rest = a.sh_dst.adam.scene_rest.p || a.sh_dst.adam.obj_rest.p, adam_sp = a.sh_dst.adam.scene_sp.p || a.sh_dst.adam.obj_sp.p;
	if ((adam_rest || adam_sp) && !raw) { set_error("preprocess backward: the in-backward Adam step needs the raw-SH path"); return -1; }
	if (adam_rest && !staged) {
		set_error("preprocess backward: the in-backward Adam step of the SH `rest` tensors needs 16 SH coefficients and the LDS row staging "
		          "(the frame's forward ran with ADGS_NO_SH_STAGING, or M != 16)");
		return -1;
	}
	if (staged) {
		const size_t lds = (size_t)BW_THREADS * (raw ? SH_ROW_REST : SH_ROW_FULL_LDS) * sizeof(float);
		hipLaunchKernelGGL(preprocess_bwd_kernel<true>, dim3(grid), dim3(BW_THREADS), lds, stream, a);
	} else {
		hipLaunchKernelGGL(preprocess_bwd_kernel<false>, dim3(grid), dim3(BW_THREADS), 0, stream, a);
	}
	ADGS_HIP_CHECK(hipGetLastError());
	// raw-SH path: d/d(shs_deform_param[m, c, k]) = w_k * dL/d(dc[m, c]) as flat coalesced passes
	if (raw && has_lin_host(a.sh_src.f)) {
		const int No = a.P - a.sh_src.Ns;
		if (launch_lin_param_grad2(a.sh_src.Ns, a.sh_dst.scene_dc, a.sh_dst.scene_sp, No, a.sh_dst.obj_dc, a.sh_dst.obj_sp, 3, 3, a.sh_src.f, stream,
		                           &a.sh_dst.adam.scene_sp, &a.sh_dst.adam.obj_sp, a.sh_dst.adam.beta1, a.sh_dst.adam.beta2, a.sh_dst.adam.eps) != 0) return -1;
	}
	return 0;
}

} // namespace adgs
