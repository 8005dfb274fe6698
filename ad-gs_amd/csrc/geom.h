// Geometry helpers shared by the forward and backward preprocess kernels.
#pragma once
#include "common.h"

namespace adgs {

// computeCov3D (forward.cu:118-152): Sigma = (S R)^T (S R), S = diag(mod * scale), R from the UN-normalised (r,x,y,z),
// upper triangle out.  Evaluated in glm's column-major order.  Contraction is pinned off inside this function (it is
// self-contained for that reason) so that the forward preprocess and the v2 backward -- which recomputes Sigma instead of
// reading it back -- produce the same bits whatever the flags of their translation units.
__device__ __forceinline__ void cov3d_from_values(float s0, float s1, float s2, float mod, float r, float x, float y, float z, float* out) {
#pragma clang fp contract(off)
	const float sx = mod * s0, sy = mod * s1, sz = mod * s2;
	// R.v[col][row] (column-major, as glm stores mat3(...) given row by row in forward.cu:133-137)
	const float Rm[3][3] = { { 1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y) },
	                         { 2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x) },
	                         { 2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y) } };
	const float Sd[3] = { sx, sy, sz };
	// M = S * R : M.v[c][rr] = sum_k S.v[k][rr] * R.v[c][k]; S diagonal -> S.v[rr][rr] * R.v[c][rr] with the two zero terms
	// added in glm's order (0*x + ... keeps the value; written out so the sums match m3mul bit for bit)
	float M[3][3];
#pragma unroll
	for (int c = 0; c < 3; c++)
#pragma unroll
		for (int rr = 0; rr < 3; rr++) {
			const float a0 = (rr == 0 ? Sd[0] : 0.f) * Rm[c][0], a1 = (rr == 1 ? Sd[1] : 0.f) * Rm[c][1], a2 = (rr == 2 ? Sd[2] : 0.f) * Rm[c][2];
			M[c][rr] = a0 + a1 + a2;
		}
	// Sigma = M^T * M : Sig.v[c][rr] = sum_k Mt.v[k][rr] * M.v[c][k],  Mt.v[k][rr] = M.v[rr][k]
	float Sg[3][3];
#pragma unroll
	for (int c = 0; c < 3; c++)
#pragma unroll
		for (int rr = 0; rr < 3; rr++)
			Sg[c][rr] = M[rr][0] * M[c][0] + M[rr][1] * M[c][1] + M[rr][2] * M[c][2];
	out[0] = Sg[0][0]; out[1] = Sg[0][1]; out[2] = Sg[0][2];
	out[3] = Sg[1][1]; out[4] = Sg[1][2]; out[5] = Sg[2][2];
}
__device__ __forceinline__ void cov3d_from_scale_rot(const float* s3, float mod, const float* q, float* out) {
	cov3d_from_values(s3[0], s3[1], s3[2], mod, q[0], q[1], q[2], q[3], out);
}

// Activations of the raw scene geometry (scene/gaussian_model.py:36-44: exp, F.normalize with eps 1e-12, sigmoid), evaluated
// by the forward preprocess AND re-evaluated by its backward: one rounding per operation so that both see the same bits.
struct SceneAct { float s[3]; float q[4]; float inv_norm; float op; };
__device__ __forceinline__ SceneAct scene_activations(const float* __restrict__ raw_scaling, const float* __restrict__ raw_rotation,
	const float* __restrict__ raw_opacity, size_t idx) {
#pragma clang fp contract(off)
	SceneAct a;
	a.s[0] = expf(raw_scaling[3 * idx]); a.s[1] = expf(raw_scaling[3 * idx + 1]); a.s[2] = expf(raw_scaling[3 * idx + 2]);
	const float4 r = *reinterpret_cast<const float4*>(raw_rotation + 4 * idx);
	a.inv_norm = 1.f / fmaxf(sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w), 1e-12f);
	a.q[0] = r.x * a.inv_norm; a.q[1] = r.y * a.inv_norm; a.q[2] = r.z * a.inv_norm; a.q[3] = r.w * a.inv_norm;
	a.op = 1.f / (1.f + expf(-raw_opacity[idx]));
	return a;
}

// d colour / d SH coefficient k for the unit view direction (x, y, z): the per-coefficient factors of
// computeColorFromSH's backward (backward.cu:44-112), coefficient 0 included; entries above the active degree are 0.
// Shared by the preprocess backward (which multiplies them by one camera's colour gradient) and by the multi-camera
// gradient expansion (exchange.hip), so that both produce the same rows.
__device__ __forceinline__ void sh_coef_factors(int deg, float x, float y, float z, float* coef) {
	const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
	const float C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f };
	const float C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
		-0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f };
#pragma unroll
	for (int k = 0; k < 16; k++) coef[k] = 0.f;
	coef[0] = C0;
	if (deg > 0) {
		coef[1] = -C1 * y; coef[2] = C1 * z; coef[3] = -C1 * x;
		if (deg > 1) {
			const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
			coef[4] = C2[0] * xy; coef[5] = C2[1] * yz; coef[6] = C2[2] * (2.f * zz - xx - yy);
			coef[7] = C2[3] * xz; coef[8] = C2[4] * (xx - yy);
			if (deg > 2) {
				coef[9] = C3[0] * y * (3.f * xx - yy);
				coef[10] = C3[1] * xy * z;
				coef[11] = C3[2] * y * (4.f * zz - xx - yy);
				coef[12] = C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
				coef[13] = C3[4] * x * (4.f * zz - xx - yy);
				coef[14] = C3[5] * z * (xx - yy);
				coef[15] = C3[6] * x * (xx - 3.f * yy);
			}
		}
	}
}

} // namespace adgs
