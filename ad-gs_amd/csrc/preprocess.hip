// Per-Gaussian forward preprocess (projection, EWA covariance, radius / tile
// rectangle, SH -> RGB) for gfx950.
//
// Reference semantics: RAST/cuda_rasterizer/forward.cu:20-256 and
// auxiliary.h:41-164.  This translation unit is
// compiled with -ffp-contract=off: the integer outputs of this stage (radii,
// tile rectangle, clamp flags, depth-key bits) must be bit-identical to a plain
// one-rounding-per-operation evaluation, and an FMA in `ceil(3*sqrt(lambda))`
// or in the tile-rectangle arithmetic would move Gaussians across tile borders.
//
// Roofline: pure HBM streaming.  Reads P*(12+12+16+4+12*M) B, writes P*4 (radii)
// + 4 (tiles_touched) and, per visible Gaussian, one 64-B Splat line + 24 B cov3D
// + 1 B clamp flags.
#include "common.h"
#include "kernels.h"
#include "geom.h"
#include <cstdlib>

#ifndef ADGS_PRE_STAGE_U4
#define ADGS_PRE_STAGE_U4 12
#endif
namespace adgs {

namespace {

struct M3 { float v[3][3]; };   // v[col][row]

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

__device__ __forceinline__ float ndc2pix(float v, int S) {
	// double arithmetic, rounded once (auxiliary.h:41-44 uses 1.0 / 0.5 literals)
	return (float)((((double)v + 1.0) * S - 1.0) * 0.5);
}

__device__ __forceinline__ void tile_rect(float px, float py, int radius, int gx, int gy,
	uint32_t& minx, uint32_t& miny, uint32_t& maxx, uint32_t& maxy) {
	minx = (uint32_t)min(gx, max(0, (int)((px - radius) / TILE_X)));
	miny = (uint32_t)min(gy, max(0, (int)((py - radius) / TILE_Y)));
	maxx = (uint32_t)min(gx, max(0, (int)((px + radius + TILE_X - 1) / TILE_X)));
	maxy = (uint32_t)min(gy, max(0, (int)((py + radius + TILE_Y - 1) / TILE_Y)));
}

// SH basis evaluation, same association order as forward.cu:20-71.
// `sh0c` is coefficient 0 of channel c; `sh[k*3+c]` must be valid for k >= 1 only
__device__ __forceinline__ float sh_channel(int deg, float sh0c, const float* sh, int c, float x, float y, float z) {
	float result = 0.28209479177387814f * sh0c;
	if (deg > 0) {
		const float C1 = 0.4886025119029199f;
		result = result - C1 * y * sh[1 * 3 + c] + C1 * z * sh[2 * 3 + c] - C1 * x * sh[3 * 3 + c];
		if (deg > 1) {
			const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
			result = result +
				1.0925484305920792f * xy * sh[4 * 3 + c] +
				-1.0925484305920792f * yz * sh[5 * 3 + c] +
				0.31539156525252005f * (2.0f * zz - xx - yy) * sh[6 * 3 + c] +
				-1.0925484305920792f * xz * sh[7 * 3 + c] +
				0.5462742152960396f * (xx - yy) * sh[8 * 3 + c];
			if (deg > 2) {
				result = result +
					-0.5900435899266435f * y * (3.0f * xx - yy) * sh[9 * 3 + c] +
					2.890611442640554f * xy * z * sh[10 * 3 + c] +
					-0.4570457994644658f * y * (4.0f * zz - xx - yy) * sh[11 * 3 + c] +
					0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12 * 3 + c] +
					-0.4570457994644658f * x * (4.0f * zz - xx - yy) * sh[13 * 3 + c] +
					1.445305721320277f * z * (xx - yy) * sh[14 * 3 + c] +
					-0.5900435899266435f * x * (xx - 3.0f * yy) * sh[15 * 3 + c];
			}
		}
	}
	return result + 0.5f;
}

// d(colour channel c) / d(unit view direction), the expressions of backward.cu:44-112 (what the preprocess backward evaluated from a second
// read of the SH row until round 5); degree 3
__device__ __forceinline__ void sh_channel_ddir(const float* sh, int c, float x, float y, float z, float& ddx, float& ddy, float& ddz) {
	const float C1 = 0.4886025119029199f;
	const float C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f };
	const float C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
		-0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f };
	const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
	float dx3 = -C1 * sh[3 * 3 + c], dy3 = -C1 * sh[1 * 3 + c], dz3 = C1 * sh[2 * 3 + c];
	dx3 += C2[0] * y * sh[4 * 3 + c] + C2[2] * 2.f * -x * sh[6 * 3 + c] + C2[3] * z * sh[7 * 3 + c] + C2[4] * 2.f * x * sh[8 * 3 + c];
	dy3 += C2[0] * x * sh[4 * 3 + c] + C2[1] * z * sh[5 * 3 + c] + C2[2] * 2.f * -y * sh[6 * 3 + c] + C2[4] * 2.f * -y * sh[8 * 3 + c];
	dz3 += C2[1] * y * sh[5 * 3 + c] + C2[2] * 2.f * 2.f * z * sh[6 * 3 + c] + C2[3] * x * sh[7 * 3 + c];
	dx3 += (C3[0] * sh[9 * 3 + c] * 3.f * 2.f * xy + C3[1] * sh[10 * 3 + c] * yz + C3[2] * sh[11 * 3 + c] * -2.f * xy +
		C3[3] * sh[12 * 3 + c] * -3.f * 2.f * xz + C3[4] * sh[13 * 3 + c] * (-3.f * xx + 4.f * zz - yy) +
		C3[5] * sh[14 * 3 + c] * 2.f * xz + C3[6] * sh[15 * 3 + c] * 3.f * (xx - yy));
	dy3 += (C3[0] * sh[9 * 3 + c] * 3.f * (xx - yy) + C3[1] * sh[10 * 3 + c] * xz + C3[2] * sh[11 * 3 + c] * (-3.f * yy + 4.f * zz - xx) +
		C3[3] * sh[12 * 3 + c] * -3.f * 2.f * yz + C3[4] * sh[13 * 3 + c] * -2.f * xy + C3[5] * sh[14 * 3 + c] * -2.f * yz +
		C3[6] * sh[15 * 3 + c] * -3.f * 2.f * xy);
	dz3 += (C3[1] * sh[10 * 3 + c] * xy + C3[2] * sh[11 * 3 + c] * 4.f * 2.f * yz + C3[3] * sh[12 * 3 + c] * 3.f * (2.f * zz - xx - yy) +
		C3[4] * sh[13 * 3 + c] * 4.f * 2.f * xz + C3[5] * sh[14 * 3 + c] * (xx - yy));
	ddx = dx3; ddy = dy3; ddz = dz3;
}

// STAGED (M == 16, SH input): the block's SH rows are loaded fully coalesced into LDS (odd row
// stride: conflict-free per-thread reads) instead of 64 scattered 192-byte rows per wave access.
constexpr int SH_ROW_FULL = 48, SH_ROW_FULL_LDS = 49, SH_ROW_REST = 45;

struct PreOut { uint32_t nfine; uint32_t zbits; };     // per Gaussian: fine tiles covered, view-space depth bits (0: not visible)

// Per-Gaussian inputs of the staged kernel, requested up front.  Read where the reference reads them they form a chain of dependent
// round trips behind the staging barrier -- position -> (near cull) -> scale / rotation / opacity -> coefficient 0 -> flow point, semantic:
// four trips of 3 - 5 us under load per workgroup, five workgroup rounds per CU.  Loaded unconditionally (clamped index, the culled
// Gaussians included) right after the SH rows have been requested, all of them ride on the staging round trip.  Same values, same
// arithmetic: only the loads move.
struct PreIn { alignas(16) float q[4]; float p[3], s[3], op, sh0[3], f[3], sem; };
__device__ __forceinline__ PreIn load_pre_in(const PreprocessArgs& a, const int idx) {
	PreIn in;
	const size_t i = (size_t)min(idx, a.P - 1);
	const bool rs = a.sh_src.scene_xyz != nullptr && (int)i < a.sh_src.Ns;
	const float* pos = rs ? a.sh_src.scene_xyz : a.means3D;
	in.p[0] = pos[3 * i]; in.p[1] = pos[3 * i + 1]; in.p[2] = pos[3 * i + 2];
	in.s[0] = in.s[1] = in.s[2] = 0.f; in.q[0] = in.q[1] = in.q[2] = in.q[3] = 0.f;
	if (!a.cov3D_precomp) {
		const float* sc = rs ? a.sh_src.scene_scaling : a.scales;
		const float* rt = rs ? a.sh_src.scene_rotation : a.rotations;
		in.s[0] = sc[3 * i]; in.s[1] = sc[3 * i + 1]; in.s[2] = sc[3 * i + 2];
		const float4 r = *reinterpret_cast<const float4*>(rt + 4 * i);
		in.q[0] = r.x; in.q[1] = r.y; in.q[2] = r.z; in.q[3] = r.w;
	}
	in.op = (rs ? a.sh_src.scene_opacity : a.opacities)[i];
	in.sh0[0] = in.sh0[1] = in.sh0[2] = 0.f;
	if (a.sh_src.scene_dc) { in.sh0[0] = a.sh0[3 * i]; in.sh0[1] = a.sh0[3 * i + 1]; in.sh0[2] = a.sh0[3 * i + 2]; }
	in.f[0] = in.f[1] = in.f[2] = 0.f;
	if (a.flow_points) {
		// a scene Gaussian of the raw-scene path does not move (its flow point is its position) and its row of flow_points is never written
		const float* fp = rs ? pos : a.flow_points;
		in.f[0] = fp[3 * i]; in.f[1] = fp[3 * i + 1]; in.f[2] = fp[3 * i + 2];
	}
	in.sem = (a.semantic && a.D_S > 0) ? a.semantic[i * a.D_S] : 0.f;
	return in;
}

template <bool STAGED>
__device__ __forceinline__ PreOut preprocess_one(const PreprocessArgs& a, const int idx, const float* s_sh, uint32_t* s_cell, const PreIn& in);

template <bool STAGED>
__global__ void __launch_bounds__(256) preprocess_fwd_kernel(PreprocessArgs a) {
	extern __shared__ float s_sh[];
	__shared__ uint32_t s_cell[MAX_CELLS];     // bucket binning: this workgroup's (cell, Gaussian) pair count per coarse cell
	const int idx = blockIdx.x * blockDim.x + threadIdx.x;
	if (a.bucket_count) for (int c = threadIdx.x; c < a.cgx * a.cgy; c += 256) s_cell[c] = 0u;
	if (idx == 0 && a.cfg_word) *a.cfg_word = a.cfg_value;
	PreIn in;
	if (STAGED) {
		const int tid = threadIdx.x, base = blockIdx.x * 256, nvalid = min(256, a.P - base);
		if (a.sh_src.scene_dc) {
			stage_rows<true, ADGS_PRE_STAGE_U4>(s_sh, SH_ROW_REST, SH_ROW_REST, base, nvalid, a.sh_src.Ns, a.sh_src.scene_rest, a.sh_src.obj_rest, tid, 256, [&]() { in = load_pre_in(a, idx); });
		} else {
			// materialised [P,16,3] SH tensor (the reference's own call path): 12 quads per thread, all requested before the first LDS
			// store (one load -> store per iteration was twelve dependent round trips), the per-Gaussian inputs with them
			const float4* src = reinterpret_cast<const float4*>(a.shs + (size_t)base * SH_ROW_FULL);
			constexpr int NQ4 = SH_ROW_FULL / 4;
			const int total4 = nvalid * NQ4;
			float4 v[NQ4];
#pragma unroll
			for (int u = 0; u < NQ4; u++) v[u] = ld_stream4(src + min(tid + u * 256, total4 - 1));
			in = load_pre_in(a, idx);
#pragma unroll
			for (int u = 0; u < NQ4; u++) {
				const int q = tid + u * 256;
				if (q < total4) {
					const int g = q / NQ4, c = (q - g * NQ4) * 4;
					float* d = s_sh + g * SH_ROW_FULL_LDS + c;
					d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
				}
			}
		}
	}
	if (STAGED || a.bucket_count) __syncthreads();
	if (a.v2 && !a.bucket_count && idx < SCAN_AUX_SLOTS) a.fine_total[idx] = 0ull;     // counters of the scan's side sum (P >= 1 block: always covered)
	PreOut o = { 0u, 0u };
	if (idx < a.P) o = preprocess_one<STAGED>(a, idx, s_sh, s_cell, in);
	if (a.bucket_count) {
		// bucket binning (binning.hip): no scan pass runs.  The pair counts per coarse cell were summed in LDS (the Gaussians of an
		// object are neighbours in index AND on the screen: global atomics serialise on a few hot cells) and go out as this
		// workgroup's row of the counts matrix; the fine-tile total (capacity bound of the chunk pool) is one atomic per workgroup.
		__shared__ uint32_t s_red[256 / WAVE];
		uint32_t nf = o.nfine;
#pragma unroll
		for (int off = WAVE / 2; off > 0; off >>= 1) nf += __shfl_xor(nf, off, WAVE);
		if ((threadIdx.x & (WAVE - 1)) == 0) s_red[threadIdx.x / WAVE] = nf;
		__syncthreads();
		for (int c = threadIdx.x; c < a.cgx * a.cgy; c += 256) a.bucket_count[(size_t)blockIdx.x * (a.cgx * a.cgy) + c] = s_cell[c];
		if (threadIdx.x == 0) {
			uint32_t n = 0;
#pragma unroll
			for (int w = 0; w < 256 / WAVE; w++) n += s_red[w];
			if (n) atomicAdd(a.fine_total + (blockIdx.x & (SCAN_AUX_SLOTS - 1)), (unsigned long long)n);
		}
	}
}

template <bool STAGED>
__device__ __forceinline__ PreOut preprocess_one(const PreprocessArgs& a, const int idx, const float* s_sh, uint32_t* s_cell, const PreIn& in) {
	const PreOut none = { 0u, 0u };
	if (idx == 0) {       // sentinels: the exclusive scans over P + 1 entries leave the totals at [P]
		a.tiles_touched[a.P] = 0;
		if (a.v2) a.fine_touched[a.P] = 0;
	}
	a.radii[idx] = 0;
	a.tiles_touched[idx] = 0;
	if (a.v2) { a.dupinfo[idx] = make_uint4(0u, 0u, 0u, 0u); a.fine_touched[idx] = 0; }      // culled: no cells

	// raw scene geometry: Gaussians idx < Ns read position / log-scale / raw rotation / opacity logit from the raw tensors
	const bool rs = a.sh_src.scene_xyz != nullptr && idx < a.sh_src.Ns;
	const float* pos = rs ? a.sh_src.scene_xyz : a.means3D;
	const float px = STAGED ? in.p[0] : pos[3 * (size_t)idx], py = STAGED ? in.p[1] : pos[3 * (size_t)idx + 1], pz = STAGED ? in.p[2] : pos[3 * (size_t)idx + 2];
	const float* V = a.view; const float* PJ = a.proj;
	// near cull only (auxiliary.h:154)
	const float vz = V[2] * px + V[6] * py + V[10] * pz + V[14];
	if (vz <= 0.2f) return none;
	const float vx = V[0] * px + V[4] * py + V[8] * pz + V[12];
	const float vy = V[1] * px + V[5] * py + V[9] * pz + V[13];
	const float hx = PJ[0] * px + PJ[4] * py + PJ[8] * pz + PJ[12];
	const float hy = PJ[1] * px + PJ[5] * py + PJ[9] * pz + PJ[13];
	const float hw = PJ[3] * px + PJ[7] * py + PJ[11] * pz + PJ[15];
	const float p_w = 1.0f / (hw + 0.0000001f);
	const float projx = hx * p_w, projy = hy * p_w;

	float c3[6];
	float opacity_in = 0.f;            // raw scene row: sigmoid of the logit (below); else read where the reference reads it
	if (a.cov3D_precomp) {
#pragma unroll
		for (int k = 0; k < 6; k++) c3[k] = a.cov3D_precomp[6 * (size_t)idx + k];
	} else {
		if (rs) {
			// (the staged kernel hands scene_activations its prefetched row: index 0 of a one-row "tensor")
			const SceneAct act = STAGED ? scene_activations(in.s, in.q, &in.op, (size_t)0)
			                            : scene_activations(a.sh_src.scene_scaling, a.sh_src.scene_rotation, a.sh_src.scene_opacity, (size_t)idx);
			cov3d_from_values(act.s[0], act.s[1], act.s[2], a.scale_modifier, act.q[0], act.q[1], act.q[2], act.q[3], c3);
			opacity_in = act.op;
		} else if (STAGED) {
			cov3d_from_scale_rot(in.s, a.scale_modifier, in.q, c3);
		} else {
			cov3d_from_scale_rot(a.scales + 3 * (size_t)idx, a.scale_modifier, a.rotations + 4 * (size_t)idx, c3);
		}
		if (a.cov3D) {        // classic pipeline keeps it for the backward; v2 recomputes it there (same function, same rounding)
#pragma unroll
			for (int k = 0; k < 6; k++) a.cov3D[6 * (size_t)idx + k] = c3[k];
		}
	}

	// EWA 2D covariance (forward.cu:74-113)
	float tx = vx, ty = vy; const float tz = vz;
	const float limx = 1.3f * a.tan_fovx, limy = 1.3f * a.tan_fovy;
	const float txtz = tx / tz, tytz = ty / tz;
	tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
	ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
	M3 J = { { { a.focal_x / tz, 0.0f, -(a.focal_x * tx) / (tz * tz) },
	           { 0.0f, a.focal_y / tz, -(a.focal_y * ty) / (tz * tz) },
	           { 0.f, 0.f, 0.f } } };
	M3 Wm = { { { V[0], V[4], V[8] }, { V[1], V[5], V[9] }, { V[2], V[6], V[10] } } };
	M3 T = m3mul(Wm, J);
	M3 Vrk = { { { c3[0], c3[1], c3[2] }, { c3[1], c3[3], c3[4] }, { c3[2], c3[4], c3[5] } } };
	M3 cov = m3mul(m3mul(m3t(T), m3t(Vrk)), T);
	const float cxx = cov.v[0][0] + 0.3f, cxy = cov.v[0][1], cyy = cov.v[1][1] + 0.3f;

	const float det = cxx * cyy - cxy * cxy;
	if (det == 0.0f) return none;
	const float det_inv = 1.f / det;
	const float conx = cyy * det_inv, cony = -cxy * det_inv, conz = cxx * det_inv;
	const float mid = 0.5f * (cxx + cyy);
	const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
	const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
	const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
	const float pix = ndc2pix(projx, a.W), piy = ndc2pix(projy, a.H);
	uint32_t minx, miny, maxx, maxy;
	tile_rect(pix, piy, (int)my_radius, a.gx, a.gy, minx, miny, maxx, maxy);
	if ((maxx - minx) * (maxy - miny) == 0) return none;

	Splat s;
	s.x = pix; s.y = piy; s.ca = conx; s.cb = cony; s.cc = conz; s.opacity = rs ? opacity_in : (STAGED ? in.op : a.opacities[idx]);
	uint8_t clamp_bits = 0;
	if (a.colors_precomp) {
		s.r = a.colors_precomp[3 * (size_t)idx]; s.g = a.colors_precomp[3 * (size_t)idx + 1]; s.b = a.colors_precomp[3 * (size_t)idx + 2];
	} else if (a.shs || a.sh_src.scene_dc) {
		float dx = px - a.campos[0], dy = py - a.campos[1], dz = pz - a.campos[2];
		const float len = sqrtf(dx * dx + dy * dy + dz * dz);
		dx = dx / len; dy = dy / len; dz = dz / len;
		const float* sh; float sh0[3];
		if (a.sh_src.scene_dc) {
			// dc + f_shs(t) || rest, straight from the raw scene/object tensors
			const bool is_obj = idx >= a.sh_src.Ns;
			const size_t m = is_obj ? idx - a.sh_src.Ns : idx;
			if (STAGED) { sh0[0] = in.sh0[0]; sh0[1] = in.sh0[1]; sh0[2] = in.sh0[2]; }
			else { sh0[0] = a.sh0[3 * (size_t)idx]; sh0[1] = a.sh0[3 * (size_t)idx + 1]; sh0[2] = a.sh0[3 * (size_t)idx + 2]; }
			if (STAGED) sh = s_sh + threadIdx.x * SH_ROW_REST - 3;
			else sh = (is_obj ? a.sh_src.obj_rest : a.sh_src.scene_rest) + m * (size_t)(a.M - 1) * 3 - 3;
		} else {
			if (STAGED) sh = s_sh + threadIdx.x * SH_ROW_FULL_LDS;
			else sh = a.shs + (size_t)idx * a.M * 3;
			sh0[0] = sh[0]; sh0[1] = sh[1]; sh0[2] = sh[2];
		}
		float rgb[3];
#pragma unroll
		for (int c = 0; c < 3; c++) {
			const float v = sh_channel(a.D, sh0[c], sh, c, dx, dy, dz);
			if (v < 0.f) clamp_bits |= (uint8_t)(1u << c);
			rgb[c] = fmaxf(v, 0.0f);
		}
		s.r = rgb[0]; s.g = rgb[1]; s.b = rgb[2];
		if (a.ddir && a.D == 3) {
			// nine floats per visible Gaussian in place of the backward's second read of its 45-float row (planes: consecutive lanes, consecutive words)
#pragma unroll
			for (int c = 0; c < 3; c++) {
				float ddx, ddy, ddz;
				sh_channel_ddir(sh, c, dx, dy, dz, ddx, ddy, ddz);
				st_stream(a.ddir + (size_t)(3 * c + 0) * a.P + idx, ddx); st_stream(a.ddir + (size_t)(3 * c + 1) * a.P + idx, ddy); st_stream(a.ddir + (size_t)(3 * c + 2) * a.P + idx, ddz);
			}
		}
	} else {
		s.r = 0.f; s.g = 0.f; s.b = 0.f;
	}
	s.dval = a.inv_depth ? (1.0f / (vz + 0.0000001f)) : vz;
	if (a.flow_points && rs) { s.fx = px; s.fy = py; s.fz = pz; }       // a scene Gaussian does not move: its flow point is its position
	else if (a.flow_points && STAGED) { s.fx = in.f[0]; s.fy = in.f[1]; s.fz = in.f[2]; }
	else if (a.flow_points) { s.fx = a.flow_points[3 * (size_t)idx]; s.fy = a.flow_points[3 * (size_t)idx + 1]; s.fz = a.flow_points[3 * (size_t)idx + 2]; }
	else { s.fx = 0.f; s.fy = 0.f; s.fz = 0.f; }
	s.sem0 = STAGED ? in.sem : ((a.semantic && a.D_S > 0) ? a.semantic[(size_t)idx * a.D_S] : 0.f);
	// alpha = opacity * exp(-0.5 d^T Q d) >= 1/255  <=>  d^T Q d <= tau = 2 ln(255 opacity); +0.02: a 1 % slack on alpha that dominates
	// every fp32 rounding in the per-pixel test (the blend forward's tile test and the rectangle shrink below use it)
	const float tau = 2.f * logf(255.f * s.opacity) + 0.02f;
	s.aux = a.v2 ? tau : vz;
	// "lean" Gaussians (render_v2.hip, eval_pixel<true>): opacity <= 0.99 and a conic that is positive definite with a relative margin of
	// 1e-4 on its determinant.  For these neither `power > 0` (forward.cu:345-346) nor the 0.99 clamp (forward.cu:353) can fire for any pixel
	// offset: the quadratic form is >= 5e-5 (A dx^2 + C dy^2) while an fp32 evaluation of it in any order is off by < 1e-6 of that sum, and
	// exp of a non-positive argument never exceeds 1 -- so the blend kernels skip both tests for them, bit for bit the same result.
	const bool lean = s.opacity <= 0.99f && s.ca > 0.f && s.cc > 0.f && s.cb * s.cb <= 0.9999f * (s.ca * s.cc);
	s.lean = lean ? 1.f : 0.f;
	// one 64-byte line, four 16-byte stores
	float4* dst = reinterpret_cast<float4*>(a.splats + idx);
	dst[0] = make_float4(s.x, s.y, s.ca, s.cb);
	dst[1] = make_float4(s.cc, s.opacity, s.r, s.g);
	dst[2] = make_float4(s.b, s.dval, s.fx, s.fy);
	dst[3] = make_float4(s.fz, s.sem0, s.aux, s.lean);
	a.clamped[idx] = clamp_bits;
	a.radii[idx] = (int)my_radius;
	if (a.gacc) {     // the blend backward accumulates into this 64-B line (rows of culled Gaussians are never read)
		float4* g = reinterpret_cast<float4*>(a.gacc + (size_t)idx * GACC_STRIDE);
		const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
		g[0] = z; g[1] = z; g[2] = z; g[3] = z;
	}
	if (!a.v2) {
		a.tiles_touched[idx] = (maxy - miny) * (maxx - minx);
		return none;
	}
	// ---- v2: shrink the reference rectangle to the tiles on which alpha can reach 1/255.
	// alpha = opacity * exp(-0.5 d^T Q d) >= 1/255  <=>  d^T Q d <= tau = 2 ln(255 opacity); the
	// axis-aligned bound of that ellipse is |dx| <= sqrt(tau * cov_xx), |dy| <= sqrt(tau * cov_yy)
	// (cov = Q^-1).  A 1% slack on alpha (+0.02 on tau) and 1e-3 relative + 0.01 px on the extents
	// dominate every fp32 rounding in the per-pixel test, so no contributing pixel is ever lost.
	uint32_t sminx = 0, sminy = 0, smaxx = 0, smaxy = 0;
	if (tau > 0.f) {
		const float ex = sqrtf(tau * cxx) * 1.001f + 0.01f, ey = sqrtf(tau * cyy) * 1.001f + 0.01f;
		// tile t covers pixel centres [16t, 16t+15]
		const int tminx = (int)ceilf((pix - ex - (float)(TILE_X - 1)) / TILE_X), tmaxx = (int)fminf(floorf((pix + ex) / TILE_X) + 1.f, 1.0e9f);
		const int tminy = (int)ceilf((piy - ey - (float)(TILE_Y - 1)) / TILE_Y), tmaxy = (int)fminf(floorf((piy + ey) / TILE_Y) + 1.f, 1.0e9f);
		// intersect in signed arithmetic (the opacity-aware bound can lie entirely off-screen)
		const int ix0 = max((int)minx, tminx), ix1 = min((int)maxx, tmaxx);
		const int iy0 = max((int)miny, tminy), iy1 = min((int)maxy, tmaxy);
		if (ix1 > ix0 && iy1 > iy0) { sminx = (uint32_t)ix0; smaxx = (uint32_t)ix1; sminy = (uint32_t)iy0; smaxy = (uint32_t)iy1; }
	}
	uint32_t ncell = 0;
	const uint32_t nfine = (smaxx - sminx) * (smaxy - sminy);
	if (nfine) {
		const uint32_t c0x = sminx / a.cell_tiles, c1x = (smaxx - 1) / a.cell_tiles, c0y = sminy / a.cell_tiles, c1y = (smaxy - 1) / a.cell_tiles;
		ncell = (c1x - c0x + 1) * (c1y - c0y + 1);
		if (a.bucket_count) {       // bucket binning: count this Gaussian into every coarse cell it covers
			for (uint32_t y = c0y; y <= c1y; y++)
				for (uint32_t x = c0x; x <= c1x; x++) atomicAdd(s_cell + y * a.cgx + x, 1u);
		}
	}
	a.dupinfo[idx] = make_uint4(sminx | (sminy << 16), smaxx | (smaxy << 16), __float_as_uint(vz), 0u);   // all the binning kernel needs, 16 B
	a.tiles_touched[idx] = ncell;
	a.fine_touched[idx] = nfine;     // scanned on the host side of the pipeline: chunk-pool capacity bound
	PreOut o; o.nfine = nfine; o.zbits = __float_as_uint(vz);
	return o;
}


__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float* __restrict__ means, const float* __restrict__ V, uint8_t* __restrict__ present) {
	const int idx = blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= P) return;
	const float vz = V[2] * means[3 * idx] + V[6] * means[3 * idx + 1] + V[10] * means[3 * idx + 2] + V[14];
	present[idx] = vz <= 0.2f ? 0 : 1;
}

// rasterizer_impl.cu:70-111: one (tile | depth) key + Gaussian index per covered tile.
__global__ void __launch_bounds__(256) duplicate_keys_kernel(int P, const Splat* __restrict__ splats, const uint32_t* __restrict__ offsets,
	const int* __restrict__ radii, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, int gx, int gy) {
	const int idx = blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= P) return;
	const int r = radii[idx];
	if (r > 0) {
		uint32_t off = offsets[idx];       // exclusive scan: start slot of this Gaussian
		const float2 xy = *reinterpret_cast<const float2*>(&splats[idx].x);
		const uint32_t dbits = __float_as_uint(splats[idx].aux);      // classic pipeline: the view-space depth
		uint32_t minx, miny, maxx, maxy;
		tile_rect(xy.x, xy.y, r, gx, gy, minx, miny, maxx, maxy);
		for (uint32_t y = miny; y < maxy; y++)
			for (uint32_t x = minx; x < maxx; x++) {
				uint64_t key = (uint64_t)(y * gx + x);
				key <<= 32; key |= dbits;
				keys[off] = key; vals[off] = (uint32_t)idx; off++;
			}
	}
}

// v2: one (cell | depth) key per covered coarse cell.
__global__ void __launch_bounds__(256) duplicate_cells_kernel(int P, const uint4* __restrict__ dupinfo,
	const uint32_t* __restrict__ offsets, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t cap, int cell_tiles, int cgx,
	uint2* __restrict__ cell_ranges, int ncells, uint32_t* __restrict__ pool_cursor, int mask_shift) {
	const int idx = blockIdx.x * blockDim.x + threadIdx.x;
	// bookkeeping resets for the stages that follow on this stream (tile_ranges, blend forward)
	for (int c = idx; c < ncells; c += gridDim.x * blockDim.x) cell_ranges[c] = make_uint2(0u, 0u);
	if (idx == 0) *pool_cursor = 0u;
	if (idx >= P) return;
	const uint4 d = dupinfo[idx];                  // (rect min, rect max, depth bits, -): one coalesced 16-byte load
	const uint32_t minx = d.x & 0xFFFFu, miny = d.x >> 16, maxx = d.y & 0xFFFFu, maxy = d.y >> 16;
	if (maxx <= minx || maxy <= miny) return;
	uint32_t off = offsets[idx];
	const uint32_t dbits = d.z;
	const uint32_t c0x = minx / cell_tiles, c1x = (maxx - 1) / cell_tiles, c0y = miny / cell_tiles, c1y = (maxy - 1) / cell_tiles;
	for (uint32_t y = c0y; y <= c1y; y++)
		for (uint32_t x = c0x; x <= c1x; x++) {
			uint64_t key = (uint64_t)(y * cgx + x);
			key <<= 32; key |= dbits;
			{
				// The key bits above (cell | depth) are not sorted on but travel with the key: they carry which tile rows and tile
				// columns OF THIS CELL the Gaussian's rectangle covers, so that the blend forward can run the rectangle test on the
				// sorted key stream alone (8 sequential bytes per candidate) and gathers the Splat line only of candidates
				// that pass it.
				const uint32_t ty0 = y * cell_tiles, tx0 = x * cell_tiles;
				const uint32_t r0 = max(miny, ty0) - ty0, r1 = min(maxy, ty0 + cell_tiles) - ty0;      // [r0, r1) within the cell
				const uint32_t q0 = max(minx, tx0) - tx0, q1 = min(maxx, tx0 + cell_tiles) - tx0;
				const uint64_t rows = ((1ull << r1) - 1ull) & ~((1ull << r0) - 1ull), cols = ((1ull << q1) - 1ull) & ~((1ull << q0) - 1ull);
				key |= (rows << mask_shift) | (cols << (mask_shift + cell_tiles));
			}
			if (off < cap) { keys[off] = key; vals[off] = (uint32_t)idx; }    // cap: speculative capacity (the host re-runs on overflow)
			off++;
		}
}

// rasterizer_impl.cu:116-138
__global__ void __launch_bounds__(256) tile_ranges_kernel(int L, const uint32_t* __restrict__ d_L, const uint64_t* __restrict__ keys, uint2* __restrict__ ranges,
	uint32_t id_mask) {
	const int idx = blockIdx.x * blockDim.x + threadIdx.x;
	if (d_L) L = (int)min((uint32_t)L, *d_L);      // device-side count, L is the capacity
	if (idx >= L) return;
	const uint32_t currtile = (uint32_t)(keys[idx] >> 32) & id_mask;      // v2 keys carry coverage masks above the cell id
	if (idx == 0) ranges[currtile].x = 0;
	else {
		const uint32_t prevtile = (uint32_t)(keys[idx - 1] >> 32) & id_mask;
		if (currtile != prevtile) { ranges[prevtile].y = idx; ranges[currtile].x = idx; }
	}
	if (idx == L - 1) ranges[currtile].y = L;
}

} // namespace

int launch_preprocess_fwd(const PreprocessArgs& a, hipStream_t stream) {
	if (a.P == 0) return 0;
	const bool raw = a.sh_src.scene_dc != nullptr;
	const bool staged = a.M == 16 && !a.colors_precomp && (raw || a.shs) && getenv("ADGS_NO_SH_STAGING") == nullptr;
	if (staged) {
		const size_t lds = (size_t)256 * (raw ? SH_ROW_REST : SH_ROW_FULL_LDS) * sizeof(float);
		hipLaunchKernelGGL(preprocess_fwd_kernel<true>, dim3((a.P + 255) / 256), dim3(256), lds, stream, a);
	} else {
		hipLaunchKernelGGL(preprocess_fwd_kernel<false>, dim3((a.P + 255) / 256), dim3(256), 0, stream, a);
	}
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
int launch_mark_visible(int P, const float* means, const float* view, uint8_t* present, hipStream_t stream) {
	if (P == 0) return 0;
	hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, means, view, present);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
int launch_duplicate_keys(int P, const Splat* splats, const uint32_t* offsets, const int* radii, uint64_t* keys, uint32_t* vals,
	int gx, int gy, hipStream_t stream) {
	if (P == 0) return 0;
	hipLaunchKernelGGL(duplicate_keys_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, splats, offsets, radii, keys, vals, gx, gy);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
int launch_duplicate_cells(int P, const uint4* dupinfo, const uint32_t* offsets, uint64_t* keys, uint32_t* vals,
	uint32_t cap, int cell_tiles, int cgx, uint2* cell_ranges, int ncells, uint32_t* pool_cursor, int mask_shift, hipStream_t stream) {
	if (P == 0) return 0;
	hipLaunchKernelGGL(duplicate_cells_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, dupinfo, offsets, keys, vals, cap, cell_tiles, cgx,
		cell_ranges, ncells, pool_cursor, mask_shift);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
int launch_tile_ranges(int L, const uint32_t* d_L, const uint64_t* keys, uint2* ranges, uint32_t id_mask, hipStream_t stream) {
	if (L == 0) return 0;
	hipLaunchKernelGGL(tile_ranges_kernel, dim3((L + 255) / 256), dim3(256), 0, stream, L, d_L, keys, ranges, id_mask);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

} // namespace adgs
