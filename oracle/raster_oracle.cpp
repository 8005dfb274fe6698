// =============================================================================
// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
//
// A plain C++ restatement of the reference AD-GS rasterizer
// (submodules/depth-diff-gaussian-rasterization, "RAST/" below).  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
//
// PARITY STATUS: "parity unpinned" by the reference itself -- the reference is
// CUDA-only (no nvcc / no GPU in the authoring container), ships no tests and no
// golden vectors.  The pins this repo adds are in
// tests/test_oracle_*.py: analytic known-answer tests, a torch-autograd
// cross-check of the backward, and central finite differences.
//
// Every function cites the reference file:line it restates.  glm is column
// major; the small M3 helper below reproduces glm's operator* evaluation order
// (RAST/third_party/glm/glm/detail/type_mat3x3.inl:486-520) so that the fp32
// build follows the reference's operation order (without FMA contraction).
//
// Built twice from this file: real=float ("f32", the comparator) and
// real=double ("f64", the numerical arbiter for continuous quantities).
// =============================================================================
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <numeric>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// RAST/cuda_rasterizer/config.h:15-19
constexpr int NUM_CHANNELS = 3;
constexpr int BLOCK_X = 16;
constexpr int BLOCK_Y = 16;
constexpr int SEMANTIC_CHANNELS = 32;
constexpr int FLOW_CHANNELS = 3;

template <typename real> struct V3 { real x, y, z; };
template <typename real> struct V4 { real x, y, z, w; };

// Column-major 3x3, value[col][row], same evaluation order as glm.
template <typename real> struct M3 {
	real v[3][3];
	// glm::mat3(a,b,c, d,e,f, g,h,i) fills COLUMNS: col0=(a,b,c) ...
	static M3 make(real a, real b, real c, real d, real e, real f, real g, real h, real i) {
		M3 m; m.v[0][0] = a; m.v[0][1] = b; m.v[0][2] = c;
		m.v[1][0] = d; m.v[1][1] = e; m.v[1][2] = f;
		m.v[2][0] = g; m.v[2][1] = h; m.v[2][2] = i; return m;
	}
};
template <typename real> M3<real> mul(const M3<real>& a, const M3<real>& b) {
	// type_mat3x3.inl:486-520
	M3<real> r;
	for (int c = 0; c < 3; c++)
		for (int rr = 0; rr < 3; rr++)
			r.v[c][rr] = a.v[0][rr] * b.v[c][0] + a.v[1][rr] * b.v[c][1] + a.v[2][rr] * b.v[c][2];
	return r;
}
template <typename real> M3<real> transpose(const M3<real>& a) {
	M3<real> r;
	for (int c = 0; c < 3; c++) for (int rr = 0; rr < 3; rr++) r.v[c][rr] = a.v[rr][c];
	return r;
}

// RAST/cuda_rasterizer/auxiliary.h:22-39
template <typename real> struct SH {
	static constexpr real C0 = (real)0.28209479177387814;
	static constexpr real C1 = (real)0.4886025119029199;
	static constexpr real C2[5] = { (real)1.0925484305920792, (real)-1.0925484305920792, (real)0.31539156525252005,
		(real)-1.0925484305920792, (real)0.5462742152960396 };
	static constexpr real C3[7] = { (real)-0.5900435899266435, (real)2.890611442640554, (real)-0.4570457994644658,
		(real)0.3731763325901154, (real)-0.4570457994644658, (real)1.445305721320277, (real)-0.5900435899266435 };
};
// The reference constants are float literals (f suffix); for the f64 build we
// still want the *same* constants, so round through float.
template <typename real> inline real fc(float x) { return (real)x; }

// auxiliary.h:41-44.  NOTE the reference computes this in double (1.0 / 0.5
// literals) and rounds to float on return.
template <typename real> inline real ndc2Pix(real v, int S) {
	return (real)((((double)v + 1.0) * S - 1.0) * 0.5);
}

// auxiliary.h:46-56
template <typename real>
inline void getRect(real px, real py, int max_radius, uint32_t& minx, uint32_t& miny, uint32_t& maxx, uint32_t& maxy, int gx, int gy) {
	minx = (uint32_t)std::min(gx, std::max(0, (int)((px - max_radius) / BLOCK_X)));
	miny = (uint32_t)std::min(gy, std::max(0, (int)((py - max_radius) / BLOCK_Y)));
	maxx = (uint32_t)std::min(gx, std::max(0, (int)((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
	maxy = (uint32_t)std::min(gy, std::max(0, (int)((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

// auxiliary.h:58-77
template <typename real> inline V3<real> transformPoint4x3(const V3<real>& p, const real* m) {
	return { m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
			 m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
			 m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14] };
}
template <typename real> inline V4<real> transformPoint4x4(const V3<real>& p, const real* m) {
	return { m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
			 m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
			 m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
			 m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15] };
}
// auxiliary.h:89-97
template <typename real> inline V3<real> transformVec4x3Transpose(const V3<real>& p, const real* m) {
	return { m[0] * p.x + m[1] * p.y + m[2] * p.z,
			 m[4] * p.x + m[5] * p.y + m[6] * p.z,
			 m[8] * p.x + m[9] * p.y + m[10] * p.z };
}
// auxiliary.h:107-117
template <typename real> inline V3<real> dnormvdv(V3<real> v, V3<real> dv) {
	real sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
	real invsum32 = (real)1.0 / std::sqrt(sum2 * sum2 * sum2);
	V3<real> r;
	r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
	r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
	r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
	return r;
}

// auxiliary.h:139-164 (prefiltered trap not reproduced: returns false)
template <typename real>
inline bool in_frustum(int idx, const real* orig_points, const real* viewmatrix, const real* projmatrix, V3<real>& p_view) {
	V3<real> p_orig = { orig_points[3 * idx], orig_points[3 * idx + 1], orig_points[3 * idx + 2] };
	p_view = transformPoint4x3(p_orig, viewmatrix);
	if (p_view.z <= fc<real>(0.2f)) return false;
	return true;
}

// rasterizer_impl.cu:35-50
inline uint32_t getHigherMsb(uint32_t n) {
	uint32_t msb = sizeof(n) * 4;
	uint32_t step = msb;
	while (step > 1) {
		step /= 2;
		if (n >> msb) msb += step; else msb -= step;
	}
	if (n >> msb) msb++;
	return msb;
}

template <typename real>
struct Oracle {
	// ---- persistent state between forward and backward (GeometryState /
	// BinningState / ImageState of rasterizer_impl.h:29-64) ----
	int P = 0, W = 0, H = 0, gx = 0, gy = 0, R = 0;
	std::vector<real> depths, means2D, cov3D, conic_opacity, rgb;
	std::vector<uint8_t> clamped;
	std::vector<int> radii;
	std::vector<uint32_t> tiles_touched, point_offsets;
	std::vector<uint64_t> keys_unsorted, keys;
	std::vector<uint32_t> list_unsorted, point_list;
	std::vector<uint32_t> ranges;   // T*2
	std::vector<uint32_t> n_contrib; // H*W
	std::vector<real> final_T;       // holds 1-T like the reference's img_opacity
	// converted copies of the matrices
	real view[16], proj[16], campos[3];
	// cached inputs converted to `real`
	std::vector<real> means3D, shs, colors_precomp, flow_points, semantic, opacities, scales, rotations, cov3D_precomp, bg;
	bool has_shs = false, has_colors = false, has_flow = false, has_sem = false, has_scales = false, has_cov = false;
	int D = 0, M = 0, D_S = 0;
	real scale_modifier = 1, tan_fovx = 1, tan_fovy = 1, focal_x = 1, focal_y = 1;
	bool inv_depth = false;

	static void cvt(std::vector<real>& dst, const float* src, size_t n, bool& has) {
		has = (src != nullptr);
		dst.resize(has ? n : 0);
		for (size_t i = 0; i < dst.size(); i++) dst[i] = (real)src[i];
	}

	// forward.cu:20-71
	V3<real> computeColorFromSH(int idx, int deg, int max_coeffs) {
		const real* m = means3D.data();
		V3<real> pos = { m[3 * idx], m[3 * idx + 1], m[3 * idx + 2] };
		V3<real> dir = { pos.x - campos[0], pos.y - campos[1], pos.z - campos[2] };
		real len = std::sqrt(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
		dir = { dir.x / len, dir.y / len, dir.z / len };
		const real* sh = shs.data() + (size_t)idx * max_coeffs * 3;
		auto S = [&](int k, int c) { return sh[k * 3 + c]; };
		real res[3];
		real x = dir.x, y = dir.y, z = dir.z;
		for (int c = 0; c < 3; c++) {
			real result = fc<real>(0.28209479177387814f) * S(0, c);
			if (deg > 0) {
				const real C1 = fc<real>(0.4886025119029199f);
				result = result - C1 * y * S(1, c) + C1 * z * S(2, c) - C1 * x * S(3, c);
				if (deg > 1) {
					real xx = x * x, yy = y * y, zz = z * z;
					real xy = x * y, yz = y * z, xz = x * z;
					const real C20 = fc<real>(1.0925484305920792f), C21 = fc<real>(-1.0925484305920792f),
						C22 = fc<real>(0.31539156525252005f), C23 = fc<real>(-1.0925484305920792f), C24 = fc<real>(0.5462742152960396f);
					result = result +
						C20 * xy * S(4, c) +
						C21 * yz * S(5, c) +
						C22 * ((real)2.0 * zz - xx - yy) * S(6, c) +
						C23 * xz * S(7, c) +
						C24 * (xx - yy) * S(8, c);
					if (deg > 2) {
						const real C30 = fc<real>(-0.5900435899266435f), C31 = fc<real>(2.890611442640554f), C32 = fc<real>(-0.4570457994644658f),
							C33 = fc<real>(0.3731763325901154f), C34 = fc<real>(-0.4570457994644658f), C35 = fc<real>(1.445305721320277f), C36 = fc<real>(-0.5900435899266435f);
						result = result +
							C30 * y * ((real)3.0 * xx - yy) * S(9, c) +
							C31 * xy * z * S(10, c) +
							C32 * y * ((real)4.0 * zz - xx - yy) * S(11, c) +
							C33 * z * ((real)2.0 * zz - (real)3.0 * xx - (real)3.0 * yy) * S(12, c) +
							C34 * x * ((real)4.0 * zz - xx - yy) * S(13, c) +
							C35 * z * (xx - yy) * S(14, c) +
							C36 * x * (xx - (real)3.0 * yy) * S(15, c);
					}
				}
			}
			result += (real)0.5;
			clamped[3 * idx + c] = (result < 0);
			res[c] = std::max(result, (real)0.0);
		}
		return { res[0], res[1], res[2] };
	}

	// forward.cu:74-113
	V3<real> computeCov2D(const V3<real>& mean, const real* c3) {
		V3<real> t = transformPoint4x3(mean, view);
		const real limx = fc<real>(1.3f) * tan_fovx;
		const real limy = fc<real>(1.3f) * tan_fovy;
		const real txtz = t.x / t.z;
		const real tytz = t.y / t.z;
		t.x = std::min(limx, std::max(-limx, txtz)) * t.z;
		t.y = std::min(limy, std::max(-limy, tytz)) * t.z;
		M3<real> J = M3<real>::make(
			focal_x / t.z, 0, -(focal_x * t.x) / (t.z * t.z),
			0, focal_y / t.z, -(focal_y * t.y) / (t.z * t.z),
			0, 0, 0);
		M3<real> Wm = M3<real>::make(
			view[0], view[4], view[8],
			view[1], view[5], view[9],
			view[2], view[6], view[10]);
		M3<real> T = mul(Wm, J);
		M3<real> Vrk = M3<real>::make(
			c3[0], c3[1], c3[2],
			c3[1], c3[3], c3[4],
			c3[2], c3[4], c3[5]);
		M3<real> cov = mul(mul(transpose(T), transpose(Vrk)), T);
		cov.v[0][0] += fc<real>(0.3f);
		cov.v[1][1] += fc<real>(0.3f);
		return { cov.v[0][0], cov.v[0][1], cov.v[1][1] };
	}

	// forward.cu:118-152
	void computeCov3D(const real* scale, real mod, const real* rot, real* out) {
		M3<real> S = M3<real>::make(1, 0, 0, 0, 1, 0, 0, 0, 1);
		S.v[0][0] = mod * scale[0];
		S.v[1][1] = mod * scale[1];
		S.v[2][2] = mod * scale[2];
		real r = rot[0], x = rot[1], y = rot[2], z = rot[3]; // NOT normalised (forward.cu:127)
		M3<real> Rm = M3<real>::make(
			(real)1 - (real)2 * (y * y + z * z), (real)2 * (x * y - r * z), (real)2 * (x * z + r * y),
			(real)2 * (x * y + r * z), (real)1 - (real)2 * (x * x + z * z), (real)2 * (y * z - r * x),
			(real)2 * (x * z - r * y), (real)2 * (y * z + r * x), (real)1 - (real)2 * (x * x + y * y));
		M3<real> Mm = mul(S, Rm);
		M3<real> Sigma = mul(transpose(Mm), Mm);
		out[0] = Sigma.v[0][0]; out[1] = Sigma.v[0][1]; out[2] = Sigma.v[0][2];
		out[3] = Sigma.v[1][1]; out[4] = Sigma.v[1][2]; out[5] = Sigma.v[2][2];
	}

	// forward.cu:155-256 (preprocessCUDA) for one Gaussian
	void preprocess_one(int idx) {
		radii[idx] = 0;
		tiles_touched[idx] = 0;
		V3<real> p_view;
		if (!in_frustum(idx, means3D.data(), view, proj, p_view)) return;
		V3<real> p_orig = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
		V4<real> p_hom = transformPoint4x4(p_orig, proj);
		real p_w = (real)1.0 / (p_hom.w + fc<real>(0.0000001f));
		V3<real> p_proj = { p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w };
		const real* c3;
		if (has_cov) c3 = cov3D_precomp.data() + (size_t)idx * 6;
		else {
			computeCov3D(scales.data() + 3 * (size_t)idx, scale_modifier, rotations.data() + 4 * (size_t)idx, cov3D.data() + 6 * (size_t)idx);
			c3 = cov3D.data() + 6 * (size_t)idx;
		}
		V3<real> cov = computeCov2D(p_orig, c3);
		real det = (cov.x * cov.z - cov.y * cov.y);
		if (det == (real)0) return;
		real det_inv = (real)1 / det;
		V3<real> conic = { cov.z * det_inv, -cov.y * det_inv, cov.x * det_inv };
		real mid = (real)0.5 * (cov.x + cov.z);
		real lambda1 = mid + std::sqrt(std::max(fc<real>(0.1f), mid * mid - det));
		real lambda2 = mid - std::sqrt(std::max(fc<real>(0.1f), mid * mid - det));
		real my_radius = std::ceil((real)3 * std::sqrt(std::max(lambda1, lambda2)));
		real pix = ndc2Pix(p_proj.x, W), piy = ndc2Pix(p_proj.y, H);
		uint32_t minx, miny, maxx, maxy;
		getRect(pix, piy, (int)my_radius, minx, miny, maxx, maxy, gx, gy);
		if ((maxx - minx) * (maxy - miny) == 0) return;
		if (!has_colors && has_shs) {
			V3<real> c = computeColorFromSH(idx, D, M);
			rgb[3 * idx + 0] = c.x; rgb[3 * idx + 1] = c.y; rgb[3 * idx + 2] = c.z;
		}
		depths[idx] = p_view.z;
		radii[idx] = (int)my_radius;
		means2D[2 * idx] = pix; means2D[2 * idx + 1] = piy;
		conic_opacity[4 * idx + 0] = conic.x; conic_opacity[4 * idx + 1] = conic.y;
		conic_opacity[4 * idx + 2] = conic.z; conic_opacity[4 * idx + 3] = opacities[idx];
		tiles_touched[idx] = (maxy - miny) * (maxx - minx);
	}

	const real* feature_ptr() const {
		return has_colors ? colors_precomp.data() : (has_shs ? rgb.data() : nullptr);
	}

	// rasterizer_impl.cu:198-352
	int forward(int P_, int D_, int M_, int D_S_, const float* bg_, int W_, int H_,
		const float* means3D_, const float* shs_, const float* colors_, const float* flow_, const float* sem_,
		const float* opac_, const float* scales_, float scale_mod_, const float* rots_, const float* cov_,
		const float* view_, const float* proj_, const float* campos_, float tfx, float tfy,
		real* out_color, real* out_depth, real* img_opacity, real* img_flow, real* img_semantic, bool inv_depth_, int* radii_out) {
		P = P_; D = D_; M = M_; D_S = D_S_; W = W_; H = H_;
		inv_depth = inv_depth_;
		scale_modifier = (real)scale_mod_; tan_fovx = (real)tfx; tan_fovy = (real)tfy;
		// rasterizer_impl.cu:229-230 (float arithmetic in the reference)
		focal_y = (real)((float)H / (2.0f * tfy));
		focal_x = (real)((float)W / (2.0f * tfx));
		if (sizeof(real) == 8) { focal_y = (real)H / ((real)2 * (real)tfy); focal_x = (real)W / ((real)2 * (real)tfx); }
		gx = (W + BLOCK_X - 1) / BLOCK_X; gy = (H + BLOCK_Y - 1) / BLOCK_Y;
		for (int i = 0; i < 16; i++) { view[i] = (real)view_[i]; proj[i] = (real)proj_[i]; }
		for (int i = 0; i < 3; i++) campos[i] = (real)campos_[i];
		bool dummy;
		cvt(means3D, means3D_, (size_t)P * 3, dummy);
		cvt(shs, shs_, (size_t)P * M * 3, has_shs);
		cvt(colors_precomp, colors_, (size_t)P * 3, has_colors);
		cvt(flow_points, flow_, (size_t)P * 3, has_flow);
		cvt(semantic, sem_, (size_t)P * D_S, has_sem);
		cvt(opacities, opac_, (size_t)P, dummy);
		cvt(scales, scales_, (size_t)P * 3, has_scales);
		cvt(rotations, rots_, (size_t)P * 4, dummy);
		cvt(cov3D_precomp, cov_, (size_t)P * 6, has_cov);
		cvt(bg, bg_, 3, dummy);

		depths.assign(P, 0); means2D.assign((size_t)P * 2, 0); cov3D.assign((size_t)P * 6, 0);
		conic_opacity.assign((size_t)P * 4, 0); rgb.assign((size_t)P * 3, 0); clamped.assign((size_t)P * 3, 0);
		radii.assign(P, 0); tiles_touched.assign(P, 0); point_offsets.assign(P, 0);

		// K1
#pragma omp parallel for schedule(static)
		for (int i = 0; i < P; i++) preprocess_one(i);
		// K2 inclusive scan (rasterizer_impl.cu:284)
		uint32_t acc = 0;
		for (int i = 0; i < P; i++) { acc += tiles_touched[i]; point_offsets[i] = acc; }
		R = P > 0 ? (int)point_offsets[P - 1] : 0;
		// K3 duplicateWithKeys (rasterizer_impl.cu:70-111)
		keys_unsorted.assign(R, 0); list_unsorted.assign(R, 0);
#pragma omp parallel for schedule(static)
		for (int idx = 0; idx < P; idx++) {
			if (radii[idx] > 0) {
				uint32_t off = (idx == 0) ? 0 : point_offsets[idx - 1];
				uint32_t minx, miny, maxx, maxy;
				getRect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], minx, miny, maxx, maxy, gx, gy);
				float depth_f = (float)depths[idx];
				uint32_t dbits; std::memcpy(&dbits, &depth_f, 4);
				for (uint32_t y = miny; y < maxy; y++)
					for (uint32_t x = minx; x < maxx; x++) {
						uint64_t key = (uint64_t)(y * gx + x);
						key <<= 32; key |= dbits;
						keys_unsorted[off] = key; list_unsorted[off] = idx; off++;
					}
			}
		}
		// K4 stable sort on bits [0, 32+bit) (rasterizer_impl.cu:307-315)
		int bit = (int)getHigherMsb((uint32_t)(gx * gy));
		uint64_t mask = (32 + bit >= 64) ? ~0ull : ((1ull << (32 + bit)) - 1);
		std::vector<uint32_t> perm(R);
		std::iota(perm.begin(), perm.end(), 0u);
		std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) {
			return (keys_unsorted[a] & mask) < (keys_unsorted[b] & mask); });
		keys.resize(R); point_list.resize(R);
		for (int i = 0; i < R; i++) { keys[i] = keys_unsorted[perm[i]]; point_list[i] = list_unsorted[perm[i]]; }
		// K5 identifyTileRanges (rasterizer_impl.cu:116-138, memset :317)
		ranges.assign((size_t)gx * gy * 2, 0);
		for (int idx = 0; idx < R; idx++) {
			uint32_t currtile = (uint32_t)(keys[idx] >> 32);
			if (idx == 0) ranges[2 * currtile] = 0;
			else {
				uint32_t prevtile = (uint32_t)(keys[idx - 1] >> 32);
				if (currtile != prevtile) { ranges[2 * prevtile + 1] = idx; ranges[2 * currtile] = idx; }
			}
			if (idx == R - 1) ranges[2 * currtile + 1] = R;
		}
		// K6 render (forward.cu:261-402)
		n_contrib.assign((size_t)W * H, 0); final_T.assign((size_t)W * H, 0);
		const real* features = feature_ptr();
		const size_t HW = (size_t)H * W;
#pragma omp parallel for schedule(dynamic, 1)
		for (int tile = 0; tile < gx * gy; tile++) {
			int tx = tile % gx, ty = tile / gx;
			uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
			for (int ly = 0; ly < BLOCK_Y; ly++) for (int lx = 0; lx < BLOCK_X; lx++) {
				int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
				if (!(px < W && py < H)) continue;
				size_t pix_id = (size_t)W * py + px;
				real pixfx = (real)px, pixfy = (real)py;
				real T = 1;
				uint32_t contributor = 0, last_contributor = 0;
				real C[NUM_CHANNELS] = { 0 }, Dp = 0, F[FLOW_CHANNELS] = { 0 }, S[SEMANTIC_CHANNELS] = { 0 };
				for (uint32_t k = r0; k < r1; k++) {
					contributor++;
					uint32_t g = point_list[k];
					real dx = means2D[2 * g] - pixfx, dy = means2D[2 * g + 1] - pixfy;
					const real* co = &conic_opacity[4 * (size_t)g];
					real power = (real)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
					if (power > 0) continue;
					real alpha = std::min(fc<real>(0.99f), co[3] * std::exp(power));
					if (alpha < fc<real>(1.0f / 255.0f)) continue;
					real test_T = T * (1 - alpha);
					if (test_T < fc<real>(0.0001f)) break; // done = true
					if (features) for (int ch = 0; ch < NUM_CHANNELS; ch++) C[ch] += features[(size_t)g * NUM_CHANNELS + ch] * alpha * T;
					if (has_flow) for (int ch = 0; ch < FLOW_CHANNELS; ch++) F[ch] += flow_points[(size_t)g * FLOW_CHANNELS + ch] * alpha * T;
					if (has_sem) for (int ch = 0; ch < D_S; ch++) S[ch] += semantic[(size_t)g * D_S + ch] * alpha * T;
					if (inv_depth) Dp += ((real)1.0 / (depths[g] + fc<real>(0.0000001f))) * alpha * T;
					else Dp += depths[g] * alpha * T;
					T = test_T;
					last_contributor = contributor;
				}
				final_T[pix_id] = (real)(1.0 - (double)T);
				img_opacity[pix_id] = final_T[pix_id];
				n_contrib[pix_id] = last_contributor;
				if (features) for (int ch = 0; ch < NUM_CHANNELS; ch++) out_color[ch * HW + pix_id] = C[ch] + T * bg[ch];
				if (has_flow) for (int ch = 0; ch < FLOW_CHANNELS; ch++) img_flow[ch * HW + pix_id] = F[ch];
				if (has_sem) for (int ch = 0; ch < D_S; ch++) img_semantic[ch * HW + pix_id] = S[ch];
				out_depth[pix_id] = Dp;
			}
		}
		if (radii_out) std::memcpy(radii_out, radii.data(), sizeof(int) * P);
		return R;
	}

	// ------------------------------------------------------------------ gate margins (test infrastructure of the test infrastructure)
	// forward.cu:345-361 has three hard gates on computed values: `power > 0`, `alpha < 1/255`, `T (1 - alpha) < 0.0001`.  Two correct
	// float32 evaluations of the same frame can take a gate differently when the gated value lies within their rounding error of the
	// threshold, and the pixel (and every Gaussian that pixel feeds a gradient to) then differs by far more than 1e-4.  This pass
	// re-walks every pixel's list and reports how close each pixel came to a gate, in units of the rounding error of a float32
	// evaluation, so that a parity test can tell an EXPLAINED deviation (margin <= a few float32 ulps) from a wrong result:
	//   power gate:  |power| / S                              S = |0.5 A dx^2| + |0.5 C dy^2| + |B dx dy|  (the terms' magnitudes)
	//   alpha gate:  |op exp(power) - 1/255| / (1/255 (1 + S))    (an absolute error e in power is a relative error e in alpha)
	//   T gate:      |T (1 - alpha) - 1e-4| / (1e-4 (1 + E))      E = sum over the factors of T so far (this one included) of
	//                1 + (1 + S) alpha / (1 - alpha): the product's roundings plus what alpha's relative error does to 1 - alpha
	//                (an alpha clamped to 0.99 is exact)
	// pix_margin[pixel] = the minimum over its walk; gauss_margin[g] = the minimum pix_margin over the pixels in whose walk g takes part
	// (passes the power and alpha gates, or misses one of them by less than 1e-3 relative) -- the walk is continued past the T stop
	// for this purpose, because a flipped stop lets the Gaussians behind it contribute.
	void gate_margins(real* pix_margin /*H*W*/, real* gauss_margin /*P*/) {
		const real INF = (real)1e30;
		for (size_t i = 0; i < (size_t)W * H; i++) pix_margin[i] = INF;
		std::vector<std::vector<real>> gm_thread;
		int nthreads = 1;
#ifdef _OPENMP
		nthreads = omp_get_max_threads();
#endif
		gm_thread.assign(nthreads, std::vector<real>());
#pragma omp parallel
		{
			int tid = 0;
#ifdef _OPENMP
			tid = omp_get_thread_num();
#endif
			std::vector<real>& gm = gm_thread[tid];
			gm.assign((size_t)P, INF);
			std::vector<uint32_t> walk;
#pragma omp for schedule(dynamic, 1)
			for (int tile = 0; tile < gx * gy; tile++) {
				int tx = tile % gx, ty = tile / gx;
				uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
				for (int ly = 0; ly < BLOCK_Y; ly++) for (int lx = 0; lx < BLOCK_X; lx++) {
					int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
					if (!(px < W && py < H)) continue;
					size_t pix_id = (size_t)W * py + px;
					real pixfx = (real)px, pixfy = (real)py;
					real T = 1, m = INF, E = 0;
					bool done = false;
					walk.clear();
					for (uint32_t k = r0; k < r1; k++) {
						uint32_t g = point_list[k];
						real dx = means2D[2 * g] - pixfx, dy = means2D[2 * g + 1] - pixfy;
						const real* co = &conic_opacity[4 * (size_t)g];
						real ta = (real)0.5 * co[0] * dx * dx, tc = (real)0.5 * co[2] * dy * dy, tb = co[1] * dx * dy;
						real S = std::abs(ta) + std::abs(tc) + std::abs(tb);
						real power = -(ta + tc) - tb;
						real araw = co[3] * std::exp(std::min(power, (real)0));
						const real amin = fc<real>(1.0f / 255.0f);
						real m_alpha = std::abs(araw - amin) / (amin * (1 + S));
						real m_pow = S > 0 ? std::abs(power) / S : INF;
						// the power gate only matters where alpha would pass
						bool near_alpha = m_alpha < (real)1e-3, near_pow = m_pow < (real)1e-3 && araw >= amin * (real)0.999;
						bool passes = !(power > 0) && !(araw < amin);
						if (!done) {
							if (near_alpha) m = std::min(m, m_alpha);
							if (near_pow) m = std::min(m, m_pow);
						}
						if (!(passes || near_alpha || near_pow)) continue;
						walk.push_back(g);
						if (!passes || done) continue;
						real alpha = std::min(fc<real>(0.99f), araw);
						real test_T = T * (1 - alpha);
						const real tstop = fc<real>(0.0001f);
						E += 1 + (araw < fc<real>(0.99f) ? (1 + S) * alpha / (1 - alpha) : (real)0);
						real m_T = std::abs(test_T - tstop) / (tstop * (1 + E));
						m = std::min(m, m_T);
						if (test_T < tstop) { done = true; continue; }
						T = test_T;
					}
					pix_margin[pix_id] = m;
					for (uint32_t g : walk) gm[g] = std::min(gm[g], m);
				}
			}
		}
		for (int g = 0; g < P; g++) {
			real v = INF;
			for (auto& gm : gm_thread) if (!gm.empty()) v = std::min(v, gm[g]);
			gauss_margin[g] = v;
		}
	}

	// Conditioning of the blend per pixel (test infrastructure, with gate_margins): sum over the contributing entries of
	// alpha_k T_k (1 + S_k), S as above.  power = -(0.5 A dx^2 + 0.5 C dy^2) - B dx dy is a sum of terms of magnitude S: a float32
	// evaluation is off by ~u S absolutely (u = 6e-8), which is a RELATIVE error u S of alpha = op exp(power) -- for needle-thin or
	// image-filling Gaussians S reaches 1e3 .. 1e4 and two correct float32 evaluations (another operation order, exp2 of a pre-scaled
	// conic) differ by 1e-4 of alpha with no gate involved.  First order, a unit-scale blended output moves by at most
	// sum_k alpha_k T_k (relative error of alpha_k) (1 + what the entries behind lose), i.e. by ~2 u cond.
	void pixel_conditioning(real* cond /*H*W*/) {
#pragma omp parallel for schedule(dynamic, 1)
		for (int tile = 0; tile < gx * gy; tile++) {
			int tx = tile % gx, ty = tile / gx;
			uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
			for (int ly = 0; ly < BLOCK_Y; ly++) for (int lx = 0; lx < BLOCK_X; lx++) {
				int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
				if (!(px < W && py < H)) continue;
				real pixfx = (real)px, pixfy = (real)py, T = 1, c = 0;
				for (uint32_t k = r0; k < r1; k++) {
					uint32_t g = point_list[k];
					real dx = means2D[2 * g] - pixfx, dy = means2D[2 * g + 1] - pixfy;
					const real* co = &conic_opacity[4 * (size_t)g];
					real ta = (real)0.5 * co[0] * dx * dx, tc = (real)0.5 * co[2] * dy * dy, tb = co[1] * dx * dy;
					real S = std::abs(ta) + std::abs(tc) + std::abs(tb);
					real power = -(ta + tc) - tb;
					if (power > 0) continue;
					real alpha = std::min(fc<real>(0.99f), co[3] * std::exp(power));
					if (alpha < fc<real>(1.0f / 255.0f)) continue;
					real test_T = T * (1 - alpha);
					if (test_T < fc<real>(0.0001f)) break;
					c += alpha * T * (1 + S);
					T = test_T;
				}
				cond[(size_t)W * py + px] = c;
			}
		}
	}

	// ------------------------------------------------------------------ backward
	// backward.cu:417-646 (renderCUDA).  The reference scatters with fp32
	// atomicAdd in an unspecified order; this restatement accumulates every
	// per-Gaussian sum in double and rounds once (documented deviation: it is the
	// order-independent value every atomic ordering approximates).
	void backward(const float* dL_dpix_, const float* dL_dpix_depth_, const float* dL_dpix_flow_, const float* dL_dpix_sem_,
		const float* grad_img_opacity_,
		real* dL_dmean2D /*P*3*/, real* dL_dconic /*P*4*/, real* dL_dopacity /*P*/, real* dL_dcolor /*P*3*/, real* dL_ddepth /*P*/,
		real* dL_dmean3D /*P*3*/, real* dL_dcov3D /*P*6*/, real* dL_dsh /*P*M*3*/, real* dL_dscale /*P*3*/, real* dL_drot /*P*4*/,
		real* dL_dflow /*P*3*/, real* dL_dsemantic /*P*D_S*/) {
		const size_t HW = (size_t)H * W;
		const real* colors = feature_ptr();
		const bool do_col = dL_dpix_ && colors;
		const bool do_flow = dL_dpix_flow_ && has_flow;
		const bool do_sem = dL_dpix_sem_ && has_sem;
		const bool do_depth = dL_dpix_depth_ != nullptr;
		const bool do_op = grad_img_opacity_ != nullptr;
		std::vector<double> a_mean2D((size_t)P * 2, 0), a_conic((size_t)P * 3, 0), a_op(P, 0), a_col((size_t)P * 3, 0), a_dep(P, 0),
			a_flow((size_t)P * 3, 0), a_sem((size_t)P * std::max(D_S, 1), 0);
		const real ddelx_dx = (real)(0.5 * W);
		const real ddely_dy = (real)(0.5 * H);
		auto acc = [](double& a, double v) {
#pragma omp atomic
			a += v;
		};
#pragma omp parallel for schedule(dynamic, 1)
		for (int tile = 0; tile < gx * gy; tile++) {
			int tx = tile % gx, ty = tile / gx;
			uint32_t r0 = ranges[2 * tile];
			for (int ly = 0; ly < BLOCK_Y; ly++) for (int lx = 0; lx < BLOCK_X; lx++) {
				int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
				if (!(px < W && py < H)) continue;
				size_t pix_id = (size_t)W * py + px;
				real pixfx = (real)px, pixfy = (real)py;
				const real T_final = (real)(1.0 - (double)final_T[pix_id]);
				real T = T_final;
				const int last_contributor = (int)n_contrib[pix_id];
				real accum_rec[NUM_CHANNELS] = { 0 }, accum_flow_rec[FLOW_CHANNELS] = { 0 }, accum_semantic_rec[SEMANTIC_CHANNELS] = { 0 };
				real dL_dpixel[NUM_CHANNELS] = { 0 }, dL_dpixel_flow[FLOW_CHANNELS] = { 0 }, dL_dpixel_semantic[SEMANTIC_CHANNELS] = { 0 };
				real dL_dpixel_depth = 0, dL_dpixel_opacity = 0, accum_depth_rec = 0;
				if (do_col) for (int i = 0; i < NUM_CHANNELS; i++) dL_dpixel[i] = (real)dL_dpix_[i * HW + pix_id];
				if (do_flow) for (int i = 0; i < FLOW_CHANNELS; i++) dL_dpixel_flow[i] = (real)dL_dpix_flow_[i * HW + pix_id];
				if (do_sem) for (int i = 0; i < D_S; i++) dL_dpixel_semantic[i] = (real)dL_dpix_sem_[i * HW + pix_id];
				if (do_depth) dL_dpixel_depth = (real)dL_dpix_depth_[pix_id];
				if (do_op) dL_dpixel_opacity = (real)grad_img_opacity_[pix_id];
				real last_alpha = 0, last_color[NUM_CHANNELS] = { 0 }, last_depth = 0, last_flow[FLOW_CHANNELS] = { 0 }, last_semantic[SEMANTIC_CHANNELS] = { 0 };
				for (int j = last_contributor - 1; j >= 0; j--) {
					const uint32_t g = point_list[r0 + j];
					const real dx = means2D[2 * g] - pixfx, dy = means2D[2 * g + 1] - pixfy;
					const real* co = &conic_opacity[4 * (size_t)g];
					const real power = (real)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
					if (power > 0) continue;
					const real G = std::exp(power);
					const real alpha = std::min(fc<real>(0.99f), co[3] * G);
					if (alpha < fc<real>(1.0f / 255.0f)) continue;
					T = T / ((real)1 - alpha);
					const real dchannel_dcolor = alpha * T;
					real dL_dalpha = 0;
					if (do_col) for (int ch = 0; ch < NUM_CHANNELS; ch++) {
						const real c = colors[(size_t)g * NUM_CHANNELS + ch];
						accum_rec[ch] = last_alpha * last_color[ch] + ((real)1 - last_alpha) * accum_rec[ch];
						last_color[ch] = c;
						const real dL_dchannel = dL_dpixel[ch];
						dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
						acc(a_col[(size_t)g * 3 + ch], (double)(dchannel_dcolor * dL_dchannel));
					}
					if (do_flow) for (int ch = 0; ch < FLOW_CHANNELS; ch++) {
						const real f = flow_points[(size_t)g * FLOW_CHANNELS + ch];
						accum_flow_rec[ch] = last_alpha * last_flow[ch] + ((real)1 - last_alpha) * accum_flow_rec[ch];
						last_flow[ch] = f;
						dL_dalpha += (f - accum_flow_rec[ch]) * dL_dpixel_flow[ch];
						acc(a_flow[(size_t)g * 3 + ch], (double)(dchannel_dcolor * dL_dpixel_flow[ch]));
					}
					if (do_sem) for (int ch = 0; ch < D_S; ch++) {
						const real s = semantic[(size_t)g * D_S + ch];
						accum_semantic_rec[ch] = last_alpha * last_semantic[ch] + ((real)1 - last_alpha) * accum_semantic_rec[ch];
						last_semantic[ch] = s;
						dL_dalpha += (s - accum_semantic_rec[ch]) * dL_dpixel_semantic[ch];
						acc(a_sem[(size_t)g * D_S + ch], (double)(dchannel_dcolor * dL_dpixel_semantic[ch]));
					}
					if (do_depth) {
						const real d = inv_depth ? ((real)1 / (depths[g] + fc<real>(0.0000001f))) : depths[g];
						accum_depth_rec = last_alpha * last_depth + ((real)1 - last_alpha) * accum_depth_rec;
						last_depth = d;
						dL_dalpha += (d - accum_depth_rec) * dL_dpixel_depth;
						acc(a_dep[g], (double)(dchannel_dcolor * dL_dpixel_depth));
					}
					// backward.cu:612-614 -- the "opacity-T quirk": added BEFORE the *= T
					if (do_op) dL_dalpha += dL_dpixel_opacity * T_final / ((real)1 - alpha);
					dL_dalpha *= T;
					last_alpha = alpha;
					real bg_dot_dpixel = 0;
					for (int i = 0; i < NUM_CHANNELS; i++) bg_dot_dpixel += bg[i] * dL_dpixel[i];
					dL_dalpha += (-T_final / ((real)1 - alpha)) * bg_dot_dpixel;
					const real dL_dG = co[3] * dL_dalpha;
					const real gdx = G * dx, gdy = G * dy;
					const real dG_ddelx = -gdx * co[0] - gdy * co[1];
					const real dG_ddely = -gdy * co[2] - gdx * co[1];
					acc(a_mean2D[(size_t)g * 2 + 0], (double)(dL_dG * dG_ddelx * ddelx_dx));
					acc(a_mean2D[(size_t)g * 2 + 1], (double)(dL_dG * dG_ddely * ddely_dy));
					acc(a_conic[(size_t)g * 3 + 0], (double)((real)-0.5 * gdx * dx * dL_dG));
					acc(a_conic[(size_t)g * 3 + 1], (double)((real)-0.5 * gdx * dy * dL_dG));
					acc(a_conic[(size_t)g * 3 + 2], (double)((real)-0.5 * gdy * dy * dL_dG));
					acc(a_op[g], (double)(G * dL_dalpha));
				}
			}
		}
		for (int g = 0; g < P; g++) {
			dL_dmean2D[3 * g] = (real)a_mean2D[2 * (size_t)g]; dL_dmean2D[3 * g + 1] = (real)a_mean2D[2 * (size_t)g + 1]; dL_dmean2D[3 * g + 2] = 0;
			// float4 .x .y .w (backward.cu:638-640); .z stays 0
			dL_dconic[4 * g] = (real)a_conic[3 * (size_t)g]; dL_dconic[4 * g + 1] = (real)a_conic[3 * (size_t)g + 1];
			dL_dconic[4 * g + 2] = 0; dL_dconic[4 * g + 3] = (real)a_conic[3 * (size_t)g + 2];
			dL_dopacity[g] = (real)a_op[g];
			for (int c = 0; c < 3; c++) dL_dcolor[3 * g + c] = (real)a_col[3 * (size_t)g + c];
			dL_ddepth[g] = (real)a_dep[g];
			for (int c = 0; c < 3; c++) dL_dflow[3 * g + c] = (real)a_flow[3 * (size_t)g + c];
			for (int c = 0; c < D_S; c++) dL_dsemantic[(size_t)g * D_S + c] = (real)a_sem[(size_t)g * D_S + c];
		}
		for (size_t i = 0; i < (size_t)P * 3; i++) { dL_dmean3D[i] = 0; dL_dscale[i] = 0; }
		for (size_t i = 0; i < (size_t)P * 6; i++) dL_dcov3D[i] = 0;
		for (size_t i = 0; i < (size_t)P * 4; i++) dL_drot[i] = 0;
		for (size_t i = 0; i < (size_t)P * M * 3; i++) dL_dsh[i] = 0;

		const real* c3all = has_cov ? cov3D_precomp.data() : cov3D.data();
#pragma omp parallel for schedule(static)
		for (int idx = 0; idx < P; idx++) {
			if (!(radii[idx] > 0)) continue;
			computeCov2D_bw(idx, c3all + 6 * (size_t)idx, dL_dconic, dL_dmean3D, dL_dcov3D);
			preprocess_bw(idx, dL_dmean2D, dL_dmean3D, dL_dcolor, dL_ddepth, dL_dcov3D, dL_dsh, dL_dscale, dL_drot);
		}
	}

	// backward.cu:144-274 (computeCov2DCUDA)
	void computeCov2D_bw(int idx, const real* c3, const real* dL_dconics, real* dL_dmeans, real* dL_dcov) {
		V3<real> mean = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
		V3<real> dL_dconic = { dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3] };
		V3<real> t = transformPoint4x3(mean, view);
		const real h_x = focal_x, h_y = focal_y;
		const real limx = fc<real>(1.3f) * tan_fovx;
		const real limy = fc<real>(1.3f) * tan_fovy;
		const real txtz = t.x / t.z;
		const real tytz = t.y / t.z;
		t.x = std::min(limx, std::max(-limx, txtz)) * t.z;
		t.y = std::min(limy, std::max(-limy, tytz)) * t.z;
		const real x_grad_mul = (txtz < -limx || txtz > limx) ? 0 : 1;
		const real y_grad_mul = (tytz < -limy || tytz > limy) ? 0 : 1;
		M3<real> J = M3<real>::make(h_x / t.z, 0, -(h_x * t.x) / (t.z * t.z),
			0, h_y / t.z, -(h_y * t.y) / (t.z * t.z),
			0, 0, 0);
		M3<real> Wm = M3<real>::make(
			view[0], view[4], view[8],
			view[1], view[5], view[9],
			view[2], view[6], view[10]);
		M3<real> Vrk = M3<real>::make(
			c3[0], c3[1], c3[2],
			c3[1], c3[3], c3[4],
			c3[2], c3[4], c3[5]);
		M3<real> T = mul(Wm, J);
		M3<real> cov2D = mul(mul(transpose(T), transpose(Vrk)), T);
		real a = cov2D.v[0][0] += fc<real>(0.3f);
		real b = cov2D.v[0][1];
		real c = cov2D.v[1][1] += fc<real>(0.3f);
		real denom = a * c - b * b;
		real dL_da = 0, dL_db = 0, dL_dc = 0;
		real denom2inv = (real)1 / ((denom * denom) + fc<real>(0.0000001f));
		auto Tm = [&](int i, int j) { return T.v[i][j]; };
		auto Vm = [&](int i, int j) { return Vrk.v[i][j]; };
		auto Wf = [&](int i, int j) { return Wm.v[i][j]; };
		if (denom2inv != 0) {
			dL_da = denom2inv * (-c * c * dL_dconic.x + 2 * b * c * dL_dconic.y + (denom - a * c) * dL_dconic.z);
			dL_dc = denom2inv * (-a * a * dL_dconic.z + 2 * a * b * dL_dconic.y + (denom - a * c) * dL_dconic.x);
			dL_db = denom2inv * 2 * (b * c * dL_dconic.x - (denom + 2 * b * b) * dL_dconic.y + a * b * dL_dconic.z);
			dL_dcov[6 * idx + 0] = (Tm(0,0) * Tm(0,0) * dL_da + Tm(0,0) * Tm(1,0) * dL_db + Tm(1,0) * Tm(1,0) * dL_dc);
			dL_dcov[6 * idx + 3] = (Tm(0,1) * Tm(0,1) * dL_da + Tm(0,1) * Tm(1,1) * dL_db + Tm(1,1) * Tm(1,1) * dL_dc);
			dL_dcov[6 * idx + 5] = (Tm(0,2) * Tm(0,2) * dL_da + Tm(0,2) * Tm(1,2) * dL_db + Tm(1,2) * Tm(1,2) * dL_dc);
			dL_dcov[6 * idx + 1] = 2 * Tm(0,0) * Tm(0,1) * dL_da + (Tm(0,0) * Tm(1,1) + Tm(0,1) * Tm(1,0)) * dL_db + 2 * Tm(1,0) * Tm(1,1) * dL_dc;
			dL_dcov[6 * idx + 2] = 2 * Tm(0,0) * Tm(0,2) * dL_da + (Tm(0,0) * Tm(1,2) + Tm(0,2) * Tm(1,0)) * dL_db + 2 * Tm(1,0) * Tm(1,2) * dL_dc;
			dL_dcov[6 * idx + 4] = 2 * Tm(0,2) * Tm(0,1) * dL_da + (Tm(0,1) * Tm(1,2) + Tm(0,2) * Tm(1,1)) * dL_db + 2 * Tm(1,1) * Tm(1,2) * dL_dc;
		}
		else {
			for (int i = 0; i < 6; i++) dL_dcov[6 * idx + i] = 0;
		}
		real dL_dT00 = 2 * (Tm(0,0) * Vm(0,0) + Tm(0,1) * Vm(0,1) + Tm(0,2) * Vm(0,2)) * dL_da +
			(Tm(1,0) * Vm(0,0) + Tm(1,1) * Vm(0,1) + Tm(1,2) * Vm(0,2)) * dL_db;
		real dL_dT01 = 2 * (Tm(0,0) * Vm(1,0) + Tm(0,1) * Vm(1,1) + Tm(0,2) * Vm(1,2)) * dL_da +
			(Tm(1,0) * Vm(1,0) + Tm(1,1) * Vm(1,1) + Tm(1,2) * Vm(1,2)) * dL_db;
		real dL_dT02 = 2 * (Tm(0,0) * Vm(2,0) + Tm(0,1) * Vm(2,1) + Tm(0,2) * Vm(2,2)) * dL_da +
			(Tm(1,0) * Vm(2,0) + Tm(1,1) * Vm(2,1) + Tm(1,2) * Vm(2,2)) * dL_db;
		real dL_dT10 = 2 * (Tm(1,0) * Vm(0,0) + Tm(1,1) * Vm(0,1) + Tm(1,2) * Vm(0,2)) * dL_dc +
			(Tm(0,0) * Vm(0,0) + Tm(0,1) * Vm(0,1) + Tm(0,2) * Vm(0,2)) * dL_db;
		real dL_dT11 = 2 * (Tm(1,0) * Vm(1,0) + Tm(1,1) * Vm(1,1) + Tm(1,2) * Vm(1,2)) * dL_dc +
			(Tm(0,0) * Vm(1,0) + Tm(0,1) * Vm(1,1) + Tm(0,2) * Vm(1,2)) * dL_db;
		real dL_dT12 = 2 * (Tm(1,0) * Vm(2,0) + Tm(1,1) * Vm(2,1) + Tm(1,2) * Vm(2,2)) * dL_dc +
			(Tm(0,0) * Vm(2,0) + Tm(0,1) * Vm(2,1) + Tm(0,2) * Vm(2,2)) * dL_db;
		real dL_dJ00 = Wf(0,0) * dL_dT00 + Wf(0,1) * dL_dT01 + Wf(0,2) * dL_dT02;
		real dL_dJ02 = Wf(2,0) * dL_dT00 + Wf(2,1) * dL_dT01 + Wf(2,2) * dL_dT02;
		real dL_dJ11 = Wf(1,0) * dL_dT10 + Wf(1,1) * dL_dT11 + Wf(1,2) * dL_dT12;
		real dL_dJ12 = Wf(2,0) * dL_dT10 + Wf(2,1) * dL_dT11 + Wf(2,2) * dL_dT12;
		real tz = (real)1 / t.z;
		real tz2 = tz * tz;
		real tz3 = tz2 * tz;
		real dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
		real dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
		real dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;
		V3<real> dL_dmean = transformVec4x3Transpose<real>({ dL_dtx, dL_dty, dL_dtz }, view);
		dL_dmeans[3 * idx] = dL_dmean.x; dL_dmeans[3 * idx + 1] = dL_dmean.y; dL_dmeans[3 * idx + 2] = dL_dmean.z;
	}

	// backward.cu:20-139 (computeColorFromSH backward)
	void sh_bw(int idx, int deg, int max_coeffs, const real* dL_dcolor, real* dL_dmeans, real* dL_dshs) {
		V3<real> pos = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
		V3<real> dir_orig = { pos.x - campos[0], pos.y - campos[1], pos.z - campos[2] };
		real len = std::sqrt(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
		V3<real> dir = { dir_orig.x / len, dir_orig.y / len, dir_orig.z / len };
		const real* sh = shs.data() + (size_t)idx * max_coeffs * 3;
		real* dL_dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
		real dL_dRGB[3];
		for (int c = 0; c < 3; c++) dL_dRGB[c] = dL_dcolor[3 * idx + c] * (clamped[3 * idx + c] ? (real)0 : (real)1);
		real x = dir.x, y = dir.y, z = dir.z;
		real dRGBdx[3] = { 0,0,0 }, dRGBdy[3] = { 0,0,0 }, dRGBdz[3] = { 0,0,0 };
		const real C0 = fc<real>(0.28209479177387814f), C1 = fc<real>(0.4886025119029199f);
		const real C2[5] = { fc<real>(1.0925484305920792f), fc<real>(-1.0925484305920792f), fc<real>(0.31539156525252005f), fc<real>(-1.0925484305920792f), fc<real>(0.5462742152960396f) };
		const real C3[7] = { fc<real>(-0.5900435899266435f), fc<real>(2.890611442640554f), fc<real>(-0.4570457994644658f), fc<real>(0.3731763325901154f),
			fc<real>(-0.4570457994644658f), fc<real>(1.445305721320277f), fc<real>(-0.5900435899266435f) };
		auto S = [&](int k, int c) { return sh[k * 3 + c]; };
		auto setsh = [&](int k, real coef) { for (int c = 0; c < 3; c++) dL_dsh[k * 3 + c] = coef * dL_dRGB[c]; };
		setsh(0, C0);
		if (deg > 0) {
			setsh(1, -C1 * y); setsh(2, C1 * z); setsh(3, -C1 * x);
			for (int c = 0; c < 3; c++) { dRGBdx[c] = -C1 * S(3, c); dRGBdy[c] = -C1 * S(1, c); dRGBdz[c] = C1 * S(2, c); }
			if (deg > 1) {
				real xx = x * x, yy = y * y, zz = z * z;
				real xy = x * y, yz = y * z, xz = x * z;
				setsh(4, C2[0] * xy); setsh(5, C2[1] * yz); setsh(6, C2[2] * ((real)2 * zz - xx - yy));
				setsh(7, C2[3] * xz); setsh(8, C2[4] * (xx - yy));
				for (int c = 0; c < 3; c++) {
					dRGBdx[c] += C2[0] * y * S(4, c) + C2[2] * (real)2 * -x * S(6, c) + C2[3] * z * S(7, c) + C2[4] * (real)2 * x * S(8, c);
					dRGBdy[c] += C2[0] * x * S(4, c) + C2[1] * z * S(5, c) + C2[2] * (real)2 * -y * S(6, c) + C2[4] * (real)2 * -y * S(8, c);
					dRGBdz[c] += C2[1] * y * S(5, c) + C2[2] * (real)2 * (real)2 * z * S(6, c) + C2[3] * x * S(7, c);
				}
				if (deg > 2) {
					setsh(9, C3[0] * y * ((real)3 * xx - yy));
					setsh(10, C3[1] * xy * z);
					setsh(11, C3[2] * y * ((real)4 * zz - xx - yy));
					setsh(12, C3[3] * z * ((real)2 * zz - (real)3 * xx - (real)3 * yy));
					setsh(13, C3[4] * x * ((real)4 * zz - xx - yy));
					setsh(14, C3[5] * z * (xx - yy));
					setsh(15, C3[6] * x * (xx - (real)3 * yy));
					for (int c = 0; c < 3; c++) {
						dRGBdx[c] += (
							C3[0] * S(9, c) * (real)3 * (real)2 * xy +
							C3[1] * S(10, c) * yz +
							C3[2] * S(11, c) * (real)-2 * xy +
							C3[3] * S(12, c) * (real)-3 * (real)2 * xz +
							C3[4] * S(13, c) * ((real)-3 * xx + (real)4 * zz - yy) +
							C3[5] * S(14, c) * (real)2 * xz +
							C3[6] * S(15, c) * (real)3 * (xx - yy));
						dRGBdy[c] += (
							C3[0] * S(9, c) * (real)3 * (xx - yy) +
							C3[1] * S(10, c) * xz +
							C3[2] * S(11, c) * ((real)-3 * yy + (real)4 * zz - xx) +
							C3[3] * S(12, c) * (real)-3 * (real)2 * yz +
							C3[4] * S(13, c) * (real)-2 * xy +
							C3[5] * S(14, c) * (real)-2 * yz +
							C3[6] * S(15, c) * (real)-3 * (real)2 * xy);
						dRGBdz[c] += (
							C3[1] * S(10, c) * xy +
							C3[2] * S(11, c) * (real)4 * (real)2 * yz +
							C3[3] * S(12, c) * (real)3 * ((real)2 * zz - xx - yy) +
							C3[4] * S(13, c) * (real)4 * (real)2 * xz +
							C3[5] * S(14, c) * (xx - yy));
					}
				}
			}
		}
		V3<real> dL_ddir = {
			dRGBdx[0] * dL_dRGB[0] + dRGBdx[1] * dL_dRGB[1] + dRGBdx[2] * dL_dRGB[2],
			dRGBdy[0] * dL_dRGB[0] + dRGBdy[1] * dL_dRGB[1] + dRGBdy[2] * dL_dRGB[2],
			dRGBdz[0] * dL_dRGB[0] + dRGBdz[1] * dL_dRGB[1] + dRGBdz[2] * dL_dRGB[2] };
		V3<real> dL_dmean = dnormvdv(dir_orig, dL_ddir);
		dL_dmeans[3 * idx] += dL_dmean.x; dL_dmeans[3 * idx + 1] += dL_dmean.y; dL_dmeans[3 * idx + 2] += dL_dmean.z;
	}

	// backward.cu:278-341 (computeCov3D backward)
	void cov3D_bw(int idx, const real* scale, real mod, const real* rot, const real* dL_dcov3Ds, real* dL_dscales, real* dL_drots) {
		real r = rot[0], x = rot[1], y = rot[2], z = rot[3];
		M3<real> Rm = M3<real>::make(
			(real)1 - (real)2 * (y * y + z * z), (real)2 * (x * y - r * z), (real)2 * (x * z + r * y),
			(real)2 * (x * y + r * z), (real)1 - (real)2 * (x * x + z * z), (real)2 * (y * z - r * x),
			(real)2 * (x * z - r * y), (real)2 * (y * z + r * x), (real)1 - (real)2 * (x * x + y * y));
		M3<real> S = M3<real>::make(1, 0, 0, 0, 1, 0, 0, 0, 1);
		real s[3] = { mod * scale[0], mod * scale[1], mod * scale[2] };
		S.v[0][0] = s[0]; S.v[1][1] = s[1]; S.v[2][2] = s[2];
		M3<real> Mm = mul(S, Rm);
		const real* d = dL_dcov3Ds + 6 * (size_t)idx;
		M3<real> dL_dSigma = M3<real>::make(
			d[0], (real)0.5 * d[1], (real)0.5 * d[2],
			(real)0.5 * d[1], d[3], (real)0.5 * d[4],
			(real)0.5 * d[2], (real)0.5 * d[4], d[5]);
		// dL_dM = 2.0f * M * dL_dSigma  (scalar*mat first, then mat*mat)
		M3<real> twoM;
		for (int c = 0; c < 3; c++) for (int rr = 0; rr < 3; rr++) twoM.v[c][rr] = Mm.v[c][rr] * (real)2;
		M3<real> dL_dM = mul(twoM, dL_dSigma);
		M3<real> Rt = transpose(Rm);
		M3<real> dL_dMt = transpose(dL_dM);
		auto dot3 = [](const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
		dL_dscales[3 * idx + 0] = dot3(Rt.v[0], dL_dMt.v[0]);
		dL_dscales[3 * idx + 1] = dot3(Rt.v[1], dL_dMt.v[1]);
		dL_dscales[3 * idx + 2] = dot3(Rt.v[2], dL_dMt.v[2]);
		for (int k = 0; k < 3; k++) { dL_dMt.v[0][k] *= s[0]; dL_dMt.v[1][k] *= s[1]; dL_dMt.v[2][k] *= s[2]; }
		auto m = [&](int i, int j) { return dL_dMt.v[i][j]; };
		real qx = 2 * z * (m(0,1) - m(1,0)) + 2 * y * (m(2,0) - m(0,2)) + 2 * x * (m(1,2) - m(2,1));
		real qy = 2 * y * (m(1,0) + m(0,1)) + 2 * z * (m(2,0) + m(0,2)) + 2 * r * (m(1,2) - m(2,1)) - 4 * x * (m(2,2) + m(1,1));
		real qz = 2 * x * (m(1,0) + m(0,1)) + 2 * r * (m(2,0) - m(0,2)) + 2 * z * (m(1,2) + m(2,1)) - 4 * y * (m(2,2) + m(0,0));
		real qw = 2 * r * (m(0,1) - m(1,0)) + 2 * x * (m(2,0) + m(0,2)) + 2 * y * (m(1,2) + m(2,1)) - 4 * z * (m(1,1) + m(0,0));
		dL_drots[4 * idx] = qx; dL_drots[4 * idx + 1] = qy; dL_drots[4 * idx + 2] = qz; dL_drots[4 * idx + 3] = qw; // no normalisation Jacobian (backward.cu:340)
	}

	// backward.cu:346-414 (preprocessCUDA backward)
	void preprocess_bw(int idx, const real* dL_dmean2D, real* dL_dmeans, const real* dL_dcolor, const real* dL_ddepth,
		const real* dL_dcov3D, real* dL_dsh, real* dL_dscale, real* dL_drot) {
		V3<real> m = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
		V4<real> m_hom = transformPoint4x4(m, proj);
		real m_w = (real)1 / (m_hom.w + fc<real>(0.0000001f));
		real mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
		real mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
		real gx2 = dL_dmean2D[3 * idx], gy2 = dL_dmean2D[3 * idx + 1];
		real dmx = (proj[0] * m_w - proj[3] * mul1) * gx2 + (proj[1] * m_w - proj[3] * mul2) * gy2;
		real dmy = (proj[4] * m_w - proj[7] * mul1) * gx2 + (proj[5] * m_w - proj[7] * mul2) * gy2;
		real dmz = (proj[8] * m_w - proj[11] * mul1) * gx2 + (proj[9] * m_w - proj[11] * mul2) * gy2;
		dL_dmeans[3 * idx] += dmx; dL_dmeans[3 * idx + 1] += dmy; dL_dmeans[3 * idx + 2] += dmz;
		real mul3 = view[2] * m.x + view[6] * m.y + view[10] * m.z + view[14];
		real demon = inv_depth ? ((real)-1 / (mul3 * mul3 + fc<real>(0.0000001f))) : (real)1;
		real d2x = (view[2] - view[3] * mul3) * dL_ddepth[idx] * demon;
		real d2y = (view[6] - view[7] * mul3) * dL_ddepth[idx] * demon;
		real d2z = (view[10] - view[11] * mul3) * dL_ddepth[idx] * demon;
		dL_dmeans[3 * idx] += d2x; dL_dmeans[3 * idx + 1] += d2y; dL_dmeans[3 * idx + 2] += d2z;
		if (has_shs) sh_bw(idx, D, M, dL_dcolor, dL_dmeans, dL_dsh);
		if (has_scales) cov3D_bw(idx, scales.data() + 3 * (size_t)idx, scale_modifier, rotations.data() + 4 * (size_t)idx, dL_dcov3D, dL_dscale, dL_drot);
	}

	// rasterizer_impl.cu:54-66,141-153
	void markVisible(int P_, const float* means, const float* view_, const float* proj_, uint8_t* present) {
		std::vector<real> mm((size_t)P_ * 3); for (size_t i = 0; i < mm.size(); i++) mm[i] = (real)means[i];
		real v[16], p[16]; for (int i = 0; i < 16; i++) { v[i] = (real)view_[i]; p[i] = (real)proj_[i]; }
		for (int i = 0; i < P_; i++) { V3<real> pv; present[i] = in_frustum(i, mm.data(), v, p, pv) ? 1 : 0; }
	}
};

} // namespace

#define ORACLE_API(SUF, real) \
extern "C" void* adgs_oracle_create_##SUF() { return new Oracle<real>(); } \
extern "C" void adgs_oracle_destroy_##SUF(void* h) { delete (Oracle<real>*)h; } \
extern "C" int adgs_oracle_forward_##SUF(void* h, int P, int D, int M, int D_S, const float* bg, int W, int H, \
	const float* means3D, const float* shs, const float* colors, const float* flow, const float* sem, const float* opac, \
	const float* scales, float scale_mod, const float* rots, const float* cov, const float* view, const float* proj, const float* campos, \
	float tfx, float tfy, real* out_color, real* out_depth, real* img_opacity, real* img_flow, real* img_semantic, int inv_depth, int* radii) { \
	return ((Oracle<real>*)h)->forward(P, D, M, D_S, bg, W, H, means3D, shs, colors, flow, sem, opac, scales, scale_mod, rots, cov, \
		view, proj, campos, tfx, tfy, out_color, out_depth, img_opacity, img_flow, img_semantic, inv_depth != 0, radii); } \
extern "C" void adgs_oracle_backward_##SUF(void* h, const float* dL_dpix, const float* dL_ddepth_pix, const float* dL_dflow_pix, const float* dL_dsem_pix, \
	const float* grad_img_opacity, real* dL_dmean2D, real* dL_dconic, real* dL_dopacity, real* dL_dcolor, real* dL_ddepth, real* dL_dmean3D, \
	real* dL_dcov3D, real* dL_dsh, real* dL_dscale, real* dL_drot, real* dL_dflow, real* dL_dsemantic) { \
	((Oracle<real>*)h)->backward(dL_dpix, dL_ddepth_pix, dL_dflow_pix, dL_dsem_pix, grad_img_opacity, dL_dmean2D, dL_dconic, dL_dopacity, \
		dL_dcolor, dL_ddepth, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, dL_dflow, dL_dsemantic); } \
extern "C" void adgs_oracle_mark_visible_##SUF(void* h, int P, const float* means, const float* view, const float* proj, uint8_t* present) { \
	((Oracle<real>*)h)->markVisible(P, means, view, proj, present); } \
extern "C" int adgs_oracle_num_rendered_##SUF(void* h) { return ((Oracle<real>*)h)->R; } \
extern "C" void adgs_oracle_gate_margins_##SUF(void* h, real* pix_margin, real* gauss_margin) { ((Oracle<real>*)h)->gate_margins(pix_margin, gauss_margin); } \
extern "C" void adgs_oracle_pixel_conditioning_##SUF(void* h, real* cond) { ((Oracle<real>*)h)->pixel_conditioning(cond); } \
extern "C" void adgs_oracle_get_state_##SUF(void* h, real* means2D, real* depths, real* cov3D, real* rgb, real* conic_opacity, \
	uint8_t* clamped, uint32_t* tiles_touched, uint32_t* point_list, uint64_t* keys, uint32_t* ranges, uint32_t* n_contrib) { \
	Oracle<real>* o = (Oracle<real>*)h; \
	if (means2D) std::memcpy(means2D, o->means2D.data(), o->means2D.size() * sizeof(real)); \
	if (depths) std::memcpy(depths, o->depths.data(), o->depths.size() * sizeof(real)); \
	if (cov3D) std::memcpy(cov3D, o->cov3D.data(), o->cov3D.size() * sizeof(real)); \
	if (rgb) std::memcpy(rgb, o->rgb.data(), o->rgb.size() * sizeof(real)); \
	if (conic_opacity) std::memcpy(conic_opacity, o->conic_opacity.data(), o->conic_opacity.size() * sizeof(real)); \
	if (clamped) std::memcpy(clamped, o->clamped.data(), o->clamped.size()); \
	if (tiles_touched) std::memcpy(tiles_touched, o->tiles_touched.data(), o->tiles_touched.size() * 4); \
	if (point_list) std::memcpy(point_list, o->point_list.data(), o->point_list.size() * 4); \
	if (keys) std::memcpy(keys, o->keys.data(), o->keys.size() * 8); \
	if (ranges) std::memcpy(ranges, o->ranges.data(), o->ranges.size() * 4); \
	if (n_contrib) std::memcpy(n_contrib, o->n_contrib.data(), o->n_contrib.size() * 4); }

ORACLE_API(f32, float)
ORACLE_API(f64, double)

extern "C" int adgs_oracle_num_threads() {
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}
extern "C" void adgs_oracle_set_num_threads(int n) {
#ifdef _OPENMP
	omp_set_num_threads(n);
#else
	(void)n;
#endif
}
extern "C" uint32_t adgs_oracle_get_higher_msb(uint32_t n) { return getHigherMsb(n); }
