// Densify / prune compaction (include/adgs_densify.h): the reference's clone -> split -> prune sequence
// (scene/gaussian_model.py:581-611, 715-861) evaluated as one row map per side and one gather per tensor.
// Everything here is HBM-bound data movement plus a few transcendental threshold tests per Gaussian; it runs once per
// densification interval (100 iterations), so the kernels are written for clarity: flat, coalesced, no LDS.
// The threshold tests decide integer outcomes (which rows exist): no FMA contraction, accurate expf / logf.
#include "common.h"
#include "../../include/adgs_densify.h"

namespace adgs {
namespace {
#pragma clang fp contract(off)

__device__ __forceinline__ float max_scale(const float* s3) { return fmaxf(fmaxf(expf(s3[0]), expf(s3[1])), expf(s3[2])); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void select_flags_kernel(adgs_densify_side s, uint32_t* clone_flag, uint32_t* split_flag) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i > s.N) return;
	uint32_t c = 0, p = 0;
	if (i < s.N) {
		float g = s.grad_accum[i] / s.denom[i];          // 0/0 -> NaN -> 0, x/0 -> inf stays (gaussian_model.py:836-838)
		if (g != g) g = 0.f;
		if (fabsf(g) >= s.grad_threshold) {
			const bool small = max_scale(s.scaling + 3 * (size_t)i) <= s.dense_extent;
			c = small ? 1u : 0u; p = small ? 0u : 1u;
		}
	}
	clone_flag[i] = c; split_flag[i] = p;                // element N is the sentinel that turns the exclusive scan into totals
}

__global__ void select_compact_kernel(int N, const uint32_t* clone_flag, const uint32_t* clone_off, const uint32_t* split_flag, const uint32_t* split_off,
	uint32_t* clone_index, uint32_t* split_index, uint32_t* counts) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < N) {
		if (clone_flag[i]) clone_index[clone_off[i]] = (uint32_t)i;
		if (split_flag[i]) split_index[split_off[i]] = (uint32_t)i;
	}
	if (i == 0) { counts[0] = clone_off[N]; counts[1] = split_off[N]; }
}

struct PlanArgs { adgs_densify_side s; const uint32_t* clone_index; int n_clone; const uint32_t* split_index; int n_split; int total; };

__device__ __forceinline__ void candidate(const PlanArgs& a, int c, uint32_t& src, uint32_t& aux, bool& child) {
	child = false;
	if (c < a.s.N) { src = (uint32_t)c; aux = ADGS_ROW_KEEP; }
	else if (c < a.s.N + a.n_clone) { src = a.clone_index[c - a.s.N]; aux = ADGS_ROW_CLONE; }
	else { const int j = c - a.s.N - a.n_clone; src = a.split_index[j % a.n_split]; aux = (uint32_t)j; child = true; }
}

__global__ void plan_keep_kernel(PlanArgs a, const uint32_t* split_flag_by_row, uint32_t* keep) {
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c > a.total) return;
	uint32_t k = 0;
	if (c < a.total) {
		uint32_t src, aux; bool child;
		candidate(a, c, src, aux, child);
		bool alive = true;
		if (c < a.s.N && split_flag_by_row[c]) alive = false;                   // split parents are removed (:765-767)
		if (alive) {
			bool prune = sigmoidf(a.s.opacity[src]) < a.s.min_opacity;              // :851-852
			if (a.s.prune_big) {
				const float* s3 = a.s.scaling + 3 * (size_t)src;
				float m;
				if (child) m = fmaxf(fmaxf(expf(logf(expf(s3[0]) / 1.6f)), expf(logf(expf(s3[1]) / 1.6f))), expf(logf(expf(s3[2]) / 1.6f)));
				else m = max_scale(s3);
				prune = prune || (m > a.s.big_extent);                                 // :854-857
			}
			alive = !prune;
		}
		k = alive ? 1u : 0u;
	}
	keep[c] = k;
}

__global__ void plan_scatter_kernel(PlanArgs a, const uint32_t* keep, const uint32_t* off, uint32_t* row_src, uint32_t* row_aux, uint32_t* count) {
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c < a.total && keep[c]) {
		uint32_t src, aux; bool child;
		candidate(a, c, src, aux, child);
		row_src[off[c]] = src; row_aux[off[c]] = aux;
	}
	if (c == 0) count[0] = off[a.total];
}

__global__ void split_flag_rows_kernel(int N, const uint32_t* split_index, int n_split, uint32_t* flag) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_split) flag[split_index[i]] = 1u;
}

__global__ void gather_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int L, size_t total, const uint32_t* __restrict__ row_src,
	const uint32_t* __restrict__ row_aux, int is_state) {
	const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= total) return;
	const size_t row = e / (size_t)L;
	const int col = (int)(e - row * (size_t)L);
	float v = 0.f;
	if (!is_state || row_aux[row] == ADGS_ROW_KEEP) v = src[(size_t)row_src[row] * L + col];
	dst[e] = v;
}

__global__ void split_rows_kernel(const float* xyz, const float* scaling, const float* rot, const float* samples, int n_out,
	const uint32_t* row_src, const uint32_t* row_aux, float* xyz_dst, float* scaling_dst) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_out) return;
	const uint32_t j = row_aux[i];
	if (j >= ADGS_ROW_CLONE) return;
	const size_t p = row_src[i];
	// build_rotation (utils/general_utils.py:79-95): normalised quaternion (r, x, y, z)
	const float q0 = rot[4 * p], q1 = rot[4 * p + 1], q2 = rot[4 * p + 2], q3 = rot[4 * p + 3];
	const float norm = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
	const float r = q0 / norm, x = q1 / norm, y = q2 / norm, z = q3 / norm;
	const float R[3][3] = { { 1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y) },
	                        { 2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x) },
	                        { 2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y) } };
	const float s0 = samples[3 * (size_t)j], s1 = samples[3 * (size_t)j + 1], s2 = samples[3 * (size_t)j + 2];
#pragma unroll
	for (int k = 0; k < 3; k++) {
		xyz_dst[3 * (size_t)i + k] = (R[k][0] * s0 + R[k][1] * s1 + R[k][2] * s2) + xyz[3 * p + k];
		scaling_dst[3 * (size_t)i + k] = logf(expf(scaling[3 * p + k]) / 1.6f);
	}
}

__global__ void reset_opacity_kernel(int N, float* o) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= N) return;
	const float x = fminf(sigmoidf(o[i]), 0.01f);
	o[i] = logf(x / (1.0f - x));
}

inline unsigned blocks(size_t n) { return (unsigned)((n + 255) / 256); }

} // namespace
} // namespace adgs

using namespace adgs;

// workspace: [flagA n+1][flagB n+1][offA n+1][offB n+1][scan temp]
static size_t ws_words(int n) { return align_up((size_t)(n + 1) * sizeof(uint32_t), 256); }
extern "C" size_t adgs_densify_workspace_bytes(int n_rows) {
	if (n_rows < 0) n_rows = 0;
	return 4 * ws_words(n_rows) + align_up(scan_temp_bytes((size_t)n_rows + 1), 256) + 256;
}

extern "C" int adgs_densify_select(const adgs_densify_side* side, uint32_t* clone_index, uint32_t* split_index, uint32_t* counts, char* workspace, void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	if (!side || !counts || !workspace) { set_error("adgs_densify_select: NULL pointer"); return -1; }
	const int N = side->N;
	if (N < 0) { set_error("adgs_densify_select: negative N"); return -1; }
	if (N > 0 && (!side->grad_accum || !side->denom || !side->scaling || !side->opacity || !clone_index || !split_index)) { set_error("adgs_densify_select: NULL pointer"); return -1; }
	uint32_t* fa = reinterpret_cast<uint32_t*>(workspace); uint32_t* fb = reinterpret_cast<uint32_t*>(workspace + ws_words(N));
	uint32_t* oa = reinterpret_cast<uint32_t*>(workspace + 2 * ws_words(N)); uint32_t* ob = reinterpret_cast<uint32_t*>(workspace + 3 * ws_words(N));
	char* temp = workspace + 4 * ws_words(N);
	hipLaunchKernelGGL(select_flags_kernel, dim3(blocks((size_t)N + 1)), dim3(256), 0, stream, *side, fa, fb);
	ADGS_HIP_CHECK(hipGetLastError());
	if (exclusive_scan_u32(fa, oa, (size_t)N + 1, temp, stream) != 0) return -1;
	if (exclusive_scan_u32(fb, ob, (size_t)N + 1, temp, stream) != 0) return -1;
	hipLaunchKernelGGL(select_compact_kernel, dim3(blocks((size_t)N + 1)), dim3(256), 0, stream, N, fa, oa, fb, ob, clone_index, split_index, counts);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

extern "C" int adgs_densify_plan(const adgs_densify_side* side, const uint32_t* clone_index, int n_clone, const uint32_t* split_index, int n_split,
	uint32_t* row_src, uint32_t* row_aux, uint32_t* count, char* workspace, void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	if (!side || !count || !workspace) { set_error("adgs_densify_plan: NULL pointer"); return -1; }
	const int N = side->N;
	if (N < 0 || n_clone < 0 || n_split < 0 || n_clone > N || n_split > N) { set_error("adgs_densify_plan: inconsistent counts"); return -1; }
	const long long total_ll = (long long)N + n_clone + 2ll * n_split;
	if (total_ll > 0x3fffffffll) { set_error("adgs_densify_plan: too many rows"); return -1; }
	const int total = (int)total_ll;
	if (total > 0 && (!row_src || !row_aux || (n_clone > 0 && !clone_index) || (n_split > 0 && !split_index))) { set_error("adgs_densify_plan: NULL pointer"); return -1; }
	uint32_t* keep = reinterpret_cast<uint32_t*>(workspace); uint32_t* off = reinterpret_cast<uint32_t*>(workspace + ws_words(total));
	uint32_t* sflag = reinterpret_cast<uint32_t*>(workspace + 2 * ws_words(total));
	char* temp = workspace + 4 * ws_words(total);
	ADGS_HIP_CHECK(hipMemsetAsync(sflag, 0, (size_t)(N + 1) * sizeof(uint32_t), stream));
	if (n_split > 0) {
		hipLaunchKernelGGL(split_flag_rows_kernel, dim3(blocks((size_t)n_split)), dim3(256), 0, stream, N, split_index, n_split, sflag);
		ADGS_HIP_CHECK(hipGetLastError());
	}
	PlanArgs a; a.s = *side; a.clone_index = clone_index; a.n_clone = n_clone; a.split_index = split_index; a.n_split = n_split; a.total = total;
	hipLaunchKernelGGL(plan_keep_kernel, dim3(blocks((size_t)total + 1)), dim3(256), 0, stream, a, sflag, keep);
	ADGS_HIP_CHECK(hipGetLastError());
	if (exclusive_scan_u32(keep, off, (size_t)total + 1, temp, stream) != 0) return -1;
	hipLaunchKernelGGL(plan_scatter_kernel, dim3(blocks((size_t)total + 1)), dim3(256), 0, stream, a, keep, off, row_src, row_aux, count);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

extern "C" int adgs_densify_gather_rows(const float* src, float* dst, int row_floats, int n_out, const uint32_t* row_src, const uint32_t* row_aux,
	int is_state, void* stream) {
	if (n_out <= 0 || row_floats <= 0) return 0;
	if (!src || !dst || !row_src || !row_aux) { set_error("adgs_densify_gather_rows: NULL pointer"); return -1; }
	const size_t total = (size_t)n_out * row_floats;
	hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks(total)), dim3(256), 0, (hipStream_t)stream, src, dst, row_floats, total, row_src, row_aux, is_state);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

extern "C" int adgs_densify_split_rows(const float* xyz_src, const float* scaling_src, const float* rotation_src, const float* samples,
	int n_out, const uint32_t* row_src, const uint32_t* row_aux, float* xyz_dst, float* scaling_dst, void* stream) {
	if (n_out <= 0) return 0;
	if (!xyz_src || !scaling_src || !rotation_src || !row_src || !row_aux || !xyz_dst || !scaling_dst) { set_error("adgs_densify_split_rows: NULL pointer"); return -1; }
	hipLaunchKernelGGL(split_rows_kernel, dim3(blocks((size_t)n_out)), dim3(256), 0, (hipStream_t)stream, xyz_src, scaling_src, rotation_src, samples, n_out,
		row_src, row_aux, xyz_dst, scaling_dst);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

extern "C" int adgs_reset_opacity(int N, float* opacity, void* stream) {
	if (N <= 0) return 0;
	if (!opacity) { set_error("adgs_reset_opacity: NULL pointer"); return -1; }
	hipLaunchKernelGGL(reset_opacity_kernel, dim3(blocks((size_t)N)), dim3(256), 0, (hipStream_t)stream, N, opacity);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
