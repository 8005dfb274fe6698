// K-nearest-neighbour index (include/adgs_knn_points.h; reference call: scene/gaussian_model.py:825-833).
//
// Exact brute force, tiled: A anchors x N points distance evaluations (A = N / K in the reference's use, so N^2 / K
// pairs: 5e9 at 200 k object Gaussians) are fp32 VALU work, not memory traffic -- each block stages a 256-point tile in
// LDS once and every lane reads it by broadcast.  One lane per anchor keeps its K best in registers (sorted, branch-
// free insertion).  The point range is split over gridDim.y so that a few hundred anchors still fill 256 CUs; a second
// kernel merges the per-split lists.  Compiled with -ffp-contract=off: distances equal a plain one-rounding-per-
// operation evaluation, so the index lists equal the oracle's.
#include "common.h"
#include "../../include/adgs_knn_points.h"
#include <cfloat>

namespace adgs {
namespace {

constexpr int KP_THREADS = 256;
constexpr int KP_TILE = 256;

template <int K>
__device__ __forceinline__ void topk_insert(float (&bd)[K], uint32_t (&bi)[K], float d, uint32_t i) {
	if (!(d < bd[K - 1])) return;
	// sorted ascending; strict '<' keeps the earlier (lower) index ahead on ties
	bd[K - 1] = d; bi[K - 1] = i;
#pragma unroll
	for (int k = K - 1; k > 0; k--) {
		const bool sw = bd[k] < bd[k - 1];
		const float td = sw ? bd[k - 1] : bd[k]; const uint32_t ti = sw ? bi[k - 1] : bi[k];
		bd[k - 1] = sw ? bd[k] : bd[k - 1]; bi[k - 1] = sw ? bi[k] : bi[k - 1];
		bd[k] = td; bi[k] = ti;
	}
}

template <int K, int D>
__global__ void __launch_bounds__(KP_THREADS) knn_points_partial_kernel(int A, const float* __restrict__ anchors, int N, const float* __restrict__ points,
	int per_split, float* __restrict__ pd, uint32_t* __restrict__ pi) {
	__shared__ float s_pts[KP_TILE * D];
	const int a = blockIdx.x * KP_THREADS + threadIdx.x;
	const int n0 = blockIdx.y * per_split, n1 = min(N, n0 + per_split);
	float q[D];
#pragma unroll
	for (int c = 0; c < D; c++) q[c] = a < A ? anchors[(size_t)a * D + c] : 0.f;
	float bd[K]; uint32_t bi[K];
#pragma unroll
	for (int k = 0; k < K; k++) { bd[k] = FLT_MAX; bi[k] = 0xffffffffu; }
	for (int t0 = n0; t0 < n1; t0 += KP_TILE) {
		const int cnt = min(KP_TILE, n1 - t0);
		__syncthreads();
		for (int e = threadIdx.x; e < cnt * D; e += KP_THREADS) s_pts[e] = points[(size_t)t0 * D + e];
		__syncthreads();
		for (int j = 0; j < cnt; j++) {
			float d = 0.f;
#pragma unroll
			for (int c = 0; c < D; c++) { const float df = q[c] - s_pts[j * D + c]; d = d + df * df; }
			topk_insert<K>(bd, bi, d, (uint32_t)(t0 + j));
		}
	}
	if (a < A) {
		const size_t o = ((size_t)blockIdx.y * A + a) * K;
#pragma unroll
		for (int k = 0; k < K; k++) { pd[o + k] = bd[k]; pi[o + k] = bi[k]; }
	}
}

template <int K>
__global__ void __launch_bounds__(KP_THREADS) knn_points_merge_kernel(int A, int S, int Kout, const float* __restrict__ pd, const uint32_t* __restrict__ pi,
	int64_t* __restrict__ idx_out, float* __restrict__ dist_out) {
	const int a = blockIdx.x * KP_THREADS + threadIdx.x;
	if (a >= A) return;
	float bd[K]; uint32_t bi[K];
#pragma unroll
	for (int k = 0; k < K; k++) { bd[k] = FLT_MAX; bi[k] = 0xffffffffu; }
	for (int s = 0; s < S; s++) {                     // splits in ascending point order: ties keep the lower index
		const size_t o = ((size_t)s * A + a) * K;
		for (int k = 0; k < K; k++) {
			const uint32_t i = pi[o + k];
			if (i == 0xffffffffu) break;
			topk_insert<K>(bd, bi, pd[o + k], i);
		}
	}
	for (int k = 0; k < Kout; k++) {
		idx_out[(size_t)a * Kout + k] = bi[k] == 0xffffffffu ? (int64_t)-1 : (int64_t)bi[k];
		if (dist_out) dist_out[(size_t)a * Kout + k] = bd[k];
	}
}

inline int pad_k(int K) { return K <= 4 ? 4 : K <= 8 ? 8 : K <= 16 ? 16 : 32; }
inline int num_splits(int A, int N) {
	const int blocks_x = (A + KP_THREADS - 1) / KP_THREADS;
	int s = (2048 + blocks_x - 1) / blocks_x;          // aim at >= 2048 workgroups (8 per CU)
	const int max_s = (N + 4 * KP_TILE - 1) / (4 * KP_TILE);   // at least four tiles per split
	if (s > max_s) s = max_s;
	return s < 1 ? 1 : s;
}

template <int K>
int run(int A, const float* anchors, int N, const float* points, int D, int Kout, int64_t* idx_out, float* dist_out, char* ws, hipStream_t stream) {
	const int S = num_splits(A, N);
	const int per_split = ((N + S - 1) / S + KP_TILE - 1) / KP_TILE * KP_TILE;
	float* pd = reinterpret_cast<float*>(ws);
	uint32_t* pi = reinterpret_cast<uint32_t*>(ws + align_up((size_t)S * A * K * sizeof(float), 256));
	const dim3 grid((A + KP_THREADS - 1) / KP_THREADS, S);
	if (D == 3) hipLaunchKernelGGL((knn_points_partial_kernel<K, 3>), grid, dim3(KP_THREADS), 0, stream, A, anchors, N, points, per_split, pd, pi);
	else hipLaunchKernelGGL((knn_points_partial_kernel<K, 4>), grid, dim3(KP_THREADS), 0, stream, A, anchors, N, points, per_split, pd, pi);
	ADGS_HIP_CHECK(hipGetLastError());
	hipLaunchKernelGGL((knn_points_merge_kernel<K>), dim3((A + KP_THREADS - 1) / KP_THREADS), dim3(KP_THREADS), 0, stream, A, S, Kout, (const float*)pd,
		(const uint32_t*)pi, idx_out, dist_out);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

} // namespace
} // namespace adgs

using namespace adgs;

extern "C" size_t adgs_knn_points_workspace_bytes(int A, int N, int K) {
	if (A <= 0 || N <= 0 || K <= 0) return 256;
	const int S = num_splits(A, N);
	return 2 * align_up((size_t)S * A * pad_k(K) * sizeof(float), 256) + 256;
}

extern "C" int adgs_knn_points(int A, const float* anchors, int N, const float* points, int D, int K, int64_t* idx_out, float* dist_out,
	char* workspace, void* stream_) {
	if (A <= 0) return 0;
	if (!anchors || !points || !idx_out || !workspace) { set_error("adgs_knn_points: NULL pointer"); return -1; }
	if (D != 3 && D != 4) { set_error("adgs_knn_points: D must be 3 or 4"); return -1; }
	if (K < 1 || K > 32 || K > N) { set_error("adgs_knn_points: need 1 <= K <= min(N, 32)"); return -1; }
	hipStream_t stream = (hipStream_t)stream_;
	switch (pad_k(K)) {
	case 4: return run<4>(A, anchors, N, points, D, K, idx_out, dist_out, workspace, stream);
	case 8: return run<8>(A, anchors, N, points, D, K, idx_out, dist_out, workspace, stream);
	case 16: return run<16>(A, anchors, N, points, D, K, idx_out, dist_out, workspace, stream);
	default: return run<32>(A, anchors, N, points, D, K, idx_out, dist_out, workspace, stream);
	}
}
