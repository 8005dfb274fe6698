// K-nearest-neighbour index (include/adgs_knn_points.h; reference call: scene/gaussian_model.py:825-833).
//
// Two exact paths with identical results (index lists equal the oracle's, ties to the lower index):
//  * SLAB SEARCH (N >= 4096): points AND anchors are sorted along the axis of largest extent (this library's radix sort on the
//    order-preserving integer image of the coordinate); a workgroup of 256 neighbouring anchors sweeps the sorted points outwards
//    from its own position, tile by tile through LDS, and a lane stops taking tiles in a direction as soon as the squared distance
//    ALONG THE AXIS ALONE to the tile's near edge exceeds its current K-th best -- every point beyond is farther (fp32 sums of
//    non-negative terms never round below a term).  At 200 k object Gaussians in 8 clusters a workgroup looks at ~10 k points
//    instead of 200 000.
//  * BRUTE FORCE, tiled (small N, and the reference for the tests): A anchors x N points distance evaluations (A = N / K in the reference's use, so N^2 / K
// pairs: 5e9 at 200 k object Gaussians) are fp32 VALU work, not memory traffic -- each block stages a 256-point tile in
// LDS once and every lane reads it by broadcast.  One lane per anchor keeps its K best in registers (sorted, branch-
// free insertion).  The point range is split over gridDim.y so that a few hundred anchors still fill 256 CUs; a second
// kernel merges the per-split lists.  Compiled with -ffp-contract=off: distances equal a plain one-rounding-per-
// operation evaluation, so the index lists equal the oracle's.
#include "common.h"
#include "../../include/adgs_knn_points.h"
#include <cfloat>
#include <cstdlib>
#include <string>
#include <algorithm>

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

// ---------------------------------------------------------------- slab search
__device__ __forceinline__ uint32_t sortable_bits(float f) {
	const uint32_t u = __float_as_uint(f);
	return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);      // unsigned order == float order (-0 < +0: only affects where the walk starts)
}
// axis of largest extent (ties: the lowest axis), one workgroup
__global__ void __launch_bounds__(1024) knn_axis_kernel(int N, const float* __restrict__ pts, int D, uint32_t* __restrict__ axis_out) {
	__shared__ float s_min[4][1024 / WAVE], s_max[4][1024 / WAVE];
	float mn[4] = { FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX }, mx[4] = { -FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX };
	for (int i = threadIdx.x; i < N; i += 1024)
		for (int c = 0; c < D; c++) { const float v = pts[(size_t)i * D + c]; mn[c] = fminf(mn[c], v); mx[c] = fmaxf(mx[c], v); }
	for (int c = 0; c < 4; c++) {
#pragma unroll
		for (int off = WAVE / 2; off > 0; off >>= 1) { mn[c] = fminf(mn[c], __shfl_xor(mn[c], off, WAVE)); mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], off, WAVE)); }
		if ((threadIdx.x & (WAVE - 1)) == 0) { s_min[c][threadIdx.x / WAVE] = mn[c]; s_max[c][threadIdx.x / WAVE] = mx[c]; }
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		int best = 0; float ext = -1.f;
		for (int c = 0; c < D; c++) {
			float a = FLT_MAX, b = -FLT_MAX;
			for (int w = 0; w < 1024 / WAVE; w++) { a = fminf(a, s_min[c][w]); b = fmaxf(b, s_max[c][w]); }
			if (b - a > ext) { ext = b - a; best = c; }
		}
		*axis_out = (uint32_t)best;
	}
}
__global__ void __launch_bounds__(256) knn_keys_kernel(int n, const float* __restrict__ pts, int D, const uint32_t* __restrict__ axis, uint32_t* __restrict__ keys,
	uint32_t* __restrict__ vals) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	keys[i] = sortable_bits(pts[(size_t)i * D + *axis]); vals[i] = (uint32_t)i;
}
__global__ void __launch_bounds__(256) knn_gather_kernel(int n, const float* __restrict__ pts, int D, const uint32_t* __restrict__ order, float4* __restrict__ out) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const float* p = pts + (size_t)order[i] * D;
	out[i] = make_float4(p[0], p[1], p[2], D == 4 ? p[3] : 0.f);
}
template <int K>
__device__ __forceinline__ void topk_insert_lex(float (&bd)[K], uint32_t (&bi)[K], float d, uint32_t i) {
	// (distance, index) in lexicographic order: the walk does not meet the points in index order
	if (!(d < bd[K - 1] || (d == bd[K - 1] && i < bi[K - 1]))) return;
	bd[K - 1] = d; bi[K - 1] = i;
#pragma unroll
	for (int k = K - 1; k > 0; k--) {
		const bool sw = bd[k] < bd[k - 1] || (bd[k] == bd[k - 1] && bi[k] < bi[k - 1]);
		const float td = sw ? bd[k - 1] : bd[k]; const uint32_t ti = sw ? bi[k - 1] : bi[k];
		bd[k - 1] = sw ? bd[k] : bd[k - 1]; bi[k - 1] = sw ? bi[k] : bi[k - 1];
		bd[k] = td; bi[k] = ti;
	}
}
// One workgroup = 64 anchors that are neighbours in the axis order.  It sweeps the sorted points tile by tile (256 points staged in
// LDS, every lane reads them by broadcast, as in the brute-force kernel), first to the right of the block's first anchor, then to
// the left; a lane stops evaluating tiles in a direction once the tile's near edge is farther along the axis than its K-th best, and
// the sweep ends when no lane needs the next tile.  (A first version let every lane walk the sorted array on its own: 64 separate
// 16-byte loads per wave instruction made it twice as slow as brute force despite 20x fewer distance evaluations.)
constexpr int KS_THREADS = 64;          // anchors per workgroup: A / 64 workgroups (A = N / K anchors: ~400 at C3) fill the chip, A / 256 would not
template <int K, int D>
__global__ void __launch_bounds__(KS_THREADS) knn_slab_kernel(int A, const float* __restrict__ anchors, const uint32_t* __restrict__ a_order, int N,
	const float4* __restrict__ spts, const uint32_t* __restrict__ sidx, const uint32_t* __restrict__ skeys, const uint32_t* __restrict__ axis_p,
	int Kout, int64_t* __restrict__ idx_out, float* __restrict__ dist_out) {
	__shared__ float4 s_pts[KP_TILE];
	__shared__ uint32_t s_id[KP_TILE];
	__shared__ int s_start;
	const int tid = threadIdx.x, t = blockIdx.x * KS_THREADS + tid;
	const bool valid = t < A;
	const uint32_t a = a_order[valid ? t : A - 1];
	const int axis = (int)*axis_p;
	float q[D];
#pragma unroll
	for (int c = 0; c < D; c++) q[c] = anchors[(size_t)a * D + c];
	const float qa = axis == 0 ? q[0] : axis == 1 ? q[1] : axis == 2 ? q[2] : q[D - 1];
	if (tid == 0) {      // where the block's first anchor sits in the sorted points
		const uint32_t kq = sortable_bits(qa);
		int lo = 0, len = N;
		while (len > 0) { const int half = len >> 1; if (skeys[lo + half] < kq) { lo += half + 1; len -= half + 1; } else len = half; }
		s_start = min(lo, N - 1) / KP_TILE;
	}
	float bd[K]; uint32_t bi[K];
#pragma unroll
	for (int k = 0; k < K; k++) { bd[k] = FLT_MAX; bi[k] = 0xffffffffu; }
	__syncthreads();
	const int T0 = s_start, NT = (N + KP_TILE - 1) / KP_TILE;
#pragma unroll 1
	for (int dir = 0; dir < 2; dir++) {
		for (int T = dir ? T0 - 1 : T0; dir ? T >= 0 : T < NT; T += dir ? -1 : 1) {
			const int base = T * KP_TILE, cnt = min(KP_TILE, N - base);
			__syncthreads();
			for (int e = tid; e < cnt; e += KS_THREADS) { s_pts[e] = spts[base + e]; s_id[e] = sidx[base + e]; }
			__syncthreads();
			// the tile's edge nearest to this direction's start: its first point going right, its last going left
			const float4 e = dir ? s_pts[cnt - 1] : s_pts[0];
			const float ea = axis == 0 ? e.x : axis == 1 ? e.y : axis == 2 ? e.z : e.w;
			const float gap = dir ? qa - ea : ea - qa;             // > 0: the whole tile lies beyond the anchor in this direction
			const bool need = valid && !(gap > 0.f && gap * gap > bd[K - 1]);
			if (!__syncthreads_or(need ? 1 : 0)) break;
			if (need) {
				// eight points per round: all LDS reads of a round are in flight together and the distances are formed before the
				// (branchy, rarely taken) insertions -- one wave per SIMD has nobody else to hide an LDS round trip per point behind
				constexpr int U = 8;
				for (int j0 = 0; j0 < cnt; j0 += U) {
					float4 p[U]; uint32_t id[U]; float d[U];
#pragma unroll
					for (int u = 0; u < U; u++) { const int j = min(j0 + u, cnt - 1); p[u] = s_pts[j]; id[u] = s_id[j]; }
#pragma unroll
					for (int u = 0; u < U; u++) {
						float acc = 0.f;
						{ const float df = q[0] - p[u].x; acc = acc + df * df; }
						{ const float df = q[1] - p[u].y; acc = acc + df * df; }
						{ const float df = q[2] - p[u].z; acc = acc + df * df; }
						if (D == 4) { const float df = q[D - 1] - p[u].w; acc = acc + df * df; }
						d[u] = (j0 + u < cnt) ? acc : FLT_MAX;           // the padded tail repeats the last point: never inserted twice
						if (j0 + u >= cnt) id[u] = 0xffffffffu;
					}
#pragma unroll
					for (int u = 0; u < U; u++) topk_insert_lex<K>(bd, bi, d[u], id[u]);
				}
			}
		}
	}
	if (!valid) return;
	for (int k = 0; k < Kout; k++) {
		idx_out[(size_t)a * Kout + k] = bi[k] == 0xffffffffu ? (int64_t)-1 : (int64_t)bi[k];
		if (dist_out) dist_out[(size_t)a * Kout + k] = bd[k];
	}
}

constexpr int SLAB_MIN_N = 4096;
struct SlabLayout { uint32_t *axis, *pk0, *pk1, *pv0, *pv1, *ak0, *ak1, *av0, *av1; float4* spts; char* sort_temp; size_t bytes; };
SlabLayout slab_layout(char* ws, size_t A, size_t N) {
	Carver c(ws); SlabLayout L;
	L.axis = c.take<uint32_t>(64);
	L.pk0 = c.take<uint32_t>(N); L.pk1 = c.take<uint32_t>(N); L.pv0 = c.take<uint32_t>(N); L.pv1 = c.take<uint32_t>(N);
	L.ak0 = c.take<uint32_t>(A); L.ak1 = c.take<uint32_t>(A); L.av0 = c.take<uint32_t>(A); L.av1 = c.take<uint32_t>(A);
	L.spts = c.take<float4>(N);
	L.sort_temp = c.take<char>(sort_temp_bytes(std::max(A, N)));
	L.bytes = c.size();
	return L;
}
template <int K>
int run_slab(int A, const float* anchors, int N, const float* points, int D, int Kout, int64_t* idx_out, float* dist_out, char* ws, hipStream_t stream) {
	SlabLayout L = slab_layout(ws, (size_t)A, (size_t)N);
	hipLaunchKernelGGL(knn_axis_kernel, dim3(1), dim3(1024), 0, stream, N, points, D, L.axis);
	hipLaunchKernelGGL(knn_keys_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, N, points, D, (const uint32_t*)L.axis, L.pk0, L.pv0);
	hipLaunchKernelGGL(knn_keys_kernel, dim3((A + 255) / 256), dim3(256), 0, stream, A, anchors, D, (const uint32_t*)L.axis, L.ak0, L.av0);
	ADGS_HIP_CHECK(hipGetLastError());
	if (radix_sort_pairs_u32(L.pk0, L.pk1, L.pv0, L.pv1, (size_t)N, 32, L.sort_temp, stream) != 0) return -1;
	if (radix_sort_pairs_u32(L.ak0, L.ak1, L.av0, L.av1, (size_t)A, 32, L.sort_temp, stream) != 0) return -1;
	hipLaunchKernelGGL(knn_gather_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, N, points, D, (const uint32_t*)L.pv1, L.spts);
	if (D == 3) hipLaunchKernelGGL((knn_slab_kernel<K, 3>), dim3((A + KS_THREADS - 1) / KS_THREADS), dim3(KS_THREADS), 0, stream, A, anchors, (const uint32_t*)L.av1, N, (const float4*)L.spts,
		(const uint32_t*)L.pv1, (const uint32_t*)L.pk1, (const uint32_t*)L.axis, Kout, idx_out, dist_out);
	else hipLaunchKernelGGL((knn_slab_kernel<K, 4>), dim3((A + KS_THREADS - 1) / KS_THREADS), dim3(KS_THREADS), 0, stream, A, anchors, (const uint32_t*)L.av1, N, (const float4*)L.spts,
		(const uint32_t*)L.pv1, (const uint32_t*)L.pk1, (const uint32_t*)L.axis, Kout, idx_out, dist_out);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

} // namespace
} // namespace adgs

using namespace adgs;

extern "C" size_t adgs_knn_points_workspace_bytes(int A, int N, int K) {
	if (A <= 0 || N <= 0 || K <= 0) return 256;
	const int S = num_splits(A, N);
	const size_t brute = 2 * align_up((size_t)S * A * pad_k(K) * sizeof(float), 256) + 256;
	return std::max(brute, slab_layout(nullptr, (size_t)A, (size_t)N).bytes);
}

extern "C" int adgs_knn_points(int A, const float* anchors, int N, const float* points, int D, int K, int64_t* idx_out, float* dist_out,
	char* workspace, void* stream_) {
	if (A <= 0) return 0;
	if (!anchors || !points || !idx_out || !workspace) { set_error("adgs_knn_points: NULL pointer"); return -1; }
	if (D != 3 && D != 4) { set_error("adgs_knn_points: D must be 3 or 4"); return -1; }
	if (K < 1 || K > 32 || K > N) { set_error("adgs_knn_points: need 1 <= K <= min(N, 32)"); return -1; }
	hipStream_t stream = (hipStream_t)stream_;
	const char* force = getenv("ADGS_KNN_POINTS");          // "brute" / "slab": force a path (tests)
	const bool slab = force ? std::string(force) == "slab" : N >= SLAB_MIN_N;
	if (slab) {
		switch (pad_k(K)) {
		case 4: return run_slab<4>(A, anchors, N, points, D, K, idx_out, dist_out, workspace, stream);
		case 8: return run_slab<8>(A, anchors, N, points, D, K, idx_out, dist_out, workspace, stream);
		case 16: return run_slab<16>(A, anchors, N, points, D, K, idx_out, dist_out, workspace, stream);
		default: return run_slab<32>(A, anchors, N, points, D, K, idx_out, dist_out, workspace, stream);
		}
	}
	switch (pad_k(K)) {
	case 4: return run<4>(A, anchors, N, points, D, K, idx_out, dist_out, workspace, stream);
	case 8: return run<8>(A, anchors, N, points, D, K, idx_out, dist_out, workspace, stream);
	case 16: return run<16>(A, anchors, N, points, D, K, idx_out, dist_out, workspace, stream);
	default: return run<32>(A, anchors, N, points, D, K, idx_out, dist_out, workspace, stream);
	}
}
