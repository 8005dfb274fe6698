// K-nearest-neighbour index (include/adgs_knn_points.h; reference call: scene/gaussian_model.py:825-833).
//
// Two exact paths with identical results (index lists equal the oracle's, ties to the lower index):
//  * SLAB SEARCH (N >= 4096): points AND anchors are sorted along the axis of largest extent (this library's radix sort on the
//    order-preserving integer image of the coordinate); ONE WAVE PER ANCHOR sweeps the sorted points outwards from the anchor's
//    position, 64 points per round (one per lane), and stops in a direction as soon as the squared distance ALONG THE AXIS ALONE
//    to the round's nearest point exceeds the current K-th best -- every point beyond is farther (fp32 sums of non-negative terms
//    never round below a term).  The K best are a sorted list across the lanes.  At 200 k object Gaussians in 8 clusters an
//    anchor looks at ~10 k points instead of 200 000.
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
// axis of largest extent (ties: the lowest axis): per-block minima / maxima of every coordinate meet in eight words through
// integer atomics on the order-preserving image of the floats (one 1024-thread workgroup reading all points took 120 us at
// 200 k points), the last block to finish picks the axis.  ext[0..3] = min, ext[4..7] = max (sortable bits), ext[8] = blocks done.
__device__ __forceinline__ float from_sortable_bits(uint32_t k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu)); }
__global__ void __launch_bounds__(256) knn_axis_init_kernel(uint32_t* __restrict__ ext) {
	if (threadIdx.x < 4) ext[threadIdx.x] = 0xffffffffu; else if (threadIdx.x < 9) ext[threadIdx.x] = 0u;
}
__global__ void __launch_bounds__(256) knn_axis_kernel(int N, const float* __restrict__ pts, int D, uint32_t* __restrict__ ext, uint32_t* __restrict__ axis_out) {
	__shared__ uint32_t s_last;
	float mn[4] = { FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX }, mx[4] = { -FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX };
	for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256)
		for (int c = 0; c < D; c++) { const float v = pts[(size_t)i * D + c]; mn[c] = fminf(mn[c], v); mx[c] = fmaxf(mx[c], v); }
	__shared__ float s_mn[4][256 / WAVE], s_mx[4][256 / WAVE];
	for (int c = 0; c < D; c++) {
#pragma unroll
		for (int off = WAVE / 2; off > 0; off >>= 1) { mn[c] = fminf(mn[c], __shfl_xor(mn[c], off, WAVE)); mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], off, WAVE)); }
		if ((threadIdx.x & (WAVE - 1)) == 0) { s_mn[c][threadIdx.x / WAVE] = mn[c]; s_mx[c][threadIdx.x / WAVE] = mx[c]; }
	}
	__syncthreads();
	if ((int)threadIdx.x < D) {      // one pair of atomics per block and coordinate: same-address atomics serialise in L2
		const int c = threadIdx.x;
		float a = s_mn[c][0], b = s_mx[c][0];
		for (int w = 1; w < 256 / WAVE; w++) { a = fminf(a, s_mn[c][w]); b = fmaxf(b, s_mx[c][w]); }
		if (a <= b) { atomicMin(&ext[c], sortable_bits(a)); atomicMax(&ext[4 + c], sortable_bits(b)); }
	}
	__threadfence();
	__syncthreads();
	if (threadIdx.x == 0) s_last = atomicAdd(&ext[8], 1u) == gridDim.x - 1 ? 1u : 0u;
	__syncthreads();
	if (!s_last || threadIdx.x != 0) return;
	__threadfence();
	int best = 0; float e = -1.f;
	for (int c = 0; c < D; c++) {
		const float a = from_sortable_bits(atomicOr(&ext[c], 0u)), b = from_sortable_bits(atomicOr(&ext[4 + c], 0u));
		if (b - a > e) { e = b - a; best = c; }
	}
	*axis_out = (uint32_t)best;
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
// One WAVE = KW_ANCHORS anchors that are neighbours in the axis order (their coordinates live in scalar registers).  The wave
// sweeps the sorted points outwards from the first anchor's position, 64 points per round (one per lane: a coalesced 1 KiB load
// shared by all its anchors), first to the right, then to the left; anchor a's K best are a sorted list ACROSS the lanes (lane k
// holds the k-th entry).  Per round and anchor: one distance per lane, one ballot of "beats the K-th best"; the rare candidates
// are inserted one by one (a ballot gives the position, a one-lane shift makes room).  An anchor stops in a direction once the
// round's nearest point along the axis is farther than its K-th best; the sweep ends when all its anchors have stopped.
// History (200 k points / 25 k anchors, K = 8, 4-D; the sweep kernel alone): every lane walking the sorted array on its own: 2x slower
// than brute force (5.5 ms); 64 anchors per workgroup, one thread each, tiles through LDS: 4.2 ms (390 lone waves; the per-lane
// insertions diverge -- a wave executed ~6x more insertion than distance instructions -- and a block sweeps the union of its anchors'
// windows, 21 k points against 9.7 k per anchor); four threads per anchor: 2.7 ms; a wave per 8 / 4 / 2 / 1 anchors: 1.0 / 0.70 / 0.6 /
// 0.45 ms -- fewer anchors per wave = less union, more waves; the 5 GB of L2 reads of the one-anchor form are not what bounds it.
#ifndef ADGS_KNN_WAVE_ANCHORS
#define ADGS_KNN_WAVE_ANCHORS 1
#endif
constexpr int KW_ANCHORS = ADGS_KNN_WAVE_ANCHORS;
constexpr int KW_WAVES = 4;             // independent waves per workgroup (no LDS, no barriers)
template <int K, int D>
__global__ void __launch_bounds__(WAVE * KW_WAVES) knn_sweep_kernel(int A, const float* __restrict__ anchors, const uint32_t* __restrict__ a_order, int N,
	const float4* __restrict__ spts, const uint32_t* __restrict__ sidx, const uint32_t* __restrict__ skeys, const uint32_t* __restrict__ axis_p,
	int Kout, int64_t* __restrict__ idx_out, float* __restrict__ dist_out) {
	const int lane = threadIdx.x & (WAVE - 1);
	const int t0 = (blockIdx.x * KW_WAVES + (threadIdx.x >> 6)) * KW_ANCHORS;
	if (t0 >= A) return;                                    // the whole wave
	const int nA = min(KW_ANCHORS, A - t0);                 // wave-uniform
	const int axis = (int)*axis_p;
	constexpr uint64_t maskK = K >= 64 ? ~0ull : ((1ull << K) - 1ull);
	// anchor ids and coordinates: lane a loads anchor a, the values are broadcast into scalars
	const uint32_t my_a = a_order[min(t0 + min(lane, KW_ANCHORS - 1), A - 1)];
	float my_q[D];
#pragma unroll
	for (int c = 0; c < D; c++) my_q[c] = anchors[(size_t)my_a * D + c];
	float q[KW_ANCHORS][D], qa[KW_ANCHORS];
#pragma unroll
	for (int a = 0; a < KW_ANCHORS; a++) {
#pragma unroll
		for (int c = 0; c < D; c++) q[a][c] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_q[c]), a));
		qa[a] = axis == 0 ? q[a][0] : axis == 1 ? q[a][1] : axis == 2 ? q[a][2] : q[a][D - 1];
	}
	// where the first anchor sits in the sorted points (wave-uniform binary search)
	int s0;
	{
		const uint32_t kq = sortable_bits(qa[0]);
		int lo = 0, len = N;
		while (len > 0) { const int half = len >> 1; if (skeys[lo + half] < kq) { lo += half + 1; len -= half + 1; } else len = half; }
		s0 = lo;
	}
	float bd[KW_ANCHORS]; uint32_t bi[KW_ANCHORS];          // lane k: the k-th best of anchor a
	float kth[KW_ANCHORS]; uint32_t kthi[KW_ANCHORS];       // scalars: the K-th entry
#pragma unroll
	for (int a = 0; a < KW_ANCHORS; a++) { bd[a] = FLT_MAX; bi[a] = 0xffffffffu; kth[a] = FLT_MAX; kthi[a] = 0xffffffffu; }
#pragma unroll 1
	for (int dir = 0; dir < 2; dir++) {
		bool done[KW_ANCHORS];
#pragma unroll
		for (int a = 0; a < KW_ANCHORS; a++) done[a] = a >= nA;
		// the next round's loads are issued before this round is evaluated (a wave alone with a long window -- the tail of the kernel: an
		// anchor in a sparse region looks at 45 k points -- otherwise pays one L2 round trip per 64 points); rolled: four rounds
		// unrolled (32 copies of the per-anchor block, 48 KiB of code) ran slower than no prefetch at all
		const int first = dir ? s0 - 1 : s0, step = dir ? -WAVE : WAVE;
		float4 pn = spts[min(max(dir ? first - lane : first + lane, 0), N - 1)];
		uint32_t idn = sidx[min(max(dir ? first - lane : first + lane, 0), N - 1)];
#pragma unroll 1
		for (int base = first; dir ? base >= 0 : base < N; base += step) {
			const int idx = dir ? base - lane : base + lane;
			const bool ok = idx >= 0 && idx < N;
			const float4 p = pn;
			const uint32_t id = idn;
			{
				const int nidx = min(max(idx + step, 0), N - 1);
				pn = spts[nidx]; idn = sidx[nidx];
			}
			// lane 0 holds the point nearest along the axis in both directions
			const float pa = axis == 0 ? p.x : axis == 1 ? p.y : axis == 2 ? p.z : p.w;
			const float ea = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(pa)));
			bool any = false;
#pragma unroll
			for (int a = 0; a < KW_ANCHORS; a++) {
				if (done[a]) continue;
				const float gap = dir ? qa[a] - ea : ea - qa[a];       // > 0: this round and everything after it lies beyond the anchor
				if (gap > 0.f && gap * gap > kth[a]) { done[a] = true; continue; }
				any = true;
				float d = 0.f;
				{ const float df = q[a][0] - p.x; d = d + df * df; }
				{ const float df = q[a][1] - p.y; d = d + df * df; }
				{ const float df = q[a][2] - p.z; d = d + df * df; }
				if (D == 4) { const float df = q[a][D - 1] - p.w; d = d + df * df; }
				uint64_t m = __ballot(ok && (d < kth[a] || (d == kth[a] && id < kthi[a])));
				while (m) {
					const int j = __builtin_ctzll(m);
					m &= m - 1;
					const float dj = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d), j));
					const uint32_t ij = __builtin_amdgcn_readlane(id, j);
					if (!(dj < kth[a] || (dj == kth[a] && ij < kthi[a]))) continue;      // an earlier candidate of this round tightened the list
					// (distance, index) order: the sweep does not meet the points in index order
					const int pos = __popcll(__ballot(bd[a] < dj || (bd[a] == dj && bi[a] < ij)) & maskK);
					const float ud = __shfl_up(bd[a], 1, WAVE); const uint32_t ui = __shfl_up(bi[a], 1, WAVE);
					bd[a] = lane < pos ? bd[a] : lane == pos ? dj : ud;
					bi[a] = lane < pos ? bi[a] : lane == pos ? ij : ui;
					kth[a] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(bd[a]), K - 1));
					kthi[a] = __builtin_amdgcn_readlane(bi[a], K - 1);
				}
			}
			if (!any) break;
		}
	}
#pragma unroll
	for (int a = 0; a < KW_ANCHORS; a++) {
		if (a >= nA) break;
		const uint32_t anchor = __builtin_amdgcn_readlane(my_a, a);
		if (lane < Kout) {
			idx_out[(size_t)anchor * Kout + lane] = bi[a] == 0xffffffffu ? (int64_t)-1 : (int64_t)bi[a];
			if (dist_out) dist_out[(size_t)anchor * Kout + lane] = bd[a];
		}
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
	hipLaunchKernelGGL(knn_axis_init_kernel, dim3(1), dim3(256), 0, stream, L.axis + 16);
	hipLaunchKernelGGL(knn_axis_kernel, dim3((unsigned)std::min(128, (N + 255) / 256)), dim3(256), 0, stream, N, points, D, L.axis + 16, L.axis);
	hipLaunchKernelGGL(knn_keys_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, N, points, D, (const uint32_t*)L.axis, L.pk0, L.pv0);
	hipLaunchKernelGGL(knn_keys_kernel, dim3((A + 255) / 256), dim3(256), 0, stream, A, anchors, D, (const uint32_t*)L.axis, L.ak0, L.av0);
	ADGS_HIP_CHECK(hipGetLastError());
	if (radix_sort_pairs_u32(L.pk0, L.pk1, L.pv0, L.pv1, (size_t)N, 32, L.sort_temp, stream) != 0) return -1;
	if (radix_sort_pairs_u32(L.ak0, L.ak1, L.av0, L.av1, (size_t)A, 32, L.sort_temp, stream) != 0) return -1;
	hipLaunchKernelGGL(knn_gather_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, N, points, D, (const uint32_t*)L.pv1, L.spts);
	const unsigned grid = (unsigned)((A + KW_ANCHORS * KW_WAVES - 1) / (KW_ANCHORS * KW_WAVES));
	if (D == 3) hipLaunchKernelGGL((knn_sweep_kernel<K, 3>), dim3(grid), dim3(WAVE * KW_WAVES), 0, stream, A, anchors, (const uint32_t*)L.av1, N, (const float4*)L.spts,
		(const uint32_t*)L.pv1, (const uint32_t*)L.pk1, (const uint32_t*)L.axis, Kout, idx_out, dist_out);
	else hipLaunchKernelGGL((knn_sweep_kernel<K, 4>), dim3(grid), dim3(WAVE * KW_WAVES), 0, stream, A, anchors, (const uint32_t*)L.av1, N, (const float4*)L.spts,
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
