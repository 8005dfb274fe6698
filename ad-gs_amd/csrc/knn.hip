// 3-nearest-neighbour mean squared distance (simple_knn._C.distCUDA2) for gfx950.
//
// Reference semantics: KNN/simple_knn.cu:45-221: exact
// 3-NN among OTHER indices, boxes of 1024 Morton-sorted points prune the search.
// MI355X design (not the reference's one-thread-walks-everything loop):
//   * no host round trips: the AABB stays in device memory (the reference copies
//     min/max to the host twice, :196-200);
//   * points are gathered once into Morton order (float4, 16-B aligned) so a box
//     is a contiguous 16-KB run;
//   * one workgroup owns one box of queries; a candidate box is staged in LDS
//     once per workgroup and scanned from LDS by the lanes that still need it
//     (broadcast reads), instead of every thread gathering points[indices[i]].
// Compiled with -ffp-contract=off so distances are bit-identical to a plain
// evaluation (the result is then exactly the oracle's).
#include "common.h"
#include "kernels.h"
#include <cfloat>

namespace adgs {
namespace {

constexpr int BOX = 1024;          // KNN/simple_knn.cu:12
constexpr int KT = 256;            // threads per block
constexpr int QPT = BOX / KT;      // queries per thread

struct MinMax { float minx, miny, minz, maxx, maxy, maxz; };

__global__ void __launch_bounds__(KT) knn_bounds_kernel(int P, const float* __restrict__ pts, uint32_t* __restrict__ mm) {
	// init {0,0,0}: only positive values can raise the max, only negative ones lower the min
	uint32_t mx[3] = { 0, 0, 0 }, mn[3] = { 0, 0, 0 };
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
#pragma unroll
		for (int c = 0; c < 3; c++) {
			const uint32_t b = __float_as_uint(pts[3 * (size_t)i + c]);
			if (b & 0x80000000u) mn[c] = max(mn[c], b); else mx[c] = max(mx[c], b);
		}
	}
#pragma unroll
	for (int c = 0; c < 3; c++) {
#pragma unroll
		for (int off = WAVE / 2; off > 0; off >>= 1) {
			mx[c] = max(mx[c], (uint32_t)__shfl_xor((int)mx[c], off, WAVE));
			mn[c] = max(mn[c], (uint32_t)__shfl_xor((int)mn[c], off, WAVE));
		}
		if ((threadIdx.x & (WAVE - 1)) == 0) { atomicMax(&mm[c], mx[c]); atomicMax(&mm[3 + c], mn[c]); }
	}
}

__device__ __forceinline__ uint32_t prep_morton(uint32_t x) {
	x = (x | (x << 16)) & 0x030000FF;
	x = (x | (x << 8)) & 0x0300F00F;
	x = (x | (x << 4)) & 0x030C30C3;
	x = (x | (x << 2)) & 0x09249249;
	return x;
}

__global__ void __launch_bounds__(KT) knn_morton_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ mm,
	uint32_t* __restrict__ codes, uint32_t* __restrict__ ids) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= P) return;
	const float maxx = __uint_as_float(mm[0]), maxy = __uint_as_float(mm[1]), maxz = __uint_as_float(mm[2]);
	const float minx = __uint_as_float(mm[3]), miny = __uint_as_float(mm[4]), minz = __uint_as_float(mm[5]);
	const float px = pts[3 * (size_t)i], py = pts[3 * (size_t)i + 1], pz = pts[3 * (size_t)i + 2];
	const uint32_t x = prep_morton((uint32_t)(((px - minx) / (maxx - minx)) * ((1 << 10) - 1)));
	const uint32_t y = prep_morton((uint32_t)(((py - miny) / (maxy - miny)) * ((1 << 10) - 1)));
	const uint32_t z = prep_morton((uint32_t)(((pz - minz) / (maxz - minz)) * ((1 << 10) - 1)));
	codes[i] = x | (y << 1) | (z << 2);
	ids[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(KT) knn_gather_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ ids, float4* __restrict__ sorted) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= P) return;
	const uint32_t id = ids[i];
	sorted[i] = make_float4(pts[3 * (size_t)id], pts[3 * (size_t)id + 1], pts[3 * (size_t)id + 2], 0.f);
}

__global__ void __launch_bounds__(KT) knn_boxes_kernel(int P, const float4* __restrict__ sorted, MinMax* __restrict__ boxes) {
	__shared__ float red[6][KT / WAVE];
	float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
	for (int k = 0; k < QPT; k++) {
		const int i = blockIdx.x * BOX + k * KT + threadIdx.x;
		if (i < P) {
			const float4 p = sorted[i];
			mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
			mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
		}
	}
#pragma unroll
	for (int c = 0; c < 3; c++) {
#pragma unroll
		for (int off = WAVE / 2; off > 0; off >>= 1) {
			mn[c] = fminf(mn[c], __shfl_xor(mn[c], off, WAVE));
			mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], off, WAVE));
		}
		if ((threadIdx.x & (WAVE - 1)) == 0) { red[c][threadIdx.x / WAVE] = mn[c]; red[3 + c][threadIdx.x / WAVE] = mx[c]; }
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		MinMax b;
		float* o = &b.minx;
		for (int c = 0; c < 3; c++) {
			float a = red[c][0], z = red[3 + c][0];
			for (int w = 1; w < KT / WAVE; w++) { a = fminf(a, red[c][w]); z = fmaxf(z, red[3 + c][w]); }
			o[c] = a; o[3 + c] = z;
		}
		boxes[blockIdx.x] = b;
	}
}

__device__ __forceinline__ float dist_box_point(const MinMax& box, float px, float py, float pz) {
	float dx = 0.f, dy = 0.f, dz = 0.f;
	if (px < box.minx || px > box.maxx) dx = fminf(fabsf(px - box.minx), fabsf(px - box.maxx));
	if (py < box.miny || py > box.maxy) dy = fminf(fabsf(py - box.miny), fabsf(py - box.maxy));
	if (pz < box.minz || pz > box.maxz) dz = fminf(fabsf(pz - box.minz), fabsf(pz - box.maxz));
	return dx * dx + dy * dy + dz * dz;
}

__device__ __forceinline__ void update3(float rx, float ry, float rz, float px, float py, float pz, float* knn) {
	const float dx = px - rx, dy = py - ry, dz = pz - rz;
	float dist = dx * dx + dy * dy + dz * dz;
#pragma unroll
	for (int j = 0; j < 3; j++) {
		if (knn[j] > dist) { const float t = knn[j]; knn[j] = dist; dist = t; }
	}
}

__global__ void __launch_bounds__(KT) knn_meandist_kernel(int P, int nboxes, const float4* __restrict__ sorted, const uint32_t* __restrict__ ids,
	const MinMax* __restrict__ boxes, float* __restrict__ dists, unsigned long long* __restrict__ staged_total) {
	__shared__ float4 s_pts[BOX];
	const int tid = threadIdx.x;
	uint32_t staged = 0;                      // candidate boxes this workgroup scanned (bench.py: boxes visited per query box)
	float qx[QPT], qy[QPT], qz[QPT], best[QPT][3], reject[QPT];
	int qi[QPT];
#pragma unroll
	for (int k = 0; k < QPT; k++) {
		const int idx = blockIdx.x * BOX + k * KT + tid;
		qi[k] = idx;
		best[k][0] = FLT_MAX; best[k][1] = FLT_MAX; best[k][2] = FLT_MAX;
		reject[k] = -1.f;
		qx[k] = qy[k] = qz[k] = 0.f;
		if (idx < P) {
			const float4 p = sorted[idx];
			qx[k] = p.x; qy[k] = p.y; qz[k] = p.z;
			float seed[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
			for (int i = max(0, idx - 3); i <= min(P - 1, idx + 3); i++) {
				if (i == idx) continue;
				const float4 o = sorted[i];
				update3(p.x, p.y, p.z, o.x, o.y, o.z, seed);
			}
			reject[k] = seed[2];
		}
	}
	for (int b = 0; b < nboxes; b++) {
		const MinMax box = boxes[b];
		bool need[QPT];
		bool any_need = false;
#pragma unroll
		for (int k = 0; k < QPT; k++) {
			need[k] = false;
			if (qi[k] < P) {
				const float d = dist_box_point(box, qx[k], qy[k], qz[k]);
				need[k] = !(d > reject[k] || d > best[k][2]);
			}
			any_need = any_need || need[k];
		}
		if (!__syncthreads_or(any_need)) continue;
		const int b0 = b * BOX;
		const int cnt = min(BOX, P - b0);
		staged++;
		for (int k = tid; k < cnt; k += KT) s_pts[k] = sorted[b0 + k];
		__syncthreads();
		if (__any(any_need)) {
			for (int i = 0; i < cnt; i++) {
				const float4 o = s_pts[i];
#pragma unroll
				for (int k = 0; k < QPT; k++) {
					if (need[k] && (b0 + i) != qi[k]) update3(qx[k], qy[k], qz[k], o.x, o.y, o.z, best[k]);
				}
			}
		}
		__syncthreads();
	}
#pragma unroll
	for (int k = 0; k < QPT; k++) {
		if (qi[k] < P) dists[ids[qi[k]]] = (best[k][0] + best[k][1] + best[k][2]) / 3.0f;
	}
	if (tid == 0) atomicAdd(staged_total, (unsigned long long)staged);
}

struct KnnWs {
	uint32_t* mm; uint32_t* codes; uint32_t* codes_sorted; uint32_t* ids; uint32_t* ids_sorted; float4* sorted; MinMax* boxes; char* sort_temp;
	static KnnWs carve(char* chunk, size_t P, size_t* bytes) {
		Carver c(chunk); KnnWs w;
		w.mm = c.take<uint32_t>(8);
		w.codes = c.take<uint32_t>(P); w.codes_sorted = c.take<uint32_t>(P);
		w.ids = c.take<uint32_t>(P); w.ids_sorted = c.take<uint32_t>(P);
		w.sorted = c.take<float4>(P);
		w.boxes = c.take<MinMax>((P + BOX - 1) / BOX);
		w.sort_temp = c.take<char>(sort_temp_bytes(P));
		if (bytes) *bytes = c.size();
		return w;
	}
};

} // namespace

size_t knn_workspace_bytes(int P) { size_t b = 0; KnnWs::carve(nullptr, (size_t)(P > 0 ? P : 0), &b); return b; }

int knn_run(int P, const float* points, float* meanDists, char* workspace, hipStream_t stream) {
	KnnWs w = KnnWs::carve(workspace, (size_t)P, nullptr);
	const int nblk = (P + KT - 1) / KT;
	const int nboxes = (P + BOX - 1) / BOX;
	ADGS_HIP_CHECK(hipMemsetAsync(w.mm, 0, 8 * sizeof(uint32_t), stream));
	hipLaunchKernelGGL(knn_bounds_kernel, dim3(nblk < 1024 ? nblk : 1024), dim3(KT), 0, stream, P, points, w.mm);
	ADGS_HIP_CHECK(hipGetLastError());
	hipLaunchKernelGGL(knn_morton_kernel, dim3(nblk), dim3(KT), 0, stream, P, points, (const uint32_t*)w.mm, w.codes, w.ids);
	ADGS_HIP_CHECK(hipGetLastError());
	if (radix_sort_pairs_u32(w.codes, w.codes_sorted, w.ids, w.ids_sorted, (size_t)P, 30, w.sort_temp, stream) != 0) return -1;
	hipLaunchKernelGGL(knn_gather_kernel, dim3(nblk), dim3(KT), 0, stream, P, points, (const uint32_t*)w.ids_sorted, w.sorted);
	ADGS_HIP_CHECK(hipGetLastError());
	hipLaunchKernelGGL(knn_boxes_kernel, dim3(nboxes), dim3(KT), 0, stream, P, (const float4*)w.sorted, w.boxes);
	ADGS_HIP_CHECK(hipGetLastError());
	hipLaunchKernelGGL(knn_meandist_kernel, dim3(nboxes), dim3(KT), 0, stream, P, nboxes, (const float4*)w.sorted, (const uint32_t*)w.ids_sorted,
		(const MinMax*)w.boxes, meanDists, reinterpret_cast<unsigned long long*>(w.mm + 6));      // words 6..7 of the workspace: boxes scanned, summed over the query boxes
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

} // namespace adgs
