// =============================================================================
// CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see raster_oracle.cpp header).
//
// Restatement of the reference simple-knn ("KNN/" = submodules/simple-knn):
// mean of the 3 smallest squared distances to OTHER indices, found through a
// Morton-sorted box pruning that is exact (KNN/simple_knn.cu:119-183), so a
// brute-force scan is an equivalent oracle; both are provided and the tests
// check them against each other (the reference ships no tests for this path:
// "parity unpinned" apart from that cross-check).
// =============================================================================
#include <cstdint>
#include <cfloat>
#include <cmath>
#include <vector>
#include <algorithm>
#include <numeric>

namespace {
constexpr int BOX_SIZE = 1024; // KNN/simple_knn.cu:12

struct F3 { float x, y, z; };
struct MinMax { F3 minn, maxx; };

// KNN/simple_knn.cu:45-53
uint32_t prepMorton(uint32_t x) {
	x = (x | (x << 16)) & 0x030000FF;
	x = (x | (x << 8)) & 0x0300F00F;
	x = (x | (x << 4)) & 0x030C30C3;
	x = (x | (x << 2)) & 0x09249249;
	return x;
}
// KNN/simple_knn.cu:55-62
uint32_t coord2Morton(F3 c, F3 minn, F3 maxx) {
	uint32_t x = prepMorton((uint32_t)(((c.x - minn.x) / (maxx.x - minn.x)) * ((1 << 10) - 1)));
	uint32_t y = prepMorton((uint32_t)(((c.y - minn.y) / (maxx.y - minn.y)) * ((1 << 10) - 1)));
	uint32_t z = prepMorton((uint32_t)(((c.z - minn.z) / (maxx.z - minn.z)) * ((1 << 10) - 1)));
	return x | (y << 1) | (z << 2);
}
// KNN/simple_knn.cu:119-129
float distBoxPoint(const MinMax& box, const F3& p) {
	F3 diff = { 0, 0, 0 };
	if (p.x < box.minn.x || p.x > box.maxx.x) diff.x = std::min(std::fabs(p.x - box.minn.x), std::fabs(p.x - box.maxx.x));
	if (p.y < box.minn.y || p.y > box.maxx.y) diff.y = std::min(std::fabs(p.y - box.minn.y), std::fabs(p.y - box.maxx.y));
	if (p.z < box.minn.z || p.z > box.maxx.z) diff.z = std::min(std::fabs(p.z - box.minn.z), std::fabs(p.z - box.maxx.z));
	return diff.x * diff.x + diff.y * diff.y + diff.z * diff.z;
}
// KNN/simple_knn.cu:131-145
void updateKBest3(const F3& ref, const F3& point, float* knn) {
	F3 d = { point.x - ref.x, point.y - ref.y, point.z - ref.z };
	float dist = d.x * d.x + d.y * d.y + d.z * d.z;
	for (int j = 0; j < 3; j++) {
		if (knn[j] > dist) { float t = knn[j]; knn[j] = dist; dist = t; }
	}
}
} // namespace

// KNN/simple_knn.cu:185-221 (SimpleKNN::knn), structure preserved.
extern "C" void adgs_oracle_knn(int P, const float* pts, float* meanDists) {
	const F3* points = (const F3*)pts;
	// cub::DeviceReduce with init {0,0,0}: the box always contains the origin (:191-200)
	F3 minn = { 0, 0, 0 }, maxx = { 0, 0, 0 };
	for (int i = 0; i < P; i++) {
		minn = { std::min(minn.x, points[i].x), std::min(minn.y, points[i].y), std::min(minn.z, points[i].z) };
		maxx = { std::max(maxx.x, points[i].x), std::max(maxx.y, points[i].y), std::max(maxx.z, points[i].z) };
	}
	std::vector<uint32_t> morton(P), indices(P);
	for (int i = 0; i < P; i++) morton[i] = coord2Morton(points[i], minn, maxx);
	std::iota(indices.begin(), indices.end(), 0u);
	std::stable_sort(indices.begin(), indices.end(), [&](uint32_t a, uint32_t b) { return morton[a] < morton[b]; });
	const int num_boxes = (P + BOX_SIZE - 1) / BOX_SIZE;
	std::vector<MinMax> boxes(num_boxes);
	// boxMinMax :78-117
	for (int b = 0; b < num_boxes; b++) {
		MinMax me = { { FLT_MAX, FLT_MAX, FLT_MAX }, { -FLT_MAX, -FLT_MAX, -FLT_MAX } };
		for (int i = b * BOX_SIZE; i < std::min(P, (b + 1) * BOX_SIZE); i++) {
			F3 p = points[indices[i]];
			me.minn = { std::min(me.minn.x, p.x), std::min(me.minn.y, p.y), std::min(me.minn.z, p.z) };
			me.maxx = { std::max(me.maxx.x, p.x), std::max(me.maxx.y, p.y), std::max(me.maxx.z, p.z) };
		}
		boxes[b] = me;
	}
	// boxMeanDist :147-183
#pragma omp parallel for schedule(dynamic, 256)
	for (int idx = 0; idx < P; idx++) {
		F3 point = points[indices[idx]];
		float best[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
		for (int i = std::max(0, idx - 3); i <= std::min(P - 1, idx + 3); i++) {
			if (i == idx) continue;
			updateKBest3(point, points[indices[i]], best);
		}
		float reject = best[2];
		best[0] = FLT_MAX; best[1] = FLT_MAX; best[2] = FLT_MAX;
		for (int b = 0; b < num_boxes; b++) {
			float dist = distBoxPoint(boxes[b], point);
			if (dist > reject || dist > best[2]) continue;
			for (int i = b * BOX_SIZE; i < std::min(P, (b + 1) * BOX_SIZE); i++) {
				if (i == idx) continue;
				updateKBest3(point, points[indices[i]], best);
			}
		}
		meanDists[indices[idx]] = (best[0] + best[1] + best[2]) / 3.0f;
	}
}

// Brute force: the definition the pruned search must equal.
extern "C" void adgs_oracle_knn_bruteforce(int P, const float* pts, float* meanDists) {
	const F3* points = (const F3*)pts;
#pragma omp parallel for schedule(dynamic, 64)
	for (int i = 0; i < P; i++) {
		float best[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
		for (int j = 0; j < P; j++) { if (j == i) continue; updateKBest3(points[i], points[j], best); }
		meanDists[i] = (best[0] + best[1] + best[2]) / 3.0f;
	}
}
