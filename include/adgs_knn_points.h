/*
 * adgs_knn_points.h -- C ABI of the K-nearest-neighbour index (libadgs_hip.so).  SURVEY.md section 8(f) row 4.
 *
 * Replaces pytorch3d.ops.knn_points (pytorch3d 0.7.x; NOT vendored in the reference, not installed here: parity is
 * unpinned at that boundary and anchored on the published semantics) as called by GaussianModel.set_obj_near_idx
 * (scene/gaussian_model.py:825-833): for each of A anchor points the K nearest of N points in D = 3 or 4 dimensions
 * (xyz, or xyz + time * scene_extent), squared Euclidean distance accumulated in fp32 over the dimensions in order,
 * results sorted by ascending distance (ties: lower point index first).
 *
 * idx_out [A,K] int64 (torch's index dtype, as knn_points returns); dist_out [A,K] fp32 squared distances or NULL.
 * K must be <= min(N, 32).
 */
#ifndef ADGS_KNN_POINTS_H
#define ADGS_KNN_POINTS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

size_t adgs_knn_points_workspace_bytes(int A, int N, int K);
int adgs_knn_points(int A, const float* anchors, int N, const float* points, int D, int K, int64_t* idx_out, float* dist_out,
	char* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif
