/*
 * adgs_envmap.h -- C ABI of the environment-map background (libadgs_hip.so).  SURVEY.md section 8(f) row 3.
 *
 * Replaces EnvironmentMap.get_image_background (scene/env.py:43-76, called from render(),
 * gaussian_renderer/__init__.py:93): per pixel
 *   ray_cam = normalize(K^-1 [x, y, 1]),  ray = normalize(R ray_cam),  R = world_view_transform[:3,:3]
 *   (az, el) = (atan2(ry, rx), atan2(rz, hypot(rx, ry)))        utils/graphics_utils.py:95-100
 *   background[c] = sigmoid(bilinear(grid_map[c], az/pi, 2 el/pi))   grid_sample, align_corners=True, zero padding
 * in one kernel (the reference caches a [H,W,3] ray tensor per camera and runs ~12 torch kernels per call),
 * and its backward into grid_map (the only trainable input).
 * R9: nine HOST floats, row-major.  All other pointers are device fp32.
 */
#ifndef ADGS_ENVMAP_H
#define ADGS_ENVMAP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* grid_map [C, Hm, Wm] -> background [C, H, W] */
int adgs_envmap_forward(int C, int Hm, int Wm, const float* grid_map, int H, int W, float focal, const float* R9,
	float* background, void* stream);

/* dL_dgrid_map [C, Hm, Wm] += d(sum dL_dbackground * background)/d grid_map  (atomic accumulation: zero-initialise it);
 * `background` is the forward's output (sigmoid' = b (1 - b)). */
int adgs_envmap_backward(int C, int Hm, int Wm, int H, int W, float focal, const float* R9,
	const float* background, const float* dL_dbackground, float* dL_dgrid_map, void* stream);

/* The same, and tile_marks[e / tile_elems] = 1 for every element e of dL_dgrid_map (flat index over [C, Hm, Wm]) that receives a
 * contribution: the byte map adgs_adam_group.tile_active of the map's optimizer (ADGS_ADAM_TILES_MARKED, include/adgs_optim.h), so
 * that neither a fill pass over the dense gradient nor a scan of it is needed per step.  tile_marks == NULL: adgs_envmap_backward. */
int adgs_envmap_backward_marked(int C, int Hm, int Wm, int H, int W, float focal, const float* R9,
	const float* background, const float* dL_dbackground, float* dL_dgrid_map, uint8_t* tile_marks, int tile_elems, void* stream);

#ifdef __cplusplus
}
#endif
#endif
