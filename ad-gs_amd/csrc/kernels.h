// Internal launcher interface between api.hip and the kernel translation units.
#pragma once
#include "common.h"

namespace adgs {

struct PreprocessArgs {
	int P, D, M, D_S;
	const float* means3D; const float* scales; float scale_modifier; const float* rotations;
	const float* opacities; const float* shs; const float* cov3D_precomp; const float* colors_precomp;
	const float* flow_points; const float* semantic;
	const float* view; const float* proj; const float* campos;
	int W, H, gx, gy;
	float tan_fovx, tan_fovy, focal_x, focal_y;
	int inv_depth;
	// outputs
	int* radii; Splat* splats; float* cov3D; uint8_t* clamped; uint32_t* tiles_touched;
};

int launch_preprocess_fwd(const PreprocessArgs& a, hipStream_t stream);
int launch_mark_visible(int P, const float* means, const float* view, uint8_t* present, hipStream_t stream);
int launch_duplicate_keys(int P, const Splat* splats, const uint32_t* offsets, const int* radii, uint64_t* keys, uint32_t* vals,
	int gx, int gy, hipStream_t stream);
int launch_tile_ranges(int L, const uint64_t* keys, uint2* ranges, hipStream_t stream);

struct RenderFwdArgs {
	const uint2* ranges; const uint32_t* point_list; const Splat* splats;
	int W, H, gx, gy, D_S;
	bool has_color, has_flow, has_sem, inv_depth;
	const float* semantic;           // [P, D_S] (only read when D_S > 1)
	const float* bg;
	float* final_T;                  // img_opacity output (holds 1 - T)
	uint32_t* n_contrib;
	float* out_color; float* out_depth; float* out_flow; float* out_semantic;
};
int launch_render_fwd(const RenderFwdArgs& a, hipStream_t stream);

struct RenderBwdArgs {
	const uint2* ranges; const uint32_t* point_list; const Splat* splats;
	int W, H, gx, gy, D_S;
	const float* semantic; const float* bg;
	const float* final_T; const uint32_t* n_contrib;
	const float* dL_dpix; const float* dL_dpix_depth; const float* dL_dpix_flow; const float* dL_dpix_sem; const float* dL_dpix_opacity;
	bool do_color, do_flow, do_sem, do_depth, do_opacity;
	float* dL_dmean2D;   // [P,3]
	float* dL_dconic;    // [P,4] (x, y, -, w)
	float* dL_dopacity;  // [P]
	float* dL_dcolor;    // [P,3]
	float* dL_ddepth;    // [P]
	float* dL_dflow;     // [P,3]
	float* dL_dsem;      // [P,D_S]
};
int launch_render_bwd(const RenderBwdArgs& a, hipStream_t stream);

struct PreprocessBwdArgs {
	int P, D, M;
	const float* means3D; const int* radii; const float* shs; const uint8_t* clamped;
	const float* scales; const float* rotations; float scale_modifier;
	const float* cov3D;              // precomputed or the forward's
	const float* view; const float* proj; const float* campos;
	float focal_x, focal_y, tan_fovx, tan_fovy;
	int inv_depth;
	const float* dL_dmean2D; const float* dL_dconic; const float* dL_dcolor; const float* dL_ddepth;
	float* dL_dmean3D; float* dL_dcov3D; float* dL_dsh; float* dL_dscale; float* dL_drot;
};
int launch_preprocess_bwd(const PreprocessBwdArgs& a, hipStream_t stream);

int knn_run(int P, const float* points, float* meanDists, char* workspace, hipStream_t stream);
size_t knn_workspace_bytes(int P);

} // namespace adgs
