// Environment-map background for gfx950: scene/env.py:11-76 in one kernel per direction.
// One thread per pixel; the ray / angle arithmetic is fp32 in the reference's order (the sample position is
// sensitive: 1 ulp of the ray is ~5e-4 texel on an 8192^2 map), bilinear weights as ATen's grid_sampler forms them.
// Forward: 4*C gathered texels in, C floats out per pixel.  Backward: 4*C fp32 atomics per pixel into the map gradient.
#include "common.h"
#include "../../include/adgs_envmap.h"
#include <cmath>

namespace adgs {
namespace {

constexpr int MAXC = 8;
struct EnvCam { float inv_f, half_w, half_h; float R[9]; int H, W, Hm, Wm, C; };

struct Taps { int x0, y0; float w[4]; bool ok[4]; };      // nw, ne, sw, se

__device__ __forceinline__ Taps env_taps(const EnvCam& a, int px, int py) {
	// K^-1 [x, y, 1] with K = [[f,0,W/2],[0,f,H/2],[0,0,1]]  (env.py:11-27), then F.normalize (eps 1e-12)
	// evaluated as (x - W/2) / f: exactly 0 at the principal point, where the azimuth is singular when the camera looks
	// along the map's pole (a fused k*x + c would leave a rounding residue there and an arbitrary azimuth)
	float vx = ((float)px - a.half_w) * a.inv_f, vy = ((float)py - a.half_h) * a.inv_f, vz = 1.0f;
	float inv = 1.f / fmaxf(sqrtf(vx * vx + vy * vy + vz * vz), 1e-12f);
	vx *= inv; vy *= inv; vz *= inv;
	// world_view_transform[:3,:3] @ ray (env.py:61), normalised again in get_env_color (env.py:70)
	float rx = a.R[0] * vx + a.R[1] * vy + a.R[2] * vz, ry = a.R[3] * vx + a.R[4] * vy + a.R[5] * vz, rz = a.R[6] * vx + a.R[7] * vy + a.R[8] * vz;
	inv = 1.f / fmaxf(sqrtf(rx * rx + ry * ry + rz * rz), 1e-12f);
	rx *= inv; ry *= inv; rz *= inv;
	const float az = atan2f(ry, rx), el = atan2f(rz, hypotf(rx, ry));          // graphics_utils.py:95-100
	const float gx = az * 0.31830988618379067f, gy = el * 0.6366197723675814f; // * (1/pi, 2/pi)
	// grid_sample, align_corners=True: ((g + 1) / 2) * (size - 1); bilinear with zero padding
	const float ix = ((gx + 1.f) / 2.f) * (float)(a.Wm - 1), iy = ((gy + 1.f) / 2.f) * (float)(a.Hm - 1);
	const float fx0 = floorf(ix), fy0 = floorf(iy);
	Taps t;
	t.x0 = (int)fx0; t.y0 = (int)fy0;
	const float x1 = fx0 + 1.f, y1 = fy0 + 1.f;
	t.w[0] = (x1 - ix) * (y1 - iy); t.w[1] = (ix - fx0) * (y1 - iy); t.w[2] = (x1 - ix) * (iy - fy0); t.w[3] = (ix - fx0) * (iy - fy0);
	const bool xa = t.x0 >= 0 && t.x0 < a.Wm, xb = t.x0 + 1 >= 0 && t.x0 + 1 < a.Wm, ya = t.y0 >= 0 && t.y0 < a.Hm, yb = t.y0 + 1 >= 0 && t.y0 + 1 < a.Hm;
	t.ok[0] = xa && ya; t.ok[1] = xb && ya; t.ok[2] = xa && yb; t.ok[3] = xb && yb;
	return t;
}

__global__ void __launch_bounds__(256) envmap_fwd_kernel(EnvCam a, const float* __restrict__ grid, float* __restrict__ bg) {
	const int px = blockIdx.x * 64 + (threadIdx.x & 63), py = blockIdx.y * 4 + (threadIdx.x >> 6);
	if (px >= a.W || py >= a.H) return;
	const Taps t = env_taps(a, px, py);
	const size_t plane = (size_t)a.Hm * a.Wm, o = (size_t)py * a.W + px;
	const size_t base = (size_t)t.y0 * a.Wm + t.x0;
	for (int c = 0; c < a.C; c++) {
		const float* g = grid + c * plane;
		float v = 0.f;
		if (t.ok[0]) v += g[base] * t.w[0];
		if (t.ok[1]) v += g[base + 1] * t.w[1];
		if (t.ok[2]) v += g[base + a.Wm] * t.w[2];
		if (t.ok[3]) v += g[base + a.Wm + 1] * t.w[3];
		bg[(size_t)c * a.H * a.W + o] = 1.f / (1.f + expf(-v));
	}
}

// Backward: neighbouring pixels hit the same texels (an 8192^2 map under a 1920-pixel, 50-degree view has ~1.7 pixels
// per texel and every pixel touches a 2x2 footprint), so the 64x4-pixel workgroup first accumulates its contributions
// in an LDS image of its texel bounding box (ds_add_f32) and flushes each touched texel with ONE global atomic --
// about an order of magnitude fewer L2 atomics than one per (pixel, corner, channel).  Workgroups whose footprint
// does not fit (the azimuth seam, the poles) fall back to direct atomics.
constexpr int TEXCAP = 2048;            // texels of the LDS footprint image (x MAXC channels would be 64 KiB: sized per launch)

__global__ void __launch_bounds__(256) envmap_bwd_kernel(EnvCam a, const float* __restrict__ bg, const float* __restrict__ g_bg, float* __restrict__ g_grid,
	uint8_t* __restrict__ marks, int tile_elems) {
	extern __shared__ float s_acc[];                     // [C][TEXCAP]
	__shared__ int s_box[4];                             // minx, miny, maxx, maxy over the valid corners of the block
	const int tid = threadIdx.x;
	const int px = blockIdx.x * 64 + (tid & 63), py = blockIdx.y * 4 + (tid >> 6);
	const bool valid = px < a.W && py < a.H;
	if (tid == 0) { s_box[0] = 0x7fffffff; s_box[1] = 0x7fffffff; s_box[2] = -0x7fffffff; s_box[3] = -0x7fffffff; }
	__syncthreads();
	Taps t;
	{
		// bounding box of the block's valid corners: reduced inside each wave first -- LDS atomics retire about one LANE per cycle and
		// CU (measured: the 4 x 256 same-address atomics of a block cost as much as a third of its 3072 accumulation atomics)
		int lox = 0x7fffffff, loy = 0x7fffffff, hix = -0x7fffffff, hiy = -0x7fffffff;
		if (valid) {
			t = env_taps(a, px, py);
			const int x0 = max(t.x0, 0), x1 = min(t.x0 + 1, a.Wm - 1), y0 = max(t.y0, 0), y1 = min(t.y0 + 1, a.Hm - 1);
			if (x0 <= x1 && y0 <= y1) { lox = x0; hix = x1; loy = y0; hiy = y1; }
		}
#pragma unroll
		for (int off = WAVE / 2; off > 0; off >>= 1) {
			lox = min(lox, __shfl_xor(lox, off, WAVE)); loy = min(loy, __shfl_xor(loy, off, WAVE));
			hix = max(hix, __shfl_xor(hix, off, WAVE)); hiy = max(hiy, __shfl_xor(hiy, off, WAVE));
		}
		if ((tid & (WAVE - 1)) == 0 && lox <= hix) { atomicMin(&s_box[0], lox); atomicMin(&s_box[1], loy); atomicMax(&s_box[2], hix); atomicMax(&s_box[3], hiy); }
	}
	__syncthreads();
	const int minx = s_box[0], miny = s_box[1];
	const int fw = s_box[2] - minx + 1, fh = s_box[3] - miny + 1;
	if (fw <= 0 || fh <= 0) return;                      // nothing inside the map
	const bool in_lds = (long long)fw * fh <= TEXCAP;    // block-uniform
	const size_t plane = (size_t)a.Hm * a.Wm, o = (size_t)py * a.W + px;
	if (in_lds) {
		const int ntex = fw * fh;
		for (int c = 0; c < a.C; c++) for (int i = tid; i < ntex; i += 256) s_acc[c * TEXCAP + i] = 0.f;
		__syncthreads();
		{
			// Neighbouring pixels of a row (lanes 2j, 2j+1) land in the same or in adjacent texel columns (~1.6 pixels per texel at
			// C3): the pair's contributions to a shared texel are summed in registers and issued once -- LDS float atomics retire about
			// one LANE per cycle and CU and were 120 of this kernel's 190 us.  Same column: the even lane issues all four taps of
			// both pixels; adjacent column: the even lane takes the odd lane's left taps into its right ones (8 -> 6 atomics).
			const int lx = t.x0 - minx, ly = t.y0 - miny;
			const int kx = valid ? t.x0 : -0x40000000, ky = valid ? t.y0 : -0x40000000;       // an invalid pixel pairs with nobody
			const int px_ = __shfl_xor(kx, 1, WAVE), py_ = __shfl_xor(ky, 1, WAVE);
			const bool odd = tid & 1;
			const int ex = odd ? px_ : kx, ey = odd ? py_ : ky, ox = odd ? kx : px_, oy = odd ? ky : py_;      // the pair's even / odd pixel
			const bool both = ex > -0x40000000 && ox > -0x40000000 && ey == oy;
			const bool same = both && ox == ex, adj = both && ox == ex + 1;
			// all channel loads first (a rolled loop over a run-time C pays one round trip per channel before its LDS atomics)
			float bv[MAXC], gv[MAXC];
#pragma unroll
			for (int c = 0; c < MAXC; c++) { const size_t oc = (size_t)min(c, a.C - 1) * a.H * a.W + (valid ? o : 0); bv[c] = bg[oc]; gv[c] = g_bg[oc]; }
#pragma unroll
			for (int c = 0; c < MAXC; c++) {
				if (c >= a.C) break;
				const float b = bv[c];
				const float gr = valid ? gv[c] * (b * (1.f - b)) : 0.f;
				float v0 = t.ok[0] && valid ? gr * t.w[0] : 0.f, v1 = t.ok[1] && valid ? gr * t.w[1] : 0.f;
				float v2 = t.ok[2] && valid ? gr * t.w[2] : 0.f, v3 = t.ok[3] && valid ? gr * t.w[3] : 0.f;
				const float p0 = __shfl_xor(v0, 1, WAVE), p1 = __shfl_xor(v1, 1, WAVE), p2 = __shfl_xor(v2, 1, WAVE), p3 = __shfl_xor(v3, 1, WAVE);
				bool i0 = t.ok[0], i1 = t.ok[1], i2 = t.ok[2], i3 = t.ok[3];          // which of its taps this lane issues
				if (!odd) {
					if (same) { v0 += p0; v1 += p1; v2 += p2; v3 += p3; }
					else if (adj) { v1 += p0; v3 += p2; }
				} else {
					if (same) i0 = i1 = i2 = i3 = false;
					else if (adj) i0 = i2 = false;
				}
				if (!valid) continue;
				float* acc = s_acc + c * TEXCAP;
				if (i0) atomicAdd(acc + ly * fw + lx, v0);
				if (i1) atomicAdd(acc + ly * fw + lx + 1, v1);
				if (i2) atomicAdd(acc + (ly + 1) * fw + lx, v2);
				if (i3) atomicAdd(acc + (ly + 1) * fw + lx + 1, v3);
			}
		}
		__syncthreads();
		for (int i = tid; i < ntex; i += 256) {
			const int ly = i / fw, lx = i - ly * fw;
			const size_t dst = (size_t)(miny + ly) * a.Wm + (minx + lx);
			for (int c = 0; c < a.C; c++) {
				const float v = s_acc[c * TEXCAP + i];
				if (v != 0.f) { atomicAdd(g_grid + c * plane + dst, v); if (marks) marks[(c * plane + dst) / (size_t)tile_elems] = 1; }
			}
		}
		return;
	}
	if (!valid) return;
	const size_t base = (size_t)t.y0 * a.Wm + t.x0;
	for (int c = 0; c < a.C; c++) {
		const float b = bg[(size_t)c * a.H * a.W + o];
		const float gr = g_bg[(size_t)c * a.H * a.W + o] * (b * (1.f - b));
		float* g = g_grid + c * plane;
		if (t.ok[0]) atomicAdd(g + base, gr * t.w[0]);
		if (t.ok[1]) atomicAdd(g + base + 1, gr * t.w[1]);
		if (t.ok[2]) atomicAdd(g + base + a.Wm, gr * t.w[2]);
		if (t.ok[3]) atomicAdd(g + base + a.Wm + 1, gr * t.w[3]);
		if (marks) {
			const size_t e = c * plane + base;
			if (t.ok[0]) marks[e / (size_t)tile_elems] = 1;
			if (t.ok[1]) marks[(e + 1) / (size_t)tile_elems] = 1;
			if (t.ok[2]) marks[(e + a.Wm) / (size_t)tile_elems] = 1;
			if (t.ok[3]) marks[(e + a.Wm + 1) / (size_t)tile_elems] = 1;
		}
	}
}

static int make_cam(EnvCam& a, int C, int Hm, int Wm, int H, int W, float focal, const float* R9, const char* who) {
	if (C <= 0 || C > MAXC || Hm < 2 || Wm < 2 || !(focal > 0.f) || !R9) { set_error(std::string(who) + ": bad arguments (1..8 channels, map >= 2x2, focal > 0, R9 != NULL)"); return -1; }
	a.inv_f = (float)(1.0 / (double)focal); a.half_w = (float)W * 0.5f; a.half_h = (float)H * 0.5f;
	for (int i = 0; i < 9; i++) a.R[i] = R9[i];
	a.H = H; a.W = W; a.Hm = Hm; a.Wm = Wm; a.C = C;
	return 0;
}

} // namespace
} // namespace adgs

using namespace adgs;

extern "C" int adgs_envmap_forward(int C, int Hm, int Wm, const float* grid_map, int H, int W, float focal, const float* R9, float* background, void* stream) {
	if (H <= 0 || W <= 0) return 0;
	EnvCam a;
	if (make_cam(a, C, Hm, Wm, H, W, focal, R9, "adgs_envmap_forward") != 0) return -1;
	if (!grid_map || !background) { set_error("adgs_envmap_forward: NULL grid_map / background"); return -1; }
	hipLaunchKernelGGL(envmap_fwd_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, grid_map, background);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

extern "C" int adgs_envmap_backward_marked(int C, int Hm, int Wm, int H, int W, float focal, const float* R9,
	const float* background, const float* dL_dbackground, float* dL_dgrid_map, uint8_t* tile_marks, int tile_elems, void* stream) {
	if (H <= 0 || W <= 0) return 0;
	EnvCam a;
	if (make_cam(a, C, Hm, Wm, H, W, focal, R9, "adgs_envmap_backward") != 0) return -1;
	if (!background || !dL_dbackground || !dL_dgrid_map) { set_error("adgs_envmap_backward: NULL pointer"); return -1; }
	if (tile_marks && tile_elems <= 0) { set_error("adgs_envmap_backward_marked: tile_elems must be positive"); return -1; }
	hipLaunchKernelGGL(envmap_bwd_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), (size_t)C * TEXCAP * sizeof(float), (hipStream_t)stream, a, background, dL_dbackground, dL_dgrid_map,
		tile_marks, tile_elems);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
extern "C" int adgs_envmap_backward(int C, int Hm, int Wm, int H, int W, float focal, const float* R9,
	const float* background, const float* dL_dbackground, float* dL_dgrid_map, void* stream) {
	return adgs_envmap_backward_marked(C, Hm, Wm, H, W, focal, R9, background, dL_dbackground, dL_dgrid_map, nullptr, 0, stream);
}
