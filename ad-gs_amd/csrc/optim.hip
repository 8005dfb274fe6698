// Fused multi-tensor Adam for gfx950: every parameter group of the model in one launch, one pass
// over (p, g, m, v) -- 16 B read + 12 B written per parameter (+4 B when the gradient is zeroed),
// the HBM-streaming floor of the update.  Reference: torch.optim.Adam as configured at
// scene/gaussian_model.py:370 (eps = 1e-15, no weight decay), stepped at train.py:163-167.
#include "common.h"
#include "adam_update.h"
#include "kernels.h"
#include "../../include/adgs_optim.h"
#include <cmath>

namespace adgs {
namespace {

constexpr int AB = 256;                 // threads per block
constexpr int AV = 4;                   // elements per thread per iteration (one 16-byte access per array)
constexpr int AI = 4;                   // iterations per thread
constexpr int ATILE = AB * AV * AI;     // elements per block
static_assert(WAVE * AV == ADGS_ADAM_TILE && ATILE % ADGS_ADAM_TILE == 0, "adgs_adam_group.tile_active: one byte per wave and iteration");

struct AdamTable {
	adgs_adam_group g[ADGS_ADAM_MAX_GROUPS];
	uint32_t first_block[ADGS_ADAM_MAX_GROUPS + 1];     // group i owns blocks [first_block[i], first_block[i+1])
	float step_size[ADGS_ADAM_MAX_GROUPS];               // lr / (1 - beta1^t)
	float inv_bc2_sqrt[ADGS_ADAM_MAX_GROUPS];            // 1 / sqrt(1 - beta2^t)
	int n;
	float beta1, beta2, eps;
	int zero_grad;
};

__global__ void __launch_bounds__(AB) adam_kernel(AdamTable t) {
	// which group does this block belong to (block-uniform binary search over <= 32 entries)
	int lo = 0, hi = t.n;
	while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (blockIdx.x >= t.first_block[mid]) lo = mid; else hi = mid; }
	const adgs_adam_group& G = t.g[lo];
	const float step_size = t.step_size[lo], ibc2 = t.inv_bc2_sqrt[lo];
	const int64_t base = (int64_t)(blockIdx.x - t.first_block[lo]) * ATILE;
	const bool zero_grad = t.zero_grad || (G.flags & ADGS_ADAM_ZERO_GRAD);
	const bool vec = ((reinterpret_cast<uintptr_t>(G.param) | reinterpret_cast<uintptr_t>(G.grad) | reinterpret_cast<uintptr_t>(G.exp_avg) |
	                   reinterpret_cast<uintptr_t>(G.exp_avg_sq)) & 15) == 0;
#pragma unroll
	for (int it = 0; it < AI; it++) {
		const int64_t i = base + ((int64_t)it * AB + threadIdx.x) * AV;
		// the 64 lanes of a wave cover one tile of ADGS_ADAM_TILE = 256 consecutive elements per iteration (i0 = its first element):
		// everything about the tile map is wave-uniform
		const int64_t i0 = base + ((int64_t)it * AB + (threadIdx.x & ~(WAVE - 1))) * AV;
		if (i0 >= G.numel) break;
		if (G.tile_active) {
			// a tile whose gradients and moments have been zero in every step so far: the update is the identity -- read the
			// gradient only (4 of the 28 bytes per element) and skip the tile while that is still all zero
			const int64_t tile = i0 / ADGS_ADAM_TILE;
			if (G.tile_active[tile] == 0) {
				if (G.flags & ADGS_ADAM_TILES_MARKED) continue;      // the gradient's producer marks the tiles it writes: nothing to look at
				bool nz = false;
				if (i < G.numel) {
					if (vec && i + AV <= G.numel) { const float4 g = *reinterpret_cast<const float4*>(G.grad + i); nz = g.x != 0.f || g.y != 0.f || g.z != 0.f || g.w != 0.f; }
					else for (int k = 0; k < AV && i + k < G.numel; k++) nz = nz || G.grad[i + k] != 0.f;
				}
				if (__ballot(nz) == 0ull) continue;
				if ((threadIdx.x & (WAVE - 1)) == 0) G.tile_active[tile] = 1;
			}
		}
		if (i >= G.numel) continue;
		if (vec && i + AV <= G.numel) {
			float4 p = ld_stream4(reinterpret_cast<const float4*>(G.param + i)), g = ld_stream4(reinterpret_cast<const float4*>(G.grad + i));
			float4 m = ld_stream4(reinterpret_cast<const float4*>(G.exp_avg + i)), v = ld_stream4(reinterpret_cast<const float4*>(G.exp_avg_sq + i));
			adam_update(p.x, m.x, v.x, g.x, t.beta1, t.beta2, t.eps, step_size, ibc2);
			adam_update(p.y, m.y, v.y, g.y, t.beta1, t.beta2, t.eps, step_size, ibc2);
			adam_update(p.z, m.z, v.z, g.z, t.beta1, t.beta2, t.eps, step_size, ibc2);
			adam_update(p.w, m.w, v.w, g.w, t.beta1, t.beta2, t.eps, step_size, ibc2);
			st_stream4(reinterpret_cast<float4*>(G.param + i), p);
			st_stream4(reinterpret_cast<float4*>(G.exp_avg + i), m);
			st_stream4(reinterpret_cast<float4*>(G.exp_avg_sq + i), v);
			if (zero_grad) *reinterpret_cast<float4*>(G.grad + i) = make_float4(0.f, 0.f, 0.f, 0.f);
		} else {
			for (int k = 0; k < AV && i + k < G.numel; k++) {
				float p = G.param[i + k], m = G.exp_avg[i + k], v = G.exp_avg_sq[i + k];
				adam_update(p, m, v, G.grad[i + k], t.beta1, t.beta2, t.eps, step_size, ibc2);
				G.param[i + k] = p; G.exp_avg[i + k] = m; G.exp_avg_sq[i + k] = v;
				if (zero_grad) G.grad[i + k] = 0.f;
			}
		}
	}
}

__global__ void __launch_bounds__(256) densification_stats_kernel(int N, const int32_t* __restrict__ radii, const float* __restrict__ g,
	float* __restrict__ accum, float* __restrict__ denom, float* __restrict__ max_r) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= N) return;
	const int r = radii[i];
	if (r <= 0) return;                                   // visibility_filter = radii > 0 (gaussian_renderer/__init__.py:101)
	const float gx = g[3 * (size_t)i], gy = g[3 * (size_t)i + 1];
	accum[i] += sqrtf(gx * gx + gy * gy);                 // torch.norm(grad[:, :2], dim=-1)
	denom[i] += 1.f;
	if (max_r) max_r[i] = fmaxf(max_r[i], (float)r);
}

} // namespace

// the host forms the bias corrections in double like torch does (python floats), then rounds once
void adam_bias_terms(float lr, int step, float beta1, float beta2, float* step_size, float* inv_bc2_sqrt) {
	const double bc1 = 1.0 - std::pow((double)beta1, (double)step), bc2 = 1.0 - std::pow((double)beta2, (double)step);
	*step_size = (float)((double)lr / bc1);
	*inv_bc2_sqrt = (float)(1.0 / std::sqrt(bc2));
}
} // namespace adgs

using namespace adgs;

extern "C" int adgs_densification_stats(int N, const int32_t* radii, const float* viewspace_grad, float* xyz_gradient_accum, float* denom,
	float* max_radii2D, void* stream) {
	if (N <= 0) return 0;
	if (!radii || !viewspace_grad || !xyz_gradient_accum || !denom) { set_error("adgs_densification_stats: NULL pointer"); return -1; }
	hipLaunchKernelGGL(densification_stats_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, radii, viewspace_grad, xyz_gradient_accum, denom, max_radii2D);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

extern "C" int adgs_adam_step(const adgs_adam_group* groups, int n_groups, float beta1, float beta2, float eps, int zero_grad, void* stream_) {
	if (n_groups <= 0) return 0;
	if (!groups || n_groups > ADGS_ADAM_MAX_GROUPS) { set_error("adgs_adam_step: between 1 and 32 groups per call"); return -1; }
	AdamTable t;
	t.n = 0; t.beta1 = beta1; t.beta2 = beta2; t.eps = eps; t.zero_grad = zero_grad;
	uint64_t blocks = 0;
	for (int i = 0; i < n_groups; i++) {
		const adgs_adam_group& g = groups[i];
		if (g.numel <= 0) continue;
		if (!g.param || !g.grad || !g.exp_avg || !g.exp_avg_sq || g.step < 1) { set_error("adgs_adam_step: NULL pointer or step < 1 in a group"); return -1; }
		t.g[t.n] = g;
		t.first_block[t.n] = (uint32_t)blocks;
		adam_bias_terms(g.lr, g.step, beta1, beta2, &t.step_size[t.n], &t.inv_bc2_sqrt[t.n]);
		blocks += (uint64_t)((g.numel + ATILE - 1) / ATILE);
		t.n++;
	}
	if (t.n == 0) return 0;
	if (blocks > 0x7fffffffull) { set_error("adgs_adam_step: too many elements for one launch"); return -1; }
	t.first_block[t.n] = (uint32_t)blocks;
	hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(AB), 0, (hipStream_t)stream_, t);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
