// Shared host/device helpers for the gfx950 AD-GS hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <string>

namespace adgs {

constexpr int TILE_X = 16;     // RAST/cuda_rasterizer/config.h:16-17
constexpr int TILE_Y = 16;
constexpr int TILE_PIX = TILE_X * TILE_Y;
constexpr int MAX_SEMANTIC = 32; // config.h:18
constexpr int WAVE = 64;

// Per-Gaussian record written by the forward preprocess and gathered by the
// blend kernels: one aligned 64-byte line per Gaussian.
struct __attribute__((aligned(64))) Splat {
	float x, y;              // pixel-space mean (means2D)
	float ca, cb, cc;        // conic (inverse 2D covariance) xx, xy, yy
	float opacity;
	float r, g, b;           // colour (SH-evaluated or colors_precomp)
	float dval;              // blended depth value: z or 1/(z+1e-7) (forward.cu:374-375)
	float fx, fy, fz;        // flow point (world position at the other time)
	float sem0;              // first semantic channel
	float aux;               // default pipeline: tau = 2 ln(255 opacity) + slack, the bound of the tile test (alpha >= 1/255 <=> d^T Q d <= tau);
	                         // stage-by-stage ("classic") pipeline: the raw view-space depth (the key of duplicate_keys)
	float lean;             // 1.0: neither `power > 0` nor the 0.99 clamp can fire for this Gaussian (preprocess.hip); else 0.0
};
static_assert(sizeof(Splat) == 64, "Splat must be one 64-byte line");

// (Until round 3 a 32-byte `FilterRec` per Gaussian -- mean, conic, tau, tile rectangle -- fed the v2 tile filter; the filter now reads the
// Splat line itself, and the rectangle travels as row / column masks in the cell lists: render_v2.hip.)

void set_error(const std::string& msg);

#define ADGS_HIP_CHECK(expr)                                                             \
	do {                                                                                 \
		hipError_t _e = (expr);                                                          \
		if (_e != hipSuccess) {                                                          \
			adgs::set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) +  \
				" (" + __FILE__ + ":" + std::to_string(__LINE__) + ")");                \
			return -1;                                                                   \
		}                                                                                \
	} while (0)

// After a kernel launch: always check the launch error; with debug also
// synchronise like the reference's CHECK_CUDA (auxiliary.h:166-173).
#define ADGS_LAUNCH_CHECK(debug, stream)                                                 \
	do {                                                                                 \
		ADGS_HIP_CHECK(hipGetLastError());                                               \
		if (debug) ADGS_HIP_CHECK(hipStreamSynchronize(stream));                         \
	} while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Sub-allocator over a caller-provided chunk (the role of `obtain` in
// RAST/cuda_rasterizer/rasterizer_impl.h:22-28); 256-byte alignment.
struct Carver {
	char* base; size_t off;
	explicit Carver(char* b) : base(b), off(0) {}
	template <typename T> T* take(size_t count) {
		off = align_up(off, 256);
		T* p = reinterpret_cast<T*>(base ? base + off : nullptr);
		off += count * sizeof(T);
		return p;
	}
	size_t size() const { return align_up(off, 256) + 256; }
};

// What a frame's first kernel does for the binning kernels behind it, beside its own work (sh0 on the raw-SH path, bin_prepare otherwise):
// zero the counters they accumulate into and take the frame's SNAPSHOT of the depth-slab bounds (the shared table is rewritten by every
// frame's sort kernels -- possibly of another stream --, and the counting and the scattering pass of a frame must bin by the same bounds).
struct FramePrologue { uint32_t* zero; int n_zero; uint32_t* copy_dst; const uint32_t* copy_src; int n_copy; };
__device__ __forceinline__ void run_frame_prologue(const FramePrologue& p) {      // call from every thread of the grid
	const int nb = min((int)gridDim.x, 8), b = (int)blockIdx.x;      // the first (up to) eight blocks share the work
	if (b < nb) {
		for (int i = b * (int)blockDim.x + threadIdx.x; i < p.n_zero; i += nb * (int)blockDim.x) p.zero[i] = 0u;
		for (int i = b * (int)blockDim.x + threadIdx.x; i < p.n_copy; i += nb * (int)blockDim.x) p.copy_dst[i] = p.copy_src[i];
	}
}

// ---- optional per-stage timing with HIP events on the launch stream (bench.py; implemented in api.hip) ----
enum Stage { ST_PREPROCESS = 0, ST_SCAN, ST_DUPLICATE, ST_SORT, ST_RANGES, ST_RENDER_FWD, ST_RENDER_BWD, ST_PREPROCESS_BWD, ST_DEFORM_FWD, ST_DEFORM_BWD,
	ST_EXPAND, ST_COUNT };
struct StageTimer {
	bool on; int stage; hipEvent_t a, b; hipStream_t s;
	StageTimer(int stage, hipStream_t stream);
	~StageTimer();
	StageTimer(const StageTimer&) = delete;
	StageTimer& operator=(const StageTimer&) = delete;
};

// ---- device primitives (primitives.hip) ----
size_t scan_temp_bytes(size_t n);
// out[i] = sum_{j<i} in[j]  (in == out allowed)
int exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, char* temp, hipStream_t stream);
constexpr int SCAN_AUX_SLOTS = 32;
int exclusive_scan_u32_sum(const uint32_t* in, uint32_t* out, size_t n, char* temp, const uint32_t* aux_in, unsigned long long* aux_total, hipStream_t stream);

size_t sort_temp_bytes(size_t n);
// Stable LSD radix sort of (key,value) pairs on key bits [0, end_bit).
// Result is left in keys_out/vals_out; *_in are clobbered.
int radix_sort_pairs_u64(uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_in, uint32_t* vals_out,
	size_t n, int end_bit, char* temp, hipStream_t stream);
int radix_sort_pairs_u64_dn(uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_in, uint32_t* vals_out,
	size_t n_cap, const uint32_t* d_n, int end_bit, char* temp, hipStream_t stream);
int radix_sort_pairs_u32(uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_in, uint32_t* vals_out,
	size_t n, int end_bit, char* temp, hipStream_t stream);

} // namespace adgs
