// Loads and stores of streams that a launch reads or writes exactly ONCE (parameter rows in, gradient rows out, images out, upstream pixel
// gradients in, Adam's p / m / v, the last read of a per-frame accumulator line): issued NON-TEMPORAL (`nt`).  Such a line is not going to be
// asked for again before a frame's other two gigabytes have passed through the 4 MiB L2 slices and the 256 MiB MALL, and leaving it there costs
// the lines that ARE re-read (Splat lines, accumulator lines, key streams) their place; the loads themselves also land sooner
// (MI355X_MICROARCH.md: issued -> landed -18 % for read-once streams).  Measured at C3 (EXPERIMENTS.md, round 5): preprocess forward 149 -> 117 us
// per frame, preprocess backward 164 -> 148, deformation 36 + 56 -> 32 + 51, frame rate +5 ... 6 %.  NOT for data the next kernel reads (Splat
// lines, deformed positions, sorted lists) and not where several load instructions of a wave share a line through the vector L1 (12-byte rows).
#pragma once
#include "common.h"

namespace adgs {

typedef float adgs_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream4(const float4* p) {
	const adgs_v4f v = __builtin_nontemporal_load(reinterpret_cast<const adgs_v4f*>(p));
	return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_stream4(float4* p, const float4 v) {
	adgs_v4f w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
	__builtin_nontemporal_store(w, reinterpret_cast<adgs_v4f*>(p));
}
__device__ __forceinline__ float ld_stream(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_stream(float* p, const float v) { __builtin_nontemporal_store(v, p); }

} // namespace adgs
