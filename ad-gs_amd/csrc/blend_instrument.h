// Experiment-build instrumentation of the v2 blend kernels (render_v2.hip).  The shipping library defines none of these macros: every
// hook below expands to nothing.  One macro family per experiment (make -C ad-gs_amd/csrc variant TAG=<tag> DEFS=-D<macro>):
//   ADGS_PROBE         tools/blend_probe.py       per wave: shader cycles (s_memtime), 100 MHz ticks (s_memrealtime), (pixel, entry) pair counts
//   ADGS_PHASE_TIMING  tools/blend_phase_timing.py  per wave: shader cycles in the phases of the forward (key-stream scan, Splat gather +
//                      tile test, batch set-up, blend loop) and of the backward (chunk header, id + Splat gather, entry loop, of it the reduction tail)
//   ADGS_TIMELINE      tools/wave_timeline.py     per tile: start / end of its wave in 100 MHz ticks
#pragma once
#include <hip/hip_runtime.h>

namespace adgs {

#ifdef ADGS_PROBE
// [0..7] forward, [8..15] backward: shader cycles, 100 MHz ticks, waves, contributing (pixel, entry) pairs, entries evaluated, entries
// with a contributing pixel, active strips
__device__ unsigned long long g_probe[16];
#define PROBE_DECL const unsigned long long pr_c0 = __builtin_readcyclecounter(), pr_r0 = wall_clock64(); unsigned pr_pairs = 0, pr_evals = 0, pr_live = 0, pr_strips = 0
#define PROBE_EVAL() pr_evals++
#define PROBE_LIVE(actm, PPL) do { pr_live++; _Pragma("unroll") for (int k_ = 0; k_ < PPL; k_++) { pr_pairs += (unsigned)__popcll(actm[k_]); pr_strips += actm[k_] != 0ull; } } while (0)
#define PROBE_FLUSH(base, lane) do { if ((lane) == 0) { \
	atomicAdd(&g_probe[(base) + 0], __builtin_readcyclecounter() - pr_c0); atomicAdd(&g_probe[(base) + 1], wall_clock64() - pr_r0); atomicAdd(&g_probe[(base) + 2], 1ull); \
	atomicAdd(&g_probe[(base) + 3], (unsigned long long)pr_pairs); atomicAdd(&g_probe[(base) + 4], (unsigned long long)pr_evals); \
	atomicAdd(&g_probe[(base) + 5], (unsigned long long)pr_live); atomicAdd(&g_probe[(base) + 6], (unsigned long long)pr_strips); } } while (0)
#else
#define PROBE_DECL
#define PROBE_EVAL()
#define PROBE_LIVE(actm, PPL)
#define PROBE_FLUSH(base, lane)
#endif

#ifdef ADGS_PHASE_TIMING
// forward [0..15]: [0] key-stream scan, [1] Splat gather + tile test, [2] batch set-up (barrier, pool block draw), [3] blend loop, [4] whole wave,
// [5] waves, [6] round-trip cycles of the scan's loads (issue -> landed), [7] scan steps, [8] round-trip cycles of the Splat gathers, [9] gather rounds
// backward [16..31]: [16] chunk header wait, [17] id + Splat gather, [18] entry loop, [19] of it: reduction + atomic, [20] whole wave, [21] waves,
// [22] chunks, [23] entries, [24] prologue (pixel state loads)
__device__ unsigned long long g_phase[32];
#define PT_DECL(n) unsigned long long t_acc[n] = {}
#define PT(var) const unsigned long long var = __builtin_readcyclecounter()
#define PT_ACC(slot, a, b) t_acc[slot] += (b) - (a)
#define PT_ADD(slot, v) t_acc[slot] += (v)
// round trip of the loads that produced the listed registers: from `t_begin` (a PT() taken BEFORE the loads were issued -- the compiler
// puts its own s_waitcnt for the asm's operands in front of the asm, so a timer started here would start after the wait) until they landed
#define PT_WAIT_VM8(slot, cnt, t_begin, r0, r1, r2, r3, r4, r5, r6, r7) do { \
	asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) :: "memory"); \
	PT(t_w1_); t_acc[slot] += t_w1_ - (t_begin); t_acc[cnt] += 1ull; } while (0)
#define PT_FLUSH(base, n, lane) do { if ((lane) == 0) { for (int i_ = 0; i_ < (n); i_++) atomicAdd(&g_phase[(base) + i_], t_acc[i_]); } } while (0)
#else
#define PT_DECL(n)
#define PT(var)
#define PT_ACC(slot, a, b)
#define PT_ADD(slot, v)
#define PT_WAIT_VM8(slot, cnt, t_begin, r0, r1, r2, r3, r4, r5, r6, r7)
#define PT_FLUSH(base, n, lane)
#endif

#ifdef ADGS_TIMELINE
#define TL_DECL const unsigned long long tl_r0 = wall_clock64()
#define TL_STORE(lane, p_start, p_end, tile) do { if ((lane) == 0 && (p_start)) { (p_start)[tile] = (uint32_t)tl_r0; (p_end)[tile] = (uint32_t)wall_clock64(); } } while (0)
#else
#define TL_DECL
#define TL_STORE(lane, p_start, p_end, tile)
#endif

} // namespace adgs

// host-side readers of the experiment builds (read and reset)
#ifdef ADGS_PROBE
extern "C" int adgs_test_probe_read(unsigned long long* out16) {
	if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(adgs::g_probe), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
	unsigned long long z[16] = { 0 };
	if (hipMemcpyToSymbol(HIP_SYMBOL(adgs::g_probe), z, sizeof(z)) != hipSuccess) return -1;
	return 0;
}
#endif
#ifdef ADGS_PHASE_TIMING
extern "C" int adgs_test_phase_timing(unsigned long long* out32) {
	if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(adgs::g_phase), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
	unsigned long long z[32] = { 0 };
	if (hipMemcpyToSymbol(HIP_SYMBOL(adgs::g_phase), z, sizeof(z)) != hipSuccess) return -1;
	return 0;
}
#endif
