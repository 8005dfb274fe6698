// Issue costs and hazards of the instruction idioms the blend kernels are made of, gfx950, measured per wave with s_memtime:
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/issue_hazards.hip -o /tmp/issue_hazards && /tmp/issue_hazards
// Every kernel runs ITER iterations of a fixed body; the table gives shader cycles per BODY and SIMD (wave life / waves per SIMD) and the
// wall-time figure next to it, at 1 / 2 / 4 / 5 / 8 waves per SIMD (256-thread workgroups: one wave per SIMD each).
// Questions (VERDICT r3, item 1c and the per-entry header of render_v2.hip):
//   * v_cndmask_b32 alone: 22.7 cycles in profiles/r03/valu_rates.txt against 3.3 paired with v_cmp -- which is the artefact?
//   * what does a lane mask cost when it travels VALU -> SGPR -> SALU (v_cmp_e64 + s_and_b64) and SALU -> VALU (s_and vcc + v_cndmask)?
//   * do SALU instructions of a wave overlap with its VALU instructions, or does every instruction cost the wave an issue slot?
//   * latency of a broadcast ds_read_b128 with an immediate wait (the per-entry Splat reads), and how much of it 8 reads in flight hide
//   * do v_exp_f32 (quarter rate) and v_mfma_f32_4x4x1 overlap with plain fp32 VALU work of the same wave / of other waves?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ITER = 2048;

typedef float f4 __attribute__((ext_vector_type(4)));

#define R2(X) X X
#define R4(X) X X X X
#define R8(X) R4(X) R4(X)

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, float seed, unsigned long long* clk) {
	__shared__ f4 s_buf[256];
	s_buf[threadIdx.x] = f4{ seed, seed * 2.f, seed * 3.f, 1.f };
	__syncthreads();
	float a[8], b = seed + threadIdx.x * 1e-7f, c = 1.0f - seed;
	f4 acc4[8];
#pragma unroll
	for (int i = 0; i < 8; i++) { a[i] = seed * (i + 1); acc4[i] = f4{ a[i], 0.f, 1.f, 2.f }; }
	uint32_t lds_addr = (uint32_t)(uintptr_t)(&s_buf[0]) + 16u * (threadIdx.x >> 6);      // wave-uniform address: a broadcast read
	asm volatile("s_mov_b32 s20, 0x55555555\n\ts_mov_b32 s21, 0x55555555\n\ts_mov_b32 vcc_lo, 0x33333333\n\ts_mov_b32 vcc_hi, 0x33333333\n\ts_mov_b64 s[48:49], -1\n\ts_mov_b32 s50, 0x3b808081\n\ts_mov_b64 s[22:23], -1\n\ts_mov_b64 s[24:25], -1\n\ts_mov_b64 s[26:27], -1"
		::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s48", "s49", "s50", "vcc");
	const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
	for (int it = 0; it < ITER; it++) {
		if (KIND == 0) {         // 32 independent v_fma_f32 (reference)
			R4(asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
				"v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
				: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));)
		} else if (KIND == 1) {  // 32 dependent v_fma_f32 (one chain)
			R4(asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"
				"v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2"
				: "+v"(a[0]) : "v"(b), "v"(c));)
		} else if (KIND == 2) {  // 32 v_cndmask_b32_e64 with an SGPR-pair mask set before the loop, distinct destinations
			R4(asm volatile("v_cndmask_b32_e64 %0, %8, %9, s[20:21]\n\tv_cndmask_b32_e64 %1, %8, %9, s[20:21]\n\tv_cndmask_b32_e64 %2, %8, %9, s[20:21]\n\tv_cndmask_b32_e64 %3, %8, %9, s[20:21]\n\t"
				"v_cndmask_b32_e64 %4, %8, %9, s[20:21]\n\tv_cndmask_b32_e64 %5, %8, %9, s[20:21]\n\tv_cndmask_b32_e64 %6, %8, %9, s[20:21]\n\tv_cndmask_b32_e64 %7, %8, %9, s[20:21]"
				: "=v"(a[0]), "=v"(a[1]), "=v"(a[2]), "=v"(a[3]), "=v"(a[4]), "=v"(a[5]), "=v"(a[6]), "=v"(a[7]) : "v"(b), "v"(c));)
		} else if (KIND == 3) {  // 32 v_cndmask_b32_e32 with vcc set before the loop (VOP2 form), dst = src0
			R4(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\t"
				"v_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc"
				: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));)
		} else if (KIND == 4) {  // 32 v_cmp_lt_f32_e64 into four rotating SGPR pairs (no consumer)
			R8(asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %4\n\tv_cmp_lt_f32_e64 s[22:23], %1, %4\n\tv_cmp_lt_f32_e64 s[24:25], %2, %4\n\tv_cmp_lt_f32_e64 s[26:27], %3, %4"
				:: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
		} else if (KIND == 5) {  // 32 independent s_and_b64 (SALU only)
			R8(asm volatile("s_and_b64 s[40:41], s[20:21], s[22:23]\n\ts_and_b64 s[42:43], s[22:23], s[24:25]\n\ts_and_b64 s[44:45], s[24:25], s[26:27]\n\ts_and_b64 s[46:47], s[26:27], s[20:21]"
				::: "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "scc");)
		} else if (KIND == 6) {  // 16 v_fma + 16 independent s_and, interleaved: do SALU and VALU of one wave overlap?
			R4(asm volatile("v_fma_f32 %0, %0, %4, %5\n\ts_and_b64 s[40:41], s[20:21], s[22:23]\n\tv_fma_f32 %1, %1, %4, %5\n\ts_and_b64 s[42:43], s[22:23], s[24:25]\n\t"
				"v_fma_f32 %2, %2, %4, %5\n\ts_and_b64 s[44:45], s[24:25], s[26:27]\n\tv_fma_f32 %3, %3, %4, %5\n\ts_and_b64 s[46:47], s[26:27], s[20:21]"
				: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b), "v"(c) : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "scc");)
		} else if (KIND == 7) {  // the header idiom: v_cmp_e64 -> s_and_b64 on its result, back to back, 4 rotating pairs (8 x [cmp, and] x 4 = 32 pairs... 8 here)
			R2(asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %4\n\ts_and_b64 s[40:41], s[20:21], s[48:49]\n\tv_cmp_lt_f32_e64 s[22:23], %1, %4\n\ts_and_b64 s[42:43], s[22:23], s[48:49]\n\t"
				"v_cmp_lt_f32_e64 s[24:25], %2, %4\n\ts_and_b64 s[44:45], s[24:25], s[48:49]\n\tv_cmp_lt_f32_e64 s[26:27], %3, %4\n\ts_and_b64 s[46:47], s[26:27], s[48:49]"
				:: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "scc");)
		} else if (KIND == 8) {  // the same 8 pairs with the four s_and hoisted behind the four v_cmp + 4 v_fma (distance hides the hazard?)
			R2(asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %4\n\tv_cmp_lt_f32_e64 s[22:23], %1, %4\n\tv_cmp_lt_f32_e64 s[24:25], %2, %4\n\tv_cmp_lt_f32_e64 s[26:27], %3, %4\n\t"
				"v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t"
				"s_and_b64 s[40:41], s[20:21], s[48:49]\n\ts_and_b64 s[42:43], s[22:23], s[48:49]\n\ts_and_b64 s[44:45], s[24:25], s[48:49]\n\ts_and_b64 s[46:47], s[26:27], s[48:49]"
				: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b), "v"(c) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "scc");)
		} else if (KIND == 9) {  // the strip idiom: v_cmp vcc; s_and vcc, vcc, mask; v_cndmask x2 (SALU-written vcc read by VALU), x4
			R4(asm volatile("v_cmp_lt_f32 vcc, %0, %2\n\ts_and_b64 vcc, vcc, s[20:21]\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc"
				: "+v"(a[0]), "+v"(a[1]) : "v"(b) : "vcc", "scc");)
		} else if (KIND == 10) { // the same without the SALU hop: v_cmp vcc; v_cndmask x2, x4
			R4(asm volatile("v_cmp_lt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc"
				: "+v"(a[0]), "+v"(a[1]) : "v"(b) : "vcc");)
		} else if (KIND == 11) { // exec-masked strip: s_and_saveexec_b64 with an SGPR mask; 4 v_fma; s_or_b64 exec, x4
			R4(asm volatile("s_and_saveexec_b64 s[40:41], s[20:21]\n\tv_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\ts_or_b64 exec, exec, s[40:41]"
				: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b), "v"(c) : "s40", "s41", "scc");)
		} else if (KIND == 12) { // broadcast ds_read_b128 + immediate wait + 4 dependent v_fma (the per-entry Splat read), x4
			R4(asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(acc4[0]) : "v"(lds_addr) : "memory");
				asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
					: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(acc4[0].x), "v"(acc4[0].w));)
		} else if (KIND == 13) { // 4 broadcast ds_read_b128 in flight, one wait, 16 v_fma
			asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
				: "=&v"(acc4[0]), "=&v"(acc4[1]), "=&v"(acc4[2]), "=&v"(acc4[3]) : "v"(lds_addr) : "memory");
			R4(asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
				: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(acc4[0].x), "v"(acc4[3].y));)
		} else if (KIND == 14) { // 4 v_exp + 28 v_fma, independent: does the transcendental unit overlap with fp32 FMAs?
			asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3" : "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
			R4(asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\tv_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5"
				: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b), "v"(c));)
		} else if (KIND == 15) { // 8 v_mfma_f32_4x4x1_16b_f32 on 8 accumulators
			R2(asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %4, %5, %0\n\tv_mfma_f32_4x4x1_16b_f32 %1, %4, %5, %1\n\tv_mfma_f32_4x4x1_16b_f32 %2, %4, %5, %2\n\tv_mfma_f32_4x4x1_16b_f32 %3, %4, %5, %3"
				: "+v"(acc4[0]), "+v"(acc4[1]), "+v"(acc4[2]), "+v"(acc4[3]) : "v"(b), "v"(c));)
		} else if (KIND == 16) { // 8 MFMA + 24 v_fma interleaved (1 : 3): the matrix pipe next to the VALU pipe of one wave
			R2(asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %8, %9, %0\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\t"
				"v_mfma_f32_4x4x1_16b_f32 %1, %8, %9, %1\n\tv_fma_f32 %7, %7, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\t"
				"v_mfma_f32_4x4x1_16b_f32 %2, %8, %9, %2\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\t"
				"v_mfma_f32_4x4x1_16b_f32 %3, %8, %9, %3\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
				: "+v"(acc4[0]), "+v"(acc4[1]), "+v"(acc4[2]), "+v"(acc4[3]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b), "v"(c));)
		} else if (KIND == 17) { // 24 v_fma alone (the VALU part of KIND 16)
			R2(asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\tv_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\t"
				"v_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\tv_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
				: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b), "v"(c));)
		} else if (KIND == 18) { // v_readfirstlane + s_cmp + s_cselect (wave-uniform test of a vector value), x8
			R8(asm volatile("v_readfirstlane_b32 s40, %0\n\ts_cmp_lt_u32 s40, s41\n\ts_cselect_b32 s41, s40, s41" :: "v"(a[0]) : "s40", "s41", "scc");)
		} else if (KIND == 19) { // 32 v_max_f32
			R4(asm volatile("v_max_f32 %0, %0, %8\n\tv_max_f32 %1, %1, %8\n\tv_max_f32 %2, %2, %8\n\tv_max_f32 %3, %3, %8\n\tv_max_f32 %4, %4, %8\n\tv_max_f32 %5, %5, %8\n\tv_max_f32 %6, %6, %8\n\tv_max_f32 %7, %7, %8"
				: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));)
		} else if (KIND == 20) { // 32 v_med3_f32
			R4(asm volatile("v_med3_f32 %0, %0, %8, %9\n\tv_med3_f32 %1, %1, %8, %9\n\tv_med3_f32 %2, %2, %8, %9\n\tv_med3_f32 %3, %3, %8, %9\n\tv_med3_f32 %4, %4, %8, %9\n\tv_med3_f32 %5, %5, %8, %9\n\tv_med3_f32 %6, %6, %8, %9\n\tv_med3_f32 %7, %7, %8, %9"
				: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));)
		} else if (KIND == 21) { // 32 v_mul_f32 with the clamp modifier (VOP3)
			R4(asm volatile("v_mul_f32_e64 %0, %0, %8 clamp\n\tv_mul_f32_e64 %1, %1, %8 clamp\n\tv_mul_f32_e64 %2, %2, %8 clamp\n\tv_mul_f32_e64 %3, %3, %8 clamp\n\tv_mul_f32_e64 %4, %4, %8 clamp\n\tv_mul_f32_e64 %5, %5, %8 clamp\n\tv_mul_f32_e64 %6, %6, %8 clamp\n\tv_mul_f32_e64 %7, %7, %8 clamp"
				: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));)
		} else if (KIND == 22) { // 32 v_cmp_lt_f32 vcc with an SGPR operand and a literal-free form (reference for KIND 4)
			R4(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n\tv_cmp_lt_f32 vcc, %1, %8\n\tv_cmp_lt_f32 vcc, %2, %8\n\tv_cmp_lt_f32 vcc, %3, %8\n\tv_cmp_lt_f32 vcc, %4, %8\n\tv_cmp_lt_f32 vcc, %5, %8\n\tv_cmp_lt_f32 vcc, %6, %8\n\tv_cmp_lt_f32 vcc, %7, %8"
				:: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b) : "vcc");)
		} else if (KIND == 23) { // a blend-header-like mix per "strip": sub, fma, fma, mul, exp, mul, min, cmp_e64, cmp(vcc) -- 4 strips interleaved by the assembler order below
			asm volatile(
				"v_sub_f32 %0, %8, %4\n\tv_sub_f32 %1, %8, %5\n\tv_sub_f32 %2, %8, %6\n\tv_sub_f32 %3, %8, %7\n\t"
				"v_fma_f32 %4, %9, %0, %8\n\tv_fma_f32 %5, %9, %1, %8\n\tv_fma_f32 %6, %9, %2, %8\n\tv_fma_f32 %7, %9, %3, %8\n\t"
				"v_fma_f32 %4, %0, %4, %9\n\tv_fma_f32 %5, %1, %5, %9\n\tv_fma_f32 %6, %2, %6, %9\n\tv_fma_f32 %7, %3, %7, %9\n\t"
				"v_mul_f32 %0, 0x3fb8aa3b, %4\n\tv_mul_f32 %1, 0x3fb8aa3b, %5\n\tv_mul_f32 %2, 0x3fb8aa3b, %6\n\tv_mul_f32 %3, 0x3fb8aa3b, %7\n\t"
				"v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
				"v_mul_f32 %0, %9, %0\n\tv_mul_f32 %1, %9, %1\n\tv_mul_f32 %2, %9, %2\n\tv_mul_f32 %3, %9, %3\n\t"
				"v_min_f32 %0, 0x3f7d70a4, %0\n\tv_min_f32 %1, 0x3f7d70a4, %1\n\tv_min_f32 %2, 0x3f7d70a4, %2\n\tv_min_f32 %3, 0x3f7d70a4, %3\n\t"
				"v_cmp_nlt_f32_e64 s[20:21], 0, %4\n\tv_cmp_nlt_f32_e64 s[22:23], 0, %5\n\tv_cmp_nlt_f32_e64 s[24:25], 0, %6\n\tv_cmp_nlt_f32_e64 s[26:27], 0, %7\n\t"
				"v_cmp_ngt_f32_e64 s[40:41], s50, %0\n\ts_and_b64 s[20:21], s[20:21], s[40:41]\n\tv_cmp_ngt_f32_e64 s[42:43], s50, %1\n\ts_and_b64 s[22:23], s[22:23], s[42:43]\n\t"
				"v_cmp_ngt_f32_e64 s[44:45], s50, %2\n\ts_and_b64 s[24:25], s[24:25], s[44:45]\n\tv_cmp_ngt_f32_e64 s[46:47], s50, %3\n\ts_and_b64 s[26:27], s[26:27], s[46:47]\n\t"
				"s_or_b64 s[40:41], s[20:21], s[22:23]\n\ts_or_b64 s[40:41], s[40:41], s[24:25]\n\ts_or_b64 s[40:41], s[40:41], s[26:27]\n\ts_cmp_eq_u64 s[40:41], 0"
				: "+v"(acc4[0].x), "+v"(acc4[0].y), "+v"(acc4[0].z), "+v"(acc4[0].w), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b), "v"(c)
				: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s50", "scc");
		} else if (KIND == 24) { // the lean header: no power test, no clamp -- sub, fma, fma, exp (pre-scaled), mul, cmp_e64; s_or chain
			asm volatile(
				"v_sub_f32 %0, %8, %4\n\tv_sub_f32 %1, %8, %5\n\tv_sub_f32 %2, %8, %6\n\tv_sub_f32 %3, %8, %7\n\t"
				"v_fma_f32 %4, %9, %0, %8\n\tv_fma_f32 %5, %9, %1, %8\n\tv_fma_f32 %6, %9, %2, %8\n\tv_fma_f32 %7, %9, %3, %8\n\t"
				"v_fma_f32 %4, %0, %4, %9\n\tv_fma_f32 %5, %1, %5, %9\n\tv_fma_f32 %6, %2, %6, %9\n\tv_fma_f32 %7, %3, %7, %9\n\t"
				"v_exp_f32 %0, %4\n\tv_exp_f32 %1, %5\n\tv_exp_f32 %2, %6\n\tv_exp_f32 %3, %7\n\t"
				"v_mul_f32 %0, %9, %0\n\tv_mul_f32 %1, %9, %1\n\tv_mul_f32 %2, %9, %2\n\tv_mul_f32 %3, %9, %3\n\t"
				"v_cmp_ngt_f32_e64 s[20:21], s50, %0\n\tv_cmp_ngt_f32_e64 s[22:23], s50, %1\n\tv_cmp_ngt_f32_e64 s[24:25], s50, %2\n\tv_cmp_ngt_f32_e64 s[26:27], s50, %3\n\t"
				"s_or_b64 s[40:41], s[20:21], s[22:23]\n\ts_or_b64 s[42:43], s[24:25], s[26:27]\n\ts_or_b64 s[40:41], s[40:41], s[42:43]\n\ts_cmp_eq_u64 s[40:41], 0"
				: "+v"(acc4[0].x), "+v"(acc4[0].y), "+v"(acc4[0].z), "+v"(acc4[0].w), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b), "v"(c)
				: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s40", "s41", "s42", "s43", "s50", "scc");
		}
	}
	const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
	float s = 0.f;
#pragma unroll
	for (int i = 0; i < 8; i++) s += a[i] + acc4[i].x + acc4[i].y + acc4[i].z + acc4[i].w;
	if ((threadIdx.x & 63) == 0) { atomicAdd(&clk[0], c1 - c0); atomicAdd(&clk[1], r1 - r0); atomicAdd(&clk[2], 1ull); }
	if (s == 12345.678f) out[threadIdx.x] = s + (float)lds_addr;
}

template <int KIND>
static void run(const char* name, float* d_out, int n_simd) {
	unsigned long long* d_clk; CHECK(hipMalloc(&d_clk, 3 * sizeof(unsigned long long)));
	printf("%-58s", name);
	for (int waves : { 1, 2, 4, 5, 8 }) {
		hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
		const int blocks = n_simd / 4 * waves;
		for (int w = 0; w < 4; w++) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 0.5f, d_clk);
		CHECK(hipDeviceSynchronize());
		CHECK(hipMemset(d_clk, 0, 3 * sizeof(unsigned long long)));
		CHECK(hipEventRecord(e0));
		constexpr int REP = 4;
		for (int w = 0; w < REP; w++) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 0.5f, d_clk);
		CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
		float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= REP;
		unsigned long long h[3]; CHECK(hipMemcpy(h, d_clk, sizeof(h), hipMemcpyDeviceToHost));
		const double clock_ghz = (double)h[0] / ((double)h[1] * 10.0);
		const double life = (double)h[0] / (double)h[2] / ITER / waves;        // SIMD cycles per body (the waves of a SIMD share it)
		const double wall = ms * 1e-3 * clock_ghz * 1e9 / ((double)ITER * waves);
		printf(" | %dw %7.1f /%7.1f", waves, life, wall);
	}
	printf("\n");
	CHECK(hipFree(d_clk));
}

int main() {
	hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
	const int n_simd = prop.multiProcessorCount * 4;
	printf("%s: %d SIMDs.  Columns: waves per SIMD, SIMD cycles per BODY by wave life / by wall time (s_memtime ticks)\n", prop.gcnArchName, n_simd);
	float* d_out; CHECK(hipMalloc(&d_out, 4096));
	run<0>("32 v_fma_f32 independent", d_out, n_simd);
	run<1>("32 v_fma_f32 dependent chain", d_out, n_simd);
	run<2>("32 v_cndmask_b32_e64 (SGPR mask, fixed)", d_out, n_simd);
	run<3>("32 v_cndmask_b32 (vcc, fixed)", d_out, n_simd);
	run<22>("32 v_cmp_lt_f32 vcc", d_out, n_simd);
	run<4>("32 v_cmp_lt_f32_e64 -> 4 SGPR pairs", d_out, n_simd);
	run<5>("32 s_and_b64 independent", d_out, n_simd);
	run<6>("16 v_fma + 16 s_and interleaved", d_out, n_simd);
	run<7>("8 x [v_cmp_e64 ; s_and on it]", d_out, n_simd);
	run<8>("8 x the same, s_and 4 instr. later (+8 v_fma)", d_out, n_simd);
	run<9>("4 x [v_cmp vcc; s_and vcc; 2 v_cndmask]", d_out, n_simd);
	run<10>("4 x [v_cmp vcc; 2 v_cndmask]", d_out, n_simd);
	run<11>("4 x [s_and_saveexec; 4 v_fma; s_or exec]", d_out, n_simd);
	run<12>("4 x [ds_read_b128 bcast; wait; 4 v_fma]", d_out, n_simd);
	run<13>("4 ds_read_b128 in flight; wait; 16 v_fma", d_out, n_simd);
	run<14>("4 v_exp + 28 v_fma", d_out, n_simd);
	run<15>("8 v_mfma_f32_4x4x1_16b", d_out, n_simd);
	run<16>("8 v_mfma_4x4x1 + 24 v_fma interleaved", d_out, n_simd);
	run<17>("24 v_fma", d_out, n_simd);
	run<18>("8 x [v_readfirstlane; s_cmp; s_cselect]", d_out, n_simd);
	run<19>("32 v_max_f32", d_out, n_simd);
	run<20>("32 v_med3_f32", d_out, n_simd);
	run<21>("32 v_mul_f32 clamp", d_out, n_simd);
	run<23>("blend header now: 4 strips (36 VALU + 7 SALU)", d_out, n_simd);
	run<24>("blend header lean: 4 strips (24 VALU + 4 SALU)", d_out, n_simd);
	return 0;
}
