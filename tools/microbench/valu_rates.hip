// VALU issue rates on gfx950, measured: wave64 instructions per SIMD and cycle for the instruction kinds the blend kernels are made
// of (plain / packed fp32 FMA, mul, min, compare + select, exp2, rcp, DPP moves), at 1 / 2 / 4 / 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
// Every kernel runs ITER iterations of 32 independent instructions of one kind (8 accumulator chains x 4), so neither dependent-issue
// latency nor the loop overhead (2 scalar instructions per 32) limits it.  Cycles are MEASURED: every wave reads s_memtime (shader-clock
// ticks, MI355X_MICROARCH.md "s_memtime tick = shader cycle") and s_memrealtime (constant 100 MHz) around its loop; their ratio is the
// sustained shader clock under that instruction mix (the chip clocks down under VALU load: DVFS), and cycles per instruction are
// s_memtime ticks, not wall time x the nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ITER = 4096;

#define BODY8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define REP4(X) X X X X

template <int KIND>
__global__ void __launch_bounds__(256) rate_kernel(float* out, float seed, unsigned long long* clk) {
	const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
	float a[8], b = seed + threadIdx.x * 1e-7f, c = 1.0f - seed;
	typedef float f2 __attribute__((ext_vector_type(2)));
	f2 p[8], pb = { b, b }, pc = { c, c };
#pragma unroll
	for (int i = 0; i < 8; i++) { a[i] = seed * (i + 1); p[i] = f2{ a[i], a[i] + 1.f }; }
	for (int it = 0; it < ITER; it++) {
		if (KIND == 0) {        // v_fma_f32
#define I(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
			REP4(BODY8(I))
#undef I
		} else if (KIND == 1) { // v_pk_fma_f32
#define I(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pb), "v"(pc));
			REP4(BODY8(I))
#undef I
		} else if (KIND == 2) { // v_mul_f32
#define I(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
			REP4(BODY8(I))
#undef I
		} else if (KIND == 3) { // v_min_f32
#define I(k) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
			REP4(BODY8(I))
#undef I
		} else if (KIND == 4) { // v_exp_f32
#define I(k) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k]));
			REP4(BODY8(I))
#undef I
		} else if (KIND == 5) { // v_rcp_f32
#define I(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
			REP4(BODY8(I))
#undef I
		} else if (KIND == 6) { // v_cmp + v_cndmask (counted as two instructions)
#define I(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[k]) : "v"(b), "v"(c) : "vcc");
			REP4(BODY8(I))
#undef I
		} else if (KIND == 7) { // v_add_f32 with a DPP row shift
#define I(k) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]) : "v"(b));
			REP4(BODY8(I))
#undef I
		} else if (KIND == 8) { // v_pk_mul_f32
#define I(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pb));
			REP4(BODY8(I))
#undef I
		} else if (KIND == 9) { // v_pk_add_f32
#define I(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pb));
			REP4(BODY8(I))
#undef I
		} else if (KIND == 10) { // v_fmac_f32 (VOP2 form)
#define I(k) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
			REP4(BODY8(I))
#undef I
		} else if (KIND == 11) { // v_sub_f32
#define I(k) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
			REP4(BODY8(I))
#undef I
		} else if (KIND == 12) { // v_pk_fma_f32 with the scalar operand broadcast by op_sel (both halves take the low dword of src1)
#define I(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[k]) : "v"(pb), "v"(pc));
			REP4(BODY8(I))
#undef I
		} else if (KIND == 13) { // v_permlane32_swap_b32 (pairs of registers)
#define I(k) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[k]), "+v"(a[(k + 4) & 7]));
			REP4(BODY8(I))
#undef I
		} else if (KIND == 14) { // v_permlane16_swap_b32
#define I(k) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[k]), "+v"(a[(k + 4) & 7]));
			REP4(BODY8(I))
#undef I
		} else if (KIND == 15) { // v_cndmask_b32 alone (vcc fixed)
#define I(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : );
			REP4(BODY8(I))
#undef I
		} else if (KIND == 16) { // v_cmp_lt_f32 alone (writes vcc)
#define I(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");
			REP4(BODY8(I))
#undef I
		} else if (KIND == 17) { // v_mov_b32
#define I(k) asm volatile("v_mov_b32 %0, %1" : "=v"(a[k]) : "v"(b));
			REP4(BODY8(I))
#undef I
		} else if (KIND == 18) { // v_add_f32 with bank-masked DPP (row_half_mirror)
#define I(k) asm volatile("v_add_f32_dpp %0, %0, %1 row_half_mirror row_mask:0xf bank_mask:0x5" : "+v"(a[k]) : "v"(b));
			REP4(BODY8(I))
#undef I
		} else if (KIND == 19) { // v_cmp_lt_f32 writing an SGPR pair (VOP3) + s_and_b64: the lane-mask idiom of the blend kernels
#define I(k) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n\ts_and_b64 s[22:23], s[20:21], s[22:23]" : : "v"(a[k]), "v"(b) : "s20", "s21", "s22", "s23", "scc");
			REP4(BODY8(I))
#undef I
		}
	}
	float s = 0.f;
#pragma unroll
	for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
	if ((threadIdx.x & 63) == 0) { atomicAdd(&clk[0], __builtin_readcyclecounter() - c0); atomicAdd(&clk[1], wall_clock64() - r0); atomicAdd(&clk[2], 1ull); }
	if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int KIND>
static void run(const char* name, int per_pair, float* d_out, int n_simd, double ghz) {
	unsigned long long* d_clk; CHECK(hipMalloc(&d_clk, 3 * sizeof(unsigned long long)));
	for (int waves : { 1, 2, 4, 8 }) {
		hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
		// 256-thread workgroups: their four waves go to the four SIMDs of a CU, so `waves` workgroups per CU put exactly `waves` waves
		// on every SIMD (one-wave workgroups are NOT spread evenly: round 2's wall-time figures measured the placement)
		const int blocks = n_simd / 4 * waves;
		// a few back-to-back launches first: the clock settles to what this instruction mix sustains
		for (int w = 0; w < 8; w++) hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 0.5f, d_clk);
		CHECK(hipDeviceSynchronize());
		CHECK(hipMemset(d_clk, 0, 3 * sizeof(unsigned long long)));
		CHECK(hipEventRecord(e0));
		constexpr int REP = 8;
		for (int w = 0; w < REP; w++) hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 0.5f, d_clk);
		CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
		float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= REP;
		unsigned long long h[3]; CHECK(hipMemcpy(h, d_clk, sizeof(h), hipMemcpyDeviceToHost));
		const double insts_per_wave = (double)ITER * 32 * per_pair;
		const double clock_ghz = (double)h[0] / ((double)h[1] * 10.0);            // ticks per 10 ns
		const double cyc_per_inst = (double)h[0] / (double)h[2] / insts_per_wave / waves;      // a SIMD's cycles per instruction it retired (waves share it)
		const double wall_cyc = ms * 1e-3 * clock_ghz * 1e9 / (insts_per_wave * waves);      // wall time x sustained clock: includes launch ramp and tail
		printf("%-34s waves/SIMD %d: %7.3f ms  clock %.3f GHz  %.2f cycles per wave64 instruction and SIMD (wave life) / %.2f (wall) = %.3f inst/SIMD/cycle\n",
			name, waves, ms, clock_ghz, cyc_per_inst, wall_cyc, 1.0 / cyc_per_inst);
	}
	CHECK(hipFree(d_clk));
}

int main() {
	hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
	const int n_simd = prop.multiProcessorCount * 4;
	const double ghz = prop.clockRate * 1e-6;
	printf("%s: %d CUs, %d SIMDs, clock %.3f GHz (runtime-reported peak; sustained clocks are lower under load)\n", prop.gcnArchName, prop.multiProcessorCount, n_simd, ghz);
	float* d_out; CHECK(hipMalloc(&d_out, 4096));
	run<0>("v_fma_f32", 1, d_out, n_simd, ghz);
	run<10>("v_fmac_f32 (VOP2)", 1, d_out, n_simd, ghz);
	run<1>("v_pk_fma_f32", 1, d_out, n_simd, ghz);
	run<12>("v_pk_fma_f32 op_sel broadcast", 1, d_out, n_simd, ghz);
	run<2>("v_mul_f32", 1, d_out, n_simd, ghz);
	run<8>("v_pk_mul_f32", 1, d_out, n_simd, ghz);
	run<9>("v_pk_add_f32", 1, d_out, n_simd, ghz);
	run<11>("v_sub_f32", 1, d_out, n_simd, ghz);
	run<3>("v_min_f32", 1, d_out, n_simd, ghz);
	run<6>("v_cmp_lt_f32 + v_cndmask_b32", 2, d_out, n_simd, ghz);
	run<7>("v_add_f32 dpp row_shr", 1, d_out, n_simd, ghz);
	run<4>("v_exp_f32", 1, d_out, n_simd, ghz);
	run<5>("v_rcp_f32", 1, d_out, n_simd, ghz);
	run<17>("v_mov_b32", 1, d_out, n_simd, ghz);
	run<15>("v_cndmask_b32", 1, d_out, n_simd, ghz);
	run<16>("v_cmp_lt_f32 (vcc)", 1, d_out, n_simd, ghz);
	run<19>("v_cmp_lt_f32 (sgpr) + s_and_b64", 1, d_out, n_simd, ghz);
	run<13>("v_permlane32_swap_b32", 1, d_out, n_simd, ghz);
	run<14>("v_permlane16_swap_b32", 1, d_out, n_simd, ghz);
	run<18>("v_add_f32 dpp bank-masked", 1, d_out, n_simd, ghz);
	return 0;
}
