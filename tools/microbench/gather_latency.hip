// Round-trip time of the memory accesses the forward blend kernel makes, under load, gfx950:
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/gather_latency.hip -o /tmp/gather_latency && /tmp/gather_latency
// Every wave repeats ITER dependent round trips of one kind (the next address depends on the data that came back) and reads s_memtime
// around the loop; the table gives shader cycles per round trip at 1 / 2 / 5 / 8 waves per SIMD (one-wave workgroups, like the blend
// kernels) over a 64 MiB array (the Splat array of C3) and the aggregate bandwidth.
//   kind 0: 64 lanes x 64 B from 64 random lines (the Splat gather: four dwordx4 per lane)
//   kind 1: 64 lanes x 32 B from 64 random records (the filter-record gather)
//   kind 2: 256 consecutive 8-byte entries at a random position (one key-stream super-round: four dwordx2 per lane)
//   kind 3: kind 0 with the 64 lines inside ONE random 16 KiB window (spatially sorted Gaussians)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ITER = 256;
constexpr size_t LINES = 1u << 20;      // 64-byte lines: 64 MiB

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int KIND>
__global__ void __launch_bounds__(64) k(const float4* __restrict__ buf, unsigned long long* clk, float* sink) {
	const int lane = threadIdx.x;
	uint32_t state = mix(blockIdx.x * 977u + 12345u);
	float acc = 0.f;
	const unsigned long long c0 = __builtin_readcyclecounter();
	for (int it = 0; it < ITER; it++) {
		float4 a, b, c, d;
		if (KIND == 0 || KIND == 3) {
			uint32_t line;
			if (KIND == 0) line = mix(state + lane * 0x9e3779b9u) & (LINES - 1);
			else line = ((mix(state) & (LINES - 1)) & ~255u) + (mix(state + lane) & 255u);
			const float4* p = buf + (size_t)line * 4;
			a = p[0]; b = p[1]; c = p[2]; d = p[3];
		} else if (KIND == 1) {
			const uint32_t rec = mix(state + lane * 0x9e3779b9u) & (2 * LINES - 1);
			const float4* p = buf + (size_t)rec * 2;
			a = p[0]; b = p[1]; c = a; d = b;
		} else {
			const uint32_t base = (mix(state) & (LINES - 1)) & ~63u;      // a 2 KiB run
			const float2* p = reinterpret_cast<const float2*>(buf) + (size_t)base * 8;
			const float2 e0 = p[lane], e1 = p[64 + lane], e2 = p[128 + lane], e3 = p[192 + lane];
			a = make_float4(e0.x, e0.y, e1.x, e1.y); b = make_float4(e2.x, e2.y, e3.x, e3.y); c = a; d = b;
		}
		const float s = (a.x + b.y) + (c.z + d.w);
		acc += s;
		// the next round trip depends on this one (wave-uniform: lane 0's data), as a tile's next batch depends on the previous
		state = mix(state + (uint32_t)__builtin_amdgcn_readfirstlane(__float_as_int(s)) + (uint32_t)it);
	}
	const unsigned long long c1 = __builtin_readcyclecounter();
	if (lane == 0) { atomicAdd(&clk[0], c1 - c0); atomicAdd(&clk[1], 1ull); }
	if (acc == 123.456f) sink[lane] = acc;
}

template <int KIND>
static void run(const char* name, const float4* buf, int n_simd, double bytes_per_trip) {
	unsigned long long* d_clk; CHECK(hipMalloc(&d_clk, 16)); float* sink; CHECK(hipMalloc(&sink, 1024));
	printf("%-64s", name);
	for (int waves : { 1, 2, 5, 8 }) {
		CHECK(hipMemset(d_clk, 0, 16));
		hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
		const int blocks = n_simd * waves;
		hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, buf, d_clk, sink);      // warm
		CHECK(hipDeviceSynchronize()); CHECK(hipMemset(d_clk, 0, 16));
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, buf, d_clk, sink);
		CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
		float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
		unsigned long long h[2]; CHECK(hipMemcpy(h, d_clk, 16, hipMemcpyDeviceToHost));
		const double cyc = (double)h[0] / (double)h[1] / ITER;
		const double tbs = bytes_per_trip * ITER * blocks / (ms * 1e-3) / 1e12;
		printf(" | %dw %7.0f cyc %5.2f TB/s", waves, cyc, tbs);
	}
	printf("\n");
	CHECK(hipFree(d_clk)); CHECK(hipFree(sink));
}

int main() {
	hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
	const int n_simd = prop.multiProcessorCount * 4;
	float4* buf; CHECK(hipMalloc(&buf, LINES * 64));
	std::vector<float> h(LINES * 16);
	for (size_t i = 0; i < h.size(); i++) h[i] = (float)(rand() & 0xffff) * 1e-3f;
	CHECK(hipMemcpy(buf, h.data(), LINES * 64, hipMemcpyHostToDevice));
	printf("%s: dependent round trips per wave, shader cycles per trip and aggregate bandwidth, by waves per SIMD (one-wave workgroups)\n", prop.gcnArchName);
	run<0>("64 lanes x 64 B, 64 random lines of 64 MiB (Splat gather)", buf, n_simd, 4096.0);
	run<1>("64 lanes x 32 B, 64 random records (filter-record gather)", buf, n_simd, 2048.0);
	run<2>("256 consecutive 8-byte entries (key-stream super-round)", buf, n_simd, 2048.0);
	run<3>("64 lanes x 64 B, lines inside one random 16 KiB window", buf, n_simd, 4096.0);
	return 0;
}
