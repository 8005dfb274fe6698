// Device-wide exclusive scan and stable LSD radix sort for gfx950 (wave64).
//
// These replace the reference's cub::DeviceScan::InclusiveSum and
// cub::DeviceRadixSort::SortPairs calls (RAST/cuda_rasterizer/rasterizer_impl.cu:284,310-315;
// KNN/simple_knn.cu:210-213).  Both are HBM-streaming passes:
//   scan : reads n*4 B twice, writes n*4 B                       (~12 B/item)
//   sort : per 8-bit pass reads keys twice + values once, writes both
//          (u64 key + u32 value: 8 + 12 + 12 = 32 B/item/pass)
#include "common.h"
#include <cstdlib>

namespace adgs {

constexpr int PB = 256;            // threads per block
constexpr int SCAN_ITEMS = 8;      // items per thread
constexpr int SCAN_TILE = PB * SCAN_ITEMS;
constexpr int SPB = 512;                   // threads per sort block
constexpr int SORT_ROUNDS = 8;
constexpr int SORT_TILE = SPB * SORT_ROUNDS;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
	for (int off = 1; off < WAVE; off <<= 1) {
		uint32_t o = __shfl_up(v, off, WAVE);
		if (lane >= off) v += o;
	}
	return v;
}

// mode 0: write per-block sums only.  mode 1: write exclusive scan (+ block offset).  mode 2: as mode 1, with block_offsets
// holding the UNSCANNED block sums (each block sums its predecessors itself).
template <int MODE>
__global__ void __launch_bounds__(PB) scan_tile_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
	size_t n, const uint32_t* __restrict__ block_offsets, uint32_t* __restrict__ block_sums,
	const uint32_t* __restrict__ aux_in, unsigned long long* __restrict__ aux_total) {
	__shared__ uint32_t wave_tot[PB / WAVE];
	const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
	const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)tid * SCAN_ITEMS;
	if (MODE == 0 && aux_in) {
		// side job of the block-sum pass: the plain total of a second array of the same length, one atomic per
		// block, spread over SCAN_AUX_SLOTS counters (same-address atomics serialise in the L2)
		__shared__ unsigned long long aux_w[PB / WAVE];
		unsigned long long s = 0;
#pragma unroll
		for (int k = 0; k < SCAN_ITEMS; k++) if (base + k < n) s += aux_in[base + k];
#pragma unroll
		for (int off = WAVE / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
		if (lane == 0) aux_w[wid] = s;
		__syncthreads();
		if (tid == 0) {
			unsigned long long t = 0;
			for (int w = 0; w < PB / WAVE; w++) t += aux_w[w];
			if (t) atomicAdd(aux_total + (blockIdx.x % SCAN_AUX_SLOTS), t);
		}
	}
	uint32_t v[SCAN_ITEMS];
	uint32_t tsum = 0;
	if (base + SCAN_ITEMS <= n) {
		const uint4* p = reinterpret_cast<const uint4*>(in + base);
		uint4 a = p[0], b = p[1];
		v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
	} else {
#pragma unroll
		for (int k = 0; k < SCAN_ITEMS; k++) v[k] = (base + k < n) ? in[base + k] : 0u;
	}
#pragma unroll
	for (int k = 0; k < SCAN_ITEMS; k++) tsum += v[k];
	uint32_t incl = wave_incl_scan(tsum, lane);
	if (lane == WAVE - 1) wave_tot[wid] = incl;
	__syncthreads();
	uint32_t wave_prefix = 0, block_total = 0;
#pragma unroll
	for (int w = 0; w < PB / WAVE; w++) {
		uint32_t t = wave_tot[w];
		if (w < wid) wave_prefix += t;
		block_total += t;
	}
	if (MODE == 0) {
		if (tid == 0) block_sums[blockIdx.x] = block_total;
		return;
	}
	uint32_t block_off = 0;
	if (MODE == 2) {
		// every block adds up the sums of the blocks before it by itself (a few hundred values from the L2): no separate
		// launch for the scan of the block sums
		__shared__ uint32_t s_part[PB / WAVE];
		uint32_t p = 0;
		for (uint32_t b = tid; b < blockIdx.x; b += PB) p += block_offsets[b];
#pragma unroll
		for (int off = WAVE / 2; off > 0; off >>= 1) p += __shfl_xor(p, off, WAVE);
		if (lane == 0) s_part[wid] = p;
		__syncthreads();
#pragma unroll
		for (int w = 0; w < PB / WAVE; w++) block_off += s_part[w];
	} else if (block_offsets) block_off = block_offsets[blockIdx.x];
	uint32_t run = block_off + wave_prefix + (incl - tsum);
	uint32_t o[SCAN_ITEMS];
#pragma unroll
	for (int k = 0; k < SCAN_ITEMS; k++) { o[k] = run; run += v[k]; }
	if (base + SCAN_ITEMS <= n) {
		uint4* q = reinterpret_cast<uint4*>(out + base);
		q[0] = make_uint4(o[0], o[1], o[2], o[3]);
		q[1] = make_uint4(o[4], o[5], o[6], o[7]);
	} else {
#pragma unroll
		for (int k = 0; k < SCAN_ITEMS; k++) if (base + k < n) out[base + k] = o[k];
	}
}

static size_t scan_blocks(size_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }

size_t scan_temp_bytes(size_t n) {
	size_t total = 0;
	while (n > (size_t)SCAN_TILE) {
		n = scan_blocks(n);
		total += align_up(n * sizeof(uint32_t), 256);
	}
	return total + 256;
}

int exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, char* temp, hipStream_t stream) {
	return exclusive_scan_u32_sum(in, out, n, temp, nullptr, nullptr, stream);
}
// exclusive scan of `in` plus, when aux_in is given, sum(aux_in[0..n)) added into the SCAN_AUX_SLOTS counters
// aux_total[] (pre-zeroed by the caller, who adds them up)
int exclusive_scan_u32_sum(const uint32_t* in, uint32_t* out, size_t n, char* temp, const uint32_t* aux_in, unsigned long long* aux_total, hipStream_t stream) {
	if (n == 0) return 0;
	const size_t nb = scan_blocks(n);
	const uint32_t* no_aux = nullptr; unsigned long long* no_tot = nullptr;
	if (nb == 1) {
		if (aux_in) {
			hipLaunchKernelGGL(scan_tile_kernel<0>, dim3(1), dim3(PB), 0, stream, in, (uint32_t*)nullptr, n, (const uint32_t*)nullptr, reinterpret_cast<uint32_t*>(temp), aux_in, aux_total);
			ADGS_HIP_CHECK(hipGetLastError());
		}
		hipLaunchKernelGGL(scan_tile_kernel<1>, dim3(1), dim3(PB), 0, stream, in, out, n, (const uint32_t*)nullptr, (uint32_t*)nullptr, no_aux, no_tot);
		ADGS_HIP_CHECK(hipGetLastError());
		return 0;
	}
	uint32_t* sums = reinterpret_cast<uint32_t*>(temp);
	char* next_temp = temp + align_up(nb * sizeof(uint32_t), 256);
	hipLaunchKernelGGL(scan_tile_kernel<0>, dim3((unsigned)nb), dim3(PB), 0, stream, in, (uint32_t*)nullptr, n, (const uint32_t*)nullptr, sums, aux_in, aux_total);
	ADGS_HIP_CHECK(hipGetLastError());
	if (nb <= 4096) {            // two launches: the blocks of the second pass add up the preceding block sums themselves
		hipLaunchKernelGGL(scan_tile_kernel<2>, dim3((unsigned)nb), dim3(PB), 0, stream, in, out, n, (const uint32_t*)sums, (uint32_t*)nullptr, no_aux, no_tot);
		ADGS_HIP_CHECK(hipGetLastError());
		return 0;
	}
	if (exclusive_scan_u32(sums, sums, nb, next_temp, stream) != 0) return -1;
	hipLaunchKernelGGL(scan_tile_kernel<1>, dim3((unsigned)nb), dim3(PB), 0, stream, in, out, n, (const uint32_t*)sums, (uint32_t*)nullptr, no_aux, no_tot);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

// ------------------------------------------------------------------ radix sort
template <typename KeyT>
__global__ void __launch_bounds__(SPB) sort_hist_kernel(const KeyT* __restrict__ keys, size_t n, const uint32_t* __restrict__ d_n, int shift, uint32_t mask,
	uint32_t* __restrict__ block_hist, uint32_t nblocks) {
	__shared__ uint32_t hist[256];
	const int tid = threadIdx.x;
	if (d_n) n = min(n, (size_t)*d_n);        // device-side count (speculative capacity launch): n is the capacity
	if (tid < 256) hist[tid] = 0;
	__syncthreads();
	const size_t tile0 = (size_t)blockIdx.x * SORT_TILE;
	KeyT kk[SORT_ROUNDS];
#pragma unroll
	for (int r = 0; r < SORT_ROUNDS; r++) {
		const size_t i = tile0 + (size_t)r * SPB + tid;
		kk[r] = keys[min(i, n ? n - 1 : (size_t)0)];      // unconditional, clamped (the buffer holds at least one key): all loads of the thread in flight
	}
#pragma unroll
	for (int r = 0; r < SORT_ROUNDS; r++) {
		const size_t i = tile0 + (size_t)r * SPB + tid;
		if (i < n) atomicAdd(&hist[(uint32_t)(kk[r] >> shift) & mask], 1u);
	}
	__syncthreads();
	if (tid < 256) block_hist[(size_t)tid * nblocks + blockIdx.x] = hist[tid];
}

// Exclusive scan of every digit row of the histogram matrix H[256][nblocks] in place (one wave per row) + the row totals.
// Together with the 256-entry scan of the totals that every scatter block does for itself, this replaces a three-launch
// device-wide scan of the matrix by one launch.
constexpr size_t SORT_ROW_SCAN_MAX_BLOCKS = 640;         // longer rows (> 2.6 M pairs) go through the generic multi-block scan (measured crossover)
__global__ void __launch_bounds__(256) sort_row_scan_kernel(uint32_t* __restrict__ block_hist, uint32_t nblocks, uint32_t* __restrict__ totals) {
	const int lane = threadIdx.x & (WAVE - 1);
	const uint32_t row = blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;        // digit
	uint32_t* h = block_hist + (size_t)row * nblocks;
	uint32_t carry = 0;
	for (uint32_t base = 0; base < nblocks; base += 8 * WAVE) {
		uint32_t v[8];
#pragma unroll
		for (int k = 0; k < 8; k++) v[k] = h[min(base + k * WAVE + lane, nblocks - 1)];
#pragma unroll
		for (int k = 0; k < 8; k++) if (base + k * WAVE + lane >= nblocks) v[k] = 0u;
#pragma unroll
		for (int k = 0; k < 8; k++) {
			const uint32_t i = base + k * WAVE + lane;
			const uint32_t incl = wave_incl_scan(v[k], lane);
			if (i < nblocks) h[i] = carry + incl - v[k];
			carry += __shfl(incl, WAVE - 1, WAVE);
		}
	}
	if (lane == 0) totals[row] = carry;
}

// The tile is ranked round by round (wave-level match-any + per-wave counters, stable), then permuted into digit order
// in LDS so that the global stores of a digit's run are contiguous: scattering 8-byte keys straight from registers
// dirties one 32-byte sector per key (measured: 13 us of a 41-us pass at 2 M pairs).
template <typename KeyT>
__global__ void __launch_bounds__(SPB) sort_scatter_kernel(const KeyT* __restrict__ keys_in, KeyT* __restrict__ keys_out,
	const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ vals_out, size_t n, const uint32_t* __restrict__ d_n, int shift, uint32_t mask,
	const uint32_t* __restrict__ block_hist_scanned, uint32_t nblocks, const uint32_t* __restrict__ totals) {
	constexpr int NW = SPB / WAVE;
	__shared__ uint32_t global_base[256];
	__shared__ uint32_t running[256];
	__shared__ uint32_t digit_start[256];
	__shared__ uint32_t s_scan[256 / WAVE];
	__shared__ uint32_t wave_cnt[NW][256];
	__shared__ uint32_t wave_base[NW][256];
	__shared__ KeyT s_keys[SORT_TILE];             // the tile in digit order (reused for the values)
	if (d_n) n = min(n, (size_t)*d_n);
	const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
	const size_t tile0 = (size_t)blockIdx.x * SORT_TILE;
	if (tile0 >= n) return;                        // block-uniform
	{	// base of digit d for this block = (keys with a smaller digit) + (keys with digit d in earlier blocks)
		const uint32_t tot = tid < 256 ? totals[tid] : 0u;
		const uint32_t incl = wave_incl_scan(tot, lane);
		if (tid < 256 && lane == WAVE - 1) s_scan[wid] = incl;
		__syncthreads();
		if (tid < 256) {
			uint32_t off = 0;
#pragma unroll
			for (int w = 0; w < 256 / WAVE; w++) if (w < wid) off += s_scan[w];
			global_base[tid] = off + (incl - tot) + block_hist_scanned[(size_t)tid * nblocks + blockIdx.x];
			running[tid] = 0;
		}
		__syncthreads();
	}
	const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (WAVE - lane));
	KeyT keys[SORT_ROUNDS]; uint32_t vals[SORT_ROUNDS], lrank[SORT_ROUNDS];
#pragma unroll
	for (int r = 0; r < SORT_ROUNDS; r++) {
		const size_t i = tile0 + (size_t)r * SPB + tid;
		const size_t ic = min(i, n ? n - 1 : (size_t)0);      // unconditional, clamped: all loads of the thread in flight
		keys[r] = keys_in[ic];
		vals[r] = vals_in[ic];
	}
#pragma unroll
	for (int r = 0; r < SORT_ROUNDS; r++) {
		const size_t i = tile0 + (size_t)r * SPB + tid;
		const bool valid = i < n;
		const uint32_t d = (uint32_t)(keys[r] >> shift) & mask;
		uint64_t peers = __ballot(valid);
#pragma unroll
		for (int b = 0; b < 8; b++) {
			const bool bit = (d >> b) & 1u;
			const uint64_t m = __ballot(valid && bit);
			peers &= bit ? m : ~m;
		}
		const uint32_t rank_in_wave = __popcll(peers & lt_mask);
		const uint32_t count = __popcll(peers);
#pragma unroll
		for (int k = 0; k < 256 / WAVE; k++) wave_cnt[wid][k * WAVE + lane] = 0;
		__syncthreads();
		if (valid && rank_in_wave == 0) wave_cnt[wid][d] = count;
		__syncthreads();
		if (tid < 256) {
			uint32_t b = running[tid];
#pragma unroll
			for (int w = 0; w < NW; w++) { wave_base[w][tid] = b; b += wave_cnt[w][tid]; }
			running[tid] = b;
		}
		__syncthreads();
		lrank[r] = wave_base[wid][d] + rank_in_wave;          // rank inside the digit's run of this tile
	}
	// start of every digit's run inside the tile: exclusive scan of the tile's histogram (= running[])
	{
		const uint32_t cnt = tid < 256 ? running[tid] : 0u;
		const uint32_t incl = wave_incl_scan(cnt, lane);
		if (tid < 256 && lane == WAVE - 1) s_scan[wid] = incl;
		__syncthreads();
		if (tid < 256) {
			uint32_t off = 0;
#pragma unroll
			for (int w = 0; w < 256 / WAVE; w++) if (w < wid) off += s_scan[w];
			digit_start[tid] = off + (incl - cnt);
		}
	}
	__syncthreads();
	const uint32_t tile_n = (uint32_t)min((size_t)SORT_TILE, n - tile0);
	uint32_t pos[SORT_ROUNDS], dsts[SORT_ROUNDS];
#pragma unroll
	for (int r = 0; r < SORT_ROUNDS; r++) {
		const uint32_t d = (uint32_t)(keys[r] >> shift) & mask;
		pos[r] = digit_start[d] + lrank[r];
		if (tile0 + (size_t)r * SPB + tid < n) s_keys[pos[r]] = keys[r];
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < SORT_ROUNDS; k++) {
		const uint32_t t = tid + k * SPB;
		if (t < tile_n) {
			const KeyT key = s_keys[t];
			const uint32_t d = (uint32_t)(key >> shift) & mask;
			dsts[k] = global_base[d] + (t - digit_start[d]);
			keys_out[dsts[k]] = key;
		}
	}
	__syncthreads();
	uint32_t* s_vals = reinterpret_cast<uint32_t*>(s_keys);
#pragma unroll
	for (int r = 0; r < SORT_ROUNDS; r++) if (tile0 + (size_t)r * SPB + tid < n) s_vals[pos[r]] = vals[r];
	__syncthreads();
#pragma unroll
	for (int k = 0; k < SORT_ROUNDS; k++) {
		const uint32_t t = tid + k * SPB;
		if (t < tile_n) vals_out[dsts[k]] = s_vals[t];
	}
}

static size_t sort_blocks(size_t n) { return (n + SORT_TILE - 1) / SORT_TILE; }

size_t sort_temp_bytes(size_t n) {
	const size_t nb = sort_blocks(n);
	return 1024 /* digit totals */ + align_up(256 * nb * sizeof(uint32_t), 256) + scan_temp_bytes(256 * nb) + 256;
}

template <typename KeyT>
static int radix_sort_pairs(KeyT* keys_in, KeyT* keys_out, uint32_t* vals_in, uint32_t* vals_out,
	size_t n, const uint32_t* d_n, int end_bit, char* temp, hipStream_t stream) {
	if (n == 0) return 0;
	const size_t nb = sort_blocks(n);
	uint32_t* totals = reinterpret_cast<uint32_t*>(temp);
	uint32_t* block_hist = reinterpret_cast<uint32_t*>(temp + 1024);
	char* scan_temp = temp + 1024 + align_up(256 * nb * sizeof(uint32_t), 256);
	const bool row_scan = nb <= SORT_ROW_SCAN_MAX_BLOCKS;
	// long rows: the whole matrix goes through the generic flat scan, which already contains the digit offsets -> totals = 0
	if (!row_scan) ADGS_HIP_CHECK(hipMemsetAsync(totals, 0, 256 * sizeof(uint32_t), stream));
	const int passes = (end_bit + 7) / 8;
	KeyT* kin = keys_in; KeyT* kout = keys_out; uint32_t* vin = vals_in; uint32_t* vout = vals_out;
	// make the final pass land in keys_out/vals_out
	if (passes % 2 == 0) { kin = keys_in; kout = keys_out; }
	for (int p = 0; p < passes; p++) {
		const int shift = p * 8;
		const int bits = (end_bit - shift) < 8 ? (end_bit - shift) : 8;
		const uint32_t mask = (1u << bits) - 1u;
		hipLaunchKernelGGL(sort_hist_kernel<KeyT>, dim3((unsigned)nb), dim3(SPB), 0, stream, (const KeyT*)kin, n, d_n, shift, mask, block_hist, (uint32_t)nb);
		ADGS_HIP_CHECK(hipGetLastError());
		if (row_scan) {
			hipLaunchKernelGGL(sort_row_scan_kernel, dim3(256 / (256 / WAVE)), dim3(256), 0, stream, block_hist, (uint32_t)nb, totals);
			ADGS_HIP_CHECK(hipGetLastError());
		} else if (exclusive_scan_u32(block_hist, block_hist, 256 * nb, scan_temp, stream) != 0) return -1;
		hipLaunchKernelGGL(sort_scatter_kernel<KeyT>, dim3((unsigned)nb), dim3(SPB), 0, stream, (const KeyT*)kin, kout,
			(const uint32_t*)vin, vout, n, d_n, shift, mask, (const uint32_t*)block_hist, (uint32_t)nb, (const uint32_t*)totals);
		ADGS_HIP_CHECK(hipGetLastError());
		KeyT* tk = kin; kin = kout; kout = tk;
		uint32_t* tv = vin; vin = vout; vout = tv;
	}
	// after the loop `kin` holds the sorted data
	if (kin != keys_out) {
		ADGS_HIP_CHECK(hipMemcpyAsync(keys_out, kin, n * sizeof(KeyT), hipMemcpyDeviceToDevice, stream));
		ADGS_HIP_CHECK(hipMemcpyAsync(vals_out, vin, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
	}
	return 0;
}

int radix_sort_pairs_u64(uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_in, uint32_t* vals_out,
	size_t n, int end_bit, char* temp, hipStream_t stream) {
	return radix_sort_pairs<uint64_t>(keys_in, keys_out, vals_in, vals_out, n, nullptr, end_bit, temp, stream);
}
// n_cap sizes the launch and the temporaries; the number of pairs actually sorted is min(*d_n, n_cap),
// read on the device (lets the host enqueue the sort before it knows the count).
int radix_sort_pairs_u64_dn(uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_in, uint32_t* vals_out,
	size_t n_cap, const uint32_t* d_n, int end_bit, char* temp, hipStream_t stream) {
	return radix_sort_pairs<uint64_t>(keys_in, keys_out, vals_in, vals_out, n_cap, d_n, end_bit, temp, stream);
}
int radix_sort_pairs_u32(uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_in, uint32_t* vals_out,
	size_t n, int end_bit, char* temp, hipStream_t stream) {
	return radix_sort_pairs<uint32_t>(keys_in, keys_out, vals_in, vals_out, n, nullptr, end_bit, temp, stream);
}

} // namespace adgs
