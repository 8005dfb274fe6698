// Bucket binning of the v2 pipeline for gfx950: depth-sorted per-cell Gaussian lists in four short launches, no device-wide sort.
//
// What it replaces (RAST/cuda_rasterizer/rasterizer_impl.cu:284-324 in the reference: InclusiveSum, duplicateWithKeys, a device-wide
// cub radix sort over 64-bit keys, identifyTileRanges; in this library's first v2 pipeline: a 2-launch scan, duplicate_cells, a
// 15-launch LSD radix sort of (cell | depth) keys and tile_ranges = 20 launches and ~175 us per C3 frame for 1.5 M pairs).
// The lists of different coarse cells are independent sorting problems; nothing has to be sorted device-wide:
//
//   preprocess_fwd   counts the (cell, Gaussian) pairs of its 256 Gaussians per coarse cell in LDS and writes the row
//                    counts[workgroup][cell] (no global atomics: the Gaussians of an object are neighbours in index AND on the
//                    screen, and atomics on a few hot cells serialise -- measured: 78 k returning atomics on 70 addresses = 70 us);
//   cell_colscan     one workgroup per cell: exclusive prefix of the cell's column over the workgroups, in place + the cell total;
//   cell_scan        ONE workgroup: exclusive scan of the cell totals -> cell ranges; cuts every cell's range into CHUNKS of at
//                    most GS_NMAX entries; publishes the frame totals to the host mailbox (the host sizes the binning buffer);
//   cell_scatter     every pair goes to cell_start + counts[workgroup][cell] + (LDS atomic inside the workgroup): one 16-byte
//                    record (depth bits, Gaussian id, rectangle mask) -- any order inside a cell;
//   chunk_sort       one workgroup per chunk: sorted on (32 depth bits, Gaussian index) by a 4-pass LSD radix sort that never
//                    leaves the CU (keys in registers, ranks by wave match-any, exchange through LDS) + an index fix-up of
//                    equal depths: exactly the order the reference's stable sort of keys emitted in index order produces;
//   chunk_merge      cells of more than one chunk: every entry (one thread each) finds its rank in the other sorted chunks of its
//                    cell by binary search (<= 13 steps per chunk, L2-resident) and moves to its final position; one-chunk
//                    cells were final.
//
// The output is what render_fwd_v2 walks: per cell a contiguous [start, end) range of (id, rectangle mask) pairs, front to back.
#include "common.h"
#include "kernels.h"

namespace adgs {
namespace {

constexpr int CS_THREADS = MAX_CELLS;                                   // cell_scan: one cell per thread
constexpr int GS_THREADS = 1024, GS_WAVES = GS_THREADS / WAVE, GS_ITEMS = GS_NMAX / GS_THREADS;      // 16 waves per CU: one chunk per CU keeps every SIMD at 4 waves

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
	for (int off = 1; off < WAVE; off <<= 1) { const uint32_t o = __shfl_up(v, off, WAVE); if (lane >= off) v += o; }
	return v;
}

// block-wide exclusive scan of one value per thread (1024 threads); returns the exclusive prefix, *total = sum over the block
__device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t v, uint32_t* s_wave /* [16] */, uint32_t* total) {
	const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
	const uint32_t incl = wave_incl_scan_u32(v, lane);
	__syncthreads();                       // s_wave may still be read from a previous call
	if (lane == WAVE - 1) s_wave[wid] = incl;
	__syncthreads();
	uint32_t off = 0, tot = 0;
#pragma unroll
	for (int w = 0; w < CS_THREADS / WAVE; w++) { const uint32_t x = s_wave[w]; if (w < wid) off += x; tot += x; }
	*total = tot;
	return off + incl - v;
}

// counts[nblocks][ncells] -> exclusive prefix over the workgroups of every cell's column (in place), column total -> cell_count
__global__ void __launch_bounds__(256) cell_colscan_kernel(uint32_t* __restrict__ counts, int nblocks, int ncells, uint32_t* __restrict__ cell_count) {
	__shared__ uint32_t s_w[256 / WAVE];
	__shared__ uint32_t s_carry;
	const int c = blockIdx.x, tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
	if (tid == 0) s_carry = 0u;
	__syncthreads();
	for (int b0 = 0; b0 < nblocks; b0 += 256 * 8) {
		uint32_t v[8], sum = 0;
#pragma unroll
		for (int k = 0; k < 8; k++) v[k] = counts[(size_t)min(b0 + tid * 8 + k, nblocks - 1) * ncells + c];      // unconditional, clamped: all eight in flight
#pragma unroll
		for (int k = 0; k < 8; k++) { if (b0 + tid * 8 + k >= nblocks) v[k] = 0u; sum += v[k]; }
		const uint32_t incl = wave_incl_scan_u32(sum, lane);
		if (lane == WAVE - 1) s_w[wid] = incl;
		__syncthreads();
		uint32_t off = s_carry, tot = 0;
#pragma unroll
		for (int w = 0; w < 256 / WAVE; w++) { const uint32_t x = s_w[w]; if (w < wid) off += x; tot += x; }
		uint32_t run = off + incl - sum;
#pragma unroll
		for (int k = 0; k < 8; k++) { const int b = b0 + tid * 8 + k; if (b < nblocks) counts[(size_t)b * ncells + c] = run; run += v[k]; }
		__syncthreads();
		if (tid == 0) s_carry += tot;
		__syncthreads();
	}
	if (tid == 0) cell_count[c] = s_carry;
}

__global__ void __launch_bounds__(CS_THREADS) cell_scan_kernel(CellScanArgs a) {
	__shared__ uint32_t s_wave[CS_THREADS / WAVE];
	const int c = threadIdx.x;
	uint32_t n = 0;
	if (c < a.ncells) n = a.cell_count[c];
	uint32_t total, nchunks_total;
	const uint32_t start = block_excl_scan_1024(n, s_wave, &total);
	const uint32_t nch = (n + GS_NMAX - 1) / GS_NMAX;
	const uint32_t g0 = block_excl_scan_1024(nch, s_wave, &nchunks_total);
	// the fullest cell (chunks): a cell of k chunks pays k - 1 rank searches per entry in the merge, so ONE hot cell (a close-up object)
	// is a cliff the average does not show; the host keeps the next frames of such a scene on the device-wide sort (api.hip)
	__shared__ uint32_t s_maxch;
	if (c == 0) s_maxch = 0u;
	__syncthreads();
	{
		uint32_t m = nch;
#pragma unroll
		for (int off = WAVE / 2; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, WAVE));
		if ((c & (WAVE - 1)) == 0) atomicMax(&s_maxch, m);
	}
	__syncthreads();
	if (c < a.ncells) {
		a.cell_start[c] = start;
		a.cell_ranges[c] = make_uint2(start, start + n);
		for (uint32_t q = 0; q < nch; q++)
			if (g0 + q < a.max_chunks) a.chunks[g0 + q] = make_uint4(start + q * GS_NMAX, min(start + (q + 1) * GS_NMAX, start + n), start, start + n);
	}
	if (c == 0) {
		a.cell_start[a.ncells] = total;
		const uint32_t over = nchunks_total > a.max_chunks ? 1u : 0u;
		unsigned long long fine = 0ull;
		for (int k = 0; k < SCAN_AUX_SLOTS; k++) fine += a.fine_total[k];
		// the binning and blend launches behind this one were enqueued against a capacity: do the totals fit?
		const uint32_t nofit = (over || total > a.cap_cells || fine > a.cap_fine) ? 1u : 0u;
		a.d_counts[0] = total; a.d_counts[1] = nchunks_total; a.d_counts[2] = over; a.d_counts[3] = nofit;
		a.box->r_cells = total; a.box->r_fine = fine; a.box->oversize = over; a.box->n_groups = nchunks_total; a.box->overflow = nofit; a.box->max_cell_chunks = s_maxch;
		if (nofit) a.box->overflow_count = a.box->overflow_count + 1u;
		a.box->cap_cells = a.cap_cells; a.box->cap_fine = a.cap_fine;
		__threadfence_system();
		a.box->seq = a.seq;                       // published last: the host spins on it
	}
}

__global__ void __launch_bounds__(256) cell_scatter_kernel(int P, const uint4* __restrict__ dupinfo, const uint32_t* __restrict__ cell_start,
	const uint32_t* __restrict__ counts, uint4* __restrict__ rec_u, uint32_t cap, int cell_tiles, int cgx, int ncells, uint32_t* __restrict__ pool_cursor) {
	__shared__ uint32_t s_cnt[MAX_CELLS];
	__shared__ uint32_t s_base[MAX_CELLS];
	const int idx = blockIdx.x * blockDim.x + threadIdx.x;
	if (idx == 0) *pool_cursor = 0u;              // bookkeeping reset for the blend forward that follows on this stream
	// this workgroup's slice of every cell's range: cell start + pairs of the workgroups before it (cell_colscan)
	const uint4 d = dupinfo[min(idx, P - 1)];      // (rect min, rect max, depth bits, -): one coalesced 16-byte load, requested before the prologue's loads and its barrier
	for (int c = threadIdx.x; c < ncells; c += 256) { s_cnt[c] = 0u; s_base[c] = cell_start[c] + counts[(size_t)blockIdx.x * ncells + c]; }
	__syncthreads();
	if (idx >= P) return;
	const uint32_t minx = d.x & 0xFFFFu, miny = d.x >> 16, maxx = d.y & 0xFFFFu, maxy = d.y >> 16;
	if (maxx <= minx || maxy <= miny) return;
	const uint32_t c0x = minx / cell_tiles, c1x = (maxx - 1) / cell_tiles, c0y = miny / cell_tiles, c1y = (maxy - 1) / cell_tiles;
	for (uint32_t y = c0y; y <= c1y; y++)
		for (uint32_t x = c0x; x <= c1x; x++) {
			const uint32_t c = y * cgx + x;
			const uint32_t pos = s_base[c] + atomicAdd(s_cnt + c, 1u);
			// which tile rows / columns OF THIS CELL the Gaussian's rectangle covers: the blend forward runs its rectangle test on
			// these 4 bytes and gathers the Splat line only of candidates that pass it
			const uint32_t ty0 = y * cell_tiles, tx0 = x * cell_tiles;
			const uint32_t r0 = max(miny, ty0) - ty0, r1 = min(maxy, ty0 + cell_tiles) - ty0;      // [r0, r1) within the cell
			const uint32_t q0 = max(minx, tx0) - tx0, q1 = min(maxx, tx0 + cell_tiles) - tx0;
			const uint32_t rows = ((1u << r1) - 1u) & ~((1u << r0) - 1u), cols = ((1u << q1) - 1u) & ~((1u << q0) - 1u);
			if (pos < cap) rec_u[pos] = make_uint4(d.z, (uint32_t)idx, rows | (cols << cell_tiles), 0u);     // cap: speculative capacity
		}
}

__global__ void __launch_bounds__(GS_THREADS) chunk_sort_kernel(ChunkSortArgs a) {
	__shared__ uint32_t s_a[GS_NMAX];
	__shared__ uint32_t s_b[GS_NMAX];
	__shared__ uint32_t s_wcnt[GS_WAVES][256];
	__shared__ uint32_t s_dbase[256];
	__shared__ uint32_t s_scan[256 / WAVE];
	const uint32_t g = blockIdx.x;
	if (g >= a.d_counts[1] || a.d_counts[2] != 0u) return;
	const uint4 ch = a.chunks[g];
	const uint32_t start = ch.x, n = ch.y - ch.x;
	if (n == 0 || n > (uint32_t)GS_NMAX || ch.w > a.cap) return;      // beyond the speculative capacity: the host re-runs with exact sizes
	const bool single = ch.x == ch.z && ch.y == ch.w;                  // the whole cell: the result is final
	const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
	const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (WAVE - lane));
	// wave w owns the contiguous index range [w * span, (w + 1) * span) of the chunk, 64 entries per round: the stable order of
	// the LSD passes is (wave, round, lane).  Padding entries carry the key 0xFFFFFFFF (no depth has these bits) and sort last.
	const uint32_t rounds = (n + GS_THREADS - 1) / GS_THREADS, span = rounds * WAVE;
	uint32_t key[GS_ITEMS], id[GS_ITEMS], msk[GS_ITEMS], pos[GS_ITEMS];
#pragma unroll
	for (int r = 0; r < GS_ITEMS; r++) {
		const uint32_t j = wid * span + r * WAVE + lane;
		const bool valid = (uint32_t)r < rounds && j < n;
		uint4 rec = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
		if (valid) rec = a.rec_u[start + j];
		key[r] = rec.x; id[r] = rec.y; msk[r] = rec.z;
	}
	for (int shift = 0; shift < 32; shift += 8) {
#pragma unroll
		for (int k = 0; k < 256 / WAVE; k++) s_wcnt[wid][k * WAVE + lane] = 0u;      // own wave's counters: LDS operations of a wave execute in order
#pragma unroll
		for (int r = 0; r < GS_ITEMS; r++) {
			if ((uint32_t)r < rounds) {              // block-uniform
				const uint32_t d = (key[r] >> shift) & 255u;
				uint64_t peers = ~0ull;
#pragma unroll
				for (int b = 0; b < 8; b++) {
					const bool bit = (d >> b) & 1u;
					const uint64_t m = __ballot(bit);
					peers &= bit ? m : ~m;
				}
				const uint32_t rank = __popcll(peers & lt_mask);
				uint32_t old = 0;
				if (rank == 0) { old = s_wcnt[wid][d]; s_wcnt[wid][d] = old + (uint32_t)__popcll(peers); }      // one leader per digit
				old = __shfl(old, __ffsll((unsigned long long)peers) - 1, WAVE);
				pos[r] = old + rank;                 // position inside this wave's run of digit d
			}
		}
		__syncthreads();
		if (tid < 256) {                             // digit tid: exclusive prefix over the waves + the digit's total
			uint32_t run = 0;
#pragma unroll
			for (int w = 0; w < GS_WAVES; w++) { const uint32_t c = s_wcnt[w][tid]; s_wcnt[w][tid] = run; run += c; }
			const uint32_t incl = wave_incl_scan_u32(run, lane);
			if (lane == WAVE - 1) s_scan[wid] = incl;
			s_dbase[tid] = incl - run;
		}
		__syncthreads();
		if (tid < 256) {
			uint32_t off = 0;
#pragma unroll
			for (int w = 0; w < 256 / WAVE; w++) if (w < wid) off += s_scan[w];
			s_dbase[tid] += off;
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < GS_ITEMS; r++) {
			if ((uint32_t)r < rounds) {
				const uint32_t d = (key[r] >> shift) & 255u;
				pos[r] += s_dbase[d] + s_wcnt[wid][d];
				s_a[pos[r]] = key[r]; s_b[pos[r]] = id[r];
			}
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < GS_ITEMS; r++) {
			if ((uint32_t)r < rounds) { const uint32_t j = wid * span + r * WAVE + lane; key[r] = s_a[j]; id[r] = s_b[j]; }
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < GS_ITEMS; r++) if ((uint32_t)r < rounds) s_a[pos[r]] = msk[r];
		__syncthreads();
#pragma unroll
		for (int r = 0; r < GS_ITEMS; r++) if ((uint32_t)r < rounds) msk[r] = s_a[wid * span + r * WAVE + lane];
		__syncthreads();
	}
	// ---- entries of equal depth: order by Gaussian index (the reference's stable sort of keys emitted in index order)
#pragma unroll
	for (int r = 0; r < GS_ITEMS; r++) {
		if ((uint32_t)r < rounds) { const uint32_t j = wid * span + r * WAVE + lane; s_a[j] = key[r]; s_b[j] = id[r]; }
	}
	__syncthreads();
#pragma unroll
	for (int r = 0; r < GS_ITEMS; r++) {
		if ((uint32_t)r < rounds) {
			const uint32_t j = wid * span + r * WAVE + lane;
			if (j < n) {
				const uint32_t k = key[r];
				uint32_t out = j;
				if ((j > 0 && s_a[j - 1] == k) || (j + 1 < n && s_a[j + 1] == k)) {
					uint32_t lo = j, hi = j + 1;
					while (lo > 0 && s_a[lo - 1] == k) lo--;
					while (hi < n && s_a[hi] == k) hi++;
					uint32_t less = 0;
					for (uint32_t i = lo; i < hi; i++) less += s_b[i] < id[r] ? 1u : 0u;
					out = lo + less;
				}
				if (single) a.ent_f[start + out] = make_uint2(id[r], msk[r]);
				else { a.key_s[start + out] = make_uint2(id[r], k); a.mask_s[start + out] = msk[r]; }      // as one 64-bit word: depth << 32 | id
			}
		}
	}
}

// One workgroup per chunk A of a multi-chunk cell.  A's (depth, id) keys stay in registers; every other chunk B of the cell is
// staged in LDS (64 KiB) and every key of A finds its rank in B by binary search there (13 steps of ~64 cycles instead of 13
// dependent L2 round trips of ~1 us each: the global-memory version of this kernel took 70 us at C3 and 480 us at C5).
// The (depth, index) order is total (indices are unique), so the ranks in the other chunks + the position in A are the final slot.
#ifndef ADGS_MG_THREADS
#define ADGS_MG_THREADS 1024
#endif
#ifndef ADGS_MG_SPLIT
#define ADGS_MG_SPLIT 2
#endif
constexpr int MG_THREADS = ADGS_MG_THREADS, MG_SPLIT = ADGS_MG_SPLIT, MG_PART = GS_NMAX / MG_SPLIT, MG_ITEMS = MG_PART / MG_THREADS;
// (MG_SPLIT workgroups per chunk A, each with a part of A's keys; two workgroups fit a CU, so one stages its next B from global
// memory while the other searches.  Measured at C3, threads x split: 1024 x 2 35 us, 512 x 4 38, 512 x 2 39, 1024 x 1 50, 256 x 8 52;
// 23 us of it is the staging of the B chunks.)
__global__ void __launch_bounds__(MG_THREADS) chunk_merge_kernel(ChunkSortArgs a) {
	// binary-search probes sit at power-of-two strides: one pad slot per 32 keys spreads them over all banks
	__shared__ unsigned long long s_key[GS_NMAX + GS_NMAX / 32];
	const uint32_t g = blockIdx.x / MG_SPLIT, part = blockIdx.x % MG_SPLIT;
	if (g >= a.d_counts[1] || a.d_counts[2] != 0u) return;
	uint4 ch = a.chunks[g];
	if (ch.w > a.cap || (ch.x == ch.z && ch.y == ch.w)) return;        // one-chunk cells are final already
	const int tid = threadIdx.x;
	const uint32_t chunk_x = ch.x;                                      // start of the whole chunk A (its slot in the cell)
	ch.x = min(ch.x + part * MG_PART, ch.y);                            // this workgroup's part of A
	ch.y = min(ch.x + (uint32_t)MG_PART, ch.y);
	const uint32_t n = ch.y - ch.x;
	if (n == 0) return;
	const unsigned long long* __restrict__ keys = reinterpret_cast<const unsigned long long*>(a.key_s);      // uint2 (id, depth): depth in the high half
	unsigned long long key[MG_ITEMS]; uint32_t rank[MG_ITEMS];
#pragma unroll
	for (int r = 0; r < MG_ITEMS; r++) {
		const uint32_t j = r * MG_THREADS + tid;
		key[r] = keys[ch.x + min(j, n - 1)];
		rank[r] = (ch.x - chunk_x) + j;                                  // position inside its own chunk
	}
	for (uint32_t q = ch.z; q < ch.w; q += GS_NMAX) {
		if (q == chunk_x) continue;                                     // block-uniform
		const uint32_t nb = min((uint32_t)GS_NMAX, ch.w - q);
		__syncthreads();
		{	// all loads of the chunk in flight at once, then the LDS stores (one load -> wait -> store per step serialises 16 L2 round
			// trips: the staging, not the search, was what this kernel's time went into)
			unsigned long long v[GS_NMAX / MG_THREADS];
#pragma unroll
			for (int k = 0; k < GS_NMAX / MG_THREADS; k++) v[k] = keys[q + min((uint32_t)(k * MG_THREADS + tid), nb - 1)];
#pragma unroll
			for (int k = 0; k < GS_NMAX / MG_THREADS; k++) {
				const uint32_t i = k * MG_THREADS + tid;
				if (i < nb) s_key[i + (i >> 5)] = v[k];
			}
		}
		__syncthreads();
		// Branch-free lower bound, the GS_ITEMS searches of a thread interleaved (a first version with `if (lo < hi)` and a
		// short-circuit two-word comparison compiled to one branch and two dependent LDS round trips per probe: 144 us at C3).
		// Invariant: the number of keys of B below key[r] lies in [lo, lo + len]; every step halves len (nb >= 1).
		uint32_t lo[MG_ITEMS];
#pragma unroll
		for (int r = 0; r < MG_ITEMS; r++) lo[r] = 0u;
		uint32_t len = nb;
		while (len > 1) {                                               // block-uniform trip count: <= 13
			const uint32_t half = len >> 1;
#pragma unroll
			for (int r = 0; r < MG_ITEMS; r++) {
				const uint32_t probe = lo[r] + half - 1;
				lo[r] = s_key[probe + (probe >> 5)] < key[r] ? lo[r] + half : lo[r];
			}
			len -= half;
		}
#pragma unroll
		for (int r = 0; r < MG_ITEMS; r++) lo[r] += s_key[lo[r] + (lo[r] >> 5)] < key[r] ? 1u : 0u;
#pragma unroll
		for (int r = 0; r < MG_ITEMS; r++) rank[r] += lo[r];
	}
#pragma unroll
	for (int r = 0; r < MG_ITEMS; r++) {
		const uint32_t j = r * MG_THREADS + tid;
		if (j < n) a.ent_f[ch.z + rank[r]] = make_uint2((uint32_t)key[r], a.mask_s[ch.x + j]);
	}
}

} // namespace

int launch_cell_scan(const CellScanArgs& a, hipStream_t stream) {
	hipLaunchKernelGGL(cell_scan_kernel, dim3(1), dim3(CS_THREADS), 0, stream, a);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
int launch_cell_colscan(uint32_t* counts, int nblocks, int ncells, uint32_t* cell_count, hipStream_t stream) {
	if (ncells == 0) return 0;
	hipLaunchKernelGGL(cell_colscan_kernel, dim3(ncells), dim3(256), 0, stream, counts, nblocks, ncells, cell_count);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
int launch_cell_scatter(int P, const uint4* dupinfo, const uint32_t* cell_start, const uint32_t* counts, uint4* rec_u, uint32_t cap,
	int cell_tiles, int cgx, int ncells, uint32_t* pool_cursor, hipStream_t stream) {
	if (P == 0) return 0;
	hipLaunchKernelGGL(cell_scatter_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, dupinfo, cell_start, counts, rec_u, cap, cell_tiles, cgx, ncells, pool_cursor);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
int launch_chunk_sort(const ChunkSortArgs& a, uint32_t grid, hipStream_t stream) {
	if (grid == 0) return 0;
	hipLaunchKernelGGL(chunk_sort_kernel, dim3(grid), dim3(GS_THREADS), 0, stream, a);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
int launch_chunk_merge(const ChunkSortArgs& a, uint32_t grid, hipStream_t stream) {
	if (grid == 0) return 0;
	hipLaunchKernelGGL(chunk_merge_kernel, dim3(grid * MG_SPLIT), dim3(MG_THREADS), 0, stream, a);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

} // namespace adgs
