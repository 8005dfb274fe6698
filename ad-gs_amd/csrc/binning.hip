// Bucket binning of the v2 pipeline for gfx950: depth-sorted per-cell Gaussian lists in five short launches, no device-wide sort.
//
// What it replaces (RAST/cuda_rasterizer/rasterizer_impl.cu:284-324 in the reference: InclusiveSum, duplicateWithKeys, a device-wide
// cub radix sort over 64-bit keys, identifyTileRanges; in this library's first v2 pipeline: a 2-launch scan, duplicate_cells, a
// 15-launch LSD radix sort of (cell | depth) keys and tile_ranges = 20 launches and ~175 us per C3 frame for 1.5 M pairs).
// The lists of different coarse cells are independent sorting problems, and a cell's list is only ever read front to back -- so it
// can be built as the concatenation of 2^lg independently sorted DEPTH SLABS.  A (cell, slab) pair is a COLUMN; the slab of a depth is
// a binary search over the cell's row of 31 bounds (ANY table contents give a monotone function of the depth, so correctness never
// depends on the bounds: only the balance does).  The bounds are the 32-quantiles of the cell's depth keys in the PREVIOUS frame this
// thread rendered, read off the sorted lists for free (round 6; rounds 3 - 5 sorted whole cells: a cell of k chunks paid k - 1 rank
// searches per entry in the merge, C5's 68 k pairs per cell went to the device-wide sort, and a counts matrix [workgroups][cells] with a
// column scan stood in front of the scatter):
//
//   (prologue)       the frame's first kernel (sh0 / bin_prepare) zeroes the column counters and takes the frame's snapshot of the bounds;
//   preprocess_fwd   writes one 16-byte record (shrunk tile rectangle, depth bits) per Gaussian;
//   bin_count        2048 Gaussians per workgroup: (column, Gaussian) pairs counted in LDS, one global atomic per touched column and
//                    workgroup (~25 k per C3 frame on ~560 addresses; one per PAIR, or per 256-Gaussian workgroup on 70 cells,
//                    serialises on the hot cells: measured 70 us in round 2);
//   col_scan         ONE workgroup: exclusive scan of the column totals -> column starts, cell ranges, the chunk table (a column of more
//                    than GS_NMAX entries is cut into chunks); publishes the frame totals to the host mailbox (the host sizes the binning buffer);
//   bin_scatter      the same enumeration again: a workgroup reserves its slice of every column with one returning atomic and writes
//                    one 16-byte record (depth bits, Gaussian id, rectangle mask) per pair -- any order inside a column;
//   chunk_bsort      one workgroup per chunk: sorted on (32 depth bits, Gaussian index) inside the CU by a histogram-equalised bucket sort
//                    (below): exactly the order the reference's stable sort of keys emitted in index order produces.  49 KiB of LDS and
//                    512 threads: three workgroups per CU, its cost follows the chunk size;
//   chunk_sort       the chunks bsort gives up on (long runs of equal depths): an LSD radix sort that never leaves the CU
//                    (keys in registers, ranks by wave match-any, exchange through LDS; passes whose digit is the same for the whole
//                    chunk are skipped) + an index fix-up of equal depths;
//   chunk_merge      columns of more than one chunk (a camera whose depths the bounds do not fit; a thread's first frame): every entry
//                    finds its rank in the other sorted chunks of its column by binary search and moves to its final position.
//
// The output is what render_fwd_v2 walks: per cell a contiguous [start, end) range of (id, rectangle mask) pairs, front to back.
#include "common.h"
#include "kernels.h"

namespace adgs {
namespace {

constexpr int CS_THREADS = 1024, CS_ITEMS = MAX_COLS / CS_THREADS;      // col_scan: CS_ITEMS consecutive columns per thread
constexpr int GS_THREADS = 512, GS_WAVES = GS_THREADS / WAVE, GS_ITEMS = GS_NMAX / GS_THREADS;
// bin_count / bin_scatter: Gaussians per workgroup.  The Gaussians of a workgroup are NOT neighbours on the screen (scene Gaussians come in
// no spatial order), so a workgroup touches min(columns, its pairs) columns and pays one global atomic for each: the atomics of a frame are
// workgroups x columns, and they serialise per cache line (~1.3 ns each: 860 k on 18 lines were 60 us at C3 with 512 Gaussians per workgroup).
#ifndef ADGS_BP_THREADS
#define ADGS_BP_THREADS 512
#endif
#ifndef ADGS_BP_ITEMS
#define ADGS_BP_ITEMS 2
#endif
constexpr int BP_THREADS = ADGS_BP_THREADS, BP_ITEMS = ADGS_BP_ITEMS, BP_GAUSS = BP_THREADS * BP_ITEMS;

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

__global__ void __launch_bounds__(256) bin_prepare_kernel(FramePrologue p) { run_frame_prologue(p); }

// Slab of a depth key among 32, from the cell's row of 31 bounds (word i = bound i; words 7, 15, 23 split the row into quarters): the
// quarter = how many of the three quarter bounds lie at or below the key, then how many of the quarter's seven bounds do -- two
// dependent steps of independent loads instead of a five-step binary search.  For sorted bounds this is "the number of bounds <= key";
// whatever the row holds, the result is a monotone function of the key (counts of thresholds are monotone, and a higher quarter ends
// above every slab of a lower one).
__device__ __forceinline__ uint32_t slab32_of(const uint32_t* row, uint32_t key) {
	const uint32_t q = (row[7] <= key ? 1u : 0u) + (row[15] <= key ? 1u : 0u) + (row[23] <= key ? 1u : 0u);
	const uint4 a = *reinterpret_cast<const uint4*>(row + 8 * q), c = *reinterpret_cast<const uint4*>(row + 8 * q + 4);
	const uint32_t r = (a.x <= key ? 1u : 0u) + (a.y <= key ? 1u : 0u) + (a.z <= key ? 1u : 0u) + (a.w <= key ? 1u : 0u) +
	                   (c.x <= key ? 1u : 0u) + (c.y <= key ? 1u : 0u) + (c.z <= key ? 1u : 0u);      // c.w is the quarter bound (or the row's pad word)
	return 8u * q + r;
}

// The (column, Gaussian) pairs of BP_GAUSS consecutive Gaussians, cells in row-major order of the Gaussian's cell rectangle.
// SCATTER = false (bin_count): the slab of every pair is looked up (the frame's snapshot of the bounds, staged in LDS when it fits), the
// pairs are counted per column in LDS and leave as one global atomic per touched column; the slabs of a Gaussian's first PACKED cells
// go to slab_words[idx] (5 bits each) so that the second pass does not search again.
// SCATTER = true (bin_scatter): counted again the same way, the workgroup's slice of every touched column reserved with ONE returning
// atomic on the column's cursor (col_scan left the column starts there), then one record per pair.
constexpr int PACKED = 6, TAB_LDS_CELLS = 128;
template <bool SCATTER>
__global__ void __launch_bounds__(BP_THREADS) bin_pairs_kernel(BinPairs b, uint32_t* __restrict__ col_word, uint32_t* __restrict__ slab_words, uint4* __restrict__ rec_u, uint32_t cap,
	uint32_t* __restrict__ pool_cursor) {
	__shared__ uint32_t s_cnt[MAX_COLS];
	__shared__ __attribute__((aligned(16))) uint32_t s_aux[SCATTER ? MAX_COLS : TAB_LDS_CELLS * SLAB_ROW];      // scatter: the reserved bases; count: the bounds table
	const int tid = threadIdx.x, ncol = b.ncells << b.lg, g0 = blockIdx.x * BP_GAUSS;
	col_word += (size_t)(blockIdx.x % BIN_COPIES) * b.cstride;      // this workgroup's copy of the column counters / cursors (both passes: the same copy)
	if (SCATTER && blockIdx.x == 0 && tid == 0) *pool_cursor = 0u;      // bookkeeping reset for the blend forward that follows on this stream
	uint4 d[BP_ITEMS]; uint32_t sw[BP_ITEMS];
#pragma unroll
	for (int k = 0; k < BP_ITEMS; k++) {      // unconditional, clamped: all loads in flight before the prologue's barrier
		const int idx = min(g0 + k * BP_THREADS + tid, b.P - 1);
		d[k] = b.dupinfo[idx];
		sw[k] = (SCATTER && b.lg > 0) ? slab_words[idx] : 0u;
	}
	for (int c = tid; c < ncol; c += BP_THREADS) s_cnt[c] = 0u;
	const bool tab_lds = !SCATTER && b.lg > 0 && b.ncells <= TAB_LDS_CELLS;
	if (tab_lds) for (int i = tid; i < b.ncells * (SLAB_ROW / 4); i += BP_THREADS) reinterpret_cast<uint4*>(s_aux)[i] = reinterpret_cast<const uint4*>(b.bounds)[i];
	__syncthreads();
	const int shift = MAX_SLAB_LG - b.lg;
	// pass 1: count (and, in bin_count, look the slabs up)
#pragma unroll
	for (int k = 0; k < BP_ITEMS; k++) {
		const int idx = g0 + k * BP_THREADS + tid;
		const uint32_t minx = d[k].x & 0xFFFFu, miny = d[k].x >> 16, maxx = d[k].y & 0xFFFFu, maxy = d[k].y >> 16;
		if (idx >= b.P || maxx <= minx || maxy <= miny) continue;
		const uint32_t c0x = minx / b.cell_tiles, c1x = (maxx - 1) / b.cell_tiles, c0y = miny / b.cell_tiles, c1y = (maxy - 1) / b.cell_tiles;
		uint32_t i = 0, packed = 0;
		for (uint32_t y = c0y; y <= c1y; y++)
			for (uint32_t x = c0x; x <= c1x; x++, i++) {
				const uint32_t cell = y * b.cgx + x;
				uint32_t s32 = 0;
				if (b.lg > 0) {
					if (SCATTER && i < PACKED) s32 = (sw[k] >> (5 * i)) & 31u;
#ifdef ADGS_DBG_NOSLAB
					else if (true) s32 = (d[k].z >> 18) & 31u;      // timing experiment: a slab function without a table (monotone in the depth bits: still a valid frame)
#endif
					else if (tab_lds) s32 = slab32_of(s_aux + cell * SLAB_ROW, d[k].z);      // (written out: through a pointer that may be either the loads become flat loads)
					else s32 = slab32_of(b.bounds + (size_t)cell * SLAB_ROW, d[k].z);
					if (!SCATTER && i < PACKED) packed |= s32 << (5 * i);
				}
				atomicAdd(s_cnt + (cell << b.lg) + (s32 >> shift), 1u);
			}
		if (!SCATTER && b.lg > 0) slab_words[idx] = packed;
	}
	__syncthreads();
	if (!SCATTER) {
		for (int c = tid; c < ncol; c += BP_THREADS) { const uint32_t n = s_cnt[c]; if (n) atomicAdd(col_word + c, n); }
		return;
	}
	for (int c = tid; c < ncol; c += BP_THREADS) { const uint32_t n = s_cnt[c]; s_aux[c] = n ? atomicAdd(col_word + c, n) : 0u; s_cnt[c] = 0u; }
	__syncthreads();
	// pass 2: one record per pair
#pragma unroll
	for (int k = 0; k < BP_ITEMS; k++) {
		const int idx = g0 + k * BP_THREADS + tid;
		const uint32_t minx = d[k].x & 0xFFFFu, miny = d[k].x >> 16, maxx = d[k].y & 0xFFFFu, maxy = d[k].y >> 16;
		if (idx >= b.P || maxx <= minx || maxy <= miny) continue;
		const uint32_t c0x = minx / b.cell_tiles, c1x = (maxx - 1) / b.cell_tiles, c0y = miny / b.cell_tiles, c1y = (maxy - 1) / b.cell_tiles;
		uint32_t i = 0;
		for (uint32_t y = c0y; y <= c1y; y++)
			for (uint32_t x = c0x; x <= c1x; x++, i++) {
				const uint32_t cell = y * b.cgx + x;
				uint32_t s32 = 0;
#ifdef ADGS_DBG_NOSLAB
				if (b.lg > 0) s32 = (d[k].z >> 18) & 31u;
#else
				if (b.lg > 0) s32 = i < PACKED ? (sw[k] >> (5 * i)) & 31u : slab32_of(b.bounds + (size_t)cell * SLAB_ROW, d[k].z);
#endif
				const uint32_t col = (cell << b.lg) + (s32 >> shift);
				const uint32_t pos = s_aux[col] + atomicAdd(s_cnt + col, 1u);
				// which tile rows / columns OF THIS CELL the Gaussian's rectangle covers: the blend forward runs its rectangle test on
				// these 4 bytes and gathers the Splat line only of candidates that pass it
				const uint32_t ty0 = y * b.cell_tiles, tx0 = x * b.cell_tiles;
				const uint32_t r0 = max(miny, ty0) - ty0, r1 = min(maxy, ty0 + b.cell_tiles) - ty0;      // [r0, r1) within the cell
				const uint32_t q0 = max(minx, tx0) - tx0, q1 = min(maxx, tx0 + b.cell_tiles) - tx0;
				const uint32_t rows = ((1u << r1) - 1u) & ~((1u << r0) - 1u), cols = ((1u << q1) - 1u) & ~((1u << q0) - 1u);
				if (pos < cap) rec_u[pos] = make_uint4(d[k].z, (uint32_t)idx, rows | (cols << b.cell_tiles), 0u);     // cap: speculative capacity
			}
	}
}

__global__ void __launch_bounds__(CS_THREADS) col_scan_kernel(ColScanArgs a) {
	__shared__ uint32_t s_wave[CS_THREADS / WAVE];
	__shared__ uint32_t s_start[MAX_COLS + 1];
	__shared__ uint32_t s_maxch;
	const int t = threadIdx.x, ncol = a.ncells << a.lg, c0 = t * CS_ITEMS;
	if (t == 0) s_maxch = 0u;
	uint32_t n[CS_ITEMS], nch[CS_ITEMS], sum = 0, chs = 0;
#pragma unroll
	for (int k = 0; k < CS_ITEMS; k++) n[k] = 0u;
	if (c0 < ncol) {
		// the column's pairs = the sum over the BIN_COPIES privatised copies; copy q's cursor starts behind the copies before it
		// (cursor = offset inside the column for now, the column start is added below)
#pragma unroll 4
		for (int q = 0; q < BIN_COPIES; q++) {
			const uint4 v = *reinterpret_cast<const uint4*>(a.col_count + (size_t)q * a.cstride + c0);      // columns beyond ncol: never counted into, zero or stale -> masked
			uint4 o = make_uint4(n[0], n[1], n[2], n[3]);
			*reinterpret_cast<uint4*>(a.col_cursor + (size_t)q * a.cstride + c0) = o;
			n[0] += v.x; n[1] += c0 + 1 < ncol ? v.y : 0u; n[2] += c0 + 2 < ncol ? v.z : 0u; n[3] += c0 + 3 < ncol ? v.w : 0u;
		}
	}
	static_assert(CS_ITEMS == 4, "col_scan reads a thread's columns as one 16-byte word");
#pragma unroll
	for (int k = 0; k < CS_ITEMS; k++) { nch[k] = (n[k] + GS_NMAX - 1) / GS_NMAX; sum += n[k]; chs += nch[k]; }
	uint32_t total, nchunks_total;
	const uint32_t start0 = block_excl_scan_1024(sum, s_wave, &total);
	uint32_t g = block_excl_scan_1024(chs, s_wave, &nchunks_total);
	{
		uint32_t run = start0;
#pragma unroll
		for (int k = 0; k < CS_ITEMS; k++) {
			if (c0 + k < ncol) s_start[c0 + k] = run;
			run += n[k];
		}
		if (t == 0) s_start[ncol] = total;
		if (c0 < ncol) {      // bin_scatter reserves from the cursors: column start + the pairs of the copies before this one
			const uint4 st = make_uint4(start0, start0 + n[0], start0 + n[0] + n[1], start0 + n[0] + n[1] + n[2]);
#pragma unroll 4
			for (int q = 0; q < BIN_COPIES; q++) {
				uint4* cur = reinterpret_cast<uint4*>(a.col_cursor + (size_t)q * a.cstride + c0);
				uint4 o = *cur;
				o.x += st.x; o.y += st.y; o.z += st.z; o.w += st.w;
				*cur = o;
			}
		}
	}
	{	// the fullest column (chunks): a column of k chunks pays k - 1 rank searches per entry in the merge, so ONE hot column (bounds that
		// do not fit this camera at all, a close-up object in a cell that has no bounds yet) is a cliff the average does not show; the host keeps the
		// next frames of such a scene on the device-wide sort for a while (api.hip)
		uint32_t m = 0;
#pragma unroll
		for (int k = 0; k < CS_ITEMS; k++) m = max(m, nch[k]);
#pragma unroll
		for (int off = WAVE / 2; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, WAVE));
		if ((t & (WAVE - 1)) == 0) atomicMax(&s_maxch, m);
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < CS_ITEMS; k++) {
		if (c0 + k >= ncol) continue;
		const uint32_t col = (uint32_t)(c0 + k), cell = col >> a.lg;
		const uint32_t st = s_start[col], cs = s_start[cell << a.lg], ce = s_start[(cell + 1u) << a.lg];
		uint4* out = reinterpret_cast<uint4*>(a.chunks);
		for (uint32_t q = 0; q < nch[k]; q++, g++)
			if (g < a.max_chunks) {
				out[2 * g] = make_uint4(st + q * GS_NMAX, min(st + (q + 1) * GS_NMAX, st + n[k]), st, st + n[k]);
				out[2 * g + 1] = make_uint4(cs, ce - cs, cell, 0u);
			}
	}
	for (int c = t; c < a.ncells; c += CS_THREADS) a.cell_ranges[c] = make_uint2(s_start[(uint32_t)c << a.lg], s_start[((uint32_t)c + 1u) << a.lg]);
	if (t == 0) {
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

// An entry that ends at rank r (0-based) of its cell's n sorted entries is the cell's j-th 32-quantile iff r == floor(j n / 32) for
// a j in 1 .. 31: its depth key is bound j - 1 of the cell's row for the NEXT frame of this thread (a cell of fewer than 32 entries
// writes fewer bounds; what stays from older frames keeps the slab function monotone, slab32_of).
__device__ __forceinline__ void publish_bound(uint32_t* __restrict__ bounds_out, uint32_t cell, uint32_t r, uint32_t n, uint32_t key) {
	const uint32_t j = (uint32_t)(((unsigned long long)r * 32ull + n - 1ull) / n);
	if (j >= 1u && j <= 31u && (uint32_t)(((unsigned long long)j * n) >> 5) == r) bounds_out[(size_t)cell * SLAB_ROW + j - 1u] = key;
}

// chunk_bsort: the chunk's entries in (depth, index) order by a histogram-equalised BUCKET sort, ~35 vector instructions per entry where
// an LSD radix pass alone costs ~50 (eight ballots and eight 64-bit per-lane selects per key and pass; rounds 3 - 5: 9 CU cycles per entry,
// 42 us at C3).  The order is total (indices are unique), so no pass has to be stable: every entry is mapped to a bucket by a MONOTONE
// function of its depth key, and its final position is the start of its bucket + the number of the bucket's entries that precede it in
// (depth, index) order, counted by walking the bucket (one or two entries on average).  The map: a linear map of the chunk's key range
// onto 256 coarse bins, a histogram of those, and inside every coarse bin a linear map onto as many fine buckets as the bin holds entries
// -- the chunk's own distribution equalised to about one entry per bucket however its depths cluster (a depth slab's first and last
// 32-quantile reach out to the nearest and the farthest Gaussian of the cell).  Monotone for any data: the coarse bin is a monotone
// function of the key, the position inside the bin too, and the buckets of a higher bin lie above those of a lower one.
// A chunk whose fullest bucket still holds more than BS_BMAX entries (thousands of EQUAL depths, a cluster 256 x narrower than its
// coarse bin) is flagged and left to chunk_sort (the radix sort): flags[g] = 1.
constexpr int BS_NB = GS_NMAX, BS_NC = 256, BS_BMAX = 32;
__global__ void __launch_bounds__(GS_THREADS) chunk_bsort_kernel(ChunkSortArgs a) {
	__shared__ __attribute__((aligned(16))) uint32_t s_cnt[BS_NB];      // entries per fine bucket -> bucket starts
	__shared__ uint32_t s_key[GS_NMAX];
	__shared__ uint32_t s_id[GS_NMAX];
	__shared__ uint32_t s_cc[BS_NC];                                    // entries per coarse bin -> first fine bucket | fine buckets << 16
	__shared__ uint32_t s_red[3 * GS_WAVES];
	const uint32_t g = blockIdx.x;
	if (g >= a.d_counts[1] || a.d_counts[2] != 0u) return;
	const uint4 ch = reinterpret_cast<const uint4*>(a.chunks)[2 * g], cc = reinterpret_cast<const uint4*>(a.chunks)[2 * g + 1];
	const uint32_t start = ch.x, n = ch.y - ch.x;
	if (n == 0 || n > (uint32_t)GS_NMAX || ch.w > a.cap) return;      // beyond the speculative capacity: the host re-runs with exact sizes
	const bool single = ch.x == ch.z && ch.y == ch.w;                  // the whole column: the result is final
	const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
	uint32_t key[GS_ITEMS], id[GS_ITEMS], msk[GS_ITEMS], fb[GS_ITEMS], arr[GS_ITEMS];
	uint32_t kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
	for (int r = 0; r < GS_ITEMS; r++) {      // entry j = r * GS_THREADS + tid (any assignment will do: nothing here depends on an order)
		const uint32_t j = r * GS_THREADS + tid;
		uint4 rec = make_uint4(0u, 0u, 0u, 0u);
		if (j < n) { rec = a.rec_u[start + j]; kmin = min(kmin, rec.x); kmax = max(kmax, rec.x); }
		key[r] = rec.x; id[r] = rec.y; msk[r] = rec.z;
	}
	for (int i = tid; i < BS_NB / 4; i += GS_THREADS) reinterpret_cast<uint4*>(s_cnt)[i] = make_uint4(0u, 0u, 0u, 0u);
	if (tid < BS_NC) s_cc[tid] = 0u;
#pragma unroll
	for (int off = WAVE / 2; off > 0; off >>= 1) { kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, WAVE)); kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, WAVE)); }
	if (lane == 0) { s_red[wid] = kmin; s_red[GS_WAVES + wid] = kmax; }
	__syncthreads();
#pragma unroll
	for (int w = 0; w < GS_WAVES; w++) { kmin = min(kmin, s_red[w]); kmax = max(kmax, s_red[GS_WAVES + w]); }
	// coarse bin of a key: floor((key - kmin) * 256 / (range + 1)), in float (conversion, product and truncation are all monotone)
	const float cscale = (float)BS_NC / ((float)(kmax - kmin) + 1.0f);
	auto coarse = [&](uint32_t k, float& cf) -> uint32_t { cf = (float)(k - kmin) * cscale; return min((uint32_t)cf, (uint32_t)(BS_NC - 1)); };
#pragma unroll
	for (int r = 0; r < GS_ITEMS; r++) if ((uint32_t)(r * GS_THREADS + tid) < n) { float cf; atomicAdd(s_cc + coarse(key[r], cf), 1u); }
	__syncthreads();
	{	// exclusive prefix of the coarse counts: bin c owns the fine buckets [base, base + count) -- as many as it holds entries
		uint32_t cnt = 0, incl = 0;
		if (tid < BS_NC) { cnt = s_cc[tid]; incl = wave_incl_scan_u32(cnt, lane); if (lane == WAVE - 1) s_red[2 * GS_WAVES + wid] = incl; }
		__syncthreads();
		if (tid < BS_NC) {
			uint32_t off = 0;
#pragma unroll
			for (int w = 0; w < BS_NC / WAVE; w++) if (w < wid) off += s_red[2 * GS_WAVES + w];
			s_cc[tid] = (off + incl - cnt) | (cnt << 16);
		}
		__syncthreads();
	}
#pragma unroll
	for (int r = 0; r < GS_ITEMS; r++) {
		fb[r] = 0u; arr[r] = 0u;
		if ((uint32_t)(r * GS_THREADS + tid) < n) {
			float cf;
			const uint32_t c = coarse(key[r], cf), v = s_cc[c], w = v >> 16;
			const float frac = fminf(fmaxf(cf - (float)c, 0.f), 1.f);      // position inside the coarse bin (a clamped bin: 1)
			fb[r] = (v & 0xffffu) + min(w - 1u, (uint32_t)(frac * (float)w));
			arr[r] = atomicAdd(s_cnt + fb[r], 1u);                         // arrival number inside the bucket: any order
		}
	}
	__syncthreads();
	{	// exclusive prefix of the fine counts (8 consecutive buckets per thread) + the fullest bucket
		uint4 c0 = reinterpret_cast<const uint4*>(s_cnt)[2 * tid], c1 = reinterpret_cast<const uint4*>(s_cnt)[2 * tid + 1];
		const uint32_t sum = c0.x + c0.y + c0.z + c0.w + c1.x + c1.y + c1.z + c1.w;
		uint32_t mx = max(max(max(c0.x, c0.y), max(c0.z, c0.w)), max(max(c1.x, c1.y), max(c1.z, c1.w)));
		const uint32_t incl = wave_incl_scan_u32(sum, lane);
#pragma unroll
		for (int off = WAVE / 2; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, WAVE));
		if (lane == WAVE - 1) s_red[wid] = incl;
		if (lane == 0) s_red[GS_WAVES + wid] = mx;
		__syncthreads();
		uint32_t off = 0;
#pragma unroll
		for (int w = 0; w < GS_WAVES; w++) { if (w < wid) off += s_red[w]; mx = max(mx, s_red[GS_WAVES + w]); }
		if (mx > (uint32_t)BS_BMAX) { if (tid == 0) a.flags[g] = 1u; return; }      // block-uniform: this chunk takes the radix sort
		if (tid == 0) a.flags[g] = 0u;
		uint32_t run = off + incl - sum;
		uint4 o0, o1;
		o0.x = run; run += c0.x; o0.y = run; run += c0.y; o0.z = run; run += c0.z; o0.w = run; run += c0.w;
		o1.x = run; run += c1.x; o1.y = run; run += c1.y; o1.z = run; run += c1.z; o1.w = run;
		reinterpret_cast<uint4*>(s_cnt)[2 * tid] = o0; reinterpret_cast<uint4*>(s_cnt)[2 * tid + 1] = o1;
		__syncthreads();
	}
#pragma unroll
	for (int r = 0; r < GS_ITEMS; r++) if ((uint32_t)(r * GS_THREADS + tid) < n) { const uint32_t slot = s_cnt[fb[r]] + arr[r]; s_key[slot] = key[r]; s_id[slot] = id[r]; }
	__syncthreads();
#pragma unroll
	for (int r = 0; r < GS_ITEMS; r++) {
		if ((uint32_t)(r * GS_THREADS + tid) < n) {
			const uint32_t st = s_cnt[fb[r]], en = fb[r] + 1u < (uint32_t)BS_NB ? s_cnt[fb[r] + 1u] : n;
			uint32_t less = 0;
			for (uint32_t i = st; i < en; i++) { const uint32_t ok = s_key[i], oi = s_id[i]; less += (ok < key[r] || (ok == key[r] && oi < id[r])) ? 1u : 0u; }
			const uint32_t out = st + less;
			if (single) {
				a.ent_f[start + out] = make_uint2(id[r], msk[r]);
				if (a.bounds_out && cc.y) publish_bound(a.bounds_out, cc.z, start + out - cc.x, cc.y, key[r]);
			} else { a.key_s[start + out] = make_uint2(id[r], key[r]); a.mask_s[start + out] = msk[r]; }      // as one 64-bit word: depth << 32 | id
		}
	}
}

// chunk_sort: the LSD radix sort of a chunk bsort flagged (a.flags[g] != 0; nullptr: every chunk).
__global__ void __launch_bounds__(GS_THREADS) chunk_sort_kernel(ChunkSortArgs a) {
	__shared__ uint32_t s_a[GS_NMAX];
	__shared__ uint32_t s_b[GS_NMAX];
	__shared__ uint32_t s_wcnt[GS_WAVES][256];
	__shared__ uint32_t s_dbase[256];
	__shared__ uint32_t s_scan[256 / WAVE];
	__shared__ uint32_t s_diff[GS_WAVES];
	const uint32_t g = blockIdx.x;
	if (g >= a.d_counts[1] || a.d_counts[2] != 0u) return;
	if (a.flags && a.flags[g] == 0u) return;                          // chunk_bsort sorted this chunk
	const uint4 ch = reinterpret_cast<const uint4*>(a.chunks)[2 * g], cc = reinterpret_cast<const uint4*>(a.chunks)[2 * g + 1];
	const uint32_t start = ch.x, n = ch.y - ch.x;
	if (n == 0 || n > (uint32_t)GS_NMAX || ch.w > a.cap) return;      // beyond the speculative capacity: the host re-runs with exact sizes
	const bool single = ch.x == ch.z && ch.y == ch.w;                  // the whole column: the result is final
	const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
	const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (WAVE - lane));
	// wave w owns the contiguous position range [w * span, (w + 1) * span) of the chunk, 64 entries per round: the stable order of
	// the LSD passes is (wave, round, lane) = the position.  Positions >= n hold nothing (no padding keys: a lane without an entry takes
	// part in no ballot and no exchange, so a pass can be SKIPPED when every entry of the chunk has the same digit).
	const uint32_t rounds = (n + GS_THREADS - 1) / GS_THREADS, span = rounds * WAVE;
	uint32_t key[GS_ITEMS], id[GS_ITEMS], msk[GS_ITEMS], pos[GS_ITEMS];
	uint32_t diff = 0;
	{
		const uint32_t k0 = a.rec_u[start].x;
#pragma unroll
		for (int r = 0; r < GS_ITEMS; r++) {
			const uint32_t j = wid * span + r * WAVE + lane;
			uint4 rec = make_uint4(k0, 0u, 0u, 0u);
			if ((uint32_t)r < rounds && j < n) rec = a.rec_u[start + j];
			key[r] = rec.x; id[r] = rec.y; msk[r] = rec.z;
			diff |= rec.x ^ k0;
		}
#pragma unroll
		for (int off = WAVE / 2; off > 0; off >>= 1) diff |= (uint32_t)__shfl_xor((int)diff, off, WAVE);
		if (lane == 0) s_diff[wid] = diff;
		__syncthreads();
		diff = 0;
#pragma unroll
		for (int w = 0; w < GS_WAVES; w++) diff |= s_diff[w];      // bits in which the chunk's depth keys differ at all
	}
	for (int shift = 0; shift < 32; shift += 8) {
		if (((diff >> shift) & 255u) == 0u) continue;          // block-uniform: one digit for the whole chunk (the upper bits of a depth slab)
#pragma unroll
		for (int k = 0; k < 256 / WAVE; k++) s_wcnt[wid][k * WAVE + lane] = 0u;      // own wave's counters: LDS operations of a wave execute in order
#pragma unroll
		for (int r = 0; r < GS_ITEMS; r++) {
			if ((uint32_t)r < rounds) {              // block-uniform
				const bool have = wid * span + r * WAVE + lane < n;
				const uint32_t d = (key[r] >> shift) & 255u;
				uint64_t peers = __ballot(have);
#pragma unroll
				for (int bq = 0; bq < 8; bq++) {
					const bool bit = (d >> bq) & 1u;
					const uint64_t m = __ballot(bit);
					peers &= bit ? m : ~m;
				}
				if (have) {
					const uint32_t rank = __popcll(peers & lt_mask);
					uint32_t old = 0;
					if (rank == 0) { old = s_wcnt[wid][d]; s_wcnt[wid][d] = old + (uint32_t)__popcll(peers); }      // one leader per digit
					old = __shfl(old, __ffsll((unsigned long long)peers) - 1, WAVE);
					pos[r] = old + rank;                 // position inside this wave's run of digit d
				}
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
			if ((uint32_t)r < rounds && wid * span + r * WAVE + lane < n) {
				const uint32_t d = (key[r] >> shift) & 255u;
				pos[r] += s_dbase[d] + s_wcnt[wid][d];
				s_a[pos[r]] = key[r]; s_b[pos[r]] = id[r];
			}
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < GS_ITEMS; r++) {
			const uint32_t j = wid * span + r * WAVE + lane;
			if ((uint32_t)r < rounds && j < n) { key[r] = s_a[j]; id[r] = s_b[j]; }
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < GS_ITEMS; r++) if ((uint32_t)r < rounds && wid * span + r * WAVE + lane < n) s_a[pos[r]] = msk[r];
		__syncthreads();
#pragma unroll
		for (int r = 0; r < GS_ITEMS; r++) { const uint32_t j = wid * span + r * WAVE + lane; if ((uint32_t)r < rounds && j < n) msk[r] = s_a[j]; }
		__syncthreads();
	}
	// ---- entries of equal depth: order by Gaussian index (the reference's stable sort of keys emitted in index order)
#pragma unroll
	for (int r = 0; r < GS_ITEMS; r++) {
		const uint32_t j = wid * span + r * WAVE + lane;
		if ((uint32_t)r < rounds && j < n) { s_a[j] = key[r]; s_b[j] = id[r]; }
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
				if (single) {
					a.ent_f[start + out] = make_uint2(id[r], msk[r]);
					if (a.bounds_out && cc.y) publish_bound(a.bounds_out, cc.z, start + out - cc.x, cc.y, k);
				} else { a.key_s[start + out] = make_uint2(id[r], k); a.mask_s[start + out] = msk[r]; }      // as one 64-bit word: depth << 32 | id
			}
		}
	}
}

// One workgroup per part of a chunk A of a multi-chunk column.  A's (depth, id) keys stay in registers; every other chunk B of the column is
// staged in LDS and every key of A finds its rank in B by binary search there (12 steps of ~64 cycles instead of 12
// dependent L2 round trips of ~1 us each: the global-memory version of this kernel took 70 us at C3 and 480 us at C5 in round 2).
// The (depth, index) order is total (indices are unique), so the ranks in the other chunks + the position in A are the final slot.
constexpr int MG_THREADS = 512, MG_SPLIT = 2, MG_PART = GS_NMAX / MG_SPLIT, MG_ITEMS = MG_PART / MG_THREADS;
// (MG_SPLIT workgroups per chunk A, each with a part of A's keys; several workgroups fit a CU, so one stages its next B from global
// memory while the others search.)
__global__ void __launch_bounds__(MG_THREADS) chunk_merge_kernel(ChunkSortArgs a) {
	// binary-search probes sit at power-of-two strides: one pad slot per 32 keys spreads them over all banks
	__shared__ unsigned long long s_key[GS_NMAX + GS_NMAX / 32];
	const uint32_t g = blockIdx.x / MG_SPLIT, part = blockIdx.x % MG_SPLIT;
	if (g >= a.d_counts[1] || a.d_counts[2] != 0u) return;
	uint4 ch = reinterpret_cast<const uint4*>(a.chunks)[2 * g];
	if (ch.w > a.cap || (ch.x == ch.z && ch.y == ch.w)) return;        // one-chunk columns are final already
	const uint4 cc = reinterpret_cast<const uint4*>(a.chunks)[2 * g + 1];
	const int tid = threadIdx.x;
	const uint32_t chunk_x = ch.x;                                      // start of the whole chunk A (its slot in the column)
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
		{	// all loads of the chunk in flight at once, then the LDS stores (one load -> wait -> store per step serialises the L2 round
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
		// Branch-free lower bound, the MG_ITEMS searches of a thread interleaved (a first version with `if (lo < hi)` and a
		// short-circuit two-word comparison compiled to one branch and two dependent LDS round trips per probe: 144 us at C3).
		// Invariant: the number of keys of B below key[r] lies in [lo, lo + len]; every step halves len (nb >= 1).
		uint32_t lo[MG_ITEMS];
#pragma unroll
		for (int r = 0; r < MG_ITEMS; r++) lo[r] = 0u;
		uint32_t len = nb;
		while (len > 1) {                                               // block-uniform trip count: <= 12
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
		if (j < n) {
			a.ent_f[ch.z + rank[r]] = make_uint2((uint32_t)key[r], a.mask_s[ch.x + j]);
			if (a.bounds_out && cc.y) publish_bound(a.bounds_out, cc.z, ch.z + rank[r] - cc.x, cc.y, (uint32_t)(key[r] >> 32));
		}
	}
}

} // namespace

int launch_bin_prepare(const FramePrologue& p, hipStream_t stream) {
	const int work = max(p.n_zero, p.n_copy);
	if (work <= 0) return 0;
	hipLaunchKernelGGL(bin_prepare_kernel, dim3(work > 2048 ? 8 : 1), dim3(256), 0, stream, p);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
int launch_bin_count(const BinPairs& b, uint32_t* col_count, uint32_t* slab_words, hipStream_t stream) {
	if (b.P == 0) return 0;
	hipLaunchKernelGGL(bin_pairs_kernel<false>, dim3((b.P + BP_GAUSS - 1) / BP_GAUSS), dim3(BP_THREADS), 0, stream, b, col_count, slab_words, (uint4*)nullptr, 0u, (uint32_t*)nullptr);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
int launch_bin_scatter(const BinPairs& b, uint32_t* col_cursor, uint32_t* slab_words, uint4* rec_u, uint32_t cap, uint32_t* pool_cursor, hipStream_t stream) {
	if (b.P == 0) return 0;
	hipLaunchKernelGGL(bin_pairs_kernel<true>, dim3((b.P + BP_GAUSS - 1) / BP_GAUSS), dim3(BP_THREADS), 0, stream, b, col_cursor, slab_words, rec_u, cap, pool_cursor);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
int launch_col_scan(const ColScanArgs& a, hipStream_t stream) {
	hipLaunchKernelGGL(col_scan_kernel, dim3(1), dim3(CS_THREADS), 0, stream, a);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
int launch_chunk_sort(const ChunkSortArgs& a, uint32_t grid, hipStream_t stream) {
	if (grid == 0) return 0;
	if (a.flags) hipLaunchKernelGGL(chunk_bsort_kernel, dim3(grid), dim3(GS_THREADS), 0, stream, a);
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
