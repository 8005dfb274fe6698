// Bucket binning of the v2 pipeline for gfx950: depth-sorted per-cell Gaussian lists in five short launches, no device-wide sort, no merge.
//
// What it replaces (RAST/cuda_rasterizer/rasterizer_impl.cu:284-324 in the reference: InclusiveSum, duplicateWithKeys, a device-wide
// cub radix sort over 64-bit keys, identifyTileRanges; in this library's first v2 pipeline: a 2-launch scan, duplicate_cells, a
// 15-launch LSD radix sort of (cell | depth) keys and tile_ranges = 20 launches and ~175 us per C3 frame for 1.5 M pairs).
// The lists of different coarse cells are independent sorting problems; nothing has to be sorted device-wide:
//
//   preprocess_fwd   counts the (cell, Gaussian) pairs of its 256 Gaussians per coarse cell in LDS and writes the row
//                    counts[workgroup][cell] (no global atomics: scene Gaussians come in no spatial order, a workgroup touches every
//                    cell, and atomics on a few hot lines serialise -- measured in rounds 2 and 6);
//   cell_colscan     one workgroup per cell: exclusive prefix of the cell's column over the workgroups, in place + the cell total;
//   cell_scan        ONE workgroup: exclusive scan of the cell totals -> cell ranges; the frame's totals and capacity check;
//   cell_scatter     every pair goes to cell_start + counts[workgroup][cell] + (LDS atomic inside the workgroup): its depth key into one
//                    array, (Gaussian id, rectangle mask) into another -- any order inside a cell;
//   slab_sort        a cell's list is only ever read front to back, so it is built as the concatenation of 2^lg independently sorted
//                    DEPTH SLABS: workgroup (cell, slab) streams the cell's depth keys (4 bytes per entry, L2-resident), keeps the
//                    entries of its slab -- the slab of a depth is a search over the cell's row of 127 bounds; ANY row
//                    contents give a monotone function of the depth, so correctness never depends on the bounds, only the balance
//                    does --, counts the entries of the slabs below it (= where its output starts: no scan over slabs, no second
//                    scatter), sorts its <= 4096 entries on (32 depth bits, Gaussian index) inside the CU (a histogram-equalised
//                    bucket sort; long runs of equal depths: an LSD radix sort) -- exactly the order the reference's stable sort of keys
//                    emitted in index order produces -- and writes them to their final positions.  The bounds are the 128-quantiles of the
//                    cell's depth keys in the SAME CAMERA's previous render (else: sampled by cell_sample), read off the sorted output.  A slab of more
//                    than 4096 entries (a thread's first frame, a camera the bounds do not fit at all) is bisected at the median of a
//                    sample and re-streamed: slower, never wrong.
//
// Rounds 3 - 5 cut a cell into position chunks of 8192, sorted each with an LSD radix sort (eight ballots and eight 64-bit selects per key
// and pass: 9 CU cycles per entry) and merged the chunks by rank search (k - 1 searches per entry, 3.1 x the algorithmic traffic): 80 us of
// a 123 us chain at C3, and C5's 68 k pairs per cell went to the device-wide sort.  A first round-6 design that made (cell, slab) COLUMNS of
// the count / scatter level was built and measured slower: two more passes over the Gaussians at ~25 us each and 16-byte record stores
// that no longer coalesce when a workgroup's pairs spread over 560 - 2240 columns (EXPERIMENTS.md).
//
// The output is what render_fwd_v2 walks: per cell a contiguous [start, end) range of (id, rectangle mask) pairs, front to back.
#include "common.h"
#include "kernels.h"

namespace adgs {
namespace {

constexpr int CS_THREADS = MAX_CELLS;                                   // cell_scan: one cell per thread
constexpr int GS_THREADS = 512, GS_WAVES = GS_THREADS / WAVE, GS_ITEMS = GS_NMAX / GS_THREADS;

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
	uint32_t total, nchunks_total, work_total;
	const uint32_t start = block_excl_scan_1024(n, s_wave, &total);
	const uint32_t nch = (n + GS_NMAX - 1) / GS_NMAX;
	(void)block_excl_scan_1024(nch, s_wave, &nchunks_total);
	// depth slabs of the cell: the smallest power of two that brings its pairs per slab to SLAB_TARGET or below -- one slab_sort workgroup each
	uint32_t lg = 0;
	while (lg < (uint32_t)MAX_SLAB_LG && (n >> lg) > a.slab_target) lg++;
	if (a.force_lg >= 0) lg = min((uint32_t)a.force_lg, (uint32_t)MAX_SLAB_LG);
	const uint32_t m = (c < a.ncells && n) ? 1u << lg : 0u;
	const uint32_t w0 = block_excl_scan_1024(m, s_wave, &work_total);
	if (c < a.ncells) { a.cell_start[c] = start; a.cell_ranges[c] = make_uint2(start, start + n); a.cell_work[c] = make_uint2(w0, lg); }
	if (c == 0) {
		a.cell_start[a.ncells] = total; a.cell_work[a.ncells] = make_uint2(work_total, 0u);
		// a frame of more than max_chunks x 4096 pairs (67 M in production; ADGS_MAX_CHUNKS: the test hook) takes the device-wide sort
		const uint32_t over = nchunks_total > a.max_chunks ? 1u : 0u;
		unsigned long long fine = 0ull;
		for (int k = 0; k < SCAN_AUX_SLOTS; k++) fine += a.fine_total[k];
		// the binning and blend launches behind this one were enqueued against a capacity: do the totals fit?
		const uint32_t nofit = (over || total > a.cap_cells || fine > a.cap_fine) ? 1u : 0u;
		a.d_counts[0] = total; a.d_counts[1] = work_total; a.d_counts[2] = over; a.d_counts[3] = nofit;
		a.d_counts[4] = 0u; a.d_counts[5] = 0u;      // slab_sort: workgroups done, the fullest slab (in units of GS_NMAX entries)
		a.d_counts[6] = (uint32_t)fine; a.d_counts[7] = (uint32_t)(fine >> 32);
		// the totals go to the host mailbox HERE, before the scatter and the slab sorts run: the host (which has the rest of the forward enqueued
		// by now) learns whether the frame fitted its capacity while ~270 us of this frame's work are still queued, and spends them enqueueing
		// the backward.  (The fullest slab -- a statistic and a policy hint -- follows from slab_sort_slow, unsequenced.)
		if (a.box) {
			a.box->r_cells = total; a.box->r_fine = fine; a.box->oversize = over; a.box->n_groups = nchunks_total; a.box->overflow = nofit;
			if (nofit) a.box->overflow_count = a.box->overflow_count + 1u;
			a.box->cap_cells = a.cap_cells; a.box->cap_fine = a.cap_fine;
			__threadfence_system();
			a.box->seq = a.seq;                       // published last: the host spins on it
		}
	}
}

__global__ void __launch_bounds__(256) cell_scatter_kernel(int P, const uint4* __restrict__ dupinfo, const uint32_t* __restrict__ cell_start,
	const uint32_t* __restrict__ counts, uint32_t* __restrict__ rec_key, uint2* __restrict__ rec_im, uint32_t cap, int cell_tiles, int cgx, int ncells, uint32_t* __restrict__ pool_cursor, uint32_t* __restrict__ slow_list) {
	__shared__ uint32_t s_cnt[MAX_CELLS];
	__shared__ uint32_t s_base[MAX_CELLS];
	const int idx = blockIdx.x * blockDim.x + threadIdx.x;
	if (idx == 0) { *pool_cursor = 0u; slow_list[0] = 0u; }      // bookkeeping resets for the kernels that follow on this stream (the blend forward; slab_sort)
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
			if (pos < cap) { rec_key[pos] = d.z; rec_im[pos] = make_uint2((uint32_t)idx, rows | (cols << cell_tiles)); }     // cap: speculative capacity
		}
}

// Slab of a depth key among 128, from the cell's row of 127 bounds (word i = bound i; words 31, 63, 95 split the row into quarters): the
// quarter = how many of the three quarter bounds lie at or below the key, then how many of the quarter's 31 bounds do.  For sorted
// bounds this is "the number of bounds <= key"; whatever the row holds, the result is a monotone function of the key (counts of
// thresholds are monotone, and a higher quarter ends above every slab of a lower one).  (The generic path only: slab_sort itself
// compares against the two bounds of its slab.)
__device__ __forceinline__ uint32_t slab128_of(const uint32_t* row /* LDS, 16-byte aligned */, uint32_t key) {
	const uint32_t q = (row[31] <= key ? 1u : 0u) + (row[63] <= key ? 1u : 0u) + (row[95] <= key ? 1u : 0u);
	uint32_t r = 0;
#pragma unroll
	for (int i = 0; i < 8; i++) {
		const uint4 v = *reinterpret_cast<const uint4*>(row + 32 * q + 4 * i);
		r += (v.x <= key ? 1u : 0u) + (v.y <= key ? 1u : 0u) + (v.z <= key ? 1u : 0u) + ((i < 7 && v.w <= key) ? 1u : 0u);      // the quarter's last word is the quarter bound (or the row's pad)
	}
	return 32u * q + r;
}

// An entry that ends at rank r (0-based) of its cell's n sorted entries is the cell's j-th 128-quantile iff r == floor(j n / 128) for
// a j in 1 .. 127: its depth key is bound j - 1 of the cell's row for the NEXT frame (a cell of fewer than 128 entries
// writes fewer bounds; what stays from older frames keeps the slab function monotone, slab128_of).
__device__ __forceinline__ void publish_bound(uint32_t* __restrict__ bounds_out, uint32_t cell, uint32_t r, uint32_t n, uint32_t key) {
	const uint32_t j = (uint32_t)(((unsigned long long)r * (unsigned)SLAB_ROW + n - 1ull) / n);
	if (j >= 1u && j < (uint32_t)SLAB_ROW && (uint32_t)(((unsigned long long)j * n) / (unsigned)SLAB_ROW) == r) bounds_out[(size_t)cell * SLAB_ROW + j - 1u] = key;
}

// LDS of slab_sort: the selected entries' keys / list positions while they are collected, then the sort's arrays
struct SlabLds {
	uint32_t a[GS_NMAX];          // collect: depth keys of the selected entries;    bsort: fine-bucket counts -> starts;  radix: keys
	uint32_t b[GS_NMAX];          // collect: their positions in the cell's list;    bsort: keys by bucket;               radix: ids
	uint32_t c[GS_NMAX];          //                                                 bsort: ids by bucket;                radix: per-wave digit counters [GS_WAVES][256]
	uint32_t cc[256];             // bsort: coarse-bin counts -> first fine bucket | fine buckets << 16;  radix: digit bases
	uint32_t red[3 * GS_WAVES];
	uint32_t row[SLAB_ROW];       // the cell's bounds (this frame's snapshot)
	unsigned long long stack[72]; // bisection: pending upper halves (lo, hi) of the (key, id) range
	uint32_t nsel, below, last;   // entries selected / entries of the cell in front of the selection / "this workgroup finished last"
};
static_assert(GS_WAVES * 256 <= GS_NMAX, "the radix sort's per-wave digit counters live in SlabLds::c");

// The n <= GS_NMAX entries (key, id) a thread block holds in registers (entry j = r * GS_THREADS + tid), ranked in (key, id) order:
// rank[r] = final position among the n.  Histogram-equalised bucket sort, ~35 vector instructions per entry where an LSD radix pass alone
// costs ~50: the order is total (ids are unique), so nothing has to be stable: every entry is mapped to a bucket by a MONOTONE function of
// its key, and its rank is the start of its bucket + the number of the bucket's entries that precede it, counted by walking the bucket
// (one or two entries on average).  The map: a linear map of the key range onto 256 coarse bins, a histogram of those, and inside every
// coarse bin a linear map onto as many fine buckets as the bin holds entries -- the entries' own distribution equalised to about one per
// bucket however the depths cluster.  Monotone for any data: the coarse bin is a monotone function of the key, the position inside the
// bin too, and the buckets of a higher bin lie above those of a lower one.  Returns false (block-uniform, ranks undefined) when a
// bucket still holds more than BS_BMAX entries: thousands of EQUAL depths -- the caller takes the radix sort.
constexpr int BS_NB = GS_NMAX, BS_NC = 256, BS_BMAX = 32;
__device__ __forceinline__ bool bucket_rank(SlabLds& s, uint32_t n, const uint32_t (&key)[GS_ITEMS], const uint32_t (&id)[GS_ITEMS], uint32_t (&rank)[GS_ITEMS]) {
	const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
	uint32_t fb[GS_ITEMS], arr[GS_ITEMS];
	uint32_t kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
	for (int r = 0; r < GS_ITEMS; r++) if ((uint32_t)(r * GS_THREADS + tid) < n) { kmin = min(kmin, key[r]); kmax = max(kmax, key[r]); }
	__syncthreads();                      // the caller's use of the arrays is over
	for (int i = tid; i < BS_NB / 4; i += GS_THREADS) reinterpret_cast<uint4*>(s.a)[i] = make_uint4(0u, 0u, 0u, 0u);
	if (tid < BS_NC) s.cc[tid] = 0u;
#pragma unroll
	for (int off = WAVE / 2; off > 0; off >>= 1) { kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, WAVE)); kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, WAVE)); }
	if (lane == 0) { s.red[wid] = kmin; s.red[GS_WAVES + wid] = kmax; }
	__syncthreads();
#pragma unroll
	for (int w = 0; w < GS_WAVES; w++) { kmin = min(kmin, s.red[w]); kmax = max(kmax, s.red[GS_WAVES + w]); }
	// coarse bin of a key: floor((key - kmin) * 256 / (range + 1)), in float (conversion, product and truncation are all monotone)
	const float cscale = (float)BS_NC / ((float)(kmax - kmin) + 1.0f);
	auto coarse = [&](uint32_t k, float& cf) -> uint32_t { cf = (float)(k - kmin) * cscale; return min((uint32_t)cf, (uint32_t)(BS_NC - 1)); };
#pragma unroll
	for (int r = 0; r < GS_ITEMS; r++) if ((uint32_t)(r * GS_THREADS + tid) < n) { float cf; atomicAdd(s.cc + coarse(key[r], cf), 1u); }
	__syncthreads();
	{	// exclusive prefix of the coarse counts: bin c owns the fine buckets [base, base + count) -- as many as it holds entries
		uint32_t cnt = 0, incl = 0;
		if (tid < BS_NC) { cnt = s.cc[tid]; incl = wave_incl_scan_u32(cnt, lane); if (lane == WAVE - 1) s.red[2 * GS_WAVES + wid] = incl; }
		__syncthreads();
		if (tid < BS_NC) {
			uint32_t off = 0;
#pragma unroll
			for (int w = 0; w < BS_NC / WAVE; w++) if (w < wid) off += s.red[2 * GS_WAVES + w];
			s.cc[tid] = (off + incl - cnt) | (cnt << 16);
		}
		__syncthreads();
	}
#pragma unroll
	for (int r = 0; r < GS_ITEMS; r++) {
		fb[r] = 0u; arr[r] = 0u;
		if ((uint32_t)(r * GS_THREADS + tid) < n) {
			float cf;
			const uint32_t c = coarse(key[r], cf), v = s.cc[c], w = v >> 16;
			const float frac = fminf(fmaxf(cf - (float)c, 0.f), 1.f);      // position inside the coarse bin (a clamped bin: 1)
			fb[r] = (v & 0xffffu) + min(w - 1u, (uint32_t)(frac * (float)w));
			arr[r] = atomicAdd(s.a + fb[r], 1u);                           // arrival number inside the bucket: any order
		}
	}
	__syncthreads();
	{	// exclusive prefix of the fine counts (8 consecutive buckets per thread) + the fullest bucket
		const uint4 c0 = reinterpret_cast<const uint4*>(s.a)[2 * tid], c1 = reinterpret_cast<const uint4*>(s.a)[2 * tid + 1];
		const uint32_t sum = c0.x + c0.y + c0.z + c0.w + c1.x + c1.y + c1.z + c1.w;
		uint32_t mx = max(max(max(c0.x, c0.y), max(c0.z, c0.w)), max(max(c1.x, c1.y), max(c1.z, c1.w)));
		const uint32_t incl = wave_incl_scan_u32(sum, lane);
#pragma unroll
		for (int off = WAVE / 2; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, WAVE));
		if (lane == WAVE - 1) s.red[wid] = incl;
		if (lane == 0) s.red[GS_WAVES + wid] = mx;
		__syncthreads();
		uint32_t off = 0;
#pragma unroll
		for (int w = 0; w < GS_WAVES; w++) { if (w < wid) off += s.red[w]; mx = max(mx, s.red[GS_WAVES + w]); }
		if (mx > (uint32_t)BS_BMAX) return false;      // block-uniform
		uint32_t run = off + incl - sum;
		uint4 o0, o1;
		o0.x = run; run += c0.x; o0.y = run; run += c0.y; o0.z = run; run += c0.z; o0.w = run; run += c0.w;
		o1.x = run; run += c1.x; o1.y = run; run += c1.y; o1.z = run; run += c1.z; o1.w = run;
		reinterpret_cast<uint4*>(s.a)[2 * tid] = o0; reinterpret_cast<uint4*>(s.a)[2 * tid + 1] = o1;
		__syncthreads();
	}
#pragma unroll
	for (int r = 0; r < GS_ITEMS; r++) if ((uint32_t)(r * GS_THREADS + tid) < n) { const uint32_t slot = s.a[fb[r]] + arr[r]; s.b[slot] = key[r]; s.c[slot] = id[r]; }
	__syncthreads();
#pragma unroll
	for (int r = 0; r < GS_ITEMS; r++) {
		rank[r] = 0u;
		if ((uint32_t)(r * GS_THREADS + tid) < n) {
			const uint32_t st = s.a[fb[r]], en = fb[r] + 1u < (uint32_t)BS_NB ? s.a[fb[r] + 1u] : n;
			uint32_t less = 0;
			for (uint32_t i = st; i < en; i++) { const uint32_t ok = s.b[i], oi = s.c[i]; less += (ok < key[r] || (ok == key[r] && oi < id[r])) ? 1u : 0u; }
			rank[r] = st + less;
		}
	}
	return true;
}

// The same ranks by an LSD radix sort on the 32 key bits that never leaves the CU (keys in registers, ranks inside a wave by match-any
// ballots, exchange through LDS; passes whose digit is the same for all entries are skipped) + an index fix-up of equal keys.  Entry
// layout here: wave w owns the contiguous position range [w * span, (w + 1) * span), 64 entries per round (the stable order of the
// passes is the position): the caller's entries (index j = r * GS_THREADS + tid) are first moved into that layout through LDS.
// On return slot r of a thread holds (key[r], id[r]) of the entry the caller had at index src[r], and rank[r] is its final position
// (0xffffffff: an empty slot).
__device__ __forceinline__ void radix_rank(SlabLds& s, uint32_t n, uint32_t (&key)[GS_ITEMS], uint32_t (&id)[GS_ITEMS], uint32_t (&src)[GS_ITEMS], uint32_t (&rank)[GS_ITEMS]) {
	const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
	const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (WAVE - lane));
	const uint32_t rounds = (n + GS_THREADS - 1) / GS_THREADS, span = rounds * WAVE;
	uint32_t (*wcnt)[256] = reinterpret_cast<uint32_t (*)[256]>(s.c);
	__syncthreads();
#pragma unroll
	for (int r = 0; r < GS_ITEMS; r++) { const uint32_t j = r * GS_THREADS + tid; if (j < n) { s.a[j] = key[r]; s.b[j] = id[r]; } }
	__syncthreads();
	uint32_t pos[GS_ITEMS], diff = 0;
	{
		const uint32_t k0 = s.a[0];
#pragma unroll
		for (int r = 0; r < GS_ITEMS; r++) {
			const uint32_t j = wid * span + r * WAVE + lane;
			const bool have = (uint32_t)r < rounds && j < n;
			key[r] = have ? s.a[j] : k0; id[r] = have ? s.b[j] : 0u; src[r] = j; pos[r] = 0u;
			diff |= key[r] ^ k0;
		}
#pragma unroll
		for (int off = WAVE / 2; off > 0; off >>= 1) diff |= (uint32_t)__shfl_xor((int)diff, off, WAVE);
		__syncthreads();
		if (lane == 0) s.red[wid] = diff;
		__syncthreads();
		diff = 0;
#pragma unroll
		for (int w = 0; w < GS_WAVES; w++) diff |= s.red[w];      // bits in which the keys differ at all
	}
	for (int shift = 0; shift < 32; shift += 8) {
		if (((diff >> shift) & 255u) == 0u) continue;          // block-uniform: one digit for all entries
		__syncthreads();
#pragma unroll
		for (int k = 0; k < 256 / WAVE; k++) wcnt[wid][k * WAVE + lane] = 0u;      // own wave's counters: LDS operations of a wave execute in order
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
					const uint32_t rk = __popcll(peers & lt_mask);
					uint32_t old = 0;
					if (rk == 0) { old = wcnt[wid][d]; wcnt[wid][d] = old + (uint32_t)__popcll(peers); }      // one leader per digit
					old = __shfl(old, __ffsll((unsigned long long)peers) - 1, WAVE);
					pos[r] = old + rk;                   // position inside this wave's run of digit d
				}
			}
		}
		__syncthreads();
		if (tid < 256) {                             // digit tid: exclusive prefix over the waves + the digit's total
			uint32_t run = 0;
#pragma unroll
			for (int w = 0; w < GS_WAVES; w++) { const uint32_t c = wcnt[w][tid]; wcnt[w][tid] = run; run += c; }
			const uint32_t incl = wave_incl_scan_u32(run, lane);
			if (lane == WAVE - 1) s.red[GS_WAVES + wid] = incl;
			s.cc[tid] = incl - run;
		}
		__syncthreads();
		if (tid < 256) {
			uint32_t off = 0;
#pragma unroll
			for (int w = 0; w < 256 / WAVE; w++) if (w < wid) off += s.red[GS_WAVES + w];
			s.cc[tid] += off;
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < GS_ITEMS; r++) {
			if ((uint32_t)r < rounds && wid * span + r * WAVE + lane < n) {
				const uint32_t d = (key[r] >> shift) & 255u;
				pos[r] += s.cc[d] + wcnt[wid][d];
				s.a[pos[r]] = key[r]; s.b[pos[r]] = id[r];
			}
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < GS_ITEMS; r++) {
			const uint32_t j = wid * span + r * WAVE + lane;
			if ((uint32_t)r < rounds && j < n) { key[r] = s.a[j]; id[r] = s.b[j]; }
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < GS_ITEMS; r++) if ((uint32_t)r < rounds && wid * span + r * WAVE + lane < n) s.a[pos[r]] = src[r];
		__syncthreads();
#pragma unroll
		for (int r = 0; r < GS_ITEMS; r++) { const uint32_t j = wid * span + r * WAVE + lane; if ((uint32_t)r < rounds && j < n) src[r] = s.a[j]; }
	}
	// ---- entries of equal key: order by id (the reference's stable sort of keys emitted in index order)
	__syncthreads();
#pragma unroll
	for (int r = 0; r < GS_ITEMS; r++) {
		const uint32_t j = wid * span + r * WAVE + lane;
		if ((uint32_t)r < rounds && j < n) { s.a[j] = key[r]; s.b[j] = id[r]; }
	}
	__syncthreads();
#pragma unroll
	for (int r = 0; r < GS_ITEMS; r++) {
		const uint32_t j = wid * span + r * WAVE + lane;
		rank[r] = 0xffffffffu;
		if ((uint32_t)r < rounds && j < n) {
			const uint32_t k = key[r];
			uint32_t out = j;
			if ((j > 0 && s.a[j - 1] == k) || (j + 1 < n && s.a[j + 1] == k)) {
				uint32_t lo = j, hi = j + 1;
				while (lo > 0 && s.a[lo - 1] == k) lo--;
				while (hi < n && s.a[hi] == k) hi++;
				uint32_t less = 0;
				for (uint32_t i = lo; i < hi; i++) less += s.b[i] < id[r] ? 1u : 0u;
				out = lo + less;
			}
			rank[r] = out;
		}
	}
}

// This frame's own slab bounds from a SAMPLE of every cell's depth keys, for frames that have no bounds of their camera's previous render
// (a camera's first render, an evaluation view): the bounds of ANOTHER camera's render misfit narrow slabs by factors -- a cluster of object
// Gaussians that moves across a bound doubles a slab -- and an oversized slab costs slab_sort_slow three streams of its cell (measured: 350 us
// of a 750 us forward-only C3 frame).  One workgroup per cell: up to GS_NMAX keys at equal strides through the cell's unsorted records (their
// order is the Gaussians' index order interleaved over the scatter workgroups: a stride sees scene and object Gaussians in proportion), ranked
// like a slab, and the sample of rank floor(q S / 128) is bound q - 1.  With S = 4096 samples a slab of 1/16 of the cell is hit 256 +- 15 times:
// slab sizes within ~20 % (3 sigma) of their target.  Cells of one slab are skipped.
__global__ void __launch_bounds__(GS_THREADS, 4) cell_sample_kernel(SlabSortArgs a, uint32_t* __restrict__ bounds /* this frame's [ncells][SLAB_ROW] */) {
	__shared__ __attribute__((aligned(16))) SlabLds s;
	const int tid = threadIdx.x;
	const uint32_t cell = blockIdx.x;
	if (!(a.d_counts[2] == 0u && a.d_counts[3] == 0u && a.d_counts[0] <= a.cap)) return;
	const uint2 range = a.cell_ranges[cell];
	const uint32_t n_c = range.y - range.x;
	if (n_c == 0u || a.cell_work[cell].y == 0u) return;      // empty, or one slab: no bounds needed
	const uint32_t S = min(n_c, (uint32_t)GS_NMAX);
	uint32_t key[GS_ITEMS], id[GS_ITEMS], rank[GS_ITEMS], src[GS_ITEMS];
#pragma unroll
	for (int r = 0; r < GS_ITEMS; r++) {
		const uint32_t i = r * GS_THREADS + tid;
		id[r] = i; src[r] = 0u;
		key[r] = a.rec_key[range.x + min((uint32_t)(((unsigned long long)(2u * i + 1u) * n_c) / (2ull * S)), n_c - 1u)];      // clamped: all loads in flight
	}
	bool radix = false;
	if (!bucket_rank(s, S, key, id, rank)) { radix = true; radix_rank(s, S, key, id, src, rank); }
#pragma unroll
	for (int r = 0; r < GS_ITEMS; r++) {
		const bool have = radix ? rank[r] != 0xffffffffu : (uint32_t)(r * GS_THREADS + tid) < S;
		if (have) publish_bound(bounds, cell, rank[r], S, key[r]);
	}
	// (a cell of fewer than 128 samples leaves some bounds unwritten: stale words -> the row may be unsorted -> the generic kernel; it has one slab anyway)
}

// One workgroup per (cell, slab): see the file header.  The common case only: a sorted bounds row (the slab is then a
// key RANGE: two compares per streamed key), a selection of at most GS_NMAX entries, a bucket sort that succeeds; anything else is appended
// to a.slow_list for slab_sort_slow (which needs twice the registers: kept out of this kernel, three workgroups of which share a CU).
__global__ void __launch_bounds__(GS_THREADS, 6) slab_sort_kernel(SlabSortArgs a) {
	__shared__ __attribute__((aligned(16))) SlabLds s;
	const int tid = threadIdx.x, lane = tid & (WAVE - 1);
	if (!(a.d_counts[2] == 0u && a.d_counts[3] == 0u && a.d_counts[0] <= a.cap) || blockIdx.x >= a.d_counts[1]) return;      // beyond the capacity / the sort fallback: the host re-runs
	// which (cell, slab) is workgroup w?  The cell whose first workgroup is the last one <= w.  cell_work and the cell ranges are staged in
	// LDS in ONE round trip (<= 1025 cells x 16 bytes), so that nothing but the bounds row is a dependent load behind them
	for (int i = tid; i <= a.ncells; i += GS_THREADS) {
		const uint2 cwv = a.cell_work[i], rg = a.cell_ranges[min(i, a.ncells - 1)];
		s.a[i] = cwv.x; s.b[i] = cwv.y; s.c[2 * i] = rg.x; s.c[2 * i + 1] = rg.y;
	}
	__syncthreads();
	uint32_t cell = 0;
	{
		uint32_t len = (uint32_t)a.ncells;      // lower bound over starts[0 .. ncells): the number of cells whose start is <= w, minus one; empty cells share a start with their successor
		uint32_t lo_i = 0;
		while (len > 0u) { const uint32_t half = len >> 1; if (s.a[lo_i + half] <= blockIdx.x) { lo_i += half + 1u; len -= half + 1u; } else len = half; }
		cell = lo_i - 1u;                       // the LAST cell with start <= w: it is the non-empty one (an empty cell owns no workgroup)
	}
	const uint32_t lg = s.b[cell], slab = blockIdx.x - s.a[cell];
	const int shift = MAX_SLAB_LG - (int)lg;
	const uint2 range = make_uint2(s.c[2 * cell], s.c[2 * cell + 1]);
	__syncthreads();
	const uint32_t n_c = range.y - range.x;
	if (n_c == 0u || slab >= (1u << lg)) return;
	// the slab's key range [lo, hi) from the cell's row; an unsorted row (stale words of another image shape beside fresh ones): every slab of the cell goes the generic way
	uint32_t lo = 0u, hi = 0u; bool open_top = true, sorted = true;
	if (lg > 0u) {
		const uint32_t* row = a.bounds + (size_t)cell * SLAB_ROW;
		const uint32_t w0 = row[min(lane, 126)], w1 = row[min(lane + 1, 126)], w2 = row[min(lane + 64, 126)], w3 = row[min(lane + 65, 126)];
		sorted = __ballot((lane < 126 && w0 > w1) || (lane + 64 < 126 && w2 > w3)) == 0ull;
		if (slab > 0u) lo = row[(slab << shift) - 1u];
		open_top = slab == (1u << lg) - 1u;
		if (!open_top) hi = row[((slab + 1u) << shift) - 1u];
	}
	if (tid == 0) { s.nsel = 0u; s.below = 0u; }
	__syncthreads();
	uint32_t my_below = 0;
	if (sorted) {
		// The cell's depth keys as 16-byte words from the aligned address below its first key: thread t of a batch owns words t, t + 512, ...,
		// eight of them in flight (32 keys per thread, 16 K keys per batch: most cells are one round trip).  Key j of the cell sits at
		// element off + j of the aligned stream.
		const uint32_t off = range.x & 3u, n_el = off + n_c;
		const uint4* __restrict__ kw = reinterpret_cast<const uint4*>(a.rec_key + (range.x - off));
		const uint32_t n_w = (n_el + 3u) >> 2;
		const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (WAVE - lane));
		constexpr int SW = 8;
		for (uint32_t wbase = 0; wbase < n_w; wbase += GS_THREADS * SW) {
			uint4 kq[SW];
#pragma unroll
			for (int u = 0; u < SW; u++) kq[u] = kw[min(wbase + u * GS_THREADS + tid, n_w - 1u)];      // unconditional, clamped (the last word may reach 12 bytes past the cell: inside the buffer, rec_key is followed by the hand-over list)
			// the wave's selected entries of this batch: ONE LDS atomic per wave and batch (one per 64 keys serialised the eight waves on a single word)
			uint32_t selbits = 0, wave_n = 0;
#pragma unroll
			for (int u = 0; u < SW; u++) {
				const uint32_t kk[4] = { kq[u].x, kq[u].y, kq[u].z, kq[u].w };
#pragma unroll
				for (int e = 0; e < 4; e++) {
					const uint32_t el = (wbase + u * GS_THREADS + tid) * 4u + e;
					const bool have = el >= off && el < n_el, sel = have && kk[e] >= lo && (open_top || kk[e] < hi);
					my_below += (have && kk[e] < lo) ? 1u : 0u;
					selbits |= sel ? 1u << (4 * u + e) : 0u;
					wave_n += (uint32_t)__popcll(__ballot(sel));
				}
			}
			if (wave_n != 0u) {      // wave-uniform
				uint32_t at0 = 0;
				if (lane == 0) at0 = atomicAdd(&s.nsel, wave_n);
				at0 = __shfl(at0, 0, WAVE);
				uint32_t run = 0;
#pragma unroll
				for (int u = 0; u < SW; u++) {
					const uint32_t kk[4] = { kq[u].x, kq[u].y, kq[u].z, kq[u].w };
#pragma unroll
					for (int e = 0; e < 4; e++) {
						const bool sel = (selbits >> (4 * u + e)) & 1u;
						const uint64_t m = __ballot(sel);
						if (m == 0ull) continue;      // wave-uniform
						const uint32_t at = at0 + run + (uint32_t)__popcll(m & lt);
						if (sel && at < (uint32_t)GS_NMAX) { s.a[at] = kk[e]; s.b[at] = (wbase + u * GS_THREADS + tid) * 4u + e - off; }
						run += (uint32_t)__popcll(m);
					}
				}
			}
		}
#pragma unroll
		for (int off = WAVE / 2; off > 0; off >>= 1) my_below += __shfl_xor(my_below, off, WAVE);
		if (lane == 0 && my_below) atomicAdd(&s.below, my_below);
	}
	__syncthreads();
	const uint32_t nsel = s.nsel, below = s.below;
	bool slow = !sorted || nsel > (uint32_t)GS_NMAX;
	uint32_t key[GS_ITEMS], id[GS_ITEMS], msk[GS_ITEMS], rank[GS_ITEMS];
	if (!slow) {
#pragma unroll
		for (int r = 0; r < GS_ITEMS; r++) {
			const uint32_t j = r * GS_THREADS + tid;
			key[r] = 0u; id[r] = 0u; msk[r] = 0u;
			if (j < nsel) { key[r] = s.a[j]; const uint2 im = a.rec_im[range.x + s.b[j]]; id[r] = im.x; msk[r] = im.y; }
		}
		slow = !bucket_rank(s, nsel, key, id, rank);
	}
	if (slow) {
		if (tid == 0) { const uint32_t at = atomicAdd(a.slow_list, 1u); if (at < a.grid) a.slow_list[1u + at] = (cell << 8) | slab; }
		return;
	}
	if (tid == 0 && nsel) atomicMax(a.d_counts + 5, 1u);
#pragma unroll
	for (int r = 0; r < GS_ITEMS; r++) {
		if ((uint32_t)(r * GS_THREADS + tid) < nsel) {
			const uint32_t at = below + rank[r];
			a.ent_f[range.x + at] = make_uint2(id[r], msk[r]);
			if (a.bounds_out) publish_bound(a.bounds_out, cell, at, n_c, key[r]);
			if (a.bounds_out2) publish_bound(a.bounds_out2, cell, at, n_c, key[r]);
		}
	}
}

// slab_sort_slow: the slabs slab_sort handed over (a.slow_list: [0] = count, then (cell << 8 | slab) words) -- more than GS_NMAX
// entries (a thread's first frame, a camera the bounds do not fit), long runs of equal depths, a bounds row that is not sorted.  The
// generic form of everything: the slab of a key by slab128_of (any row), a selection that does not fit is bisected at the median of the
// first GS_NMAX entries it collected and both halves are streamed again, the bucket sort falls back to the radix sort.  A few workgroups
// walk the list; the last one to finish reports the frame's fullest slab to the host mailbox.
__global__ void __launch_bounds__(GS_THREADS, 4) slab_sort_slow_kernel(SlabSortArgs a) {
	__shared__ __attribute__((aligned(16))) SlabLds s;
	const int tid = threadIdx.x, lane = tid & (WAVE - 1);
	const uint32_t total = a.d_counts[0];
	const bool run = a.d_counts[2] == 0u && a.d_counts[3] == 0u && total <= a.cap;      // else: beyond the capacity / the sort fallback: the host re-runs
	uint32_t max_rounds = 0;
	const uint32_t n_slow = run ? min(a.slow_list[0], a.grid) : 0u;
	for (uint32_t item = blockIdx.x; item < n_slow; item += gridDim.x) {
		__syncthreads();
		const uint32_t word = a.slow_list[1 + item], cell = word >> 8, slab = word & 255u;
		const uint32_t lg = a.cell_work[cell].y;
		const int shift = MAX_SLAB_LG - (int)lg;
		const uint2 range = a.cell_ranges[cell];
		const uint32_t n_c = range.y - range.x;
		if (n_c == 0u) continue;
		{
		if (lg > 0u && tid < SLAB_ROW) s.row[tid] = a.bounds[(size_t)cell * SLAB_ROW + tid];
		const uint32_t* __restrict__ keys = a.rec_key + range.x;
		// The selection: the entries of this slab whose (key, id) lies in [lo, hi) -- the whole slab at first (`all`).  A slab of more than
		// GS_NMAX entries is bisected at the median of the first GS_NMAX it collected, and both halves are streamed again.
		// `below`: entries of the cell in front of the selection = entries of lower slabs + entries of this slab below lo.
		unsigned long long lo = 0ull, hi = ~0ull; bool all = true;
		int sp = 0;      // s.stack: pending upper halves
		while (true) {
			if (tid == 0) { s.nsel = 0u; s.below = 0u; }
			__syncthreads();
			uint32_t my_below = 0;
			for (uint32_t base = 0; base < n_c; base += GS_THREADS * 8) {
				uint32_t k[8], idv[8];
#pragma unroll
				for (int u = 0; u < 8; u++) k[u] = keys[min(base + u * GS_THREADS + tid, n_c - 1u)];      // unconditional, clamped: eight loads in flight
#pragma unroll
				for (int u = 0; u < 8; u++) idv[u] = all ? 0u : a.rec_im[range.x + min(base + u * GS_THREADS + tid, n_c - 1u)].x;
#pragma unroll
				for (int u = 0; u < 8; u++) {
					const uint32_t j = base + u * GS_THREADS + tid;
					const bool have = j < n_c;
					const uint32_t sl = lg > 0u ? slab128_of(s.row, k[u]) >> shift : 0u;
					bool sel = have && sl == slab, blw = have && sl < slab;
					if (!all && sel) {
						const unsigned long long kk = ((unsigned long long)k[u] << 32) | idv[u];
						blw = kk < lo; sel = kk >= lo && kk < hi;
					}
					my_below += blw ? 1u : 0u;
					const uint64_t m = __ballot(sel);
					if (m != 0ull) {      // wave-uniform
						uint32_t at = 0;
						if (lane == 0) at = atomicAdd(&s.nsel, (uint32_t)__popcll(m));
						at = __shfl(at, 0, WAVE) + (uint32_t)__popcll(m & ((lane == 0) ? 0ull : (~0ull >> (WAVE - lane))));
						if (sel && at < (uint32_t)GS_NMAX) { s.a[at] = k[u]; s.b[at] = j; }
					}
				}
			}
#pragma unroll
			for (int off = WAVE / 2; off > 0; off >>= 1) my_below += __shfl_xor(my_below, off, WAVE);
			if (lane == 0 && my_below) atomicAdd(&s.below, my_below);
			__syncthreads();
			const uint32_t nsel = s.nsel, below = s.below;
			max_rounds = max(max_rounds, (nsel + GS_NMAX - 1) / GS_NMAX);
			// the (up to GS_NMAX) collected entries into registers, with their ids and masks
			const uint32_t n = min(nsel, (uint32_t)GS_NMAX);
			uint32_t key[GS_ITEMS], id[GS_ITEMS], msk[GS_ITEMS], rank[GS_ITEMS], src[GS_ITEMS];
#pragma unroll
			for (int r = 0; r < GS_ITEMS; r++) {
				const uint32_t j = r * GS_THREADS + tid;
				key[r] = 0u; id[r] = 0u; msk[r] = 0u; src[r] = 0u;
				if (j < n) { key[r] = s.a[j]; const uint2 im = a.rec_im[range.x + s.b[j]]; id[r] = im.x; msk[r] = im.y; }
			}
			bool radix = false;
			if (!bucket_rank(s, n, key, id, rank)) { radix = true; radix_rank(s, n, key, id, src, rank); }
			if (nsel <= (uint32_t)GS_NMAX) {
				if (radix) {
					// the radix sort moved keys and ids between threads: the masks follow through LDS (slot r now holds the entry the thread block had at index src[r])
					__syncthreads();
#pragma unroll
					for (int r = 0; r < GS_ITEMS; r++) { const uint32_t j = r * GS_THREADS + tid; if (j < n) s.c[j] = msk[r]; }
					__syncthreads();
#pragma unroll
					for (int r = 0; r < GS_ITEMS; r++) if (rank[r] != 0xffffffffu) msk[r] = s.c[src[r]];
				}
#pragma unroll
				for (int r = 0; r < GS_ITEMS; r++) {
					const bool have = radix ? rank[r] != 0xffffffffu : (uint32_t)(r * GS_THREADS + tid) < n;
					if (have) {
						const uint32_t at = below + rank[r];
						a.ent_f[range.x + at] = make_uint2(id[r], msk[r]);
						if (a.bounds_out) publish_bound(a.bounds_out, cell, at, n_c, key[r]);
						if (a.bounds_out2) publish_bound(a.bounds_out2, cell, at, n_c, key[r]);
					}
				}
				if (sp == 0) break;
				__syncthreads();
				hi = s.stack[sp - 1]; lo = s.stack[sp - 2]; sp -= 2; all = false;      // block-uniform: every thread reads the same LDS words
				__syncthreads();
				continue;
			}
			// more than fits: the median (key, id) of the collected sample splits the range; the lower half next, the upper half later
			// (lo < pivot < hi, both halves lose at least half the sample: the bisection ends)
			__syncthreads();
#pragma unroll
			for (int r = 0; r < GS_ITEMS; r++) {
				const bool have = radix ? rank[r] != 0xffffffffu : (uint32_t)(r * GS_THREADS + tid) < n;
				if (have && rank[r] == (uint32_t)GS_NMAX / 2) s.stack[sp] = ((unsigned long long)key[r] << 32) | id[r];
			}
			__syncthreads();
			const unsigned long long pivot = s.stack[sp];
			__syncthreads();
			if (tid == 0) s.stack[sp + 1] = hi;      // pending: [pivot, hi)
			sp += 2;
			hi = pivot; all = false;
			if (sp > 68) break;                     // (cannot happen: 34 nested bisections of a 64-bit range)
			__syncthreads();
		}
	}
	}
	// the last workgroup to finish leaves the fullest slab of the frame (in units of GS_NMAX entries) in the host mailbox: a statistic
	// (adgs_get_frame_status) and the hot-cell hint of the next frames' binning policy; the totals went out with cell_scan
	__syncthreads();
	if (tid == 0) {
		if (max_rounds) atomicMax(a.d_counts + 5, max_rounds);
		__threadfence();
		s.last = atomicAdd(a.d_counts + 4, 1u) == gridDim.x - 1u ? 1u : 0u;
	}
	__syncthreads();
	if (s.last && tid == 0 && a.box) {
		__threadfence();
		a.box->max_cell_chunks = *(volatile uint32_t*)(a.d_counts + 5);
		__threadfence_system();
	}
}

// The sorted (device-wide radix sort) path leaves the same learning material: thread j of workgroup `cell` reads the key at the cell's j-th
// 128-quantile position.  keys: (cell | depth) 64-bit sort keys, depth in the low 32 bits.
__global__ void __launch_bounds__(SLAB_ROW) bounds_from_sorted_kernel(const unsigned long long* __restrict__ keys, const uint2* __restrict__ cell_ranges, const uint32_t* __restrict__ d_total,
	uint32_t cap, uint32_t* __restrict__ bounds_out) {
	const uint32_t cell = blockIdx.x, j = threadIdx.x;
	const uint2 r = cell_ranges[cell];
	const uint32_t n = r.y - r.x;
	if (j < 1u || j >= (uint32_t)SLAB_ROW || n == 0u || r.y > cap || (d_total && r.y > *d_total)) return;
	bounds_out[(size_t)cell * SLAB_ROW + j - 1u] = (uint32_t)keys[r.x + (uint32_t)(((unsigned long long)j * n) / (unsigned)SLAB_ROW)];
}

} // namespace

int launch_bin_prepare(const FramePrologue& p, hipStream_t stream) {
	const int work = max(p.n_zero, p.n_copy);
	if (work <= 0) return 0;
	hipLaunchKernelGGL(bin_prepare_kernel, dim3(work > 2048 ? 8 : 1), dim3(256), 0, stream, p);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
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
int launch_cell_scatter(int P, const uint4* dupinfo, const uint32_t* cell_start, const uint32_t* counts, uint32_t* rec_key, uint2* rec_im, uint32_t cap,
	int cell_tiles, int cgx, int ncells, uint32_t* pool_cursor, uint32_t* slow_list, hipStream_t stream) {
	if (P == 0) return 0;
	hipLaunchKernelGGL(cell_scatter_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, dupinfo, cell_start, counts, rec_key, rec_im, cap, cell_tiles, cgx, ncells, pool_cursor, slow_list);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
int launch_slab_sort(const SlabSortArgs& a, hipStream_t stream) {
	if (a.ncells <= 0) return 0;
	hipLaunchKernelGGL(slab_sort_kernel, dim3(a.grid), dim3(GS_THREADS), 0, stream, a);
	hipLaunchKernelGGL(slab_sort_slow_kernel, dim3(128), dim3(GS_THREADS), 0, stream, a);      // the handed-over slabs + the mailbox (its last workgroup)
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
int launch_cell_sample(const SlabSortArgs& a, uint32_t* bounds, hipStream_t stream) {
	if (a.ncells <= 0) return 0;
	hipLaunchKernelGGL(cell_sample_kernel, dim3((unsigned)a.ncells), dim3(GS_THREADS), 0, stream, a, bounds);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
int launch_bounds_from_sorted(const uint64_t* keys, const uint2* cell_ranges, int ncells, const uint32_t* d_total, uint32_t cap, uint32_t* bounds_out, hipStream_t stream) {
	if (ncells <= 0 || !bounds_out) return 0;
	hipLaunchKernelGGL(bounds_from_sorted_kernel, dim3(ncells), dim3(SLAB_ROW), 0, stream, reinterpret_cast<const unsigned long long*>(keys), cell_ranges, d_total, cap, bounds_out);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

} // namespace adgs
