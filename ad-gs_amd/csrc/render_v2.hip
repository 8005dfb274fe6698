// v2 blend kernels for gfx950: coarse-binned, lazily filtered, one wave64 per 16x16 tile.
//
// Why (measured on MI355X, profiles/r01/v1_baseline_*): with the classic pipeline a 1M-Gaussian
// 1920x1280 frame has 47M (tile, Gaussian) pairs; sorting them costs 4 ms and the blend kernels
// only ever consume the front ~5% of every tile list because pixels saturate.  v2 therefore
//   * bins/sorts Gaussians only into COARSE cells (128x128 px: ~20x fewer pairs), by
//     (cell | depth) with the same stable radix sort, and
//   * lets every 16x16 tile walk its cell's depth-sorted list lazily: 64 entries at a time are
//     tested against the tile (the Gaussian's tile rectangle = the reference rectangle shrunk by
//     an opacity-aware bound, so only entries that can reach alpha >= 1/255 on the tile survive),
//     survivors are compacted IN ORDER with a wave ballot (no sort needed) and blended until
//     all 256 pixels are saturated.  Entries behind the saturation depth are never touched.
// The per-pixel result is identical to the reference's: it depends only on the ordered sequence
// of Gaussians that pass the per-pixel tests, and removed entries are exactly those that fail
// `alpha >= 1/255` on every pixel of the tile (or lie outside the reference's tile rectangle).
//
// One workgroup = one wave64 = one tile (or half of one for small images: PPL = 2); each lane owns PPL pixels
// (column x = lane&15, rows (lane>>4) + 4k).  Per-Gaussian data is broadcast from LDS once per PPL pixels, 16x4-pixel
// strips an entry does not reach are skipped wave-uniformly, the backward's cross-lane reductions (through LDS:
// wave_sum14_lds) are amortised over the whole tile and the 14 per-Gaussian partial sums go out as ONE atomic
// instruction onto one 64-byte line.
// The consumed (tile, Gaussian) sequence is written to a chunk pool (linked 64-entry chunks) so
// the backward replays exactly what the forward blended, back to front.
#include "common.h"
#include "kernels.h"
#include "blend_instrument.h"
#include <cstdlib>

namespace adgs {
namespace {

constexpr float ALPHA_MAX = 0.99f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_STOP = 0.0001f;
constexpr float PIXEL_DONE = 3.0e38f;      // row coordinate of a pixel that takes no further entries (forward): exp -> 0
// pixels per lane: 4 (one wave = one 16x16 tile) or, for small images that would leave the chip idle, 2 (one wave = a 16x8
// half tile: twice the waves, a shorter dependent chain per wave); PPL is a template parameter of both blend kernels
constexpr uint32_t NO_CHUNK = 0xFFFFFFFFu;
constexpr int SPLAT_ROW = 5;       // float4 quads between two gathered Splat rows in LDS (4 data quads + 1 pad quad: conflict-free row stores, render_fwd_v2_kernel)

// The exponent is kept in log2 units (entry_geom scales the conic by log2 e once per entry): G = 2^pw is ONE v_exp_f32 per pixel
// (~1 ulp), not v_mul + v_exp.  ADGS_PRECISE_EXP: libm's exp2f (parity experiments: the share of gate flips the fast exp owns).
#ifndef ADGS_PRECISE_EXP
#define ADGS_EXP2(x) __builtin_amdgcn_exp2f(x)
#else
#define ADGS_EXP2(x) exp2f(x)
#endif
constexpr float LOG2E = 1.4426950408889634f;
// (The A/B switches of rounds 3 - 5 -- ADGS_LEAN, ADGS_SETPRIO, ADGS_BWD_PF2, ADGS_FWD_DMA, ADGS_FWD_SCAN_ROUNDS, ADGS_FWD_KEY_RING, ADGS_FWD_WAVES /
// ADGS_BWD_WAVES, ADGS_SPLAT_ROW -- are retired: every one was measured, the winner is the code below, the losers are in EXPERIMENTS.md.)

// Can the Gaussian of Splat line (q0 = x y ca cb, cc, tau) reach alpha >= 1/255 on any pixel centre of the wave's tile (column tx, pixel rows
// row0 .. row0 + rows - 1)?  Exact minimum of the quadratic form d^T Q d over the tile's pixel-centre rectangle (a lower bound of the
// minimum over its integer pixels), compared with tau (which carries the slack).  The tile lies inside the Gaussian's tile rectangle:
// the list's row / column masks said so.
__device__ __forceinline__ bool tile_may_contribute(const float4 q0, float C, float tau, uint32_t tx, uint32_t row0, int rows) {
	const float A = q0.z, B = q0.w;
	// rectangle relative to the mean: d = pixel - mean (the form is symmetric in the sign of d)
	const float x0 = (float)(tx * TILE_X) - q0.x, x1 = x0 + (float)(TILE_X - 1);
	const float y0 = (float)row0 - q0.y, y1 = y0 + (float)(rows - 1);      // the wave's own rows (a 16x16 tile or a half of one)
	if (x0 <= 0.f && x1 >= 0.f && y0 <= 0.f && y1 >= 0.f) return true;      // mean inside the tile
	// minimum over the four edges: fix one coordinate, clamp the unconstrained minimiser of the other
	// (any point of an edge bounds its minimum from above and the form is flat at the minimiser, so the
	// 1-ulp reciprocals only perturb `best` in second order -- far inside the slack carried by tau)
	float best = 3.0e38f;
	const float nBrC = -B * __builtin_amdgcn_rcpf(C), nBrA = -B * __builtin_amdgcn_rcpf(A);
#pragma unroll
	for (int e = 0; e < 2; e++) {
		const float dx = e ? x1 : x0;
		const float dy = fminf(fmaxf(nBrC * dx, y0), y1);
		best = fminf(best, A * dx * dx + 2.f * B * dx * dy + C * dy * dy);
		const float ey = e ? y1 : y0;
		const float ex = fminf(fmaxf(nBrA * ey, x0), x1);
		best = fminf(best, A * ex * ex + 2.f * B * ex * ey + C * ey * ey);
	}
	return best <= tau * 1.0005f + 1e-3f;
}

// Measured on gfx950 and NOT used: packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 issue at half the rate of their scalar
// counterparts -- a SIMD already retires a wave64 v_fma_f32 in 2 cycles -- ; this file is built with -fno-slp-vectorize so the
// compiler does not pack either.  Re-measured in round 4 with clean code -- the channel accumulations as v_pk_fma_f32 on the Splat
// line's natural pairs, op_sel broadcast, no register moves, 12 instead of 16 / 20 instead of 27 vector instructions per strip --:
// forward 0.234 -> 0.248 ms, backward 0.339 -> 0.343 ms, EXPERIMENTS.md), and branch-free predicated per-pixel code (the
// wave-uniform per-strip branches below skip ~22 % of the strip work).

// Per-entry, per-lane part of the Gaussian evaluation (the lane's 4 pixels share the column x):
//   power * log2 e = log2 e (-0.5 (A dx^2 + C dy^2) - B dx dy) = a0 + dy (b0 + c0 dy)
// Forward and backward evaluate alpha through this one function, so both take identical
// per-pixel decisions (power > 0, alpha < 1/255) on identical bits.
// LEAN entries (Splat::lean, set by the preprocess: opacity <= 0.99 and a conic safely inside the positive-definite cone) need neither
// the `power > 0` test nor the 0.99 clamp -- both provably never fire for them (preprocess.hip) -- which is 2 of 9 vector
// instructions and 1 of 2 compares per pixel; the choice is wave-uniform per entry (a bit of a 64-bit scalar mask per batch).
struct EntryGeom { float y, a0, b0, c0, op; };
// wave-uniform "does any lane ...": the ballot compared on the scalar unit (hipcc turns __any() of a value that already lives in
// an SGPR mask into v_cndmask + v_cmp_ne + s_cbranch_vccz: two vector instructions, ~7 cycles of the SIMD, per test)
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ EntryGeom entry_geom(const float4 q0, const float4 q1, float dx) {
	EntryGeom g;
	g.y = q0.y; g.a0 = ((-0.5f * LOG2E) * q0.z * dx) * dx; g.b0 = (-LOG2E * q0.w) * dx; g.c0 = (-0.5f * LOG2E) * q1.x; g.op = q1.y;
	return g;
}
template <bool LEAN>
__device__ __forceinline__ void eval_pixel(const EntryGeom& g, float py, float& dy, float& pw, float& G, float& al) {
	dy = g.y - py;
	pw = fmaf(dy, fmaf(g.c0, dy, g.b0), g.a0);
	G = ADGS_EXP2(pw);
	al = LEAN ? g.op * G : fminf(ALPHA_MAX, g.op * G);
}
// the four (PPL) pixels of a lane against one entry: alpha and the lane masks of the pixels that pass the gates (forward.cu:345-356)
template <bool LEAN, int PPL>
__device__ __forceinline__ uint64_t eval_entry_fwd(const EntryGeom& eg, const float (&pyf)[PPL], float (&alpha)[PPL], uint64_t (&actm)[PPL]) {
	uint64_t any_m = 0ull;
#pragma unroll
	for (int k = 0; k < PPL; k++) {
		float dy, pw, G;
		// a finished (or out-of-image) pixel sits at row 3e38: its exponent is -inf, alpha 0, so it needs no flag here
		eval_pixel<LEAN>(eg, pyf[k], dy, pw, G, alpha[k]);
		// "which pixels does the entry reach" as 64-bit lane masks on the scalar unit: the ballot of one comparison IS the
		// comparison's result register, two are combined by s_and and tested by s_cmp (a ballot of `a && b` would be
		// rebuilt from a per-lane 0 / 1 value: two vector instructions per strip and entry)
		actm[k] = LEAN ? __builtin_amdgcn_ballot_w64(!(alpha[k] < ALPHA_MIN))
		               : (__builtin_amdgcn_ballot_w64(!(pw > 0.0f)) & __builtin_amdgcn_ballot_w64(!(alpha[k] < ALPHA_MIN)));
		any_m |= actm[k];
	}
	return any_m;
}

// PUBLISH: the training forward (the consumed (tile, Gaussian) sequence goes to the chunk pool, per-pixel contributor counts and per-tile
// bookkeeping are written for the backward).  false: the forward-only render (adgs_raster_render: evaluation under torch.no_grad(),
// /root/reference/render.py:52-55,156) -- same pixels bit for bit, nothing kept for a backward that will not come.
template <int PPL, bool PUBLISH>
__global__ void __launch_bounds__(WAVE) render_fwd_v2_kernel(RenderV2FwdArgs a) {
	constexpr int ROWS = 4 * PPL, SUB = TILE_Y / ROWS;       // rows per wave tile; wave tiles per 16x16 tile
	// Gathered Splat lines, one ROW per candidate.  Rows are SPLAT_ROW = 5 quads apart (80 bytes), not 4: lane l stores its line with four
	// ds_write_b128 at l * 64 bytes otherwise, and the sixteen lanes of a pass land on four bank quads (round 4: SQ_LDS_BANK_CONFLICT = 23 % of
	// the kernel's LDS-active cycles); at 80 bytes the sixteen start banks 20 l mod 64 are all different.  The fifth quad of the 64 rows is
	// exactly the 256-entry candidate ring (s_cand): the kernel's LDS footprint stays 7680 bytes = 21 workgroups per CU.
	__shared__ uint32_t s_pub[PUBLISH ? 2 * WAVE : 1];         // live ids waiting to leave as a full chunk
	// key-stream scan: SCAN_ROUNDS x 64 list entries per step; ring: < 64 waiting + one step's survivors, power of two.  With the
	// staged key stream (below) a step is half a staged block: 4096 + 512 + 1024 + 2048 bytes of LDS = 21 workgroups per CU.
	constexpr int SCAN_ROUNDS = 2, CAND_RING = 2 * WAVE * SCAN_ROUNDS;
	static_assert(CAND_RING == 4 * WAVE && SPLAT_ROW == 5, "the candidate ring lives in the pad quads of the 64 gathered rows");
	__shared__ float4 s_splat[WAVE * SPLAT_ROW];
	auto cand = [&](uint32_t i) -> uint32_t& { return reinterpret_cast<uint32_t*>(s_splat)[(i >> 2) * (4 * SPLAT_ROW) + 16 + (i & 3u)]; };
	// The key stream of a cell is SEQUENTIAL and the tile's position in it is known long before the entries are needed: the next
	// block of KEY_BLOCK (id, mask) entries is copied global -> LDS by the DMA path of the load unit (global_load_lds_dwordx4: no
	// registers, two instructions per block) as soon as the previous block has been scanned, i.e. it flies under the filter round, the
	// Splat gather and the whole blend loop of the batch.  Until round 3 the scan loaded 256 entries into registers when it needed
	// them: ~10 exposed round trips per tile at several microseconds each under load (tools/blend_phase_timing.py) -- a third of a
	// wave's life.  Bucket-binned frames only (cell_entries); the device-wide-sort fallback keeps the register path.
	constexpr int KEY_BLOCK = 2 * SCAN_ROUNDS * WAVE, KEY_RING = 1;      // a block = two scan steps; one DMA instruction copies 128 entries; ONE staged block per wave (2 / 3: slower, below)
	__shared__ __attribute__((aligned(16))) uint2 s_keys[KEY_RING * KEY_BLOCK];
	const int lane = threadIdx.x;
	// Dispatch order.  The backward knows every tile's length and starts the longest first (launch_tile_order).  The forward does not know
	// its lengths -- but it knows the lengths of the LAST render of the same camera (api.hip: OrderHints; a training run returns to every
	// camera once per epoch, the bench every 16 steps): order_mode 2 walks the tiles longest-first by that hint.  9600 one-wave workgroups
	// over 5120 slots ran 1.9 rounds with 28 % of the kernel below half occupancy (profiles/r04/wave_timeline_c3.json: wave-time / slots =
	// 165 us against 238 us): with the hint 0.231 -> 0.198 ms at C3 (round 5; round 2's forward, which had no such tail, gained nothing).
	// Without a hint: bottom-up (order_mode 1) -- in driving scenes (the reference's KITTI / Waymo data, and the road-plane objects of
	// the synthetic configs) the lower image rows carry the long lists: 0.409 -> 0.400 ms at the time; ADGS_FWD_ORDER=0: top-down.
	// Any order gives the same images bit for bit: a tile's result does not depend on when it runs.
	// A camera is recognised by the addresses of its matrices (api.hip); the hint carries the view matrix it was made under, and a hint of
	// ANOTHER pose (recycled addresses, a camera that moved) is ignored here -- measured: a neighbouring camera's order is worse than bottom-up.
	bool use_hint = a.order_mode == 2;
	if (use_hint) {
#pragma unroll
		for (int i = 0; i < 16; i++) use_hint = use_hint && (a.fwd_view[i] == a.fwd_sig[i]);      // 32 cached scalar loads
	}
	uint32_t tile = use_hint ? a.fwd_order[blockIdx.x] : (a.order_mode != 0 ? gridDim.x - 1u - blockIdx.x : blockIdx.x);
	const uint32_t tx = tile % a.gx, ty = tile / a.gx;
	if (*a.overflow_flag != 0u) {
		// The frame does not fit the capacity this launch was enqueued against (api.hip: the totals are compared on the device): the
		// lists are incomplete.  Blend nothing, leave an empty replay state and a DEFINED empty render (background colour, opacity 0,
		// no contributors: a graph replay that overflowed hands these to the caller, whose loss and optimizer step must not see stale
		// pool memory -- a NaN that has reached the Adam moments cannot be redone); the host enqueues binning and blend again with
		// exact sizes (or, under stream capture, reports the overflow through adgs_get_frame_status).
		if (PUBLISH && lane == 0) { a.tile_last_chunk[tile] = NO_CHUNK; a.tile_consumed[tile] = 0u; a.tile_scanned[tile] = 0u; a.tile_batches[tile] = 0u; }
		const uint32_t ex = tx * TILE_X + (lane & 15);
		const size_t eHW = (size_t)a.H * a.W;
#pragma unroll
		for (int k = 0; k < PPL; k++) {
			const uint32_t ey = ty * (4 * PPL) + (lane >> 4) + 4 * k;
			if (ex < (uint32_t)a.W && ey < (uint32_t)a.H) {
				const size_t pix_id = (size_t)a.W * ey + ex;
				a.final_T[pix_id] = 0.f; a.n_contrib[pix_id] = 0u; a.out_depth[pix_id] = 0.f;
				if (a.has_color) for (int c = 0; c < 3; c++) a.out_color[c * eHW + pix_id] = a.bg[c] + (a.bg_image ? a.bg_image[c * eHW + pix_id] : 0.f);
				if (a.has_flow) for (int c = 0; c < 3; c++) a.out_flow[c * eHW + pix_id] = 0.f;
				if (a.has_sem) a.out_semantic[pix_id] = 0.f;
			}
		}
		return;
	}
	const uint32_t ty16 = ty / SUB;                           // row of the 16x16 tile grid the binning works on
	const uint32_t cell = (ty16 / a.cell_tiles) * a.cgx + (tx / a.cell_tiles);
	const uint2 range = a.cell_ranges[cell];
	const uint32_t px = tx * TILE_X + (lane & 15);
	const uint32_t py0 = ty * ROWS + (lane >> 4);
	const float pxf = (float)px;
	bool inside[PPL];
	float pyf[PPL], T[PPL], C0[PPL], C1[PPL], C2[PPL], Dp[PPL], F0[PPL], F1[PPL], F2[PPL], S0[PPL];
	uint32_t last_contrib[PPL];
#pragma unroll
	for (int k = 0; k < PPL; k++) {
		const uint32_t py = py0 + 4 * k;
		inside[k] = px < (uint32_t)a.W && py < (uint32_t)a.H;
		pyf[k] = inside[k] ? (float)py : PIXEL_DONE;
		last_contrib[k] = 0;
		T[k] = 1.f;
		C0[k] = C1[k] = C2[k] = Dp[k] = F0[k] = F1[k] = F2[k] = S0[k] = 0.f;
	}
	uint32_t pos = range.x, consumed = 0, prev_chunk = NO_CHUNK;
	uint32_t blk_next = tile * POOL_BLOCK, blk_left = POOL_BLOCK;      // chunk slots: the tile's own block first (kernels.h)
	const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (WAVE - lane));
	// stage-1 state: candidates whose rectangle holds this tile (ring buffer), the prefetched next batch of the key stream
	uint32_t chead = 0, ccount = 0;
	const bool staged = a.cell_entries != nullptr;
	// The staged blocks form a ring: block j of the cell's list (positions range.x + j KEY_BLOCK ...) lives in slot j % KEY_RING and is
	// requested as soon as that slot is free, i.e. KEY_RING blocks before the scan reaches it.  With ONE slot (the default) a block is
	// requested when the previous one is exhausted: the scan of a batch needs ~2.4 blocks and most of them are waited for at full latency
	// (6 us; 41 % of a wave's life, tools/blend_phase_timing.py) -- but two slots (9.5 KiB of LDS: 16 instead of 21 waves per CU) make the
	// kernel 20 us SLOWER and three 60 us (0.229 -> 0.250 -> 0.290 ms at C3): the waits are covered by the other waves of the SIMD, and a
	// resident wave is worth more to this kernel than a shorter wave.  bpos / bslot / bhalf: the block being
	// scanned, its slot, its next unscanned half; issued: end of the requested part of the list; landed: end of the part KNOWN to have
	// arrived -- memory operations complete in order, so everything requested before a wait for younger loads (the Splat gather of a
	// batch) or before an explicit vmcnt(0) is there, and a block is only waited for when it is not (a blanket vmcnt(0) per block also
	// waits for the acknowledgement of the chunk stores the blend loop has just issued).
	uint32_t bpos = range.x, bslot = 0, bhalf = 0, issued = range.x, landed = range.x;
	auto stage_block = [&](uint32_t p, uint32_t slot) {          // wave-uniform p < range.y
		// lane l copies entries p + 2l, p + 2l + 1 (16 bytes); lanes past the end of the list re-read its last entry (the word behind a
		// list is inside the binning buffer: BinStateV2::carve_buckets), the scan ignores positions >= range.y
		// Written as inline assembly, not __builtin_amdgcn_global_load_lds: the compiler cannot tell which LDS array a DMA writes and
		// puts s_waitcnt vmcnt(0) in front of EVERY later LDS read (the queue read that precedes the Splat gather, the blend loop's
		// rows) -- the round trip would be exposed again, once per batch.  Unknown to the compiler, the two copies only ever make its
		// own vmcnt waits conservative (memory operations complete in order); the scan waits for them itself (vmcnt(0) below).
		const uint32_t last = range.y - 1u;
		const uint2* g0 = a.cell_entries + min(p + 2u * lane, last);
		const uint2* g1 = a.cell_entries + min(p + (uint32_t)(KEY_BLOCK / 2) + 2u * lane, last);
		const uint32_t l0 = (uint32_t)(uintptr_t)s_keys + slot * (uint32_t)(KEY_BLOCK * sizeof(uint2)), l1 = l0 + (uint32_t)(KEY_BLOCK / 2 * sizeof(uint2));
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // m0 is a reserved register: named as clobbered on purpose (the DMA's LDS base travels in it)
		if (KEY_BLOCK > 2 * WAVE)
			asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, off\n\ts_mov_b32 m0, %3\n\tglobal_load_lds_dwordx4 %1, off"
				:: "v"(g0), "v"(g1), "s"(l0), "s"(l1) : "memory", "m0");
		else
			asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g0), "s"(l0) : "memory", "m0");
#pragma clang diagnostic pop
		issued = p + KEY_BLOCK;
	};
	if (staged) {
#pragma unroll
		for (int j = 0; j < KEY_RING; j++) if (range.x + (uint32_t)(j * KEY_BLOCK) < range.y) stage_block(range.x + (uint32_t)(j * KEY_BLOCK), (uint32_t)j);
	}
	const int row_bit = (int)(ty16 % a.cell_tiles), col_bit = a.cell_tiles + (int)(tx % a.cell_tiles);
	// a candidate = (Gaussian id, rectangle-coverage mask): one 8-byte entry (bucket binning) or the list id + the bits above
	// (cell | depth) of its sort key
	PT_DECL(10); PT(t_wave0); TL_DECL; PROBE_DECL;

	// Chunk publication (the backward's replay list): live entries collect in s_pub and leave as FULL chunks of 64 -- a batch of the direct
	// refill holds ~45 entries, and every chunk boundary is a gather round trip for the backward.  Chunk slots: the tile's own block
	// first; a new block of POOL_BLOCK slots is drawn from the shared cursor when the block is used up -- before the blend loop of the
	// batch that may need it, so that the atomic's round trip hides behind the loop.
	uint32_t pub_n = 0, pre_block = 0; bool pre_drawn = false;
	auto flush_chunk = [&](uint32_t cnt) {       // wave-uniform cnt in [1, 64]: the first cnt ids of s_pub become a chunk, the rest moves down
		if (!PUBLISH) return;
		if (blk_left == 0) {
			if (!pre_drawn) { uint32_t nb = 0; if (lane == 0) nb = atomicAdd(a.pool_cursor, (uint32_t)POOL_BLOCK); pre_block = nb; }
			blk_next = gridDim.x * (uint32_t)POOL_BLOCK + __shfl(pre_block, 0, WAVE); blk_left = POOL_BLOCK; pre_drawn = false;
		}
		const uint32_t chunk = blk_next;
		blk_next++; blk_left--;
		uint32_t* c = a.pool + (size_t)chunk * CHUNK_WORDS;
		const uint32_t mine = s_pub[lane], behind = s_pub[WAVE + lane];
		if ((uint32_t)lane < cnt) c[lane] = mine;
		if (lane == 0) { c[CHUNK_PREV] = prev_chunk; c[CHUNK_COUNT] = cnt; }
		prev_chunk = chunk;
		s_pub[lane] = behind;                     // (a partial chunk is the tile's last: what moves down then is never read)
		pub_n -= cnt;
	};
	uint32_t entries_in = 0;                     // entries handed to the blend loop (statistics)

	while (true) {
		bool mine_done = true;
#pragma unroll
		for (int k = 0; k < PPL; k++) mine_done = mine_done && (pyf[k] == PIXEL_DONE);
		const bool all_done = !wave_any(!mine_done);
		if (all_done) break;
		// The memory phases (refill, gather) are a few dozen instructions between long waits: raised priority lets the wave issue its loads
		// at once when it returns from one; the blend loops of the other waves, which only need throughput, fill the rest.
		__builtin_amdgcn_s_setprio(2);
		// ---- refill.  Stage 1 scans the cell's depth-sorted list: every entry carries which tile rows and columns of the cell the
		// Gaussian's rectangle covers (cell_scatter / duplicate_cells), so the rectangle test costs 8 sequential bytes and two shifts per
		// candidate -- at C3 a tile scans ~2400 candidates to find ~250.  Stage 2 gathers the 64-byte Splat line of up to 64 of them,
		// runs the exact ellipse-against-tile test on it and keeps the survivors' lines, compacted in order: ONE scattered line per
		// candidate.  (Until round 3: a 32-byte filter record per candidate, then the Splat line of every survivor -- 478 instead of
		// 253 scattered lines per tile and one more dependent round trip per batch.  The chip serves ~58 G scattered 64-byte lines per
		// second whatever the occupancy -- tools/microbench/gather_latency.hip -- : the forward's 7.5 M lines per frame were 130 us.)
		PT(t_s1a);
		while (ccount < WAVE && pos < range.y) {
			uint32_t key[SCAN_ROUNDS], id[SCAN_ROUNDS];
			if (staged) {
				// half a staged block per step (pos == bpos + bhalf * KEY_BLOCK / 2)
				PT(t_l0);
				if (bhalf == 0 && bpos >= landed) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); landed = issued; }      // not known to have arrived: everything requested so far has now
#pragma unroll
				for (int r = 0; r < SCAN_ROUNDS; r++) { const uint2 v = s_keys[bslot * KEY_BLOCK + bhalf * (KEY_BLOCK / 2) + r * WAVE + lane]; id[r] = v.x; key[r] = v.y; }
				PT_WAIT_VM8(6, 7, t_l0, key[0], key[1], id[0], id[1], key[0], key[1], id[0], id[1]);
				bhalf++;
				if (bhalf == 2) {
					if (issued < range.y) {
						asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // every lane has its entries: the slot may be overwritten
						stage_block(issued, bslot);                                // block j + KEY_RING takes the slot of block j
					}
					bpos += KEY_BLOCK; bhalf = 0; bslot = bslot + 1 == (uint32_t)KEY_RING ? 0u : bslot + 1;
				}
			} else {
			// UNCONDITIONAL loads at clamped indices, the bucket / sort distinction outside the unrolled loop: a load inside a
			// divergent `if (e < end)` is followed by s_waitcnt vmcnt(0) at the end of its block (its value is copied into the
			// merged register there), which serialises the round trips again -- the one-round prefetch this replaces never
			// overlapped anything for that reason.
			const uint32_t last = range.y - 1u;
			PT(t_l0);
			if (a.cell_entries) {
				uint2 v[SCAN_ROUNDS];
#pragma unroll
				for (int r = 0; r < SCAN_ROUNDS; r++) v[r] = a.cell_entries[min(pos + r * WAVE + lane, last)];
#pragma unroll
				for (int r = 0; r < SCAN_ROUNDS; r++) { id[r] = v[r].x; key[r] = v[r].y; }
			} else {
				unsigned long long kk[SCAN_ROUNDS];
#pragma unroll
				for (int r = 0; r < SCAN_ROUNDS; r++) {
					const uint32_t e = min(pos + r * WAVE + lane, last);
					id[r] = a.cell_list[e];
					kk[r] = a.cell_keys[e];
				}
#pragma unroll
				for (int r = 0; r < SCAN_ROUNDS; r++) key[r] = (uint32_t)(kk[r] >> a.mask_shift);
			}
			PT_WAIT_VM8(6, 7, t_l0, key[0], key[1], key[SCAN_ROUNDS - 1], key[SCAN_ROUNDS - 2], id[0], id[1], id[SCAN_ROUNDS - 1], id[SCAN_ROUNDS - 2]);
			}
#pragma unroll
			for (int r = 0; r < SCAN_ROUNDS; r++) {
				const bool have = pos + r * WAVE + lane < range.y;
				const bool rp = have && (((key[r] >> row_bit) & (key[r] >> col_bit)) & 1u);
				const uint64_t m = __ballot(rp);
				if (rp) cand((chead + ccount + __popcll(m & lt_mask)) & (CAND_RING - 1)) = id[r];
				ccount += __popcll(m);
			}
			pos += SCAN_ROUNDS * WAVE;
		}
		PT(t_s1b); PT_ACC(0, t_s1a, t_s1b);
		const uint32_t nc = min(ccount, (uint32_t)WAVE);
		if (nc == 0) break;
		__syncthreads();
		// lane l's candidate lands in row l of s_splat whether it passes or not: the blend loop walks the set bits of the pass mask (no
		// compaction -- sixteen registers of gathered line per lane next to the blend accumulators cost a wave per SIMD)
		bool pass = false, my_lean = false; uint32_t myid = 0;
		if ((uint32_t)lane < nc) {
			myid = cand((chead + lane) & (CAND_RING - 1));
			PT(t_f0);
			const float4* src = reinterpret_cast<const float4*>(a.splats + myid);
			float4 g0 = src[0], g1 = src[1];
			const float4 g2 = src[2], g3 = src[3];
			PT_WAIT_VM8(8, 9, t_f0, g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w);
			s_splat[lane * SPLAT_ROW + 2] = g2; s_splat[lane * SPLAT_ROW + 3] = g3;
			s_splat[lane * SPLAT_ROW + 0] = g0; s_splat[lane * SPLAT_ROW + 1] = g1;
			pass = tile_may_contribute(g0, g1.x, g3.z, tx, ty * ROWS, ROWS);      // g3.z: tau (Splat::aux)
			my_lean = pass && g3.w != 0.f;
		}
		landed = issued;          // the gather's lines (nc >= 1 lanes took part) are younger than every block requested so far: those have arrived
		chead = (chead + nc) & (CAND_RING - 1); ccount -= nc;
		const uint64_t pm = __ballot(pass);                                     // bit j: row j of s_splat is an entry of this batch
		const uint64_t lean_m = __builtin_amdgcn_ballot_w64(my_lean);           // bit j: entry j takes the lean evaluation
		const uint32_t n = (uint32_t)__popcll(pm);
		PT(t_s2b); PT_ACC(1, t_s1b, t_s2b);
		if (n == 0) continue;
		PT(t_g0);
		__syncthreads();
		entries_in += n;
		if (PUBLISH && blk_left == 0 && !pre_drawn && pub_n + n >= (uint32_t)WAVE) {      // this batch may complete a chunk and the block is used up
			uint32_t nb = 0;
			if (lane == 0) nb = atomicAdd(a.pool_cursor, (uint32_t)POOL_BLOCK);
			pre_block = nb; pre_drawn = true;
		}
		// ---- blend.  `live` collects the entries at least one pixel evaluates as contributing: only those are published for
		// the backward replay (at C3 22 % of the entries that pass the tile test are blended by no pixel -- the test is a bound
		// over the tile rectangle, and pixels saturate), and positions (n_contrib) count live entries only.
		uint64_t live = 0ull;
		PT(t_b0); PT_ACC(2, t_g0, t_b0);
		__builtin_amdgcn_s_setprio(0);
		for (uint64_t todo = pm; todo != 0ull; todo &= todo - 1ull) {
			const uint32_t j = (uint32_t)__builtin_ctzll(todo);
			const float4 q0 = s_splat[j * SPLAT_ROW + 0];      // x y ca cb
			const float4 q1 = s_splat[j * SPLAT_ROW + 1];      // cc op r g
			const EntryGeom eg = entry_geom(q0, q1, q0.x - pxf);
			PROBE_EVAL();
			float alpha[PPL]; uint64_t actm[PPL];
			const uint64_t any_m = ((lean_m >> j) & 1ull) ? eval_entry_fwd<true, PPL>(eg, pyf, alpha, actm) : eval_entry_fwd<false, PPL>(eg, pyf, alpha, actm);
			if (any_m == 0ull) continue;

			PROBE_LIVE(actm, PPL);
			uint32_t position = 0u;
			if (PUBLISH) {
				position = consumed + (uint32_t)__popcll(live) + 1u;      // 1-based position in the published sequence
				asm volatile("" : "+v"(position));        // one copy into a vector register per entry (else: one v_mov per strip)
			}
			live |= 1ull << j;
			const float4 q2 = s_splat[j * SPLAT_ROW + 2];      // b dval fx fy
			const float4 q3 = s_splat[j * SPLAT_ROW + 3];      // fz sem0 zview lean
#pragma unroll
			for (int k = 0; k < PPL; k++) {
				if (actm[k] == 0ull) continue;           // wave-uniform: a 16x4 pixel strip the entry does not reach costs nothing
				const float test_T = T[k] * (1.f - alpha[k]);
				const uint64_t stopm = actm[k] & __builtin_amdgcn_ballot_w64(test_T < T_STOP);
				const bool stop = __builtin_amdgcn_inverse_ballot_w64(stopm);
				const bool up = __builtin_amdgcn_inverse_ballot_w64(actm[k] & ~stopm);
				pyf[k] = stop ? PIXEL_DONE : pyf[k];
				const float w = up ? alpha[k] * T[k] : 0.f;   // pixels that do not blend this entry add exactly 0
				C0[k] = fmaf(q1.z, w, C0[k]); C1[k] = fmaf(q1.w, w, C1[k]); C2[k] = fmaf(q2.x, w, C2[k]);
				F0[k] = fmaf(q2.z, w, F0[k]); F1[k] = fmaf(q2.w, w, F1[k]); F2[k] = fmaf(q3.x, w, F2[k]);
				Dp[k] = fmaf(q2.y, w, Dp[k]); S0[k] = fmaf(q3.y, w, S0[k]);
				T[k] = up ? test_T : T[k];
				if (PUBLISH) last_contrib[k] = up ? position : last_contrib[k];
			}
		}
		PT(t_b1); PT_ACC(3, t_b0, t_b1);
		// ---- the live entries of this batch join the pending chunk, in order
		const uint32_t nlive = (uint32_t)__popcll(live);
		if (PUBLISH) {
			if ((live >> lane) & 1ull) s_pub[pub_n + (uint32_t)__popcll(live & lt_mask)] = myid;
			pub_n += nlive;
		}
		consumed += nlive;
		if (PUBLISH && pub_n >= (uint32_t)WAVE) flush_chunk(WAVE);
	}
	if (PUBLISH && pub_n > 0) flush_chunk(pub_n);

	if (PUBLISH && lane == 0) { a.tile_last_chunk[tile] = prev_chunk; a.tile_consumed[tile] = consumed; a.tile_scanned[tile] = min(pos, range.y) - range.x; a.tile_batches[tile] = entries_in; }
	TL_STORE(lane, a.tile_scanned, a.tile_batches, tile);      // timeline build: the wave's life instead of the statistics
	PROBE_FLUSH(0, lane);
	{ PT(t_wave1); PT_ACC(4, t_wave0, t_wave1); PT_ADD(5, 1ull); PT_FLUSH(0, 10, lane); }
	const size_t HW = (size_t)a.H * a.W;
#pragma unroll
	for (int k = 0; k < PPL; k++) {
		if (inside[k]) {
			const size_t pix_id = (size_t)a.W * (py0 + 4 * k) + px;
			a.final_T[pix_id] = (float)(1.0 - (double)T[k]);
			if (PUBLISH) a.n_contrib[pix_id] = last_contrib[k];
			if (a.has_color) {
				// background composite in the epilogue: constant colour (forward.cu:392-394) or, with an environment map, the per-pixel
				// background of gaussian_renderer/__init__.py:93-94 (render = foreground + (1 - O) * background, T = 1 - O)
				const float b0 = a.bg[0] + (a.bg_image ? a.bg_image[0 * HW + pix_id] : 0.f), b1 = a.bg[1] + (a.bg_image ? a.bg_image[1 * HW + pix_id] : 0.f),
					b2 = a.bg[2] + (a.bg_image ? a.bg_image[2 * HW + pix_id] : 0.f);
				st_stream(a.out_color + 0 * HW + pix_id, C0[k] + T[k] * b0);
				st_stream(a.out_color + 1 * HW + pix_id, C1[k] + T[k] * b1);
				st_stream(a.out_color + 2 * HW + pix_id, C2[k] + T[k] * b2);
			}
			if (a.has_flow) { st_stream(a.out_flow + 0 * HW + pix_id, F0[k]); st_stream(a.out_flow + 1 * HW + pix_id, F1[k]); st_stream(a.out_flow + 2 * HW + pix_id, F2[k]); }
			if (a.has_sem) st_stream(a.out_semantic + pix_id, S0[k]);
			st_stream(a.out_depth + pix_id, Dp[k]);
		}
	}
}

// Fourteen wave64 sums through LDS (round 2 used a transposing register butterfly of permlane swaps and bank-masked DPP adds: git
// history, EXPERIMENTS.md).  Measured on gfx950 (tools/microbench/valu_rates.hip, profiles/r03/valu_rates.txt): v_permlane32_swap /
// v_permlane16_swap retire at a QUARTER of the fp32 rate (8.2 cycles per wave64 instruction, like v_exp_f32), DPP adds at half rate
// (4.3) -- the butterfly's 11 swaps + 11 adds + 10 DPP adds were ~165 VALU cycles per entry -- while the LDS pipe idled.  Here every
// lane stores its 14 partial sums ([value][lane], conflict-free; rows of exactly 64 floats, so that seven ds_write2st64_b32 address
// all rows from ONE base register), lane l reads back a quarter (l & 3) of value row l >> 2 as four 16-byte words and adds its 16
// numbers; two quad DPP adds finish: 15 full-rate adds + 2 DPP adds of VALU work.  With rows of 64 floats the sixteen lanes of a
// ds_read_b128 group (four rows x four quarters) would hit the same four bank quads, so the i-th read of a lane starts at word
// (i + row) & 3 of its quarter: the four rows of a group (0,3,5,6 / 1,2,4,7 / 8,11,13,10 / 9,10,12,11 -- lanes 56..63 re-read rows
// 10 and 11, nobody uses their sums) differ mod 4.  The LDS operations of one wave execute in order: no barrier.
constexpr int RED_ROWS = 14, RED_STRIDE = WAVE;      // floats per value row
struct RedAddr { const float4* r[4]; };
__device__ __forceinline__ RedAddr red_addresses(const float* s_red, int lane) {
	const int row = (lane >> 2) < RED_ROWS ? (lane >> 2) : (lane >> 2) - 4;
	RedAddr ra;
#pragma unroll
	for (int i = 0; i < 4; i++) ra.r[i] = reinterpret_cast<const float4*>(s_red + row * RED_STRIDE + (lane & 3) * 16 + 4 * ((i + row) & 3));
	return ra;
}
__device__ __forceinline__ float wave_sum14_lds(float* s_red, const RedAddr& ra, int lane, float x0, float x1, float x2, float x3, float x4, float x5, float x6,
	float x7, float x8, float x9, float x10, float x11, float x12, float x13) {
	float* w = s_red + lane;
	w[0 * RED_STRIDE] = x0; w[1 * RED_STRIDE] = x1; w[2 * RED_STRIDE] = x2; w[3 * RED_STRIDE] = x3; w[4 * RED_STRIDE] = x4;
	w[5 * RED_STRIDE] = x5; w[6 * RED_STRIDE] = x6; w[7 * RED_STRIDE] = x7; w[8 * RED_STRIDE] = x8; w[9 * RED_STRIDE] = x9;
	w[10 * RED_STRIDE] = x10; w[11 * RED_STRIDE] = x11; w[12 * RED_STRIDE] = x12; w[13 * RED_STRIDE] = x13;
	const float4 a = *ra.r[0], b = *ra.r[1], c = *ra.r[2], d = *ra.r[3];
	float v = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w));
	v += ((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w));
	int x = __float_as_int(v);
	v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false)); x = __float_as_int(v);      // quad_perm [1,0,3,2]
	v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false));                               // quad_perm [2,3,0,1]
	return v;
}

// One 16x4 strip of one entry in the backward replay (backward.cu:545-644 for the lane's pixel of that strip).
struct BwdSums { float op, mx, my, ca, cb, cc, c0, c1, c2, d, f0, f1, f2, s; };      // per-Gaussian partial sums of this lane
struct BwdEntry { float r, g, b, dval, fx, fy, fz, sem, dx; };                       // the entry's payload (wave-uniform) and the lane's dx
struct BwdPixel { float gC0, gC1, gC2, gD, gF0, gF1, gF2, gS, tfo, tfb; };           // the pixel's upstream gradients
template <bool INIT>
__device__ __forceinline__ void bwd_acc(float& s, float a, float b) { s = INIT ? a * b : fmaf(a, b, s); }
template <bool INIT>
__device__ __forceinline__ void bwd_strip(BwdSums& v, const BwdEntry& e, const BwdPixel& p, float al, float G, float dy, float& T, float& Bsum,
	bool do_color, bool do_flow, bool do_sem, bool do_depth, bool do_opacity) {
	const float rinv = __builtin_amdgcn_rcpf(1.f - al);
	T = T * rinv;
	const float dch = al * T;
	// backward.cu:578-607 keeps, per channel, the blend A_ch of everything behind this entry
	// (A' = A + alpha (c - A)) and forms dL/dalpha = sum_ch (c_ch - A_ch) g_ch.  Only the scalar
	// B = sum_ch A_ch g_ch is ever used, and it obeys the same recurrence
	//   B' = B + alpha (cg - B),  cg = sum_ch c_ch g_ch,
	// so one accumulator per pixel replaces the eight per-channel ones.
	float cg = 0.f;
	if (do_color) {
		cg += e.r * p.gC0; cg += e.g * p.gC1; cg += e.b * p.gC2;
		bwd_acc<INIT>(v.c0, dch, p.gC0); bwd_acc<INIT>(v.c1, dch, p.gC1); bwd_acc<INIT>(v.c2, dch, p.gC2);
	} else if (INIT) { v.c0 = v.c1 = v.c2 = 0.f; }
	if (do_flow) {
		cg += e.fx * p.gF0; cg += e.fy * p.gF1; cg += e.fz * p.gF2;
		bwd_acc<INIT>(v.f0, dch, p.gF0); bwd_acc<INIT>(v.f1, dch, p.gF1); bwd_acc<INIT>(v.f2, dch, p.gF2);
	} else if (INIT) { v.f0 = v.f1 = v.f2 = 0.f; }
	if (do_sem) { cg += e.sem * p.gS; bwd_acc<INIT>(v.s, dch, p.gS); } else if (INIT) { v.s = 0.f; }
	if (do_depth) { cg += e.dval * p.gD; bwd_acc<INIT>(v.d, dch, p.gD); } else if (INIT) { v.d = 0.f; }
	float dL_dalpha = cg - Bsum;
	Bsum += al * dL_dalpha;
	if (do_opacity) dL_dalpha += p.tfo * rinv;  // before the *= T: reference quirk (backward.cu:612-614)
	dL_dalpha = dL_dalpha * T - p.tfb * rinv;
	// the lane's four pixels share the column, i.e. dx: only S0 = sum L, Sy = sum L dy, Syy = sum L dy^2 are accumulated per strip;
	// Sx = dx S0, Sxx = dx^2 S0, Sxy = dx Sy are formed once per entry (bwd_finish_moments)
	const float L = G * dL_dalpha;
	const float Ly = L * dy;
	bwd_acc<INIT>(v.op, G, dL_dalpha); bwd_acc<INIT>(v.my, L, dy); bwd_acc<INIT>(v.cc, Ly, dy);
}
__device__ __forceinline__ void bwd_finish_moments(BwdSums& v, float dx) {
	v.mx = dx * v.op; v.ca = dx * v.mx; v.cb = dx * v.my;
}

// The lane's pixels against one entry of the replay: alpha, G, dy and the lane masks of the pixels the entry contributed to.  LEAN: the
// entry needs neither the power test nor the clamp (eval_pixel) AND lies before every pixel's last contributor (wave-uniform: the
// position test of backward.cu:553 is then true for all of them).
template <bool LEAN, int PPL>
__device__ __forceinline__ uint64_t eval_entry_bwd(const EntryGeom& eg, float pyf0, int contributor, const int (&last_contributor)[PPL],
	float (&alpha)[PPL], float (&G)[PPL], float (&dy)[PPL], uint64_t (&actm)[PPL]) {
	uint64_t any_m = 0ull;
#pragma unroll
	for (int k = 0; k < PPL; k++) {
		float power;
		eval_pixel<LEAN>(eg, pyf0 + (float)(4 * k), dy[k], power, G[k], alpha[k]);
		actm[k] = LEAN ? __builtin_amdgcn_ballot_w64(!(alpha[k] < ALPHA_MIN))
		               : (__builtin_amdgcn_ballot_w64(contributor < last_contributor[k]) & __builtin_amdgcn_ballot_w64(!(power > 0.0f)) &
		                  __builtin_amdgcn_ballot_w64(!(alpha[k] < ALPHA_MIN)));
		any_m |= actm[k];
	}
	return any_m;
}

// FULL: colour, depth, opacity, flow and semantic gradients all present (the training configuration) --
// the channel switches fold at compile time; otherwise they are wave-uniform run-time flags.
template <int PPL, bool FULL>
__global__ void __launch_bounds__(WAVE) render_bwd_v2_kernel(RenderV2BwdArgs a) {
	constexpr int ROWS = 4 * PPL;
	const bool do_color = FULL || a.do_color, do_flow = FULL || a.do_flow, do_sem = FULL || a.do_sem, do_depth = FULL || a.do_depth, do_opacity = FULL || a.do_opacity;
	// 4096 + 3584 bytes: 21 one-wave workgroups per CU (8768 bytes until round 3 capped the kernel at 18 = 4.5 waves per SIMD whatever
	// its registers allowed)
	// rows 80 bytes apart: conflict-free ds_write_b128 (see the forward) where the registers cap the kernel at 16 workgroups per CU anyway
	// (PPL = 4: 5120 + 3584 bytes = 18 per CU); the half-tile kernels (88 registers: 20 per CU) keep the dense 7680-byte layout
	constexpr int BROW = PPL == 4 ? SPLAT_ROW : 4;
	__shared__ float4 s_splat[WAVE * BROW];
	__shared__ __attribute__((aligned(16))) float s_red[RED_ROWS * RED_STRIDE];
	const int lane = threadIdx.x;
	const uint32_t tile = a.tile_order ? a.tile_order[blockIdx.x] : blockIdx.x;
	const uint32_t tx = tile % a.gx, ty = tile / a.gx;
	const uint32_t px = tx * TILE_X + (lane & 15);
	const uint32_t py0 = ty * ROWS + (lane >> 4);
	const float pxf = (float)px;
	const size_t HW = (size_t)a.H * a.W;
	bool inside[PPL];
	const float pyf0 = (float)py0;
	float T[PPL], tfo[PPL], tfb[PPL];      // tfo = dL/dO * T_final, tfb = T_final * (bg . dL/dC)
	int last_contributor[PPL];
	float Bsum[PPL];          // sum_ch A_ch * dL/dC_ch of the suffix blend A behind the current entry (see below)
	float gC0[PPL], gC1[PPL], gC2[PPL], gF0[PPL], gF1[PPL], gF2[PPL], gD[PPL], gS[PPL];
	int max_contrib = 0, min_contrib = 0x7fffffff;
	const int slot = lane >> 2;                      // the gacc slot this lane's quad finishes in wave_sum14_lds
	const RedAddr red = red_addresses(s_red, lane);
	const bool writer = (lane & 3) == 0 && slot < GACC_USED;
#pragma unroll
	for (int k = 0; k < PPL; k++) {
		const uint32_t py = py0 + 4 * k;
		inside[k] = px < (uint32_t)a.W && py < (uint32_t)a.H;
		const size_t pix_id = (size_t)a.W * py + px;
		const float T_final = inside[k] ? (float)(1.0 - (double)a.final_T[pix_id]) : 0.f;
		float gO = 0.f;
		T[k] = T_final;
		last_contributor[k] = inside[k] ? (int)a.n_contrib[pix_id] : 0;
		max_contrib = max(max_contrib, last_contributor[k]);
		min_contrib = min(min_contrib, last_contributor[k]);      // an out-of-image pixel has 0: a tile with one never takes the lean path
		Bsum[k] = 0.f;
		gC0[k] = gC1[k] = gC2[k] = gF0[k] = gF1[k] = gF2[k] = gD[k] = gS[k] = 0.f;
		if (inside[k]) {
			if (do_color) { gC0[k] = ld_stream(a.dL_dpix + 0 * HW + pix_id); gC1[k] = ld_stream(a.dL_dpix + 1 * HW + pix_id); gC2[k] = ld_stream(a.dL_dpix + 2 * HW + pix_id); }
			if (do_flow) { gF0[k] = ld_stream(a.dL_dpix_flow + 0 * HW + pix_id); gF1[k] = ld_stream(a.dL_dpix_flow + 1 * HW + pix_id); gF2[k] = ld_stream(a.dL_dpix_flow + 2 * HW + pix_id); }
			if (do_sem) gS[k] = ld_stream(a.dL_dpix_sem + pix_id);
			if (do_depth) gD[k] = ld_stream(a.dL_dpix_depth + pix_id);
			if (do_opacity && a.dL_dpix_opacity) gO = ld_stream(a.dL_dpix_opacity + pix_id);
		}
		float b = 0.f;
		b += a.bg[0] * gC0[k]; b += a.bg[1] * gC1[k]; b += a.bg[2] * gC2[k];
		if (a.bg_image && inside[k]) {
			// The reference composites the environment map in Python: render = fg + (1 - O) * bg_image, so -sum_c bg_c dL/dC_c reaches the
			// rasterizer as part of grad_img_opacity and takes the opacity path WITH its extra factor T (backward.cu:612-614, the
			// "opacity-T quirk") -- unlike the constant background above, which the rasterizer differentiates itself.
			gO -= a.bg_image[0 * HW + pix_id] * gC0[k] + a.bg_image[1 * HW + pix_id] * gC1[k] + a.bg_image[2 * HW + pix_id] * gC2[k];
			if (a.dL_dbg_image) {      // d render / d background = T_final
				st_stream(a.dL_dbg_image + 0 * HW + pix_id, T_final * gC0[k]); st_stream(a.dL_dbg_image + 1 * HW + pix_id, T_final * gC1[k]); st_stream(a.dL_dbg_image + 2 * HW + pix_id, T_final * gC2[k]);
			}
		}
		tfo[k] = gO * T_final; tfb[k] = T_final * b;
	}
#pragma unroll
	for (int off = WAVE / 2; off > 0; off >>= 1) {
		max_contrib = max(max_contrib, __shfl_xor(max_contrib, off, WAVE));
		min_contrib = min(min_contrib, __shfl_xor(min_contrib, off, WAVE));
	}
	max_contrib = __builtin_amdgcn_readfirstlane(max_contrib);      // wave-uniform from here: the per-entry tests against them are scalar
	min_contrib = __builtin_amdgcn_readfirstlane(min_contrib);
	uint32_t chunk = a.tile_last_chunk[tile];
	int base = (int)a.tile_consumed[tile];        // one past the last position of the current chunk
	TL_DECL; PROBE_DECL;
	PT_DECL(9); PT(t_wave0);
	{
		// timing build: the prologue's pixel-state loads have landed when the first of them can be used
		float pt_probe = gC0[0] + T[0];
		(void)pt_probe;
#ifdef ADGS_PHASE_TIMING
		asm volatile("s_waitcnt vmcnt(0)" : "+v"(pt_probe) :: "memory");
#endif
		PT(t_p1); PT_ACC(8, t_wave0, t_p1);
	}
	// The replay list is a chain of chunks ([64 ids, prev, count]), walked from the tile's last chunk to its first.  Ids and link words of a
	// chunk are two independent vector loads.  The walk is pipelined two chunks deep: while chunk c is replayed, the Splat lines of chunk
	// c - 1 are in flight into registers (its ids arrived while c + 1 was replayed) and the ids of chunk c - 2 are requested -- at a
	// chunk boundary nothing but a wait for loads that landed long ago and four LDS stores is left.  (Until round 3: header -> ids ->
	// gather, three dependent round trips per 64 entries; a gather of 64 scattered lines is ~6 us under load.)
	uint32_t my_id = 0, link = 0;                 // current chunk -- lane j: id of entry j; lanes 0 / 1: prev chunk / entry count
	uint32_t nx_id = 0, nx_link = 0;              // the chunk behind it in the walk (`prev`)
	float4 nx0, nx1, nx2, nx3;                    // ... and this lane's Splat line of it
	nx0 = nx1 = nx2 = nx3 = make_float4(0.f, 0.f, 0.f, 0.f);
	auto request_rows = [&](uint32_t id, int cnt, float4& r0, float4& r1, float4& r2, float4& r3) {
		if (lane < cnt) {
			const float4* src = reinterpret_cast<const float4*>(a.splats + id);
			r0 = src[0]; r1 = src[1]; r2 = src[2]; r3 = src[3];
			if (!FULL && a.sem_src) r3.y = a.sem_src[(size_t)id * a.sem_stride];      // an extra semantic channel's replay
		}
	};
	if (chunk != NO_CHUNK) {
		const uint32_t* c = a.pool + (size_t)chunk * CHUNK_WORDS;
		my_id = c[lane]; link = c[CHUNK_PREV + (lane & 1)];
	}
	bool rows_ready = false;                      // nx0 .. nx3 hold the CURRENT chunk's lines (requested one chunk ago)
	while (chunk != NO_CHUNK) {
		PT(t_c0);
		__builtin_amdgcn_s_setprio(2);      // a chunk boundary is a dozen instructions between two waits: do not starve behind the older waves' entry loops
		const uint32_t prev = (uint32_t)__builtin_amdgcn_readlane((int)link, 0);
		const int n = __builtin_amdgcn_readlane((int)link, 1);
#ifdef ADGS_PHASE_TIMING
		asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
		PT(t_c1); PT_ACC(0, t_c0, t_c1); PT_ADD(6, 1ull);
		base -= n;
		const uint32_t cur_id = my_id;
		float4 c0, c1, c2, c3;
		if (rows_ready) { c0 = nx0; c1 = nx1; c2 = nx2; c3 = nx3; }
		else {
			// the tile's first chunk: its lines now, the next chunk's ids at the same time
			if (prev != NO_CHUNK) { const uint32_t* cn = a.pool + (size_t)prev * CHUNK_WORDS; nx_id = cn[lane]; nx_link = cn[CHUNK_PREV + (lane & 1)]; }
			c0 = c1 = c2 = c3 = make_float4(0.f, 0.f, 0.f, 0.f);
			request_rows(cur_id, n, c0, c1, c2, c3);
		}
		{
			__syncthreads();
			bool my_lean = false;
			if (lane < n) {
				s_splat[lane * BROW + 0] = c0; s_splat[lane * BROW + 1] = c1; s_splat[lane * BROW + 2] = c2; s_splat[lane * BROW + 3] = c3;
				my_lean = c3.w != 0.f;
			}
			// the chunk behind this one: its lines fly while this chunk is replayed; the ids of the one behind that follow
			uint32_t nn_id = 0, nn_link = 0;
			if (prev != NO_CHUNK) {
				{
					const uint32_t pprev = (uint32_t)__builtin_amdgcn_readlane((int)nx_link, 0);
					request_rows(nx_id, __builtin_amdgcn_readlane((int)nx_link, 1), nx0, nx1, nx2, nx3);
					if (pprev != NO_CHUNK) { const uint32_t* cp = a.pool + (size_t)pprev * CHUNK_WORDS; nn_id = cp[lane]; nn_link = cp[CHUNK_PREV + (lane & 1)]; }
					rows_ready = true;
				}
			}
			my_id = nx_id; link = nx_link; nx_id = nn_id; nx_link = nn_link;
			// bit j: entry j takes the lean evaluation -- its Gaussian allows it and its position lies before every pixel's last contributor
			// (contributor = base + j < min_contrib: the position test holds for all pixels)
			const int n_before = min_contrib - base;
			const uint64_t lean_m = __builtin_amdgcn_ballot_w64(my_lean) & (n_before >= WAVE ? ~0ull : n_before <= 0 ? 0ull : ((1ull << n_before) - 1ull));
			__syncthreads();
			__builtin_amdgcn_s_setprio(0);
			PT(t_g1); PT_ACC(1, t_c1, t_g1);
			// entries at or behind every pixel's last contributor are not replayed: the loop starts below them
			const int j_first = min(n - 1, max_contrib - base - 1);
			for (int j = j_first; j >= 0; j--) {
				const int contributor = base + j;
				PT_ADD(7, 1ull);
				const float4 q0 = s_splat[j * BROW + 0], q1 = s_splat[j * BROW + 1];
				const float dx = q0.x - pxf;
				const EntryGeom eg = entry_geom(q0, q1, dx);
				// the entry's Gaussian: a scalar from the id register (lane j holds entry j), early -- its only use is the atomic's address
				const uint32_t gid = (uint32_t)__builtin_amdgcn_readlane((int)cur_id, j);
				PROBE_EVAL();
				float alpha[PPL], G[PPL], dy[PPL]; uint64_t actm[PPL];      // lane masks on the scalar unit, as in the forward
				const uint64_t any_m = ((lean_m >> j) & 1ull) ? eval_entry_bwd<true, PPL>(eg, pyf0, contributor, last_contributor, alpha, G, dy, actm)
				                            : eval_entry_bwd<false, PPL>(eg, pyf0, contributor, last_contributor, alpha, G, dy, actm);
				if (any_m == 0ull) continue;
				const float4 q2 = s_splat[j * BROW + 2];
				const float4 q3 = s_splat[j * BROW + 3];
				PROBE_LIVE(actm, PPL);
				// geometric part: with L = G * dL/dalpha per pixel, the reference's six sums are linear in
				//   S0 = sum L, Sx = sum L dx, Sy = sum L dy, Sxx = sum L dx^2, Sxy = sum L dx dy, Syy = sum L dy^2
				// (dL/dmean2D = -op*(ca Sx + cb Sy)*W/2 ..., dL/dconic = -op/2 * S.., dL/dopacity = S0); the
				// per-Gaussian factors are applied once per Gaussian in the preprocess backward.
				BwdSums v;
				const BwdEntry be = { q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, dx };
				// The first strip the entry reaches INITIALISES the 14 sums (every lane writes: a lane the entry does not reach computes
				// with alpha = G = 0, i.e. adds exactly nothing and leaves its T and B as they are); the others accumulate under exec.
				// (Only strip 0 can initialise: one straight-line path per possible first strip, or a flag carried through the unrolled
				// loop, leaves hipcc with 10 - 30 register copies per entry at the joins -- more than the 14 zeroing moves it saves.)
#define ADGS_BWD_PIXEL(k) const BwdPixel bp = { gC0[k], gC1[k], gC2[k], gD[k], gF0[k], gF1[k], gF2[k], gS[k], tfo[k], tfb[k] }
#define ADGS_BWD_INITK(k) { ADGS_BWD_PIXEL(k); const bool act = __builtin_amdgcn_inverse_ballot_w64(actm[k]); \
	bwd_strip<true>(v, be, bp, act ? alpha[k] : 0.f, act ? G[k] : 0.f, dy[k], T[k], Bsum[k], do_color, do_flow, do_sem, do_depth, do_opacity); }
#define ADGS_BWD_ACCK(k) if (__builtin_amdgcn_inverse_ballot_w64(actm[k])) { ADGS_BWD_PIXEL(k); \
	bwd_strip<false>(v, be, bp, alpha[k], G[k], dy[k], T[k], Bsum[k], do_color, do_flow, do_sem, do_depth, do_opacity); }
				if (actm[0] != 0ull) { ADGS_BWD_INITK(0) }
				else v.op = v.my = v.cc = v.c0 = v.c1 = v.c2 = v.d = v.f0 = v.f1 = v.f2 = v.s = 0.f;
#pragma unroll
				for (int k = 1; k < PPL; k++) { ADGS_BWD_ACCK(k) }
				bwd_finish_moments(v, dx);
#undef ADGS_BWD_PIXEL
#undef ADGS_BWD_INITK
#undef ADGS_BWD_ACCK
				// 14 wave sums through LDS (slot k of the 64-byte gradient line ends up in the quad of lanes 4k .. 4k+3) -> one atomic
				// instruction on one 64-byte line.  Absent channels stay exactly 0.
				PT(t_r0);
				const float out = wave_sum14_lds(s_red, red, lane, v.op, v.mx, v.my, v.ca, v.cb, v.cc, v.c0, v.c1, v.c2, v.d, v.f0, v.f1, v.f2, v.s);
				if (!FULL && a.sem_dst) {
					if (writer && (slot < 6 || slot == GACC_USED - 1)) atomicAdd(slot == GACC_USED - 1 ? a.sem_dst + (size_t)gid * a.sem_stride : a.gacc + (size_t)gid * GACC_STRIDE + slot, out);
				} else if (writer) atomicAdd(a.gacc + (size_t)gid * GACC_STRIDE + slot, out);
				PT(t_r1); PT_ACC(3, t_r0, t_r1);
			}
			PT(t_e1); PT_ACC(2, t_g1, t_e1);
		}
		chunk = prev;
	}
	__builtin_amdgcn_s_setprio(0);
	{ PT(t_wave1); PT_ACC(4, t_wave0, t_wave1); PT_ADD(5, 1ull); PT_FLUSH(16, 9, lane); }
	TL_STORE(lane, a.tl_start, a.tl_end, tile);
	PROBE_FLUSH(8, lane);
}

// Semantic channels 1 .. D_S-1 of the forward image (RenderV2SemFwdArgs): the replay loop of the backward without the gradients.
template <int PPL>
__global__ void __launch_bounds__(WAVE, 8) render_sem_fwd_v2_kernel(RenderV2SemFwdArgs a) {
	constexpr int ROWS = 4 * PPL, NCH = 4;
	__shared__ float4 s_geo[WAVE * 2];
	__shared__ float4 s_sem[WAVE];
	const int lane = threadIdx.x;
	const uint32_t tile = blockIdx.x;
	const uint32_t tx = tile % a.gx, ty = tile / a.gx;
	const uint32_t px = tx * TILE_X + (lane & 15);
	const uint32_t py0 = ty * ROWS + (lane >> 4);
	const float pxf = (float)px, pyf0 = (float)py0;
	const size_t HW = (size_t)a.H * a.W;
	bool inside[PPL]; int last_contributor[PPL]; float S[PPL][NCH];
	int max_contrib = 0;
#pragma unroll
	for (int k = 0; k < PPL; k++) {
		const uint32_t py = py0 + 4 * k;
		inside[k] = px < (uint32_t)a.W && py < (uint32_t)a.H;
		last_contributor[k] = inside[k] ? (int)a.n_contrib[(size_t)a.W * py + px] : 0;
		max_contrib = max(max_contrib, last_contributor[k]);
#pragma unroll
		for (int i = 0; i < NCH; i++) S[k][i] = 0.f;
	}
#pragma unroll
	for (int off = WAVE / 2; off > 0; off >>= 1) max_contrib = max(max_contrib, __shfl_xor(max_contrib, off, WAVE));
	uint32_t chunk = a.tile_last_chunk[tile];
	int base = (int)a.tile_consumed[tile];
	while (chunk != NO_CHUNK) {
		const uint32_t* c = a.pool + (size_t)chunk * CHUNK_WORDS;
		const uint32_t prev = c[CHUNK_PREV];
		const int n = (int)c[CHUNK_COUNT];
		base -= n;
		if (base < max_contrib) {
			__syncthreads();
			if (lane < n) {
				const uint32_t id = c[lane];
				const float4* src = reinterpret_cast<const float4*>(a.splats + id);
				s_geo[lane * 2 + 0] = src[0];
				s_geo[lane * 2 + 1] = src[1];
				const float* sv = a.semantic + (size_t)id * a.D_S + a.c0;
				float4 q = make_float4(sv[0], 0.f, 0.f, 0.f);
				if (a.nch > 1) q.y = sv[1];
				if (a.nch > 2) q.z = sv[2];
				if (a.nch > 3) q.w = sv[3];
				s_sem[lane] = q;
			}
			__syncthreads();
			for (int j = n - 1; j >= 0; j--) {
				const int contributor = base + j;
				if (contributor >= max_contrib) continue;
				const float4 q0 = s_geo[j * 2 + 0], q1 = s_geo[j * 2 + 1];
				const EntryGeom eg = entry_geom(q0, q1, q0.x - pxf);
				const float4 sv = s_sem[j];
#pragma unroll
				for (int k = 0; k < PPL; k++) {
					float dy, power, G, alpha;
					eval_pixel<false>(eg, pyf0 + (float)(4 * k), dy, power, G, alpha);      // same bits as the lean form where that one applies
					const bool act = contributor < last_contributor[k] && !(power > 0.0f) && !(alpha < ALPHA_MIN);
					const float al = act ? alpha : 0.f;
					S[k][0] = fmaf(al, sv.x - S[k][0], S[k][0]); S[k][1] = fmaf(al, sv.y - S[k][1], S[k][1]);
					S[k][2] = fmaf(al, sv.z - S[k][2], S[k][2]); S[k][3] = fmaf(al, sv.w - S[k][3], S[k][3]);
				}
			}
		}
		chunk = prev;
	}
#pragma unroll
	for (int k = 0; k < PPL; k++) {
		if (!inside[k]) continue;
		const size_t pix_id = (size_t)a.W * (py0 + 4 * k) + px;
#pragma unroll
		for (int i = 0; i < NCH; i++) if (i < a.nch) a.out_semantic[(size_t)(a.c0 + i) * HW + pix_id] = S[k][i];
	}
}

// per-tile bookkeeping reset + pool cursor
__global__ void __launch_bounds__(256) reset_tiles_kernel(uint32_t T, uint32_t* last_chunk, uint32_t* consumed, uint32_t* cursor) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < T) { last_chunk[i] = NO_CHUNK; consumed[i] = 0; }
	if (i == 0) *cursor = 0;
}

} // namespace

int launch_render_fwd_v2(const RenderV2FwdArgs& a, hipStream_t stream) {
	const uint32_t T = (uint32_t)a.gx * a.gy;           // a.gy counts WAVE tiles (16 x 4*ppl pixels)
	if (!a.publish) {
		if (a.ppl == 1) hipLaunchKernelGGL((render_fwd_v2_kernel<1, false>), dim3(T), dim3(WAVE), 0, stream, a);
		else if (a.ppl == 2) hipLaunchKernelGGL((render_fwd_v2_kernel<2, false>), dim3(T), dim3(WAVE), 0, stream, a);
		else hipLaunchKernelGGL((render_fwd_v2_kernel<4, false>), dim3(T), dim3(WAVE), 0, stream, a);
	} else if (a.ppl == 1) hipLaunchKernelGGL((render_fwd_v2_kernel<1, true>), dim3(T), dim3(WAVE), 0, stream, a);
	else if (a.ppl == 2) hipLaunchKernelGGL((render_fwd_v2_kernel<2, true>), dim3(T), dim3(WAVE), 0, stream, a);
	else hipLaunchKernelGGL((render_fwd_v2_kernel<4, true>), dim3(T), dim3(WAVE), 0, stream, a);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
// Longest-list-first order of the tiles for the backward (the forward recorded how many entries every tile consumed).  9600 one-wave
// workgroups over 4096 wave slots are ~2.3 rounds; without this the last round ends with whatever long tiles happen to sit at the end of
// the grid (backward 506 -> 466 us at C3).  Since round 5 the FORWARD launches it, right behind its blend kernel: the order serves this
// frame's backward (img.tile_order) and, as a copy in the camera's hint buffer, the next forward of the same camera.  (Ordering the forward by
// the length of the tile's cell list was measured and does not help.)
// Round 6: K workgroups instead of ONE (a 1024-bucket counting sort whose LDS atomics serialised on the hot buckets: 11.4 us alone on the
// critical path between the two blend kernels).  Workgroup w owns the tiles t = w (mod K) -- a 1-in-K sample of the whole image, so all
// slices have the same length distribution --, ranks them among each other by counting (<= 256 broadcast LDS reads per tile, no atomics),
// and the slices are interleaved: slot = rank * K + w.  The result is the exact descending order within every slice and within a few
// positions of it globally, which is all a dispatch order needs.
namespace {
constexpr int TO_THREADS = 256;
__global__ void __launch_bounds__(TO_THREADS) tile_order_kernel(int T, int K, const uint32_t* __restrict__ consumed, uint32_t* __restrict__ order, uint32_t* __restrict__ order_copy,
	const float* __restrict__ view) {
	__shared__ __attribute__((aligned(16))) uint32_t s_key[TO_THREADS];
	const int w = blockIdx.x, tid = threadIdx.x;
	if (w == 0 && order_copy && view && tid < 16) reinterpret_cast<float*>(order_copy + T)[tid] = view[tid];      // the pose this hint belongs to
	const int n = (T - w + K - 1) / K;             // tiles of this slice: w, w + K, ...  (n <= TO_THREADS: launch_tile_order)
	const int t = w + tid * K;
	// one word per tile, unique inside the slice: (entries consumed, clamped) above (255 - position) -- a longer list first, equal lengths in tile order
	const uint32_t mine = tid < n ? (min(consumed[t], 0xffffffu) << 8) | (uint32_t)(TO_THREADS - 1 - tid) : 0u;
	s_key[tid] = mine;
	__syncthreads();
	if (tid >= n) return;
	uint32_t rank = 0;
#pragma unroll 4
	for (int j = 0; j < TO_THREADS / 4; j++) {      // (the words behind the slice are 0: never above a tile's word)
		const uint4 o = reinterpret_cast<const uint4*>(s_key)[j];
		rank += (o.x > mine ? 1u : 0u) + (o.y > mine ? 1u : 0u) + (o.z > mine ? 1u : 0u) + (o.w > mine ? 1u : 0u);
	}
	// slot = rank * K + w < T: slices differ in length by at most one, and the longer ones are the first T mod K
	const uint32_t at = rank * (uint32_t)K + (uint32_t)w;
	order[at] = (uint32_t)t;
	if (order_copy) order_copy[at] = (uint32_t)t;
}
} // namespace
int launch_tile_order(int ntiles, const uint32_t* tile_consumed, uint32_t* order, hipStream_t stream, uint32_t* order_copy, const float* view) {
	if (ntiles <= 0) return 0;
	const int K = (ntiles + TO_THREADS - 1) / TO_THREADS;
	hipLaunchKernelGGL(tile_order_kernel, dim3(K), dim3(TO_THREADS), 0, stream, ntiles, K, tile_consumed, order, order_copy, view);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

int launch_render_sem_fwd_v2(const RenderV2SemFwdArgs& a, hipStream_t stream) {
	const uint32_t T = (uint32_t)a.gx * a.gy;
	if (a.ppl == 1) hipLaunchKernelGGL(render_sem_fwd_v2_kernel<1>, dim3(T), dim3(WAVE), 0, stream, a);
	else if (a.ppl == 2) hipLaunchKernelGGL(render_sem_fwd_v2_kernel<2>, dim3(T), dim3(WAVE), 0, stream, a);
	else hipLaunchKernelGGL(render_sem_fwd_v2_kernel<4>, dim3(T), dim3(WAVE), 0, stream, a);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

int launch_render_bwd_v2(const RenderV2BwdArgs& a, hipStream_t stream) {
	const uint32_t T = (uint32_t)a.gx * a.gy;
	const bool full = a.do_color && a.do_flow && a.do_sem && a.do_depth && a.do_opacity && a.dL_dpix_opacity != nullptr && !a.sem_src && !a.sem_dst;
	if (a.ppl == 1) {
		if (full) hipLaunchKernelGGL((render_bwd_v2_kernel<1, true>), dim3(T), dim3(WAVE), 0, stream, a);
		else hipLaunchKernelGGL((render_bwd_v2_kernel<1, false>), dim3(T), dim3(WAVE), 0, stream, a);
	} else if (a.ppl == 2) {
		if (full) hipLaunchKernelGGL((render_bwd_v2_kernel<2, true>), dim3(T), dim3(WAVE), 0, stream, a);
		else hipLaunchKernelGGL((render_bwd_v2_kernel<2, false>), dim3(T), dim3(WAVE), 0, stream, a);
	} else {
		if (full) hipLaunchKernelGGL((render_bwd_v2_kernel<4, true>), dim3(T), dim3(WAVE), 0, stream, a);
		else hipLaunchKernelGGL((render_bwd_v2_kernel<4, false>), dim3(T), dim3(WAVE), 0, stream, a);
	}
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

} // namespace adgs

