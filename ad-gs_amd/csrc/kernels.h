// Internal launcher interface between api.hip and the kernel translation units.
#pragma once
#include "common.h"
#include "func_eval.h"

namespace adgs {

struct PreprocessArgs {
	int P, D, M, D_S;
	const float* means3D; const float* scales; float scale_modifier; const float* rotations;
	const float* opacities; const float* shs; const float* cov3D_precomp; const float* colors_precomp;
	const float* flow_points; const float* semantic;
	const float* view; const float* proj; const float* campos;
	int W, H, gx, gy;
	float tan_fovx, tan_fovy, focal_x, focal_y;
	int inv_depth;
	// outputs
	int* radii; Splat* splats; float* cov3D; uint8_t* clamped; uint32_t* tiles_touched;
	// v2 (coarse-binned) extras; v2 == 0 selects the classic behaviour
	int v2;
	uint32_t* fine_touched;          // #fine tiles covered per Gaussian (its sum bounds the chunk pool)
	uint4* dupinfo;                  // (shrunk rect min, max, depth bits, -) per Gaussian: the binning kernel's only input
	int cell_tiles, cgx, cgy;        // coarse cell = cell_tiles x cell_tiles fine tiles
	ShSource sh_src;                 // raw SH source (sh_src.scene_dc != nullptr) instead of `shs`
	const float* sh0;                // raw SH source: precomputed coefficient 0 [P,3] (launch_sh0)
	float* gacc;                     // v2: [P][GACC_STRIDE] accumulator lines, zeroed here for every visible Gaussian (nullptr: skip)
	unsigned long long* fine_total;  // v2: reset here; the scan pass adds up fine_touched into it
	// v2 bucket binning (binning.hip; bucket_count == nullptr: sort-based binning): every visible Gaussian is counted into the
	// coarse cells it covers -- bucket_count[workgroup][cell], every entry written -- and fine_total is accumulated here (the frame's
	// prologue zeroed it)
	uint32_t* bucket_count;
	// the frame's configuration word, written into the header of the image state (api.hip: frame_cfg_word): a backward that cannot find
	// its forward in the host-side frame table (cloned / offloaded state buffers) reads it back instead of consulting the environment
	uint32_t* cfg_word; uint32_t cfg_value;
	// raw-SH path, 16 coefficients: d(colour channel c)/d(view direction) of every visible Gaussian, [9][P] (dx / dy / dz of channels 0..2 in
	// planes 0..8) -- 36 bytes that spare the preprocess BACKWARD its second pass over the 180-byte `rest` rows (nullptr: not wanted)
	float* ddir;
};

int launch_preprocess_fwd(const PreprocessArgs& a, hipStream_t stream);
int launch_mark_visible(int P, const float* means, const float* view, uint8_t* present, hipStream_t stream);
int launch_duplicate_keys(int P, const Splat* splats, const uint32_t* offsets, const int* radii, uint64_t* keys, uint32_t* vals,
	int gx, int gy, hipStream_t stream);
int launch_tile_ranges(int L, const uint32_t* d_L, const uint64_t* keys, uint2* ranges, uint32_t id_mask, hipStream_t stream);

struct RenderFwdArgs {
	const uint2* ranges; const uint32_t* point_list; const Splat* splats;
	int W, H, gx, gy, D_S;
	bool has_color, has_flow, has_sem, inv_depth;
	const float* semantic;           // [P, D_S] (only read when D_S > 1)
	const float* bg;
	float* final_T;                  // img_opacity output (holds 1 - T)
	uint32_t* n_contrib;
	float* out_color; float* out_depth; float* out_flow; float* out_semantic;
	int order_mode;
};
int launch_render_fwd(const RenderFwdArgs& a, hipStream_t stream);

struct RenderBwdArgs {
	const uint2* ranges; const uint32_t* point_list; const Splat* splats;
	int W, H, gx, gy, D_S;
	const float* semantic; const float* bg;
	const float* final_T; const uint32_t* n_contrib;
	const float* dL_dpix; const float* dL_dpix_depth; const float* dL_dpix_flow; const float* dL_dpix_sem; const float* dL_dpix_opacity;
	bool do_color, do_flow, do_sem, do_depth, do_opacity;
	float* dL_dmean2D;   // [P,3]
	float* dL_dconic;    // [P,4] (x, y, -, w)
	float* dL_dopacity;  // [P]
	float* dL_dcolor;    // [P,3]
	float* dL_ddepth;    // [P]
	float* dL_dflow;     // [P,3]
	float* dL_dsem;      // [P,D_S]
};
int launch_render_bwd(const RenderBwdArgs& a, hipStream_t stream);

struct PreprocessBwdArgs {
	int P, D, M;
	const float* means3D; const int* radii; const float* shs; const uint8_t* clamped;
	const float* scales; const float* rotations; float scale_modifier;
	const float* cov3D;              // precomputed or the forward's
	const float* view; const float* proj; const float* campos;
	float focal_x, focal_y, tan_fovx, tan_fovy;
	int inv_depth;
	const float* dL_dmean2D; const float* dL_dconic; const float* dL_dcolor; const float* dL_ddepth;
	float* dL_dmean3D; float* dL_dcov3D; float* dL_dsh; float* dL_dscale; float* dL_drot;
	// v2: per-Gaussian sums arrive packed in gacc ([P][16]); they are unpacked into the ABI outputs
	// below (which are then written, not read).  gacc == nullptr selects the classic inputs above.
	float* gacc; const Splat* splats; int W, H;
	ShSource sh_src; ShGradDst sh_dst;   // raw-SH path: gradients go straight to the raw tensors' layout
	float* out_mean2D; float* out_conic; float* out_opacity; float* out_color; float* out_depth; float* out_flow; float* out_sem;
	int D_S;
	int sh_staging;                  // 0: ADGS_NO_SH_STAGING was set when the FORWARD of this frame ran (api.hip: FrameCfg)
	const float* ddir;               // [9][P] written by this frame's preprocess forward (PreprocessArgs.ddir) or nullptr: read the `rest` rows again
};
int launch_preprocess_bwd(const PreprocessBwdArgs& a, hipStream_t stream);

// ---- v2 pipeline (render_v2.hip) ----
constexpr int CHUNK_WORDS = 2 + WAVE;   // [64 Gaussian ids, prev chunk, count]: the ids first, so that a wave reads `c[lane]` and the link words with two
constexpr int CHUNK_PREV = WAVE, CHUNK_COUNT = WAVE + 1;      // independent vector loads (the backward prefetches the next chunk of its list)
// Chunk slots are handed out in blocks of POOL_BLOCK: tile t owns slots [t * POOL_BLOCK, (t + 1) * POOL_BLOCK) outright and draws further
// blocks from the shared cursor (which counts from #tiles * POOL_BLOCK).  One returning atomic per BATCH on one address -- 28 800 per
// C3 frame -- was what the forward blend kernel's time consisted of: the serialised atomics back up the CUs' memory pipelines, and
// every load of every wave queues behind them (the kernel took 375 us whatever its arithmetic, occupancy or locality were changed to).
constexpr int POOL_BLOCK = 4;
constexpr int GACC_STRIDE = 16;         // one 64-byte line of gradient accumulators per Gaussian
constexpr int GACC_USED = 14;           // S0 Sx Sy Sxx Sxy Syy c0 c1 c2 d f0 f1 f2 s0

// ---- bucket binning (binning.hip) ----
constexpr int MAX_CELLS = 1024;         // coarse cells one cell_scan workgroup (and one LDS histogram) handles
constexpr int SLAB_ROW = 128;           // words per cell in a table of depth-slab bounds: the 127 inner 128-quantiles of the cell's depth keys (+ 1 pad)
constexpr int MAX_SLAB_LG = 7;          // a cell's list is built as 2^lg <= 128 independently sorted depth slabs; lg per cell, from the cell's pair count
constexpr int SLAB_TARGET = 3072;       // ... the smallest lg that brings the cell's pairs per slab to this or below (3/4 of what one sort holds: with the camera's
                                        // own bounds the slabs of a cell are equal to within its depth ties; 2048 made 1120 workgroups of C3's 70 cells: two rounds)
constexpr int SLAB_TARGET_FOREIGN = 2048;      // ... when the bounds are not the camera's own (sampled by cell_sample; another render's on small images): room for a 2 x misfit
constexpr int GS_NMAX = 4096;           // entries one slab_sort workgroup sorts at a time inside its CU
constexpr int MAX_CHUNKS = 16384;       // a frame of more than MAX_CHUNKS x GS_NMAX pairs takes the device-wide sort
// pinned host mailbox the device publishes the frame totals to (api.hip: the host polls `seq`)
// overflow: the totals exceed the capacity the frame's launches were enqueued against; overflow_count: such frames since the mailbox exists
struct Mailbox { volatile uint32_t seq; uint32_t r_cells; unsigned long long r_fine; uint32_t oversize, n_groups, overflow, overflow_count, max_cell_chunks;      // max_cell_chunks: bucket binning, the fullest slab in units of GS_NMAX entries
	uint32_t cap_cells; unsigned long long cap_fine; };      // the capacity THIS frame was enqueued against, as its kernels saw it (a graph replay reports its capture's)
int launch_bin_prepare(const FramePrologue& p, hipStream_t stream);
struct CellScanArgs {
	const uint32_t* cell_count;         // pairs per cell (cell_colscan)
	uint32_t* cell_start;               // [ncells + 1]
	uint2* cell_ranges; int ncells;
	uint32_t max_chunks;                // more GS_NMAX-entry units than this: d_counts[2] (the host re-bins with the device-wide sort)
	uint2* cell_work;                   // [ncells + 1] <- (first slab_sort workgroup of the cell, its lg); [ncells] = (workgroups in all, 0)
	int force_lg;                       // >= 0: every cell gets 2^force_lg slabs (ADGS_SLABS_LG: tests)
	uint32_t slab_target;               // the smallest lg that brings a cell's pairs per slab to this or below
	uint32_t* d_counts;                 // [0] pairs, [1] slab_sort workgroups, [2] more than max_chunks x GS_NMAX pairs, [3] the totals exceed the capacity,
	                                    // [4] slab_sort workgroups done, [5] the fullest slab (GS_NMAX units), [6..7] fine-tile total
	const unsigned long long* fine_total;
	uint32_t cap_cells; unsigned long long cap_fine;      // capacities of the launches that follow
	Mailbox* box; uint32_t seq;         // non-null: publish the totals here (a host that waits for them before it enqueues the binning)
};
int launch_cell_scan(const CellScanArgs& a, hipStream_t stream);
int launch_cell_colscan(uint32_t* counts, int nblocks, int ncells, uint32_t* cell_count, hipStream_t stream);
// rec_key[pos] = depth bits, rec_im[pos] = (Gaussian id, rectangle mask): a cell's unsorted records, struct-of-arrays (slab_sort streams the keys alone)
int launch_cell_scatter(int P, const uint4* dupinfo, const uint32_t* cell_start, const uint32_t* counts, uint32_t* rec_key, uint2* rec_im, uint32_t cap,
	int cell_tiles, int cgx, int ncells, uint32_t* pool_cursor, uint32_t* slow_list, hipStream_t stream);      // slow_list[0] <- 0
struct SlabSortArgs {
	const uint2* cell_ranges; int ncells;
	const uint2* cell_work;             // (first workgroup, lg) per cell (cell_scan): 2^lg depth slabs, one workgroup each
	uint32_t grid;                      // workgroups launched (>= d_counts[1] whenever the frame fits its capacity)
	const uint32_t* rec_key; const uint2* rec_im;  // the cells' unsorted records (cell_scatter)
	uint2* ent_f;                       // final (id, mask) entries
	uint32_t cap;                       // records the binning buffer holds
	const uint32_t* bounds;             // this frame's snapshot of the slab bounds [ncells][SLAB_ROW]
	uint32_t* bounds_out;               // the thread's table [MAX_CELLS][SLAB_ROW]: this frame's 128-quantiles per cell, for the next frame (nullptr: none)
	uint32_t* bounds_out2;              // ... and the camera's own table (api.hip: OrderHints), for its next render (nullptr: none)
	uint32_t* slow_list;                // [1 + grid] slabs slab_sort hands to slab_sort_slow, (cell << 8) | slab: [0] = count (zeroed by cell_scatter)
	uint32_t* d_counts;
	Mailbox* box;                       // the last workgroup to finish reports the fullest slab (max_cell_chunks)
};
int launch_slab_sort(const SlabSortArgs& a, hipStream_t stream);
// this frame's bounds [ncells][SLAB_ROW] from a sample of every cell's keys (frames without bounds of their camera's previous render)
int launch_cell_sample(const SlabSortArgs& a, uint32_t* bounds, hipStream_t stream);
// sorted frames (device-wide radix sort): the 128-quantiles of every cell from the sorted (cell | depth) keys
int launch_bounds_from_sorted(const uint64_t* keys, const uint2* cell_ranges, int ncells, const uint32_t* d_total, uint32_t cap, uint32_t* bounds_out, hipStream_t stream);

int launch_duplicate_cells(int P, const uint4* dupinfo, const uint32_t* offsets, uint64_t* keys, uint32_t* vals,
	uint32_t cap, int cell_tiles, int cgx, uint2* cell_ranges, int ncells, uint32_t* pool_cursor, int mask_shift, hipStream_t stream);

struct RenderV2FwdArgs {
	const uint2* cell_ranges; const uint32_t* cell_list; const Splat* splats;
	const uint64_t* cell_keys; int mask_shift;   // sorted keys with the rectangle-coverage masks at bit mask_shift (sorted frames; api.hip picks a cell size whose masks fit)
	const uint2* cell_entries;                   // bucket binning: (id, mask) per list entry (then cell_keys / cell_list == nullptr)
	int W, H, gx, gy, cell_tiles, cgx;      // gy: rows of WAVE tiles (16 x 4*ppl pixels), not of 16x16 tiles
	int ppl;                                // pixels per lane: 4 or 2 (v2_pixels_per_lane)
	bool has_color, has_flow, has_sem;
	const float* bg;
	const float* bg_image;                  // [3,H,W] per-pixel background (environment map) or nullptr: color = C + T * bg_image instead of + T * bg
	uint32_t* pool; uint32_t* pool_cursor; uint32_t* tile_last_chunk; uint32_t* tile_consumed; uint32_t* tile_scanned; uint32_t* tile_batches;
	float* final_T; uint32_t* n_contrib;
	float* out_color; float* out_depth; float* out_flow; float* out_semantic;
	int order_mode;                         // 1: workgroups walk the tiles bottom-up, 0: top-down, 2: in the order `fwd_order` gives
	const uint32_t* fwd_order;              // order_mode 2: workgroup -> tile, the longest-first order of THIS CAMERA'S PREVIOUS render (api.hip: OrderHints)
	const float* fwd_view; const float* fwd_sig;      // this frame's view matrix / the one the hint was made under (16 floats each): a hint of another pose is ignored
	const uint32_t* overflow_flag;          // device word: != 0 = the frame does not fit the capacity of this launch (blend nothing)
	bool publish;                           // true: the training forward; false: nothing is kept for a backward (pool / tile_* / n_contrib are not touched, may be NULL)
};
int launch_render_fwd_v2(const RenderV2FwdArgs& a, hipStream_t stream);

struct RenderV2BwdArgs {
	const Splat* splats; const uint32_t* pool; const uint32_t* tile_last_chunk; const uint32_t* tile_consumed;
	int W, H, gx, gy, ppl;                  // gy: rows of WAVE tiles, as in the forward
	const float* bg; const float* final_T; const uint32_t* n_contrib;
	const float* bg_image; float* dL_dbg_image;      // per-pixel background of the forward and its gradient T_final * dL/dC ([3,H,W], every pixel written)
	const float* dL_dpix; const float* dL_dpix_depth; const float* dL_dpix_flow; const float* dL_dpix_sem; const float* dL_dpix_opacity;
	bool do_color, do_flow, do_sem, do_depth, do_opacity;
	float* gacc;                     // [P][GACC_STRIDE], zero-initialised
	const uint32_t* tile_order;      // workgroup -> tile, longest lists first (launch_tile_order); nullptr: identity
	uint32_t* tl_start; uint32_t* tl_end;      // -DADGS_TIMELINE experiment build: per-tile wave start / end (100 MHz ticks); else unused
	// extra semantic channels (D_S > 1): one more replay per channel c >= 1 with only do_sem set -- the entry's value is read from
	// sem_src[id * sem_stride] instead of the Splat, its gradient sum goes to sem_dst[id * sem_stride] instead of the gacc line; the
	// geometric sums add into the gacc line as in the main pass (dL/dalpha is linear in the channels).  nullptr: the main pass.
	const float* sem_src; float* sem_dst; int sem_stride;
};
int launch_render_bwd_v2(const RenderV2BwdArgs& a, hipStream_t stream);

// Forward of the semantic channels 1 .. D_S-1 (channel 0 rides in the Splat and is blended by render_fwd_v2): a back-to-front
// replay of the chunk lists the main forward published, S = s a + (1 - a) S per contributing entry (the same sum as forward.cu:372
// written as a Horner recurrence), up to 4 channels per launch.
struct RenderV2SemFwdArgs {
	const Splat* splats; const uint32_t* pool; const uint32_t* tile_last_chunk; const uint32_t* tile_consumed; const uint32_t* n_contrib;
	int W, H, gx, gy, ppl;
	const float* semantic; int D_S, c0, nch;      // channels c0 .. c0 + nch - 1 of semantic[P, D_S]
	float* out_semantic;                          // [D_S, H, W]
};
int launch_render_sem_fwd_v2(const RenderV2SemFwdArgs& a, hipStream_t stream);
// order[i] = tile with the i-th largest number of consumed entries (bucketed): the backward starts the long tiles first
// order_copy: the same permutation a second time (the camera's hint buffer), followed by the 16 floats of `view` (the pose the hint belongs to)
int launch_tile_order(int ntiles, const uint32_t* tile_consumed, uint32_t* order, hipStream_t stream, uint32_t* order_copy = nullptr, const float* view = nullptr);

// flat coalesced d/dparam[m, d, k] = w_k * g[m * gstride + d] for the linear families (deform.hip)
int launch_lin_param_grad(int count, int D, const float* g, int gstride, float* out, const adgs_func_eval& f, hipStream_t stream);
// adam_a / adam_b (p != nullptr): the Adam step on that side's parameter rows in place of the store to out_a / out_b
int launch_lin_param_grad2(int count_a, const float* g_a, float* out_a, int count_b, const float* g_b, float* out_b, int D, int gstride,
	const adgs_func_eval& f, hipStream_t stream, const AdamSlot* adam_a = nullptr, const AdamSlot* adam_b = nullptr, float beta1 = 0.f, float beta2 = 0.f,
	float eps = 0.f);
// lr / (1 - beta1^t) and 1 / sqrt(1 - beta2^t), formed in double and rounded once (optim.hip)
void adam_bias_terms(float lr, int step, float beta1, float beta2, float* step_size, float* inv_bc2_sqrt);
// sh0[N,3] = dc + f_shs(t); optionally runs the frame's prologue (counters zeroed, slab bounds snapshot) for the binning kernels behind it
int launch_sh0(int N, const ShSource& s, float* out, hipStream_t stream, const FramePrologue* prologue = nullptr, int ostride = 3);      // ostride: floats per Gaussian in `out`
inline bool has_lin_host(const adgs_func_eval& f) { return (f.n_terms[0] + f.n_terms[1] + f.n_terms[2]) > 0 && f.n_params > 0; }

int knn_run(int P, const float* points, float* meanDists, char* workspace, hipStream_t stream);
size_t knn_workspace_bytes(int P);

} // namespace adgs
