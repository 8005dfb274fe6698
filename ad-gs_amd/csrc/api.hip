// C ABI of libadgs_hip.so (declared in include/adgs_rasterizer.h): host-side
// orchestration of the forward / backward kernel sequences.
//
// Replaces CudaRasterizer::Rasterizer::{forward,backward,markVisible}
// (RAST/cuda_rasterizer/rasterizer_impl.cu:141-153,198-352,356-476).
#include "common.h"
#include "kernels.h"
#include "../../include/adgs_rasterizer.h"
#include "../../include/adgs_optim.h"

#include <atomic>
#include <chrono>
#include <mutex>
#include <vector>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>

namespace adgs {

static thread_local std::string g_last_error;
// What a forward remembers for its caller and for the next forward -- per (host thread, device), like the mailbox: two threads (or one thread
// on two GPUs) rendering at the same time do not see each other's statistics or capacity hints.  Only the forward touches it (autograd runs the
// backward on a thread of its own: the backward keys what it needs by the state buffers, remember_frame / lookup_frame).  The stage profiler
// (adgs_profile_*) stays process-wide on purpose: it collects the events of both directions.
struct FrameContext {
	adgs_frame_stats stats;
	size_t hint_cells, hint_fine;          // speculative binning capacities (v2 forward): previous frames' counts + 25 %
	unsigned hint_max_cell_chunks;         // the fullest slab (GS_NMAX units) of the last bucket-binned frames, decaying
	bool bucket_frame_pending;             // the last forward was bucket-binned: its fullest slab is (or will be) in the mailbox
	long long reruns;                      // forwards whose capacity was too small (binning + blend enqueued twice)
	int last_order_hint;                   // the last forward was handed a tile order an earlier forward of the same camera and stream left (OrderHints)
	// Depth-slab bounds of the bucket binning (binning.hip): [MAX_CELLS][SLAB_ROW] device words, the 128-quantiles of every cell's depth keys
	// as the most recent bucket-binned frame of this thread left them.  Every frame takes its own snapshot in its first kernel and bins
	// by that; the sort kernels of the frame rewrite the table for the next one.  ANY contents are valid bounds (a torn snapshot beside
	// another stream's writes, another camera's or another image shape's quantiles: the slab function stays monotone in the depth) --
	// they only decide how evenly a cell's entries spread over its slabs.
	uint32_t* slab_bounds;
	hipStream_t service;                   // private non-blocking stream: one-off initialisations that must not become part of a capture
};
static FrameContext* frame_context() {
	constexpr int MAX_DEV = 64;
	static thread_local FrameContext ctx[MAX_DEV] = {};
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) dev = 0;
	return &ctx[dev];
}
void set_error(const std::string& msg) { g_last_error = msg; }

// rasterizer_impl.cu:35-50 (next-highest bit of the MSB)
static uint32_t higher_msb(uint32_t n) {
	uint32_t msb = sizeof(n) * 4;
	uint32_t step = msb;
	while (step > 1) {
		step /= 2;
		if (n >> msb) msb += step; else msb -= step;
	}
	if (n >> msb) msb++;
	return msb;
}

// ---- state carved from the three caller-owned byte buffers (opaque to the caller) ----
struct GeomState {
	Splat* splats; float* cov3D; uint8_t* clamped; uint32_t* tiles_touched; uint32_t* offsets; char* scan_temp;
	static GeomState carve(char* chunk, size_t P, size_t* bytes) {
		Carver c(chunk); GeomState g;
		g.splats = c.take<Splat>(P);
		g.cov3D = c.take<float>(P * 6);
		g.clamped = c.take<uint8_t>(P);
		g.tiles_touched = c.take<uint32_t>(P + 1);
		g.offsets = c.take<uint32_t>(P + 1);
		g.scan_temp = c.take<char>(scan_temp_bytes(P + 1));
		if (bytes) *bytes = c.size();
		return g;
	}
};
// Both image states start with the same 64-byte header: word 0 is the frame's configuration (frame_cfg_word), written by the forward's
// preprocess kernel -- whatever the layout behind it, a backward can read how its forward carved it.
constexpr size_t IMG_HEADER_WORDS = 16;
struct ImgState {
	uint32_t* header; uint32_t* n_contrib; uint2* ranges;
	static ImgState carve(char* chunk, size_t npix, size_t ntiles, size_t* bytes) {
		Carver c(chunk); ImgState s;
		s.header = c.take<uint32_t>(IMG_HEADER_WORDS);
		s.n_contrib = c.take<uint32_t>(npix);
		s.ranges = c.take<uint2>(ntiles);
		if (bytes) *bytes = c.size();
		return s;
	}
};
struct BinState {
	uint64_t* keys_unsorted; uint64_t* keys; uint32_t* list_unsorted; uint32_t* list; char* sort_temp;
	static BinState carve(char* chunk, size_t R, size_t* bytes) {
		Carver c(chunk); BinState b;
		b.list = c.take<uint32_t>(R);
		b.list_unsorted = c.take<uint32_t>(R);
		b.keys = c.take<uint64_t>(R);
		b.keys_unsorted = c.take<uint64_t>(R);
		b.sort_temp = c.take<char>(sort_temp_bytes(R));
		if (bytes) *bytes = c.size();
		return b;
	}
};

// ---- v2 (coarse-binned) state ----
struct GeomStateV2 {
	Splat* splats; uint8_t* clamped; uint32_t* cells_touched; uint32_t* offsets; uint32_t* fine_touched; uint4* dupinfo;
	float* gacc; float* sh0; char* scan_temp; unsigned long long* fine_total;
	float* ddir;      // [9][P]: d colour / d view direction of the raw-SH path (PreprocessArgs.ddir)
	// bucket binning (binning.hip): ONE block of counters the frame's prologue zeroes -- pair counts per cell, the fine-tile total slots,
	// the device-side (pairs, overflow, ...) words -- then the cell starts, the frame's snapshot of the slab bounds and the counts matrix
	uint32_t* counters; uint32_t* cell_start; uint2* cell_work; uint32_t* bounds; uint32_t* counts;      // counts[ceil(P / 256)][ncells]: pairs per (preprocess workgroup, cell); bounds[ncells][SLAB_ROW]; cell_work[ncells + 1]
	static constexpr size_t COUNTER_WORDS = MAX_CELLS + 2 * SCAN_AUX_SLOTS + 8;
	uint32_t* cell_count() const { return counters; }
	unsigned long long* bucket_fine_total() const { return reinterpret_cast<unsigned long long*>(counters + MAX_CELLS); }
	uint32_t* d_counts() const { return counters + MAX_CELLS + 2 * SCAN_AUX_SLOTS; }
	// with_ddir: the frame hands d colour / d direction from its forward to its backward (raw-SH path, degree 3, a training frame) -- both
	// sides derive the flag from the same arguments; ncells: forward only
	static GeomStateV2 carve(char* chunk, size_t P, size_t* bytes, bool with_ddir, size_t ncells = 0) {
		Carver c(chunk); GeomStateV2 g;
		g.counters = c.take<uint32_t>(COUNTER_WORDS);
		g.cell_start = c.take<uint32_t>(MAX_CELLS + 1);
		g.cell_work = c.take<uint2>(MAX_CELLS + 1);
		g.splats = c.take<Splat>(P);
		g.gacc = c.take<float>(P * GACC_STRIDE);
		g.dupinfo = c.take<uint4>(P);
		g.clamped = c.take<uint8_t>(P);
		g.cells_touched = c.take<uint32_t>(P + 1);
		g.offsets = c.take<uint32_t>(P + 1);
		g.fine_touched = c.take<uint32_t>(P + 1);
		g.sh0 = c.take<float>(P * 3);
		g.fine_total = c.take<unsigned long long>(SCAN_AUX_SLOTS);
		g.scan_temp = c.take<char>(scan_temp_bytes(P + 1));
		g.ddir = with_ddir ? c.take<float>(P * 9) : nullptr;
		const size_t nc = std::min<size_t>(ncells, (size_t)MAX_CELLS + 1);      // last: the backward carves without them
		g.bounds = c.take<uint32_t>(nc * SLAB_ROW);
		g.counts = c.take<uint32_t>(((P + 255) / 256) * nc);
		if (bytes) *bytes = c.size();
		return g;
	}
};
struct ImgStateV2 {
	uint32_t* header; uint32_t* n_contrib; uint2* cell_ranges; uint32_t* tile_last_chunk; uint32_t* tile_consumed; uint32_t* tile_order; uint32_t* tile_scanned; uint32_t* tile_batches;
	static ImgStateV2 carve(char* chunk, size_t npix, size_t ntiles, size_t ncells, size_t* bytes) {
		Carver c(chunk); ImgStateV2 s;
		s.header = c.take<uint32_t>(IMG_HEADER_WORDS);
		s.n_contrib = c.take<uint32_t>(npix);
		s.cell_ranges = c.take<uint2>(ncells);
		s.tile_last_chunk = c.take<uint32_t>(ntiles);
		s.tile_consumed = c.take<uint32_t>(ntiles);
		s.tile_order = c.take<uint32_t>(ntiles);
		s.tile_scanned = c.take<uint32_t>(ntiles);          // candidates of the cell list the tile's walk went through (statistics)
		s.tile_batches = c.take<uint32_t>(ntiles);          // batches of <= 64 entries handed to the blend loop (statistics)
		if (bytes) *bytes = c.size();
		return s;
	}
};
struct BinStateV2 {
	uint32_t* pool_cursor; uint32_t* pool;     // first, so that the backward finds them without knowing the sizes
	uint64_t* keys_unsorted; uint64_t* keys; uint32_t* list_unsorted; uint32_t* list; char* sort_temp;
	// bucket binning: unsorted (cell-grouped) records as two arrays -- depth keys; (id, mask) --, final (id, mask) entries
	uint32_t* rec_key; uint2* rec_im; uint2* entries; uint32_t* slow_list;
	// slab_sort workgroups a frame of R_cells pairs needs at most: a cell of n pairs has 2^lg <= max(1, 2 n / SLAB_TARGET) slabs
	static size_t slab_grid(size_t R_cells, size_t ncells) { return std::max<size_t>(2 * R_cells / SLAB_TARGET_FOREIGN + ncells + 1, ncells << forced_lg()); }
	static int forced_lg() { const char* v = getenv("ADGS_SLABS_LG"); return (v && *v) ? std::min(std::max(atoi(v), 0), (int)MAX_SLAB_LG) : 0; }      // test hook: every cell gets 2^lg slabs (read by the FORWARD only)
	// every tile's own block + blocks drawn from the cursor, each of which may end partly used
	static size_t pool_chunks(size_t R_fine, size_t ntiles) { return R_fine / WAVE + (2 * (size_t)POOL_BLOCK + 1) * ntiles + 1; }
	static BinStateV2 carve_buckets(char* chunk, size_t R_cells, size_t R_fine, size_t ntiles, size_t* bytes, size_t ncells = 0) {
		Carver c(chunk); BinStateV2 b;
		b.pool_cursor = c.take<uint32_t>(64);
		b.pool = c.take<uint32_t>(pool_chunks(R_fine, ntiles) * CHUNK_WORDS);
		b.entries = c.take<uint2>(R_cells + 1);      // (+ 1: the blend forward's staged reads run one entry past a list, render_v2.hip)
		b.rec_im = c.take<uint2>(R_cells);
		b.rec_key = c.take<uint32_t>(R_cells);
		b.slow_list = c.take<uint32_t>(1 + slab_grid(R_cells, ncells));
		b.keys = nullptr; b.keys_unsorted = nullptr; b.sort_temp = nullptr; b.list = nullptr; b.list_unsorted = nullptr;
		if (bytes) *bytes = c.size();
		return b;
	}
	static BinStateV2 carve(char* chunk, size_t R_cells, size_t R_fine, size_t ntiles, size_t* bytes) {
		Carver c(chunk); BinStateV2 b;
		b.pool_cursor = c.take<uint32_t>(64);
		b.pool = c.take<uint32_t>(pool_chunks(R_fine, ntiles) * CHUNK_WORDS);
		b.list = c.take<uint32_t>(R_cells);
		b.list_unsorted = c.take<uint32_t>(R_cells);
		b.keys = c.take<uint64_t>(R_cells);
		b.keys_unsorted = c.take<uint64_t>(R_cells);
		b.sort_temp = c.take<char>(sort_temp_bytes(R_cells));
		b.rec_key = nullptr; b.rec_im = nullptr; b.entries = nullptr; b.slow_list = nullptr;
		if (bytes) *bytes = c.size();
		return b;
	}
};

// Every environment switch of the rasterizer is read through these two (and counted: adgs_test_env_reads -- the backward of a frame
// must not read any: what the forward decided from the environment travels in the frame table, FrameCfg).
static std::atomic<unsigned long long> g_env_reads{0};
static const char* env_str(const char* name) { g_env_reads.fetch_add(1, std::memory_order_relaxed); return getenv(name); }
static int env_int(const char* name, int dflt) {
	const char* v = env_str(name);
	return (v && *v) ? atoi(v) : dflt;
}
// "classic" reproduces the reference pipeline stage by stage (full (tile|depth) sort, num_rendered
// equal to the reference's); "v2" (default) is the coarse-binned lazy pipeline.  v2 blends semantic channel 0 in the main kernels
// and replays the published lists once more per extra channel (render_sem_fwd_v2 / render_bwd_v2 with sem_src; AD-GS uses 1 channel.
// C3 with 4 / 32 channels: 2.6 / 14.6 ms per frame against 23.8 / 46.5 ms on the classic kernels).  ADGS_V2_MAX_SEMANTIC (default 32
// = all) sends larger D_S to the classic kernels.
// env = false: the built-in defaults (what a backward over foreign state buffers assumes; it never reads the environment).
static bool use_v2(int D_S, bool env = true) {
	if (!env) return D_S <= (int)MAX_SEMANTIC;
	const char* m = env_str("ADGS_RASTER_MODE");
	if (m && std::string(m) == "classic") return false;
	return D_S <= std::min(env_int("ADGS_V2_MAX_SEMANTIC", 32), (int)MAX_SEMANTIC);
}

// Pixels per lane of the v2 blend kernels: one wave per 16x16 tile (4 px/lane) when that already gives the chip enough
// waves; 16x8 half tiles (2 px/lane) for small images (KITTI-sized frames have < 2 waves per SIMD otherwise).
// ADGS_V2_PPL=2|4 overrides; forward and backward of a frame must see the same value.
// Edge of a coarse cell in 16x16 tiles.  Larger cells mean fewer (cell, Gaussian) pairs to sort but longer candidate lists for
// every tile to filter; measured on MI355X (frames/s, cell edge 8 / 10 / 12 tiles): C3 1920x1280 (9600 tiles) 636 / 640 / 643,
// C5 395 / 400 / 405, C2 1242x375 (1872 tiles) 1324 / 1279 / 1152 -- so 12 for large tile grids (round 1, device-wide sort).
// Re-measured in round 3 with bucket binning (the cells' lists are sorted inside the CUs: more, smaller cells cost little): C2 cell edge
// 8 / 6 / 5 / 4 / 3: binning 73 / 66 / 60 / 56 / 51 us, forward 138 / 131 / 128 / 126 / 127 us; C1 as graph replays 10 380 / 10 790 / - / 10 980
// frames/s; C3 12 / 10 / 8: 857 / 860 / 826 frames/s.  Small tile grids therefore aim at ~110 cells.
// Coarse-cell edge in tiles: 12 for large tile grids, ~110 cells per image for small ones (EXPERIMENTS.md), ADGS_CELL_TILES overrides.
// The cell lists carry one mask bit per tile row and per tile column of a cell (the blend forward's rectangle test); in a sorted
// (not bucket-binned) frame these 2 x edge bits share the upper key word with the cell index: the edge shrinks until they fit.
static bool v2_keys_fit(int gx, int gy, int c) { return 2 * c + (int)higher_msb((uint32_t)(((gx + c - 1) / c) * ((gy + c - 1) / c))) <= 32; }
// fit_keys: shrink the edge until the mask bits fit the sort key -- only a frame that takes the device-wide SORT needs that (bucket
// binning carries separate (id, mask) entries); a bucket-binned frame keeps the edge it was asked for.
static int v2_cell_tiles(int gx, int gy, bool fit_keys, bool env = true) {
	const size_t ntiles16 = (size_t)gx * gy;
	const int small = std::min(8, std::max(3, (int)std::lround(std::sqrt((double)ntiles16 / 110.0))));
	const int dflt = ntiles16 >= 4096 ? 12 : small;
	int c = std::max(1, env ? env_int("ADGS_CELL_TILES", dflt) : dflt);
	while (fit_keys && c > 1 && !v2_keys_fit(gx, gy, c)) c--;
	return c;
}
static int v2_pixels_per_lane(size_t ntiles16, bool env = true) {
	const int e = env ? env_int("ADGS_V2_PPL", 0) : 0;
	if (e == 1 || e == 2 || e == 4) return e;
	return ntiles16 < 4096 ? 2 : 4;
}


// ---- optional per-stage timing with HIP events on the launch stream (bench.py); Stage / StageTimer are declared in common.h ----
static const char* const kStageNames[ST_COUNT] = { "preprocess_fwd", "scan", "duplicate_keys", "radix_sort", "tile_ranges",
	"render_fwd", "render_bwd", "preprocess_bwd", "deform_fwd", "deform_bwd", "grad_expand" };
struct ProfRec { int stage; hipEvent_t a, b; };
// process-wide: torch.autograd calls the backward from its own worker thread
static std::atomic<unsigned> g_prof_mask{0};          // bit i: stage i is timed
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;
static hipEvent_t prof_event() {
	std::lock_guard<std::mutex> lk(g_prof_mu);
	if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
	hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
}
StageTimer::StageTimer(int stage_, hipStream_t stream) : on((g_prof_mask.load(std::memory_order_relaxed) >> stage_) & 1u), stage(stage_), a(nullptr), b(nullptr), s(stream) {
	if (on) { a = prof_event(); b = prof_event(); (void)hipEventRecord(a, s); }
}
StageTimer::~StageTimer() {
	if (on) { (void)hipEventRecord(b, s); std::lock_guard<std::mutex> lk(g_prof_mu); g_prof_recs.push_back(ProfRec{ stage, a, b }); }
}

// pinned host word for the one device->host read-back of num_rendered
// (the reference does a blocking cudaMemcpy, rasterizer_impl.cu:288)
static uint32_t* pinned_word() {
	static thread_local uint32_t* p = nullptr;
	if (!p) { if (hipHostMalloc((void**)&p, 1024, hipHostMallocDefault) != hipSuccess) p = nullptr; }
	return p;
}

// Mailbox in pinned host memory, written by the GPU itself and polled by the host: the v2 forward needs two totals in the
// middle of the frame, and a driver-level wait (hipEventSynchronize / hipStreamSynchronize) for them was measured to fall
// back to a ~10 ms timeout per call in the first process on a freshly booted box (frames at 11 ms instead of 1.7 ms with
// every kernel at its normal duration).  A volatile read of host memory has no such mode.
struct MailboxRef { Mailbox* host; Mailbox* dev; uint32_t next_seq; size_t cap_cells, cap_fine;      // cap_*: capacities of the last host-side forward
	long long repaired; };      // eager frames whose overflow this library repaired itself (they bump the device-side overflow_count too)
// one mailbox per (host thread, device): the device pointer of a mapped allocation belongs to the device that was current when
// it was taken, so a thread that renders on several GPUs gets one per GPU (portable pinned memory)
static MailboxRef* mailbox() {
	constexpr int MAX_DEV = 64;
	static thread_local MailboxRef boxes[MAX_DEV] = {};
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
	MailboxRef& m = boxes[dev];
	if (!m.host) {
		void* h = nullptr; void* d = nullptr;
		if (hipHostMalloc(&h, 256, hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess) return nullptr;
		if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipHostFree(h); return nullptr; }
		memset(h, 0, 256);
		m.host = (Mailbox*)h; m.dev = (Mailbox*)d; m.next_seq = 1; m.cap_cells = m.cap_fine = 0;
	}
	return &m;
}
// Forward tile order from the camera's previous render (render_v2.hip: order_mode 2).  One hint buffer (a permutation of the wave tiles,
// written by launch_tile_order right behind every blend forward) per (camera, STREAM), per (host thread, device).  A camera is recognised by
// the device addresses of its view / projection matrices -- the reference's Camera keeps them on the GPU (scene/cameras.py:77-80);
// gaussian_renderer.render() keeps one contiguous copy per camera object -- together with the image shape.  A wrong guess (recycled
// addresses, a camera that moved) costs speed only: the kernel compares the pose the hint was made under with the frame's and ignores a
// stale one, and ANY permutation renders the same images.
// A hint is only ever read and rewritten by launches on the stream that produced it (round-5 advisor finding: a second forward of the same
// camera on another stream -- an evaluation stream, a graph replay beside eager frames -- could read the buffer while tile_order of the
// other stream rewrote it: the pose signature matches, the order is a mix of two permutations, tiles are rendered twice or never).  Work
// of one stream is ordered, so reader and writer of an entry can never overlap; another stream gets an entry of its own.  Entries created
// under stream capture belong to the graph being captured (`captured`): eager frames never touch them, whatever stream the graph is
// replayed on, and they stay allocated for the graph's lifetime.  (Two graphs of the SAME camera captured on the same stream share an
// entry: replaying them concurrently on different streams is the caller's race, like replaying one graph twice at once.)
// The table is bounded (OLDEST entry recycled: no hipFree / hipMalloc in steady state); under capture only existing entries are used.
struct OrderHints {
	struct Entry { const void* view; const void* proj; int W, H; size_t tiles; hipStream_t stream; bool captured; uint32_t* buf; unsigned long long used; bool written; size_t extra; };
	// buf: [tiles] the order | [16] the pose it was made under | [extra] the camera's own depth-slab bounds (binning.hip), ncells x SLAB_ROW words
	// written: a frame has been enqueued that fills buf (tile_order; slab_sort before it)
	std::vector<Entry> entries;
	unsigned long long clock = 0, lookups = 0, hits = 0;      // hits: forwards that found an entry (adgs_frame_status.order_hint: the last forward did)
	hipStream_t service = nullptr;                            // a private non-blocking stream: buffer initialisation that is never part of a capture
	static constexpr size_t MAX_ENTRIES = 1024;
	Entry* find(const void* view, const void* proj, int W, int H, size_t tiles, hipStream_t stream, bool captured) {
		for (Entry& e : entries)
			if (e.view == view && e.proj == proj && e.W == W && e.H == H && e.tiles == tiles && e.captured == captured && (captured || e.stream == stream)) { e.used = ++clock; return &e; }
		return nullptr;
	}
	// A fresh buffer carries an all-zero pose signature (no view matrix is all zero: the kernel ignores the hint until tile_order has written
	// one).  Zeroed on the service stream and waited for, so that it is neither captured nor ordered against the caller's stream by luck.
	bool reset_signature(uint32_t* buf, size_t tiles, size_t extra) {      // ... and the bounds rows start as "everything in slab 0"
		if (!service && hipStreamCreateWithFlags(&service, hipStreamNonBlocking) != hipSuccess) { service = nullptr; return false; }
		return hipMemsetAsync(buf + tiles, 0, 16 * sizeof(uint32_t), service) == hipSuccess &&
		       (extra == 0 || hipMemsetAsync(buf + tiles + 16, 0xff, extra * sizeof(uint32_t), service) == hipSuccess) && hipStreamSynchronize(service) == hipSuccess;
	}
	// capturing: the caller's stream is being captured into a graph (no synchronisation of it, no recycling of a live entry)
	Entry* create(const void* view, const void* proj, int W, int H, size_t tiles, hipStream_t stream, bool captured, size_t extra) {
		Entry* e = nullptr;
		if (entries.size() >= MAX_ENTRIES) {      // recycle the least recently used entry that no graph holds
			if (captured) return nullptr;
			for (Entry& c : entries) if (!c.captured && (!e || c.used < e->used)) e = &c;
			if (!e) return nullptr;
			// the old owner's stream may still have launches in flight that read or write the buffer: drain it before the buffer changes hands
			(void)hipStreamSynchronize(e->stream);
			if (e->tiles != tiles || e->extra != extra) { (void)hipFree(e->buf); e->buf = nullptr; e->tiles = 0; e->view = nullptr; if (hipMalloc((void**)&e->buf, (tiles + 16 + extra) * sizeof(uint32_t)) != hipSuccess) { e->buf = nullptr; return nullptr; } }
		} else {
			Entry fresh{ nullptr, nullptr, 0, 0, 0, nullptr, false, nullptr, 0, false, 0 };
			if (hipMalloc((void**)&fresh.buf, (tiles + 16 + extra) * sizeof(uint32_t)) != hipSuccess) return nullptr;      // the permutation + the 16 floats of the view matrix it was made under + the bounds
			entries.push_back(fresh);
			e = &entries.back();
		}
		e->view = view; e->proj = proj; e->W = W; e->H = H; e->tiles = tiles; e->stream = stream; e->captured = captured; e->used = ++clock; e->written = false; e->extra = extra;
		if (!reset_signature(e->buf, tiles, extra)) { e->view = nullptr; e->tiles = 0; (void)hipFree(e->buf); e->buf = nullptr; return nullptr; }      // (an unusable slot: find() never matches tiles == 0 / buf == nullptr is never handed out)
		return e;
	}
};
static OrderHints* order_hints() {
	constexpr int MAX_DEV = 64;
	static thread_local OrderHints hints[MAX_DEV];
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
	return &hints[dev];
}
// What the forward decided from the environment (pipeline, cell size, pixels per lane), remembered per forward state so that the
// backward of THAT forward carves the saved buffers the same way even if the environment changed in between.  The buffers are opaque
// device memory, so the key is host-side: the (image, geometry) buffer addresses autograd hands back TOGETHER WITH the frame's shape
// (W, H, P).  A state that reaches the backward under another address (cloned / offloaded saved tensors) or another shape is "not
// found": the backward then carves by today's environment and zeroes the accumulator lines itself -- it never trusts a stale entry.
// tile_order / sh_staging / timeline: the backward's own measurement knobs (ADGS_TILE_ORDER, ADGS_NO_SH_STAGING, ADGS_TIMELINE_BWD), read by the FORWARD
struct FrameCfg { int v2; int cell_tiles; int ppl; int tile_order = 1; int sh_staging = 1; int timeline = 0; int order_ready = 0; int backwards = 0; };      // order_ready: the forward already built img.tile_order      // backwards: how many backward passes have consumed this forward's accumulator lines
// FrameCfg <-> the configuration word in the image state's header (kernels.h: PreprocessArgs::cfg_word)
static uint32_t frame_cfg_word(const FrameCfg& c) {
	return 0xAD600000u | ((uint32_t)(c.v2 & 1) << 19) | ((uint32_t)(c.tile_order & 1) << 18) | ((uint32_t)(c.sh_staging & 1) << 17) | ((uint32_t)(c.timeline & 1) << 16) |
	       ((uint32_t)(c.order_ready & 1) << 15) |
	       ((uint32_t)(c.ppl & 0x7f) << 8) | (uint32_t)(c.cell_tiles & 0xff);
}
static bool frame_cfg_from_word(uint32_t w, FrameCfg* c) {
	if ((w & 0xFFF00000u) != 0xAD600000u) return false;
	*c = FrameCfg{ (int)((w >> 19) & 1u), (int)(w & 0xffu), (int)((w >> 8) & 0x7fu) };
	c->tile_order = (int)((w >> 18) & 1u); c->sh_staging = (int)((w >> 17) & 1u); c->timeline = (int)((w >> 16) & 1u); c->order_ready = (int)((w >> 15) & 1u);
	return c->cell_tiles >= 1 && (c->ppl == 1 || c->ppl == 2 || c->ppl == 4);
}
struct FrameKey {
	const void* img; const void* geom; int W, H, P;
	bool operator==(const FrameKey& o) const { return img == o.img && geom == o.geom && W == o.W && H == o.H && P == o.P; }
};
static std::mutex g_cfg_mu;
static std::vector<std::pair<FrameKey, FrameCfg>> g_cfg_table;
static void remember_frame(const FrameKey& k, const FrameCfg& c) {
	std::lock_guard<std::mutex> lk(g_cfg_mu);
	// a new forward into these addresses supersedes whatever lived there (either buffer may have been recycled for another shape)
	for (auto it = g_cfg_table.begin(); it != g_cfg_table.end();) it = (it->first.img == k.img || it->first.geom == k.geom) ? g_cfg_table.erase(it) : it + 1;
	if (g_cfg_table.size() >= 256) g_cfg_table.erase(g_cfg_table.begin());
	g_cfg_table.emplace_back(k, c);
}
// counts a backward over this forward state; returns how many ran before it, or -1 if the forward is not in the table (any more)
static int note_backward(const FrameKey& k) {
	std::lock_guard<std::mutex> lk(g_cfg_mu);
	for (auto it = g_cfg_table.rbegin(); it != g_cfg_table.rend(); ++it) if (it->first == k) return it->second.backwards++;
	return -1;
}
static bool lookup_frame(const FrameKey& k, FrameCfg* c) {
	std::lock_guard<std::mutex> lk(g_cfg_mu);
	for (auto it = g_cfg_table.rbegin(); it != g_cfg_table.rend(); ++it) if (it->first == k) { *c = it->second; return true; }
	return false;
}
// the test hooks only know the image buffer and the image size
static bool lookup_frame_by_image(const void* img, int W, int H, FrameCfg* c) {
	std::lock_guard<std::mutex> lk(g_cfg_mu);
	for (auto it = g_cfg_table.rbegin(); it != g_cfg_table.rend(); ++it)
		if (it->first.img == img && it->first.W == W && it->first.H == H) { *c = it->second; return true; }
	return false;
}
__global__ void publish_counts_kernel(const uint32_t* __restrict__ total_cells, const unsigned long long* __restrict__ fine_slots, Mailbox* box, uint32_t seq,
	uint32_t cap_cells, unsigned long long cap_fine, uint32_t* __restrict__ overflow_flag) {
	unsigned long long f = threadIdx.x < SCAN_AUX_SLOTS ? fine_slots[threadIdx.x] : 0ull;
#pragma unroll
	for (int off = WAVE / 2; off > 0; off >>= 1) f += __shfl_xor(f, off, WAVE);
	if (threadIdx.x == 0) {
		const uint32_t total = *total_cells;
		const uint32_t over = (total > cap_cells || f > cap_fine) ? 1u : 0u;      // the launches queued against the capacity must not blend
		*overflow_flag = over;
		box->r_cells = total; box->r_fine = f; box->oversize = 0u; box->n_groups = 0u; box->overflow = over; box->max_cell_chunks = 0u;
		if (over) box->overflow_count = box->overflow_count + 1u;
		box->cap_cells = cap_cells; box->cap_fine = cap_fine;
		__threadfence_system();
		box->seq = seq;                       // published last: the host spins on it
	}
}
// returns 0 when the GPU published `seq`; falls back to draining the stream if that takes implausibly long
static int wait_mailbox(const MailboxRef* m, uint32_t seq, hipStream_t stream) {
	const auto t0 = std::chrono::steady_clock::now();
	for (unsigned spin = 0;; spin++) {
		if (m->host->seq == seq) return 0;
		if ((spin & 0xFFFF) == 0xFFFF && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) break;
	}
	ADGS_HIP_CHECK(hipStreamSynchronize(stream));
	if (m->host->seq == seq) return 0;
	set_error("the device never published the binning counts"); return -1;
}

} // namespace adgs

using namespace adgs;

extern "C" const char* adgs_last_error(void) { return g_last_error.c_str(); }

extern "C" void adgs_get_frame_stats(adgs_frame_stats* out) { if (out) *out = frame_context()->stats; }

// Totals the device published for the calling thread's most recent v2 forward on the current device, and whether they fitted the
// capacity that forward (eager or captured in a HIP graph) was enqueued against.  Reads host memory only; meaningful once the
// stream the frame ran on has been synchronised (a graph replay publishes into the same mailbox as its capture did).
extern "C" int adgs_get_frame_status(adgs_frame_status* out) {
	if (!out) return -1;
	memset(out, 0, sizeof(*out));
	MailboxRef* mb = mailbox();
	if (!mb) { set_error("hipHostMalloc (mapped) failed"); return -1; }
	out->pairs = (int64_t)mb->host->r_cells; out->fine_pairs = (int64_t)mb->host->r_fine;
	// the capacity the frame's own kernels compared against (a graph replay: that of its capture; 0xffffffff / ~0: an eager frame that
	// waited for the exact counts)
	out->capacity_pairs = mb->host->cap_cells == 0xffffffffu ? (int64_t)mb->host->r_cells : (int64_t)mb->host->cap_cells;
	out->capacity_fine_pairs = mb->host->cap_fine == ~0ull ? (int64_t)mb->host->r_fine : (int64_t)mb->host->cap_fine;
	out->overflow = (int32_t)mb->host->overflow; out->overflow_count = (int64_t)mb->host->overflow_count;
	out->eager_reruns = (int64_t)frame_context()->reruns;
	out->order_hint = frame_context()->last_order_hint;
	out->fullest_slab_units = (int64_t)mb->host->max_cell_chunks;
	if (OrderHints* oh = order_hints()) { out->order_hint_lookups = (int64_t)oh->lookups; out->order_hint_hits = (int64_t)oh->hits; }
	out->unrepaired_overflow_count = (int64_t)mb->host->overflow_count - (int64_t)mb->repaired;
	return 0;
}

// 1 if the caller must zero-fill the outputs of forward/backward for this D_S (the classic,
// atomics-into-outputs pipeline; also the reference's contract), 0 if every element is written.
extern "C" int adgs_raster_needs_zero_init(int D_S) { return use_v2(D_S) ? 0 : 1; }
// The same question for the BACKWARD of a given forward: answered from the frame table (what that forward decided), never from the
// environment; a state the table does not know is answered "zero-fill" (always safe).
extern "C" int adgs_raster_backward_needs_zero_init(const char* geom_buffer, const char* img_buffer, int width, int height, int P) {
	FrameCfg cfg;
	if (!lookup_frame(FrameKey{ img_buffer, geom_buffer, width, height, P }, &cfg)) return 1;
	return cfg.v2 ? 0 : 1;
}

extern "C" void adgs_profile_enable(int stage_mask) { g_prof_mask.store((unsigned)stage_mask); }
// Pre-creates event objects so that a measurement loop never calls hipEventCreate (a driver call that can block).
extern "C" int adgs_profile_reserve(int n_events) {
	std::vector<hipEvent_t> fresh;
	for (int i = 0; i < n_events; i++) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) break; fresh.push_back(e); }
	std::lock_guard<std::mutex> lk(g_prof_mu);
	g_prof_pool.insert(g_prof_pool.end(), fresh.begin(), fresh.end());
	return (int)fresh.size();
}
extern "C" int adgs_profile_num_stages(void) { return ST_COUNT; }
extern "C" const char* adgs_profile_stage_name(int i) { return (i >= 0 && i < ST_COUNT) ? kStageNames[i] : ""; }
// Resolves all recorded (start, stop) event pairs: the caller must have synchronised the stream.
// total_ms[i] += elapsed, counts[i] += launches; arrays have adgs_profile_num_stages() entries.
extern "C" int adgs_profile_collect(double* total_ms, int64_t* counts) {
	std::lock_guard<std::mutex> lk(g_prof_mu);
	for (auto& r : g_prof_recs) {
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { total_ms[r.stage] += ms; counts[r.stage] += 1; }
		g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b);
	}
	g_prof_recs.clear();
	return 0;
}

extern "C" int adgs_device_check(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { set_error("no HIP device"); return -1; }
	hipDeviceProp_t prop;
	int dev = 0;
	ADGS_HIP_CHECK(hipGetDevice(&dev));
	ADGS_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
	if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
		set_error(std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
		return -2;
	}
	return 0;
}

// training = false: the forward-only render (adgs_raster_render*): nothing is kept for a backward
static int raster_forward_impl(const ShSource* sh_src, bool training,
	adgs_alloc_fn geometryBuffer, void* geometryUser,
	adgs_alloc_fn binningBuffer, void* binningUser,
	adgs_alloc_fn imageBuffer, void* imageUser,
	int P, int D, int M, int D_S,
	const float* background, int width, int height,
	const float* means3D, const float* shs, const float* colors_precomp, const float* flow_points, const float* semantic,
	const float* opacities, const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
	const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
	float* out_color, float* out_depth, float* img_opacity, float* img_flow, float* img_semantic,
	int inv_depth, int* radii, int debug, void* stream_) {
	(void)prefiltered;   // the reference only uses it to trap on an impossible cull (auxiliary.h:156-160)
	hipStream_t stream = (hipStream_t)stream_;
	FrameContext* fc = frame_context();
	if (P <= 0) return 0;
	if (D_S > MAX_SEMANTIC) { set_error("D_S exceeds 32 semantic channels"); return -1; }
	if (!means3D || !opacities || (!cov3D_precomp && (!scales || !rotations)) || !radii) { set_error("missing required input pointer"); return -1; }
	const int gx = (width + TILE_X - 1) / TILE_X, gy = (height + TILE_Y - 1) / TILE_Y;
	const size_t ntiles = (size_t)gx * gy, npix = (size_t)width * height;
	if (gx > 65535 || gy > 65535) { set_error("image too large"); return -1; }
	if (ntiles * 4 * (2 * (size_t)POOL_BLOCK + 1) > 0xffffffffull) { set_error("image too large: chunk slots are 32-bit"); return -1; }      // 4: wave tiles per 16x16 tile at most

	// semantic channels beyond the first are blended by a replay of the published lists: such a frame publishes them whatever it is for
	if (D_S > 1) training = true;
	if (use_v2(D_S)) {
		// the cell edge as asked for; a frame that takes the sorted path shrinks it until the mask bits fit its keys (decided below)
		int cell_tiles = v2_cell_tiles(gx, gy, false);
		int cgx = (gx + cell_tiles - 1) / cell_tiles, cgy = (gy + cell_tiles - 1) / cell_tiles;
		size_t ncells = (size_t)cgx * cgy;
		// Binning: bucket binning (binning.hip: per-cell lists built as independently sorted depth slabs, inside the CUs) unless the cell grid
		// has more than MAX_CELLS cells, ADGS_BINNING=sort asks for the device-wide radix sort of (cell | depth) keys, the previous frames
		// held more pairs than the largest slab count can split into sortable pieces (hundreds of millions), or ONE slab of the last
		// bucket-binned frame held more than ADGS_BUCKET_MAX_CELL_CHUNKS (16) x GS_NMAX entries (such a slab is bisected and its cell streamed
		// again once per half: a cliff the average does not show; the figure decays, so the bucket path is tried again -- with the bounds the
		// sorted frames have taught it in the meantime).  Slabs per cell: decided on the device from the cell's own pair count (cell_scan).
		// Until round 5 whole cells were cut into position chunks, sorted and merged, and C5 (9 chunks per cell) took the device-wide sort:
		// EXPERIMENTS.md.
		const char* binning_env = env_str("ADGS_BINNING");
		const std::string binning_mode = binning_env ? binning_env : "";
		{	// the hot-slab hint: what the device last reported (see the end of this block), decaying
			const MailboxRef* mbp = mailbox();
			const unsigned old_m = fc->hint_max_cell_chunks;
			const unsigned seen = (mbp && fc->bucket_frame_pending) ? mbp->host->max_cell_chunks : 0u;
			fc->hint_max_cell_chunks = std::max(seen, old_m - std::max(1u, old_m / 16) * (old_m ? 1u : 0u));
			fc->bucket_frame_pending = false;
		}
		bool buckets = ncells <= (size_t)MAX_CELLS && cell_tiles <= 16 && binning_mode != "sort" &&
			(binning_mode == "bucket" || (fc->hint_cells <= (size_t)std::max(1, env_int("ADGS_BUCKET_MAX_CHUNKS", 4)) * GS_NMAX * (ncells << MAX_SLAB_LG) &&
			                              fc->hint_max_cell_chunks <= (unsigned)std::max(1, env_int("ADGS_BUCKET_MAX_CELL_CHUNKS", 16))));
		if (!buckets && !v2_keys_fit(gx, gy, cell_tiles)) {
			cell_tiles = v2_cell_tiles(gx, gy, true);
			cgx = (gx + cell_tiles - 1) / cell_tiles; cgy = (gy + cell_tiles - 1) / cell_tiles; ncells = (size_t)cgx * cgy;
		}
		const bool sort_fallback_fits = v2_keys_fit(gx, gy, cell_tiles);      // may this frame still fall back to the sort (chunk table full)?
		size_t gb = 0, ib = 0;
		const size_t count_cells = buckets ? ncells : 0;      // the counts matrix [ceil(P / 256)][ncells] and the snapshot of the slab bounds exist for bucket binning only
		const bool with_ddir = training && sh_src && M == 16;
		GeomStateV2::carve(nullptr, P, &gb, with_ddir, count_cells);
		char* gch = geometryBuffer(geometryUser, gb);
		const int ppl = v2_pixels_per_lane(ntiles), sub = TILE_Y / (4 * ppl);
		const int wgy = (height + 4 * ppl - 1) / (4 * ppl);                    // rows of wave tiles
		const size_t wtiles = (size_t)gx * wgy;
		ImgStateV2::carve(nullptr, npix, wtiles, ncells, &ib);
		char* ich = imageBuffer(imageUser, ib);
		if (!gch || !ich) { set_error("buffer allocator returned NULL"); return -1; }
		GeomStateV2 geom = GeomStateV2::carve(gch, P, nullptr, with_ddir, count_cells);
		ImgStateV2 img = ImgStateV2::carve(ich, npix, wtiles, ncells, nullptr);
		uint32_t frame_word = 0; bool order_tiles = false;
		{
			FrameCfg fcfg{ 1, cell_tiles, ppl };
			fcfg.tile_order = env_int("ADGS_TILE_ORDER", 1) != 0; fcfg.sh_staging = env_str("ADGS_NO_SH_STAGING") == nullptr; fcfg.timeline = env_int("ADGS_TIMELINE_BWD", 0) != 0;
			// fewer tiles than wave slots: nothing to balance, neither in the backward nor in the forward
			order_tiles = wtiles >= 2048 && fcfg.tile_order;
			if (!training) order_tiles = false;      // no backward to balance, and an evaluation render does not come back to its camera
			fcfg.order_ready = order_tiles ? 1 : 0;
			if (training) remember_frame(FrameKey{ ich, gch, width, height, P }, fcfg);
			frame_word = frame_cfg_word(fcfg);
		}
		// ADGS_FWD_ORDER: 2 (default) = this camera's previous render decides the forward's tile order (bottom-up without one), 1 = bottom-up, 0 = top-down
		const int fwd_order_mode = env_int("ADGS_FWD_ORDER", 2);
		OrderHints::Entry* hint = nullptr;
		const uint32_t* hint_read = nullptr;
		const bool fwd_hint_wanted = wtiles >= 2048 && env_int("ADGS_TILE_ORDER", 1) != 0 && fwd_order_mode == 2;
		if (fwd_hint_wanted) {
			hipStreamCaptureStatus cs0 = hipStreamCaptureStatusNone;
			(void)hipStreamIsCapturing(stream, &cs0);
			if (OrderHints* oh = order_hints()) {
				const bool cap0 = cs0 == hipStreamCaptureStatusActive;
				hint = oh->find(viewmatrix, projmatrix, width, height, wtiles, stream, cap0);
				// a camera's first render has no hint (bottom-up).  Another camera's order is no substitute: measured with the bench's jittered
				// cameras (0.04 rad of yaw: the image shifts by a few tiles) it is WORSE than bottom-up, 910 - 920 against 928 frames/s.
				if (!hint && training) {
					// an entry is allocated outside the capture's rules: hipMalloc and the service stream's memset + wait are "potentially
					// unsafe" calls that a global-mode capture (torch.cuda.graph's default) forbids unless this thread switches to relaxed mode
					hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
					if (cap0) (void)hipThreadExchangeStreamCaptureMode(&mode);
					hint = oh->create(viewmatrix, projmatrix, width, height, wtiles, stream, cap0, std::min<size_t>(ncells, (size_t)MAX_CELLS) * SLAB_ROW);
					if (cap0) (void)hipThreadExchangeStreamCaptureMode(&mode);
					(void)hipGetLastError();      // a failed allocation costs the hint, not the frame
				}
				oh->lookups++;
				if (hint && hint->buf) { hint_read = hint->buf; if (hint->written) oh->hits++; fc->last_order_hint = hint->written ? 1 : 0; } else { hint = nullptr; fc->last_order_hint = 0; }
			}
		}

		PreprocessArgs pa;
		pa.P = P; pa.D = D; pa.M = M; pa.D_S = D_S;
		pa.means3D = means3D; pa.scales = scales; pa.scale_modifier = scale_modifier; pa.rotations = rotations;
		pa.opacities = opacities; pa.shs = shs; pa.cov3D_precomp = cov3D_precomp; pa.colors_precomp = colors_precomp;
		pa.flow_points = flow_points; pa.semantic = semantic;
		pa.view = viewmatrix; pa.proj = projmatrix; pa.campos = cam_pos;
		pa.W = width; pa.H = height; pa.gx = gx; pa.gy = gy;
		pa.tan_fovx = tan_fovx; pa.tan_fovy = tan_fovy;
		pa.focal_y = height / (2.0f * tan_fovy); pa.focal_x = width / (2.0f * tan_fovx);
		pa.inv_depth = inv_depth;
		pa.radii = radii; pa.splats = geom.splats; pa.cov3D = nullptr; pa.clamped = geom.clamped; pa.tiles_touched = geom.cells_touched;     // cov3D: recomputed by the backward
		pa.v2 = 1; pa.dupinfo = geom.dupinfo; pa.fine_touched = geom.fine_touched; pa.cell_tiles = cell_tiles; pa.cgx = cgx; pa.cgy = cgy;
		memset(&pa.sh_src, 0, sizeof(pa.sh_src));
		pa.sh0 = geom.sh0; pa.gacc = training ? geom.gacc : nullptr; pa.fine_total = geom.fine_total;      // forward-only: no accumulator lines to zero
		pa.bucket_count = nullptr;
		pa.cfg_word = img.header; pa.cfg_value = frame_word;
		pa.ddir = geom.ddir;      // raw-SH path: the backward will not read the `rest` rows a second time
		// bucket binning accumulates the fine-tile total and a few device words, and splits its cells by a snapshot of slab bounds (the
		// camera's own, else the thread's latest): zeroed / copied by the sh0 kernel on the raw-SH path (no launch of its own), by bin_prepare otherwise
		FramePrologue pro{ nullptr, 0, nullptr, nullptr, 0 };
		bool own_bounds = false, sample_bounds = false;
		if (buckets) {
			if (!fc->slab_bounds) {
				// first bucket-binned frame of this thread on this device: the table starts as "everything in slab 0" (all bounds 0xffffffff).
				// Allocated and filled outside the rules of a capture that may be running on `stream` (see OrderHints::create)
				hipStreamCaptureStatus csb = hipStreamCaptureStatusNone;
				(void)hipStreamIsCapturing(stream, &csb);
				hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
				if (csb == hipStreamCaptureStatusActive) (void)hipThreadExchangeStreamCaptureMode(&mode);
				bool ok = hipMalloc((void**)&fc->slab_bounds, (size_t)MAX_CELLS * SLAB_ROW * sizeof(uint32_t)) == hipSuccess;
				ok = ok && (fc->service || hipStreamCreateWithFlags(&fc->service, hipStreamNonBlocking) == hipSuccess);
				ok = ok && hipMemsetAsync(fc->slab_bounds, 0xff, (size_t)MAX_CELLS * SLAB_ROW * sizeof(uint32_t), fc->service) == hipSuccess && hipStreamSynchronize(fc->service) == hipSuccess;
				if (csb == hipStreamCaptureStatusActive) (void)hipThreadExchangeStreamCaptureMode(&mode);
				if (!ok) { if (fc->slab_bounds) (void)hipFree(fc->slab_bounds); fc->slab_bounds = nullptr; (void)hipGetLastError(); }
			}
			pro.zero = reinterpret_cast<uint32_t*>(geom.bucket_fine_total()); pro.n_zero = 2 * SCAN_AUX_SLOTS + 8;
			if (fc->slab_bounds) {
				// the camera's own bounds (its previous render left them in its hint entry) fit best; a camera's first render takes the thread's latest
				// ... the thread's latest on images too small for camera entries; else the frame samples its own (cell_sample)
				// (a frame being captured into a graph: its entry belongs to the graph, starts as "everything in slab 0" and is rewritten by every
				// replay -- the first replay bisects, the others split by their own previous bounds; baking "sample" in would cost every replay the pass)
				// A forward-only render never takes an entry's bounds: it does not rewrite them, so an entry found under RECYCLED matrix addresses
				// (another camera's, freed since) would misfit every render of the view -- a training frame heals such an entry with its first
				// render (measured: the bench's forward-only line at 1 585 instead of 1 980 frames/s behind the training lines of the same process).
				const bool own = training && hint && (hint->written || hint->captured) && hint->extra >= ncells * SLAB_ROW;
				own_bounds = own;
				sample_bounds = !own && fwd_hint_wanted && env_int("ADGS_SLAB_SAMPLE", 1) != 0;
				pro.copy_dst = geom.bounds; pro.copy_src = own ? hint->buf + wtiles + 16 : fc->slab_bounds; pro.n_copy = (int)(ncells * SLAB_ROW);
			}
			if (!sh_src && launch_bin_prepare(pro, stream) != 0) return -1;
			pa.bucket_count = geom.counts; pa.fine_total = geom.bucket_fine_total();
		}
		if (sh_src) { pa.sh_src = *sh_src; StageTimer t(ST_PREPROCESS, stream); if (launch_sh0(P, *sh_src, geom.sh0, stream, buckets ? &pro : nullptr) != 0) return -1; }
		if (const int evict_mb = env_int("ADGS_DBG_EVICT_MB", 0)) {
			// measurement hook (tools/stage_cache_experiment.py): a fill of `evict_mb` MiB between the sh0 kernel and the preprocess pushes what
			// the previous kernels wrote (deformed parameters, sh0) out of the 256 MiB Infinity Cache -- the state rocprofv3's serialised,
			// instrumented dispatches leave the preprocess in, and the reason its kernel-trace duration exceeds the HIP-event bracket of the
			// undisturbed frame
			static thread_local void* scratch = nullptr; static thread_local size_t scratch_bytes = 0;
			const size_t want = (size_t)evict_mb << 20;
			if (scratch_bytes < want) { if (scratch) (void)hipFree(scratch); scratch = nullptr; scratch_bytes = 0; if (hipMalloc(&scratch, want) == hipSuccess) scratch_bytes = want; }
			if (scratch) ADGS_HIP_CHECK(hipMemsetAsync(scratch, 0, want, stream));
		}
		{ StageTimer t(ST_PREPROCESS, stream); if (launch_preprocess_fwd(pa, stream) != 0) return -1; }
		ADGS_LAUNCH_CHECK(debug, stream);
		// The two totals (coarse (cell, Gaussian) pairs; fine-tile bound of the chunk pool) size the binning buffer.  Instead of
		// draining the stream for them (the reference's blocking cudaMemcpy, rasterizer_impl.cu:288) the WHOLE rest of the forward --
		// binning, sort, ranges and the blend -- is enqueued against a capacity (previous frames' counts + 25%) before the host knows
		// them; the device compares the exact totals with the capacity (cell_scan / publish_counts), and a frame that does not fit
		// blends nothing (render_fwd_v2 leaves an empty replay state).  Only then does the host look at the totals the device
		// published to its mailbox: the GPU has the complete forward queued by that time, so it never waits for the host, and in the
		// rare case that the capacity was too small the binning and the blend are enqueued again with exact sizes.
		// Under stream capture (HIP graphs) nothing can be read back: the frame is enqueued against the capacity and that is all;
		// the caller checks adgs_get_frame_status() after a replay and re-captures after an eager frame when it reports an overflow.
		MailboxRef* mb = mailbox();
		if (!mb) { set_error("hipHostMalloc (mapped) failed"); return -1; }
		const uint32_t seq = mb->next_seq++;
		hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
		(void)hipStreamIsCapturing(stream, &cap_status);
		const bool capturing = cap_status == hipStreamCaptureStatusActive;
		const bool speculate = capturing || env_int("ADGS_NO_SPECULATION", 0) == 0;
		const size_t cap_cells = std::min<size_t>(std::max<size_t>(fc->hint_cells, (size_t)P + 4096), 0x7fffffffu);
		const size_t cap_fine = std::max<size_t>(fc->hint_fine, (size_t)8 * P + 4096);
		mb->cap_cells = cap_cells; mb->cap_fine = cap_fine;
		uint32_t* overflow_flag = geom.d_counts() + 3;         // written by cell_scan / publish_counts in every frame
		CellScanArgs sa;
		sa.cell_count = geom.cell_count(); sa.cell_start = geom.cell_start; sa.cell_ranges = img.cell_ranges; sa.ncells = (int)ncells;
		sa.max_chunks = (uint32_t)std::min(MAX_CHUNKS, std::max(1, env_int("ADGS_MAX_CHUNKS", MAX_CHUNKS))); sa.d_counts = geom.d_counts();      // ADGS_MAX_CHUNKS: test hook for the overflow fallback
		sa.fine_total = geom.bucket_fine_total(); sa.cell_work = geom.cell_work; sa.force_lg = buckets ? env_int("ADGS_SLABS_LG", -1) : -1;
		sa.slab_target = own_bounds ? (uint32_t)SLAB_TARGET : (uint32_t)SLAB_TARGET_FOREIGN;      // sampled bounds carry sampling noise: measured, a 3072 target overflows a slab in one view of a few
		sa.cap_cells = speculate ? (uint32_t)cap_cells : 0xffffffffu; sa.cap_fine = speculate ? (unsigned long long)cap_fine : ~0ull;
		// cell_scan publishes the frame's totals to the host mailbox (the first kernel that knows them: until the middle of round 6 the LAST
		// kernel of the binning did, and the host came back from its wait with only the blend left in the queue -- 0.2 ms to get the backward's
		// first kernel enqueued, which a slow host misses: C3 at 1.01 ms per step against 0.96 ms of kernels on such a box)
		sa.box = mb->dev; sa.seq = seq;
		if (buckets) {
			StageTimer t(ST_SCAN, stream);
			if (launch_cell_colscan(geom.counts, (P + 255) / 256, (int)ncells, geom.cell_count(), stream) != 0) return -1;
			if (launch_cell_scan(sa, stream) != 0) return -1;
		} else {
			{
				StageTimer t(ST_SCAN, stream);
				// offsets of the (cell, Gaussian) pairs; the fine-tile bound of the chunk pool only needs its total
				if (exclusive_scan_u32_sum(geom.cells_touched, geom.offsets, (size_t)P + 1, geom.scan_temp, geom.fine_touched, geom.fine_total, stream) != 0) return -1;
			}
			hipLaunchKernelGGL(publish_counts_kernel, dim3(1), dim3(WAVE), 0, stream, (const uint32_t*)(geom.offsets + P), (const unsigned long long*)geom.fine_total, mb->dev, seq,
				speculate ? (uint32_t)cap_cells : 0xffffffffu, speculate ? (unsigned long long)cap_fine : ~0ull, overflow_flag);
			ADGS_HIP_CHECK(hipGetLastError());
		}
		ADGS_LAUNCH_CHECK(debug, stream);
		const int bit = (int)higher_msb((uint32_t)ncells);
		BinStateV2 bin;
		// rectangle-coverage masks in the key bits above (cell | depth): one bit per tile row and per tile column of a cell
		const int mask_shift = 32 + bit;          // 2 * cell_tiles + bit <= 32: v2_cell_tiles
		auto enqueue_binning = [&](size_t cells, size_t fine, const uint32_t* d_count) -> int {
			size_t bb = 0;
			if (buckets) {
				BinStateV2::carve_buckets(nullptr, cells, training ? fine * sub : 0, training ? wtiles : 0, &bb, ncells);      // forward-only: no chunk pool
				char* bch = binningBuffer(binningUser, bb);
				if (!bch) { set_error("binning allocator returned NULL"); return -1; }
				bin = BinStateV2::carve_buckets(bch, cells, training ? fine * sub : 0, training ? wtiles : 0, nullptr, ncells);
				if (cells == 0) { ADGS_HIP_CHECK(hipMemsetAsync(bin.pool_cursor, 0, sizeof(uint32_t), stream)); return 0; }     // cell_ranges: all (0, 0) from bucket_scan
				const uint32_t cap = (uint32_t)std::min<size_t>(cells, 0xffffffffu);
				{ StageTimer t(ST_DUPLICATE, stream);
				  if (launch_cell_scatter(P, geom.dupinfo, geom.cell_start, geom.counts, bin.rec_key, bin.rec_im, cap, cell_tiles, cgx, (int)ncells, bin.pool_cursor, bin.slow_list, stream) != 0) return -1; }
				ADGS_LAUNCH_CHECK(debug, stream);
				SlabSortArgs ga;
				ga.cell_ranges = img.cell_ranges; ga.ncells = (int)ncells; ga.cell_work = geom.cell_work; ga.grid = (uint32_t)BinStateV2::slab_grid(cells, ncells); ga.rec_key = bin.rec_key; ga.rec_im = bin.rec_im; ga.ent_f = bin.entries; ga.cap = cap;
				ga.bounds = geom.bounds; ga.bounds_out = fc->slab_bounds; ga.d_counts = geom.d_counts(); ga.slow_list = bin.slow_list;
				ga.bounds_out2 = (training && hint && hint->extra >= ncells * SLAB_ROW) ? hint->buf + wtiles + 16 : nullptr;
				ga.box = mb->dev;
				{ StageTimer t(ST_SORT, stream);
				  if (sample_bounds && launch_cell_sample(ga, geom.bounds, stream) != 0) return -1;
				  if (launch_slab_sort(ga, stream) != 0) return -1; }
				ADGS_LAUNCH_CHECK(debug, stream);
				return 0;
			}
			BinStateV2::carve(nullptr, cells, training ? fine * sub : 0, training ? wtiles : 0, &bb);        // a Gaussian can enter both halves of a tile
			char* bch = binningBuffer(binningUser, bb);
			if (!bch) { set_error("binning allocator returned NULL"); return -1; }
			bin = BinStateV2::carve(bch, cells, training ? fine * sub : 0, training ? wtiles : 0, nullptr);
			if (cells == 0) {
				ADGS_HIP_CHECK(hipMemsetAsync(img.cell_ranges, 0, ncells * sizeof(uint2), stream));
				ADGS_HIP_CHECK(hipMemsetAsync(bin.pool_cursor, 0, sizeof(uint32_t), stream));
				return 0;
			}
			{ StageTimer t(ST_DUPLICATE, stream);
			  if (launch_duplicate_cells(P, geom.dupinfo, geom.offsets, bin.keys_unsorted, bin.list_unsorted, (uint32_t)std::min<size_t>(cells, 0xffffffffu),
			        cell_tiles, cgx, img.cell_ranges, (int)ncells, bin.pool_cursor, mask_shift, stream) != 0) return -1; }
			ADGS_LAUNCH_CHECK(debug, stream);
			{ StageTimer t(ST_SORT, stream);
			  if (radix_sort_pairs_u64_dn(bin.keys_unsorted, bin.keys, bin.list_unsorted, bin.list, cells, d_count, 32 + bit, bin.sort_temp, stream) != 0) return -1; }
			ADGS_LAUNCH_CHECK(debug, stream);
			{ StageTimer t(ST_RANGES, stream);
			  if (launch_tile_ranges((int)cells, d_count, bin.keys, img.cell_ranges, bit >= 32 ? 0xffffffffu : ((1u << bit) - 1u), stream) != 0) return -1;
			  // a sorted frame teaches the bucket path its depth-slab bounds too (the cells' 128-quantiles, read off the sorted keys)
			  if (fc->slab_bounds && ncells <= (size_t)MAX_CELLS &&
			      launch_bounds_from_sorted(bin.keys, img.cell_ranges, (int)ncells, d_count, (uint32_t)std::min<size_t>(cells, 0xffffffffu), fc->slab_bounds, stream) != 0) return -1; }
			ADGS_LAUNCH_CHECK(debug, stream);
			return 0;
		};
		auto launch_blend = [&]() -> int {      // on the binning state `bin` of the last enqueue_binning
			RenderV2FwdArgs ra;
			ra.cell_ranges = img.cell_ranges; ra.cell_list = bin.list; ra.splats = geom.splats;
			ra.cell_keys = bin.keys; ra.mask_shift = mask_shift; ra.cell_entries = buckets ? bin.entries : nullptr;
			ra.W = width; ra.H = height; ra.gx = gx; ra.gy = wgy; ra.ppl = ppl; ra.cell_tiles = cell_tiles; ra.cgx = cgx;
			ra.has_color = (colors_precomp != nullptr) || (shs != nullptr) || (sh_src != nullptr);
			ra.has_flow = flow_points != nullptr; ra.has_sem = (semantic != nullptr) && D_S > 0;
			ra.bg = background; ra.bg_image = sh_src ? sh_src->bg_image : nullptr;
			ra.pool = bin.pool; ra.pool_cursor = bin.pool_cursor; ra.tile_last_chunk = img.tile_last_chunk; ra.tile_consumed = img.tile_consumed; ra.tile_scanned = img.tile_scanned; ra.tile_batches = img.tile_batches;
			ra.final_T = img_opacity; ra.n_contrib = img.n_contrib;
			ra.out_color = out_color; ra.out_depth = out_depth; ra.out_flow = img_flow; ra.out_semantic = img_semantic;
			ra.order_mode = fwd_order_mode == 2 ? 1 : fwd_order_mode; ra.fwd_order = nullptr;
			ra.fwd_view = viewmatrix; ra.fwd_sig = nullptr;
			if (hint_read) { ra.order_mode = 2; ra.fwd_order = hint_read; ra.fwd_sig = reinterpret_cast<const float*>(hint_read + wtiles); }
			ra.overflow_flag = overflow_flag; ra.publish = training;
			{ StageTimer t(ST_RENDER_FWD, stream);
			  if (launch_render_fwd_v2(ra, stream) != 0) return -1;
			  for (int c0 = 1; ra.has_sem && c0 < D_S; c0 += 4) {      // semantic channels beyond the Splat's slot: a replay of the published lists
				RenderV2SemFwdArgs sa;
				sa.splats = geom.splats; sa.pool = bin.pool; sa.tile_last_chunk = img.tile_last_chunk; sa.tile_consumed = img.tile_consumed; sa.n_contrib = img.n_contrib;
				sa.W = width; sa.H = height; sa.gx = gx; sa.gy = wgy; sa.ppl = ppl;
				sa.semantic = semantic; sa.D_S = D_S; sa.c0 = c0; sa.nch = std::min(4, D_S - c0); sa.out_semantic = img_semantic;
				if (launch_render_sem_fwd_v2(sa, stream) != 0) return -1;
			  } }
			// the longest-first order of the tiles: for this frame's backward (it used to launch this itself) and, as a copy, for the next forward of this camera
			if (order_tiles) {
				StageTimer t(ST_RENDER_FWD, stream);
				if (launch_tile_order((int)wtiles, img.tile_consumed, img.tile_order, stream, hint ? hint->buf : nullptr, viewmatrix) != 0) return -1;
				if (hint) hint->written = true;      // (a second blend of this call -- the capacity re-run -- reads the order the first one left)
			}
			ADGS_LAUNCH_CHECK(debug, stream);
			return 0;
		};
		if (speculate) {
			if (enqueue_binning(cap_cells, cap_fine, buckets ? nullptr : geom.offsets + P) != 0) return -1;
			if (launch_blend() != 0) return -1;
		}
		if (capturing) {
			// nothing of this frame can be read back inside a capture: the totals stay in the mailbox for adgs_get_frame_status()
			fc->stats.num_rendered = (int64_t)cap_cells; fc->stats.tiles = (int32_t)ntiles; fc->stats.sort_bits = buckets ? 32 : 32 + bit;
			fc->stats.sort_passes = buckets ? 4 : (32 + bit + 7) / 8; fc->stats.reserved = buckets ? 1 : 0; fc->stats.fine_pairs = (int64_t)cap_fine;
			return (int)cap_cells;
		}
		if (wait_mailbox(mb, seq, stream) != 0) return -1;
		const size_t R_cells = mb->host->r_cells, R_fine = (size_t)mb->host->r_fine;
		if (mb->host->overflow) mb->repaired++;      // an eager frame that did not fit is enqueued again below, before this call returns
		const bool chunk_table_full = buckets && mb->host->oversize;
		if (!speculate || chunk_table_full || R_cells > cap_cells || R_fine > cap_fine) {
			if (speculate) fc->reruns++;
			// the re-run fits by construction; the device word can also be set without speculation (cell_scan raises it when the chunk
			// table is full), and a blend launched with it set renders nothing
			ADGS_HIP_CHECK(hipMemsetAsync(overflow_flag, 0, sizeof(uint32_t), stream));
			if (buckets && !chunk_table_full) {      // slab_sort's completion counter / fullest slab / hand-over list
				ADGS_HIP_CHECK(hipMemsetAsync(geom.d_counts() + 4, 0, 2 * sizeof(uint32_t), stream));
			}
			if (chunk_table_full) {
				// more chunks than the chunk table holds (> 100 M pairs): this frame takes the device-wide radix sort, which needs the
				// per-Gaussian pair offsets first
				if (!sort_fallback_fits) { set_error("frame exceeds the bucket chunk table and its cell edge does not fit the sort keys: set ADGS_BINNING=sort (or a smaller ADGS_CELL_TILES)"); return -1; }
				buckets = false;
				ADGS_HIP_CHECK(hipMemsetAsync(geom.fine_total, 0, SCAN_AUX_SLOTS * sizeof(unsigned long long), stream));
				StageTimer t(ST_SCAN, stream);
				if (exclusive_scan_u32_sum(geom.cells_touched, geom.offsets, (size_t)P + 1, geom.scan_temp, geom.fine_touched, geom.fine_total, stream) != 0) return -1;
			}
			if (enqueue_binning(R_cells, R_fine, nullptr) != 0) return -1;
			if (launch_blend() != 0) return -1;
		}
		{	// capacity hints for the next frame: 25% head-room over this frame, slow decay of older peaks
			const size_t want_c = R_cells + R_cells / 4 + 4096, want_f = R_fine + R_fine / 4 + 4096;
			const size_t old_c = fc->hint_cells, old_f = fc->hint_fine;
			fc->hint_cells = std::max(want_c, old_c - old_c / 16);
			fc->hint_fine = std::max(want_f, old_f - old_f / 16);
			// the fullest slab of a bucket-binned frame: the LATEST figure the device has reported (slab_sort_slow writes it when the frame's
			// binning is through -- this frame's if the caller drains the stream between frames, an earlier one in a free-running loop: it is a
			// policy hint); a sorted frame does not report one: the old figure decays, so that a scene with a persistent hot cell re-tries
			// the bucket path once in a few dozen frames instead of every other frame
			fc->bucket_frame_pending = buckets;
		}
		fc->stats.num_rendered = (int64_t)R_cells; fc->stats.tiles = (int32_t)ntiles; fc->stats.sort_bits = buckets ? 32 : 32 + bit;
		fc->stats.sort_passes = buckets ? 4 : (32 + bit + 7) / 8;
		fc->stats.reserved = buckets ? 1 : 0; fc->stats.fine_pairs = (int64_t)R_fine;       // reserved: 1 = bucket binning (the sort passes stay inside the CUs)
		return (int)R_cells;
	}

	size_t gbytes = 0, ibytes = 0;
	GeomState::carve(nullptr, P, &gbytes);
	char* gchunk = geometryBuffer(geometryUser, gbytes);
	ImgState::carve(nullptr, npix, ntiles, &ibytes);
	char* ichunk = imageBuffer(imageUser, ibytes);
	if (!gchunk || !ichunk) { set_error("buffer allocator returned NULL"); return -1; }
	GeomState geom = GeomState::carve(gchunk, P, nullptr);
	ImgState img = ImgState::carve(ichunk, npix, ntiles, nullptr);
	uint32_t frame_word = 0;
	{
		FrameCfg fcfg{ 0, 1, 4 };
		fcfg.sh_staging = env_str("ADGS_NO_SH_STAGING") == nullptr;
		remember_frame(FrameKey{ ichunk, gchunk, width, height, P }, fcfg);
		frame_word = frame_cfg_word(fcfg);
	}

	PreprocessArgs pa;
	pa.P = P; pa.D = D; pa.M = M; pa.D_S = D_S;
	pa.means3D = means3D; pa.scales = scales; pa.scale_modifier = scale_modifier; pa.rotations = rotations;
	pa.opacities = opacities; pa.shs = shs; pa.cov3D_precomp = cov3D_precomp; pa.colors_precomp = colors_precomp;
	pa.flow_points = flow_points; pa.semantic = semantic;
	pa.view = viewmatrix; pa.proj = projmatrix; pa.campos = cam_pos;
	pa.W = width; pa.H = height; pa.gx = gx; pa.gy = gy;
	pa.tan_fovx = tan_fovx; pa.tan_fovy = tan_fovy;
	pa.focal_y = height / (2.0f * tan_fovy);     // rasterizer_impl.cu:229-230
	pa.focal_x = width / (2.0f * tan_fovx);
	pa.inv_depth = inv_depth;
	pa.radii = radii; pa.splats = geom.splats; pa.cov3D = geom.cov3D; pa.clamped = geom.clamped; pa.tiles_touched = geom.tiles_touched;
	pa.v2 = 0; pa.dupinfo = nullptr; pa.fine_touched = nullptr; pa.cell_tiles = 1; pa.cgx = gx; pa.cgy = gy;
	memset(&pa.sh_src, 0, sizeof(pa.sh_src)); pa.sh0 = nullptr; pa.gacc = nullptr; pa.fine_total = nullptr;
	pa.bucket_count = nullptr;
	pa.cfg_word = img.header; pa.cfg_value = frame_word; pa.ddir = nullptr;
	if (sh_src) { set_error("the raw-SH entry points need the default (v2) pipeline (not ADGS_RASTER_MODE=classic, D_S <= ADGS_V2_MAX_SEMANTIC)"); return -1; }
	{ StageTimer t(ST_PREPROCESS, stream); if (launch_preprocess_fwd(pa, stream) != 0) return -1; }
	ADGS_LAUNCH_CHECK(debug, stream);

	// tiles_touched[P] = 0 sentinel so that the exclusive scan yields the total at [P]
	{
		StageTimer t(ST_SCAN, stream);
		if (exclusive_scan_u32(geom.tiles_touched, geom.offsets, (size_t)P + 1, geom.scan_temp, stream) != 0) return -1;
	}
	ADGS_LAUNCH_CHECK(debug, stream);

	uint32_t* host_word = pinned_word();
	if (!host_word) { set_error("hipHostMalloc failed"); return -1; }
	ADGS_HIP_CHECK(hipMemcpyAsync(host_word, geom.offsets + P, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
	ADGS_HIP_CHECK(hipStreamSynchronize(stream));
	const int num_rendered = (int)*host_word;

	size_t bbytes = 0;
	BinState::carve(nullptr, (size_t)num_rendered, &bbytes);
	char* bchunk = binningBuffer(binningUser, bbytes);
	if (!bchunk) { set_error("binning allocator returned NULL"); return -1; }
	BinState bin = BinState::carve(bchunk, (size_t)num_rendered, nullptr);

	{ StageTimer t(ST_DUPLICATE, stream);
	  if (launch_duplicate_keys(P, geom.splats, geom.offsets, radii, bin.keys_unsorted, bin.list_unsorted, gx, gy, stream) != 0) return -1; }
	ADGS_LAUNCH_CHECK(debug, stream);
	const int bit = (int)higher_msb((uint32_t)ntiles);
	{ StageTimer t(ST_SORT, stream);
	  if (radix_sort_pairs_u64(bin.keys_unsorted, bin.keys, bin.list_unsorted, bin.list, (size_t)num_rendered, 32 + bit, bin.sort_temp, stream) != 0) return -1; }
	ADGS_LAUNCH_CHECK(debug, stream);
	{ StageTimer t(ST_RANGES, stream);
	  ADGS_HIP_CHECK(hipMemsetAsync(img.ranges, 0, ntiles * sizeof(uint2), stream));
	  if (launch_tile_ranges(num_rendered, nullptr, bin.keys, img.ranges, 0xffffffffu, stream) != 0) return -1; }
	ADGS_LAUNCH_CHECK(debug, stream);

	RenderFwdArgs ra;
	ra.ranges = img.ranges; ra.point_list = bin.list; ra.splats = geom.splats;
	ra.W = width; ra.H = height; ra.gx = gx; ra.gy = gy; ra.D_S = D_S;
	ra.has_color = (colors_precomp != nullptr) || (shs != nullptr);
	ra.has_flow = flow_points != nullptr; ra.has_sem = (semantic != nullptr) && D_S > 0; ra.inv_depth = inv_depth != 0;
	ra.semantic = semantic; ra.bg = background;
	ra.final_T = img_opacity; ra.n_contrib = img.n_contrib;
	ra.out_color = out_color; ra.out_depth = out_depth; ra.out_flow = img_flow; ra.out_semantic = img_semantic;
	{ StageTimer t(ST_RENDER_FWD, stream); if (launch_render_fwd(ra, stream) != 0) return -1; }
	ADGS_LAUNCH_CHECK(debug, stream);

	fc->stats.num_rendered = num_rendered; fc->stats.tiles = (int32_t)ntiles; fc->stats.sort_bits = 32 + bit; fc->stats.sort_passes = (32 + bit + 7) / 8;
	fc->stats.fine_pairs = num_rendered;
	return num_rendered;
}

static int raster_backward_impl(const ShSource* sh_src, const ShGradDst* sh_dst,
	int P, int D, int M, int R, int D_S,
	const float* background, int width, int height,
	const float* means3D, const float* shs, const float* colors_precomp, const float* flow_points, const float* semantic,
	const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
	const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
	const int* radii, char* geom_buffer, char* binning_buffer, char* img_buffer,
	const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dpix_flow, const float* dL_dpix_semantic,
	float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D,
	float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dflow, float* dL_dsemantic,
	const float* grad_img_opacity, const float* img_opacity, int inv_depth, int debug, void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	if (P <= 0) return 0;
	if (!geom_buffer || !img_buffer || (R > 0 && !binning_buffer)) { set_error("backward called without forward state buffers"); return -1; }
	const int gx = (width + TILE_X - 1) / TILE_X, gy = (height + TILE_Y - 1) / TILE_Y;
	const size_t ntiles = (size_t)gx * gy, npix = (size_t)width * height;
	FrameCfg cfg;
	const FrameKey fkey{ img_buffer, geom_buffer, width, height, P };
	if (!lookup_frame(fkey, &cfg)) {
		// State buffers this library did not hand out at these addresses (cloned / offloaded saved tensors), or a forward that has
		// dropped out of the frame table.  The backward never reads the environment: the forward's preprocess kernel wrote the frame's
		// configuration into the header of the image state, and this rare path reads that word back (one blocking 4-byte copy).
		hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
		(void)hipStreamIsCapturing(stream, &cs);
		if (cs == hipStreamCaptureStatusActive) { set_error("backward over state buffers the captured forward did not allocate: their configuration cannot be read back inside a stream capture"); return -1; }
		uint32_t word = 0;
		ADGS_HIP_CHECK(hipMemcpyAsync(&word, img_buffer, sizeof(word), hipMemcpyDeviceToHost, stream));
		ADGS_HIP_CHECK(hipStreamSynchronize(stream));
		if (!frame_cfg_from_word(word, &cfg)) { set_error("backward: the image state buffer does not carry a forward's configuration word (not written by adgs_raster_forward?)"); return -1; }
	}
	if (cfg.v2) {
		const int cell_tiles = cfg.cell_tiles;
		const size_t ncells = (size_t)((gx + cell_tiles - 1) / cell_tiles) * ((gy + cell_tiles - 1) / cell_tiles);
		GeomStateV2 geom = GeomStateV2::carve(geom_buffer, P, nullptr, sh_src && M == 16);
		const int ppl = cfg.ppl;
		const int wgy = (height + 4 * ppl - 1) / (4 * ppl);
		const size_t wtiles = (size_t)gx * wgy;
		ImgStateV2 img = ImgStateV2::carve(img_buffer, npix, wtiles, ncells, nullptr);
		BinStateV2 bin = BinStateV2::carve(binning_buffer, 0, 0, wtiles, nullptr);     // only pool_cursor / pool are used
		const bool has_color = (colors_precomp != nullptr) || (shs != nullptr) || (sh_src != nullptr);
		RenderV2BwdArgs ra;
		ra.splats = geom.splats; ra.pool = bin.pool; ra.tile_last_chunk = img.tile_last_chunk; ra.tile_consumed = img.tile_consumed;
		ra.W = width; ra.H = height; ra.gx = gx; ra.gy = wgy; ra.ppl = ppl;
		ra.bg = background; ra.final_T = img_opacity; ra.n_contrib = img.n_contrib;
		ra.bg_image = sh_src ? sh_src->bg_image : nullptr; ra.dL_dbg_image = (sh_src && sh_dst && ra.bg_image && dL_dpix) ? sh_dst->bg_image : nullptr;
		ra.dL_dpix = dL_dpix; ra.dL_dpix_depth = dL_dpix_depth; ra.dL_dpix_flow = dL_dpix_flow; ra.dL_dpix_sem = dL_dpix_semantic;
		ra.dL_dpix_opacity = grad_img_opacity;
		ra.do_color = dL_dpix && has_color;
		ra.do_flow = dL_dpix_flow && flow_points;
		ra.do_sem = dL_dpix_semantic && semantic && D_S > 0;
		ra.do_depth = dL_dpix_depth != nullptr;
		ra.do_opacity = grad_img_opacity != nullptr || (ra.bg_image != nullptr && ra.do_color);      // the per-pixel background's term rides on the opacity path
		ra.gacc = geom.gacc;
		ra.tile_order = nullptr;
		ra.tl_start = cfg.timeline ? img.tile_scanned : nullptr; ra.tl_end = img.tile_batches;      // experiment build only
		{
			StageTimer t(ST_RENDER_BWD, stream);
			if (wtiles >= 2048 && cfg.tile_order) {      // fewer tiles than wave slots: nothing to balance
				// the forward built the order behind its blend kernel (order_ready); a state from a library build that did not: here
				if (!cfg.order_ready && launch_tile_order((int)wtiles, img.tile_consumed, img.tile_order, stream) != 0) return -1;
				ra.tile_order = img.tile_order;
			}
			// geom.gacc lines of the visible Gaussians were zeroed by the forward preprocess: the FIRST backward over a forward
			// accumulates into them as they are.  A second backward over the same forward state (retain_graph) -- or one whose
			// forward has dropped out of the frame table -- gets them zeroed again here (64 B per Gaussian: the preprocess
			// backward used to re-zero every line after reading it, 64 MB of writes per frame at C3 for a case that is rare).
			if (note_backward(fkey) != 0) ADGS_HIP_CHECK(hipMemsetAsync(geom.gacc, 0, (size_t)P * GACC_STRIDE * sizeof(float), stream));
			ra.sem_src = nullptr; ra.sem_dst = nullptr; ra.sem_stride = 0;
			if (binning_buffer && launch_render_bwd_v2(ra, stream) != 0) return -1;
			if (binning_buffer && ra.do_sem && D_S > 1) {
				// channels 1 .. D_S-1: dL/dalpha is linear in the channels, so each channel's replay adds its share of the geometric
				// sums to the gacc lines and its own sum_k alpha_k T_k dL/dS to dL_dsemantic[:, c] (zeroed here; channel 0 is written
				// by the preprocess backward with every other output row)
				ADGS_HIP_CHECK(hipMemsetAsync(dL_dsemantic, 0, (size_t)P * D_S * sizeof(float), stream));
				RenderV2BwdArgs rc = ra;
				rc.do_color = rc.do_flow = rc.do_depth = rc.do_opacity = false; rc.do_sem = true;
				rc.dL_dpix = nullptr; rc.dL_dpix_depth = nullptr; rc.dL_dpix_flow = nullptr; rc.dL_dpix_opacity = nullptr;
				rc.bg_image = nullptr; rc.dL_dbg_image = nullptr; rc.sem_stride = D_S;
				for (int c = 1; c < D_S; c++) {
					rc.dL_dpix_sem = dL_dpix_semantic + (size_t)c * npix; rc.sem_src = semantic + c; rc.sem_dst = dL_dsemantic + c;
					if (launch_render_bwd_v2(rc, stream) != 0) return -1;
				}
			}
		}
		ADGS_LAUNCH_CHECK(debug, stream);
		PreprocessBwdArgs pa;
		pa.P = P; pa.D = D; pa.M = M;
		pa.means3D = means3D; pa.radii = radii; pa.shs = shs; pa.clamped = geom.clamped;
		pa.scales = scales; pa.rotations = rotations; pa.scale_modifier = scale_modifier;
		pa.cov3D = cov3D_precomp;                // NULL: recomputed from scales / rotations
		pa.view = viewmatrix; pa.proj = projmatrix; pa.campos = campos;
		pa.focal_y = height / (2.0f * tan_fovy); pa.focal_x = width / (2.0f * tan_fovx);
		pa.tan_fovx = tan_fovx; pa.tan_fovy = tan_fovy; pa.inv_depth = inv_depth;
		pa.dL_dmean2D = nullptr; pa.dL_dconic = nullptr; pa.dL_dcolor = nullptr; pa.dL_ddepth = nullptr;
		pa.dL_dmean3D = dL_dmean3D; pa.dL_dcov3D = dL_dcov3D; pa.dL_dsh = dL_dsh; pa.dL_dscale = dL_dscale; pa.dL_drot = dL_drot;
		pa.gacc = geom.gacc; pa.splats = geom.splats; pa.W = width; pa.H = height;
		memset(&pa.sh_src, 0, sizeof(pa.sh_src)); pa.sh_dst = ShGradDst{};
		if (sh_src) { pa.sh_src = *sh_src; if (sh_dst) pa.sh_dst = *sh_dst; }
		pa.out_mean2D = dL_dmean2D; pa.out_conic = dL_dconic; pa.out_opacity = dL_dopacity; pa.out_color = dL_dcolor; pa.out_depth = dL_ddepth;
		pa.out_flow = ra.do_flow ? dL_dflow : nullptr; pa.out_sem = ra.do_sem ? dL_dsemantic : nullptr; pa.D_S = D_S;
		pa.sh_staging = cfg.sh_staging;
		pa.ddir = geom.ddir;
		{ StageTimer t(ST_PREPROCESS_BWD, stream); if (launch_preprocess_bwd(pa, stream) != 0) return -1; }
		ADGS_LAUNCH_CHECK(debug, stream);
		return 0;
	}
	GeomState geom = GeomState::carve(geom_buffer, P, nullptr);
	ImgState img = ImgState::carve(img_buffer, npix, ntiles, nullptr);
	BinState bin = BinState::carve(binning_buffer, (size_t)R, nullptr);

	const bool has_color = (colors_precomp != nullptr) || (shs != nullptr);
	RenderBwdArgs ra;
	ra.ranges = img.ranges; ra.point_list = bin.list; ra.splats = geom.splats;
	ra.W = width; ra.H = height; ra.gx = gx; ra.gy = gy; ra.D_S = D_S;
	ra.semantic = semantic; ra.bg = background;
	ra.final_T = img_opacity; ra.n_contrib = img.n_contrib;
	ra.dL_dpix = dL_dpix; ra.dL_dpix_depth = dL_dpix_depth; ra.dL_dpix_flow = dL_dpix_flow; ra.dL_dpix_sem = dL_dpix_semantic;
	ra.dL_dpix_opacity = grad_img_opacity;
	// gating as in backward.cu:497-506
	ra.do_color = dL_dpix && has_color;
	ra.do_flow = dL_dpix_flow && flow_points;
	ra.do_sem = dL_dpix_semantic && semantic && D_S > 0;
	ra.do_depth = dL_dpix_depth != nullptr;
	ra.do_opacity = grad_img_opacity != nullptr;
	ra.dL_dmean2D = dL_dmean2D; ra.dL_dconic = dL_dconic; ra.dL_dopacity = dL_dopacity; ra.dL_dcolor = dL_dcolor;
	ra.dL_ddepth = dL_ddepth; ra.dL_dflow = dL_dflow; ra.dL_dsem = dL_dsemantic;
	if (R > 0) {
		{ StageTimer t(ST_RENDER_BWD, stream); if (launch_render_bwd(ra, stream) != 0) return -1; }
		ADGS_LAUNCH_CHECK(debug, stream);
	}

	PreprocessBwdArgs pa;
	pa.P = P; pa.D = D; pa.M = M;
	pa.means3D = means3D; pa.radii = radii; pa.shs = shs; pa.clamped = geom.clamped;
	pa.scales = scales; pa.rotations = rotations; pa.scale_modifier = scale_modifier;
	pa.cov3D = cov3D_precomp ? cov3D_precomp : geom.cov3D;
	pa.view = viewmatrix; pa.proj = projmatrix; pa.campos = campos;
	pa.focal_y = height / (2.0f * tan_fovy); pa.focal_x = width / (2.0f * tan_fovx);
	pa.tan_fovx = tan_fovx; pa.tan_fovy = tan_fovy; pa.inv_depth = inv_depth;
	pa.dL_dmean2D = dL_dmean2D; pa.dL_dconic = dL_dconic; pa.dL_dcolor = dL_dcolor; pa.dL_ddepth = dL_ddepth;
	pa.dL_dmean3D = dL_dmean3D; pa.dL_dcov3D = dL_dcov3D; pa.dL_dsh = dL_dsh; pa.dL_dscale = dL_dscale; pa.dL_drot = dL_drot;
	if (sh_src) { set_error("the raw-SH entry points need the default (v2) pipeline (not ADGS_RASTER_MODE=classic, D_S <= ADGS_V2_MAX_SEMANTIC)"); return -1; }
	memset(&pa.sh_src, 0, sizeof(pa.sh_src)); pa.sh_dst = ShGradDst{};
	pa.gacc = nullptr; pa.splats = nullptr; pa.W = width; pa.H = height; pa.out_mean2D = nullptr; pa.out_conic = nullptr; pa.out_opacity = nullptr; pa.out_color = nullptr; pa.out_depth = nullptr;
	pa.out_flow = nullptr; pa.out_sem = nullptr; pa.D_S = D_S;
	pa.sh_staging = cfg.sh_staging; pa.ddir = nullptr;
	{ StageTimer t(ST_PREPROCESS_BWD, stream); if (launch_preprocess_bwd(pa, stream) != 0) return -1; }
	ADGS_LAUNCH_CHECK(debug, stream);
	return 0;
}

extern "C" int adgs_raster_forward(
	adgs_alloc_fn geometryBuffer, void* geometryUser, adgs_alloc_fn binningBuffer, void* binningUser, adgs_alloc_fn imageBuffer, void* imageUser,
	int P, int D, int M, int D_S, const float* background, int width, int height,
	const float* means3D, const float* shs, const float* colors_precomp, const float* flow_points, const float* semantic,
	const float* opacities, const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
	const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
	float* out_color, float* out_depth, float* img_opacity, float* img_flow, float* img_semantic,
	int inv_depth, int* radii, int debug, void* stream) {
	return raster_forward_impl(nullptr, true, geometryBuffer, geometryUser, binningBuffer, binningUser, imageBuffer, imageUser, P, D, M, D_S, background,
		width, height, means3D, shs, colors_precomp, flow_points, semantic, opacities, scales, scale_modifier, rotations, cov3D_precomp,
		viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, out_depth, img_opacity, img_flow, img_semantic,
		inv_depth, radii, debug, stream);
}

// The forward-only render: adgs_raster_forward's arguments and images, nothing kept for a backward (include/adgs_rasterizer.h)
extern "C" int adgs_raster_render(
	adgs_alloc_fn geometryBuffer, void* geometryUser, adgs_alloc_fn binningBuffer, void* binningUser, adgs_alloc_fn imageBuffer, void* imageUser,
	int P, int D, int M, int D_S, const float* background, int width, int height,
	const float* means3D, const float* shs, const float* colors_precomp, const float* flow_points, const float* semantic,
	const float* opacities, const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
	const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
	float* out_color, float* out_depth, float* img_opacity, float* img_flow, float* img_semantic,
	int inv_depth, int* radii, int debug, void* stream) {
	return raster_forward_impl(nullptr, false, geometryBuffer, geometryUser, binningBuffer, binningUser, imageBuffer, imageUser, P, D, M, D_S, background,
		width, height, means3D, shs, colors_precomp, flow_points, semantic, opacities, scales, scale_modifier, rotations, cov3D_precomp,
		viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, out_depth, img_opacity, img_flow, img_semantic,
		inv_depth, radii, debug, stream);
}

extern "C" int adgs_raster_backward(
	int P, int D, int M, int R, int D_S, const float* background, int width, int height,
	const float* means3D, const float* shs, const float* colors_precomp, const float* flow_points, const float* semantic,
	const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
	const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
	const int* radii, char* geom_buffer, char* binning_buffer, char* img_buffer,
	const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dpix_flow, const float* dL_dpix_semantic,
	float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D,
	float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dflow, float* dL_dsemantic,
	const float* grad_img_opacity, const float* img_opacity, int inv_depth, int debug, void* stream) {
	return raster_backward_impl(nullptr, nullptr, P, D, M, R, D_S, background, width, height, means3D, shs, colors_precomp, flow_points, semantic,
		scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer,
		img_buffer, dL_dpix, dL_dpix_depth, dL_dpix_flow, dL_dpix_semantic, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_ddepth, dL_dmean3D,
		dL_dcov3D, dL_dsh, dL_dscale, dL_drot, dL_dflow, dL_dsemantic, grad_img_opacity, img_opacity, inv_depth, debug, stream);
}

// scene_dc != NULL is what switches the kernels to the raw-SH source, and an empty side (a model without scene or without
// object Gaussians: its tensors have no storage) is never dereferenced: give it the other side's pointers.
static ShSource to_sh_source(const adgs_sh_source* s) {
	ShSource r;
	r.Ns = s->Ns; r.scene_dc = s->scene_dc; r.obj_dc = s->obj_dc; r.scene_rest = s->scene_rest; r.obj_rest = s->obj_rest;
	r.scene_sp = s->scene_deform; r.obj_sp = s->obj_deform; r.f = s->f;
	const bool raw_geo = s->Ns > 0 && s->scene_xyz && s->scene_scaling && s->scene_rotation && s->scene_opacity;
	r.scene_xyz = raw_geo ? s->scene_xyz : nullptr; r.scene_scaling = raw_geo ? s->scene_scaling : nullptr;
	r.scene_rotation = raw_geo ? s->scene_rotation : nullptr; r.scene_opacity = raw_geo ? s->scene_opacity : nullptr;
	r.bg_image = s->bg_image;
	if (!r.scene_dc) { r.scene_dc = r.obj_dc; r.scene_rest = r.obj_rest; }
	if (!r.obj_dc) { r.obj_dc = r.scene_dc; r.obj_rest = r.scene_rest; }
	return r;
}
static int check_sh_source(const adgs_sh_source* sh, int P, int M, const char* who) {
	const bool need_scene = sh && sh->Ns > 0, need_obj = sh && sh->Ns < P;
	if (sh && (sh->scene_xyz || sh->scene_scaling || sh->scene_rotation || sh->scene_opacity) &&
	    !(sh->scene_xyz && sh->scene_scaling && sh->scene_rotation && sh->scene_opacity)) {
		set_error(std::string(who) + ": raw scene geometry needs all four tensors"); return -1;
	}
	if (!sh || sh->Ns < 0 || sh->Ns > P || (need_scene && (!sh->scene_dc || (M > 1 && !sh->scene_rest))) || (need_obj && (!sh->obj_dc || (M > 1 && !sh->obj_rest)))) {
		set_error(std::string(who) + ": incomplete SH source"); return -1;
	}
	return 0;
}

extern "C" int adgs_raster_forward_rawsh(
	adgs_alloc_fn geometryBuffer, void* geometryUser, adgs_alloc_fn binningBuffer, void* binningUser, adgs_alloc_fn imageBuffer, void* imageUser,
	int P, int D, int M, int D_S, const float* background, int width, int height,
	const float* means3D, const adgs_sh_source* sh, const float* flow_points, const float* semantic,
	const float* opacities, const float* scales, float scale_modifier, const float* rotations,
	const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
	float* out_color, float* out_depth, float* img_opacity, float* img_flow, float* img_semantic,
	int inv_depth, int* radii, int debug, void* stream) {
	if (P > 0 && check_sh_source(sh, P, M, "adgs_raster_forward_rawsh") != 0) return -1;
	if (P <= 0) return 0;
	const ShSource src = to_sh_source(sh);
	return raster_forward_impl(&src, true, geometryBuffer, geometryUser, binningBuffer, binningUser, imageBuffer, imageUser, P, D, M, D_S, background,
		width, height, means3D, nullptr, nullptr, flow_points, semantic, opacities, scales, scale_modifier, rotations, nullptr,
		viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, 0, out_color, out_depth, img_opacity, img_flow, img_semantic,
		inv_depth, radii, debug, stream);
}

extern "C" int adgs_raster_render_rawsh(
	adgs_alloc_fn geometryBuffer, void* geometryUser, adgs_alloc_fn binningBuffer, void* binningUser, adgs_alloc_fn imageBuffer, void* imageUser,
	int P, int D, int M, int D_S, const float* background, int width, int height,
	const float* means3D, const adgs_sh_source* sh, const float* flow_points, const float* semantic,
	const float* opacities, const float* scales, float scale_modifier, const float* rotations,
	const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
	float* out_color, float* out_depth, float* img_opacity, float* img_flow, float* img_semantic,
	int inv_depth, int* radii, int debug, void* stream) {
	if (P > 0 && check_sh_source(sh, P, M, "adgs_raster_render_rawsh") != 0) return -1;
	if (P <= 0) return 0;
	const ShSource src = to_sh_source(sh);
	return raster_forward_impl(&src, false, geometryBuffer, geometryUser, binningBuffer, binningUser, imageBuffer, imageUser, P, D, M, D_S, background,
		width, height, means3D, nullptr, nullptr, flow_points, semantic, opacities, scales, scale_modifier, rotations, nullptr,
		viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, 0, out_color, out_depth, img_opacity, img_flow, img_semantic,
		inv_depth, radii, debug, stream);
}

extern "C" int adgs_raster_backward_rawsh(
	int P, int D, int M, int R, int D_S, const float* background, int width, int height,
	const float* means3D, const adgs_sh_source* sh, const float* flow_points, const float* semantic,
	const float* scales, float scale_modifier, const float* rotations,
	const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
	const int* radii, char* geom_buffer, char* binning_buffer, char* img_buffer,
	const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dpix_flow, const float* dL_dpix_semantic,
	float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D,
	float* dL_dcov3D, const adgs_sh_grads* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dflow, float* dL_dsemantic,
	const float* grad_img_opacity, const float* img_opacity, int inv_depth, int debug, void* stream) {
	if (!sh || !dL_dsh) { set_error("adgs_raster_backward_rawsh: NULL SH source / gradients"); return -1; }
	// the gradient block is versioned by its size: what the caller's header did not know yet is absent (NULL)
	adgs_sh_grads grads_full;
	memset(&grads_full, 0, sizeof(grads_full));
	if (dL_dsh->struct_bytes < offsetof(adgs_sh_grads, scene_xyz) || dL_dsh->struct_bytes > 4096) {
		set_error("adgs_raster_backward_rawsh: adgs_sh_grads.struct_bytes must be sizeof(adgs_sh_grads) of the caller's header"); return -1;
	}
	memcpy(&grads_full, dL_dsh, std::min<size_t>((size_t)dL_dsh->struct_bytes, sizeof(grads_full)));
	dL_dsh = &grads_full;
	if (P > 0 && check_sh_source(sh, P, M, "adgs_raster_backward_rawsh") != 0) return -1;
	const ShSource src = to_sh_source(sh);
	ShGradDst dst;
	dst.scene_dc = dL_dsh->scene_dc; dst.obj_dc = dL_dsh->obj_dc; dst.scene_rest = dL_dsh->scene_rest; dst.obj_rest = dL_dsh->obj_rest;
	dst.scene_sp = dL_dsh->scene_deform; dst.obj_sp = dL_dsh->obj_deform;
	dst.rgb_factor = dL_dsh->rgb_factor;
	dst.bg_image = dL_dsh->bg_image;
	dst.scene_xyz = dL_dsh->scene_xyz; dst.scene_scaling = dL_dsh->scene_scaling; dst.scene_rotation = dL_dsh->scene_rotation; dst.scene_opacity = dL_dsh->scene_opacity;
	if (src.scene_xyz && !(dst.scene_xyz && dst.scene_scaling && dst.scene_rotation && dst.scene_opacity)) {
		set_error("adgs_raster_backward_rawsh: the source carries raw scene geometry, its four gradient destinations are required"); return -1;
	}
	if (dL_dsh->adam && P > 0) {
		// the Adam step in place of the gradient stores (include/adgs_optim.h: adgs_sh_adam)
		const adgs_sh_adam& ad = *dL_dsh->adam;
		hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
		(void)hipStreamIsCapturing((hipStream_t)stream, &cs);
		if (cs != hipStreamCaptureStatusNone) { set_error("adgs_raster_backward_rawsh: the in-backward Adam step cannot be captured into a graph (its bias corrections are launch arguments)"); return -1; }
		if (!(ad.beta1 >= 0.f && ad.beta1 < 1.f && ad.beta2 >= 0.f && ad.beta2 < 1.f && ad.eps >= 0.f)) { set_error("adgs_raster_backward_rawsh: invalid Adam hyper-parameter"); return -1; }
		dst.adam.beta1 = ad.beta1; dst.adam.beta2 = ad.beta2; dst.adam.eps = ad.eps;
		const int No = P - src.Ns;
		struct { const adgs_adam_slot* in; AdamSlot* out; const float* source; float* grad_dst; int rows; const char* name; } slots[4] = {
			{ &ad.scene_rest, &dst.adam.scene_rest, src.scene_rest, dst.scene_rest, src.Ns, "scene_rest" },
			{ &ad.obj_rest, &dst.adam.obj_rest, src.obj_rest, dst.obj_rest, No, "obj_rest" },
			{ &ad.scene_deform, &dst.adam.scene_sp, src.scene_sp, dst.scene_sp, src.Ns, "scene_deform" },
			{ &ad.obj_deform, &dst.adam.obj_sp, src.obj_sp, dst.obj_sp, No, "obj_deform" } };
		for (auto& sl : slots) {
			if (!sl.in->param || sl.rows <= 0) continue;
			if (sl.in->param != sl.source) { set_error(std::string("adgs_raster_backward_rawsh: Adam slot ") + sl.name + " does not point at the tensor the frame reads"); return -1; }
			if (sl.grad_dst) { set_error(std::string("adgs_raster_backward_rawsh: ") + sl.name + " has a gradient destination AND an Adam slot"); return -1; }
			if (!sl.in->exp_avg || !sl.in->exp_avg_sq || sl.in->step < 1) { set_error(std::string("adgs_raster_backward_rawsh: Adam slot ") + sl.name + ": NULL moments or step < 1"); return -1; }
			sl.out->p = sl.in->param; sl.out->m = sl.in->exp_avg; sl.out->v = sl.in->exp_avg_sq;
			adam_bias_terms(sl.in->lr, sl.in->step, ad.beta1, ad.beta2, &sl.out->step_size, &sl.out->inv_bc2_sqrt);
		}
		// one side of a tensor pair fused and the other stored would leave the caller with half a gradient
		if ((dst.adam.scene_rest.p && dst.obj_rest) || (dst.adam.obj_rest.p && dst.scene_rest) || (dst.adam.scene_sp.p && dst.obj_sp) || (dst.adam.obj_sp.p && dst.scene_sp)) {
			set_error("adgs_raster_backward_rawsh: the scene and object halves of a tensor pair take the Adam step together or not at all"); return -1;
		}
		if ((dst.adam.scene_sp.p && !dst.scene_dc) || (dst.adam.obj_sp.p && !dst.obj_dc)) {
			set_error("adgs_raster_backward_rawsh: the Adam step of the SH deformation rows needs the dc gradient destinations (their rows are multiples of it)"); return -1;
		}
	}
	return raster_backward_impl(&src, &dst, P, D, M, R, D_S, background, width, height, means3D, nullptr, nullptr, flow_points, semantic,
		scales, scale_modifier, rotations, nullptr, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer,
		img_buffer, dL_dpix, dL_dpix_depth, dL_dpix_flow, dL_dpix_semantic, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_ddepth, dL_dmean3D,
		dL_dcov3D, nullptr, dL_dscale, dL_drot, dL_dflow, dL_dsemantic, grad_img_opacity, img_opacity, inv_depth, debug, stream);
}

extern "C" int adgs_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present, void* stream_) {
	(void)projmatrix;   // in_frustum only tests the view-space depth (auxiliary.h:154)
	if (P <= 0) return 0;
	if (launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream_) != 0) return -1;
	return 0;
}

extern "C" size_t adgs_knn_workspace_bytes(int P) { return knn_workspace_bytes(P); }

extern "C" int adgs_knn_dist2(int P, const float* points, float* meanDists, char* workspace, void* stream_) {
	if (P <= 0) return 0;
	if (!points || !meanDists || !workspace) { set_error("adgs_knn_dist2: NULL pointer"); return -1; }
	return knn_run(P, points, meanDists, workspace, (hipStream_t)stream_);
}

// ---- test-only hooks (include/adgs_testing.h) ----
#include "../../include/adgs_testing.h"
#include "../../include/adgs_optim.h"
// the image state of a v2 forward, carved the way that forward carved it
namespace {
struct V2ImageView { ImgStateV2 img; size_t wtiles, ncells; };
bool v2_image_view(const char* img_buffer, int width, int height, V2ImageView* v) {
	if (!img_buffer || width <= 0 || height <= 0) return false;
	const int gx = (width + TILE_X - 1) / TILE_X, gy = (height + TILE_Y - 1) / TILE_Y;
	const size_t ntiles = (size_t)gx * gy, npix = (size_t)width * height;
	FrameCfg cfg;
	if (!lookup_frame_by_image(img_buffer, width, height, &cfg)) { cfg.v2 = 1; cfg.cell_tiles = v2_cell_tiles(gx, gy, true, false); cfg.ppl = v2_pixels_per_lane(ntiles, false); }      // test hooks over foreign buffers: the defaults
	const int cell_tiles = cfg.cell_tiles, ppl = cfg.ppl;
	v->ncells = (size_t)((gx + cell_tiles - 1) / cell_tiles) * ((gy + cell_tiles - 1) / cell_tiles);
	v->wtiles = (size_t)gx * ((height + 4 * ppl - 1) / (4 * ppl));
	v->img = ImgStateV2::carve(const_cast<char*>(img_buffer), npix, v->wtiles, v->ncells, nullptr);
	return true;
}
long long sum_tile_words(const uint32_t* d_words, size_t n, hipStream_t stream) {
	std::vector<uint32_t> h(n);
	if (hipMemcpyAsync(h.data(), d_words, n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream) != hipSuccess) return -1;
	if (hipStreamSynchronize(stream) != hipSuccess) return -1;
	long long total = 0;
	for (uint32_t x : h) total += x;
	return total;
}
} // namespace
extern "C" long long adgs_test_v2_published_entries(const char* img_buffer, int width, int height, void* stream_) {
	V2ImageView v;
	if (!v2_image_view(img_buffer, width, height, &v)) return -1;
	return sum_tile_words(v.img.tile_consumed, v.wtiles, (hipStream_t)stream_);
}
// per-tile counters of the last forward: out_consumed / out_scanned receive one uint32 per wave tile (returns the tile count)
extern "C" long long adgs_test_v2_tile_counters(const char* img_buffer, int width, int height, uint32_t* out_consumed, uint32_t* out_scanned, long long capacity, void* stream_) {
	V2ImageView v;
	if (!v2_image_view(img_buffer, width, height, &v)) return -1;
	if ((long long)v.wtiles > capacity) return (long long)v.wtiles;
	if (out_consumed && hipMemcpyAsync(out_consumed, v.img.tile_consumed, v.wtiles * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream_) != hipSuccess) return -1;
	if (out_scanned && hipMemcpyAsync(out_scanned, v.img.tile_scanned, v.wtiles * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream_) != hipSuccess) return -1;
	if (hipStreamSynchronize((hipStream_t)stream_) != hipSuccess) return -1;
	return (long long)v.wtiles;
}
// experiment builds (-DADGS_TIMELINE): the raw tile_scanned / tile_batches words of an image state
extern "C" long long adgs_test_v2_tile_words(const char* img_buffer, int width, int height, uint32_t* out_scanned, uint32_t* out_batches, long long capacity, void* stream_) {
	V2ImageView v;
	if (!v2_image_view(img_buffer, width, height, &v)) return -1;
	if ((long long)v.wtiles > capacity) return (long long)v.wtiles;
	if (out_scanned && hipMemcpyAsync(out_scanned, v.img.tile_scanned, v.wtiles * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream_) != hipSuccess) return -1;
	if (out_batches && hipMemcpyAsync(out_batches, v.img.tile_batches, v.wtiles * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream_) != hipSuccess) return -1;
	if (hipStreamSynchronize((hipStream_t)stream_) != hipSuccess) return -1;
	return (long long)v.wtiles;
}
// number of batches (of <= 64 entries that passed the tile test) the forward handed to its blend loop, summed over the tiles
extern "C" long long adgs_test_v2_blend_batches(const char* img_buffer, int width, int height, void* stream_) {
	V2ImageView v;
	if (!v2_image_view(img_buffer, width, height, &v)) return -1;
	return sum_tile_words(v.img.tile_batches, v.wtiles, (hipStream_t)stream_);
}
// per-cell (start, end) ranges of the depth-sorted candidate lists of the last forward (returns the cell count)
extern "C" long long adgs_test_v2_cell_ranges(const char* img_buffer, int width, int height, uint32_t* out_ranges, long long capacity, void* stream_) {
	V2ImageView v;
	if (!v2_image_view(img_buffer, width, height, &v)) return -1;
	if ((long long)v.ncells > capacity) return (long long)v.ncells;
	if (out_ranges && hipMemcpyAsync(out_ranges, v.img.cell_ranges, v.ncells * sizeof(uint2), hipMemcpyDeviceToHost, (hipStream_t)stream_) != hipSuccess) return -1;
	if (hipStreamSynchronize((hipStream_t)stream_) != hipSuccess) return -1;
	return (long long)v.ncells;
}
// sum over the wave tiles of the candidates of the cell list each tile's walk went through before all its pixels were saturated
extern "C" long long adgs_test_v2_scanned_candidates(const char* img_buffer, int width, int height, void* stream_) {
	V2ImageView v;
	if (!v2_image_view(img_buffer, width, height, &v)) return -1;
	return sum_tile_words(v.img.tile_scanned, v.wtiles, (hipStream_t)stream_);
}
// sizeof of the structs that cross the ABI by pointer: lets a binding check its mirror (which: 0 adgs_sh_source, 1 adgs_sh_grads,
// 2 adgs_frame_stats, 3 adgs_frame_status, 4 adgs_func_eval, 5 adgs_adam_group, 6 adgs_sh_adam)
extern "C" unsigned long long adgs_test_env_reads(void) { return g_env_reads.load(); }
extern "C" size_t adgs_test_abi_sizeof(int which) {
	switch (which) {
	case 0: return sizeof(adgs_sh_source);
	case 1: return sizeof(adgs_sh_grads);
	case 2: return sizeof(adgs_frame_stats);
	case 3: return sizeof(adgs_frame_status);
	case 4: return sizeof(adgs_func_eval);
	case 5: return sizeof(adgs_adam_group);
	case 6: return sizeof(adgs_sh_adam);
	default: return 0;
	}
}
extern "C" void adgs_test_set_capacity_hints(long long pairs, long long fine_pairs) {
	FrameContext* fc = frame_context();
	fc->hint_cells = (size_t)std::max(0ll, pairs); fc->hint_fine = (size_t)std::max(0ll, fine_pairs); fc->hint_max_cell_chunks = 0u; fc->bucket_frame_pending = false;
	// ... and what the binning has learned about depths: the thread's slab bounds back to "everything in slab 0", the cameras' own tables forgotten
	(void)hipDeviceSynchronize();
	if (fc->slab_bounds) (void)hipMemset(fc->slab_bounds, 0xff, (size_t)MAX_CELLS * SLAB_ROW * sizeof(uint32_t));
	if (OrderHints* oh = order_hints()) for (auto& e : oh->entries) { e.written = false; if (e.buf) (void)oh->reset_signature(e.buf, e.tiles, e.extra); }
}
// fills the calling thread's table of depth-slab bounds with pseudo-random words (an LCG): ANY contents must give the same lists
// (binning.hip: the slab of a depth is a monotone function of it whatever the row holds) -- returns 0, or -1 when no bucket-binned frame
// has created the table yet
extern "C" int adgs_test_scramble_slab_bounds(unsigned seed) {
	FrameContext* fc = frame_context();
	if (!fc->slab_bounds) return -1;
	std::vector<uint32_t> h((size_t)MAX_CELLS * SLAB_ROW);
	uint32_t x = seed * 2654435761u + 12345u;
	for (auto& w : h) { x = x * 1664525u + 1013904223u; w = (x >> 3) ^ (x << 7); }
	if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(fc->slab_bounds, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) return -1;
	if (OrderHints* oh = order_hints()) for (auto& e : oh->entries) e.written = false;      // the cameras' own tables are not consulted
	return 0;
}
extern "C" size_t adgs_test_scan_temp_bytes(size_t n) { return scan_temp_bytes(n); }
extern "C" int adgs_test_exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, char* temp, void* stream) {
	return exclusive_scan_u32(in, out, n, temp, (hipStream_t)stream);
}
extern "C" size_t adgs_test_sort_temp_bytes(size_t n) { return sort_temp_bytes(n); }
extern "C" int adgs_test_sort_pairs_u64(uint64_t* ki, uint64_t* ko, uint32_t* vi, uint32_t* vo, size_t n, int end_bit, char* temp, void* stream) {
	return radix_sort_pairs_u64(ki, ko, vi, vo, n, end_bit, temp, (hipStream_t)stream);
}
extern "C" int adgs_test_sort_pairs_u32(uint32_t* ki, uint32_t* ko, uint32_t* vi, uint32_t* vo, size_t n, int end_bit, char* temp, void* stream) {
	return radix_sort_pairs_u32(ki, ko, vi, vo, n, end_bit, temp, (hipStream_t)stream);
}
