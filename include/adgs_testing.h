/*
 * adgs_testing.h -- test-only entry points of libadgs_hip.so (device primitives that
 * the rasterizer and kNN paths are built from).  Not part of the drop-in boundary;
 * used by tests/ to check the hand-written scan / radix sort directly.
 */
#ifndef ADGS_TESTING_H
#define ADGS_TESTING_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
size_t adgs_test_scan_temp_bytes(size_t n);
/* out[i] = sum_{j<i} in[j]; in == out allowed */
int adgs_test_exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, char* temp, void* stream);
size_t adgs_test_sort_temp_bytes(size_t n);
/* stable LSD radix sort on key bits [0,end_bit); *_in are clobbered */
int adgs_test_sort_pairs_u64(uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, size_t n, int end_bit, char* temp, void* stream);
int adgs_test_sort_pairs_u32(uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, size_t n, int end_bit, char* temp, void* stream);
/* Number of (tile, Gaussian) entries the default (v2) forward published for the backward replay of a frame: the sum of the
 * per-tile counts in the forward's image-state buffer.  Synchronises the stream; statistics only (bench.py). */
long long adgs_test_v2_published_entries(const char* img_buffer, int width, int height, void* stream);
/* Sum over the tiles of the candidates of the coarse cell's depth-sorted list each tile's walk went through before all of
 * its pixels were saturated (same buffer, same conditions). */
long long adgs_test_v2_scanned_candidates(const char* img_buffer, int width, int height, void* stream);
/* The two per-tile counters themselves (host arrays of `capacity` uint32 each, either may be NULL); returns the number of tiles. */
long long adgs_test_v2_tile_counters(const char* img_buffer, int width, int height, uint32_t* out_consumed, uint32_t* out_scanned, long long capacity, void* stream);
/* Batches of <= 64 tile-test survivors the forward handed to its blend loop, summed over the tiles (same buffer, same conditions). */
long long adgs_test_v2_blend_batches(const char* img_buffer, int width, int height, void* stream);
/* (start, end) of every coarse cell's depth-sorted candidate list (host array of 2 x `capacity` uint32); returns the number of cells. */
long long adgs_test_v2_cell_ranges(const char* img_buffer, int width, int height, uint32_t* out_ranges, long long capacity, void* stream);

/* sizeof of the structs that cross the ABI by pointer (0 adgs_sh_source, 1 adgs_sh_grads, 2 adgs_frame_stats, 3 adgs_frame_status,
 * 4 adgs_func_eval, 5 adgs_adam_group, 6 adgs_sh_adam): a binding checks its mirror against it. */
size_t adgs_test_abi_sizeof(int which);

/* How many times the rasterizer has read an ADGS_* environment switch so far (process-wide).  The backward of a frame must not read any:
 * what the forward decided from the environment travels with the frame (api.hip: FrameCfg); a test reads this before and after. */
unsigned long long adgs_test_env_reads(void);

/* Overwrites the process-wide capacity hints the next default-pipeline forward is enqueued against (pairs, fine pairs; the
 * forward still uses at least P + 4096 / 8 P + 4096): lets a test force the "frame does not fit its capacity" path. */
void adgs_test_set_capacity_hints(long long pairs, long long fine_pairs);      /* also forgets the depth-slab bounds the bucket binning has learned */

/* Fills the calling thread's table of depth-slab bounds (bucket binning) with pseudo-random words: any contents must give the same
 * per-cell lists.  Returns 0, -1 when no bucket-binned frame has created the table yet. */
int adgs_test_scramble_slab_bounds(unsigned seed);

#ifdef __cplusplus
}
#endif
#endif
