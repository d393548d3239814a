/*
 * psascan_amd_extras.h -- synthetic-input preparation (bench / full-size property tests).
 * NOT part of the drop-in boundary (see psascan_amd.h for that): the reference sorts
 * half-blocks on the host (inmem_psascan_src/inmem_psascan.hpp:64-304) and construct_sa does
 * the same.  These helpers exist so that bench.py can build valid multi-GiB hot-path inputs
 * (text, partial SA, BWT, i0, gt_begin) inside HBM within minutes.
 */
#ifndef PSASCAN_AMD_EXTRAS_H
#define PSASCAN_AMD_EXTRAS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* seeded text: mode 0 = uniform bytes 0..254, 1 = DNA (ACGT), 2 = `sigma` letters from 'a',
 * 3 = English-like (Zipfian words over a skewed 26-letter alphabet, blanks, full stops: sigma = 28) */
int psgx_gen_text(uint8_t *d_text, int64_t n, int mode, int sigma, uint64_t seed);
/* partial SA (relative to beg), BWT (dummy 0 at i0), i0 and gt_begin (bit u <-> position
 * end-u, u in [0,size)) of text[beg..end) ordered as suffixes of the whole text.  Prefix-key
 * radix sort + refinement rounds on the following symbols (groups of equal prefixes are re-sorted by
 * the next 8-12 symbols, up to 24 rounds) + comparison of what is left: for texts whose repeats are
 * at most a few hundred symbols long (random, DNA, the English-like generator).              */
int psgx_sort_halfblock(const uint8_t *d_text, int64_t n, int64_t beg, int64_t end, uint32_t *d_psa,
                        uint8_t *d_bwt, int64_t *i0, uint32_t *d_gt_begin, int64_t *tie_groups);
/* the same when only the window text[text_begin .. n) is on the device (d_text points at position 0 all the same): a rank
 * of the block-per-GPU schedule holds its own block and a look-ahead, `n` is where its window ends                    */
int psgx_sort_halfblock_window(const uint8_t *d_text, int64_t text_begin, int64_t n, int64_t beg, int64_t end, uint32_t *d_psa,
                               uint8_t *d_bwt, int64_t *i0, uint32_t *d_gt_begin, int64_t *tie_groups);
/* property check of `count` uint40 entries: sum of entries mod 2^64, and the number of sampled
 * adjacent pairs (k, k+1) that are NOT in suffix order.                                      */
int psgx_check_sa5(const uint8_t *d_text, int64_t n, const uint8_t *d_sa5, int64_t count, int64_t samples,
                   uint64_t seed, int64_t *bad_pairs, uint64_t *sum);
/* the same; *undecided_pairs (may be NULL) = sampled pairs that agree on 2^24 symbols and were not followed further */
int psgx_check_sa5_ex(const uint8_t *d_text, int64_t n, const uint8_t *d_sa5, int64_t count, int64_t samples,
                      uint64_t seed, int64_t *bad_pairs, uint64_t *sum, int64_t *undecided_pairs);
/* binds all threads of the process to the CPUs of the current device's NUMA node (a placement hint for the pinned
   buffers and the .sa5 writer; never fails).  *node = the node, or -1 if nothing was changed.  PSG_NO_NUMA_BIND=1 disables. */
int psgx_bind_threads_near_device(int *node);
/* what the device-memory arena has cost so far: seconds inside hipMalloc for its segments, how many, their bytes */
int psgx_arena_stats(double *driver_seconds, int64_t *segments, int64_t *bytes);
/* gap[v] += #{k : log[k] == v}, v in [0,m]; entries 0xFFFFFFFF are ignored; the log is clobbered.
 * (the atomics-free gap update of psg_stream_gap, exposed for tests)                          */
int psgx_gap_hist(uint32_t *d_log, int64_t nlog, int64_t m, uint32_t *d_gap);
#ifdef __cplusplus
}
#endif
#endif
