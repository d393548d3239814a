/*
 * psascan_amd.h -- C ABI of the MI355X-native streaming-gap + merge path of pSAscan.
 *
 * This is the drop-in boundary.  The reference has no FFI layer: its
 * `process_block` (include/partial_sufsort.hpp:67-551) and `pSAscan`
 * (include/psascan.hpp:53-131) call a handful of C++ templates with raw pointers.
 * Each entry point below replaces one of those call sites; the cited file:line is
 * relative to the reference checkout.
 *
 * Conventions
 *   - plain pointers and sizes only; all sizes/positions are int64_t.
 *   - `d_` pointers are device (HBM) pointers, `h_` pointers are host pointers.
 *   - every function returns 0 on success, a negative PSG_E* code otherwise, never
 *     exits or throws; psg_last_error() describes the last failure of the calling thread.
 *   - bit arrays are little-endian arrays of uint32_t, LSB-first (bit i = word i/32,
 *     bit i%32) -- byte-identical to the reference's bitvector (bitvector.hpp:61-67).
 *   - a "gt" bit array that belongs to the position range (lo, hi] stores the bit of text
 *     position j at index u = hi - j.  This is the reference's reversed indexing (bit n-j
 *     of the multifile, compute_gap.hpp:118-119, stream.hpp:104-106) shifted by n-hi.
 *   - ONE host thread per process drives ONE device (the library keeps its stream, device arena and
 *     staging buffers in process-wide state; the multi-GPU driver runs one process per GPU); work is
 *     enqueued on the library's stream and the call returns after the result is complete unless
 *     stated otherwise.  psg_last_error() is per thread.
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails.
 */
#ifndef PSASCAN_AMD_H
#define PSASCAN_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSG_OK 0
#define PSG_EINVAL (-1)   /* bad argument                                  */
#define PSG_EDEVICE (-2)  /* HIP runtime error / no device                 */
#define PSG_ENOMEM (-3)   /* device allocation failed                      */
#define PSG_EOVERFLOW (-4)/* (unused since the gap arrays carry an excess list) */
#define PSG_ECHECK (-5)   /* an internal invariant check failed            */

/* ---- runtime ------------------------------------------------------------------------ */
int psg_init(int device);                 /* select device, create the stream            */
const char *psg_last_error(void);
int psg_device_name(char *buf, int cap);
int psg_malloc(void **d_ptr, int64_t bytes);
int psg_free(void *d_ptr);                /* returns the block to the library's cache       */
int psg_trim(void);                       /* gives all cached device memory back to the driver */
int psg_memset(void *d_ptr, int value, int64_t bytes);
int psg_h2d(void *d_dst, const void *h_src, int64_t bytes);
int psg_d2h(void *h_dst, const void *d_src, int64_t bytes);
int psg_d2d(void *d_dst, const void *d_src, int64_t bytes);
int psg_sync(void);
/* page-locked host memory: psg_h2d / psg_d2h / psg_merge_stream copy from / to it without a staging step */
int psg_host_alloc(void **h_ptr, int64_t bytes);
int psg_host_free(void *h_ptr);
/* download in the background: a worker thread with its own stream and pinned staging drains d_src into (pageable)
 * host memory while the library's stream goes on; h_dst stays untouched until psg_copy_wait (which returns the
 * copy's status and frees the handle).  free_src != 0: d_src (from psg_malloc) is handed to the library and
 * freed the moment it is drained; otherwise it stays the caller's and must live until psg_copy_wait.  The work
 * enqueued before the call is complete when it returns.                                                       */
typedef struct psg_copy psg_copy_t;
int psg_d2h_begin(void *h_dst, void *d_src, int64_t bytes, int free_src, psg_copy_t **out);
/* the other direction: h_src (pageable) goes up into d_dst in the background; d_dst must not be used by the
 * library's stream until psg_copy_wait (the tail chunks of a text that stays in host memory: stream.hpp:104-106) */
int psg_h2d_begin(void *d_dst, const void *h_src, int64_t bytes, psg_copy_t **out);
int psg_copy_wait(psg_copy_t *copy);
/* device memory: bytes handed out by psg_malloc right now / the highest value so far / held from the driver */
int psg_mem_stats(int64_t *in_use, int64_t *peak_in_use, int64_t *reserved);
/* the device's memory as the driver sees it (hipMemGetInfo) */
int psg_device_memory(int64_t *free_bytes, int64_t *total_bytes);
/* a budget below the device's memory: psg_malloc and every internal allocation fail with PSG_ENOMEM once more than
 * `bytes` would be handed out (0 = no budget); psg_device_memory then reports the budget.  The reference runs under a
 * RAM budget the same way (-m, psascan.hpp:73-91); construct_sa --hbm-limit uses this to run its spill paths (partial
 * SAs, merge bitvectors and gt bits in host memory, the text uploaded chunk by chunk) on a device that would hold all. */
int psg_set_memory_limit(int64_t bytes);
/* Use an externally owned hipStream_t (e.g. torch's current stream); NULL = own stream. */
int psg_set_stream(void *hip_stream);

/* ---- rank over a block BWT: replaces `new rank4n<>(bwt, m, threads)`,
 *      partial_sufsort.hpp:403,500 ; semantics rank.hpp:566-568, m_count rank.hpp:112 --- */
/* ---- gap arrays: buffered_gap_array (gap_array.hpp:55-383).  A gap array over m+1 slots is psg_gap_words(m)
 *      uint32 words: words [0, m] are the counters; behind them (16-byte aligned) an EXCESS LIST: a slot whose
 *      counter wraps gets an entry appended by an atomic cursor and keeps counting from 0, exactly as the
 *      reference's u8 counters do with their excess list (update.hpp:88-96, gap_array.hpp:79-88):
 *          value(j) = counter[j] + 2^32 * #{entries equal to j}           (gap_array.hpp:116-124)
 *      Every consumer (psg_gap_to_bitvector, psg_split_gap, psg_gap_values, ...) works on value(j).  An all-zero
 *      array is a valid empty gap array; PSG_GAP_UNINITIALIZED passes initialise it themselves.  n <= 2^40 bounds
 *      the list at 256 entries; it has room for PSG_GAP_EXCESS_CAP.  (Tests shrink the counters to 8 or 16 bits
 *      with PSG_GAP_COUNTER_BITS so that the excess path runs at small sizes.)                                  */
#define PSG_GAP_EXCESS_CAP 65536
#define PSG_GAP_HDR_WORD(m) ((((int64_t)(m) + 1) + 3) & ~(int64_t)3)
#define PSG_GAP_WORDS(m) (PSG_GAP_HDR_WORD(m) + 4 + 2 * (int64_t)PSG_GAP_EXCESS_CAP)
int64_t psg_gap_words(int64_t m);
/* d_out[j] = value(j) for j in [0, m] (counters + excess), as 64-bit values */
int psg_gap_values(const uint32_t *d_gap, int64_t m, uint64_t *d_out);

typedef struct psg_rank psg_rank_t;
/* data_bytes_per_block: 0 = choose from the alphabet (64, or 48 when sigma <= 4).        */
int psg_rank_build(const uint8_t *d_bwt, int64_t m, int data_bytes_per_block, psg_rank_t **out);
int psg_rank_counts(const psg_rank_t *r, int64_t counts[256]);
int64_t psg_rank_device_bytes(const psg_rank_t *r);
/* batch query, used by the parity tests: out[k] = rank(i[k], c[k])                      */
int psg_rank_query(const psg_rank_t *r, const int64_t *d_i, const uint8_t *d_c, int64_t nq, int64_t *d_out);
void psg_rank_free(psg_rank_t *r);

/* ---- one streaming pass: replaces compute_gap<T>(...), partial_sufsort.hpp:412-414 and
 *      :512-514 (compute_gap.hpp:61-157 -> stream.hpp:147-158 + update.hpp:86-96) -------
 *  d_tail           text[tail_begin .. tail_end)               (tail_len bytes)
 *  d_gt_in          gt of positions (tail_begin, tail_end] w.r.t. the END of the block the
 *                   rank was built on; bit u <-> position tail_end - u; NULL = all zero
 *  rank_at_tail_end number of block suffixes smaller than text[tail_end..n)
 *                   (the reference's initial_ranks.back(), stream.hpp:66,108)
 *  d_gap            gap array of psg_gap_words(m) words, INCREMENTED (zero all of it first for a fresh array);
 *                   value semantics of buffered_gap_array (gap_array.hpp:116-124)
 *  d_gt_out         tail_len bits written: bit u = [text[tail_end-u..n) > text[block_beg..n)]
 *                   (stream.hpp:150); may be NULL
 *  max_chains       0 = auto (fill the chip); the tail is cut into that many independent
 *                   backward-search chains whose start ranks are found on the device
 *  h_final_rank     out: rank of text[tail_begin..n) among the block suffixes; may be NULL
 */
typedef struct {
  int64_t n_chains;        /* chains actually used                                  */
  int64_t chain_len;       /* steps per chain                                       */
  int64_t warmup_steps;    /* warm-up steps per chain boundary (last attempt)       */
  int64_t unresolved;      /* chain starts the warm-up left open (found by search, or run one after another) */
  int64_t rounds;          /* stream kernel launches                                */
  double kernel_ms;        /* time of the stream kernel launches (HIP events)       */
  double total_ms;         /* whole call                                            */
  double hist_ms;          /* rank-log sort + histogram (0 when atomics are used)   */
} psg_stream_stats;

int psg_stream_gap(const psg_rank_t *rank, int64_t block_i0, int block_last_symbol,
                   const uint8_t *d_tail, int64_t tail_len, const uint32_t *d_gt_in,
                   int64_t rank_at_tail_end, uint32_t *d_gap, uint32_t *d_gt_out,
                   int64_t max_chains, int64_t *h_final_rank, psg_stream_stats *stats);

/* Same pass over a sub-range of a tail (the reference cuts the tail into per-thread ranges the
 * same way, compute_gap.hpp:68-69,114-124): `right_context` more text bytes / gt bits are valid
 * to the right of the range.  d_tail = text[tail_begin .. tail_end + right_context), gt_in bit u
 * <-> position (tail_end + right_context) - u, and rank_at_context_end is the exact rank at
 * tail_end + right_context, or -1 if it is not known.  The start rank at tail_end is found on the device inside the context
 * (PSG_ECHECK if the text is too repetitive for that).  gt_out bit u <-> position tail_end - u.
 * right_context must be a multiple of 64.                                                    */
int psg_stream_gap_ctx(const psg_rank_t *rank, int64_t block_i0, int block_last_symbol,
                       const uint8_t *d_tail, int64_t tail_len, int64_t right_context,
                       const uint32_t *d_gt_in, int64_t rank_at_context_end, uint32_t *d_gap,
                       uint32_t *d_gt_out, int64_t max_chains, int64_t *h_final_rank,
                       psg_stream_stats *stats);

/* The reference hands compute_gap a freshly constructed (zeroed) buffered_gap_array
 * (partial_sufsort.hpp:405-407, 503-505).  Same pass as psg_stream_gap_ctx with
 * flags & PSG_GAP_UNINITIALIZED: d_gap may hold anything on entry and holds exactly this pass's
 * counts on return -- the library zero-fills or overwrites as suits the update mode (saves the
 * caller's memset and the read half of the read-modify-write of every counter).              */
#define PSG_GAP_UNINITIALIZED 1
int psg_stream_gap_ex(const psg_rank_t *rank, int64_t block_i0, int block_last_symbol,
                      const uint8_t *d_tail, int64_t tail_len, int64_t right_context,
                      const uint32_t *d_gt_in, int64_t rank_at_context_end, uint32_t *d_gap,
                      uint32_t *d_gt_out, int64_t max_chains, int flags, int64_t *h_final_rank,
                      psg_stream_stats *stats);

/* ---- em_compute_initial_ranks, partial_sufsort.hpp:189,311,385 (em_compute_initial_ranks.hpp:222-319,
 *      513-561; lcp_compare :54-76): exact rank of a tail suffix among the suffixes of a block, by string search
 *      over the block's partial suffix array(s).  A comparison that reaches `cmp_end` (the end of the block being
 *      processed) is decided by the gt bit of the position the pattern has reached.  The whole text is one device
 *      array.  A streaming pass given a search context resolves every chain start whose warm-up interval does not
 *      close (text with long repeats) this way, in one launch, instead of running those chains one after another. */
typedef struct {
  const uint8_t *d_text;          /* text[0..n) on the device                                               */
  int64_t n;
  int64_t cmp_end;                /* comparisons run on text symbols while the block suffix is below cmp_end */
  const uint32_t *d_gt_cmp_end;   /* bit (n - j) = [text[j..n) > text[cmp_end..n)], j in (cmp_end, n]; may be
                                     NULL when cmp_end == n                                                  */
  int nparts;                     /* 1 or 2 sorted parts (half-blocks), all below cmp_end                    */
  struct { int64_t beg, size; const uint32_t *d_psa_lo; const uint8_t *d_psa_hi; } part[2];
  int64_t text_begin, text_end;   /* 0, 0: all of text[0..n) is on the device.  Otherwise only text[text_begin ..
                                     text_end) is (d_text still points at position 0: d_text = window - text_begin);
                                     a comparison that would read outside fails the call with PSG_EWINDOW -- a text
                                     that stays in host memory is searched through a window per half-block          */
  const uint8_t *d_text2;         /* optional SECOND window for the searched positions (psg_initial_ranks only): they and */
  int64_t text2_begin, text2_end; /* what follows them are read from text[text2_begin .. text2_end) (d_text2 points at
                                     position 0 likewise), the parts' suffixes from the first window -- a rank of the
                                     block-per-GPU schedule holds its own block and a piece of the far block whose end it
                                     searches for, not the text in between.  NULL, 0, 0: one window.                */
} psg_search_ctx;
#define PSG_EWINDOW (-7)
/* h_ranks[k] = sum over the parts of #{s in part : text[s..n) < text[h_positions[k]..n)}; positions lie at or
 * behind the end of the last part (position n: rank 0).                                                       */
int psg_initial_ranks(const psg_search_ctx *sc, const int64_t *h_positions, int64_t count, int64_t *h_ranks);

/* One streaming pass, all arguments in a struct (psg_stream_gap_ex + search context).  With `search` set, chain
 * starts the warm-up cannot determine are found by psg_initial_ranks' search: `tail_begin_abs` = text position of
 * d_tail[0].  flags: PSG_GAP_UNINITIALIZED, PSG_FAIL_IF_UNRESOLVED (return PSG_EUNRESOLVED instead of running
 * unresolved chains one after another when no search context is given: the caller uploads the partial SAs and
 * calls again).                                                                                               */
#define PSG_FAIL_IF_UNRESOLVED 2
/* with `search`: find EVERY chain start of a small pass (at most 2^17 chains) by string search and skip the warm-up --
 * for callers that know the text has no long repeats around here (construct_sa's leaf merging: the leaves were sorted
 * with a bounded look-ahead); on periodic text every search costs a block length of comparisons                    */
#define PSG_SEARCH_ALL_STARTS 4
#define PSG_EUNRESOLVED (-6)
typedef struct {
  const psg_rank_t *rank;
  int64_t block_i0;
  int block_last_symbol;
  const uint8_t *d_tail;
  int64_t tail_len, right_context;
  const uint32_t *d_gt_in;
  int64_t rank_at_context_end;
  uint32_t *d_gap;
  uint32_t *d_gt_out;
  int64_t max_chains;
  int flags;
  const psg_search_ctx *search;   /* may be NULL */
  int64_t tail_begin_abs;         /* used with `search` */
} psg_stream_args;
int psg_stream_gap_args(const psg_stream_args *a, int64_t *h_final_rank, psg_stream_stats *stats);

/* ---- buffered_gap_array::convert_to_bitvector, partial_sufsort.hpp:441
 *      (gap_array.hpp:273-364): for j=0..m: gap[j] ones, then a zero (none after j=m).
 *      d_bv needs room for m + sum(gap) bits rounded up to 32; *nbits = m + sum(gap). ---- */
int psg_gap_to_bitvector(const uint32_t *d_gap, int64_t m, uint32_t *d_bv, int64_t bv_capacity_bits,
                         int64_t *nbits);

/* ---- merge_bwt, partial_sufsort.hpp:470-471 (bwt_merge.hpp:66-140) --------------------- */
int psg_merge_bwt(const uint8_t *d_left_bwt, const uint8_t *d_right_bwt, int64_t ml, int64_t mr,
                  int64_t left_i0, int64_t right_i0, int left_last_symbol, const uint32_t *d_bv,
                  uint8_t *d_out_bwt, int64_t *block_i0);

/* ---- compute_right_gap + compute_left_gap, partial_sufsort.hpp:541-542
 *      (compute_right_gap.hpp:161-305, compute_left_gap.hpp:162-306, gap_array_2n
 *      gap_array.hpp:386-529).  The half-block gaps are produced in their unary
 *      ("merge bitvector") form, which is what psg_merge consumes:
 *        mbv = for r = 0..size: gap[r] ones, then (r < size) a zero.
 *      d_mbv_left  : ml + mr + tail_len bits,  d_mbv_right : mr + tail_len bits.
 *      tail_len must equal sum(d_block_gap) (checked).                               ---- */
int psg_split_gap(const uint32_t *d_block_gap, const uint32_t *d_bv, int64_t ml, int64_t mr,
                  int64_t tail_len, uint32_t *d_mbv_left, uint32_t *d_mbv_right);

/* gap values (what the reference writes to its .gap files) out of a merge bitvector.      */
int psg_mbv_to_gap(const uint32_t *d_mbv, int64_t nbits, int64_t size, uint64_t *d_gap_out /* size+1 */);
/* vbyte codec of the gap files (utils/parallel_utils.hpp:47-136;
 * io/async_vbyte_stream_reader.hpp:49-186): replaces save_to_file, partial_sufsort.hpp:422 */
int psg_vbyte_encode(const uint64_t *d_vals, int64_t count, uint8_t *d_out, int64_t capacity, int64_t *nbytes);

/* ---- merge<T>, psascan.hpp:120,124 (merge.hpp:55-180) ---------------------------------
 *  Half-blocks sorted by beg.  psa = positions relative to beg, low 32 bits (+ optional
 *  high byte array for half-blocks larger than 2^32).  d_mbv = NULL for the last one.
 *  Output: entries [out_begin, out_begin+out_count) of the suffix array as 40-bit
 *  little-endian integers (types/uint40.hpp:42-104), 5*out_count bytes at d_out_sa5.      */
typedef struct {
  int64_t beg, size;
  const uint32_t *d_psa_lo;
  const uint8_t *d_psa_hi;   /* may be NULL */
  const uint32_t *d_mbv;     /* size + (sizes of all later half-blocks) bits; NULL for last */
} psg_hb_desc;
typedef struct psg_merge_plan psg_merge_plan_t;
int psg_merge_plan_create(const psg_hb_desc *hbs, int H, psg_merge_plan_t **out);
int psg_merge_run(const psg_merge_plan_t *plan, int64_t out_begin, int64_t out_count, uint8_t *d_out_sa5);
void psg_merge_plan_free(psg_merge_plan_t *plan);

/* ---- in-memory pSAscan pieces (inmem_psascan_src/inmem_psascan.hpp:64-304): a range too large for one host sort
 *      is cut into sub-ranges that ARE sorted on the host; their partial SAs are merged on the device exactly like
 *      the blocks of the text (one streaming pass per sub-range over the sub-ranges to its right, gap ->
 *      bitvector, psg_merge_plan_create), which yields the range's own partial SA:                          */
/* the merged order of a plan as u32 values: every level's beg + psa value must be below 2^32, i.e. the plan's
 * `beg` are relative to the enclosing range                                                                  */
int psg_merge_run_u32(const psg_merge_plan_t *plan, int64_t out_begin, int64_t out_count, uint32_t *d_out);
/* BWT (dummy 0 at i0, inmem_bwt_from_sa.hpp:47-83), i0 and gt_begin (bit u <-> position beg + size - u: the suffix
 * is ranked after the range's first suffix; bit 0 by comparison) of text[beg .. beg + size) from its partial SA.
 * sc: text + comparison end + gt bits as for psg_initial_ranks (parts unused); d_gt_begin may be NULL.        */
int psg_halfblock_from_psa(const psg_search_ctx *sc, int64_t beg, int64_t size, const uint32_t *d_psa, uint8_t *d_bwt,
                           int64_t *i0, uint32_t *d_gt_begin);
/* the two above for ranges of 2^32 positions or more (the 8 GiB half-blocks of BASELINE configs[3]'s 16 GiB blocks):
 * values of up to 40 bits in two planes, d_lo[k] = low 32 bits, d_hi[k] = bits 32..39 -- the layout psg_hb_desc
 * and the search parts take.  d_psa_hi may be NULL when size < 2^32.                                            */
int psg_merge_run_planes(const psg_merge_plan_t *plan, int64_t out_begin, int64_t out_count, uint32_t *d_lo, uint8_t *d_hi);
int psg_halfblock_from_psa40(const psg_search_ctx *sc, int64_t beg, int64_t size, const uint32_t *d_psa_lo, const uint8_t *d_psa_hi,
                             uint8_t *d_bwt, int64_t *i0, uint32_t *d_gt_begin);

/* ---- the merging half of the in-memory pSAscan in batches (inmem_psascan.hpp:64-304: max_threads sub-blocks are
 *      suffix-sorted, then merged by streaming; initial_partial_sufsort.hpp:61-319 for the sub-block sorts).  The host
 *      cores sort many small LEAVES of the range [range_beg, range_beg + range_size) -- leaf l = text positions
 *      [h_leaf_beg[l], h_leaf_beg[l+1]), ordered as suffixes of the WHOLE text -- and hand over only their partial
 *      SAs: d_leaf_psa holds them back to back in text order (entry k of leaf l at index h_leaf_beg[l] - range_beg + k),
 *      positions relative to the leaf's begin, psa_bytes = 2 (leaves of at most 65 536 positions) or 4 bytes each.
 *      The device derives BWT / i0 / gt bits of every leaf and merges neighbours pairwise, level by level, every
 *      level in ONE launch sequence for all its pairs (one rank structure over the level's BWT array, one stream
 *      kernel launch, one histogram, one merge).  Results as from psg_merge_run_u32 + psg_halfblock_from_psa: the
 *      range's partial SA (u32, relative to range_beg; range_size < 2^32 - 16), BWT (dummy 0 at *i0), i0, gt_begin
 *      (bit u <-> position range_beg + range_size - u).  sc: the text (comparison end = n; optionally a window).
 *      PSG_EWINDOW: a comparison left the text window; PSG_EUNRESOLVED: a comparison between leaves exceeded its
 *      budget of 2^20 symbols (long repeats across leaves) -- the caller sorts that range another way.            */
typedef struct {
  int64_t levels, passes, suffixes;     /* tree levels, pair merges, tail suffixes streamed over all levels */
  double prepare_ms, rank_ms, search_ms, stream_ms, hist_ms, bitvector_ms, merge_ms;   /* HIP events, summed over the levels */
  double total_ms;                      /* wall time of the call */
} psg_leaf_merge_stats;
int psg_merge_leaves(const psg_search_ctx *sc, int64_t range_beg, int64_t range_size, const int64_t *h_leaf_beg, int64_t n_leaves,
                     const void *d_leaf_psa, int psa_bytes, uint32_t *d_psa_out, uint8_t *d_bwt_out, int64_t *i0, uint32_t *d_gt_begin_out,
                     psg_leaf_merge_stats *stats);

/* ---- merge<T> with the partial suffix arrays in HOST memory.  The reference keeps every partial SA in part files
 *      (io/distributed_file.hpp:58-67) and streams them back during the merge (merge.hpp:72-81, 143; parts
 *      deleted as consumed, distributed_file.hpp:159-171); here they stay where the host sorter left them and
 *      are staged through pinned buffers slice by slice: slice k+1 is copied in while slice k is merged and
 *      slice k-1 is copied out / handed to the sink.  The merge bitvectors stay in HBM.
 *  h_psa_lo/h_psa_hi  host arrays (pageable or pinned), positions relative to beg
 *  sink               called in output order with `n_entries` packed uint40 values (5 bytes each, pinned host
 *                     memory, valid until it returns); non-zero return aborts the merge (PSG_ECHECK).
 *                     NULL: the output is produced in HBM and dropped (use with `check`).
 *  check              optional property check of every slice on the device: sum of all entries (mod 2^64; a
 *                     permutation of 0..n-1 gives n(n-1)/2) and sampled adjacent pairs out of suffix order.       */
typedef struct {
  int64_t beg, size;
  const uint32_t *h_psa_lo;
  const uint8_t *h_psa_hi;   /* may be NULL */
  const uint32_t *d_mbv;     /* device; size + (sizes of all later half-blocks) bits; NULL for the last */
  const uint32_t *d_psa_lo;  /* optional: this half-block's partial SA is ALREADY in HBM (h_psa_* are then ignored): */
  const uint8_t *d_psa_hi;   /* a caller with HBM to spare keeps some of them resident and saves their PCIe transfer */
  /* optional: the merge bitvector is in HOST memory (d_mbv NULL; all or none of the half-blocks) -- the reference keeps
   * its gap arrays in files and streams them through the merge (merge.hpp:80,145; gap_array.hpp:156-182).  h_mbv: the
   * words as psg_mbv_spill wrote them, h_mbv_samp: its rank samples.  Every output slice then uploads the words of
   * every level it touches (128-word aligned pieces) next to the pieces of the partial SAs.                       */
  const uint32_t *h_mbv;
  const uint64_t *h_mbv_samp;
} psg_hb_host_desc;
/* copy a finished merge bitvector of nbits bits out of HBM: h_words receives (nbits + 31) / 32 words (+ up to 3 of
 * padding: room for psg_mbv_spill_words(nbits) words), h_samp the number of one bits in front of every group of 4096
 * bits ((nbits + 4095) / 4096 + 1 values, the last one = all ones).  Both may be pageable memory.                  */
int64_t psg_mbv_spill_words(int64_t nbits);
int psg_mbv_spill(const uint32_t *d_mbv, int64_t nbits, uint32_t *h_words, uint64_t *h_samp);
/* the same in the background: h_samp is complete on return, h_words once psg_copy_wait(*out) has returned; d_mbv (from
 * psg_malloc) is handed to the library and freed when drained.  After the wait, psg_mbv_spill_finish clears the bits
 * behind nbits (the padding words included).                                                                      */
int psg_mbv_spill_begin(uint32_t *d_mbv, int64_t nbits, uint32_t *h_words, uint64_t *h_samp, psg_copy_t **out);
int psg_mbv_spill_finish(uint32_t *h_words, int64_t nbits);
typedef int (*psg_sink_fn)(void *ctx, const uint8_t *h_sa5, int64_t first_entry, int64_t n_entries);
typedef struct {
  const uint8_t *d_text;     /* the whole text on the device */
  int64_t n;
  int64_t samples_per_slice;
  uint64_t seed;
  uint64_t sum;              /* out */
  int64_t bad_pairs;         /* out */
  int64_t undecided_pairs;   /* out: sampled pairs that agree on 2^24 symbols (periodic text) -- not compared to the end, not counted as bad */
} psg_merge_check;
typedef struct {
  int64_t slices;
  double total_ms;           /* wall time of the call                                   */
  double kernel_ms;          /* merge kernels (HIP events)                              */
  double stage_ms;           /* host time spent copying PSA pieces into pinned buffers  */
  double sink_ms;            /* host time spent inside the sink                         */
  int64_t h2d_bytes, d2h_bytes;
} psg_merge_stream_stats;
int psg_merge_stream(const psg_hb_host_desc *hbs, int H, int64_t slice_entries, psg_merge_check *check,
                     psg_sink_fn sink, void *sink_ctx, psg_merge_stream_stats *stats);

/* ---- block-per-GPU schedule (psascan_amd/blockdist.py): pieces of the final merge when
 *      every half-block's merge bitvector and partial SA live on the rank that owns the block and every rank
 *      merges one range of the output.                                                                         */
/* h_ones[k] = number of one bits in d_bits[0 .. h_pos[k]) (rank1; ranksel_support.hpp:45-187), 0 <= pos <= nbits */
int psg_bits_rank1(const uint32_t *d_bits, int64_t nbits, const int64_t *h_pos, int64_t count, int64_t *h_ones);
/* A merge plan over SLICES: of every level only the part this rank's output range [out_begin, out_end) touches is
 * present.  Level h: d_mbv_words holds the words [first_word, first_word + n_words) of the level's merge bitvector,
 * first_word a multiple of 128 (a 4096-bit group), ones_before = one bits in front of that word; d_psa_lo (and
 * d_psa_hi) hold the elements [psa_first, psa_first + psa_count) of the half-block's partial SA.  psg_merge_run on
 * the plan accepts output ranges inside [out_begin, out_end).                                                   */
typedef struct {
  int64_t beg, size;             /* the half-block                                                   */
  int64_t nbits;                 /* length of the level's merge bitvector (0 for the last level)       */
  const uint32_t *d_mbv_words;   /* NULL for the last level                                            */
  int64_t first_word, n_words, ones_before;
  const uint32_t *d_psa_lo;
  const uint8_t *d_psa_hi;       /* may be NULL                                                       */
  int64_t psa_first, psa_count;
} psg_hb_slice_desc;
int psg_merge_plan_create_sliced(const psg_hb_slice_desc *levels, int H, psg_merge_plan_t **out);

/* ---- multi-GPU building blocks: one pass sharded over the TAIL (the reference's own parallel
 *      axis, compute_gap.hpp:68-69,114-124), the gap array sharded by index range.  Each rank
 *      streams its tail range into a rank log, the logs are exchanged so that every rank holds
 *      the entries of its gap slice (all-to-all), and slices are counted / turned into
 *      bitvector bits locally.  These replace the shared gap array + updater of
 *      update.hpp:60-220 across devices.                                                   ---- */
/* stream a tail range; instead of counting, hand back the log: *nlog u32 entries (0xFFFFFFFF =
 * no entry, arbitrary order) in a device buffer the caller releases with psg_free.           */
int psg_stream_gap_log(const psg_rank_t *rank, int64_t block_i0, int block_last_symbol,
                       const uint8_t *d_tail, int64_t tail_len, int64_t right_context,
                       const uint32_t *d_gt_in, int64_t rank_at_context_end, uint32_t *d_gt_out,
                       int64_t max_chains, int64_t *h_final_rank, psg_stream_stats *stats,
                       uint32_t **d_log, int64_t *nlog);
/* split the valid entries into nparts contiguous value ranges: part p occupies
 * d_out[h_offsets[p] .. h_offsets[p+1]) and holds the values [h_value_bounds[p],
 * h_value_bounds[p+1]); the bounds depend only on (m, nparts).                              */
int psg_log_partition(const uint32_t *d_log, int64_t nlog, int64_t m, int nparts, uint32_t *d_out,
                      int64_t *h_offsets, int64_t *h_value_bounds);
/* d_gap_slice[v - value_base] += #{log entries equal to v}, v in [value_base, value_base+count);
 * the log is clobbered.                                                                     */
int psg_gap_hist(uint32_t *d_log, int64_t nlog, int64_t value_base, int64_t count, uint32_t *d_gap_slice);
/* slice form of convert_to_bitvector: sets bit j + ps_before + sum_{t in slice, t<=j} gap[t] for
 * every j in [j0, j0+count) with j < m in a ZERO-initialised array of the global size; after
 * summing the arrays of all ranks (disjoint bits) psg_bits_not gives the bitvector.          */
int psg_gap_slice_to_bits(const uint32_t *d_gap_slice, int64_t j0, int64_t count, int64_t m,
                          uint64_t ps_before, uint32_t *d_bits);
int psg_bits_not(uint32_t *d_bits, int64_t nbits);

/* ---- small device utilities the host orchestration needs ------------------------------ */
/* dst bits [dst_bit, dst_bit+nbits) = src bits [src_bit, src_bit+nbits) (non-overlapping) */
int psg_bitcopy(uint32_t *d_dst, int64_t dst_bit, const uint32_t *d_src, int64_t src_bit, int64_t nbits);
/* number of one bits in [0, nbits)                                                        */
int psg_popcount(const uint32_t *d_bits, int64_t nbits, int64_t *ones);
/* timing of the last kernel sequence of the named entry point, in ms (HIP events)          */
double psg_last_kernel_ms(void);

#ifdef __cplusplus
}
#endif
#endif /* PSASCAN_AMD_H */
