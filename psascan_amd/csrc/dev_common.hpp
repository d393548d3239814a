// dev_common.hpp -- shared host/device helpers of the HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/psascan_amd.h"

typedef int64_t i64;
typedef uint64_t u64;
typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t u8;

namespace psg {

// ---- error plumbing ------------------------------------------------------------------
void set_error(const std::string &s);
hipStream_t stream();                       // the stream every launch of the library goes to
hipStream_t side_stream();                  // a second stream for work that overlaps the main one (created on first use)
struct StreamScope {                        // within the scope, stream() is `s`
  hipStream_t prev;
  explicit StreamScope(hipStream_t s);
  ~StreamScope();
};
void note_kernel_ms(double ms);

#define PSG_HIP(call)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      psg::set_error(std::string(#call) + ": " + hipGetErrorString(e_));                     \
      return e_ == hipErrorOutOfMemory ? PSG_ENOMEM : PSG_EDEVICE;                           \
    }                                                                                        \
  } while (0)

#define PSG_REQUIRE(cond, msg)                                                               \
  do {                                                                                       \
    if (!(cond)) { psg::set_error(std::string("invalid argument: ") + msg); return PSG_EINVAL; } \
  } while (0)

// caching device allocator: hipMalloc/hipFree of multi-GiB buffers cost ~100s of ms each
// (page-table work + implicit sync); freed blocks are kept and reused (single stream =>
// stream-ordered reuse is safe).  pool_trim() returns everything to the driver.
hipError_t pool_alloc(void **p, size_t bytes);
void pool_free(void *p);
void pool_trim();
size_t pool_cached_bytes();   // bytes held in the cache (reusable without asking the driver)
size_t mem_available();       // what a new structure may take: the rest of the budget (psg_set_memory_limit) or free + cached device memory

// grow-only pinned host buffers (slot 0..15): 0-3 stream pass (per-chain arrays, read-backs), 4 wide-log slab
// bounds, 7 copy staging, 8-11 streamed merge (PSA pieces in, .sa5 slices out), 12 merge cursors, 13 the LDS tables of a pass; 5-6 search positions / ranks
void *pinned_buf(int slot, size_t bytes);
// Synchronous host<->device copies staged through pinned memory.  Pageable host pointers are never
// handed to HIP: ROCr registers them as userptr ranges, and when the host later unmaps / trims
// that memory the KFD MMU notifier evicts and restores the process' queues -- measured as 20-30 ms
// stalls of the next HIP call in a run-dependent fraction of the processes.
int copy_h2d(void *d, const void *h, size_t bytes);
int copy_d2h(void *h, const void *d, size_t bytes);

// RAII device buffer for temporaries
struct DevBuf {
  void *p = nullptr;
  i64 bytes = 0;
  DevBuf() {}
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { if (p) psg::pool_free(p); }
  int alloc(i64 b) {
    if (p) { psg::pool_free(p); p = nullptr; }
    bytes = b < 16 ? 16 : b;
    hipError_t e = psg::pool_alloc(&p, (size_t)bytes);
    if (e != hipSuccess) { p = nullptr; psg::set_error(std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e)); return PSG_ENOMEM; }
    return 0;
  }
  template <class T> T *as() const { return (T *)p; }
};

// Wait for the library stream by POLLING an event: hipStreamSynchronize's interrupt-driven wait was
// measured to return 15-25 ms late for ~6 ms of GPU work in some processes (MI355X, ROCm 7.2).
hipError_t sync_stream();

// events are recycled: hipEventCreate/Destroy per call showed up as multi-ms host stalls
hipEvent_t event_acquire();
void event_release(hipEvent_t e);
struct EventTimer {
  hipEvent_t a = nullptr, b = nullptr;
  EventTimer() { a = event_acquire(); b = event_acquire(); }
  ~EventTimer() { event_release(a); event_release(b); }
  EventTimer(const EventTimer &) = delete;
  EventTimer &operator=(const EventTimer &) = delete;
  void start() { (void)hipEventRecord(a, stream()); }
  void stop() { (void)hipEventRecord(b, stream()); }
  double ms() { float f = 0; (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&f, a, b); return f; }
};

static inline i64 cdiv(i64 a, i64 b) { return (a + b - 1) / b; }

// ---- gap arrays with an in-band excess list (include/psascan_amd.h) ------------------------------------------
// header words behind the counters: [0] number of entries, [1] counter bits (0 = 32), [2] capacity-exceeded flag,
// [3] capacity of the list (0 = PSG_GAP_EXCESS_CAP)
struct GapExcess {          // what a kernel needs to append
  u32 *hdr;                 // null: no excess handling wanted
  u64 *ent;
  int bits;                 // counter width (32 in production; 8/16 in tests)
};
static inline u32 *gap_hdr(u32 *gap, i64 m) { return gap + PSG_GAP_HDR_WORD(m); }
static inline GapExcess gap_excess(u32 *gap, i64 m, int bits) { return GapExcess{gap_hdr(gap, m), (u64 *)(gap_hdr(gap, m) + 4), bits}; }
// counter width of a gap array as recorded in its header (host round trip); initialises the header of a fresh or
// all-zero array (PSG_GAP_COUNTER_BITS = 8 | 16: tests).  fresh: the header holds garbage.
int gap_prepare(u32 *d_gap, i64 m, bool fresh, int *bits);
// the excess entries sorted by slot, for consumers: *d_sorted is a pool buffer the caller releases (null when n == 0)
struct ExcessView { const u64 *sorted; u32 n; int bits; };
int gap_excess_view(const u32 *d_gap, i64 m, ExcessView *view, void **owned);

// gap[v] += #{entries of log equal to v} for v in [0, m]; entries 0xFFFFFFFF are skipped.
// Sorts the log by its high bits (in place semantics: log is clobbered) and histograms
// LDS-sized windows -- replaces one random atomic per streamed suffix (gap_hist.hip).
// overwrite: d_gap holds garbage on entry and exactly the histogram on return (no zero-fill needed)
// excess: where counters that wrap leave their carries (slot numbers are offset by slot_base)
int gap_hist_from_log(u32 *d_log, i64 nlog, i64 m, u32 *d_gap, double *ms, bool overwrite, GapExcess ex = GapExcess{nullptr, nullptr, 32}, i64 slot_base = 0);
// the same for ranks of up to 40 bits (blocks of >= 2^32 - 1 symbols): log_lo holds the low words, log_hi one byte per
// entry with bits 32..39 (no entry = 0xFFFFFFFF / 0xFF).  The log is first split into slabs of 2^31 counters (values
// relative to the slab start), each slab is then histogrammed like a 32-bit log.  Both buffers are released.
int gap_hist_from_wide_log(DevBuf &log_lo, DevBuf &log_hi, i64 nlog, i64 m, u32 *d_gap, double *ms, bool overwrite, GapExcess ex);
// the same in two halves: launch enqueues everything on stream() without waiting for the device, wait
// blocks until the job is done, checks the overflow flag and releases the job's buffers
struct HistJob {
  DevBuf part1, part2, counts, off, bin_base, win_off, cnt, tot, ovf;
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  hipStream_t s = nullptr;
  bool active = false;
  bool own_excess = false;   // the job's own stand-in excess area: a carry is an error
};
int gap_hist_launch(HistJob &job, u32 *d_log, i64 nlog, i64 m, u32 *d_gap, bool overwrite, GapExcess ex = GapExcess{nullptr, nullptr, 32}, i64 slot_base = 0);
// two-plane log, arrays of up to 2^33 counters (one slab: gap_hist_wide_one_slab): launch only, on stream()
bool gap_hist_wide_one_slab(i64 m);
int gap_hist_wide_launch(HistJob &job, const u32 *log_lo, const u8 *log_hi, i64 nlog, i64 m, u32 *d_gap, bool overwrite, GapExcess ex);
int gap_hist_wait(HistJob &job, double *ms);

// property check of `count` packed uint40 entries (prep.hip): acc[0] += sum of the entries (mod 2^64),
// acc[1] += number of `samples` random adjacent pairs that are NOT in suffix order.  Enqueued on stream().
int check_sa5_accumulate(const u8 *d_text, i64 n, const u8 *d_sa5, i64 count, i64 samples, u64 seed, unsigned long long *d_acc);

// K8 (search.hip): enqueue the string search for npos device-resident positions; ranks land in d_rank
int search_ranks_launch(const psg_search_ctx *sc, const i64 *d_pos, i64 npos, i64 *d_rank);
int search_window_check();   // after the search has completed: PSG_EWINDOW if a comparison left the context's text window

// set around a psg_rank_build call: one bit per 3072-position build segment of the symbol-major layout, 0 = nothing will
// query inside that segment, its entries need not be written (device memory; nullptr = build everything)
extern const u32 *rank_build_seg_mask;
constexpr i64 RANK_BUILD_SEG = 3072;

// ---- batched passes: many (block, tail) pairs of one level of the leaf merging in one launch (rank_stream.hip,
// bits_merge.hip, leaf_tree.hip).  Positions are relative to the begin of the enclosing range; the blocks' BWTs lie
// at their positions in ONE array over which ONE rank structure is built.
struct BatchGeom {
  i64 lbeg, m, T;            // block = [lbeg, lbeg + m), tail = the m.. T positions behind it
  i64 kbase;                 // first chain of this pass among all chains of the launch
  i64 gap_base;              // first slot of the pass in the shared gap array (slot m of a pass = slot 0 of the next)
  i64 ones_before;           // sum of T over the passes in front = one bits in front of lbeg in the merge bitvector
  i64 gt_in_word, gt_l_word; // word offsets of the tail's / the block's gt array in the level's current gt buffer
  i64 gt_out_word;           // word offset of the parent's gt array in the next level's buffer
  i64 node;                  // index of the block's node (i0 array)
};
struct BatchTables;          // per-launch tables of the rank structure (owned by the caller of stream_batch_setup)
BatchTables *stream_batch_tables_create(const psg_rank_t *r, i64 npass);
void stream_batch_tables_free(BatchTables *t);
// fills the pass table of a launch on the device: per-pass C arrays folded with the structure's counts in front of the
// block (LDS tables), the pass parameters.  Enqueue only.
int stream_batch_setup(const psg_rank_t *r, BatchTables *tabs, const BatchGeom *d_geom, i64 npass, const i64 *d_i0_nodes, const u8 *d_text_range,
                       const u32 *d_gt_cur, u32 *d_gt_new, i64 L, i64 *d_init, i64 *d_fin, u32 *d_log, i64 Ktotal, u32 *d_gap, GapExcess ex);
// mode 0 / 1 / 2: atomics, atomics with carries, rank log (32-bit).  Enqueue only.
int stream_batch_launch(const psg_rank_t *r, const BatchTables *tabs, const u32 *d_wg_pass, const u32 *d_wg_local, i64 nwg, int mode);
int stream_batch_blocks_per_cu(const psg_rank_t *r, int mode);
// out[x] = bit x of bv ? tail element : block element, for every pass of the level (positions [0, nbits)): partial SA
// values and BWT symbols move together; the tail's dummy symbol is patched with the block's last symbol
// (bwt_merge.hpp:128), i0_out[pass] = where the block's first suffix lands (bwt_merge.hpp:133).  Enqueue only.
int merge_pairs_launch(const u32 *d_bv, i64 nbits, const BatchGeom *d_geom, i64 npass, const u32 *d_tile_pass, const u32 *d_psa, const u8 *d_bwt,
                       const u8 *d_text_range, u32 *d_psa_out, u8 *d_bwt_out, i64 *d_i0_out);
int merge_pairs_tile();      // output slots per tile (d_tile_pass[t] = pass that holds slot t * tile)

// single-workgroup exclusive scan of n u64 values in place; total -> d_total (may be null)
int scan_u64_inplace(u64 *d_vals, i64 n, u64 *d_total);

}  // namespace psg

// ---- device helpers ----------------------------------------------------------------
#define PSG_WG 256

__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63; }

// inclusive scan across the 64 lanes of a wave
template <class T> __device__ __forceinline__ T wave_incl_scan(T v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T o = __shfl_up(v, d, 64);
    if ((int)lane_id() >= d) v += o;
  }
  return v;
}

// exclusive scan over a 256-thread workgroup; `total` gets the workgroup sum.
// scratch: 8 elements of T in LDS.  Contains two barriers.
template <class T> __device__ __forceinline__ T block_excl_scan(T v, T *scratch, T &total) {
  T inc = wave_incl_scan(v);
  int w = threadIdx.x >> 6;
  if (lane_id() == 63) scratch[w] = inc;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < PSG_WG / 64; ++k) {
    T s = scratch[k];
    if (k < w) base += s;
    tot += s;
  }
  __syncthreads();
  total = tot;
  return base + inc - v;
}

template <class T> __device__ __forceinline__ T block_sum(T v, T *scratch) {
  T tot;
  (void)block_excl_scan(v, scratch, tot);
  return tot;
}

// Load through a pointer whose address space the compiler cannot see (rebuilt from an integer, or
// read from a descriptor in memory).  Forces global_load: a flat_load also counts against lgkmcnt,
// so every wait for an LDS access would wait for the outstanding global loads as well.
template <class T> __device__ __forceinline__ T gload(const T *p) {
  return *(const __attribute__((address_space(1))) T *)p;
}
typedef unsigned int psg_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 gload(const uint4 *p) {
  psg_u32x4 v = *(const __attribute__((address_space(1))) psg_u32x4 *)p;
  return make_uint4(v.x, v.y, v.z, v.w);
}

// bits [pos, pos+cnt) (cnt <= 32) of an LSB-first u32 bit array; reads at most words
// pos/32 and pos/32+1 (the second only if needed and < nwords).
__device__ __forceinline__ u32 get_bits(const u32 *bv, i64 pos, int cnt, i64 nwords) {
  i64 w = pos >> 5;
  int sh = (int)(pos & 31);
  u64 lo = gload(bv + w);   // bit arrays always live in HBM
  u64 hi = (sh + cnt > 32 && w + 1 < nwords) ? gload(bv + w + 1) : 0;
  u64 x = (lo | (hi << 32)) >> sh;
  return cnt >= 32 ? (u32)x : (u32)x & ((1u << cnt) - 1u);
}

// ---- excess list, device side ---------------------------------------------------------------------------------
// `carries` carries out of slot j: append that many entries (each stands for 2^bits)
__device__ __forceinline__ void excess_append(const psg::GapExcess &X, u64 j, u32 carries) {
  if (!carries) return;
  const u32 pos = atomicAdd(X.hdr, carries);
  const u32 cap = X.hdr[3] ? X.hdr[3] : (u32)PSG_GAP_EXCESS_CAP;
  for (u32 t = 0; t < carries; ++t) {
    if (pos + t < cap) X.ent[pos + t] = j;
    else X.hdr[2] = 1u;       // cannot happen with 32-bit counters (n <= 2^40 gives at most 256 carries)
  }
}
// counter += c (not atomic: the caller owns the slot); returns the new counter value, appends the carries
__device__ __forceinline__ u32 excess_add_owned(const psg::GapExcess &X, u64 j, u32 old, u32 c) {
  const u64 sum = (u64)old + c;
  if (X.bits >= 32) { if (sum >> 32) excess_append(X, j, 1u); return (u32)sum; }
  excess_append(X, j, (u32)(sum >> X.bits));
  return (u32)(sum & ((1ull << X.bits) - 1ull));
}
// counter += c with an atomic (other threads add to the same slot): every multiple of 2^bits the running value
// crosses is turned into one entry; with narrow counters the crossing thread also takes 2^bits off again
__device__ __forceinline__ void excess_add_atomic(const psg::GapExcess &X, u32 *cell, u64 j, u32 c) {
  const u32 old = atomicAdd(cell, c);
  if (X.bits >= 32) { if ((u32)(old + c) < old) excess_append(X, j, 1u); return; }
  const u32 carries = (u32)((((u64)old + c) >> X.bits) - ((u64)old >> X.bits));
  if (carries) { atomicSub(cell, carries << X.bits); excess_append(X, j, carries); }
}
// values of the slots [base, base + 8): g[q] += 2^bits for every entry equal to base + q
__device__ __forceinline__ void excess_apply8(const psg::ExcessView &X, i64 base, u64 (&g)[8]) {
  if (!X.n) return;
  u32 lo = 0, hi = X.n;
  while (lo < hi) { const u32 md = (lo + hi) >> 1; if (X.sorted[md] < (u64)base) lo = md + 1; else hi = md; }
  for (; lo < X.n && X.sorted[lo] < (u64)base + 8; ++lo) g[X.sorted[lo] - (u64)base] += 1ull << X.bits;
}
