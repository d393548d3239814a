// gap_hist.hip -- gap increments without random atomics (replaces update.hpp:86-96's one random
// read-modify-write per streamed suffix; random u32 atomics top out at 17.9 G/s on MI355X,
// profiles/r01_membench.txt).
//
// The stream kernel (MODE 2) logs the rank of every streamed suffix with coalesced stores.  Here
// the log is partitioned so that every WINDOW of 2^WBITS consecutive gap counters becomes one
// contiguous piece of it, and each window is histogrammed in LDS and added to (or, for a fresh gap
// array, stored into) the gap array with coalesced 16-byte accesses.  The partition is a
// hand-written two-level MSD radix split (<= 512 bins per level) with deterministic offsets: a
// count pass gives every workgroup private output segments, so there are no global atomics and no
// ordering requirements.  Two versions live here: the exact-size scatter (tiles staged in LDS, written
// as per-bin runs) behind psg_log_partition, whose output goes over the wire, and "partition v2"
// (per-bin staging in LDS across tiles, whole 64-byte units only, padded segments) behind the
// histogram path.  (The reference does the same thing for cache locality on the CPU:
// stream.hpp:160-232 buckets every buffer of ranks by value range before the updaters run.)
#include "dev_common.hpp"

#include <algorithm>
#include <cstring>

using namespace psg;

#define WBITS 15           // counters per window: two 16-bit counters share an LDS word (64 KiB per histogram)
#define WSIZE (1 << WBITS)
#define CAP 61440          // log entries per histogram work item: < 2^16, so a 16-bit counter cannot wrap inside one item
#define PBINS 512          // bins per partition level
#ifndef PT
#define PT 16384           // entries per partition tile (runs of ~128 B per bin: full-line writes, tools/membench)
#endif
#define PNT 1024           // threads of a partition workgroup (8 entries per thread and tile)
#define PAD 0xFFFFFFFFu

struct Tile {              // LDS of one partition workgroup (~76 KiB -> 2 workgroups per CU)
  u32 stage[PT];
  u32 h[PBINS];
  u32 loff[PBINS];
  u64 gbase[PBINS];
  u64 cur[PBINS];
  u32 scratch[PNT / 64];
};

// exclusive scan over the PNT threads of a partition workgroup
__device__ __forceinline__ u32 part_scan(u32 v, u32 *scratch, u32 &total) {
  u32 inc = wave_incl_scan(v);
  int w = threadIdx.x >> 6;
  if (lane_id() == 63) scratch[w] = inc;
  __syncthreads();
  u32 base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < PNT / 64; ++k) { u32 x = scratch[k]; if (k < w) base += x; tot += x; }
  __syncthreads();
  total = tot;
  return base + inc - v;
}

// scatter one tile [beg, end) (end - beg <= PT) of `keys` into `out` at the workgroup's cursors
__device__ __forceinline__ void scatter_tile(Tile &S, const u32 *keys, i64 beg, i64 end, int shift, u32 mask, u32 *out) {
  u32 v[PT / PNT];
  u32 r[PT / PNT];
  for (int b = threadIdx.x; b < PBINS; b += PNT) S.h[b] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < PT / PNT; ++j) {
    i64 k = beg + j * PNT + threadIdx.x;
    v[j] = k < end ? keys[k] : PAD;
    if (v[j] != PAD) r[j] = atomicAdd(&S.h[(v[j] >> shift) & mask], 1u);
  }
  __syncthreads();
  // exclusive scan over the 512 bins (one per thread < 512) + advance the cursors
  u32 hb = threadIdx.x < PBINS ? S.h[threadIdx.x] : 0, tot;
  u32 pre = part_scan(hb, S.scratch, tot);
  if (threadIdx.x < PBINS) {
    S.loff[threadIdx.x] = pre;
    S.gbase[threadIdx.x] = S.cur[threadIdx.x] - pre;   // staged slot s of this bin goes to out[gbase + s]
    S.cur[threadIdx.x] += hb;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < PT / PNT; ++j) {
    if (v[j] != PAD) {
      u32 bin = (v[j] >> shift) & mask;
      u32 slot = S.loff[bin] + r[j];
      S.stage[slot] = v[j];
    }
  }
  __syncthreads();
  for (u32 s = threadIdx.x; s < tot; s += PNT) {
    u32 x = S.stage[s];
    u32 bin = (x >> shift) & mask;
    out[S.gbase[bin] + s] = x;
  }
  __syncthreads();
}

// level 1, step A: persistent workgroup g counts its tiles (g, g+G, ...)
__global__ __launch_bounds__(PNT) void part_count_kernel(const u32 *keys, i64 n, int shift, u32 *counts) {
  __shared__ u32 h[PBINS];
  for (int b = threadIdx.x; b < PBINS; b += PNT) h[b] = 0;
  __syncthreads();
  i64 ntiles = (n + PT - 1) / PT;
  for (i64 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    i64 beg = tile * PT;
    for (int j0 = 0; j0 < PT / PNT; j0 += 8) {   // 8 independent loads in flight, then the LDS atomics
      u32 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { i64 k = beg + (j0 + j) * PNT + threadIdx.x; v[j] = k < n ? keys[k] : PAD; }
#pragma unroll
      for (int j = 0; j < 8; ++j) if (v[j] != PAD) atomicAdd(&h[v[j] >> shift], 1u);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < PBINS; b += PNT) counts[(i64)blockIdx.x * PBINS + b] = h[b];
}

// level 1, step B (one workgroup of 512 threads): off[g][b] = start of workgroup g's run inside bin b
__global__ __launch_bounds__(PBINS) void part_offsets_kernel(const u32 *counts, int G, u64 *off, u64 *bin_base) {
  __shared__ u64 tot[PBINS];
  int b = threadIdx.x;
  u64 t = 0;
  for (int g = 0; g < G; ++g) t += counts[(i64)g * PBINS + b];
  tot[b] = t;
  __syncthreads();
  if (b == 0) {  // 512-entry exclusive scan; tiny
    u64 run = 0;
    for (int k = 0; k < PBINS; ++k) { u64 x = tot[k]; tot[k] = run; run += x; }
    bin_base[PBINS] = run;
  }
  __syncthreads();
  u64 run = tot[b];
  bin_base[b] = run;
  for (int g = 0; g < G; ++g) { off[(i64)g * PBINS + b] = run; run += counts[(i64)g * PBINS + b]; }
}

// level 1, step C: persistent workgroup g scatters its tiles at its private cursors
__global__ __launch_bounds__(PNT) void part_scatter_kernel(const u32 *keys, i64 n, int shift, const u64 *off, u32 *out) {
  __shared__ Tile S;
  for (int b = threadIdx.x; b < PBINS; b += PNT) S.cur[b] = off[(i64)blockIdx.x * PBINS + b];
  __syncthreads();
  i64 ntiles = (n + PT - 1) / PT;
  for (i64 tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
    scatter_tile(S, keys, tile * PT, std::min<i64>(tile * PT + PT, n), shift, 0xFFFFFFFFu, out);
}

// level 2: workgroup b splits segment b of level 1 by the next bits; records the window starts
__global__ __launch_bounds__(PNT) void part_level2_kernel(const u32 *keys, const u64 *bin_base, int bits2, u32 *out, u64 *win_off) {
  __shared__ Tile S;
  const i64 sb = (i64)bin_base[blockIdx.x], se = (i64)bin_base[blockIdx.x + 1];
  const u32 mask = (1u << bits2) - 1u;
  for (int b = threadIdx.x; b < PBINS; b += PNT) S.h[b] = 0;
  __syncthreads();
  for (i64 k0 = sb; k0 < se; k0 += 8 * PNT) {
    u32 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { i64 k = k0 + j * PNT + threadIdx.x; v[j] = k < se ? keys[k] : PAD; }
#pragma unroll
    for (int j = 0; j < 8; ++j) if (v[j] != PAD) atomicAdd(&S.h[(v[j] >> WBITS) & mask], 1u);
  }
  __syncthreads();
  u32 hb = threadIdx.x < PBINS ? S.h[threadIdx.x] : 0, tot;
  u32 pre = part_scan(hb, S.scratch, tot);
  if (threadIdx.x < PBINS) S.cur[threadIdx.x] = (u64)sb + pre;
  __syncthreads();
  for (u32 c = threadIdx.x; c <= mask; c += PNT) win_off[((i64)blockIdx.x << bits2) + c] = S.cur[c];
  __syncthreads();
  for (i64 beg = sb; beg < se; beg += PT) scatter_tile(S, keys, beg, std::min<i64>(beg + PT, se), WBITS, mask, out);
}

// ---------------------------------------------------------------------------------------
// Partition, second version (used by gap_hist_from_log; psg_log_partition keeps the exact-size
// scatter above).  What the first version gets wrong on this memory system (tools/membench):
// a run of ~32 entries per bin and tile starts and ends in the middle of a 64-byte sector, and a
// partially written sector costs a read-modify-write in HBM.  Here every workgroup keeps 32
// staging slots per bin in LDS ACROSS its tiles and writes whole, aligned 16-entry units only;
// the per-workgroup segment of a bin is rounded up to a multiple of 16 entries and the holes hold
// the "no entry" value, which every consumer of the log already skips.
// ---------------------------------------------------------------------------------------
#ifndef P2L2_A
#define P2L2_A 128
#define P2L2_B 64
#define P2L2_EA 4
#define P2L2_EB 2
#endif
#ifndef P2TS
#define P2TS 2048          // entries per level-1 tile
#endif
#define P2T 512            // threads per workgroup (2 workgroups per CU)
#define P2C 32             // staging slots per bin
#define P2U 16             // unit = 16 entries = one 64-byte sector
#define P2SLACK ((i64)P2U * PBINS)   // rounding growth bound of one partition call per (workgroup | segment)

struct Stage2 {            // ~71 KiB
  u32 cnt[PBINS];
  __attribute__((aligned(16))) u32 buf[PBINS][P2C];
  u64 cur[PBINS];
  u16 list[PBINS];
  u32 nlist;
  u32 scratch[P2T / 64];
};

__device__ __forceinline__ u32 p2_scan(u32 v, u32 *scratch, u32 &total) {   // exclusive scan over the P2T threads
  u32 inc = wave_incl_scan(v);
  int w = threadIdx.x >> 6;
  if (lane_id() == 63) scratch[w] = inc;
  __syncthreads();
  u32 base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < P2T / 64; ++k) { u32 x = scratch[k]; if (k < w) base += x; tot += x; }
  __syncthreads();
  total = tot;
  return base + inc - v;
}

// bin of a value: (v >> shift) - base, the last bin (511) also takes everything above it -- the value
// range need not be a power of two, so the top bin of level 1 may own a few windows more than the others
__device__ __forceinline__ u32 p2_bin(u32 v, int shift, u32 base) {
  const u32 b = (v >> shift) - base;
  return b < PBINS - 1 ? b : PBINS - 1;
}

// Key sources of the level-1 pass: entry -> (bin, value RELATIVE to the start of the bin).  Level 1 writes bin-relative
// values, so whatever the width of the ranks, everything behind level 1 works on 32-bit values: the window of a
// value is (bin << bits2) + (value >> WBITS), its counter value & (WSIZE - 1).
struct Keys32 {            // a 32-bit rank log
  const u32 *k; int shift; int relative;   // relative = 0: keep the value (level 2: the bin base is handled by p2_bin's base)
  u32 base;
  template <int EPT> __device__ __forceinline__ void load_tile(u32 (&v)[EPT], u32 (&b)[EPT], i64 beg, i64 end) const {
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      i64 i = beg + j * P2T + threadIdx.x;
      const u32 x = i < end ? k[i] : PAD;
      b[j] = p2_bin(x, shift, base);
      v[j] = (x == PAD || !relative) ? x : x - (b[j] << shift);
    }
  }
};
struct Keys40 {            // ranks of up to 40 bits in two planes (stream kernel MODE 3), restricted to the counters
  const u32 *lo; const u8 *hi; int shift;   // [slab_lo, slab_hi) (everything else reads as "no entry"); beg, end multiples of 4
  u64 slab_lo, slab_hi;
  template <int EPT> __device__ __forceinline__ void load_tile(u32 (&v)[EPT], u32 (&b)[EPT], i64 beg, i64 end) const {
    static_assert(EPT % 4 == 0, "four consecutive entries per thread and load");
#pragma unroll
    for (int jj = 0; jj < EPT / 4; ++jj) {
      const i64 i = beg + ((i64)jj * P2T + threadIdx.x) * 4;
      uint4 l = make_uint4(PAD, PAD, PAD, PAD);
      u32 h = 0xFFFFFFFFu;
      if (i < end) { l = *(const uint4 *)(lo + i); h = *(const u32 *)(hi + i); }
      const u32 lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const u32 hb = (h >> (8 * q)) & 255u;
        const u64 x = ((u64)hb << 32) | lw[q];
        const bool in = x >= slab_lo && x < slab_hi && !(lw[q] == PAD && hb == 0xFFu);
        const u64 xr = x - slab_lo;
        const u64 bin = xr >> shift;
        b[4 * jj + q] = bin < PBINS - 1 ? (u32)bin : PBINS - 1;
        v[4 * jj + q] = in ? (u32)(xr - ((u64)b[4 * jj + q] << shift)) : PAD;
      }
    }
  }
};

// EPT entries per thread: a tile should bring ~4 entries per bin in use, so that the 16 slots of headroom
// above the flush threshold are practically never exceeded
template <int EPT>
__device__ __forceinline__ void p2_insert_tile(Stage2 &S, const u32 (&v)[EPT], const u32 (&bn)[EPT], u32 *out) {
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    if (v[j] == PAD) continue;
    const u32 b = bn[j];
    const u32 slot = atomicAdd(&S.cnt[b], 1u);
    if (slot < P2C) S.buf[b][slot] = v[j];
    else out[atomicAdd((unsigned long long *)&S.cur[b], 1ull)] = v[j];   // bin full between two flushes: rare, written directly
  }
}

// write out the complete units of every bin
__device__ __forceinline__ void p2_flush(Stage2 &S, u32 *out) {
  __syncthreads();
  for (int b = threadIdx.x; b < PBINS; b += P2T) {
    const u32 n = S.cnt[b] < P2C ? S.cnt[b] : P2C;
    if (n >= P2U) S.list[atomicAdd(&S.nlist, 1u)] = (u16)b;
    S.cnt[b] = n;
  }
  __syncthreads();
  const u32 nl = S.nlist;
  const int li = threadIdx.x & 3;
  for (u32 f = threadIdx.x >> 2; f < nl; f += P2T / 4) {   // 4 lanes x 16 bytes = one unit
    const u32 b = S.list[f], n = S.cnt[b];
    const u64 pos = S.cur[b];
    const uint4 lo = ((const uint4 *)S.buf[b])[li], hi = ((const uint4 *)S.buf[b])[4 + li];
    const bool two = n >= P2C;                               // both units are complete
    u32 *dst = out + pos;
    if ((pos & 3) == 0) {
      ((uint4 *)dst)[li] = lo;
      if (two) ((uint4 *)dst)[4 + li] = hi;
    } else {                                                 // cursor knocked off its alignment by a direct write
      dst[4 * li] = lo.x; dst[4 * li + 1] = lo.y; dst[4 * li + 2] = lo.z; dst[4 * li + 3] = lo.w;
      if (two) { dst[16 + 4 * li] = hi.x; dst[17 + 4 * li] = hi.y; dst[18 + 4 * li] = hi.z; dst[19 + 4 * li] = hi.w; }
    }
    if (!two) ((uint4 *)S.buf[b])[li] = hi;                  // entries 16.. move to the front (LDS accesses of a wave stay in order)
    if (li == 0) { S.cnt[b] = two ? 0 : n - P2U; S.cur[b] = pos + (two ? 2 * P2U : P2U); }
  }
  __syncthreads();
  if (threadIdx.x == 0) S.nlist = 0;
}

// end of a workgroup's input: the leftovers (< one unit per bin after p2_flush), then "no entry" up to the
// next unit boundary -- segments are whole units, so the written length is exactly the segment's
__device__ __forceinline__ void p2_flush_final(Stage2 &S, u32 *out) {
  p2_flush(S, out);
  for (int b = threadIdx.x; b < PBINS; b += P2T) {
    const u32 n = S.cnt[b];
    u64 pos = S.cur[b];
    for (u32 k = 0; k < n; ++k) out[pos++] = S.buf[b][k];
    while (pos & (P2U - 1)) out[pos++] = PAD;
    S.cur[b] = pos;
    S.cnt[b] = 0;
  }
}

// all tiles of the entries [beg0, end0): the loads of the next tile are in flight while this one is inserted and flushed
template <int EPT, class KEYS>
__device__ __forceinline__ void p2_run(Stage2 &S, const KEYS &keys, i64 beg0, i64 end0, u32 *out) {
  u32 v[EPT], b[EPT], vn[EPT], bn[EPT];
  keys.template load_tile<EPT>(v, b, beg0, end0);
  for (i64 beg = beg0; beg < end0; beg += EPT * P2T) {
    keys.template load_tile<EPT>(vn, bn, beg + EPT * P2T, end0);
    p2_insert_tile<EPT>(S, v, b, out);
    p2_flush(S, out);
#pragma unroll
    for (int j = 0; j < EPT; ++j) { v[j] = vn[j]; b[j] = bn[j]; }
  }
  p2_flush_final(S, out);
}

// level 1, step A: workgroup g counts the entries of its chunk [g*chunk, (g+1)*chunk)
template <class KEYS>
__global__ __launch_bounds__(P2T) void p2_count_kernel(KEYS keys, i64 n, i64 chunk, u32 *counts) {
  __shared__ u32 h[PBINS];
  for (int b = threadIdx.x; b < PBINS; b += P2T) h[b] = 0;
  __syncthreads();
  const i64 cb = (i64)blockIdx.x * chunk, ce = std::min<i64>(cb + chunk, n);
  for (i64 k0 = cb; k0 < ce; k0 += 8 * P2T) {   // 8 independent loads in flight, then the LDS atomics
    u32 v[8], bn[8];
    keys.template load_tile<8>(v, bn, k0, ce);
#pragma unroll
    for (int j = 0; j < 8; ++j) if (v[j] != PAD) atomicAdd(&h[bn[j]], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < PBINS; b += P2T) counts[(i64)blockIdx.x * PBINS + b] = h[b];
}

// level 1, step B: off[g][b] = start of workgroup g's segment of bin b; segments are whole units
__global__ __launch_bounds__(PBINS) void p2_offsets_kernel(const u32 *counts, int G, u64 *off, u64 *bin_base) {
  __shared__ u64 tot[PBINS];
  int b = threadIdx.x;
  u64 t = 0;
  for (int g = 0; g < G; ++g) t += (counts[(i64)g * PBINS + b] + (P2U - 1)) / P2U * P2U;
  tot[b] = t;
  __syncthreads();
  if (b == 0) {
    u64 run = 0;
    for (int k = 0; k < PBINS; ++k) { u64 x = tot[k]; tot[k] = run; run += x; }
    bin_base[PBINS] = run;
  }
  __syncthreads();
  u64 run = tot[b];
  bin_base[b] = run;
  for (int g = 0; g < G; ++g) { off[(i64)g * PBINS + b] = run; run += (counts[(i64)g * PBINS + b] + (P2U - 1)) / P2U * P2U; }
}

// level 1, step C
template <class KEYS>
__global__ __launch_bounds__(P2T) void p2_scatter_kernel(KEYS keys, i64 n, i64 chunk, const u64 *off, u32 *out) {
  __shared__ Stage2 S;
  for (int b = threadIdx.x; b < PBINS; b += P2T) { S.cur[b] = off[(i64)blockIdx.x * PBINS + b]; S.cnt[b] = 0; }
  if (threadIdx.x == 0) S.nlist = 0;
  __syncthreads();
  const i64 cb = (i64)blockIdx.x * chunk, ce = std::min<i64>(cb + chunk, n);
  p2_run<P2TS / P2T>(S, keys, cb, ce, out);
}

// level 2: workgroup b splits level-1 bin b (keys[bin_base[b] .. bin_base[b+1]), padding included) into its
// windows (b << bits2) .. ; the top bin owns all windows up to nwin (at most 512).  Its output region starts
// at bin_base[b] + b * P2SLACK (room for the rounding of its windows); window starts go to win_off; the
// unused end of the region is marked "no entry".
__global__ __launch_bounds__(P2T) void p2_level2_kernel(const u32 *keys, const u64 *bin_base, int bits2, i64 nwin, u32 *out, u64 *win_off) {
  __shared__ Stage2 S;
  const i64 sb = (i64)bin_base[blockIdx.x], se = (i64)bin_base[blockIdx.x + 1];
  const u32 base = (u32)blockIdx.x << bits2;
  const i64 nsub = std::max<i64>(0, blockIdx.x == PBINS - 1 ? nwin - (i64)base : std::min<i64>((i64)1 << bits2, nwin - (i64)base));
  const u64 obase = (u64)sb + (u64)blockIdx.x * P2SLACK, onext = (u64)se + (u64)(blockIdx.x + 1) * P2SLACK;
  for (int b = threadIdx.x; b < PBINS; b += P2T) S.cnt[b] = 0;
  if (threadIdx.x == 0) S.nlist = 0;
  __syncthreads();
  for (i64 k0 = sb; k0 < se; k0 += 8 * P2T) {
    u32 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { i64 k = k0 + j * P2T + threadIdx.x; v[j] = k < se ? keys[k] : PAD; }
#pragma unroll
    for (int j = 0; j < 8; ++j) if (v[j] != PAD) atomicAdd(&S.cnt[p2_bin(v[j], WBITS, 0u)], 1u);   // values are relative to the bin
  }
  __syncthreads();
  u32 tot;
  const u32 hb = (S.cnt[threadIdx.x] + (P2U - 1)) / P2U * P2U;   // P2T == PBINS: one window per thread
  const u32 pre = p2_scan(hb, S.scratch, tot);
  S.cur[threadIdx.x] = obase + pre;
  S.cnt[threadIdx.x] = 0;
  if ((i64)threadIdx.x < nsub) win_off[(i64)base + threadIdx.x] = obase + pre;
  if (blockIdx.x == PBINS - 1 && threadIdx.x == 0) win_off[nwin] = onext;
  for (u64 k = obase + tot + threadIdx.x; k < onext; k += P2T) out[k] = PAD;
  __syncthreads();
  const Keys32 K2{keys, WBITS, 0, 0u};
  if (nsub > P2L2_A) p2_run<P2L2_EA>(S, K2, sb, se, out);
  else if (nsub > P2L2_B) p2_run<P2L2_EB>(S, K2, sb, se, out);
  else p2_run<1>(S, K2, sb, se, out);
}

// work items per window; all_windows: at least one each (the overwriting histogram must visit empty windows too)
__global__ __launch_bounds__(PSG_WG) void item_count_kernel(const u64 *off, i64 nwin, u64 *cnt, int all_windows) {
  i64 w = (i64)blockIdx.x * PSG_WG + threadIdx.x;
  if (w >= nwin) return;
  u64 c = off[w + 1] - off[w];
  u64 k = (c + CAP - 1) / CAP;
  cnt[w] = (all_windows && k == 0) ? 1 : k;
}

// overwriting histogram: a window with several work items is accumulated with atomics, so it starts from zero
__global__ __launch_bounds__(PSG_WG) void zero_multi_item_windows_kernel(const u64 *off, i64 nwin, i64 m, u32 *gap) {
  const i64 w = blockIdx.x;
  if ((i64)(off[w + 1] - off[w]) <= CAP) return;
  for (i64 k = (w << WBITS) + threadIdx.x; k < ((w + 1) << WBITS) && k <= m; k += PSG_WG) gap[k] = 0;
}

// one work item = up to CAP log entries of one window: LDS histogram, coalesced add to the gap array
// OVERWRITE: the gap array holds garbage on entry; single-item windows are stored, not added
// 512 threads: the 64 KiB histogram allows two workgroups per CU, and the kernel needs loads in flight
#define HWG 512
template <bool OVERWRITE>
__global__ __launch_bounds__(HWG) void hist_items_kernel(const u32 *keys, const u64 *off, const u64 *item_pref, const u64 *n_items, i64 nwin, i64 m, u32 *gap,
                                                           GapExcess ex, i64 slot_base) {
  __shared__ __attribute__((aligned(16))) u32 h[WSIZE / 2];   // counter c = half (c & 1) of word c >> 1; an item adds at most CAP < 2^16 to it
  __shared__ i64 s_w;
  i64 item = blockIdx.x;
  if (item >= (i64)*n_items) return;   // the grid is an upper bound (no host round trip for the item count)
  if (threadIdx.x == 0) {  // window of this item: last w with item_pref[w] <= item
    i64 lo = 0, hi = nwin;
    while (lo + 1 < hi) {
      i64 md = (lo + hi) >> 1;
      if ((i64)item_pref[md] <= item) lo = md; else hi = md;
    }
    s_w = lo;
  }
  for (int k = threadIdx.x; k < WSIZE / 8; k += HWG) ((uint4 *)h)[k] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  i64 w = s_w;
  i64 sub = item - (i64)item_pref[w];
  i64 beg = (i64)off[w] + sub * CAP, end = std::min<i64>(beg + CAP, (i64)off[w + 1]);
  bool single = (i64)(off[w + 1] - off[w]) <= CAP;
  for (i64 k0 = beg; k0 < end; k0 += 8 * HWG) {   // 8 independent loads in flight per thread
    u32 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { i64 k = k0 + j * HWG + threadIdx.x; v[j] = k < end ? keys[k] : PAD; }
#pragma unroll
    for (int j = 0; j < 8; ++j) if (v[j] != PAD) atomicAdd(&h[(v[j] & (WSIZE - 1)) >> 1], 1u << (16 * (v[j] & 1u)));
  }
  __syncthreads();
  i64 base = w << WBITS;
  bool vec = single && base + WSIZE - 1 <= m && ((uintptr_t)(gap + base) & 15) == 0;
  if (vec) {   // whole window inside the array: coalesced 16-byte read-modify-writes, 4 counters (2 LDS words) per access
    for (int k = threadIdx.x; k < WSIZE / 4; k += HWG) {
      const uint2 p = ((const uint2 *)h)[k];
      const uint4 c = make_uint4(p.x & 0xFFFFu, p.x >> 16, p.y & 0xFFFFu, p.y >> 16);
      const u64 j0 = (u64)(slot_base + base) + 4 * (u64)k;
      if (OVERWRITE && ex.bits >= 32) ((uint4 *)(gap + base))[k] = c;
      else if (OVERWRITE) {     // narrow counters (tests): a window count may already exceed the counter
        ((uint4 *)(gap + base))[k] = make_uint4(excess_add_owned(ex, j0, 0u, c.x), excess_add_owned(ex, j0 + 1, 0u, c.y),
                                                excess_add_owned(ex, j0 + 2, 0u, c.z), excess_add_owned(ex, j0 + 3, 0u, c.w));
      } else if (p.x | p.y) {
        uint4 *gp = (uint4 *)(gap + base) + k;
        uint4 g = *gp;
        g.x = excess_add_owned(ex, j0, g.x, c.x); g.y = excess_add_owned(ex, j0 + 1, g.y, c.y);
        g.z = excess_add_owned(ex, j0 + 2, g.z, c.z); g.w = excess_add_owned(ex, j0 + 3, g.w, c.w);   // a wrapping counter leaves a carry in the excess list
        *gp = g;
      }
    }
  } else {
    for (int k = threadIdx.x; k < WSIZE; k += HWG) {
      u32 c = (h[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
      i64 idx = base + k;
      if (OVERWRITE && single) { if (idx <= m) gap[idx] = excess_add_owned(ex, (u64)(slot_base + idx), 0u, c); }
      else if (c && idx <= m) {
        if (single) gap[idx] = excess_add_owned(ex, (u64)(slot_base + idx), gap[idx], c);
        else excess_add_atomic(ex, &gap[idx], (u64)(slot_base + idx), c);
      }
    }
  }
}

// KEYS: Keys32 / Keys40 with every field but `shift` filled in
template <class KEYS>
static int hist_job_launch(HistJob &J, KEYS K1, i64 nlog, i64 m, u32 *d_gap, bool overwrite, GapExcess ex, i64 slot_base) {
  static_assert(P2T == PBINS, "p2_level2_kernel maps one sub-bin to one thread");
  J.s = stream();
  J.ev_begin = event_acquire(); J.ev_end = event_acquire();
  (void)hipEventRecord(J.ev_begin, J.s);
  const i64 nwin = ((m + 1) + WSIZE - 1) >> WBITS;   // <= 2^18: up to 2^33 counters per call
  // level 2 only when there are > 512 windows: 511 level-1 bins of 2^bits2 windows + a top bin that takes the
  // rest (<= 512 windows).  m + 1 = 2^32 + 1 (a 4 GiB block) thus uses 257 full bins, not 512 half-empty ones.
  int bits2 = 0;
  if (nwin > PBINS) { bits2 = 1; while (nwin - (i64)(PBINS - 1) * ((i64)1 << bits2) > PBINS) ++bits2; }
  PSG_REQUIRE(bits2 <= 9, "gap histogram: more than 2^33 counters in one call");
  const int shift1 = WBITS + bits2;
  K1.shift = shift1;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int G = (int)std::min<i64>((i64)cus * 2, std::max<i64>(1, cdiv(nlog, P2TS)));
  const i64 chunk = cdiv(cdiv(nlog, G), P2TS) * P2TS;
  int rc;
  if ((rc = J.ovf.alloc(16 + 64))) return rc;       // stand-in excess area for callers without one (carries flag an error)
  PSG_HIP(hipMemsetAsync(J.ovf.p, 0, 16 + 64, J.s));
  PSG_HIP(hipMemsetAsync(J.ovf.as<u32>() + 3, 8, 1, J.s));   // header word 3 = capacity of the list behind it: 8 entries
  J.own_excess = ex.hdr == nullptr;
  if (J.own_excess) ex = GapExcess{J.ovf.as<u32>(), (u64 *)(J.ovf.as<u32>() + 4), 32};
  const i64 nwin_slots = (bits2 ? ((i64)PBINS << bits2) : PBINS) + PBINS + 2;
  const i64 cap1 = nlog + P2SLACK * G + 64;                 // level-1 output incl. the rounding of every (workgroup, bin) segment
  const i64 cap2 = cap1 + P2SLACK * (PBINS + 1);
  if ((rc = J.part1.alloc(cap1 * 4)) || (rc = J.counts.alloc((i64)G * PBINS * 4)) || (rc = J.off.alloc((i64)G * PBINS * 8)) ||
      (rc = J.bin_base.alloc((PBINS + 1) * 8)) || (rc = J.win_off.alloc((nwin_slots + 1) * 8)) || (rc = J.cnt.alloc(nwin_slots * 8)) || (rc = J.tot.alloc(8)))
    return rc;
  hipLaunchKernelGGL(p2_count_kernel<KEYS>, dim3(G), dim3(P2T), 0, J.s, K1, nlog, chunk, J.counts.as<u32>());
  hipLaunchKernelGGL(p2_offsets_kernel, dim3(1), dim3(PBINS), 0, J.s, J.counts.as<u32>(), G, J.off.as<u64>(), J.bin_base.as<u64>());
  hipLaunchKernelGGL(p2_scatter_kernel<KEYS>, dim3(G), dim3(P2T), 0, J.s, K1, nlog, chunk, J.off.as<u64>(), J.part1.as<u32>());
  PSG_HIP(hipGetLastError());
  const u32 *sorted = J.part1.as<u32>();
  const u64 *woff = J.bin_base.as<u64>();
  if (bits2) {
    if ((rc = J.part2.alloc(cap2 * 4))) return rc;
    hipLaunchKernelGGL(p2_level2_kernel, dim3(PBINS), dim3(P2T), 0, J.s, J.part1.as<u32>(), J.bin_base.as<u64>(), bits2, nwin, J.part2.as<u32>(), J.win_off.as<u64>());
    PSG_HIP(hipGetLastError());
    sorted = J.part2.as<u32>();
    woff = J.win_off.as<u64>();
  }
  hipLaunchKernelGGL(item_count_kernel, dim3((unsigned)cdiv(nwin, PSG_WG)), dim3(PSG_WG), 0, J.s, woff, nwin, J.cnt.as<u64>(), overwrite ? 1 : 0);
  if (overwrite) hipLaunchKernelGGL(zero_multi_item_windows_kernel, dim3((unsigned)nwin), dim3(PSG_WG), 0, J.s, woff, nwin, m, d_gap);
  PSG_HIP(hipGetLastError());
  if ((rc = scan_u64_inplace(J.cnt.as<u64>(), nwin, J.tot.as<u64>()))) return rc;
  // work items: one per window (overwrite) or per non-empty window, plus one per CAP entries; upper bound, the
  // kernel reads the exact count on the device
  const i64 items_max = nwin + (bits2 ? cap2 : cap1) / CAP + 1;
  if (overwrite) hipLaunchKernelGGL(hist_items_kernel<true>, dim3((unsigned)items_max), dim3(HWG), 0, J.s, sorted, woff, J.cnt.as<u64>(), J.tot.as<u64>(), nwin, m, d_gap, ex, slot_base);
  else hipLaunchKernelGGL(hist_items_kernel<false>, dim3((unsigned)items_max), dim3(HWG), 0, J.s, sorted, woff, J.cnt.as<u64>(), J.tot.as<u64>(), nwin, m, d_gap, ex, slot_base);
  PSG_HIP(hipGetLastError());
  (void)hipEventRecord(J.ev_end, J.s);
  J.active = true;
  return 0;
}

int psg::gap_hist_launch(HistJob &J, u32 *d_log, i64 nlog, i64 m, u32 *d_gap, bool overwrite, GapExcess ex, i64 slot_base) {
  return hist_job_launch(J, Keys32{d_log, 0, 1, 0u}, nlog, m, d_gap, overwrite, ex, slot_base);
}

int psg::gap_hist_wait(HistJob &J, double *ms) {
  if (!J.active) { if (ms) *ms = 0; return 0; }
  J.active = false;
  int h_ovf = 0, rc = 0;
  {
    StreamScope sc(J.s);
    rc = psg::copy_d2h(&h_ovf, J.ovf.p, 4);   // orders behind the job on its stream and waits for it
  }
  float f = 0;
  (void)hipEventElapsedTime(&f, J.ev_begin, J.ev_end);
  if (ms) *ms = f;
  event_release(J.ev_begin); event_release(J.ev_end);
  J.ev_begin = J.ev_end = nullptr;
  J.part1.alloc(16); J.part2.alloc(16);     // give the big buffers back to the pool
  if (rc) return rc;
  if (h_ovf && J.own_excess) { set_error("gap histogram: a counter wrapped and the caller gave no excess list"); return PSG_ECHECK; }
  return 0;
}

int psg::gap_hist_from_log(u32 *d_log, i64 nlog, i64 m, u32 *d_gap, double *ms, bool overwrite, GapExcess ex, i64 slot_base) {
  HistJob job;
  if (int rc = gap_hist_launch(job, d_log, nlog, m, d_gap, overwrite, ex, slot_base)) return rc;
  return gap_hist_wait(job, ms);
}

// Ranks of up to 40 bits (m >= 2^32 - 1).  One call of the partition covers 2^33 counters (2^18 windows), so the gap
// array is cut into slabs of 2^33 counters and the two-plane log is partitioned once per slab, entries of the other
// slabs reading as "no entry" (one slab for blocks up to 8 Gi symbols; a separate slab-split pass cost 14 ms per 2^31
// entries, more than a level of the partition itself).  PSG_LOG_SLAB_SHIFT makes the slabs small so that tests
// cross several of them.
bool psg::gap_hist_wide_one_slab(i64 m) {
  int slab_shift = 33;
  if (const char *e = getenv("PSG_LOG_SLAB_SHIFT")) { int v = atoi(e); if (v >= 8 && v <= 33) slab_shift = v; }
  return (m >> slab_shift) == 0;
}
int psg::gap_hist_wide_launch(HistJob &job, const u32 *log_lo, const u8 *log_hi, i64 nlog, i64 m, u32 *d_gap, bool overwrite, GapExcess ex) {
  const Keys40 K{log_lo, log_hi, 0, (u64)0, (u64)m + 1};
  return hist_job_launch(job, K, nlog, m, d_gap, overwrite, ex, 0);
}

int psg::gap_hist_from_wide_log(DevBuf &log_lo, DevBuf &log_hi, i64 nlog, i64 m, u32 *d_gap, double *ms, bool overwrite, GapExcess ex) {
  int slab_shift = 33;
  if (const char *e = getenv("PSG_LOG_SLAB_SHIFT")) { int v = atoi(e); if (v >= 8 && v <= 33) slab_shift = v; }
  const i64 nslab = (m >> slab_shift) + 1;
  double total = 0;
  for (i64 sl = 0; sl < nslab; ++sl) {
    const i64 base = sl << slab_shift, ms_ = std::min<i64>(((i64)1 << slab_shift) - 1, m - base);   // counters [0, ms_] of this slab
    HistJob job;
    const Keys40 K{log_lo.as<u32>(), log_hi.as<u8>(), 0, (u64)base, (u64)base + (u64)ms_ + 1};
    if (int rc = hist_job_launch(job, K, nlog, ms_, d_gap + base, overwrite, ex, base)) return rc;
    double t = 0;
    if (int rc = gap_hist_wait(job, &t)) return rc;
    total += t;
  }
  log_lo.alloc(16); log_hi.alloc(16);
  if (ms) *ms = total;
  return 0;
}

// ---------------------------------------------------------------------------------------
// multi-GPU building blocks (tail-sharded pass with a rank-log all-to-all)
// ---------------------------------------------------------------------------------------
static void level1_geometry(i64 m, int *shift1, int *bits2) {
  const i64 nwin = ((m + 1) + WSIZE - 1) >> WBITS;
  int bits_total = 0;
  while (((i64)1 << bits_total) < nwin) ++bits_total;
  *bits2 = bits_total > 9 ? bits_total - 9 : 0;
  *shift1 = WBITS + *bits2;
}

// Split the valid entries of a rank log into `nparts` contiguous value ranges (part p = level-1
// bins [nb*p/nparts, nb*(p+1)/nparts), nb = bins in use): d_out receives the parts back to back, h_offsets[p] is the
// start of part p in d_out (h_offsets[nparts] = number of valid entries), h_value_bounds[p] the
// first value of part p.  The bounds depend only on (m, nparts), so every rank computes the same.
extern "C" int psg_log_partition(const uint32_t *d_log, int64_t nlog, int64_t m, int nparts, uint32_t *d_out,
                                 int64_t *h_offsets, int64_t *h_value_bounds) {
  PSG_REQUIRE(d_out && h_offsets && h_value_bounds && nlog >= 0 && m >= 0 && m < 0xFFFFFFFFll && nparts >= 1 && nparts <= PBINS,
              "psg_log_partition");
  int shift1, bits2;
  level1_geometry(m, &shift1, &bits2);
  const i64 nb = ((m + 1) + ((i64)1 << shift1) - 1) >> shift1;   // level-1 bins that can hold a value (<= 512)
  for (int p = 0; p <= nparts; ++p) {
    i64 v = (nb * p / nparts) << shift1;
    h_value_bounds[p] = p == nparts ? std::max<i64>(v, m + 1) : v;
  }
  if (nlog == 0) { for (int p = 0; p <= nparts; ++p) h_offsets[p] = 0; return 0; }
  PSG_REQUIRE(d_log, "psg_log_partition: log required");
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int G = (int)std::min<i64>((i64)cus * 2, std::max<i64>(1, cdiv(nlog, PT)));
  DevBuf counts, off, bin_base;
  int rc;
  if ((rc = counts.alloc((i64)G * PBINS * 4)) || (rc = off.alloc((i64)G * PBINS * 8)) || (rc = bin_base.alloc((PBINS + 1) * 8))) return rc;
  hipLaunchKernelGGL(part_count_kernel, dim3(G), dim3(PNT), 0, stream(), d_log, nlog, shift1, counts.as<u32>());
  hipLaunchKernelGGL(part_offsets_kernel, dim3(1), dim3(PBINS), 0, stream(), counts.as<u32>(), G, off.as<u64>(), bin_base.as<u64>());
  hipLaunchKernelGGL(part_scatter_kernel, dim3(G), dim3(PNT), 0, stream(), d_log, nlog, shift1, off.as<u64>(), d_out);
  PSG_HIP(hipGetLastError());
  u64 bb[PBINS + 1];
  if (int rc_ = psg::copy_d2h(bb, bin_base.p, (size_t)(sizeof bb))) return rc_;
  PSG_HIP(psg::sync_stream());
  for (int p = 0; p <= nparts; ++p) h_offsets[p] = (i64)bb[p == nparts ? PBINS : nb * p / nparts];
  return 0;
}

__global__ __launch_bounds__(PSG_WG) void sub_base_kernel(u32 *log, i64 n, u32 base, u32 count) {
  i64 k = (i64)blockIdx.x * PSG_WG + threadIdx.x;
  if (k >= n) return;
  u32 v = log[k];
  log[k] = (v != PAD && v >= base && v - base < count) ? v - base : PAD;
}

// gap_slice[v - value_base] += #{entries equal to v} for v in [value_base, value_base + count);
// other entries are ignored; the log is clobbered.
extern "C" int psg_gap_hist(uint32_t *d_log, int64_t nlog, int64_t value_base, int64_t count, uint32_t *d_gap_slice) {
  PSG_REQUIRE(d_gap_slice && nlog >= 0 && value_base >= 0 && count >= 0 && value_base + count <= 0xFFFFFFFFll, "psg_gap_hist");
  if (nlog == 0 || count == 0) return 0;
  PSG_REQUIRE(d_log, "psg_gap_hist: log required");
  hipLaunchKernelGGL(sub_base_kernel, dim3((unsigned)cdiv(nlog, PSG_WG)), dim3(PSG_WG), 0, stream(), d_log, nlog, (u32)value_base, (u32)count);
  PSG_HIP(hipGetLastError());
  return psg::gap_hist_from_log(d_log, nlog, count - 1, d_gap_slice, nullptr, false);
}

// test entry (include/psascan_amd_extras.h): histogram an explicit rank log
extern "C" int psgx_gap_hist(uint32_t *d_log, int64_t nlog, int64_t m, uint32_t *d_gap) {
  PSG_REQUIRE(d_log && d_gap && nlog >= 0 && m >= 0 && m < 0xFFFFFFFFll, "psgx_gap_hist");
  if (nlog == 0) return 0;
  return psg::gap_hist_from_log(d_log, nlog, m, d_gap, nullptr, false);
}
