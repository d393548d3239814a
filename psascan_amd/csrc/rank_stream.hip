// rank_stream.hip -- K1 (rank structure build) and K2 (streaming gap kernel) for gfx950.
//
// Default rank layout: symbol-major entries, one 8/16-byte load per query (rank_sm.hpp), for blocks of any
// size (counts are kept relative to superblocks of 2^31 positions, the bases live in the LDS table of the
// pass).  Fallback when that structure does not fit the HBM budget:
// "interleaved blocks" (HBM-resident): the BWT is cut into blocks of B data
// bytes; each block is stored as  [CNT x u32 counters][B data bytes]  (STRIDE = 4*CNT+B).
// counter[code] = #occurrences of that symbol before the block, relative to the enclosing
// superblock (2^SB_SHIFT blocks); the superblock bases (u64) are folded with the C array of
// the pass into one LDS table, so a query is: 1 LDS read + 1 counter sector + 1 data sector.
//   sigma_eff <= 4  : CNT=4,  B=48  -> one 64-byte sector per query
//   sigma_eff <= 16 : CNT=16, B=64  -> one 128-byte line per query
//   otherwise       : CNT=256,B=64/128/256 (identity code)
// Semantics restated from rank4n<>::rank (reference include/rank.hpp:566-568): only the
// semantics -- the reference's 16 MiB/1 MiB cache-oriented trunks are not reproduced.
#include "dev_common.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace psg;

#ifndef PSG_STREAM_MIN_WAVES
#define PSG_STREAM_MIN_WAVES 1      // waves per SIMD requested from the register allocator (B <= 64 layouts)
#endif
#define SB_SHIFT_DEFAULT 24         // interleaved blocks: blocks per superblock = 2^24 (PSG_BLOCK_SB_SHIFT: tests)
#define SEG_BLOCKS 64               // blocks per build segment (one workgroup)
#define GROUP_SEGS 256              // segments per scan group
#define CODE_SHIFT 56
#define VAL_MASK ((1ull << CODE_SHIFT) - 1ull)

struct psg_rank {
  i64 m = 0;
  int cnt = 0, B = 0, stride = 0;
  i64 nblk = 0, nseg = 0;
  int nsb = 0;
  int sb_shift = 0;               // interleaved blocks: log2(blocks per superblock)
  i64 sb_size = 0;                // symbol-major: positions per superblock (a multiple of SM_SEG = 3072: not a power of two)
  u8 *d_blocks = nullptr;
  i64 blocks_bytes = 0;
  u64 *d_sb = nullptr;            // [nsb][cnt]
  std::vector<u64> h_sb;          // host copy
  u8 *d_aux = nullptr;            // symbol-major layout: overflow pool
  u64 t2[256] = {0};              // symbol-major layout: per-symbol descriptor (offset/16 | mode << 62)
  u8 code[256];                   // symbol -> code (0xFF = absent)
  i64 count[256];                 // occurrences per symbol (m_count, rank.hpp:112)
};

// ---------------------------------------------------------------------------------------
// device: count bytes equal to c among the first `off` bytes of a B-byte data area
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ u32 swar_eq_mask(u32 w, u32 c4) {
  u32 x = w ^ c4;
  u32 t = ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu;
  return ~t;  // 0x80 in every byte of w equal to c
}

template <int B> __device__ __forceinline__ u32 count_prefix(const u8 *data, u32 c, int off) {
  const uint4 *p = (const uint4 *)data;
  u32 c4 = c * 0x01010101u;
  u32 acc = 0;
#pragma unroll
  for (int q = 0; q < B / 16; ++q) {
    uint4 v = p[q];
    u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int rem = off - (q * 16 + k * 4);  // valid bytes in this word (may be <=0 or >=4)
      u32 keep = rem >= 4 ? 0x80808080u : (rem <= 0 ? 0u : (0x80808080u >> (32 - 8 * rem)));
      acc += __popc(swar_eq_mask(w[k], c4) & keep);
    }
  }
  return acc;
}

// bytes equal to c among bytes [lo, hi) of an NCH*16-byte region (lo, hi relative to it)
template <int NCH> __device__ __forceinline__ u32 count_range(const u8 *data, u32 c, int lo, int hi) {
  const uint4 *p = (const uint4 *)data;
  u32 c4 = c * 0x01010101u;
  u32 acc = 0;
#pragma unroll
  for (int q = 0; q < NCH; ++q) {
    uint4 v = p[q];
    u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int o = q * 16 + k * 4;
      int rh = hi - o, rl = lo - o;
      u32 kh = rh >= 4 ? 0x80808080u : (rh <= 0 ? 0u : (0x80808080u >> (32 - 8 * rh)));
      u32 kl = rl >= 4 ? 0x80808080u : (rl <= 0 ? 0u : (0x80808080u >> (32 - 8 * rl)));
      acc += __popc(swar_eq_mask(w[k], c4) & kh & ~kl);
    }
  }
  return acc;
}

// The block counter holds the count at the block's MIDPOINT (not its start): a query then only
// needs the half of the data area that lies between the midpoint and i -- at most MID bytes, i.e.
// 2 dwordx4 loads for B = 64 instead of 4.  Measured on MI355X (tools/membench): a step of
// 1 dword + 2 dwordx4 dependent random loads runs at 25 G steps/s, 1 dword + 4 dwordx4 (same two
// sectors!) only at 15.7 G steps/s -- the load count per step, not the sector count, was the limit.
// full-block count (first `off` bytes) and first-part count (first min(off, MID) bytes) in one pass
template <int B, int MID> __device__ __forceinline__ u32 count_two(const u8 *data, u32 c, int off, u32 &first) {
  const uint4 *p = (const uint4 *)data;
  u32 c4 = c * 0x01010101u;
  u32 acc = 0, accf = 0;
#pragma unroll
  for (int q = 0; q < B / 16; ++q) {
    uint4 v = p[q];
    u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int rem = off - (q * 16 + k * 4);
      u32 keep = rem >= 4 ? 0x80808080u : (rem <= 0 ? 0u : (0x80808080u >> (32 - 8 * rem)));
      u32 n = __popc(swar_eq_mask(w[k], c4) & keep);
      acc += n;
      if (q * 16 + k * 4 < MID) accf += n;   // compile-time condition (MID is a multiple of 4)
    }
  }
  first = accf;
  return acc;
}

template <int CNT, int B> struct RankView {   // CNT == 0: symbol-major layout (rank_sm.hpp)
  const u8 *blocks;
  i64 m;
  const u8 *aux;                                     // symbol-major layout: overflow bitmap pool
  int sb_shift;                                      // interleaved blocks: log2(blocks per superblock)
  i64 sb_size;                                       // symbol-major: positions per superblock
  static constexpr int STRIDE = 4 * CNT + B;
  static constexpr int MID = B == 48 ? 32 : B / 2;   // multiple of 16, >= B - MID
};

#include "rank_sm.hpp"

// One LF step = C[c] + rank(i, c) (before the delta / gt corrections), split in two halves so that a
// lane can have the loads of SEVERAL chains in flight before it needs any of the data:
//   rank_issue  : LDS lookups, address computation, issue of the global loads
//   rank_finish : arithmetic on the loaded registers
// T1: LDS table [nsb][256] of (C[c] + superblock base) | code << 56 ; tot: LDS [256] (+ [256] descriptors).
template <int CNT, int B> struct RankReq {
  i64 res;       // kind == 0: the result; kind == 1: C[c] (+ superblock base)
  u32 kind;
  // symbol-major
  u64 t2;
  u32 off;
  uint4 e;
  // interleaved blocks
  u32 ctr;
  int lo, hi;
  bool upper;
  uint4 d[CNT == 0 ? 1 : RankView<CNT, B>::MID / 16];
};

template <int CNT, int B>
__device__ __forceinline__ void rank_issue(const RankView<CNT, B> &R, const u64 *T1, const u64 *tot, i64 i, u32 c, RankReq<CNT, B> &q) {
  // symbol-major layout: the entry's count is relative to the superblock of i, whose base is folded into T1
  i64 sbi = 0;
  if (CNT == 0 && i > 0 && i < R.m)
    for (i64 t = R.sb_size; t <= i; t += R.sb_size) ++sbi;   // (one to three superblocks in practice: blocks of up to 2^32+ symbols)
  u64 e0 = T1[sbi * 256 + c];
  u32 code = (u32)(e0 >> CODE_SHIFT);
  i64 Cc = (i64)(e0 & VAL_MASK);
  q.kind = 0;
  q.res = Cc;
  if (i <= 0) return;
  if (i >= R.m) { q.res = Cc + (i64)tot[c]; return; }
  if constexpr (CNT == 0) {   // symbol-major: one 16-byte load (descriptors live behind tot[] in LDS)
    u64 t2 = tot[256 + c];
    u32 mode = (u32)(t2 >> SM_MODE_SHIFT);
    if (mode == SM_ABSENT) return;
    const uint4 *E = (const uint4 *)R.blocks + (t2 & SM_OFF_MASK);
    bool bm = mode == SM_BITMAP;
    const u32 b96 = (u32)(((u64)(u32)(i >> 5) * 0xAAAAAAABull) >> 33);   // i / 96 (i < 2^37)
    if (mode == SM_LIST8) { uint2 x = ((const uint2 *)E)[i >> 8]; q.e = make_uint4(x.x, x.y, 0u, 0u); }
    else q.e = E[bm ? (i64)b96 : (i >> 8)];
    q.off = bm ? (u32)(i - (i64)b96 * SM_BW) : ((u32)i & 255u);
    q.t2 = t2;
    q.kind = 1;
  } else {
    if (CNT < 256 && code == 0xFFu) return;
    i64 blk = i / B;
    int off = (int)(i - blk * B);
    const u8 *p = R.blocks + blk * (i64)RankView<CNT, B>::STRIDE;
    constexpr int MID = RankView<CNT, B>::MID;
    q.ctr = *(const u32 *)(p + 4 * code);
    q.upper = off >= MID;
    i64 valid = R.m - blk * B;                     // symbols stored in this block (last block may be short)
    int lim = valid < MID ? (int)valid : MID;
    q.lo = q.upper ? 0 : off;
    q.hi = q.upper ? off - MID : lim;
    const uint4 *dp = (const uint4 *)(p + 4 * CNT + (q.upper ? MID : 0));
#pragma unroll
    for (int k = 0; k < MID / 16; ++k) q.d[k] = dp[k];
    i64 sb = blk >> R.sb_shift;
    q.res = sb ? (i64)(T1[sb * 256 + c] & VAL_MASK) : Cc;
    q.kind = 1;
  }
}

template <int CNT, int B>
__device__ __forceinline__ i64 rank_finish(const RankView<CNT, B> &R, u32 c, const RankReq<CNT, B> &q) {
  if (q.kind == 0) return q.res;
  if constexpr (CNT == 0) {
    return q.res + (i64)sm_rank_entry(q.e, R.aux, q.t2, q.off);
  } else {
    u32 cnt = count_range<RankView<CNT, B>::MID / 16>((const u8 *)q.d, c, q.lo, q.hi);
    return q.res + q.ctr + (q.upper ? (i64)cnt : -(i64)cnt);
  }
}

template <int CNT, int B>
__device__ __forceinline__ i64 lf_core(const RankView<CNT, B> &R, const u64 *T1, const u64 *tot, i64 i, u32 c) {
  RankReq<CNT, B> q;
  rank_issue<CNT, B>(R, T1, tot, i, c, q);
  return rank_finish<CNT, B>(R, c, q);
}

// ---------------------------------------------------------------------------------------
// K1 kernels
// ---------------------------------------------------------------------------------------
// (16 copies of the histogram, copy = lane & 15, bin-major so that the copies of a bin lie in different banks: on
// natural-language BWTs a few symbols take most of the increments and 64 lanes hitting one LDS word serialise)
__global__ __launch_bounds__(PSG_WG) void hist256_kernel(const u8 *bwt, i64 m, unsigned long long *out) {
  __shared__ u32 h[256 * 16];
  for (int k = threadIdx.x; k < 256 * 16; k += PSG_WG) h[k] = 0;
  __syncthreads();
  const u32 cp = threadIdx.x & 15u;
  for (i64 k = ((i64)blockIdx.x * PSG_WG + threadIdx.x) * 16; k < m; k += (i64)gridDim.x * PSG_WG * 16) {
    if (k + 16 <= m) {
      uint4 v = *(const uint4 *)(bwt + k);
      u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int b = 0; b < 4; ++b) atomicAdd(&h[(((w[q] >> (8 * b)) & 255) << 4) | cp], 1u);
    } else {
      for (i64 j = k; j < m; ++j) atomicAdd(&h[((u32)bwt[j] << 4) | cp], 1u);
    }
  }
  __syncthreads();
  u32 tot = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) tot += h[(threadIdx.x << 4) | ((k + threadIdx.x) & 15)];
  if (tot) atomicAdd(&out[threadIdx.x], (unsigned long long)tot);
}

template <int CNT, int B>
__global__ __launch_bounds__(PSG_WG) void seg_hist_kernel(const u8 *bwt, i64 m, const u8 *code_g, u32 *seg_cnt) {
  constexpr int SEGSYM = SEG_BLOCKS * B;
  constexpr int CP = CNT <= 256 ? 8 : 1;            // copies of the histogram (see hist256_kernel)
  __shared__ u32 h[CNT * CP];
  __shared__ u8 code[256];
  code[threadIdx.x] = code_g[threadIdx.x];
  for (int k = threadIdx.x; k < CNT * CP; k += PSG_WG) h[k] = 0;
  __syncthreads();
  const u32 cp = threadIdx.x & (CP - 1);
  i64 base = (i64)blockIdx.x * SEGSYM;
  if (SEGSYM == 16 * PSG_WG && base + SEGSYM <= m && ((uintptr_t)bwt & 15) == 0) {   // whole segment: one 16-byte load per thread
    const uint4 v = ((const uint4 *)(bwt + base))[threadIdx.x];
    const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      u32 cd = code[(w[k >> 2] >> (8 * (k & 3))) & 255u];
      if (cd != 0xFFu) atomicAdd(&h[cd * CP + cp], 1u);
    }
  } else {
    for (int k = threadIdx.x; k < SEGSYM; k += PSG_WG) {
      i64 p = base + k;
      if (p < m) {
        u32 cd = code[bwt[p]];
        if (cd != 0xFFu) atomicAdd(&h[cd * CP + cp], 1u);
      }
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < CNT; k += PSG_WG) {
    u32 tot = 0;
#pragma unroll
    for (int q = 0; q < CP; ++q) tot += h[k * CP + ((q + k) & (CP - 1))];
    seg_cnt[(i64)blockIdx.x * CNT + k] = tot;
  }
}

// thread per (group, col): serial exclusive prefix over the group's segments (in place)
__global__ __launch_bounds__(PSG_WG) void group_prefix_kernel(u32 *seg_cnt, i64 nseg, int cnt, u64 *group_sum, i64 ngroups) {
  i64 t = (i64)blockIdx.x * PSG_WG + threadIdx.x;
  if (t >= ngroups * cnt) return;
  i64 g = t / cnt;
  int col = (int)(t - g * cnt);
  i64 r0 = g * GROUP_SEGS, r1 = std::min<i64>(r0 + GROUP_SEGS, nseg);
  u64 run = 0;
  for (i64 r = r0; r < r1; ++r) {
    u32 v = seg_cnt[r * cnt + col];
    seg_cnt[r * cnt + col] = (u32)run;
    run += v;
  }
  group_sum[g * cnt + col] = run;
}

// one workgroup; thread per col: serial exclusive scan over groups (in place)
__global__ __launch_bounds__(PSG_WG) void group_scan_kernel(u64 *group_sum, i64 ngroups, int cnt) {
  int col = threadIdx.x;
  if (col >= cnt) return;
  u64 run = 0;
  for (i64 g = 0; g < ngroups; ++g) {
    u64 v = group_sum[g * cnt + col];
    group_sum[g * cnt + col] = run;
    run += v;
  }
}

// One workgroup per segment of 64 blocks.  Per-(block, code) occurrence counts are built with LDS
// atomics -- one per symbol, instead of every code scanning every byte -- in two passes of 32
// blocks; each u32 holds the count of the block's first part (< MID, low half) and of the rest
// (high half), so the midpoint counter and the running prefix come out of one array.
template <int CNT, int B>
__global__ __launch_bounds__(PSG_WG) void rank_fill_kernel(const u8 *bwt, i64 m, const u8 *code_g, const u32 *seg_pref,
                                                             const u64 *group_base, u8 *blocks, i64 nblk, int sb_shift) {
  constexpr int SEGSYM = SEG_BLOCKS * B;
  constexpr int STRIDE = 4 * CNT + B;
  constexpr int MID = RankView<CNT, B>::MID;
  constexpr int HB = SEG_BLOCKS / 2;
  __shared__ __attribute__((aligned(16))) u8 sym[SEGSYM];
  __shared__ u32 cnt[HB * CNT];
  __shared__ u8 code[256];
  const i64 seg = blockIdx.x;
  const i64 base = seg * SEGSYM;
  code[threadIdx.x] = code_g[threadIdx.x];
  if (base + SEGSYM <= m && ((uintptr_t)bwt & 15) == 0) {
    for (int k = threadIdx.x; k < SEGSYM / 16; k += PSG_WG) ((uint4 *)sym)[k] = ((const uint4 *)(bwt + base))[k];
  } else {
    for (int k = threadIdx.x; k < SEGSYM; k += PSG_WG) sym[k] = base + k < m ? bwt[base + k] : 0;
  }
  // running count (relative to the superblock) at the start of the segment, for the code of this thread
  const i64 g = seg / GROUP_SEGS;
  const i64 sb = (seg * SEG_BLOCKS) >> sb_shift;
  const i64 sb_group = (sb << sb_shift) / SEG_BLOCKS / GROUP_SEGS;
  u32 run = 0;
  if (threadIdx.x < CNT) run = (u32)(group_base[g * CNT + threadIdx.x] + seg_pref[seg * CNT + threadIdx.x] - group_base[sb_group * CNT + threadIdx.x]);
  for (int half = 0; half < 2; ++half) {
    for (int k = threadIdx.x; k < HB * CNT; k += PSG_WG) cnt[k] = 0;
    __syncthreads();
    for (int p = half * HB * B + threadIdx.x; p < (half + 1) * HB * B; p += PSG_WG) {
      if (base + p < m) {
        u32 cd = code[sym[p]];
        if (cd != 0xFFu || CNT == 256) {
          int blk = p / B, off = p - blk * B;
          atomicAdd(&cnt[(blk - half * HB) * CNT + cd], off < MID ? 1u : 0x10000u);
        }
      }
    }
    __syncthreads();
    if (threadIdx.x < CNT) {
      for (int blk = 0; blk < HB; ++blk) {
        u32 v = cnt[blk * CNT + threadIdx.x];
        i64 gb = seg * SEG_BLOCKS + half * HB + blk;
        if (gb < nblk) *(u32 *)(blocks + gb * STRIDE + 4 * threadIdx.x) = run + (v & 0xFFFFu);   // count at the block midpoint
        run += (v & 0xFFFFu) + (v >> 16);
      }
    }
    __syncthreads();
  }
  // data bytes
  for (int k = threadIdx.x; k < SEGSYM / 4; k += PSG_WG) {
    int blk = (k * 4) / B, o = (k * 4) % B;
    i64 gb = seg * SEG_BLOCKS + blk;
    if (gb >= nblk) continue;
    *(u32 *)(blocks + gb * STRIDE + 4 * CNT + o) = *(const u32 *)(sym + k * 4);
  }
}

// per-segment symbol histograms of the symbol-major build (segments of SM_SEG positions)
__global__ __launch_bounds__(PSG_WG) void sm_seg_hist_kernel(const u8 *bwt, i64 m, u32 *seg_cnt) {
  constexpr int CP = 8;                              // copies of the histogram (see hist256_kernel)
  __shared__ u32 h[256 * CP];
  for (int k = threadIdx.x; k < 256 * CP; k += PSG_WG) h[k] = 0;
  __syncthreads();
  const u32 cp = threadIdx.x & (CP - 1);
  const i64 base = (i64)blockIdx.x * SM_SEG;
  if (base + SM_SEG <= m && ((uintptr_t)bwt & 15) == 0) {
    if (threadIdx.x < SM_SEG / 16) {
      const uint4 v = ((const uint4 *)(bwt + base))[threadIdx.x];
      const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 16; ++k) atomicAdd(&h[((w[k >> 2] >> (8 * (k & 3))) & 255u) * CP + cp], 1u);
    }
  } else {
    for (int k = threadIdx.x; k < SM_SEG; k += PSG_WG) if (base + k < m) atomicAdd(&h[(u32)bwt[base + k] * CP + cp], 1u);
  }
  __syncthreads();
  u32 tot = 0;
#pragma unroll
  for (int q = 0; q < CP; ++q) tot += h[threadIdx.x * CP + ((q + threadIdx.x) & (CP - 1))];
  seg_cnt[(i64)blockIdx.x * 256 + threadIdx.x] = tot;
}

// LDS table loader shared by the query / warm-up / stream kernels
__device__ __forceinline__ void load_tables(u64 *lds, const u64 *g_T1, const u64 *g_tot, int nsb) {
  for (int k = threadIdx.x; k < nsb * 256; k += blockDim.x) lds[k] = g_T1[k];
  for (int k = threadIdx.x; k < 512; k += blockDim.x) lds[nsb * 256 + k] = g_tot[k];   // tot[256] + symbol-major descriptors[256]
  __syncthreads();
}

template <int CNT, int B>
__global__ __launch_bounds__(PSG_WG) void rank_query_kernel(RankView<CNT, B> R, const u64 *g_T1, const u64 *g_tot, int nsb,
                                                              const i64 *qi, const u8 *qc, i64 nq, i64 *out) {
  extern __shared__ u64 lds[];
  load_tables(lds, g_T1, g_tot, nsb);
  i64 k = (i64)blockIdx.x * PSG_WG + threadIdx.x;
  if (k >= nq) return;
  out[k] = lf_core<CNT, B>(R, lds, lds + nsb * 256, qi[k], qc[k]);
}

// ---------------------------------------------------------------------------------------
// K2: stream kernel.  One backward-search chain per lane.
//   chain k covers steps u in [k*L, min((k+1)*L, T));  step u handles text position
//   j = te - u: consumes symbol tail[T-1-u], gt_in bit u, emits gt_out bit u.
// ---------------------------------------------------------------------------------------
struct StreamParams {
  const u8 *tail;
  i64 T;              // tail_len + right context
  i64 ctx;            // right context length (multiple of 64): steps u < ctx are not streamed
  const u32 *gt_in;
  u32 *gt_out;
  u32 *gap;
  i64 i0;
  u32 last;
  i64 L;
  i64 nchains;        // entries in this launch
  const i64 *list;    // chain ids (null = identity)
  const i64 *init;    // [K] start rank of chain k
  i64 *fin;           // [K] rank after the chain's last step
  const u64 *g_T1;
  const u64 *g_tot;
  int nsb;
  int *ovf_flag;
  u32 *log;           // MODE 2/3: rank log (low 32 bits), entry ((step >> 2) * K + chain) * 4 + (step & 3)
  i64 K;              // total number of chains (log row length)
  u32 *log_hi;        // MODE 3: bits 32..39 of the ranks, one byte per entry at the same index (0xFF + low word PAD = no entry)
  psg::GapExcess ex;  // MODE 1: where wrapping counters leave their carries
  // batched passes (stream_batch_kernel: many small passes over ONE rank structure built on the concatenation of their
  // blocks): the block of this pass starts at position rank_off of the structure, its gap slots at gap_base of the
  // shared gap array; 0 / 0 for an ordinary pass
  i64 rank_off;
  i64 gap_base;
};

// MODE 0: atomicAdd on the gap counters (a fresh array of 32-bit counters and a tail below 2^32: nothing can wrap);
// MODE 1: the same with carries into the excess list (update.hpp:88-96);
// MODE 2: no atomics -- the ranks are logged (coalesced dwordx4 stores) and histogrammed afterwards
//         (gap_hist.hip).
// MODE 3: MODE 2 for blocks of >= 2^32 - 1 symbols: a second plane takes bits 32..39 of every rank (one dword
//         store per 4 steps); the log is split into slabs of 2^31 counters before the histogram.
// CPL = chains per lane: the steps of CPL independent chains are interleaved so that a lane has CPL
//       rank loads in flight (memory-level parallelism beyond what 8 waves per SIMD give).
struct Chain {
  i64 i, k, u0, u1, w;
  u64 thi, tlo;       // current 16-byte text chunk, next byte on top
  int tcnt;           // bytes left in it
  int nbuf;           // chunks still buffered in b0..b2 (b0 is next)
  uint4 b0, b1, b2;   // rest of the current 64-byte text block: ONE memory access per 64 steps (every private
  const uint4 *bp;    // 16-byte load of a chain costs a full random HBM access, ~like a rank load)
  uintptr_t last_addr;
  u32 gin, gin_next, gout;
  u32 hiacc;          // MODE 3: high bytes of the last (t & 3) ranks
};

// the 64-byte text block at bp; its 16-byte chunks are consumed in descending order starting with chunk
// `first`: cur = that chunk, b0.. = the following ones
struct TextBlock { uint4 cur, b0, b1, b2; };
__device__ __forceinline__ TextBlock load_text_block(const uint4 *bp, int first) {
  const uint4 z = make_uint4(0, 0, 0, 0);
  uint4 x0 = gload(bp), x1 = z, x2 = z, x3 = z;
  if (first >= 1) x1 = gload(bp + 1);
  if (first >= 2) x2 = gload(bp + 2);
  if (first >= 3) x3 = gload(bp + 3);
  TextBlock t;
  t.cur = first == 3 ? x3 : (first == 2 ? x2 : (first == 1 ? x1 : x0));
  t.b0 = first == 3 ? x2 : (first == 2 ? x1 : x0);
  t.b1 = first == 3 ? x1 : x0;
  t.b2 = x0;
  return t;
}
#define PSG_TEXT_BLOCK(c, first_)                                \
  do {                                                           \
    const int f_ = (first_);                                     \
    const TextBlock tb_ = load_text_block((c).bp, f_);           \
    (c).b0 = tb_.b0; (c).b1 = tb_.b1; (c).b2 = tb_.b2;           \
    (c).nbuf = f_;                                               \
    (c).tlo = (u64)tb_.cur.x | ((u64)tb_.cur.y << 32);           \
    (c).thi = (u64)tb_.cur.z | ((u64)tb_.cur.w << 32);           \
  } while (0)

// The body of the stream kernel: lane `gid` of the pass runs its CPL chains.  BATCH: the pass is one of many over a
// shared rank structure (P.rank_off, P.gap_base); T1 then holds C[c] - rank(rank_off, c) modulo 2^56.
template <int CNT, int B, int MODE, int CPL, bool BATCH>
__device__ __forceinline__ void stream_body(const RankView<CNT, B> &R, const StreamParams &P, const u64 *T1, const u64 *tot, u32 *lstage, u32 *gstage, i64 gid) {
  const bool gt16 = P.gt_out && ((uintptr_t)P.gt_out & 15) == 0 && (P.L & 127) == 0;
  const i64 nlanes = (P.nchains + CPL - 1) / CPL;
  if (gid >= nlanes) return;
  Chain S[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    Chain &c = S[q];
    i64 idx = gid + q * nlanes;
    bool act = idx < P.nchains;
    c.k = act ? (P.list ? P.list[idx] : idx) : -1;
    c.u0 = act ? P.ctx + c.k * P.L : 0;
    c.u1 = act ? std::min<i64>(c.u0 + P.L, P.T) : 0;
    c.i = act ? P.init[c.k] : 0;
    c.w = c.u0 >> 5;
    c.gin = 0; c.gin_next = 0; c.gout = 0; c.hiacc = 0;
    c.tcnt = 1; c.thi = 0; c.tlo = 0; c.bp = nullptr; c.last_addr = 0; c.nbuf = 0;
    c.b0 = make_uint4(0, 0, 0, 0); c.b1 = c.b0; c.b2 = c.b0;
    if (act) {
      // text cursor: descending bytes starting at tail[T-1-u0], consumed from a 128-bit shift
      // register (no dynamically indexed registers: hipcc would spill those to scratch memory)
      uintptr_t addr = (uintptr_t)(P.tail + (P.T - 1 - c.u0));
      c.last_addr = (uintptr_t)(P.tail + (P.T - c.u1));   // address of the last byte this chain needs
      c.bp = (const uint4 *)(addr & ~(uintptr_t)63);
      c.tcnt = (int)(addr & 15) + 1;                       // bytes left in the current chunk
      PSG_TEXT_BLOCK(c, (int)((addr >> 4) & 3));
      int sh = (16 - c.tcnt) * 8;                          // bring byte (tcnt-1) of the chunk to the top
      if (sh >= 64) { c.thi = c.tlo << (sh - 64); c.tlo = 0; }
      else if (sh > 0) { c.thi = (c.thi << sh) | (c.tlo >> (64 - sh)); c.tlo <<= sh; }
      c.gin = P.gt_in ? P.gt_in[c.w] : 0u;
    }
  }
  for (i64 g = 0; g < P.L; g += 32) {
    int steps[CPL], smax = 0;
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      Chain &c = S[q];
      i64 rem = c.u1 - c.u0 - g;
      steps[q] = rem <= 0 ? 0 : (rem < 32 ? (int)rem : 32);
      smax = steps[q] > smax ? steps[q] : smax;
      c.gin_next = (steps[q] && P.gt_in && c.u0 + g + 32 < c.u1) ? P.gt_in[c.w + 1] : 0u;   // prefetch the next gt word
      c.gout = 0;
    }
    if (smax == 0) break;
    for (int t = 0; t < smax; ++t) {
      RankReq<CNT, B> req[CPL];
      u32 sym[CPL];
      bool gti0[CPL];
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        if (t < steps[q]) {
          Chain &c = S[q];
          sym[q] = (u32)(c.thi >> 56);
          c.thi = (c.thi << 8) | (c.tlo >> 56);
          c.tlo <<= 8;
          gti0[q] = c.i > P.i0;
          c.gout |= (u32)gti0[q] << t;
          rank_issue<CNT, B>(R, T1, tot, BATCH ? c.i + P.rank_off : c.i, sym[q], req[q]);
        }
      }
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        if (t < steps[q]) {
          Chain &c = S[q];
          i64 ni = rank_finish<CNT, B>(R, sym[q], req[q]);
          if (BATCH) ni &= (i64)VAL_MASK;
          ni -= (gti0[q] && sym[q] == 0) ? 1 : 0;
          ni += (sym[q] == P.last && ((c.gin >> t) & 1u)) ? 1 : 0;
          c.i = ni;
          if (MODE >= 2) {
            // 4 consecutive ranks of a chain leave as ONE dwordx4 store (memory instructions per step
            // are the scarce resource of this kernel); lane-private LDS slots, no barrier needed
            u32 *ls = lstage + q * 4 * PSG_WG;
            ls[(t & 3) * PSG_WG + threadIdx.x] = (u32)(BATCH ? ni + P.gap_base : ni);
            if (MODE == 3) c.hiacc |= (u32)((u64)ni >> 32) << (8 * (t & 3));
            if ((t & 3) == 3) {
              uint4 q4 = make_uint4(ls[threadIdx.x], ls[PSG_WG + threadIdx.x], ls[2 * PSG_WG + threadIdx.x], ls[3 * PSG_WG + threadIdx.x]);
              ((uint4 *)P.log)[((g + t) >> 2) * P.K + c.k] = q4;
              if (MODE == 3) { P.log_hi[((g + t) >> 2) * P.K + c.k] = c.hiacc; c.hiacc = 0; }
            }
          } else if (MODE == 1) {
            excess_add_atomic(P.ex, &P.gap[BATCH ? ni + P.gap_base : ni], (u64)(BATCH ? ni + P.gap_base : ni), 1u);
          } else {
            atomicAdd(&P.gap[BATCH ? ni + P.gap_base : ni], 1u);
          }
          if (--c.tcnt == 0) {
            c.tcnt = 16;
            if (c.nbuf > 0) {
              --c.nbuf;
              c.tlo = (u64)c.b0.x | ((u64)c.b0.y << 32);
              c.thi = (u64)c.b0.z | ((u64)c.b0.w << 32);
              c.b0 = c.b1; c.b1 = c.b2;
            } else if ((uintptr_t)c.bp > c.last_addr) {    // the chain needs bytes below this block
              c.bp -= 4;
              PSG_TEXT_BLOCK(c, 3);
            }
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      if (steps[q]) {
        Chain &c = S[q];
        if (P.gt_out) {
          // scattered 4-byte stores are read-modify-writes of a whole sector in HBM (5 ms of a 57 ms pass):
          // stage the words in lane-private LDS slots and write 16 aligned bytes per 128 steps
          const int grp = (int)(g >> 5) & 3;
          u32 *gs = gstage + q * 4 * PSG_WG;
          const i64 wi = c.w - (P.ctx >> 5);
          if (!gt16) P.gt_out[wi] = c.gout;
          else {
            gs[grp * PSG_WG + threadIdx.x] = c.gout;
            const bool last_group = c.u0 + g + 32 >= c.u1;
            if (grp == 3) {
              *(uint4 *)(P.gt_out + (wi - 3)) = make_uint4(gs[threadIdx.x], gs[PSG_WG + threadIdx.x], gs[2 * PSG_WG + threadIdx.x], c.gout);
            } else if (last_group) {
              for (int j = 0; j <= grp; ++j) P.gt_out[wi - grp + j] = gs[j * PSG_WG + threadIdx.x];
            }
          }
        }
        c.gin = c.gin_next;
        ++c.w;
        if (MODE >= 2 && (steps[q] & 3)) {   // ragged end of the last chain: flush the partial group, rest = no entry
          int full = steps[q] & ~3;
          const u32 *ls = lstage + q * 4 * PSG_WG;
          uint4 q4 = make_uint4(ls[threadIdx.x], (steps[q] & 3) > 1 ? ls[PSG_WG + threadIdx.x] : 0xFFFFFFFFu,
                                (steps[q] & 3) > 2 ? ls[2 * PSG_WG + threadIdx.x] : 0xFFFFFFFFu, 0xFFFFFFFFu);
          ((uint4 *)P.log)[((g + full) >> 2) * P.K + c.k] = q4;
          if (MODE == 3) { P.log_hi[((g + full) >> 2) * P.K + c.k] = c.hiacc | (0xFFFFFFFFu << (8 * (steps[q] & 3))); c.hiacc = 0; }
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    Chain &c = S[q];
    if (c.k < 0) continue;
    if (MODE >= 2)   // a short (last) chain marks the rest of its log column as "no entry"
      for (i64 st = ((c.u1 - c.u0) + 3) & ~(i64)3; st < P.L; st += 4) {
        ((uint4 *)P.log)[(st >> 2) * P.K + c.k] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        if (MODE == 3) P.log_hi[(st >> 2) * P.K + c.k] = 0xFFFFFFFFu;
      }
    P.fin[c.k] = c.i;
  }
}

template <int CNT, int B, int MODE, int CPL>
// MODE 3 needs 82 VGPRs as the compiler allocates freely = 5 waves per SIMD (MODE 2: 76 = 6 waves); the kernel hides its
// memory latency with resident chains, and 5 instead of 6 workgroups per CU cost 12 % per suffix at configs[2]: ask
// for 6 (<= 80 VGPRs) there
__global__ __launch_bounds__(PSG_WG, (CNT == 0 && MODE == 3 && PSG_STREAM_MIN_WAVES < 6 ? 6 : (B <= 64 ? PSG_STREAM_MIN_WAVES : 1))) void stream_kernel(RankView<CNT, B> R, StreamParams P) {
  extern __shared__ u64 lds[];
  __shared__ u32 lstage[MODE >= 2 ? CPL * 4 * PSG_WG : 1];
  __shared__ u32 gstage[CPL * 4 * PSG_WG];   // 4 gt_out words (128 steps) of a chain leave as one 16-byte store
  load_tables(lds, P.g_T1, P.g_tot, P.nsb);
  stream_body<CNT, B, MODE, CPL, false>(R, P, lds, lds + P.nsb * 256, lstage, gstage, (i64)blockIdx.x * PSG_WG + threadIdx.x);
}

// Many small passes in ONE launch (the merging of host-sorted leaves, inmem_psascan.hpp:64-304: every pair of
// neighbouring sub-ranges of a level is one pass): workgroup w serves pass wg_pass[w] -- its parameters come from a
// table in device memory -- with the lanes wg_local[w] * 256 .. of that pass.  All passes share ONE rank structure,
// built over the concatenation of their blocks' BWTs.
template <int CNT, int B, int MODE>
__global__ __launch_bounds__(PSG_WG, (B <= 64 ? PSG_STREAM_MIN_WAVES : 1)) void stream_batch_kernel(RankView<CNT, B> R, const StreamParams *passes, const u32 *wg_pass, const u32 *wg_local) {
  extern __shared__ u64 lds[];
  __shared__ u32 lstage[MODE >= 2 ? 4 * PSG_WG : 1];
  __shared__ u32 gstage[4 * PSG_WG];
  const u32 w = __builtin_amdgcn_readfirstlane(wg_pass[blockIdx.x]);
  const StreamParams P = passes[w];
  load_tables(lds, P.g_T1, P.g_tot, P.nsb);
  stream_body<CNT, B, MODE, 1, true>(R, P, lds, lds + P.nsb * 256, lstage, gstage, (i64)__builtin_amdgcn_readfirstlane(wg_local[blockIdx.x]) * PSG_WG + threadIdx.x);
}

// Warm-up: find the start rank of chain k by running the recurrence on an interval
// [lo, hi] (monotone map) for W steps right of the chain start.  If the interval closes
// (lo == hi) the start rank is exact.  Chains within W steps of the tail end start from
// the exact rank_at_tail_end.
struct WarmParams {
  const u8 *tail;
  i64 T;
  i64 ctx;
  const u32 *gt_in;
  i64 i0;
  u32 last;
  i64 L, W;
  i64 m;
  i64 rank_at_end;
  i64 nitems;
  const i64 *list;   // chain ids to process (null = all chains 0..nitems-1)
  i64 *lo, *hi;      // [K]
  const u64 *g_T1;
  const u64 *g_tot;
  int nsb;
};

template <int CNT, int B>
__global__ __launch_bounds__(PSG_WG) void warmup_kernel(RankView<CNT, B> R, WarmParams P) {
  extern __shared__ u64 lds[];
  load_tables(lds, P.g_T1, P.g_tot, P.nsb);
  const u64 *T1 = lds, *tot = lds + P.nsb * 256;
  i64 gid = (i64)blockIdx.x * PSG_WG + threadIdx.x;
  if (gid >= P.nitems) return;
  i64 k = P.list ? P.list[gid] : gid;
  i64 uend = P.ctx + k * P.L;        // the chain starts at step uend; warm-up covers [ubeg, uend)
  i64 ubeg = uend - P.W;
  i64 lo, hi;
  if (ubeg <= 0 && P.rank_at_end >= 0) { ubeg = 0; lo = hi = P.rank_at_end; }   // exact start
  else { if (ubeg < 0) ubeg = 0; lo = 0; hi = P.m; }
  for (i64 u = ubeg; u < uend; ++u) {
    u32 c = P.tail[P.T - 1 - u];
    u32 g = P.gt_in ? (P.gt_in[u >> 5] >> (u & 31)) & 1u : 0u;
    i64 add = (c == P.last && g) ? 1 : 0;
    i64 nlo = lf_core<CNT, B>(R, T1, tot, lo, c) - ((lo > P.i0 && c == 0) ? 1 : 0) + add;
    i64 nhi = lo == hi ? nlo : lf_core<CNT, B>(R, T1, tot, hi, c) - ((hi > P.i0 && c == 0) ? 1 : 0) + add;
    lo = nlo; hi = nhi;
  }
  P.lo[k] = lo;
  P.hi[k] = hi;
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
#define DISPATCH_LAYOUT(r, F, ...)                                                                   \
  do {                                                                                               \
    if ((r)->cnt == 0) { F<0, 0>(__VA_ARGS__); }                                                     \
    else if ((r)->cnt == 4 && (r)->B == 48) { F<4, 48>(__VA_ARGS__); }                               \
    else if ((r)->cnt == 16 && (r)->B == 64) { F<16, 64>(__VA_ARGS__); }                             \
    else if ((r)->cnt == 256 && (r)->B == 32) { F<256, 32>(__VA_ARGS__); }                           \
    else if ((r)->cnt == 256 && (r)->B == 64) { F<256, 64>(__VA_ARGS__); }                           \
    else if ((r)->cnt == 256 && (r)->B == 128) { F<256, 128>(__VA_ARGS__); }                         \
    else if ((r)->cnt == 256 && (r)->B == 256) { F<256, 256>(__VA_ARGS__); }                         \
    else { set_error("unsupported rank layout"); return PSG_EINVAL; }                                \
  } while (0)

template <int CNT, int B>
static void launch_build(const u8 *d_bwt, i64 m, const u8 *d_code, u32 *seg_cnt, u64 *group_sum, psg_rank *r, i64 ngroups) {
  hipLaunchKernelGGL((seg_hist_kernel<CNT, B>), dim3((unsigned)r->nseg), dim3(PSG_WG), 0, stream(), d_bwt, m, d_code, seg_cnt);
  hipLaunchKernelGGL(group_prefix_kernel, dim3((unsigned)cdiv(ngroups * CNT, PSG_WG)), dim3(PSG_WG), 0, stream(), seg_cnt,
                     r->nseg, CNT, group_sum, ngroups);
  hipLaunchKernelGGL(group_scan_kernel, dim3(1), dim3(PSG_WG), 0, stream(), group_sum, ngroups, CNT);
  hipLaunchKernelGGL((rank_fill_kernel<CNT, B>), dim3((unsigned)r->nseg), dim3(PSG_WG), 0, stream(), d_bwt, m, d_code, seg_cnt,
                     group_sum, r->d_blocks, r->nblk, r->sb_shift);
}

namespace psg { const u32 *rank_build_seg_mask = nullptr; }   // see dev_common.hpp
static_assert(SM_SEG == psg::RANK_BUILD_SEG, "the segment mask is per build segment");
// Symbol-major layout (rank_sm.hpp).  *fell_back = true: not applicable / did not fit / overflow
// pool exhausted -- the caller builds a block layout instead.
static int sm_build(psg_rank *r, const u8 *d_bwt, i64 m, const u64 *h, double budget_bytes, bool allow_list8, bool *fell_back, bool *list8_failed) {
  *fell_back = true;
  // counts inside the entries are relative to superblocks of 2^31 positions (LIST8 keeps bit 31 of the count as
  // its "dense bucket" flag); PSG_SM_SB_SHIFT makes them small so that tests run several superblocks
  // superblocks: multiples of the build segment (3072 = 12 LIST buckets = 32 BITMAP buckets), just under 2^31 positions
  i64 sb_segs = SM_SB_SEGS_DEFAULT;
  if (const char *e = getenv("PSG_SM_SB_SHIFT")) { int v = atoi(e); if (v >= 12 && v < 31) sb_segs = (i64)1 << (v - 12); }
  const i64 sb_size = sb_segs * SM_SEG;
  if ((m - 1) / sb_size + 1 > 64) return 0;
  const int nsb = (int)((m - 1) / sb_size + 1);
  if (nsb > 64) return 0;   // LDS table of the pass: 2 KiB per superblock
  // per-symbol mode: BITMAP (bucket 64) when a 256-bucket would hold > 3 occurrences on average; the other
  // symbols get 8-byte LIST8 entries (4 positions inline) if every one of them averages <= 1.5 per bucket
  // -- a byte-uniform text: 8 instead of 16 bytes of structure per symbol -- else 16-byte LIST entries
  u64 off16 = 0;
  u64 t2[256];
  int nbitmap = 0;
  bool list8 = allow_list8 && !(getenv("PSG_SM_LIST8") && !strcmp(getenv("PSG_SM_LIST8"), "0"));
  *list8_failed = false;
  double dense8 = 0;   // expected number of LIST8 buckets with more than 4 occurrences (Poisson)
  for (int c = 0; c < 256; ++c) {
    if (!h[c]) continue;
    const double lam = (double)h[c] * 256.0 / (double)m;
    if (lam > 3.0) { ++nbitmap; continue; }
    if (lam > 1.5) list8 = false;
    double p = std::exp(-lam), tail = 1.0 - p;
    for (int k = 1; k <= SM_CAP8; ++k) { p *= lam / k; tail -= p; }
    dense8 += std::max(0.0, tail) * (double)cdiv(m, 256);
  }
  if (nbitmap > SM_MAX_BITMAP) return 0;   // too many frequent symbols for the fill kernel's LDS budget
  for (int c = 0; c < 256; ++c) {
    if (!h[c]) { t2[c] = 0; continue; }
    bool bitmap = (double)h[c] * 256.0 > 3.0 * (double)m;
    u64 nbk = bitmap ? (u64)cdiv(m, SM_BW) : (u64)cdiv(m, 256);
    if (!bitmap && list8) nbk = (nbk + 1) / 2;   // 8-byte entries, counted in 16-byte units
    t2[c] = off16 | ((u64)(bitmap ? SM_BITMAP : (list8 ? SM_LIST8 : SM_LIST)) << SM_MODE_SHIFT);
    off16 += (nbk + 7) / 8 * 8;   // regions start on 128-byte lines (the fill kernel writes whole lines)
  }
  const i64 entries_bytes = (i64)off16 * 16;
  // 32-byte bitmaps for dense buckets: LIST keeps 11 positions inline (dense = runs of one symbol); LIST8 keeps 4
  const u32 pool_cap = (u32)std::min<double>(4.0e9, std::max<double>(1024.0, (double)m / 512.0 + (list8 ? 2.0 * dense8 + 65536.0 : 0.0)));
  if ((double)entries_bytes + 32.0 * pool_cap > budget_bytes) return 0;
  const i64 nseg = cdiv(m, SM_SEG), ngroups = cdiv(nseg, GROUP_SEGS);
  DevBuf code_d, seg_cnt, group_sum, t2_d, misc, sb_d;
  int rc;
  u8 ident[256];
  for (int c = 0; c < 256; ++c) ident[c] = (u8)c;
  u8 *entries = nullptr, *pool = nullptr;
  if (psg::pool_alloc((void **)&entries, (size_t)entries_bytes + 64) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (psg::pool_alloc((void **)&pool, (size_t)pool_cap * 32) != hipSuccess) { (void)hipGetLastError(); psg::pool_free(entries); return 0; }
  auto fail = [&](int code) { psg::pool_free(entries); psg::pool_free(pool); return code; };
  if ((rc = code_d.alloc(256)) || (rc = seg_cnt.alloc(nseg * 256 * 4)) || (rc = group_sum.alloc(ngroups * 256 * 8)) || (rc = t2_d.alloc(256 * 8)) || (rc = misc.alloc(8)) ||
      (rc = sb_d.alloc((i64)nsb * 256 * 8)))
    return fail(rc);
  if ((rc = psg::copy_h2d(code_d.p, ident, 256)) || (rc = psg::copy_h2d(t2_d.p, t2, sizeof t2))) return fail(rc);
  if (hipMemsetAsync(misc.p, 0, 8, stream()) != hipSuccess) { set_error("sm_build: memset failed"); return fail(PSG_EDEVICE); }
  hipLaunchKernelGGL(sm_seg_hist_kernel, dim3((unsigned)nseg), dim3(PSG_WG), 0, stream(), d_bwt, m, seg_cnt.as<u32>());
  hipLaunchKernelGGL(group_prefix_kernel, dim3((unsigned)cdiv(ngroups * 256, PSG_WG)), dim3(PSG_WG), 0, stream(), seg_cnt.as<u32>(), nseg, 256, group_sum.as<u64>(), ngroups);
  hipLaunchKernelGGL(group_scan_kernel, dim3(1), dim3(PSG_WG), 0, stream(), group_sum.as<u64>(), ngroups, 256);
  hipLaunchKernelGGL(sm_sb_base_kernel, dim3((unsigned)nsb), dim3(256), 0, stream(), seg_cnt.as<u32>(), group_sum.as<u64>(), sb_segs, sb_d.as<u64>());
  if (list8) hipLaunchKernelGGL(sm_fill_kernel<true>, dim3((unsigned)nseg), dim3(256), 0, stream(), d_bwt, m, t2_d.as<u64>(), seg_cnt.as<u32>(), group_sum.as<u64>(),
                                (uint4 *)entries, (u32 *)pool, misc.as<u32>(), pool_cap, (int *)(misc.as<u32>() + 1), sb_segs, psg::rank_build_seg_mask);
  else hipLaunchKernelGGL(sm_fill_kernel<false>, dim3((unsigned)nseg), dim3(256), 0, stream(), d_bwt, m, t2_d.as<u64>(), seg_cnt.as<u32>(), group_sum.as<u64>(),
                          (uint4 *)entries, (u32 *)pool, misc.as<u32>(), pool_cap, (int *)(misc.as<u32>() + 1), sb_segs, psg::rank_build_seg_mask);
  u32 st[2] = {0, 0};
  hipError_t e4 = hipGetLastError();
  if (e4 != hipSuccess) { set_error(std::string("sm_build: ") + hipGetErrorString(e4)); return fail(PSG_EDEVICE); }
  if ((rc = psg::copy_d2h(st, misc.p, 8))) return fail(rc);
  if (st[1] || st[0] > pool_cap) {   // too many dense buckets (runs of a symbol): the caller retries with roomier entries, then blocks
    psg::pool_free(entries); psg::pool_free(pool);
    *list8_failed = list8;
    return 0;
  }
  r->h_sb.assign((size_t)nsb * 256, 0);
  if ((rc = psg::copy_d2h(r->h_sb.data(), sb_d.p, (size_t)nsb * 256 * 8))) return fail(rc);
  r->cnt = 0; r->B = 0; r->stride = 0; r->nblk = 0; r->nseg = nseg; r->nsb = nsb; r->sb_shift = 0; r->sb_size = sb_size;
  r->d_blocks = entries; r->d_aux = pool; r->blocks_bytes = entries_bytes + (i64)pool_cap * 32;
  for (int c = 0; c < 256; ++c) { r->t2[c] = t2[c]; r->code[c] = (u8)c; }
  *fell_back = false;
  return 0;
}

extern "C" int psg_rank_build(const uint8_t *d_bwt, int64_t m, int data_bytes, psg_rank_t **out) {
  PSG_REQUIRE(out && m >= 1 && d_bwt, "psg_rank_build: m >= 1 and non-null pointers required");
  PSG_REQUIRE(stream() != nullptr, "psg_init() not called");
  EventTimer tm;
  tm.start();
  // phase 0: global histogram -> alphabet -> layout
  DevBuf hist;
  if (int rc = hist.alloc(256 * 8)) return rc;
  PSG_HIP(hipMemsetAsync(hist.p, 0, 256 * 8, stream()));
  {
    unsigned grid = (unsigned)std::min<i64>(cdiv(m, (i64)PSG_WG * 16), 2048);
    hipLaunchKernelGGL(hist256_kernel, dim3(grid), dim3(PSG_WG), 0, stream(), d_bwt, m, hist.as<unsigned long long>());
    PSG_HIP(hipGetLastError());
  }
  u64 h[256];
  if (int rc_ = psg::copy_d2h(h, hist.p, (size_t)(sizeof h))) return rc_;
  PSG_HIP(psg::sync_stream());
  psg_rank *r = new psg_rank();
  r->m = m;
  int sigma = 0;
  for (int c = 0; c < 256; ++c) { r->count[c] = (i64)h[c]; sigma += h[c] != 0; }
  {
    // symbol-major layout (one 16-byte load per query) by default for every alphabet -- DNA: 64-position
    // bitmaps, 1.1 B/symbol, 48 instead of 69 ms per 2^31 steps against the 4-counter blocks, whose counter +
    // two data loads per step are bound by the request rate.  PSG_RANK_LAYOUT=block disables it.
    const char *env = getenv("PSG_RANK_LAYOUT");
    bool want_sm = data_bytes == 1 || (data_bytes == 0 && !(env && !strcmp(env, "block")));
    if (want_sm) {
      bool fell_back = true, l8_failed = false;
      int rc = sm_build(r, d_bwt, m, h, 0.35 * (double)mem_available(), true, &fell_back, &l8_failed);
      if (!rc && fell_back && l8_failed) {   // a BWT with runs overflows the 4 inline positions: 16-byte entries (one failed fill ~ 10 ms)
        rc = sm_build(r, d_bwt, m, h, 0.35 * (double)mem_available(), false, &fell_back, &l8_failed);
      }
      if (rc) { delete r; return rc; }
      if (!fell_back) {
        tm.stop();
        PSG_HIP(psg::sync_stream());
        note_kernel_ms(tm.ms());
        *out = r;
        return 0;
      }
      if (data_bytes == 1) data_bytes = 0;   // forced but not applicable: block layout
    }
  }
  if (data_bytes == 0) {
    if (sigma <= 4) { r->cnt = 4; r->B = 48; }
    else if (sigma <= 16) { r->cnt = 16; r->B = 64; }
    else {
      // general alphabet: the smaller the block, the fewer load instructions per query
      // (B=32: counter dword + ONE dwordx4; B=64: + two; ...) at 33 / 17 / 9 / 5 bytes per symbol.
      // Take the smallest block whose structure stays within a quarter of the free HBM.
      double budget = 0.35 * (double)mem_available();
      r->cnt = 256;
      r->B = 33.0 * m <= budget ? 32 : 17.0 * m <= budget ? 64 : 9.0 * m <= budget ? 128 : 256;
    }
  } else if (data_bytes == 48 && sigma <= 4) { r->cnt = 4; r->B = 48; }
  else if (data_bytes == 64 && sigma <= 16) { r->cnt = 16; r->B = 64; }
  else if (data_bytes == 32 || data_bytes == 64 || data_bytes == 128 || data_bytes == 256) { r->cnt = 256; r->B = data_bytes; }
  else if (data_bytes == -64) { r->cnt = 256; r->B = 64; }   // force the general layout (tests)
  else { delete r; set_error("psg_rank_build: data_bytes_per_block must be 0, 32, 48, 64, 128, 256"); return PSG_EINVAL; }
  if (r->cnt == 256) for (int c = 0; c < 256; ++c) r->code[c] = (u8)c;
  else { int k = 0; for (int c = 0; c < 256; ++c) r->code[c] = h[c] ? (u8)k++ : 0xFF; }
  r->stride = 4 * r->cnt + r->B;
  r->nblk = cdiv(m, r->B);
  r->nseg = cdiv(r->nblk, SEG_BLOCKS);
  r->sb_shift = SB_SHIFT_DEFAULT;   // superblocks start on scan-group boundaries: >= log2(SEG_BLOCKS * GROUP_SEGS) = 14
  if (const char *e = getenv("PSG_BLOCK_SB_SHIFT")) { int v = atoi(e); if (v >= 14 && v <= 30) r->sb_shift = v; }
  r->nsb = (int)(((r->nblk - 1) >> r->sb_shift) + 1);
  if (r->nsb > 64) { delete r; set_error("psg_rank_build: too many superblocks for the LDS table"); return PSG_EINVAL; }
  i64 ngroups = cdiv(r->nseg, GROUP_SEGS);
  r->blocks_bytes = r->nblk * (i64)r->stride;
  int rc = 0;
  DevBuf code_d, seg_cnt, group_sum;
  hipError_t e = psg::pool_alloc((void **)&r->d_blocks, (size_t)r->blocks_bytes + 64);   // +64: the upper-part read of the last block may run 16 B over
  if (e != hipSuccess) { set_error(std::string("rank blocks hipMalloc ") + std::to_string(r->blocks_bytes) + ": " + hipGetErrorString(e)); delete r; return PSG_ENOMEM; }
  if ((rc = code_d.alloc(256)) || (rc = seg_cnt.alloc(r->nseg * r->cnt * 4)) || (rc = group_sum.alloc(ngroups * r->cnt * 8))) { psg_rank_free(r); return rc; }
  if (int rc_ = psg::copy_h2d(code_d.p, r->code, (size_t)(256))) { psg_rank_free(r); return rc_; }
  if (r->cnt == 4) launch_build<4, 48>(d_bwt, m, code_d.as<u8>(), seg_cnt.as<u32>(), group_sum.as<u64>(), r, ngroups);
  else if (r->cnt == 16) launch_build<16, 64>(d_bwt, m, code_d.as<u8>(), seg_cnt.as<u32>(), group_sum.as<u64>(), r, ngroups);
  else if (r->B == 32) launch_build<256, 32>(d_bwt, m, code_d.as<u8>(), seg_cnt.as<u32>(), group_sum.as<u64>(), r, ngroups);
  else if (r->B == 64) launch_build<256, 64>(d_bwt, m, code_d.as<u8>(), seg_cnt.as<u32>(), group_sum.as<u64>(), r, ngroups);
  else if (r->B == 128) launch_build<256, 128>(d_bwt, m, code_d.as<u8>(), seg_cnt.as<u32>(), group_sum.as<u64>(), r, ngroups);
  else launch_build<256, 256>(d_bwt, m, code_d.as<u8>(), seg_cnt.as<u32>(), group_sum.as<u64>(), r, ngroups);
  PSG_HIP(hipGetLastError());
  // superblock bases = group bases at the superblock starts
  r->h_sb.assign((size_t)r->nsb * r->cnt, 0);
  for (int s = 0; s < r->nsb; ++s) {
    i64 g = (((i64)s << r->sb_shift) / SEG_BLOCKS) / GROUP_SEGS;
    if (int rc_ = psg::copy_d2h(&r->h_sb[(size_t)s * r->cnt], group_sum.as<u64>() + g * r->cnt, (size_t)((size_t)r->cnt * 8))) { psg_rank_free(r); return rc_; }
  }
  tm.stop();
  PSG_HIP(psg::sync_stream());
  note_kernel_ms(tm.ms());
  *out = r;
  return 0;
}

extern "C" void psg_rank_free(psg_rank_t *r) {
  if (!r) return;
  if (r->d_blocks) psg::pool_free(r->d_blocks);
  if (r->d_aux) psg::pool_free(r->d_aux);
  delete r;
}
extern "C" int psg_rank_counts(const psg_rank_t *r, int64_t counts[256]) {
  PSG_REQUIRE(r && counts, "psg_rank_counts");
  for (int c = 0; c < 256; ++c) counts[c] = r->count[c];
  return 0;
}
extern "C" int64_t psg_rank_device_bytes(const psg_rank_t *r) { return r ? r->blocks_bytes : 0; }

// T1[sb][c] = (Cadd[c] + sb_base[sb][code[c]]) | code << 56 ; tot[c] = count[c]
static int make_tables(const psg_rank *r, const i64 *Cadd, DevBuf &T1, DevBuf &tot) {
  // built in a pinned buffer and copied asynchronously: the kernels that read the tables are ordered behind the copy on
  // the stream, and the next call (which reuses the buffer) only comes after this pass has been waited for
  const size_t n1 = (size_t)r->nsb * 256;
  u64 *h = (u64 *)pinned_buf(13, (n1 + 512) * 8);
  if (!h) { set_error("make_tables: pinned host allocation failed"); return PSG_ENOMEM; }
  u64 *t = h + n1;
  for (int s = 0; s < r->nsb; ++s)
    for (int c = 0; c < 256; ++c) {
      u8 cd = r->code[c];
      u64 base = r->cnt == 0 ? r->h_sb[(size_t)s * 256 + c] : (cd == 0xFF ? 0 : r->h_sb[(size_t)s * r->cnt + cd]);
      h[(size_t)s * 256 + c] = ((u64)(Cadd ? Cadd[c] : 0) + base) | ((u64)cd << CODE_SHIFT);
    }
  for (int c = 0; c < 256; ++c) { t[c] = (u64)r->count[c]; t[256 + c] = r->t2[c]; }
  if (int rc = T1.alloc((i64)n1 * 8)) return rc;
  if (int rc = tot.alloc(512 * 8)) return rc;
  PSG_HIP(hipMemcpyAsync(T1.p, h, n1 * 8, hipMemcpyHostToDevice, stream()));
  PSG_HIP(hipMemcpyAsync(tot.p, t, 512 * 8, hipMemcpyHostToDevice, stream()));
  return 0;
}

template <int CNT, int B>
static void launch_query(const psg_rank *r, const u64 *T1, const u64 *tot, const i64 *qi, const u8 *qc, i64 nq, i64 *out) {
  RankView<CNT, B> R{r->d_blocks, r->m, r->d_aux, r->sb_shift, r->sb_size};
  size_t lds = ((size_t)r->nsb * 256 + 512) * 8;
  hipLaunchKernelGGL((rank_query_kernel<CNT, B>), dim3((unsigned)cdiv(nq, PSG_WG)), dim3(PSG_WG), lds, stream(), R, T1, tot,
                     r->nsb, qi, qc, nq, out);
}

extern "C" int psg_rank_query(const psg_rank_t *r, const int64_t *d_i, const uint8_t *d_c, int64_t nq, int64_t *d_out) {
  PSG_REQUIRE(r && nq >= 0, "psg_rank_query");
  if (nq == 0) return 0;
  DevBuf T1, tot;
  if (int rc = make_tables(r, nullptr, T1, tot)) return rc;
  DISPATCH_LAYOUT(r, launch_query, r, T1.as<u64>(), tot.as<u64>(), d_i, d_c, nq, d_out);
  PSG_HIP(hipGetLastError());
  PSG_HIP(psg::sync_stream());
  return 0;
}

template <int CNT, int B> static void launch_warm(const psg_rank *r, WarmParams P) {
  RankView<CNT, B> R{r->d_blocks, r->m, r->d_aux, r->sb_shift, r->sb_size};
  size_t lds = ((size_t)r->nsb * 256 + 512) * 8;
  hipLaunchKernelGGL((warmup_kernel<CNT, B>), dim3((unsigned)cdiv(P.nitems, PSG_WG)), dim3(PSG_WG), lds, stream(), R, P);
}
// chains per lane: 2 for the layouts with at most two loads per query (symbol-major, sigma <= 16),
// PSG_CPL=1|2 overrides
static int chains_per_lane(const psg_rank *r) {
  if (const char *e = getenv("PSG_CPL")) return atoi(e) == 2 ? 2 : 1;
  (void)r;
  return 1;   // measured (MI355X, 4 GiB bench): 2 chains/lane at 4 waves/SIMD == 1 chain/lane at 8 waves/SIMD
}
template <int CNT, int B> static void launch_stream(const psg_rank *r, StreamParams P, int mode, int cpl) {
  RankView<CNT, B> R{r->d_blocks, r->m, r->d_aux, r->sb_shift, r->sb_size};
  size_t lds = ((size_t)r->nsb * 256 + 512) * 8;
  dim3 grid((unsigned)cdiv(cdiv(P.nchains, cpl), PSG_WG));
#define PSG_LAUNCH(MODE_)                                                                                          \
  do {                                                                                                             \
    if (cpl == 2) hipLaunchKernelGGL((stream_kernel<CNT, B, MODE_, 2>), grid, dim3(PSG_WG), lds, stream(), R, P);  \
    else hipLaunchKernelGGL((stream_kernel<CNT, B, MODE_, 1>), grid, dim3(PSG_WG), lds, stream(), R, P);           \
  } while (0)
  if (mode == 3) PSG_LAUNCH(3);
  else if (mode == 2) PSG_LAUNCH(2);
  else if (mode == 1) PSG_LAUNCH(1);
  else PSG_LAUNCH(0);
#undef PSG_LAUNCH
}

// ---------------------------------------------------------------------------------------
// batched passes (leaf_tree.hip): pass table + per-pass LDS tables, built on the device
// ---------------------------------------------------------------------------------------
struct psg::BatchTables {
  DevBuf T1g, tot;            // the structure's own tables (C = 0)
  DevBuf T1p;                 // [npass][nsb][256]
  DevBuf passes;              // [npass] StreamParams
  i64 npass = 0;
};
psg::BatchTables *psg::stream_batch_tables_create(const psg_rank_t *r, i64 npass) {
  BatchTables *t = new BatchTables();
  t->npass = npass;
  if (make_tables(r, nullptr, t->T1g, t->tot) || t->T1p.alloc(npass * (i64)r->nsb * 256 * 8) || t->passes.alloc(npass * (i64)sizeof(StreamParams))) { delete t; return nullptr; }
  return t;
}
void psg::stream_batch_tables_free(BatchTables *t) { delete t; }

// workgroup p, thread c: counts of symbol c in front of / inside the block of pass p, the pass's C array
// (compute_gap.hpp:77-85), T1p[p][sb][c] = C[c] - rank(lbeg, c) + base of superblock sb (mod 2^56); thread 0 writes
// the pass parameters.
template <int CNT, int B>
__global__ __launch_bounds__(256) void batch_setup_kernel(RankView<CNT, B> R, const u64 *g_T1, const u64 *g_tot, int nsb, const psg::BatchGeom *geom,
                                                           const i64 *i0_nodes, const u8 *text, const u32 *gt_cur, u32 *gt_new, i64 L, i64 *init, i64 *fin,
                                                           u32 *log, i64 Ktotal, u32 *gap, psg::GapExcess ex, u64 *T1p, StreamParams *passes) {
  extern __shared__ u64 lds[];
  __shared__ i64 scratch[8];
  load_tables(lds, g_T1, g_tot, nsb);
  const u64 *T1 = lds, *tot = lds + nsb * 256;
  const psg::BatchGeom G = geom[blockIdx.x];
  const u32 c = threadIdx.x;
  const u32 last = text[G.lbeg + G.m - 1];
  const i64 before = lf_core<CNT, B>(R, T1, tot, G.lbeg, c), upto = lf_core<CNT, B>(R, T1, tot, G.lbeg + G.m, c);
  i64 total;
  const i64 v = (upto - before) + (c == last ? 1 : 0) - (c == 0 ? 1 : 0);
  const i64 C = block_excl_scan<i64>(v, scratch, total);
  u64 *out = T1p + (i64)blockIdx.x * nsb * 256;
  for (int sb = 0; sb < nsb; ++sb) {
    const u64 g = T1[sb * 256 + c];
    out[sb * 256 + c] = (((g & VAL_MASK) + (u64)(C - before)) & VAL_MASK) | (g & ~VAL_MASK);
  }
  if (c == 0) {
    StreamParams P{};
    P.tail = text + G.lbeg + G.m; P.T = G.T; P.ctx = 0;
    P.gt_in = gt_cur + G.gt_in_word; P.gt_out = gt_new + G.gt_out_word; P.gap = gap;
    P.i0 = i0_nodes[G.node]; P.last = last; P.L = L; P.nchains = (G.T + L - 1) / L; P.list = nullptr;
    P.init = init + G.kbase; P.fin = fin + G.kbase; P.g_T1 = out; P.g_tot = g_tot; P.nsb = nsb; P.ovf_flag = nullptr;
    P.log = log ? log + 4 * G.kbase : nullptr; P.K = Ktotal; P.log_hi = nullptr; P.ex = ex;
    P.rank_off = G.lbeg; P.gap_base = G.gap_base;
    passes[blockIdx.x] = P;
  }
}
template <int CNT, int B>
static void launch_batch_setup(const psg_rank *r, psg::BatchTables *t, const psg::BatchGeom *geom, i64 npass, const i64 *i0_nodes, const u8 *text, const u32 *gt_cur,
                               u32 *gt_new, i64 L, i64 *init, i64 *fin, u32 *log, i64 Ktotal, u32 *gap, psg::GapExcess ex) {
  RankView<CNT, B> R{r->d_blocks, r->m, r->d_aux, r->sb_shift, r->sb_size};
  const size_t lds = ((size_t)r->nsb * 256 + 512) * 8;
  hipLaunchKernelGGL((batch_setup_kernel<CNT, B>), dim3((unsigned)npass), dim3(256), lds, stream(), R, t->T1g.as<u64>(), t->tot.as<u64>(), r->nsb, geom, i0_nodes, text,
                     gt_cur, gt_new, L, init, fin, log, Ktotal, gap, ex, t->T1p.as<u64>(), t->passes.as<StreamParams>());
}
int psg::stream_batch_setup(const psg_rank_t *r, BatchTables *t, const BatchGeom *d_geom, i64 npass, const i64 *d_i0_nodes, const u8 *d_text_range, const u32 *d_gt_cur,
                            u32 *d_gt_new, i64 L, i64 *d_init, i64 *d_fin, u32 *d_log, i64 Ktotal, u32 *d_gap, GapExcess ex) {
  PSG_REQUIRE(r && t && t->npass == npass && npass >= 1 && L >= 4 && (L & 3) == 0, "stream_batch_setup");
  DISPATCH_LAYOUT(r, launch_batch_setup, r, t, d_geom, npass, d_i0_nodes, d_text_range, d_gt_cur, d_gt_new, L, d_init, d_fin, d_log, Ktotal, d_gap, ex);
  PSG_HIP(hipGetLastError());
  return 0;
}
template <int CNT, int B>
static void launch_stream_batch(const psg_rank *r, const psg::BatchTables *t, const u32 *wg_pass, const u32 *wg_local, i64 nwg, int mode) {
  RankView<CNT, B> R{r->d_blocks, r->m, r->d_aux, r->sb_shift, r->sb_size};
  const size_t lds = ((size_t)r->nsb * 256 + 512) * 8;
  const dim3 grid((unsigned)nwg);
  const StreamParams *passes = t->passes.as<StreamParams>();
  if (mode == 2) hipLaunchKernelGGL((stream_batch_kernel<CNT, B, 2>), grid, dim3(PSG_WG), lds, stream(), R, passes, wg_pass, wg_local);
  else if (mode == 1) hipLaunchKernelGGL((stream_batch_kernel<CNT, B, 1>), grid, dim3(PSG_WG), lds, stream(), R, passes, wg_pass, wg_local);
  else hipLaunchKernelGGL((stream_batch_kernel<CNT, B, 0>), grid, dim3(PSG_WG), lds, stream(), R, passes, wg_pass, wg_local);
}
int psg::stream_batch_launch(const psg_rank_t *r, const BatchTables *t, const u32 *d_wg_pass, const u32 *d_wg_local, i64 nwg, int mode) {
  PSG_REQUIRE(r && t && nwg >= 1 && mode >= 0 && mode <= 2, "stream_batch_launch");
  DISPATCH_LAYOUT(r, launch_stream_batch, r, t, d_wg_pass, d_wg_local, nwg, mode);
  PSG_HIP(hipGetLastError());
  return 0;
}
template <int CNT, int B> static void batch_occupancy(const psg_rank *r, int mode, int *blocks) {
  const size_t lds = ((size_t)r->nsb * 256 + 512) * 8;
  *blocks = 0;
  if (mode == 2) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, stream_batch_kernel<CNT, B, 2>, PSG_WG, 0);
  else if (mode == 1) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, stream_batch_kernel<CNT, B, 1>, PSG_WG, 0);
  else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, stream_batch_kernel<CNT, B, 0>, PSG_WG, 0);
  (void)hipGetLastError();
  const int by_lds = (int)(((size_t)160 << 10) / (lds + 8192 + 256));
  if (*blocks > by_lds) *blocks = by_lds;
  if (*blocks < 1) *blocks = 4;
  if (*blocks > 8) *blocks = 8;
}
int psg::stream_batch_blocks_per_cu(const psg_rank_t *r, int mode) {
  int blocks = 4;
  if (r->cnt == 0) batch_occupancy<0, 0>(r, mode, &blocks);
  return blocks;
}

// resident workgroups per CU of the stream kernel that will be launched (occupancy API)
template <int CNT, int B> static void query_occupancy(const psg_rank *r, int mode, int cpl, int *blocks) {
  // The occupancy API budgets 64 KiB of LDS per CU; gfx950 has 160 KiB, and a block with three superblocks (a 4 GiB
  // block: 10 KiB of tables) was planned at 5 workgroups per CU where 6 run (measured: 56.7 -> 55.8 ms per 2^31
  // suffixes, no second wave).  So: registers and static LDS from the API, the dynamic LDS against the real size.
  const size_t lds = ((size_t)r->nsb * 256 + 512) * 8;
  *blocks = 0;
  size_t static_lds = 8192;
#define PSG_OCC(MODE_)                                                                                                     \
  do {                                                                                                                     \
    hipFuncAttributes fa{};                                                                                                \
    if (cpl == 2) {                                                                                                        \
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, stream_kernel<CNT, B, MODE_, 2>, PSG_WG, 0);              \
      if (hipFuncGetAttributes(&fa, (const void *)stream_kernel<CNT, B, MODE_, 2>) == hipSuccess) static_lds = fa.sharedSizeBytes; \
    } else {                                                                                                               \
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, stream_kernel<CNT, B, MODE_, 1>, PSG_WG, 0);              \
      if (hipFuncGetAttributes(&fa, (const void *)stream_kernel<CNT, B, MODE_, 1>) == hipSuccess) static_lds = fa.sharedSizeBytes; \
    }                                                                                                                      \
  } while (0)
  if (mode == 3) PSG_OCC(3);
  else if (mode == 2) PSG_OCC(2);
  else if (mode == 1) PSG_OCC(1);
  else PSG_OCC(0);
#undef PSG_OCC
  (void)hipGetLastError();
  const int by_lds = (int)(((size_t)160 << 10) / (lds + static_lds + 256));     // + allocation granularity
  if (*blocks > by_lds) *blocks = by_lds;
}

// everything one pass needs (the public entry points differ only in how they fill this in)
struct PassArgs {
  const psg_rank_t *r;
  i64 i0;
  int last_sym;
  const u8 *d_tail;
  i64 T, ctx;
  const u32 *d_gt_in;
  i64 rank_at_end;
  u32 *d_gap;
  u32 *d_gt_out;
  i64 max_chains;
  bool fresh;                     // PSG_GAP_UNINITIALIZED
  bool fail_if_unresolved;        // PSG_FAIL_IF_UNRESOLVED
  bool search_all;                // PSG_SEARCH_ALL_STARTS
  const psg_search_ctx *search;   // K8 for chain starts the warm-up leaves open
  i64 tail_begin_abs;
  u32 **log_out;
  i64 *nlog_out;
  struct DeferredHist *defer;     // chunked pass: the histogram of a chunk's rank log runs behind the next chunk's kernel
};
// The partition + histogram of a chunk's rank log on the side stream, while the main stream runs the next chunk's
// kernel (which only appends to its own log): the job, the log it reads, and where its time is reported.
struct DeferredHist {
  HistJob job;
  DevBuf log_lo, log_hi;
  bool active = false;
  int wait(double *ms) {
    if (!active) { if (ms) *ms = 0; return 0; }
    active = false;
    int rc;
    { StreamScope sc(side_stream()); rc = gap_hist_wait(job, ms); }
    log_lo.alloc(16); log_hi.alloc(16);
    return rc;
  }
};
static int stream_impl(const PassArgs &A, int64_t *h_final_rank, psg_stream_stats *stats);
static int stream_chunk(const PassArgs &A, int64_t *h_final_rank, psg_stream_stats *stats);

extern "C" int psg_stream_gap(const psg_rank_t *r, int64_t i0, int last_sym, const uint8_t *d_tail, int64_t T,
                              const uint32_t *d_gt_in, int64_t rank_at_end, uint32_t *d_gap, uint32_t *d_gt_out,
                              int64_t max_chains, int64_t *h_final_rank, psg_stream_stats *stats) {
  return psg_stream_gap_ctx(r, i0, last_sym, d_tail, T, 0, d_gt_in, rank_at_end, d_gap, d_gt_out, max_chains, h_final_rank, stats);
}

extern "C" int psg_stream_gap_ctx(const psg_rank_t *r, int64_t i0, int last_sym, const uint8_t *d_tail, int64_t T,
                                  int64_t ctx, const uint32_t *d_gt_in, int64_t rank_at_end, uint32_t *d_gap,
                                  uint32_t *d_gt_out, int64_t max_chains, int64_t *h_final_rank, psg_stream_stats *stats) {
  PSG_REQUIRE(d_gap, "psg_stream_gap: gap array required");
  return stream_impl(PassArgs{r, i0, last_sym, d_tail, T, ctx, d_gt_in, rank_at_end, d_gap, d_gt_out, max_chains, false, false, false, nullptr, 0, nullptr, nullptr}, h_final_rank, stats);
}
// the gap array is uninitialised on entry (PSG_GAP_UNINITIALIZED): zero-filled or overwritten by the pass
extern "C" int psg_stream_gap_ex(const psg_rank_t *r, int64_t i0, int last_sym, const uint8_t *d_tail, int64_t T,
                                 int64_t ctx, const uint32_t *d_gt_in, int64_t rank_at_end, uint32_t *d_gap,
                                 uint32_t *d_gt_out, int64_t max_chains, int flags, int64_t *h_final_rank, psg_stream_stats *stats) {
  PSG_REQUIRE(d_gap, "psg_stream_gap: gap array required");
  PSG_REQUIRE((flags & ~(PSG_GAP_UNINITIALIZED | PSG_FAIL_IF_UNRESOLVED)) == 0, "psg_stream_gap_ex: unknown flag");
  return stream_impl(PassArgs{r, i0, last_sym, d_tail, T, ctx, d_gt_in, rank_at_end, d_gap, d_gt_out, max_chains, (flags & PSG_GAP_UNINITIALIZED) != 0,
                              (flags & PSG_FAIL_IF_UNRESOLVED) != 0, false, nullptr, 0, nullptr, nullptr}, h_final_rank, stats);
}
extern "C" int psg_stream_gap_args(const psg_stream_args *a, int64_t *h_final_rank, psg_stream_stats *stats) {
  PSG_REQUIRE(a && a->d_gap, "psg_stream_gap_args: arguments and gap array required");
  PSG_REQUIRE((a->flags & ~(PSG_GAP_UNINITIALIZED | PSG_FAIL_IF_UNRESOLVED | PSG_SEARCH_ALL_STARTS)) == 0, "psg_stream_gap_args: unknown flag");
  PSG_REQUIRE(!a->search || (a->tail_begin_abs >= 0 && a->tail_begin_abs + a->tail_len + a->right_context <= a->search->n),
              "psg_stream_gap_args: the tail lies outside the text of the search context");
  return stream_impl(PassArgs{a->rank, a->block_i0, a->block_last_symbol, a->d_tail, a->tail_len, a->right_context, a->d_gt_in, a->rank_at_context_end,
                              a->d_gap, a->d_gt_out, a->max_chains, (a->flags & PSG_GAP_UNINITIALIZED) != 0, (a->flags & PSG_FAIL_IF_UNRESOLVED) != 0,
                              (a->flags & PSG_SEARCH_ALL_STARTS) != 0 && a->search != nullptr, a->search, a->tail_begin_abs, nullptr, nullptr}, h_final_rank, stats);
}

// same pass, but the ranks are handed back as a log (one u32 per streamed suffix, 0xFFFFFFFF =
// no entry, arbitrary order) instead of being counted: the multi-GPU driver partitions the log by
// owner of the gap slice and exchanges it (all-to-all) before histogramming.
extern "C" int psg_stream_gap_log(const psg_rank_t *r, int64_t i0, int last_sym, const uint8_t *d_tail, int64_t T,
                                  int64_t ctx, const uint32_t *d_gt_in, int64_t rank_at_end, uint32_t *d_gt_out,
                                  int64_t max_chains, int64_t *h_final_rank, psg_stream_stats *stats, uint32_t **d_log,
                                  int64_t *nlog) {
  PSG_REQUIRE(d_log && nlog, "psg_stream_gap_log: output pointers required");
  PSG_REQUIRE(r && r->m < 0xFFFFFFFFll, "psg_stream_gap_log: block too large for a 32-bit rank log");
  *d_log = nullptr; *nlog = 0;
  return stream_impl(PassArgs{r, i0, last_sym, d_tail, T, ctx, d_gt_in, rank_at_end, nullptr, d_gt_out, max_chains, false, false, false, nullptr, 0, d_log, nlog}, h_final_rank, stats);
}

// A pass over a long tail is cut into chunks of at most 2^31 suffixes, streamed right to left with
// the exact hand-over rank: bounds the rank log (8 + 8 GiB) and keeps every chunk in rank-log mode.
#define PSG_PASS_CHUNK ((int64_t)1 << 31)
static int stream_impl(const PassArgs &A, int64_t *h_final_rank, psg_stream_stats *stats) {
  // (a block below 2^31 symbols -- construct_sa's default -m gives 646 MiB blocks -- takes chunks of 2^29: 390 000 chains
  // of 1376 steps keep the kernel as busy as longer ones, and the rank log + partition buffers of a chunk are 6 instead
  // of 24 GiB that the arena has to get from the driver at 25-30 ms per GiB)
  int64_t chunk = A.r && A.r->m < ((int64_t)1 << 31) ? PSG_PASS_CHUNK >> 2 : PSG_PASS_CHUNK;
  if (const char *e = getenv("PSG_PASS_CHUNK")) { int64_t v = atoll(e); if (v >= 64) chunk = v / 64 * 64; }   // tests
  const i64 T = A.T;
  if (A.log_out || T <= chunk) return stream_chunk(A, h_final_rank, stats);
  psg_stream_stats acc = {};
  int64_t fin = A.rank_at_end;
  DeferredHist deferred;
  // Off unless PSG_HIST_OVERLAP=1.  Measured at configs[2]: the step gains 2.4 % (6.46 -> 6.31 s), but the
  // partition's streaming traffic and its waves slow the latency-bound stream kernel by 11 % (3.63 -> 4.01 s per step),
  // and the histogram itself stretches 3x at the occupancy the kernel leaves it -- the two do not hide behind each other.
  const char *ov = getenv("PSG_HIST_OVERLAP");
  const bool overlap = ov && !strcmp(ov, "1") && A.d_gap;
  for (int64_t u_lo = 0; u_lo < T; u_lo += chunk) {     // u = distance from the tail end
    int64_t u_hi = std::min<int64_t>(T, u_lo + chunk);
    psg_stream_stats st = {};
    // first chunk: the caller's context / start rank; later chunks start exactly where the previous one ended
    PassArgs C = A;
    C.d_tail = A.d_tail + (T - u_hi);
    C.T = u_hi - u_lo;
    C.ctx = u_lo == 0 ? A.ctx : 0;
    C.d_gt_in = A.d_gt_in ? A.d_gt_in + ((u_lo + (u_lo == 0 ? 0 : A.ctx)) >> 5) : nullptr;
    C.rank_at_end = u_lo == 0 ? A.rank_at_end : fin;
    C.d_gt_out = A.d_gt_out ? A.d_gt_out + (u_lo >> 5) : nullptr;
    C.fresh = A.fresh && u_lo == 0;
    C.tail_begin_abs = A.tail_begin_abs + (T - u_hi);
    C.defer = overlap ? &deferred : nullptr;
    int rc = stream_chunk(C, &fin, &st);
    if (rc) { (void)deferred.wait(nullptr); return rc; }
    acc.n_chains = std::max(acc.n_chains, st.n_chains); acc.chain_len = st.chain_len;
    acc.warmup_steps = std::max(acc.warmup_steps, st.warmup_steps);
    acc.unresolved += st.unresolved; acc.rounds += st.rounds;
    acc.kernel_ms += st.kernel_ms; acc.total_ms += st.total_ms; acc.hist_ms += st.hist_ms;
  }
  if (deferred.active) {                                  // the last chunk's histogram, and the excess list's overflow flag after it
    double t = 0;
    if (int rc = deferred.wait(&t)) return rc;
    acc.hist_ms += t; acc.total_ms += t;
    int gbits = 32;
    if (int rc = gap_prepare(A.d_gap, A.r->m, false, &gbits)) return rc;
    GapExcess gex = gap_excess(A.d_gap, A.r->m, gbits);
    u32 xflag = 0;
    if (gex.hdr) { if (int rc = psg::copy_d2h(&xflag, gex.hdr + 2, 4)) return rc; }
    if (xflag) { set_error("stream: excess list capacity exceeded"); return PSG_ECHECK; }
  }
  note_kernel_ms(acc.kernel_ms);
  if (h_final_rank) *h_final_rank = fin;
  if (stats) *stats = acc;
  return 0;
}

static int stream_chunk(const PassArgs &A, int64_t *h_final_rank, psg_stream_stats *stats) {
  const psg_rank_t *r = A.r;
  const i64 i0 = A.i0, T = A.T, ctx = A.ctx, rank_at_end = A.rank_at_end;
  const int last_sym = A.last_sym;
  const u8 *d_tail = A.d_tail;
  const u32 *d_gt_in = A.d_gt_in;
  u32 *d_gap = A.d_gap, *d_gt_out = A.d_gt_out;
  u32 **log_out = A.log_out;
  const bool fresh = A.fresh;
  PSG_REQUIRE(ctx >= 0 && (ctx & 63) == 0, "psg_stream_gap_ctx: right context must be a multiple of 64");
  PSG_REQUIRE(r && (d_gap || log_out), "psg_stream_gap: rank and gap required");
  PSG_REQUIRE(T >= 0 && i0 >= 0 && i0 < r->m && last_sym >= 0 && last_sym < 256, "psg_stream_gap: bad scalar argument");
  PSG_REQUIRE((rank_at_end >= 0 && rank_at_end <= r->m) || (rank_at_end == -1 && ctx > 0),
              "psg_stream_gap: rank_at_tail_end out of range (-1 = unknown is only allowed with a right context)");
  psg_stream_stats st = {};
  if (T == 0) {
    if (fresh && d_gap) PSG_HIP(hipMemsetAsync(d_gap, 0, (size_t)PSG_GAP_WORDS(r->m) * 4, stream()));
    if (h_final_rank) *h_final_rank = rank_at_end; if (stats) *stats = st; return 0;
  }
  PSG_REQUIRE(d_tail, "psg_stream_gap: tail text required");
  EventTimer total_tm, ktm;
  total_tm.start();
  // C array of the pass: compute_gap.hpp:77-85
  i64 C[256], s = 0;
  for (int c = 0; c < 256; ++c) { i64 t = r->count[c] + (c == last_sym) - (c == 0); C[c] = s; s += t; }
  DevBuf T1, tot;
  if (int rc = make_tables(r, C, T1, tot)) return rc;
  // the array's counter width / excess list (32 bits unless a test narrows it)
  int gbits = 32;
  GapExcess gex{nullptr, nullptr, 32};
  if (d_gap) {
    if (int rc = gap_prepare(d_gap, r->m, fresh, &gbits)) return rc;
    gex = gap_excess(d_gap, r->m, gbits);
  }
  // gap update mode: log + histogram needs enough work to pay for the partition
  int mode = (fresh && gbits == 32 && T < 0xFFFFFFFFll) ? 0 : 1;
  {
    const char *e = getenv("PSG_GAP_MODE");   // "atomic" | "log" | unset = auto
    bool want_log = e ? !strcmp(e, "log") : (T >= (1 << 22));
    if (e && !strcmp(e, "atomic")) want_log = false;
    if (e && !strcmp(e, "ovf")) { want_log = false; mode = 1; }   // tests: force the carry-checking atomic kernel
    const bool wide = r->m >= 0xFFFFFFFFll || getenv("PSG_LOG_WIDE") != nullptr;   // ranks need more than 32 bits (tests: forced)
    if (want_log) mode = wide ? 3 : 2;
    if (log_out) mode = 2;   // the caller wants the log itself (32-bit ranks: checked by psg_stream_gap_log)
  }
  double prev_hist_ms = 0;
  if (A.defer && A.defer->active && mode != 3) {   // this chunk updates the gap array itself: the histogram still running on the side stream comes first
    if (int rc = A.defer->wait(&prev_hist_ms)) return rc;
  }
  if (fresh && mode < 2) PSG_HIP(hipMemsetAsync(d_gap, 0, (size_t)(r->m + 1) * 4, stream()));   // the atomics need zeroes; the histogram overwrites
  // chain plan: exactly one resident wave of workgroups (a partial second wave would double the
  // pass time: every chain has the same length)
  const int cpl = chains_per_lane(r);
  i64 Ktarget = A.max_chains;
  if (Ktarget <= 0) {
    int blocks = 0, dev = 0, cus = 256;
    DISPATCH_LAYOUT(r, query_occupancy, r, mode, cpl, &blocks);
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (blocks < 1) blocks = 4;
    if (blocks > 8) blocks = 8;
    if (const char *e = getenv("PSG_STREAM_BLOCKS")) { int v = atoi(e); if (v >= 1 && v <= 8) blocks = v; }   // experiments
    Ktarget = (i64)blocks * cus * PSG_WG * cpl;
  }
  DevBuf lo_d, hi_d, fin_d, list_d, flag_d;
  i64 *lo = nullptr, *hi = nullptr, *fin = nullptr;
  std::vector<i64> list;
  std::vector<char> resolved;
  i64 L = 0, K = 0, nun = 0;
  int rc;
  // plan + warm-up; a text on which most chain starts stay open (periodic, long runs) is re-planned ONCE with few,
  // long chains: the search then resolves a few thousand starts instead of hundreds of thousands, and the
  // latency-bound stream kernel still runs thousands of chains side by side
  for (int plan = 0; plan < 2; ++plan) {
    L = cdiv(cdiv(T, Ktarget), 128) * 128;   // multiple of 128: a chain's gt_out words start on a 16-byte boundary
    K = cdiv(T, L);
    st.n_chains = K; st.chain_len = L;
    if ((rc = lo_d.alloc(K * 8)) || (rc = hi_d.alloc(K * 8)) || (rc = fin_d.alloc(K * 8)) || (rc = list_d.alloc(K * 8)) || (rc = flag_d.alloc(4))) return rc;
    PSG_HIP(hipMemsetAsync(flag_d.p, 0, 4, stream()));
    // pinned host mirrors: read-backs into pageable memory go through a slow staging path
    lo = (i64 *)pinned_buf(0, (size_t)K * 8); hi = (i64 *)pinned_buf(1, (size_t)K * 8); fin = (i64 *)pinned_buf(2, (size_t)K * 8);
    if (!lo || !hi || !fin) { set_error("stream: pinned host allocation failed"); return PSG_ENOMEM; }
    for (i64 k = 0; k < K; ++k) fin[k] = -1;
    resolved.assign((size_t)K, 0);
    WarmParams WP{d_tail, T + ctx, ctx, d_gt_in, i0, (u32)last_sym, L, 32, r->m, rank_at_end, K, nullptr, lo_d.as<i64>(), hi_d.as<i64>(), T1.as<u64>(), tot.as<u64>(), r->nsb};
    // warm-up with growing W for the chains that did not resolve; with a search context (or a caller that will come
    // back with one) two attempts suffice: a search costs about as much as a few hundred warm-up steps
    const int attempts = (A.search || A.fail_if_unresolved) ? 2 : 4;
    nun = 0;
    // A small pass with a search context skips the warm-up: W warm-up steps per chain would cost more than the pass
    // itself (K = 49 152 chains of 128 steps for a 6 Mi tail: 512 warm-up steps each), a string search per start does
    // not.  Chain 0 starts from the exact rank at the tail end.
    const bool search_all = A.search_all && A.search && K <= ((i64)1 << 17) && ctx == 0 && rank_at_end >= 0;
    if (search_all) {
      list.clear();
      lo[0] = hi[0] = rank_at_end; resolved[0] = 1;
      for (i64 k = 1; k < K; ++k) { lo[k] = 0; hi[k] = r->m; resolved[k] = 0; list.push_back(k); }
      nun = (i64)list.size();
      st.warmup_steps = 0;
      if (nun == 0) PSG_HIP(hipMemcpyAsync(lo_d.p, lo, K * 8, hipMemcpyHostToDevice, stream()));   // a single chain: its start is all there is
    }
    for (int attempt = 0; attempt < attempts && !search_all; ++attempt) {
      DISPATCH_LAYOUT(r, launch_warm, r, WP);
      PSG_HIP(hipGetLastError());
      PSG_HIP(hipMemcpyAsync(lo, lo_d.p, K * 8, hipMemcpyDeviceToHost, stream()));
      PSG_HIP(hipMemcpyAsync(hi, hi_d.p, K * 8, hipMemcpyDeviceToHost, stream()));
      PSG_HIP(psg::sync_stream());
      list.clear();
      for (i64 k = 0; k < K; ++k) { resolved[k] = lo[k] == hi[k]; if (!resolved[k]) list.push_back(k); }
      st.warmup_steps = WP.W;
      nun = (i64)list.size();
      if (nun == 0) break;
      WP.W *= 16;
      WP.nitems = nun;
      WP.list = list_d.as<i64>();
      if (int rc_ = psg::copy_h2d(list_d.p, list.data(), (size_t)(nun * 8))) return rc_;
      PSG_HIP(psg::sync_stream());
    }
    if (search_all) break;
    const i64 few = 16384;
    if (plan == 0 && A.max_chains <= 0 && nun > K / 4 && K > few) { Ktarget = few; continue; }
    break;
  }
  st.unresolved = nun;
  if (nun > 0 && A.search) {
    // K8: the open chain starts by string search.  Chain k starts at text position tail_end - k * L.
    const i64 tail_end_abs = A.tail_begin_abs + T;
    i64 *pos = (i64 *)pinned_buf(5, (size_t)nun * 8), *rk = (i64 *)pinned_buf(6, (size_t)nun * 8);
    if (!pos || !rk) { set_error("stream: pinned host allocation failed"); return PSG_ENOMEM; }
    for (i64 q = 0; q < nun; ++q) pos[q] = tail_end_abs - list[(size_t)q] * L;
    DevBuf pos_d, rk_d;
    if ((rc = pos_d.alloc(nun * 8)) || (rc = rk_d.alloc(nun * 8))) return rc;
    PSG_HIP(hipMemcpyAsync(pos_d.p, pos, nun * 8, hipMemcpyHostToDevice, stream()));
    if ((rc = psg::search_ranks_launch(A.search, pos_d.as<i64>(), nun, rk_d.as<i64>()))) return rc;
    PSG_HIP(hipMemcpyAsync(rk, rk_d.p, nun * 8, hipMemcpyDeviceToHost, stream()));
    PSG_HIP(psg::sync_stream());
    if (A.search->text_end > 0 && (rc = psg::search_window_check())) return rc;
    for (i64 q = 0; q < nun; ++q) {
      const i64 k = list[(size_t)q];
      if (rk[q] < lo[k] || rk[q] > hi[k]) {   // the warm-up interval always contains the true rank
        set_error("stream: searched start rank of chain " + std::to_string(k) + " (" + std::to_string(rk[q]) + ") lies outside its warm-up interval [" +
                  std::to_string(lo[k]) + ", " + std::to_string(hi[k]) + "]");
        return PSG_ECHECK;
      }
      lo[k] = hi[k] = rk[q];
      resolved[k] = 1;
    }
    PSG_HIP(hipMemcpyAsync(lo_d.p, lo, K * 8, hipMemcpyHostToDevice, stream()));
    nun = 0;
  }
  if (nun > 0 && A.fail_if_unresolved) {
    set_error("stream: " + std::to_string(nun) + " of " + std::to_string(K) + " chain starts not determined by the warm-up (pass a search context)");
    return PSG_EUNRESOLVED;
  }
  if (!resolved[0]) { set_error("stream: start rank of the first chain not determined inside the right context (text too repetitive for this context length)"); return PSG_ECHECK; }
  DevBuf log_d, loghi_d;
  if (mode >= 2) {
    if ((rc = log_d.alloc(K * L * 4))) return rc;   // every entry is written by its chain (0xFFFFFFFF = no entry)
    if (mode == 3 && (rc = loghi_d.alloc(K * L))) return rc;
  }
  StreamParams SP{d_tail, T + ctx, ctx, d_gt_in, d_gt_out, d_gap, i0, (u32)last_sym, L, K, nullptr, lo_d.as<i64>(), fin_d.as<i64>(), T1.as<u64>(), tot.as<u64>(), r->nsb, flag_d.as<int>(), log_d.as<u32>(), K, loghi_d.as<u32>(), gex, 0, 0};
  double kms = 0;
  i64 ndone = 0;
  // rounds: every chain whose start rank is known runs; an unresolved chain k becomes
  // known once chain k-1 (to its right in the text) has finished: fin[k-1] == init[k].
  // Host work per round is proportional to the chains of that round (a text whose chains all wait for their
  // right neighbour runs K rounds of one chain: scanning all K chains per round would be quadratic).
  std::vector<i64> ready, next, need;
  const bool all_first = nun == 0;   // the usual case: every start rank came out of the warm-up (or the search)
  if (!all_first)
    for (i64 k = 0; k < K; ++k) if (resolved[k]) ready.push_back(k);
  const i64 SMALL = 256;             // up to this many values are moved one by one instead of as whole arrays
  while (ndone < K) {
    const bool all_ready = all_first || (i64)ready.size() == K;
    if (!all_ready && ready.empty()) { set_error("stream: no runnable chain (internal error)"); return PSG_ECHECK; }
    if (all_ready) { SP.list = nullptr; SP.nchains = K; }
    else {
      if (int rc_ = psg::copy_h2d(list_d.p, ready.data(), (size_t)(ready.size() * 8))) return rc_;
      SP.list = list_d.as<i64>(); SP.nchains = (i64)ready.size();
    }
    ktm.start();
    DISPATCH_LAYOUT(r, launch_stream, r, SP, mode, cpl);
    ktm.stop();
    PSG_HIP(hipGetLastError());
    // final ranks that start a waiting chain: chain k+1 unresolved, chain k just run
    need.clear();
    if (!all_ready) for (i64 k : ready) if (k + 1 < K && !resolved[k + 1]) need.push_back(k);
    if ((i64)need.size() > SMALL) {
      PSG_HIP(hipMemcpyAsync(fin, fin_d.p, K * 8, hipMemcpyDeviceToHost, stream()));
      PSG_HIP(psg::sync_stream());
    } else {
      for (i64 k : need) if (int rc_ = psg::copy_d2h(&fin[k], fin_d.as<i64>() + k, 8)) return rc_;
      PSG_HIP(psg::sync_stream());
    }
    kms += ktm.ms();
    ndone += all_ready ? K : (i64)ready.size();
    st.rounds++;
    next.clear();
    for (i64 k : need) { lo[k + 1] = fin[k]; resolved[k + 1] = 1; next.push_back(k + 1); }
    if (!next.empty()) {   // the patched start ranks go to the device
      if ((i64)next.size() > SMALL) PSG_HIP(hipMemcpyAsync(lo_d.p, lo, K * 8, hipMemcpyHostToDevice, stream()));
      else for (i64 k : next) if (int rc_ = psg::copy_h2d(lo_d.as<i64>() + k, &lo[k], 8)) return rc_;
    }
    ready.swap(next);
  }
  PSG_HIP(hipMemcpyAsync(fin, fin_d.p, K * 8, hipMemcpyDeviceToHost, stream()));   // all final ranks for the check below
  PSG_HIP(psg::sync_stream());
  // invariant: the rank a chain ends with is the start rank of the next chain
  for (i64 k = 1; k < K; ++k)
    if (fin[k - 1] != lo[k]) {
      set_error("stream: chain hand-over check failed at chain " + std::to_string(k) + " (fin=" + std::to_string(fin[k - 1]) + " init=" + std::to_string(lo[k]) + ")");
      return PSG_ECHECK;
    }
  if (ctx == 0 && lo[0] != rank_at_end) { set_error("stream: chain 0 did not start at rank_at_tail_end"); return PSG_ECHECK; }
  double hist_ms = 0;
  if (mode == 2 && log_out) {       // hand the log to the caller (ownership moves; psg_free)
    *log_out = log_d.as<u32>();
    *A.nlog_out = K * L;
    log_d.p = nullptr;
  } else if (mode == 2) {
    if ((rc = psg::gap_hist_from_log(log_d.as<u32>(), K * L, r->m, d_gap, &hist_ms, fresh, gex))) return rc;
    log_d.alloc(16);   // give the log back to the pool before returning
  } else if (mode == 3 && A.defer && psg::gap_hist_wide_one_slab(r->m)) {
    // behind the next chunk's kernel: the previous chunk's job (same side stream, its buffers) is collected first
    if ((rc = A.defer->wait(&prev_hist_ms))) return rc;
    std::swap(A.defer->log_lo.p, log_d.p); std::swap(A.defer->log_lo.bytes, log_d.bytes);
    std::swap(A.defer->log_hi.p, loghi_d.p); std::swap(A.defer->log_hi.bytes, loghi_d.bytes);
    {
      StreamScope sc(side_stream());
      if ((rc = psg::gap_hist_wide_launch(A.defer->job, A.defer->log_lo.as<u32>(), A.defer->log_hi.as<u8>(), K * L, r->m, d_gap, fresh, gex))) return rc;
    }
    A.defer->active = true;
  } else if (mode == 3) {
    if (A.defer && (rc = A.defer->wait(&prev_hist_ms))) return rc;
    if ((rc = psg::gap_hist_from_wide_log(log_d, loghi_d, K * L, r->m, d_gap, &hist_ms, fresh, gex))) return rc;
  }
  st.hist_ms = hist_ms + prev_hist_ms;
  u32 xflag = 0;
  if (gex.hdr) PSG_HIP(hipMemcpyAsync(pinned_buf(3, 64), gex.hdr + 2, 4, hipMemcpyDeviceToHost, stream()));   // pinned: pageable read-backs stall
  total_tm.stop();
  PSG_HIP(psg::sync_stream());
  if (gex.hdr) memcpy(&xflag, pinned_buf(3, 64), 4);
  if (xflag) { set_error("stream: excess list capacity exceeded"); return PSG_ECHECK; }
  st.kernel_ms = kms;
  st.total_ms = total_tm.ms();
  note_kernel_ms(kms);
  if (h_final_rank) *h_final_rank = fin[K - 1];
  if (stats) *stats = st;
  return 0;
}
