// rank_sm.hpp -- "symbol-major" rank layout: ONE 16-byte load per query for general alphabets.
// (included by rank_stream.hip)
//
// The interleaved-block layout needs two HBM sectors per query (counter sector + data sector) and
// MI355X serves ~51 G random sectors/s (profiles/r01_membench.txt), so the stream kernel saturates
// at ~22 G steps/s.  Here every symbol c owns an array E_c of 16-byte entries, one per BUCKET of
// the BWT, that holds the running count AND the positions of c inside the bucket:
//   LIST   (rare symbols, bucket = 256 positions):
//            word0 = #c before the bucket ; byte4 = n ; bytes 5..15 = positions (0xFF = unused)
//            n > 11  ->  byte4 = 0xFF, word2 = index of a 256-bit bitmap in the overflow pool
//   BITMAP (frequent symbols, bucket = 96 positions):
//            word0 = #c before the bucket ; words 1,2,3 = 96-bit occupancy bitmap
//            (round 2 kept 64 positions per entry and left word 1 unused: a third more table for the same one load --
//             and the chip's random-sector rate drops with the size of the table, profiles/r02_membench_footprint.txt)
// rank(i, c) = word0 + #positions < (i mod bucket): one sector.  Space = 16 B x (m/256 | m/96) per
// symbol: 16 B/symbol of text for a uniform byte alphabet (the block layout with B=32 needs 33).
// Superblocks (the counts inside the entries are relative to them) and build segments are multiples of
// lcm(256, 96) = 768 positions, so that no bucket of either kind straddles one.
// Semantics are those of rank4n<>::rank (rank.hpp:566-568).
#pragma once

#define SM_ABSENT 0u
#define SM_LIST 1u
#define SM_BITMAP 2u
#define SM_LIST8 3u          // 8-byte entries: word0 = #c before the bucket, bytes 4..7 = positions (0xFF = unused);
                             // more than 4 occurrences: word0 |= 2^31, word1 = index of a 256-bit bitmap in the pool
#define SM_CAP8 4
#define SM_MODE_SHIFT 62
#define SM_OFF_MASK ((1ull << SM_MODE_SHIFT) - 1ull)
#define SM_SEG 3072          // positions per build segment: 12 LIST buckets of 256 = 32 BITMAP buckets of 96
#define SM_LB (SM_SEG / 256) // LIST buckets per segment
#define SM_BW 96             // positions per BITMAP bucket
#define SM_BB (SM_SEG / SM_BW)   // BITMAP buckets per segment
#define SM_SB_SEGS_DEFAULT 699050   // segments per superblock: 3072 * 699050 = 2 147 481 600 < 2^31 (LIST8 keeps bit 31 of its count as a flag)
#define SM_CAP 11            // positions that fit into a LIST entry

// rank inside a loaded entry: e = E_c[bucket], off = i mod bucket size, t2 = descriptor of the symbol
__device__ __forceinline__ u32 sm_pool_rank(const u8 *pool, u32 idx, u32 off) {   // ones below bit `off` of pool bitmap idx
  const uint4 *bp = (const uint4 *)(pool + (size_t)idx * 32);
  uint4 a = bp[0], b = bp[1];
  u32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  u32 cnt = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int rem = (int)off - 32 * k;
    u32 mask = rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << rem) - 1u));
    cnt += __popc(w[k] & mask);
  }
  return cnt;
}

__device__ __forceinline__ u32 sm_rank_entry(const uint4 &e, const u8 *pool, u64 t2, u32 off) {
  const u32 mode = (u32)(t2 >> SM_MODE_SHIFT);
  if (mode == SM_BITMAP) {   // ones below bit `off` (< 96) of the bitmap e.y | e.z << 32 | e.w << 64
    const u32 m0 = off >= 32 ? 0xFFFFFFFFu : ((1u << off) - 1u);
    const u32 m1 = off >= 64 ? 0xFFFFFFFFu : (off > 32 ? ((1u << (off - 32)) - 1u) : 0u);
    const u32 m2 = off > 64 ? ((1u << (off - 64)) - 1u) : 0u;
    return e.x + (u32)__popc(e.y & m0) + (u32)__popc(e.z & m1) + (u32)__popc(e.w & m2);
  }
  if (mode == SM_LIST8) {   // e.x = count (| 2^31), e.y = four positions or the pool index
    if (e.x & 0x80000000u) return (e.x & 0x7FFFFFFFu) + sm_pool_rank(pool, e.y, off);
    u32 cnt = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) cnt += ((e.y >> (8 * j)) & 255u) < off;   // unused slots hold 0xFF, never < off (off <= 255)
    return e.x + cnt;
  }
  u32 n = e.y & 255u;
  if (n != 0xFFu) {   // unused slots hold 0xFF, which is never < off (off <= 255)
    u32 cnt = 0;
#pragma unroll
    for (int j = 1; j < 4; ++j) cnt += ((e.y >> (8 * j)) & 255u) < off;
#pragma unroll
    for (int j = 0; j < 4; ++j) cnt += ((e.z >> (8 * j)) & 255u) < off;
#pragma unroll
    for (int j = 0; j < 4; ++j) cnt += ((e.w >> (8 * j)) & 255u) < off;
    return e.x + cnt;
  }
  return e.x + sm_pool_rank(pool, e.z, off);   // dense bucket: 256-bit bitmap in the pool
}

// bit k of the result = (byte k of the 4-byte word == c)
__device__ __forceinline__ u32 sm_eq_nibble(u32 w, u32 c4) {
  u32 y = (swar_eq_mask(w, c4) >> 7) & 0x01010101u;
  return (y & 1u) | ((y >> 7) & 2u) | ((y >> 14) & 4u) | ((y >> 21) & 8u);
}

#define SM_MAX_BITMAP 32     // symbols that may use BITMAP mode (LDS budget of the fill kernel)

// One workgroup per segment of 4096 positions.  Entries are assembled in LDS and written out with
// neighbouring lanes covering neighbouring entries of the same symbol (128-byte / 1-KiB runs):
// a thread-per-symbol store pattern would issue 2 G fully divergent 16-byte stores for a 2 GiB BWT.
#define SM_NB 6                                  // LIST buckets per phase (two phases per segment): runs of 6 entries = 96 bytes per symbol
#define SM_OUT_STRIDE (256 + 2)                  // padded so the (symbol, bucket) -> lane transpose reads conflict-free
struct SmListLds {                               // LIST phase (SM_NB buckets of 256 positions), ~33 KiB -> 4 workgroups per CU
  // slot (bucket j, symbol c): {count, 12 position bytes}; rewritten IN PLACE into the finished entry
  uint4 slot[SM_NB * SM_OUT_STRIDE];
};
struct SmList8Lds {                              // LIST8 phase: all 12 buckets of the segment at once, 8-byte slots (~25 KiB)
  uint2 slot[SM_LB * SM_OUT_STRIDE];             // {count, 4 position bytes} -> the finished entry, in place
};
struct SmBitmapLds {                             // BITMAP phase (32 sub-buckets of 96)
  u32 b0[SM_MAX_BITMAP * SM_BB], b1[SM_MAX_BITMAP * SM_BB], b2[SM_MAX_BITMAP * SM_BB];
  u32 cum[SM_MAX_BITMAP * SM_BB];
};
union SmPhaseLds { SmListLds L; SmList8Lds L8; SmBitmapLds B; };

// Dense buckets (more occurrences than an entry holds inline) of one phase of one workgroup: collected in a
// list, ONE global atomic reserves their pool slots (a global atomic per bucket serialises on the cursor:
// 8 M of them took 65 ms), then 8 lanes per bucket build the 256-bit bitmap.
#define SM_DENSE_CAP 1024
struct SmDense {
  u32 n;          // dense buckets of this phase
  u32 base;       // first pool slot of this phase
  u16 item[SM_DENSE_CAP];   // (bucket within the segment) << 8 | symbol
};
// workgroup-wide; call between the barrier after the build loop and the store loop.  patch(j, c, idx) stores the
// final pool index into the entry of (bucket j of this phase, symbol c).
template <class Patch>
__device__ __forceinline__ void sm_dense_flush(SmDense &D, const u8 *sym, i64 base, i64 m, int first_bucket, u32 *pool, u32 *pool_cursor,
                                               u32 pool_cap, int *err, Patch patch) {
  __syncthreads();
  const u32 nd = D.n < SM_DENSE_CAP ? D.n : SM_DENSE_CAP;
  if (threadIdx.x == 0) {
    if (D.n > SM_DENSE_CAP) *err = 1;
    D.base = nd ? atomicAdd(pool_cursor, nd) : 0u;
  }
  __syncthreads();
  const u32 pbase = D.base;
  const int w = threadIdx.x & 7;
  for (u32 it = threadIdx.x >> 3; it < nd; it += 32) {
    const u32 j = D.item[it] >> 8, c = D.item[it] & 255u, idx = pbase + it;
    if (idx < pool_cap) {
      const u32 *bs = (const u32 *)(sym + (first_bucket + (int)j) * 256) + w * 8;
      const u32 c4 = c * 0x01010101u;
      u32 bits = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) bits |= sm_eq_nibble(bs[k], c4) << (4 * k);
      const i64 lim = m - (base + (first_bucket + (int)j) * 256) - 32 * w;   // valid positions from this word on
      if (lim < 32) bits &= lim <= 0 ? 0u : ((1u << lim) - 1u);
      pool[(size_t)idx * 8 + w] = bits;
    } else *err = 1;
    if (w == 0) patch((int)j, (int)c, idx);
  }
  __syncthreads();
  if (threadIdx.x == 0) D.n = 0;
}

// occurrences of every symbol before superblock sb (superblock = sbs segments): workgroup sb, thread c
__global__ __launch_bounds__(256) void sm_sb_base_kernel(const u32 *seg_pref, const u64 *group_base, i64 sbs, u64 *sb_base) {
  const i64 seg0 = (i64)blockIdx.x * sbs;
  sb_base[(i64)blockIdx.x * 256 + threadIdx.x] = group_base[(seg0 / GROUP_SEGS) * 256 + threadIdx.x] + seg_pref[seg0 * 256 + threadIdx.x];
}

// L8: the LIST symbols of this structure use 8-byte entries (SM_LIST8) -- all of them or none
// sbs: segments per superblock; the counts stored in the entries are relative to the superblock start
template <bool L8>
__global__ __launch_bounds__(256) void sm_fill_kernel(const u8 *bwt, i64 m, const u64 *t2g, const u32 *seg_pref, const u64 *group_base,
                                                       uint4 *entries, u32 *pool, u32 *pool_cursor, u32 pool_cap, int *err, i64 sbs, const u32 *seg_mask) {
  // seg_mask (optional): one bit per segment, 0 = no query will ever read this segment's entries (the right children of a
  // level of psg_merge_leaves are only streamed, never ranked): their entries stay unwritten
  if (seg_mask && !((seg_mask[blockIdx.x >> 5] >> (blockIdx.x & 31)) & 1u)) return;
  __shared__ __attribute__((aligned(16))) u8 sym[SM_SEG];
  __shared__ __attribute__((aligned(16))) SmPhaseLds P;
  __shared__ u64 t2S[256];
  __shared__ u8 bmSym[SM_MAX_BITMAP];            // symbols in BITMAP mode
  __shared__ int nbm;
  __shared__ SmDense D;
  const int c = threadIdx.x;
  const i64 seg = blockIdx.x, base = seg * SM_SEG;
  const u64 t2 = t2g[c];
  const u32 mymode = (u32)(t2 >> SM_MODE_SHIFT);
  t2S[c] = t2;
  if (c == 0) { nbm = 0; D.n = 0; }
  if (base + SM_SEG <= m && ((uintptr_t)bwt & 15) == 0) {
    if (c < SM_SEG / 16) ((uint4 *)sym)[c] = ((const uint4 *)(bwt + base))[c];
  } else {
    for (int k = c; k < SM_SEG; k += 256) sym[k] = base + k < m ? bwt[base + k] : 0;
  }
  __syncthreads();
  if (mymode == SM_BITMAP) { int k = atomicAdd(&nbm, 1); if (k < SM_MAX_BITMAP) bmSym[k] = (u8)c; }
  const i64 seg0 = seg / sbs * sbs;       // first segment of this superblock
  u32 run = (u32)(group_base[(seg / GROUP_SEGS) * 256 + c] + seg_pref[seg * 256 + c] - (group_base[(seg0 / GROUP_SEGS) * 256 + c] + seg_pref[seg0 * 256 + c]));
  // ---------------- LIST8 symbols: one phase over the 12 buckets, runs of 12 entries = 96 bytes ----------------
  if (L8) {
    for (int k = c; k < SM_LB * SM_OUT_STRIDE; k += 256) P.L8.slot[k] = make_uint2(0u, ~0u);
    __syncthreads();
    for (int j = 0; j < SM_LB; ++j) {
      int q = j * 256 + c;
      if (base + q < m) {
        u32 s = sym[q];
        if ((u32)(t2S[s] >> SM_MODE_SHIFT) == SM_LIST8) {
          uint2 *sl = &P.L8.slot[j * SM_OUT_STRIDE + s];
          u32 k = atomicAdd(&sl->x, 1u);
          if (k < SM_CAP8) ((u8 *)sl)[4 + k] = (u8)c;
        }
      }
    }
    __syncthreads();
    if (mymode == SM_LIST8) {
      for (int j = 0; j < SM_LB; ++j) {
        const uint2 raw = P.L8.slot[j * SM_OUT_STRIDE + c];
        const u32 n = raw.x;
        uint2 e = make_uint2(run, raw.y);
        if (n > SM_CAP8) {   // dense bucket: 256-bit bitmap in the overflow pool (index patched in by sm_dense_flush)
          u32 k = atomicAdd(&D.n, 1u);
          if (k < SM_DENSE_CAP) D.item[k] = (u16)((j << 8) | c);
          e = make_uint2(run | 0x80000000u, 0u);
        }
        P.L8.slot[j * SM_OUT_STRIDE + c] = e;
        run += n;
      }
    }
    sm_dense_flush(D, sym, base, m, 0, pool, pool_cursor, pool_cap, err, [&](int j, int cc, u32 idx) { P.L8.slot[j * SM_OUT_STRIDE + cc].y = idx; });
    for (int idx = c; idx < 256 * SM_LB; idx += 256) {   // 12 consecutive lanes write the 12 entries of one symbol
      int s = idx / SM_LB, j = idx % SM_LB;
      u64 ts = t2S[s];
      i64 bk = seg * SM_LB + j;
      if ((u32)(ts >> SM_MODE_SHIFT) == SM_LIST8 && bk * 256 < m) ((uint2 *)(entries + (ts & SM_OFF_MASK)))[bk] = P.L8.slot[j * SM_OUT_STRIDE + s];
    }
    __syncthreads();
  }
  // ---------------- LIST symbols: SM_LB/SM_NB phases of SM_NB buckets ----------------
  for (int ph = 0; !L8 && ph < SM_LB / SM_NB; ++ph) {
    for (int k = c; k < SM_NB * SM_OUT_STRIDE; k += 256) P.L.slot[k] = make_uint4(0u, ~0u, ~0u, ~0u);
    __syncthreads();
    for (int j = 0; j < SM_NB; ++j) {
      int q = (ph * SM_NB + j) * 256 + c;
      if (base + q < m) {
        u32 s = sym[q];
        if ((u32)(t2S[s] >> SM_MODE_SHIFT) == SM_LIST) {
          uint4 *sl = &P.L.slot[j * SM_OUT_STRIDE + s];
          u32 k = atomicAdd(&sl->x, 1u);
          if (k < SM_CAP) ((u8 *)sl)[4 + k] = (u8)c;   // position inside the bucket = c
        }
      }
    }
    __syncthreads();
    if (mymode == SM_LIST) {
      for (int j = 0; j < SM_NB; ++j) {
        const uint4 raw = P.L.slot[j * SM_OUT_STRIDE + c];
        const u32 n = raw.x;
        uint4 e;
        e.x = run;
        if (n <= SM_CAP) {
          e.y = n | (raw.y << 8);
          e.z = (raw.y >> 24) | (raw.z << 8);
          e.w = (raw.z >> 24) | (raw.w << 8);
        } else {   // dense bucket: 256-bit bitmap in the overflow pool (index patched in by sm_dense_flush)
          u32 k = atomicAdd(&D.n, 1u);
          if (k < SM_DENSE_CAP) D.item[k] = (u16)((j << 8) | c);
          e.y = 0xFFu; e.z = 0; e.w = 0;
        }
        P.L.slot[j * SM_OUT_STRIDE + c] = e;
        run += n;
      }
    }
    sm_dense_flush(D, sym, base, m, ph * SM_NB, pool, pool_cursor, pool_cap, err, [&](int j, int cc, u32 idx) { P.L.slot[j * SM_OUT_STRIDE + cc].z = idx; });
    for (int idx = c; idx < 256 * SM_NB; idx += 256) {   // SM_NB consecutive lanes write consecutive entries of one symbol
      int s = idx / SM_NB, j = idx % SM_NB;
      u64 ts = t2S[s];
      i64 bk = seg * SM_LB + ph * SM_NB + j;
      if ((u32)(ts >> SM_MODE_SHIFT) == SM_LIST && bk * 256 < m) entries[(ts & SM_OFF_MASK) + bk] = P.L.slot[j * SM_OUT_STRIDE + s];
    }
    __syncthreads();
  }
  // ---------------- BITMAP symbols: (symbol, sub-bucket of 96 positions) items ----------------
  const int nb = nbm < SM_MAX_BITMAP ? nbm : SM_MAX_BITMAP;
  for (int item = c; item < nb * SM_BB; item += 256) {
    int s = bmSym[item / SM_BB], sub = item % SM_BB;
    const u32 *ws = (const u32 *)(sym + sub * SM_BW);
    u32 s4 = (u32)s * 0x01010101u;
    u32 b[3] = {0u, 0u, 0u};
#pragma unroll
    for (int w = 0; w < SM_BW / 4; ++w) b[w >> 3] |= sm_eq_nibble(ws[w], s4) << (4 * (w & 7));
    i64 lim = m - (base + sub * SM_BW);       // valid positions of this sub-bucket
    if (lim < SM_BW) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { const i64 l = lim - 32 * k; if (l < 32) b[k] &= l <= 0 ? 0u : ((1u << l) - 1u); }
    }
    P.B.b0[item] = b[0]; P.B.b1[item] = b[1]; P.B.b2[item] = b[2];
  }
  __syncthreads();
  if (mymode == SM_BITMAP) {   // running counts over the sub-buckets of this symbol
    int k = 0;
    while (k < nb && bmSym[k] != c) ++k;
    if (k < nb)
      for (int sub = 0; sub < SM_BB; ++sub) {
        const int it = k * SM_BB + sub;
        P.B.cum[it] = run;
        run += (u32)__popc(P.B.b0[it]) + (u32)__popc(P.B.b1[it]) + (u32)__popc(P.B.b2[it]);
      }
  }
  __syncthreads();
  for (int item = c; item < nb * SM_BB; item += 256) {
    int s = bmSym[item / SM_BB], sub = item % SM_BB;
    i64 bk = seg * SM_BB + sub;
    if (bk * SM_BW < m) entries[(t2S[s] & SM_OFF_MASK) + bk] = make_uint4(P.B.cum[item], P.B.b0[item], P.B.b1[item], P.B.b2[item]);
  }
}
