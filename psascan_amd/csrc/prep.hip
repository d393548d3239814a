// prep.hip -- synthetic-input preparation for bench.py and the full-size property tests.
// NOT part of the drop-in boundary: the reference sorts half-blocks on the host
// (inmem_psascan.hpp:64-304) and so does construct_sa.  To time the hot path at
// BASELINE.json's full sizes the bench needs valid (partial SA, BWT, i0, gt_begin) inputs
// for multi-GiB half-blocks within minutes, so for texts with short repeats (uniform random
// bytes, i.i.d. DNA) they are produced on the device: sort by a packed 64-bit prefix key
// (rocPRIM radix sort) and finish the rare equal-key groups by direct suffix comparison.
#include "dev_common.hpp"

#include <cmath>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include <algorithm>
#include <vector>

using namespace psg;

__device__ __forceinline__ u64 splitmix64(u64 x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// mode 0: uniform bytes 0..254 ; mode 1: DNA (ACGT) ; mode 2: uniform over `sigma` symbols starting at 'a'
__global__ __launch_bounds__(PSG_WG) void gen_text_kernel(u8 *text, i64 n, int mode, int sigma, u64 seed) {
  i64 k = ((i64)blockIdx.x * PSG_WG + threadIdx.x) * 8;
  if (k >= n) return;
  u64 r = splitmix64(seed * 0x100000001B3ull + (u64)(k >> 3));
  u64 r2 = splitmix64(r);
  for (int q = 0; q < 8 && k + q < n; ++q) {
    u32 v = (u32)((q < 4 ? r >> (16 * q) : r2 >> (16 * (q - 4))) & 0xFFFF);
    u8 c;
    if (mode == 0) c = (u8)((v * 255u) >> 16);
    else if (mode == 1) c = (u8)("ACGT"[v & 3]);
    else c = (u8)('a' + (v * (u32)sigma >> 16));
    text[k + q] = c;
  }
}

// mode 3: "English-like" text (BASELINE configs[2]; no corpus ships with the image): words drawn from a Zipfian
// table of 4096 words over a skewed 26-letter alphabet, separated by blanks, a full stop now and then -- sigma = 28,
// frequent symbols next to rare ones, repeats of tens of symbols.  Every thread fills one 256-byte segment from its
// own seed (words are cut at segment ends), so the text is a pure function of (seed, n).
#define ENG_WORDS 4096
#define ENG_SEG 256
struct EngTable { u32 cum[ENG_WORDS]; u16 off[ENG_WORDS + 1]; u8 letters[ENG_WORDS * 10]; };
__global__ __launch_bounds__(PSG_WG) void gen_english_kernel(u8 *text, i64 n, const EngTable *tab, u64 seed) {
  __shared__ __attribute__((aligned(16))) u8 buf[PSG_WG][ENG_SEG + 16];   // +16: rows on different banks
  const i64 seg = (i64)blockIdx.x * PSG_WG + threadIdx.x;
  u8 *b = buf[threadIdx.x];
  u64 r = splitmix64(seed * 0x100000001B3ull + (u64)seg);
  int len = 0;
  while (len < ENG_SEG) {
    r = splitmix64(r);
    const u32 x = (u32)(r >> 32);
    int lo = 0, hi = ENG_WORDS - 1;                      // first word with cum >= x
    while (lo < hi) { int md = (lo + hi) >> 1; if (tab->cum[md] >= x) hi = md; else lo = md + 1; }
    const int o = tab->off[lo], l = tab->off[lo + 1] - o;
    for (int q = 0; q < l && len < ENG_SEG; ++q) b[len++] = tab->letters[o + q];
    if (len < ENG_SEG) b[len++] = ((u32)r & 15u) == 0 ? '.' : ' ';
    if (len < ENG_SEG && b[len - 1] == '.') b[len++] = ' ';
  }
  __syncthreads();
  // coalesced copy-out: 16 lanes x 16 bytes per segment
  const i64 base = (i64)blockIdx.x * PSG_WG * ENG_SEG;
  for (int k = threadIdx.x; k < PSG_WG * (ENG_SEG / 16); k += PSG_WG) {
    const int sg = k / (ENG_SEG / 16), q = k % (ENG_SEG / 16);
    const i64 p = base + (i64)sg * ENG_SEG + q * 16;
    if (p + 16 <= n) *(uint4 *)(text + p) = *(const uint4 *)(buf[sg] + q * 16);
    else for (int j = 0; j < 16 && p + j < n; ++j) text[p + j] = buf[sg][q * 16 + j];
  }
}

static void build_english_table(EngTable &T) {
  u64 r = 0x5EED5EEDull;
  auto next = [&] { r += 0x9E3779B97F4A7C15ull; u64 x = r; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); };
  const char *letters = "etaoinshrdlcumwfgypbvkjxqz";
  double lp[26], lsum = 0;
  for (int k = 0; k < 26; ++k) { lp[k] = 1.0 / std::pow(k + 1.0, 0.9); lsum += lp[k]; }
  int o = 0;
  for (int w = 0; w < ENG_WORDS; ++w) {
    T.off[w] = (u16)o;
    const int l = 2 + (int)(next() % 8);
    for (int q = 0; q < l; ++q) {
      double x = (double)(next() >> 11) / 9007199254740992.0 * lsum;
      int k = 0;
      while (k < 25 && x >= lp[k]) { x -= lp[k]; ++k; }
      T.letters[o++] = (u8)letters[k];
    }
  }
  T.off[ENG_WORDS] = (u16)o;
  double wsum = 0, run = 0;
  for (int w = 0; w < ENG_WORDS; ++w) wsum += 1.0 / (w + 1.0);
  for (int w = 0; w < ENG_WORDS; ++w) { run += 1.0 / (w + 1.0) / wsum; T.cum[w] = w == ENG_WORDS - 1 ? 0xFFFFFFFFu : (u32)std::min(4294967295.0, run * 4294967296.0); }
}

extern "C" int psgx_gen_text(uint8_t *d_text, int64_t n, int mode, int sigma, uint64_t seed) {
  PSG_REQUIRE(d_text && n >= 0 && mode >= 0 && mode <= 3, "psgx_gen_text");
  if (n == 0) return 0;
  if (mode == 3) {
    PSG_REQUIRE(((uintptr_t)d_text & 15) == 0, "psgx_gen_text: text must be 16-byte aligned");
    static EngTable T;
    static bool built = false;
    if (!built) { build_english_table(T); built = true; }
    DevBuf tab;
    if (int rc = tab.alloc(sizeof(EngTable))) return rc;
    if (int rc = psg::copy_h2d(tab.p, &T, sizeof(EngTable))) return rc;
    hipLaunchKernelGGL(gen_english_kernel, dim3((unsigned)cdiv(cdiv(n, ENG_SEG), PSG_WG)), dim3(PSG_WG), 0, stream(), d_text, n, tab.as<EngTable>(), seed);
    PSG_HIP(hipGetLastError());
    PSG_HIP(psg::sync_stream());
    return 0;
  }
  hipLaunchKernelGGL(gen_text_kernel, dim3((unsigned)cdiv(cdiv(n, 8), PSG_WG)), dim3(PSG_WG), 0, stream(), d_text, n, mode, sigma, seed);
  PSG_HIP(hipGetLastError());
  PSG_HIP(psg::sync_stream());
  return 0;
}

struct KeyCfg { int bits, per_key; u8 code[256]; };

// flags[c] = 1 for every byte value that occurs in text[0..n)
__global__ __launch_bounds__(PSG_WG) void present_kernel(const u8 *text, i64 n, u32 *flags) {
  __shared__ u32 f[256];
  f[threadIdx.x] = 0;
  __syncthreads();
  for (i64 k = ((i64)blockIdx.x * PSG_WG + threadIdx.x) * 16; k < n; k += (i64)gridDim.x * PSG_WG * 16) {
    if (k + 16 <= n && ((uintptr_t)text & 15) == 0) {
      const uint4 v = *(const uint4 *)(text + k);
      const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 16; ++q) f[(w[q >> 2] >> (8 * (q & 3))) & 255u] = 1u;
    } else {
      for (i64 j = k; j < n && j < k + 16; ++j) f[text[j]] = 1u;
    }
  }
  __syncthreads();
  if (f[threadIdx.x]) flags[threadIdx.x] = 1u;
}

__global__ __launch_bounds__(PSG_WG) void make_keys_kernel(const u8 *text, i64 n, i64 beg, i64 size, KeyCfg cfg, u64 *keys, u32 *idx) {
  i64 s = (i64)blockIdx.x * PSG_WG + threadIdx.x;
  if (s >= size) return;
  u64 key = 0;
  i64 p = beg + s;
  for (int q = 0; q < cfg.per_key; ++q) {
    u64 v = p + q < n ? (u64)cfg.code[text[p + q]] + 1 : 0;   // 0 = past the end of the text (smallest)
    key = (key << cfg.bits) | v;
  }
  keys[s] = key;
  idx[s] = (u32)s;
}

// text[a..n) < text[b..n) ?  (a != b), comparison starts at offset `skip`
__device__ bool suffix_less(const u8 *text, i64 n, i64 a, i64 b, i64 skip) {
  i64 k = skip;
  while (a + k < n && b + k < n) {
    u8 x = text[a + k], y = text[b + k];
    if (x != y) return x < y;
    ++k;
  }
  return a + k >= n && b + k < n ? true : (a + k >= n && b + k >= n ? a > b : false);
}

// the same with a bound on the work: after `budget` more equal symbols the comparison stops and *spent_out is set --
// a single thread must not run for minutes on periodic text (the caller gives up and hands over to the host sorter)
__device__ bool suffix_less_bounded(const u8 *text, i64 n, i64 a, i64 b, i64 skip, i64 &budget, int *spent_out) {
  i64 k = skip;
  while (a + k < n && b + k < n) {
    u8 x = text[a + k], y = text[b + k];
    if (x != y) return x < y;
    ++k;
    if (--budget < 0) { *spent_out = 1; return false; }
  }
  return a + k >= n && b + k < n ? true : (a + k >= n && b + k >= n ? a > b : false);
}

// ---- refinement rounds: groups of suffixes that agree on their first d symbols are re-sorted by the next
// per_key symbols (keys read straight from the text, so nothing depends on ranks of other positions).
// head[k] = 1: sorted position k starts a group.
__global__ __launch_bounds__(PSG_WG) void heads_from_keys_kernel(const u64 *keys, i64 size, u8 *head) {
  i64 k = (i64)blockIdx.x * PSG_WG + threadIdx.x;
  if (k < size) head[k] = (k == 0 || keys[k] != keys[k - 1]) ? 1 : 0;
}
// unresolved = member of a group with more than one element
__global__ __launch_bounds__(PSG_WG) void unresolved_kernel(const u8 *head, i64 size, u8 *flag) {
  i64 k = (i64)blockIdx.x * PSG_WG + threadIdx.x;
  if (k < size) flag[k] = (head[k] && (k + 1 == size || head[k + 1])) ? 0 : 1;
}
// compacted element j (sorted position upos[j]): group start marker (its own index where a group starts, else 0)
__global__ __launch_bounds__(PSG_WG) void group_start_kernel(const u32 *upos, const u8 *head, i64 nu, u32 *start) {
  i64 j = (i64)blockIdx.x * PSG_WG + threadIdx.x;
  if (j < nu) start[j] = head[upos[j]] ? (u32)j : 0u;
}
__global__ __launch_bounds__(PSG_WG) void round_keys_kernel(const u8 *text, i64 n, i64 beg, KeyCfg cfg, i64 depth, const u32 *upos, const u32 *idx,
                                                             const u32 *gid, i64 nu, u64 *key2, u64 *packed) {
  i64 j = (i64)blockIdx.x * PSG_WG + threadIdx.x;
  if (j >= nu) return;
  const u32 s = idx[upos[j]];
  u64 key = 0;
  const i64 p = beg + (i64)s + depth;
  for (int q = 0; q < cfg.per_key; ++q) {
    u64 v = p + q < n ? (u64)cfg.code[text[p + q]] + 1 : 0;
    key = (key << cfg.bits) | v;
  }
  key2[j] = key;
  packed[j] = ((u64)gid[j] << 32) | s;
}
__global__ __launch_bounds__(PSG_WG) void round_scatter_kernel(const u32 *upos, const u64 *packed, const u64 *key2, i64 nu, u32 *idx, u8 *head) {
  i64 j = (i64)blockIdx.x * PSG_WG + threadIdx.x;
  if (j >= nu) return;
  const u32 k = upos[j];
  idx[k] = (u32)packed[j];
  head[k] = (j == 0 || (packed[j] >> 32) != (packed[j - 1] >> 32) || key2[j] != key2[j - 1]) ? 1 : 0;
}

// what is left after the rounds: one thread per group (the group head does the work), insertion sort by suffix comparison
__global__ __launch_bounds__(PSG_WG) void fix_ties_kernel(const u8 *text, i64 n, i64 beg, i64 size, const u8 *head, u32 *idx, i64 skip,
                                                            int max_group, unsigned long long *groups, int *too_big) {
  i64 k = (i64)blockIdx.x * PSG_WG + threadIdx.x;
  if (k >= size) return;
  if (!head[k]) return;                                   // not a group head
  if (k + 1 >= size || head[k + 1]) return;               // singleton
  i64 e = k + 1;
  while (e < size && !head[e]) ++e;
  if (e - k > max_group) { *too_big = 1; return; }
  atomicAdd(groups, 1ull);
  i64 budget = (i64)1 << 22;                               // symbols this thread may compare: ~50 ms; periodic text runs out
  for (i64 a = k + 1; a < e; ++a) {
    u32 v = idx[a];
    i64 b = a - 1;
    while (b >= k && suffix_less_bounded(text, n, beg + v, beg + idx[b], skip, budget, too_big)) { idx[b + 1] = idx[b]; --b; }
    if (budget < 0) return;
    idx[b + 1] = v;
  }
}

__global__ __launch_bounds__(PSG_WG) void find_i0_kernel(const u32 *psa, i64 size, i64 *i0) {
  i64 k = (i64)blockIdx.x * PSG_WG + threadIdx.x;
  if (k < size && psa[k] == 0) *i0 = k;
}

__global__ __launch_bounds__(PSG_WG) void bwt_gt_kernel(const u8 *text, i64 n, i64 beg, i64 size, const u32 *psa, const i64 *i0p, u8 *bwt, u32 *gt, int *too_long) {
  i64 k = (i64)blockIdx.x * PSG_WG + threadIdx.x;
  if (k >= size) return;
  i64 i0 = *i0p;
  u32 s = psa[k];
  bwt[k] = s ? text[beg + s - 1] : 0;                     // dummy 0 at i0 (inmem_bwt_from_sa.hpp:51-54)
  if (gt && s && k > i0) { i64 u = size - s; atomicOr(&gt[u >> 5], 1u << (u & 31)); }
  if (gt && k == 0) {                                     // bit u=0: position j = end
    i64 end = beg + size;
    i64 budget = (i64)1 << 24;                             // a range that repeats itself for longer is left to the host sorter
    bool g = end < n ? suffix_less_bounded(text, n, beg, end, 0, budget, too_long) : false;   // text[end..) > text[beg..) ?
    if (g) atomicOr(&gt[0], 1u);
  }
}

static i64 g_sort_text_begin = 0;   // psgx_sort_halfblock_window: only text[text_begin .. n) is on the device
extern "C" int psgx_sort_halfblock(const uint8_t *d_text, int64_t n, int64_t beg, int64_t end, uint32_t *d_psa, uint8_t *d_bwt,
                                   int64_t *i0, uint32_t *d_gt_begin, int64_t *tie_groups);
// the same for a text of which only the window text[text_begin .. n) is resident (d_text still points at position 0):
// the alphabet is taken from the window, and `n` is the end of the text as the sorter's comparisons see it
extern "C" int psgx_sort_halfblock_window(const uint8_t *d_text, int64_t text_begin, int64_t n, int64_t beg, int64_t end, uint32_t *d_psa,
                                          uint8_t *d_bwt, int64_t *i0, uint32_t *d_gt_begin, int64_t *tie_groups) {
  PSG_REQUIRE(text_begin >= 0 && text_begin <= beg, "psgx_sort_halfblock_window");
  g_sort_text_begin = text_begin;
  const int rc = psgx_sort_halfblock(d_text, n, beg, end, d_psa, d_bwt, i0, d_gt_begin, tie_groups);
  g_sort_text_begin = 0;
  return rc;
}
extern "C" int psgx_sort_halfblock(const uint8_t *d_text, int64_t n, int64_t beg, int64_t end, uint32_t *d_psa, uint8_t *d_bwt,
                                   int64_t *i0, uint32_t *d_gt_begin, int64_t *tie_groups) {
  i64 size = end - beg;
  PSG_REQUIRE(d_text && d_psa && d_bwt && i0 && size >= 1 && size < (1ll << 32) && end <= n && beg >= 0, "psgx_sort_halfblock");
  // alphabet of the WHOLE text -> bits per symbol (a symbol that first appears late must get its own code)
  DevBuf hist;
  int rc;
  if ((rc = hist.alloc(256 * 4))) return rc;
  PSG_HIP(hipMemsetAsync(hist.p, 0, 256 * 4, stream()));
  hipLaunchKernelGGL(present_kernel, dim3((unsigned)std::min<i64>(cdiv(n - g_sort_text_begin, (i64)PSG_WG * 16), 4096)), dim3(PSG_WG), 0, stream(), d_text + g_sort_text_begin,
                     n - g_sort_text_begin, hist.as<u32>());
  PSG_HIP(hipGetLastError());
  u32 hflags[256];
  if (int rc_ = psg::copy_d2h(hflags, hist.p, sizeof hflags)) return rc_;
  PSG_HIP(psg::sync_stream());
  bool present[256];
  for (int c = 0; c < 256; ++c) present[c] = hflags[c] != 0;
  KeyCfg cfg;
  int sigma = 0;
  for (int c = 0; c < 256; ++c) sigma += present[c];
  bool dense = sigma <= 15;      // small alphabets are packed; otherwise raw bytes (code = byte)
  if (dense) { int k = 0; for (int c = 0; c < 256; ++c) cfg.code[c] = present[c] ? (u8)k++ : (u8)0; }
  else for (int c = 0; c < 256; ++c) cfg.code[c] = (u8)c;
  int maxv = dense ? sigma : 255;  // codes+1 must fit
  cfg.bits = 1; while ((1 << cfg.bits) <= maxv) ++cfg.bits;
  if (!dense) cfg.bits = 8;        // bytes 0..254 -> 1..255
  cfg.per_key = 64 / cfg.bits;
  DevBuf keys_a, keys_b, idx_b, tmp, misc;
  if ((rc = keys_a.alloc(size * 8)) || (rc = keys_b.alloc(size * 8)) || (rc = idx_b.alloc(size * 4)) || (rc = misc.alloc(32))) return rc;
  hipLaunchKernelGGL(make_keys_kernel, dim3((unsigned)cdiv(size, PSG_WG)), dim3(PSG_WG), 0, stream(), d_text, n, beg, size, cfg, keys_a.as<u64>(), idx_b.as<u32>());
  PSG_HIP(hipGetLastError());
  size_t tbytes = 0;
  PSG_HIP(rocprim::radix_sort_pairs(nullptr, tbytes, keys_a.as<u64>(), keys_b.as<u64>(), idx_b.as<u32>(), d_psa, (size_t)size, 0, 64, stream()));
  if ((rc = tmp.alloc((i64)tbytes))) return rc;
  PSG_HIP(rocprim::radix_sort_pairs(tmp.p, tbytes, keys_a.as<u64>(), keys_b.as<u64>(), idx_b.as<u32>(), d_psa, (size_t)size, 0, 64, stream()));
  PSG_HIP(hipMemsetAsync(misc.p, 0, 32, stream()));
  // ---- refinement rounds (English-like text leaves most suffixes tied after the first 8-12 symbols)
  DevBuf head;
  if ((rc = head.alloc(size + 16))) return rc;
  hipLaunchKernelGGL(heads_from_keys_kernel, dim3((unsigned)cdiv(size, PSG_WG)), dim3(PSG_WG), 0, stream(), keys_b.as<u64>(), size, head.as<u8>());
  PSG_HIP(hipGetLastError());
  keys_a.alloc(16); keys_b.alloc(16); idx_b.alloc(16); tmp.alloc(16);
  i64 depth = cfg.per_key;
  const int max_rounds = 24;
  for (int round = 0; round < max_rounds; ++round, depth += cfg.per_key) {
    DevBuf flag, upos, cnt;
    if ((rc = flag.alloc(size)) || (rc = upos.alloc(size * 4)) || (rc = cnt.alloc(8))) return rc;
    hipLaunchKernelGGL(unresolved_kernel, dim3((unsigned)cdiv(size, PSG_WG)), dim3(PSG_WG), 0, stream(), head.as<u8>(), size, flag.as<u8>());
    PSG_HIP(hipGetLastError());
    size_t tb2 = 0;
    rocprim::counting_iterator<u32> positions(0);
    PSG_HIP(rocprim::select(nullptr, tb2, positions, flag.as<u8>(), upos.as<u32>(), cnt.as<u64>(), (size_t)size, stream()));
    DevBuf t2;
    if ((rc = t2.alloc((i64)tb2))) return rc;
    PSG_HIP(rocprim::select(t2.p, tb2, positions, flag.as<u8>(), upos.as<u32>(), cnt.as<u64>(), (size_t)size, stream()));
    u64 nu64 = 0;
    if (int rc_ = psg::copy_d2h(&nu64, cnt.p, 8)) return rc_;
    const i64 nu = (i64)nu64;
    flag.alloc(16);
    if (nu == 0) break;
    DevBuf start, gid, key2a, key2b, pka, pkb;
    if ((rc = start.alloc(nu * 4)) || (rc = gid.alloc(nu * 4)) || (rc = key2a.alloc(nu * 8)) || (rc = key2b.alloc(nu * 8)) || (rc = pka.alloc(nu * 8)) || (rc = pkb.alloc(nu * 8))) return rc;
    const unsigned gnu = (unsigned)cdiv(nu, PSG_WG);
    hipLaunchKernelGGL(group_start_kernel, dim3(gnu), dim3(PSG_WG), 0, stream(), upos.as<u32>(), head.as<u8>(), nu, start.as<u32>());
    size_t tb3 = 0;
    PSG_HIP(rocprim::inclusive_scan(nullptr, tb3, start.as<u32>(), gid.as<u32>(), (size_t)nu, rocprim::maximum<u32>(), stream()));
    if ((rc = t2.alloc((i64)tb3))) return rc;
    PSG_HIP(rocprim::inclusive_scan(t2.p, tb3, start.as<u32>(), gid.as<u32>(), (size_t)nu, rocprim::maximum<u32>(), stream()));
    hipLaunchKernelGGL(round_keys_kernel, dim3(gnu), dim3(PSG_WG), 0, stream(), d_text, n, beg, cfg, depth, upos.as<u32>(), d_psa, gid.as<u32>(), nu, key2a.as<u64>(), pka.as<u64>());
    PSG_HIP(hipGetLastError());
    start.alloc(16); gid.alloc(16);
    // stable sort by the new key, then by the group: (group, key) order
    const unsigned kbits = (unsigned)(cfg.per_key * cfg.bits);
    unsigned gbits = 1;
    while (((u64)1 << gbits) < (u64)nu) ++gbits;
    size_t tb4 = 0, tb5 = 0;
    PSG_HIP(rocprim::radix_sort_pairs(nullptr, tb4, key2a.as<u64>(), key2b.as<u64>(), pka.as<u64>(), pkb.as<u64>(), (size_t)nu, 0, kbits, stream()));
    PSG_HIP(rocprim::radix_sort_pairs(nullptr, tb5, pkb.as<u64>(), pka.as<u64>(), key2b.as<u64>(), key2a.as<u64>(), (size_t)nu, 32, 32 + gbits, stream()));
    if ((rc = t2.alloc((i64)std::max(tb4, tb5)))) return rc;
    PSG_HIP(rocprim::radix_sort_pairs(t2.p, tb4, key2a.as<u64>(), key2b.as<u64>(), pka.as<u64>(), pkb.as<u64>(), (size_t)nu, 0, kbits, stream()));
    PSG_HIP(rocprim::radix_sort_pairs(t2.p, tb5, pkb.as<u64>(), pka.as<u64>(), key2b.as<u64>(), key2a.as<u64>(), (size_t)nu, 32, 32 + gbits, stream()));
    hipLaunchKernelGGL(round_scatter_kernel, dim3(gnu), dim3(PSG_WG), 0, stream(), upos.as<u32>(), pka.as<u64>(), key2a.as<u64>(), nu, d_psa, head.as<u8>());
    PSG_HIP(hipGetLastError());
    PSG_HIP(psg::sync_stream());   // the round's buffers go back to the arena
  }
  hipLaunchKernelGGL(fix_ties_kernel, dim3((unsigned)cdiv(size, PSG_WG)), dim3(PSG_WG), 0, stream(), d_text, n, beg, size, head.as<u8>(), d_psa,
                     depth, 4096, misc.as<unsigned long long>(), (int *)((u8 *)misc.p + 8));
  PSG_HIP(hipGetLastError());
  {
    u64 h0[2];
    if (int rc_ = psg::copy_d2h(h0, misc.p, 16)) return rc_;
    if ((int)(h0[1] & 0xFFFFFFFF)) { set_error("psgx_sort_halfblock: an equal-prefix group is too large (text too repetitive for the prefix-key sorter)"); return PSG_ECHECK; }
  }
  hipLaunchKernelGGL(find_i0_kernel, dim3((unsigned)cdiv(size, PSG_WG)), dim3(PSG_WG), 0, stream(), d_psa, size, (i64 *)((u8 *)misc.p + 16));
  if (d_gt_begin) PSG_HIP(hipMemsetAsync(d_gt_begin, 0, (size_t)(((size + 31) >> 5) * 4), stream()));
  hipLaunchKernelGGL(bwt_gt_kernel, dim3((unsigned)cdiv(size, PSG_WG)), dim3(PSG_WG), 0, stream(), d_text, n, beg, size, d_psa, (const i64 *)((u8 *)misc.p + 16), d_bwt, d_gt_begin, (int *)((u8 *)misc.p + 8));
  PSG_HIP(hipGetLastError());
  u64 h[4];
  if (int rc_ = psg::copy_d2h(h, misc.p, (size_t)(32))) return rc_;
  PSG_HIP(psg::sync_stream());
  if ((int)(h[1] & 0xFFFFFFFF)) { set_error("psgx_sort_halfblock: an equal-prefix group is too large (text too repetitive for the prefix-key sorter)"); return PSG_ECHECK; }
  if (tie_groups) *tie_groups = (i64)h[0];
  *i0 = (i64)h[2];
  return 0;
}

// ---------------------------------------------------------------------------------------
// full-size property check of a .sa5 buffer: (a) sum of all entries (mod 2^64) -- equals
// n(n-1)/2 for a permutation of 0..n-1; (b) `samples` random adjacent pairs are in suffix order.
// d_acc: three 64-bit accumulators -- sum, pairs out of order, pairs undecided within the comparison budget.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ u64 load_u40(const u8 *p) {
  return (u64)p[0] | ((u64)p[1] << 8) | ((u64)p[2] << 16) | ((u64)p[3] << 24) | ((u64)p[4] << 32);
}
__global__ __launch_bounds__(PSG_WG) void sa5_sum_kernel(const u8 *sa5, i64 cnt, unsigned long long *sum) {
  __shared__ u64 scratch[8];
  u64 acc = 0;
  for (i64 k = (i64)blockIdx.x * PSG_WG + threadIdx.x; k < cnt; k += (i64)gridDim.x * PSG_WG) acc += load_u40(sa5 + 5 * k);
  u64 tot = block_sum<u64>(acc, scratch);
  if (threadIdx.x == 0) atomicAdd(sum, (unsigned long long)tot);
}
__global__ __launch_bounds__(PSG_WG) void sa5_order_kernel(const u8 *text, i64 n, const u8 *sa5, i64 cnt, i64 samples, u64 seed, unsigned long long *bad) {
  i64 t = (i64)blockIdx.x * PSG_WG + threadIdx.x;
  if (t >= samples || cnt < 2) return;
  i64 k = (i64)(splitmix64(seed + (u64)t) % (u64)(cnt - 1));
  i64 a = (i64)load_u40(sa5 + 5 * k), b = (i64)load_u40(sa5 + 5 * (k + 1));
  if (a >= n || b >= n || a == b) { atomicAdd(bad, 1ull); return; }
  i64 budget = (i64)1 << 24;                               // a sampled pair that agrees on 16 Mi symbols is not followed further (periodic text)
  int spent = 0;
  const bool less = suffix_less_bounded(text, n, a, b, 0, budget, &spent);
  if (spent) atomicAdd(bad + 1, 1ull);                     // undecided within the budget: reported, not counted as in order
  else if (!less) atomicAdd(bad, 1ull);
}
int psg::check_sa5_accumulate(const u8 *d_text, i64 n, const u8 *d_sa5, i64 count, i64 samples, u64 seed, unsigned long long *d_acc) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(sa5_sum_kernel, dim3((unsigned)std::min<i64>(cdiv(count, PSG_WG), 8192)), dim3(PSG_WG), 0, stream(), d_sa5, count, d_acc);
  if (samples > 0) hipLaunchKernelGGL(sa5_order_kernel, dim3((unsigned)cdiv(samples, PSG_WG)), dim3(PSG_WG), 0, stream(), d_text, n, d_sa5, count, samples, seed, d_acc + 1);
  PSG_HIP(hipGetLastError());
  return 0;
}

extern "C" int psgx_check_sa5_ex(const uint8_t *d_text, int64_t n, const uint8_t *d_sa5, int64_t count, int64_t samples, uint64_t seed,
                                 int64_t *bad_pairs, uint64_t *sum, int64_t *undecided_pairs) {
  PSG_REQUIRE(d_text && d_sa5 && bad_pairs && sum && count >= 0, "psgx_check_sa5");
  DevBuf acc;
  if (int rc = acc.alloc(24)) return rc;
  PSG_HIP(hipMemsetAsync(acc.p, 0, 24, stream()));
  if (int rc = psg::check_sa5_accumulate(d_text, n, d_sa5, count, samples, seed, acc.as<unsigned long long>())) return rc;
  u64 h[3];
  if (int rc_ = psg::copy_d2h(h, acc.p, (size_t)(24))) return rc_;
  PSG_HIP(psg::sync_stream());
  *sum = h[0]; *bad_pairs = (i64)h[1];
  if (undecided_pairs) *undecided_pairs = (i64)h[2];
  return 0;
}
extern "C" int psgx_check_sa5(const uint8_t *d_text, int64_t n, const uint8_t *d_sa5, int64_t count, int64_t samples, uint64_t seed,
                              int64_t *bad_pairs, uint64_t *sum) {
  return psgx_check_sa5_ex(d_text, n, d_sa5, count, samples, seed, bad_pairs, sum, nullptr);
}
