// prep.hip -- synthetic-input preparation for bench.py and the full-size property tests.
// NOT part of the drop-in boundary: the reference sorts half-blocks on the host
// (inmem_psascan.hpp:64-304) and so does construct_sa.  To time the hot path at
// BASELINE.json's full sizes the bench needs valid (partial SA, BWT, i0, gt_begin) inputs
// for multi-GiB half-blocks within minutes, so for texts with short repeats (uniform random
// bytes, i.i.d. DNA) they are produced on the device: sort by a packed 64-bit prefix key
// (rocPRIM radix sort) and finish the rare equal-key groups by direct suffix comparison.
#include "dev_common.hpp"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

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

extern "C" int psgx_gen_text(uint8_t *d_text, int64_t n, int mode, int sigma, uint64_t seed) {
  PSG_REQUIRE(d_text && n >= 0 && mode >= 0 && mode <= 2, "psgx_gen_text");
  if (n == 0) return 0;
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

// one thread per equal-key group (the group head does the work): insertion sort by suffix comparison
__global__ __launch_bounds__(PSG_WG) void fix_ties_kernel(const u8 *text, i64 n, i64 beg, i64 size, const u64 *keys, u32 *idx, int skip,
                                                            int max_group, unsigned long long *groups, int *too_big) {
  i64 k = (i64)blockIdx.x * PSG_WG + threadIdx.x;
  if (k >= size) return;
  if (k > 0 && keys[k - 1] == keys[k]) return;           // not a group head
  if (k + 1 >= size || keys[k + 1] != keys[k]) return;   // singleton
  i64 e = k + 1;
  while (e < size && keys[e] == keys[k]) ++e;
  if (e - k > max_group) { *too_big = 1; return; }
  atomicAdd(groups, 1ull);
  for (i64 a = k + 1; a < e; ++a) {
    u32 v = idx[a];
    i64 b = a - 1;
    while (b >= k && suffix_less(text, n, beg + v, beg + idx[b], skip)) { idx[b + 1] = idx[b]; --b; }
    idx[b + 1] = v;
  }
}

__global__ __launch_bounds__(PSG_WG) void find_i0_kernel(const u32 *psa, i64 size, i64 *i0) {
  i64 k = (i64)blockIdx.x * PSG_WG + threadIdx.x;
  if (k < size && psa[k] == 0) *i0 = k;
}

__global__ __launch_bounds__(PSG_WG) void bwt_gt_kernel(const u8 *text, i64 n, i64 beg, i64 size, const u32 *psa, const i64 *i0p, u8 *bwt, u32 *gt) {
  i64 k = (i64)blockIdx.x * PSG_WG + threadIdx.x;
  if (k >= size) return;
  i64 i0 = *i0p;
  u32 s = psa[k];
  bwt[k] = s ? text[beg + s - 1] : 0;                     // dummy 0 at i0 (inmem_bwt_from_sa.hpp:51-54)
  if (gt && s && k > i0) { i64 u = size - s; atomicOr(&gt[u >> 5], 1u << (u & 31)); }
  if (gt && k == 0) {                                     // bit u=0: position j = end
    i64 end = beg + size;
    bool g = end < n ? suffix_less(text, n, beg, end, 0) : false;   // text[end..) > text[beg..) ?
    if (g) atomicOr(&gt[0], 1u);
  }
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
  hipLaunchKernelGGL(present_kernel, dim3((unsigned)std::min<i64>(cdiv(n, (i64)PSG_WG * 16), 4096)), dim3(PSG_WG), 0, stream(), d_text, n, hist.as<u32>());
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
  hipLaunchKernelGGL(fix_ties_kernel, dim3((unsigned)cdiv(size, PSG_WG)), dim3(PSG_WG), 0, stream(), d_text, n, beg, size, keys_b.as<u64>(), d_psa,
                     cfg.per_key, 4096, misc.as<unsigned long long>(), (int *)((u8 *)misc.p + 8));
  PSG_HIP(hipGetLastError());
  hipLaunchKernelGGL(find_i0_kernel, dim3((unsigned)cdiv(size, PSG_WG)), dim3(PSG_WG), 0, stream(), d_psa, size, (i64 *)((u8 *)misc.p + 16));
  if (d_gt_begin) PSG_HIP(hipMemsetAsync(d_gt_begin, 0, (size_t)(((size + 31) >> 5) * 4), stream()));
  hipLaunchKernelGGL(bwt_gt_kernel, dim3((unsigned)cdiv(size, PSG_WG)), dim3(PSG_WG), 0, stream(), d_text, n, beg, size, d_psa, (const i64 *)((u8 *)misc.p + 16), d_bwt, d_gt_begin);
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
  if (a >= n || b >= n || a == b || !suffix_less(text, n, a, b, 0)) atomicAdd(bad, 1ull);
}
int psg::check_sa5_accumulate(const u8 *d_text, i64 n, const u8 *d_sa5, i64 count, i64 samples, u64 seed, unsigned long long *d_acc) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(sa5_sum_kernel, dim3((unsigned)std::min<i64>(cdiv(count, PSG_WG), 8192)), dim3(PSG_WG), 0, stream(), d_sa5, count, d_acc);
  if (samples > 0) hipLaunchKernelGGL(sa5_order_kernel, dim3((unsigned)cdiv(samples, PSG_WG)), dim3(PSG_WG), 0, stream(), d_text, n, d_sa5, count, samples, seed, d_acc + 1);
  PSG_HIP(hipGetLastError());
  return 0;
}

extern "C" int psgx_check_sa5(const uint8_t *d_text, int64_t n, const uint8_t *d_sa5, int64_t count, int64_t samples, uint64_t seed,
                              int64_t *bad_pairs, uint64_t *sum) {
  PSG_REQUIRE(d_text && d_sa5 && bad_pairs && sum && count >= 0, "psgx_check_sa5");
  DevBuf acc;
  if (int rc = acc.alloc(16)) return rc;
  PSG_HIP(hipMemsetAsync(acc.p, 0, 16, stream()));
  if (int rc = psg::check_sa5_accumulate(d_text, n, d_sa5, count, samples, seed, acc.as<unsigned long long>())) return rc;
  u64 h[2];
  if (int rc_ = psg::copy_d2h(h, acc.p, (size_t)(16))) return rc_;
  PSG_HIP(psg::sync_stream());
  *sum = h[0]; *bad_pairs = (i64)h[1];
  return 0;
}
