// bits_merge.hip -- K3 gap->bitvector, K4 BWT merge, K5 gap split (unary / "merge
// bitvector" form), K6 gap values + vbyte, K7 final merge.  All are tile-scan + scatter:
//   tile reduce -> single-workgroup scan of the tile sums -> tile apply.
// Reference loops restated (semantics only): gap_array.hpp:273-364, bwt_merge.hpp:66-140,
// compute_right_gap.hpp:55-122, compute_left_gap.hpp:55-123, utils/parallel_utils.hpp:47-136,
// merge.hpp:110-159.
#include "dev_common.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <thread>
#include <vector>

using namespace psg;

#define TILE_V 2048   // values per tile (8 per thread)
#define TILE_B 4096   // bits / outputs per tile (16 per thread)

// =======================================================================================
// small kernels shared by several entry points
// =======================================================================================
// g[q] = v[base + q] for base + q < n (else 0): two 16-byte loads when the group is whole and v is 16-byte aligned
__device__ __forceinline__ void load8_u32(const u32 *v, i64 base, i64 n, u32 (&g)[8]) {
  if (base + 8 <= n && ((uintptr_t)v & 15) == 0) {
    const uint4 a = *(const uint4 *)(v + base), b = *(const uint4 *)(v + base + 4);
    g[0] = a.x; g[1] = a.y; g[2] = a.z; g[3] = a.w; g[4] = b.x; g[5] = b.y; g[6] = b.z; g[7] = b.w;
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) g[q] = base + q < n ? v[base + q] : 0u;
  }
}

// the 8 gap VALUES of slots slot0 + base + q (counters v[base + q] + their excess entries); 0 beyond n
__device__ __forceinline__ void load8_gap(const u32 *v, i64 base, i64 n, const ExcessView &X, i64 slot0, u64 (&g)[8]) {
  u32 c[8];
  load8_u32(v, base, n, c);
#pragma unroll
  for (int q = 0; q < 8; ++q) g[q] = c[q];
  excess_apply8(X, slot0 + base, g);
}

__global__ __launch_bounds__(PSG_WG) void tile_sum_u32_kernel(const u32 *v, i64 n, u64 *tile_sum, ExcessView X, i64 slot0) {
  __shared__ u64 scratch[8];
  i64 base = (i64)blockIdx.x * TILE_V + (i64)threadIdx.x * 8;
  u64 g[8], s = 0;
  load8_gap(v, base, n, X, slot0, g);
#pragma unroll
  for (int q = 0; q < 8; ++q) s += g[q];
  u64 tot = block_sum<u64>(s, scratch);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}

// ones (or zeros) per TILE_B-bit tile of a bit array of nbits bits
template <bool ZEROS>
__global__ __launch_bounds__(PSG_WG) void tile_popc_kernel(const u32 *bv, i64 nbits, u64 *tile_cnt) {
  __shared__ u64 scratch[8];
  i64 nwords = (nbits + 31) >> 5;
  i64 b0 = (i64)blockIdx.x * TILE_B + (i64)threadIdx.x * 16;
  int n = (int)std::max<i64>(0, std::min<i64>(16, nbits - b0));
  u32 bits = n > 0 ? get_bits(bv, b0, n, nwords) : 0;
  u64 c = ZEROS ? (u64)(n - __popc(bits)) : (u64)__popc(bits);
  u64 tot = block_sum<u64>(c, scratch);
  if (threadIdx.x == 0) tile_cnt[blockIdx.x] = tot;
}

__global__ void mask_tail_kernel(u32 *bv, i64 nbits) {
  if (nbits & 31) bv[nbits >> 5] &= (1u << (nbits & 31)) - 1u;
}

static int fill_ones(u32 *d_bv, i64 nbits) {
  if (nbits <= 0) return 0;
  PSG_HIP(hipMemsetAsync(d_bv, 0xFF, (size_t)(((nbits + 31) >> 5) * 4), stream()));
  hipLaunchKernelGGL(mask_tail_kernel, dim3(1), dim3(1), 0, stream(), d_bv, nbits);
  PSG_HIP(hipGetLastError());
  return 0;
}

// explicit address spaces: when an LDS and a global atomic sit in the two arms of a branch the
// compiler merges them into ONE flat atomic on a selected pointer, which is slower for both
__device__ __forceinline__ void global_and(u32 *p, u32 x) {
  (void)__hip_atomic_fetch_and((__attribute__((address_space(1))) u32 *)p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lds_and(u32 *p, u32 x) {
  (void)__hip_atomic_fetch_and((__attribute__((address_space(3))) u32 *)p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void clear_bit(u32 *bv, i64 pos) { global_and(&bv[pos >> 5], ~(1u << (pos & 31))); }

// =======================================================================================
// K3: gap -> bitvector
// =======================================================================================
// Zero bits of one tile are (almost always) confined to a few hundred consecutive words: they are
// collected in an LDS bitmap window and merged into the ones-filled output with one atomicAnd per
// touched WORD (coalesced) instead of one per bit; bits beyond the window fall back to clear_bit.
#define BMW 512   // words in the LDS bitmap window (16384 bits)

struct BitWindow {
  u32 w[BMW];
  i64 base_word;
};

__device__ __forceinline__ void bw_init(BitWindow &W, i64 first_pos) {
  for (int k = threadIdx.x; k < BMW; k += PSG_WG) W.w[k] = 0xFFFFFFFFu;
  if (threadIdx.x == 0) W.base_word = first_pos >> 5;
}
__device__ __forceinline__ void bw_clear(BitWindow &W, u32 *bv, i64 pos) {
  i64 k = (pos >> 5) - W.base_word;
  if (k >= 0 && k < BMW) lds_and(&W.w[k], ~(1u << (pos & 31)));
  else clear_bit(bv, pos);
}
__device__ __forceinline__ void bw_flush(BitWindow &W, u32 *bv) {
  for (int k = threadIdx.x; k < BMW; k += PSG_WG) {
    u32 x = W.w[k];
    if (x != 0xFFFFFFFFu) global_and(&bv[W.base_word + k], x);
  }
}

__global__ __launch_bounds__(PSG_WG) void gap_to_bv_apply_kernel(const u32 *gap, i64 m, const u64 *tile_pref, u32 *bv, ExcessView X) {
  __shared__ u64 scratch[8];
  __shared__ BitWindow W;
  __shared__ u64 first_gap;
  i64 base = (i64)blockIdx.x * TILE_V + (i64)threadIdx.x * 8;
  u64 g[8];
  u64 s = 0;
  load8_gap(gap, base, m + 1, X, 0, g);
#pragma unroll
  for (int q = 0; q < 8; ++q) s += g[q];
  if (threadIdx.x == 0) first_gap = g[0];
  u64 tot;
  u64 tp = tile_pref[blockIdx.x];
  u64 pre = tp + block_excl_scan<u64>(s, scratch, tot);   // (contains barriers: first_gap is visible)
  // first zero position of the tile: element j0 = blockIdx*TILE_V sits at j0 + tp + gap[j0]
  bw_init(W, (i64)blockIdx.x * TILE_V + (i64)tp + (i64)first_gap);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    pre += g[q];
    i64 j = base + q;
    if (j < m) bw_clear(W, bv, j + (i64)pre);
  }
  __syncthreads();
  bw_flush(W, bv);
}

extern "C" int psg_gap_to_bitvector(const uint32_t *d_gap, int64_t m, uint32_t *d_bv, int64_t cap_bits, int64_t *nbits) {
  PSG_REQUIRE(d_gap && d_bv && m >= 0 && nbits, "psg_gap_to_bitvector");
  static const bool dbg = getenv("PSG_TIMING") != nullptr;
  auto now = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; };
  if (dbg) { double d0 = now(); (void)psg::sync_stream(); fprintf(stderr, "[psg_gap_to_bitvector]   drain at entry %.2f ms\n", now() - d0); }
  double w0 = now();
  EventTimer tm; tm.start();
  i64 n = m + 1, ntiles = cdiv(n, TILE_V);
  DevBuf ts, tot;
  int rc;
  if ((rc = ts.alloc(ntiles * 8)) || (rc = tot.alloc(8))) return rc;
  double w1 = now();
  ExcessView X;
  struct Owned { void *p = nullptr; ~Owned() { if (p) psg::pool_free(p); } } owned;
  if ((rc = gap_excess_view(d_gap, m, &X, &owned.p))) return rc;
  hipLaunchKernelGGL(tile_sum_u32_kernel, dim3((unsigned)ntiles), dim3(PSG_WG), 0, stream(), d_gap, n, ts.as<u64>(), X, (i64)0);
  PSG_HIP(hipGetLastError());
  double x1 = now();
  if ((rc = scan_u64_inplace(ts.as<u64>(), ntiles, tot.as<u64>()))) return rc;
  double x2 = now();
  u64 total = 0;
  PSG_HIP(hipMemcpyAsync(pinned_buf(3, 64), tot.p, 8, hipMemcpyDeviceToHost, stream()));   // pinned: pageable read-backs stall
  double x3 = now();
  PSG_HIP(psg::sync_stream());
  if (dbg) fprintf(stderr, "[psg_gap_to_bitvector]   launch sum %.2f  launch scan %.2f  memcpyAsync %.2f  sync %.2f ms\n", x1 - w1, x2 - x1, x3 - x2, now() - x3);
  memcpy(&total, pinned_buf(3, 64), 8);
  double w2 = now();
  *nbits = m + (i64)total;
  if (*nbits > cap_bits) { set_error("psg_gap_to_bitvector: output capacity too small"); return PSG_EINVAL; }
  if ((rc = fill_ones(d_bv, *nbits))) return rc;
  double w3 = now();
  hipLaunchKernelGGL(gap_to_bv_apply_kernel, dim3((unsigned)ntiles), dim3(PSG_WG), 0, stream(), d_gap, m, ts.as<u64>(), d_bv, X);
  PSG_HIP(hipGetLastError());
  tm.stop();
  PSG_HIP(psg::sync_stream());
  note_kernel_ms(tm.ms());
  if (dbg) fprintf(stderr, "[psg_gap_to_bitvector] alloc %.2f  sum+scan+readback %.2f  fill_ones(host) %.2f  apply+sync %.2f ms\n", w1 - w0, w2 - w1, w3 - w2, now() - w3);
  return 0;
}

// ---- slice form for the multi-GPU driver: rank d owns gap[j0 .. j0+count) and knows how many
// tail suffixes fall into lower slices (ps_before).  It marks its own zero-bit positions
// j + ps_before + sum_{t<=j, t in slice} gap[t] with ONE bits in a zero-initialised array of the
// global size; the arrays of all ranks are summed (disjoint bits -> OR) and inverted.
__global__ __launch_bounds__(PSG_WG) void gap_slice_bits_kernel(const u32 *gap, i64 j0, i64 count, i64 m, u64 ps_before,
                                                                  const u64 *tile_pref, u32 *bits) {
  __shared__ u64 scratch[8];
  __shared__ u32 win[BMW];
  __shared__ i64 base_word;
  i64 base = (i64)blockIdx.x * TILE_V + (i64)threadIdx.x * 8;
  u32 g[8];
  u64 s = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) { g[q] = base + q < count ? gap[base + q] : 0; s += g[q]; }
  u64 tot;
  u64 tp = ps_before + tile_pref[blockIdx.x];
  u64 pre = tp + block_excl_scan<u64>(s, scratch, tot);
  for (int k = threadIdx.x; k < BMW; k += PSG_WG) win[k] = 0;
  if (threadIdx.x == 0) base_word = (j0 + (i64)blockIdx.x * TILE_V + (i64)tp) >> 5;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    pre += g[q];
    i64 j = j0 + base + q;
    if (base + q < count && j < m) {
      i64 pos = j + (i64)pre, k = (pos >> 5) - base_word;
      if (k >= 0 && k < BMW) atomicOr(&win[k], 1u << (pos & 31));
      else atomicOr(&bits[pos >> 5], 1u << (pos & 31));
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < BMW; k += PSG_WG) if (win[k]) atomicOr(&bits[base_word + k], win[k]);
}

extern "C" int psg_gap_slice_to_bits(const uint32_t *d_gap_slice, int64_t j0, int64_t count, int64_t m, uint64_t ps_before,
                                     uint32_t *d_bits) {
  PSG_REQUIRE(d_bits && j0 >= 0 && count >= 0 && m >= 0, "psg_gap_slice_to_bits");
  if (count == 0) return 0;
  PSG_REQUIRE(d_gap_slice, "psg_gap_slice_to_bits: slice required");
  i64 ntiles = cdiv(count, TILE_V);
  DevBuf ts;
  int rc;
  if ((rc = ts.alloc(ntiles * 8))) return rc;
  hipLaunchKernelGGL(tile_sum_u32_kernel, dim3((unsigned)ntiles), dim3(PSG_WG), 0, stream(), d_gap_slice, count, ts.as<u64>(), ExcessView{nullptr, 0, 32}, (i64)0);
  PSG_HIP(hipGetLastError());
  if ((rc = scan_u64_inplace(ts.as<u64>(), ntiles, nullptr))) return rc;
  hipLaunchKernelGGL(gap_slice_bits_kernel, dim3((unsigned)ntiles), dim3(PSG_WG), 0, stream(), d_gap_slice, j0, count, m, (u64)ps_before,
                     ts.as<u64>(), d_bits);
  PSG_HIP(hipGetLastError());
  PSG_HIP(psg::sync_stream());
  return 0;
}

__global__ __launch_bounds__(PSG_WG) void bits_not_kernel(u32 *bits, i64 nbits) {
  i64 w = (i64)blockIdx.x * PSG_WG + threadIdx.x, nw = (nbits + 31) >> 5;
  if (w >= nw) return;
  u32 x = ~bits[w];
  if (w == nw - 1 && (nbits & 31)) x &= (1u << (nbits & 31)) - 1u;
  bits[w] = x;
}
extern "C" int psg_bits_not(uint32_t *d_bits, int64_t nbits) {
  PSG_REQUIRE(d_bits && nbits >= 0, "psg_bits_not");
  if (nbits == 0) return 0;
  hipLaunchKernelGGL(bits_not_kernel, dim3((unsigned)cdiv((nbits + 31) >> 5, PSG_WG)), dim3(PSG_WG), 0, stream(), d_bits, nbits);
  PSG_HIP(hipGetLastError());
  return 0;
}

// =======================================================================================
// K4: BWT merge
// =======================================================================================
__global__ __launch_bounds__(PSG_WG) void merge_bwt_apply_kernel(const u8 *L, const u8 *R, i64 ml, i64 mr, i64 left_i0, i64 right_i0,
                                                                   u32 left_last, const u32 *bv, const u64 *tile_pref, u8 *out,
                                                                   i64 *block_i0) {
  __shared__ u32 scratch[8];
  __shared__ __attribute__((aligned(4))) u8 sL[TILE_B + 8], sR[TILE_B + 8];   // + the bytes in front of an unaligned start
  __shared__ __attribute__((aligned(4))) u8 sO[TILE_B];
  i64 block = ml + mr, nwords = (block + 31) >> 5;
  i64 k0 = (i64)blockIdx.x * TILE_B;
  int nvalid = (int)std::min<i64>(TILE_B, block - k0);
  int e0 = threadIdx.x * 16;
  int n = std::max(0, std::min(16, nvalid - e0));
  u32 bits = n > 0 ? get_bits(bv, k0 + e0, n, nwords) : 0;
  u32 nr;
  u32 o = block_excl_scan<u32>((u32)__popc(bits), scratch, nr);
  i64 r0 = (i64)tile_pref[blockIdx.x], l0 = k0 - r0;
  int nl = nvalid - (int)nr;
  // both source ranges are staged with aligned 4-byte loads (a byte per lane is a quarter of the bandwidth):
  // the staged copy starts at the 4-byte boundary below the range, offL / offR bytes in front of it
  const int offL = (int)((uintptr_t)(L + l0) & 3), offR = (int)((uintptr_t)(R + r0) & 3);
  {
    const u32 *pL = (const u32 *)(L + l0 - offL), *pR = (const u32 *)(R + r0 - offR);
    for (int k = threadIdx.x; k * 4 < nl + offL; k += PSG_WG) ((u32 *)sL)[k] = pL[k];
    for (int k = threadIdx.x; k * 4 < (int)nr + offR; k += PSG_WG) ((u32 *)sR)[k] = pR[k];
  }
  __syncthreads();
  int z = e0 - (int)o;
  for (int q = 0; q < n; ++q) {
    u32 bit = (bits >> q) & 1u;
    int below = __popc(bits & ((1u << q) - 1u));
    u8 v;
    if (bit) {
      int ri = (int)o + below;
      v = (r0 + ri == right_i0) ? (u8)left_last : sR[offR + ri];       // bwt_merge.hpp:128
    } else {
      int li = z + q - below;
      v = sL[offL + li];
      if (l0 + li == left_i0) *block_i0 = k0 + e0 + q;           // bwt_merge.hpp:133
    }
    sO[e0 + q] = v;
  }
  __syncthreads();
  for (int k = threadIdx.x * 4; k < nvalid; k += PSG_WG * 4) {
    if (k + 4 <= nvalid) *(u32 *)(out + k0 + k) = *(const u32 *)(sO + k);
    else for (int q = k; q < nvalid; ++q) out[k0 + q] = sO[q];
  }
}

extern "C" int psg_merge_bwt(const uint8_t *d_l, const uint8_t *d_r, int64_t ml, int64_t mr, int64_t left_i0, int64_t right_i0,
                             int left_last, const uint32_t *d_bv, uint8_t *d_out, int64_t *block_i0) {
  PSG_REQUIRE(d_l && d_r && d_bv && d_out && block_i0 && ml >= 1 && mr >= 1, "psg_merge_bwt");
  PSG_REQUIRE(left_i0 >= 0 && left_i0 < ml && right_i0 >= 0 && right_i0 < mr, "psg_merge_bwt: i0 out of range");
  PSG_REQUIRE(((uintptr_t)d_out & 3) == 0, "psg_merge_bwt: output must be 4-byte aligned");
  EventTimer tm; tm.start();
  i64 block = ml + mr, ntiles = cdiv(block, TILE_B);
  DevBuf tc, tot, bi0;
  int rc;
  if ((rc = tc.alloc(ntiles * 8)) || (rc = tot.alloc(8)) || (rc = bi0.alloc(8))) return rc;
  hipLaunchKernelGGL((tile_popc_kernel<false>), dim3((unsigned)ntiles), dim3(PSG_WG), 0, stream(), d_bv, block, tc.as<u64>());
  PSG_HIP(hipGetLastError());
  if ((rc = scan_u64_inplace(tc.as<u64>(), ntiles, tot.as<u64>()))) return rc;
  u64 ones = 0;
  PSG_HIP(hipMemcpyAsync(pinned_buf(3, 64), tot.p, 8, hipMemcpyDeviceToHost, stream()));   // pinned: pageable read-backs stall
  PSG_HIP(psg::sync_stream());
  memcpy(&ones, pinned_buf(3, 64), 8);
  if ((i64)ones != mr) { set_error("psg_merge_bwt: bitvector has " + std::to_string(ones) + " ones, expected " + std::to_string(mr)); return PSG_ECHECK; }
  PSG_HIP(hipMemsetAsync(bi0.p, 0xFF, 8, stream()));
  hipLaunchKernelGGL(merge_bwt_apply_kernel, dim3((unsigned)ntiles), dim3(PSG_WG), 0, stream(), d_l, d_r, ml, mr, left_i0, right_i0,
                     (u32)left_last, d_bv, tc.as<u64>(), d_out, bi0.as<i64>());
  PSG_HIP(hipGetLastError());
  tm.stop();
  if (int rc_ = psg::copy_d2h(block_i0, bi0.p, (size_t)(8))) return rc_;
  PSG_HIP(psg::sync_stream());
  note_kernel_ms(tm.ms());
  if (*block_i0 < 0) { set_error("psg_merge_bwt: block_i0 not found"); return PSG_ECHECK; }
  return 0;
}

// =======================================================================================
// K5: split the block gap into the merge bitvectors of the two half-blocks.
//   PS[k] = sum_{t<=k} block_gap[t];  r1(k) = #ones of bv before k
//   bv[k]==0 (left suffix)  -> mbv_left  has a zero at k + PS[k]
//   bv[k]==1 (right suffix) -> mbv_right has a zero at r1(k) + PS[k]
// (closed form of the segment sums of compute_left_gap.hpp:55-123 /
//  compute_right_gap.hpp:55-122)
// =======================================================================================
__global__ __launch_bounds__(PSG_WG) void split_reduce_kernel(const u32 *gap, const u32 *bv, i64 block, u64 *tile_g, u64 *tile_o, ExcessView X) {
  __shared__ u64 scratch[8];
  i64 base = (i64)blockIdx.x * TILE_V + (i64)threadIdx.x * 8;
  u64 s = 0;
  u64 g[8];
  load8_gap(gap, base, block + 1, X, 0, g);
#pragma unroll
  for (int q = 0; q < 8; ++q) s += g[q];
  int n = (int)std::max<i64>(0, std::min<i64>(8, block - base));
  u32 bits = n > 0 ? get_bits(bv, base, n, (block + 31) >> 5) : 0;
  u64 tg = block_sum<u64>(s, scratch);
  u64 to = block_sum<u64>((u64)__popc(bits), scratch);
  if (threadIdx.x == 0) { tile_g[blockIdx.x] = tg; tile_o[blockIdx.x] = to; }
}

__global__ __launch_bounds__(PSG_WG) void split_apply_kernel(const u32 *gap, const u32 *bv, i64 block, const u64 *tile_g,
                                                               const u64 *tile_o, u32 *mbv_left, u32 *mbv_right, ExcessView X) {
  __shared__ u64 scratch[8];
  __shared__ BitWindow WL, WR;
  i64 base = (i64)blockIdx.x * TILE_V + (i64)threadIdx.x * 8;
  u64 g[8];
  u64 s = 0;
  load8_gap(gap, base, block + 1, X, 0, g);
#pragma unroll
  for (int q = 0; q < 8; ++q) s += g[q];
  int n = (int)std::max<i64>(0, std::min<i64>(8, block - base));
  u32 bits = n > 0 ? get_bits(bv, base, n, (block + 31) >> 5) : 0;
  u64 t0, t1;
  u64 tg = tile_g[blockIdx.x], to = tile_o[blockIdx.x];
  u64 ps = tg + block_excl_scan<u64>(s, scratch, t0);
  u64 r1 = to + block_excl_scan<u64>((u64)__popc(bits), scratch, t1);
  // windows start at the tile's first possible positions: k0 + PS (left), r1 + PS (right)
  i64 k0 = (i64)blockIdx.x * TILE_V;
  bw_init(WL, k0 + (i64)tg);
  bw_init(WR, (i64)to + (i64)tg);
  __syncthreads();
  for (int q = 0; q < n; ++q) {
    ps += g[q];
    i64 k = base + q;
    if ((bits >> q) & 1u) { bw_clear(WR, mbv_right, (i64)(r1 + ps)); ++r1; }
    else bw_clear(WL, mbv_left, k + (i64)ps);
  }
  __syncthreads();
  bw_flush(WL, mbv_left);
  bw_flush(WR, mbv_right);
}

extern "C" int psg_split_gap(const uint32_t *d_gap, const uint32_t *d_bv, int64_t ml, int64_t mr, int64_t tail_len,
                             uint32_t *d_mbv_left, uint32_t *d_mbv_right) {
  PSG_REQUIRE(d_gap && d_bv && d_mbv_left && d_mbv_right && ml >= 1 && mr >= 1 && tail_len >= 0, "psg_split_gap");
  EventTimer tm; tm.start();
  i64 block = ml + mr, ntiles = cdiv(block + 1, TILE_V);
  DevBuf tg, to, tot;
  int rc;
  if ((rc = tg.alloc(ntiles * 8)) || (rc = to.alloc(ntiles * 8)) || (rc = tot.alloc(16))) return rc;
  ExcessView X;
  struct Owned { void *p = nullptr; ~Owned() { if (p) psg::pool_free(p); } } owned;
  if ((rc = gap_excess_view(d_gap, block, &X, &owned.p))) return rc;
  hipLaunchKernelGGL(split_reduce_kernel, dim3((unsigned)ntiles), dim3(PSG_WG), 0, stream(), d_gap, d_bv, block, tg.as<u64>(), to.as<u64>(), X);
  PSG_HIP(hipGetLastError());
  if ((rc = scan_u64_inplace(tg.as<u64>(), ntiles, tot.as<u64>()))) return rc;
  if ((rc = scan_u64_inplace(to.as<u64>(), ntiles, tot.as<u64>() + 1))) return rc;
  u64 t[2];
  if (int rc_ = psg::copy_d2h(t, tot.p, (size_t)(16))) return rc_;
  PSG_HIP(psg::sync_stream());
  if ((i64)t[0] != tail_len) { set_error("psg_split_gap: sum(block_gap)=" + std::to_string(t[0]) + " != tail_len=" + std::to_string(tail_len)); return PSG_ECHECK; }
  if ((i64)t[1] != mr) { set_error("psg_split_gap: bitvector ones=" + std::to_string(t[1]) + " != right size " + std::to_string(mr)); return PSG_ECHECK; }
  if ((rc = fill_ones(d_mbv_left, block + tail_len)) || (rc = fill_ones(d_mbv_right, mr + tail_len))) return rc;
  hipLaunchKernelGGL(split_apply_kernel, dim3((unsigned)ntiles), dim3(PSG_WG), 0, stream(), d_gap, d_bv, block, tg.as<u64>(), to.as<u64>(),
                     d_mbv_left, d_mbv_right, X);
  PSG_HIP(hipGetLastError());
  tm.stop();
  PSG_HIP(psg::sync_stream());
  note_kernel_ms(tm.ms());
  return 0;
}

// =======================================================================================
// K6: merge bitvector -> gap values; vbyte
// =======================================================================================
__global__ __launch_bounds__(PSG_WG) void zpos_kernel(const u32 *mbv, i64 nbits, const u64 *tile_pref, u64 *zpos) {
  __shared__ u32 scratch[8];
  i64 b0 = (i64)blockIdx.x * TILE_B + (i64)threadIdx.x * 16;
  int n = (int)std::max<i64>(0, std::min<i64>(16, nbits - b0));
  u32 bits = n > 0 ? get_bits(mbv, b0, n, (nbits + 31) >> 5) : 0;
  u32 tot;
  u32 zr = block_excl_scan<u32>((u32)(n - __popc(bits)), scratch, tot);
  u64 r = tile_pref[blockIdx.x] + zr;
  for (int q = 0; q < n; ++q) if (!((bits >> q) & 1u)) zpos[r++] = (u64)(b0 + q);
}
__global__ __launch_bounds__(PSG_WG) void zpos_to_gap_kernel(const u64 *zpos, i64 size, i64 nbits, u64 *gap) {
  i64 r = (i64)blockIdx.x * PSG_WG + threadIdx.x;
  if (r > size) return;
  u64 prev_end = r ? zpos[r - 1] + 1 : 0;
  gap[r] = (r < size ? zpos[r] : (u64)nbits) - prev_end;
}

extern "C" int psg_mbv_to_gap(const uint32_t *d_mbv, int64_t nbits, int64_t size, uint64_t *d_gap_out) {
  PSG_REQUIRE(d_mbv && d_gap_out && nbits >= size && size >= 0, "psg_mbv_to_gap");
  i64 ntiles = std::max<i64>(1, cdiv(nbits, TILE_B));
  DevBuf tc, tot, zp;
  int rc;
  if ((rc = tc.alloc(ntiles * 8)) || (rc = tot.alloc(8)) || (rc = zp.alloc(size * 8))) return rc;
  hipLaunchKernelGGL((tile_popc_kernel<true>), dim3((unsigned)ntiles), dim3(PSG_WG), 0, stream(), d_mbv, nbits, tc.as<u64>());
  PSG_HIP(hipGetLastError());
  if ((rc = scan_u64_inplace(tc.as<u64>(), ntiles, tot.as<u64>()))) return rc;
  u64 zeros = 0;
  PSG_HIP(hipMemcpyAsync(pinned_buf(3, 64), tot.p, 8, hipMemcpyDeviceToHost, stream()));   // pinned: pageable read-backs stall
  PSG_HIP(psg::sync_stream());
  memcpy(&zeros, pinned_buf(3, 64), 8);
  if ((i64)zeros != size) { set_error("psg_mbv_to_gap: " + std::to_string(zeros) + " zero bits, expected " + std::to_string(size)); return PSG_ECHECK; }
  hipLaunchKernelGGL(zpos_kernel, dim3((unsigned)ntiles), dim3(PSG_WG), 0, stream(), d_mbv, nbits, tc.as<u64>(), zp.as<u64>());
  hipLaunchKernelGGL(zpos_to_gap_kernel, dim3((unsigned)cdiv(size + 1, PSG_WG)), dim3(PSG_WG), 0, stream(), zp.as<u64>(), size, nbits, d_gap_out);
  PSG_HIP(hipGetLastError());
  PSG_HIP(psg::sync_stream());
  return 0;
}

__device__ __forceinline__ int vbyte_len(u64 x) { int l = 1; while (x > 127) { x >>= 7; ++l; } return l; }

__global__ __launch_bounds__(PSG_WG) void vbyte_len_kernel(const u64 *v, i64 n, u64 *tile_sum) {
  __shared__ u64 scratch[8];
  i64 base = (i64)blockIdx.x * TILE_V + (i64)threadIdx.x * 8;
  u64 s = 0;
  for (int q = 0; q < 8; ++q) if (base + q < n) s += vbyte_len(v[base + q]);
  u64 tot = block_sum<u64>(s, scratch);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}
__global__ __launch_bounds__(PSG_WG) void vbyte_write_kernel(const u64 *v, i64 n, const u64 *tile_pref, u8 *out, i64 cap) {
  __shared__ u64 scratch[8];
  i64 base = (i64)blockIdx.x * TILE_V + (i64)threadIdx.x * 8;
  u64 s = 0;
  for (int q = 0; q < 8; ++q) if (base + q < n) s += vbyte_len(v[base + q]);
  u64 tot;
  u64 p = tile_pref[blockIdx.x] + block_excl_scan<u64>(s, scratch, tot);
  for (int q = 0; q < 8; ++q) {
    if (base + q >= n) break;
    u64 x = v[base + q];
    while (x > 127) { if ((i64)p < cap) out[p] = (u8)((x & 0x7f) | 0x80); ++p; x >>= 7; }
    if ((i64)p < cap) out[p] = (u8)x;
    ++p;
  }
}

extern "C" int psg_vbyte_encode(const uint64_t *d_vals, int64_t count, uint8_t *d_out, int64_t cap, int64_t *nbytes) {
  PSG_REQUIRE(d_vals && d_out && nbytes && count >= 0, "psg_vbyte_encode");
  *nbytes = 0;
  if (count == 0) return 0;
  i64 ntiles = cdiv(count, TILE_V);
  DevBuf ts, tot;
  int rc;
  if ((rc = ts.alloc(ntiles * 8)) || (rc = tot.alloc(8))) return rc;
  hipLaunchKernelGGL(vbyte_len_kernel, dim3((unsigned)ntiles), dim3(PSG_WG), 0, stream(), d_vals, count, ts.as<u64>());
  PSG_HIP(hipGetLastError());
  if ((rc = scan_u64_inplace(ts.as<u64>(), ntiles, tot.as<u64>()))) return rc;
  u64 total = 0;
  PSG_HIP(hipMemcpyAsync(pinned_buf(3, 64), tot.p, 8, hipMemcpyDeviceToHost, stream()));   // pinned: pageable read-backs stall
  PSG_HIP(psg::sync_stream());
  memcpy(&total, pinned_buf(3, 64), 8);
  *nbytes = (i64)total;
  if ((i64)total > cap) { set_error("psg_vbyte_encode: output capacity too small"); return PSG_EINVAL; }
  hipLaunchKernelGGL(vbyte_write_kernel, dim3((unsigned)ntiles), dim3(PSG_WG), 0, stream(), d_vals, count, ts.as<u64>(), d_out, cap);
  PSG_HIP(hipGetLastError());
  PSG_HIP(psg::sync_stream());
  return 0;
}

// =======================================================================================
// K7: final merge.  M_h = merged order of half-blocks h..H-1.  mbv_h marks, for every
// slot of M_h, whether it holds an own suffix of half-block h (0) or the next element of
// M_{h+1} (1).  A tile of 4096 output slots walks down the levels: the elements that
// survive level h form a contiguous range of M_{h+1}.  This is the closed form of the
// "leftmost half-block whose gap head is 0" rule of merge.hpp:123-158.
// =======================================================================================
struct MergeLevel {
  const u32 *mbv;   // null on the last level
  i64 nbits;
  const u64 *samp;  // ones before every TILE_B-bit group
  const u32 *lo;
  const u8 *hi;
  i64 beg, size;
};

struct psg_merge_plan {
  int H = 0;
  i64 n = 0;
  std::vector<MergeLevel> levels;
  MergeLevel *d_levels = nullptr;
  std::vector<void *> owned;
};

#ifndef MT
#define MT 2048
#endif
//#define MT 2048                 // output slots per merge tile: 18 KiB of LDS -> 8 workgroups per CU (a tile is a chain of
#define MEPT (MT / PSG_WG)      // dependent loads, so the kernel is latency-bound: residency matters more than tile size)

// everything a level needs from memory for the range [q0, q0+cnt) of its merge bitvector: the
// thread's MEPT bits, its share of the partial popcount in front of q0, the group's rank sample
__device__ __forceinline__ void merge_level_loads(const MergeLevel &L, i64 q0, int cnt, u32 &bits, u32 &part, i64 &samp) {
  const i64 nwords = (L.nbits + 31) >> 5;
  const i64 g = q0 >> 12, gbase = g << 12;
  part = 0;
  if (threadIdx.x < 128) {
    i64 wb = gbase + (i64)threadIdx.x * 32;
    if (wb < q0) {
      u32 w = gload(L.mbv + (wb >> 5));
      i64 nb = q0 - wb;
      if (nb < 32) w &= (1u << nb) - 1u;
      part = __popc(w);
    }
  }
  const int e0 = threadIdx.x * MEPT;
  int n = std::max(0, std::min(MEPT, cnt - e0));
  bits = n > 0 ? get_bits(L.mbv, q0 + e0, n, nwords) : 0;
  samp = (i64)gload(L.samp + g);
}

// General case (H half-blocks).  A tile of MT output slots walks down the levels; a thread owns 8 consecutive
// positions of the current level: its own elements (bit 0) are consecutive elements of this half-block's PSA
// and are gathered straight into the output slots, its survivors (bit 1) are compacted -- slot numbers only --
// into the other slot buffer and form a contiguous range of the next level.
// OUT32: the values (below 2^32: positions inside one half-block) leave as plain u32 instead of packed uint40 -- the
// merge of sub-blocks into a half-block's partial SA (in-memory pSAscan, inmem_psascan.hpp:64-304)
// OUT = 2: two planes, the low 32 bits and the bits above -- the same merge when the enclosing range has 2^32 positions
// or more (half-blocks of configs[3]'s 16 GiB blocks)
template <bool HI, int OUT = 0>
__global__ __launch_bounds__(PSG_WG) void merge_kernel(const MergeLevel *lv, int H, i64 out_begin, i64 count, u8 *out, u8 *out_hi) {
  __shared__ u32 scratch[8];
  __shared__ __attribute__((aligned(16))) u16 cur[2][MT];
  __shared__ __attribute__((aligned(16))) u32 vlo[MT];
  __shared__ u8 vhi[MT];
  const int e0 = threadIdx.x * MEPT;
  const i64 x0 = out_begin + (i64)blockIdx.x * MT;
  const int len = (int)std::min<i64>(MT, out_begin + count - x0);
  i64 q0 = x0;
  int cnt = len, s = 0;
  bool identity = true;
  for (int h = 0; h < H && cnt > 0; ++h) {
    const MergeLevel L = lv[h];
    const int n = std::max(0, std::min(MEPT, cnt - e0));
    if (h == H - 1) {   // last half-block: everything left is its own, in order
      const u32 *p = L.lo + q0 + e0;
      const u8 *ph = L.hi ? L.hi + q0 + e0 : nullptr;
      u32 lo[MEPT], hi[MEPT];
#pragma unroll
      for (int q = 0; q < MEPT; ++q) { lo[q] = q < n ? gload(p + q) : 0u; hi[q] = (HI && ph && q < n) ? gload(ph + q) : 0u; }
#pragma unroll
      for (int q = 0; q < MEPT; ++q)
        if (q < n) {
          const int slot = identity ? e0 + q : cur[s][e0 + q];
          const u64 v = (u64)L.beg + lo[q] + (HI ? ((u64)hi[q] << 32) : 0);
          vlo[slot] = (u32)v; vhi[slot] = (u8)(v >> 32);
        }
      break;
    }
    u32 bits, part;
    i64 samp;
    merge_level_loads(L, q0, cnt, bits, part, samp);
    const u32 part_tot = block_sum<u32>(part, scratch);
    const i64 ones_q0 = samp + part_tot, zeros_q0 = q0 - ones_q0;
    u32 tot1;
    const u32 o = block_excl_scan<u32>((u32)__popc(bits), scratch, tot1);
    const u32 *p = L.lo + zeros_q0 + (e0 - (int)o);
    const u8 *ph = L.hi ? L.hi + zeros_q0 + (e0 - (int)o) : nullptr;
    u32 lo[MEPT], hi[MEPT];
#pragma unroll
    for (int q = 0; q < MEPT; ++q) {   // the gathers of the own elements, issued together
      const bool own = q < n && !((bits >> q) & 1u);
      const int k = q - __popc(bits & ((1u << q) - 1u));
      lo[q] = own ? gload(p + k) : 0u;
      hi[q] = (HI && own && ph) ? gload(ph + k) : 0u;
    }
#pragma unroll
    for (int q = 0; q < MEPT; ++q)
      if (q < n) {
        const int slot = identity ? e0 + q : cur[s][e0 + q];
        if ((bits >> q) & 1u) cur[s ^ 1][o + __popc(bits & ((1u << q) - 1u))] = (u16)slot;
        else {
          const u64 v = (u64)L.beg + lo[q] + (HI ? ((u64)hi[q] << 32) : 0);
          vlo[slot] = (u32)v; vhi[slot] = (u8)(v >> 32);
        }
      }
    __syncthreads();
    q0 = ones_q0; cnt = (int)tot1; s ^= 1; identity = false;
  }
  __syncthreads();
  if (OUT != 0) {
    u32 *o32 = (u32 *)out + (x0 - out_begin);
    for (int k = threadIdx.x; k < len; k += PSG_WG) o32[k] = vlo[k];
    if (OUT == 2) {
      u8 *o8 = out_hi + (x0 - out_begin);
      for (int k = threadIdx.x; k < len; k += PSG_WG) o8[k] = vhi[k];
    }
    return;
  }
  // pack 40-bit little-endian (types/uint40.hpp:42-104): 4 entries -> 5 dwords, staged in LDS (the slot
  // buffers are dead now) so that the tile leaves as whole 16-byte stores
  u32 *packed = (u32 *)&cur[0][0];                 // 8 KiB: half a tile (1024 entries = 5 KiB) at a time
  u8 *obase = out + 5 * (x0 - out_begin);
  const bool al16 = ((uintptr_t)obase & 15) == 0;
  for (int eb = 0; eb < len; eb += MT / 2) {
    const int ne = std::min(MT / 2, len - eb), nq = ne >> 2;
    if (eb) __syncthreads();
    for (int g4 = threadIdx.x; g4 < nq; g4 += PSG_WG) {
      int e = eb + 4 * g4;
      u32 l0 = vlo[e], l1 = vlo[e + 1], l2 = vlo[e + 2], l3 = vlo[e + 3];
      u32 h0 = vhi[e], h1 = vhi[e + 1], h2 = vhi[e + 2], h3 = vhi[e + 3];
      u32 *dst = packed + 5 * g4;
      dst[0] = l0;
      dst[1] = h0 | (l1 << 8);
      dst[2] = (l1 >> 24) | (h1 << 8) | (l2 << 16);
      dst[3] = (l2 >> 16) | (h2 << 16) | (l3 << 24);
      dst[4] = (l3 >> 8) | (h3 << 24);
    }
    __syncthreads();
    u32 *dst = (u32 *)(obase + 5 * eb);            // 5 * 1024 bytes per half: keeps the 16-byte alignment
    const int ndw = nq * 5;
    if (al16) {
      for (int k = threadIdx.x; k < (ndw >> 2); k += PSG_WG) ((uint4 *)dst)[k] = ((const uint4 *)packed)[k];
      for (int k = (ndw & ~3) + threadIdx.x; k < ndw; k += PSG_WG) dst[k] = packed[k];
    } else {
      for (int k = threadIdx.x; k < ndw; k += PSG_WG) dst[k] = packed[k];
    }
    for (int bb = nq * 20 + threadIdx.x; bb < 5 * ne; bb += PSG_WG) {   // ragged tail of the last tile
      int e = eb + bb / 5, rr = bb % 5;
      obase[5 * eb + bb] = (u8)(rr < 4 ? (vlo[e] >> (8 * rr)) & 255u : vhi[e]);
    }
  }
}

// ---- the general merge off its latency chain.  merge_kernel above walks the levels one after the other: the range of
// level h+1 a tile reads is known only when level h's rank sample and bits have arrived -- 16 dependent round trips per
// tile at 16 half-blocks (8.9 % of the HBM roofline, profiles/r02_configs2_kernel_stats.csv).  Here a cheap pre-pass
// walks the levels for every TILE BOUNDARY (one wave per boundary: the 4096-bit group of the position is popcounted by
// its 64 lanes, merge.hpp:123-158 for one position) and leaves the cursors q_h(t); a tile then knows all its ranges
// [q_h(t), q_h(t+1)) up front and issues the bit loads of all levels at once, resolves which output slot takes which
// element of which level in LDS only, and gathers its 8 values per thread in one go: three round trips instead of 2 H.
#define MCUR_MAXH 48
#define MCUR_G 8             // levels resolved per group: their bits and scan results live in registers
__global__ __launch_bounds__(256) void merge_tile_cursor_kernel(const MergeLevel *lv, int H, i64 out_begin, i64 count, i64 tile0, i64 nb, i64 *cur) {
  const i64 b = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);          // boundary handled by this wave
  if (b >= nb) return;
  const int lane = (int)lane_id();
  i64 q = std::min(out_begin + (tile0 + b) * MT, out_begin + count);
  for (int h = 0; h < H; ++h) {
    if (lane == 0) cur[b * H + h] = q;
    if (h == H - 1) break;
    const MergeLevel L = lv[h];
    i64 ones;
    if (q >= L.nbits) ones = L.nbits - L.size;                      // everything: all elements of the later half-blocks
    else {
      const i64 g = q >> 12, w0 = (g << 7) + 2 * lane;              // this lane's two words of the group
      u32 part = 0;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const i64 wb = (w0 + k) << 5;
        if (wb < q) { u32 w = gload(L.mbv + w0 + k); const i64 nbq = q - wb; if (nbq < 32) w &= (1u << nbq) - 1u; part += __popc(w); }
      }
      const u32 inc = wave_incl_scan(part);
      ones = (i64)gload(L.samp + g) + (i64)__shfl(inc, 63, 64);
    }
    q = ones;
  }
}

template <bool HI, int OUT>
__global__ __launch_bounds__(PSG_WG) __attribute__((amdgpu_waves_per_eu(8, 8))) void merge_kernel_cur(const MergeLevel *__restrict__ lv, int H, i64 out_begin, i64 count, const i64 *__restrict__ cur, i64 tile0, u8 *out, u8 *out_hi) {
  __shared__ i64 q0s[MCUR_MAXH], begs[MCUR_MAXH];       // per level, for the gather (indexed by each element's level)
  __shared__ const u32 *los[MCUR_MAXH];
  __shared__ const u8 *his[MCUR_MAXH];
  // 16 KiB: two slot buffers and the info words while the levels are resolved, the packed 40-bit output afterwards
  __shared__ __attribute__((aligned(16))) u32 work[MT * 2];
  u16 (*slotbuf)[MT] = (u16 (*)[MT])work;    // [2][MT]
  u32 *info = work + MT;                     // output slot -> level << 16 | index among the tile's own elements of that level
  u32 *packed = work;                        // MT * 5 / 4 words
  __shared__ u32 wsum[MCUR_G][4];
  const i64 t = blockIdx.x;
  const i64 x0 = out_begin + (tile0 + t) * MT;
  const int len = (int)std::min<i64>(MT, out_begin + count - x0);
  const int e0 = threadIdx.x * MEPT, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const i64 wbase = (i64)wave * 64 * MEPT;     // a level with no more than this many positions has nothing for this wave:
                                               // the ranges only shrink from level to level, so it then skips to the barriers
  for (int h = threadIdx.x; h < H; h += PSG_WG) {
    const MergeLevel L = lv[h];
    q0s[h] = cur[t * H + h];
    begs[h] = L.beg; los[h] = L.lo; his[h] = L.hi;
  }
  __syncthreads();
  int s = 0;
  bool identity = true;
  for (int h0 = 0; h0 < H - 1; h0 += MCUR_G) {
    const int ng = std::min(MCUR_G, H - 1 - h0);
    u32 bits[MCUR_G], o[MCUR_G];
    u32 act = 0;                               // wave-uniform: the levels of the group that have positions for this wave
    // the bits of all levels of the group: independent loads, issued together (bit 8.. of bits[g]: how many are valid)
#pragma unroll
    for (int g = 0; g < MCUR_G; ++g) {
      bits[g] = 0;
      if (g < ng) {
        const int h = h0 + g;                  // uniform: cursors and level descriptor come through scalar loads
        const i64 q0 = cur[t * H + h], c = cur[(t + 1) * H + h] - q0;
        if (c > wbase) {
          act |= 1u << g;
          const int n = (int)std::max<i64>(0, std::min<i64>(MEPT, c - e0));
          if (n > 0) bits[g] = get_bits(lv[h].mbv, q0 + e0, n, (lv[h].nbits + 31) >> 5) | ((u32)n << 8);
        }
      }
    }
    // ones in front of this thread's positions, for every level of the group: wave scans + one exchange (an idle wave
    // stands behind every busy one, nobody reads its sum)
#pragma unroll
    for (int g = 0; g < MCUR_G; ++g) {
      o[g] = 0;
      if ((act >> g) & 1u) {
        const u32 pc = (u32)__popc(bits[g] & 255u);
        const u32 inc = wave_incl_scan(pc);
        o[g] = inc - pc;
        if (lane_id() == 63) wsum[g][wave] = inc;
      }
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < MCUR_G; ++g)
      if ((act >> g) & 1u) for (int w = 0; w < wave; ++w) o[g] += wsum[g][w];
    // which output slot does position j of level h stand for?  own elements get their level and their index among the
    // tile's own elements of that level, survivors move on to the next level (slot numbers only, in LDS)
#pragma unroll
    for (int g = 0; g < MCUR_G; ++g) {
      if (g < ng) {
       if ((act >> g) & 1u) {
        const int npg = (int)(bits[g] >> 8);
        const u32 hbits = (u32)(h0 + g) << 16;
        // the thread's 8 slot numbers in ONE 16-byte LDS read (lane stride 16 bytes: conflict-free; eight 2-byte reads at
        // that stride were 8-way bank conflicts), then fully unrolled: survivors are appended to the other buffer, own
        // elements record level | index in the slot's info word
        const uint4 raw = identity ? make_uint4(0u, 0u, 0u, 0u) : *(const uint4 *)&slotbuf[s][e0];
        const u32 w4[4] = {raw.x, raw.y, raw.z, raw.w};
        u32 pos = o[g], zi = (u32)e0 - o[g];
        const u32 b = bits[g];
#pragma unroll
        for (int q = 0; q < MEPT; ++q) {
          if (q < npg) {
            const u32 slot = identity ? (u32)(e0 + q) : ((w4[q >> 1] >> (16 * (q & 1))) & 0xFFFFu);
            if ((b >> q) & 1u) slotbuf[s ^ 1][pos++] = (u16)slot;
            else info[slot] = hbits | zi++;
          }
        }
       }
        __syncthreads();
        s ^= 1; identity = false;
      }
    }
  }
  {   // the last half-block takes what is left, in order
    const i64 c = cur[(t + 1) * H + H - 1] - cur[t * H + H - 1];
    const int n = c > wbase ? (int)std::max<i64>(0, std::min<i64>(MEPT, c - e0)) : 0;
    const uint4 raw = identity ? make_uint4(0u, 0u, 0u, 0u) : *(const uint4 *)&slotbuf[s][e0];
    const u32 w4[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int q = 0; q < MEPT; ++q)
      if (q < n) {
        const u32 slot = identity ? (u32)(e0 + q) : ((w4[q >> 1] >> (16 * (q & 1))) & 0xFFFFu);
        info[slot] = ((u32)(H - 1) << 16) | (u32)(e0 + q);
      }
  }
  __syncthreads();
  // ---- the thread's 8 output slots: one gather each, all in flight together
  const int n = std::max(0, std::min(MEPT, len - e0));
  u64 v[MEPT];
#pragma unroll
  for (int q = 0; q < MEPT; ++q) {
    v[q] = 0;
    if (q < n) {
      const u32 inf = info[e0 + q];
      const int h = (int)(inf >> 16);
      const i64 idx = (h == H - 1 ? q0s[h] : q0s[h] - q0s[h + 1]) + (i64)(inf & 0xFFFFu);
      u64 x = (u64)gload(los[h] + idx);
      if (HI && his[h]) x |= (u64)gload(his[h] + idx) << 32;
      v[q] = (u64)begs[h] + x;
    }
  }
  if (OUT != 0) {
    u32 *o32 = (u32 *)out + (x0 - out_begin) + e0;
    if (n == MEPT && ((uintptr_t)o32 & 15) == 0) {
      ((uint4 *)o32)[0] = make_uint4((u32)v[0], (u32)v[1], (u32)v[2], (u32)v[3]);
      ((uint4 *)o32)[1] = make_uint4((u32)v[4], (u32)v[5], (u32)v[6], (u32)v[7]);
    } else for (int q = 0; q < n; ++q) o32[q] = (u32)v[q];
    if (OUT == 2) { u8 *o8 = out_hi + (x0 - out_begin) + e0; for (int q = 0; q < n; ++q) o8[q] = (u8)(v[q] >> 32); }
    return;
  }
  __syncthreads();                             // every info word has been read: the buffer becomes the packed output
  // values -> 40-bit little-endian (types/uint40.hpp:42-104), 8 entries = 10 dwords per thread, through LDS
  u32 *dst = packed + 10 * threadIdx.x;
#pragma unroll
  for (int g4 = 0; g4 < 2; ++g4) {
    const u32 l0 = (u32)v[4 * g4], l1 = (u32)v[4 * g4 + 1], l2 = (u32)v[4 * g4 + 2], l3 = (u32)v[4 * g4 + 3];
    const u32 b0 = (u32)(v[4 * g4] >> 32) & 255u, b1 = (u32)(v[4 * g4 + 1] >> 32) & 255u, b2 = (u32)(v[4 * g4 + 2] >> 32) & 255u,
              b3 = (u32)(v[4 * g4 + 3] >> 32) & 255u;
    dst[5 * g4 + 0] = l0;
    dst[5 * g4 + 1] = b0 | (l1 << 8);
    dst[5 * g4 + 2] = (l1 >> 24) | (b1 << 8) | (l2 << 16);
    dst[5 * g4 + 3] = (l2 >> 16) | (b2 << 16) | (l3 << 24);
    dst[5 * g4 + 4] = (l3 >> 8) | (b3 << 24);
  }
  __syncthreads();
  u8 *obase = out + 5 * (x0 - out_begin);
  const int nbytes = 5 * len, ndw = nbytes >> 2;
  if (((uintptr_t)obase & 15) == 0) {
    for (int k = threadIdx.x; k < (ndw >> 2); k += PSG_WG) ((uint4 *)obase)[k] = ((const uint4 *)packed)[k];
    for (int k = (ndw & ~3) + threadIdx.x; k < ndw; k += PSG_WG) ((u32 *)obase)[k] = packed[k];
  } else {
    for (int k = threadIdx.x; k < ndw; k += PSG_WG) ((u32 *)obase)[k] = packed[k];
  }
  for (int bb = 4 * ndw + threadIdx.x; bb < nbytes; bb += PSG_WG) obase[bb] = ((const u8 *)packed)[bb];   // ragged end of the last tile
}

// cursors + merge for the output range [out_begin, out_begin + out_count), in chunks of at most 2^20 tiles
template <bool HI, int OUT>
static int launch_merge_cur(int H, const MergeLevel *d_levels, i64 out_begin, i64 out_count, u8 *d_out, u8 *d_out_hi) {
  const i64 ntiles = cdiv(out_count, MT), CH = (i64)1 << 20;
  DevBuf cur;
  if (int rc = cur.alloc((std::min(ntiles, CH) + 1) * H * 8)) return rc;
  for (i64 t0 = 0; t0 < ntiles; t0 += CH) {
    const i64 nt = std::min(CH, ntiles - t0);
    hipLaunchKernelGGL(merge_tile_cursor_kernel, dim3((unsigned)cdiv(nt + 1, 4)), dim3(256), 0, stream(), d_levels, H, out_begin, out_count, t0, nt + 1, cur.as<i64>());
    hipLaunchKernelGGL((merge_kernel_cur<HI, OUT>), dim3((unsigned)nt), dim3(PSG_WG), 0, stream(), d_levels, H, out_begin, out_count, cur.as<i64>(), t0, d_out, d_out_hi);
  }
  PSG_HIP(hipGetLastError());
  return 0;
}
// Measured at configs[2]'s 16 half-blocks (profiles/r03_merge_kernel_pmc.txt): neither kernel waits on memory -- an element
// passes through H/2 levels on average and each is a scan + an LDS round trip, so the time is instruction issue and
// occupancy.  This kernel issues 0.66x the LDS and 0.8x the VALU instructions of the level-by-level one; at 4 waves per
// SIMD (34 KiB of LDS, 88 VGPRs) it was no faster, at 8 (19 KiB, 63 VGPRs) it takes 0.61x the time.  PSG_MERGE_CHAIN=1
// selects the level-by-level kernel, which also serves H > MCUR_MAXH.
static bool merge_cur_enabled(int H) { return H >= 3 && H <= MCUR_MAXH && !getenv("PSG_MERGE_CHAIN"); }

// Two half-blocks (one block): out[x] = bit ? psa1[rank1(x)] : psa0[rank0(x)].  No slot compaction in LDS:
// a thread's 8 output slots take consecutive elements of the two arrays, so its gathers walk two short
// runs (neighbouring lanes share sectors; the 8 loads are independent and issued together).
template <bool HI>
__global__ __launch_bounds__(PSG_WG) void merge2_kernel(MergeLevel L0, MergeLevel L1, i64 out_begin, i64 count, u8 *out) {
  __shared__ u32 scratch[8];
  __shared__ __attribute__((aligned(16))) u32 packed[MT * 5 / 4];
  const i64 x0 = out_begin + (i64)blockIdx.x * MT;
  const int len = (int)std::min<i64>(MT, out_begin + count - x0);
  const int e0 = threadIdx.x * MEPT;
  u32 bits, part;
  i64 samp;
  merge_level_loads(L0, x0, len, bits, part, samp);
  const u32 part_tot = block_sum<u32>(part, scratch);
  const i64 ones_q0 = samp + part_tot, zeros_q0 = x0 - ones_q0;
  const int n = std::max(0, std::min(MEPT, len - e0));
  u32 tot1;
  const u32 o = block_excl_scan<u32>((u32)__popc(bits), scratch, tot1);
  const u32 *p1 = L1.lo + ones_q0 + o, *p0 = L0.lo + zeros_q0 + (e0 - (int)o);
  const u8 *h1 = L1.hi ? L1.hi + ones_q0 + o : nullptr, *h0 = L0.hi ? L0.hi + zeros_q0 + (e0 - (int)o) : nullptr;
  u32 lo[MEPT], hi[MEPT];
#pragma unroll
  for (int q = 0; q < MEPT; ++q) {
    const bool one = (bits >> q) & 1u;
    const int below1 = __popc(bits & ((1u << q) - 1u)), k = one ? below1 : q - below1;
    lo[q] = 0; hi[q] = 0;
    if (q < n) {
      lo[q] = gload((one ? p1 : p0) + k);
      if (HI) { const u8 *hp = one ? h1 : h0; hi[q] = hp ? gload(hp + k) : 0u; }
    }
  }
  // values -> 40-bit little-endian (types/uint40.hpp:42-104), 8 entries = 10 dwords per thread, through LDS
  u64 v[MEPT];
#pragma unroll
  for (int q = 0; q < MEPT; ++q) {
    const bool one = (bits >> q) & 1u;
    v[q] = (u64)(one ? L1.beg : L0.beg) + lo[q] + (HI ? ((u64)hi[q] << 32) : 0);
  }
  u32 *dst = packed + 10 * threadIdx.x;
#pragma unroll
  for (int g4 = 0; g4 < 2; ++g4) {
    const u32 l0 = (u32)v[4 * g4], l1 = (u32)v[4 * g4 + 1], l2 = (u32)v[4 * g4 + 2], l3 = (u32)v[4 * g4 + 3];
    const u32 b0 = (u32)(v[4 * g4] >> 32) & 255u, b1 = (u32)(v[4 * g4 + 1] >> 32) & 255u, b2 = (u32)(v[4 * g4 + 2] >> 32) & 255u,
              b3 = (u32)(v[4 * g4 + 3] >> 32) & 255u;
    dst[5 * g4 + 0] = l0;
    dst[5 * g4 + 1] = b0 | (l1 << 8);
    dst[5 * g4 + 2] = (l1 >> 24) | (b1 << 8) | (l2 << 16);
    dst[5 * g4 + 3] = (l2 >> 16) | (b2 << 16) | (l3 << 24);
    dst[5 * g4 + 4] = (l3 >> 8) | (b3 << 24);
  }
  __syncthreads();
  u8 *obase = out + 5 * (x0 - out_begin);
  const int nbytes = 5 * len, ndw = nbytes >> 2;
  if (((uintptr_t)obase & 15) == 0) {
    for (int k = threadIdx.x; k < (ndw >> 2); k += PSG_WG) ((uint4 *)obase)[k] = ((const uint4 *)packed)[k];
    for (int k = (ndw & ~3) + threadIdx.x; k < ndw; k += PSG_WG) ((u32 *)obase)[k] = packed[k];
  } else {
    for (int k = threadIdx.x; k < ndw; k += PSG_WG) ((u32 *)obase)[k] = packed[k];
  }
  for (int bb = 4 * ndw + threadIdx.x; bb < nbytes; bb += PSG_WG) obase[bb] = ((const u8 *)packed)[bb];   // ragged end of the last tile
}

// Many two-way merges in one launch: the pairs (block, tail) of one level of the leaf merging (leaf_tree.hip).  bv is
// the concatenation of the pairs' merge bitvectors -- bit x belongs to the pass whose parent range holds position x --
// and the arrays are indexed by position: the block of pass p at [lbeg, lbeg + m), its tail behind it.  Partial SA
// values (relative to the enclosing range, u32) and BWT symbols move together (merge.hpp:123-158 for two half-blocks +
// bwt_merge.hpp:66-140 in one sweep).
__global__ __launch_bounds__(PSG_WG) void merge_pairs_kernel(const u32 *bv, i64 nbits, const u64 *samp, const psg::BatchGeom *geom, i64 npass, const u32 *tile_pass,
                                                               const u32 *psa, const u8 *bwt, const u8 *text, u32 *psa_out, u8 *bwt_out, i64 *i0_out) {
  __shared__ u32 scratch[8];
  const i64 x0 = (i64)blockIdx.x * MT;
  const int len = (int)std::min<i64>(MT, nbits - x0);
  const int e0 = threadIdx.x * MEPT;
  MergeLevel L0{};
  L0.mbv = bv; L0.nbits = nbits; L0.samp = samp;
  u32 bits, part;
  i64 sm;
  merge_level_loads(L0, x0, len, bits, part, sm);
  const u32 part_tot = block_sum<u32>(part, scratch);
  const i64 ones_x0 = sm + part_tot;
  const int n = std::max(0, std::min(MEPT, len - e0));
  u32 tot1;
  const u32 o = block_excl_scan<u32>((u32)__popc(bits), scratch, tot1);
  if (n <= 0) return;
  i64 p = tile_pass[blockIdx.x];
  psg::BatchGeom G = geom[p];
  u32 v[MEPT], c[MEPT];
#pragma unroll
  for (int q = 0; q < MEPT; ++q) {
    v[q] = 0; c[q] = 0;
    if (q < n) {
      const i64 x = x0 + e0 + q;
      while (x >= G.lbeg + G.m + G.T && p + 1 < npass) { ++p; G = geom[p]; }
      const bool one = (bits >> q) & 1u;
      const i64 r1 = ones_x0 + o + __popc(bits & ((1u << q) - 1u)) - G.ones_before;   // tail elements of this pass in front of x
      const i64 src = one ? G.lbeg + G.m + r1 : x - r1;
      v[q] = gload(psa + src);
      c[q] = gload(bwt + src);
      if (one && v[q] == (u32)(G.lbeg + G.m)) c[q] = text[G.lbeg + G.m - 1];          // the tail's first suffix: dummy -> the block's last symbol
      if (!one && v[q] == (u32)G.lbeg) i0_out[p] = x - G.lbeg;                        // the block's first suffix is the parent's
    }
  }
  u32 *po = psa_out + x0 + e0;
  u8 *bo = bwt_out + x0 + e0;
  if (n == MEPT) {
    ((uint4 *)po)[0] = make_uint4(v[0], v[1], v[2], v[3]);
    ((uint4 *)po)[1] = make_uint4(v[4], v[5], v[6], v[7]);
    *(uint2 *)bo = make_uint2(c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24), c[4] | (c[5] << 8) | (c[6] << 16) | (c[7] << 24));
  } else {
    for (int q = 0; q < n; ++q) { po[q] = v[q]; bo[q] = (u8)c[q]; }
  }
}

int psg::merge_pairs_tile() { return MT; }

int psg::merge_pairs_launch(const u32 *d_bv, i64 nbits, const BatchGeom *d_geom, i64 npass, const u32 *d_tile_pass, const u32 *d_psa, const u8 *d_bwt,
                            const u8 *d_text_range, u32 *d_psa_out, u8 *d_bwt_out, i64 *d_i0_out) {
  static_assert(MEPT == 8, "merge_pairs_kernel stores 8 entries per thread");
  PSG_REQUIRE(d_bv && nbits >= 1 && d_geom && npass >= 1 && d_tile_pass && d_psa && d_bwt && d_psa_out && d_bwt_out && d_i0_out, "merge_pairs_launch");
  PSG_REQUIRE(((uintptr_t)d_psa_out & 15) == 0 && ((uintptr_t)d_bwt_out & 7) == 0, "merge_pairs_launch: output alignment");
  const i64 ntiles = cdiv(nbits, TILE_B);
  DevBuf samp;
  if (int rc = samp.alloc((ntiles + 1) * 8)) return rc;
  hipLaunchKernelGGL((tile_popc_kernel<false>), dim3((unsigned)ntiles), dim3(PSG_WG), 0, stream(), d_bv, nbits, samp.as<u64>());
  if (int rc = scan_u64_inplace(samp.as<u64>(), ntiles, nullptr)) return rc;
  hipLaunchKernelGGL(merge_pairs_kernel, dim3((unsigned)cdiv(nbits, MT)), dim3(PSG_WG), 0, stream(), d_bv, nbits, samp.as<u64>(), d_geom, npass, d_tile_pass, d_psa, d_bwt,
                     d_text_range, d_psa_out, d_bwt_out, d_i0_out);
  PSG_HIP(hipGetLastError());
  return 0;   // (samp goes back to the pool: reuse is stream-ordered)
}

extern "C" void psg_merge_plan_free(psg_merge_plan_t *p) {
  if (!p) return;
  for (void *q : p->owned) psg::pool_free(q);
  if (p->d_levels) psg::pool_free(p->d_levels);
  delete p;
}

// need_psa = false: the partial SAs are not on the device (psg_merge_stream); lo/hi of the levels stay null
// need_mbv = false: neither are the merge bitvectors (they come slice by slice from host memory): mbv/samp stay null
static int plan_build(const psg_hb_desc *hbs, int H, bool need_psa, psg_merge_plan_t **out, bool need_mbv = true) {
  psg_merge_plan *p = new psg_merge_plan();
  p->H = H;
  std::vector<i64> nh(H + 1, 0);
  for (int h = H - 1; h >= 0; --h) {
    if (hbs[h].size < 1 || (need_psa && !hbs[h].d_psa_lo) || (need_mbv && h + 1 < H && !hbs[h].d_mbv) || (h > 0 && hbs[h].beg < hbs[h - 1].beg)) {
      psg_merge_plan_free(p); set_error("psg_merge_plan_create: bad half-block descriptor " + std::to_string(h)); return PSG_EINVAL;
    }
    nh[h] = nh[h + 1] + hbs[h].size;
  }
  p->n = nh[0];
  int rc = 0;
  for (int h = 0; h < H; ++h) {
    MergeLevel L{};
    L.lo = hbs[h].d_psa_lo; L.hi = hbs[h].d_psa_hi; L.beg = hbs[h].beg; L.size = hbs[h].size;
    if (h + 1 < H && !need_mbv) { L.mbv = nullptr; L.nbits = nh[h]; L.samp = nullptr; }
    else if (h + 1 < H) {
      L.mbv = hbs[h].d_mbv; L.nbits = nh[h];
      i64 ntiles = cdiv(L.nbits, TILE_B);
      void *samp = nullptr; DevBuf tot;
      hipError_t e = psg::pool_alloc(&samp, (size_t)(ntiles + 1) * 8);
      if (e != hipSuccess) { psg_merge_plan_free(p); set_error("merge plan: hipMalloc failed"); return PSG_ENOMEM; }
      p->owned.push_back(samp);
      if ((rc = tot.alloc(8))) { psg_merge_plan_free(p); return rc; }
      hipLaunchKernelGGL((tile_popc_kernel<false>), dim3((unsigned)ntiles), dim3(PSG_WG), 0, stream(), L.mbv, L.nbits, (u64 *)samp);
      if ((rc = scan_u64_inplace((u64 *)samp, ntiles, tot.as<u64>()))) { psg_merge_plan_free(p); return rc; }
      u64 ones = 0;
      if ((rc = psg::copy_d2h(&ones, tot.p, 8))) { psg_merge_plan_free(p); return rc; }
      if ((i64)ones != nh[h + 1]) {
        psg_merge_plan_free(p);
        set_error("merge plan: level " + std::to_string(h) + " has " + std::to_string(ones) + " ones, expected " + std::to_string(nh[h + 1]));
        return PSG_ECHECK;
      }
      L.samp = (const u64 *)samp;
    }
    p->levels.push_back(L);
  }
  hipError_t e = psg::pool_alloc((void **)&p->d_levels, sizeof(MergeLevel) * (size_t)H);
  if (e != hipSuccess) { psg_merge_plan_free(p); set_error("merge plan: hipMalloc failed"); return PSG_ENOMEM; }
  if ((rc = psg::copy_h2d(p->d_levels, p->levels.data(), sizeof(MergeLevel) * (size_t)H))) { psg_merge_plan_free(p); return rc; }
  *out = p;
  return 0;
}

extern "C" int psg_merge_plan_create(const psg_hb_desc *hbs, int H, psg_merge_plan_t **out) {
  PSG_REQUIRE(hbs && H >= 1 && out, "psg_merge_plan_create");
  return plan_build(hbs, H, true, out);
}

// enqueue the merge kernel for the output range [out_begin, out_begin + out_count); lv/L0/L1 carry the PSA pointers
static int merge_launch(int H, const MergeLevel *d_levels, const MergeLevel &L0, const MergeLevel &L1, bool any_hi,
                        i64 out_begin, i64 out_count, u8 *d_out) {
  const unsigned grid = (unsigned)cdiv(out_count, MT);
  if (H == 2 && !getenv("PSG_MERGE_GENERAL")) {   // one block: the two-way kernel
    if (any_hi) hipLaunchKernelGGL(merge2_kernel<true>, dim3(grid), dim3(PSG_WG), 0, stream(), L0, L1, out_begin, out_count, d_out);
    else hipLaunchKernelGGL(merge2_kernel<false>, dim3(grid), dim3(PSG_WG), 0, stream(), L0, L1, out_begin, out_count, d_out);
  } else if (merge_cur_enabled(H)) {
    if (any_hi) return launch_merge_cur<true, 0>(H, d_levels, out_begin, out_count, d_out, nullptr);
    return launch_merge_cur<false, 0>(H, d_levels, out_begin, out_count, d_out, nullptr);
  } else {
    if (any_hi) hipLaunchKernelGGL(merge_kernel<true>, dim3(grid), dim3(PSG_WG), 0, stream(), d_levels, H, out_begin, out_count, d_out, (u8 *)nullptr);
    else hipLaunchKernelGGL(merge_kernel<false>, dim3(grid), dim3(PSG_WG), 0, stream(), d_levels, H, out_begin, out_count, d_out, (u8 *)nullptr);
  }
  PSG_HIP(hipGetLastError());
  return 0;
}

// the merged order as u32 values (every beg + psa value must be below 2^32: the plan's `beg` are relative to the
// enclosing range).  Replaces the merging stage of the reference's in-memory sorter (inmem_psascan.hpp:233-304).
extern "C" int psg_merge_run_u32(const psg_merge_plan_t *p, int64_t out_begin, int64_t out_count, uint32_t *d_out) {
  PSG_REQUIRE(p && d_out && out_begin >= 0 && out_count >= 0 && out_begin + out_count <= p->n, "psg_merge_run_u32: range");
  for (const MergeLevel &L : p->levels) PSG_REQUIRE(!L.hi && L.beg + L.size <= 0x100000000ll, "psg_merge_run_u32: values must fit 32 bits");
  if (out_count == 0) return 0;
  EventTimer tm; tm.start();
  if (merge_cur_enabled(p->H)) { if (int rc = launch_merge_cur<false, 1>(p->H, p->d_levels, out_begin, out_count, (u8 *)d_out, nullptr)) return rc; }
  else hipLaunchKernelGGL((merge_kernel<false, 1>), dim3((unsigned)cdiv(out_count, MT)), dim3(PSG_WG), 0, stream(), p->d_levels, p->H, out_begin, out_count, (u8 *)d_out, (u8 *)nullptr);
  PSG_HIP(hipGetLastError());
  tm.stop();
  PSG_HIP(psg::sync_stream());
  note_kernel_ms(tm.ms());
  return 0;
}

// the same with values of up to 40 bits, as two planes (d_lo[k] = low 32 bits, d_hi[k] = bits 32..39): the partial SA
// of a range of 2^32 positions or more in the layout every consumer of partial SAs takes (psg_hb_desc, search parts)
extern "C" int psg_merge_run_planes(const psg_merge_plan_t *p, int64_t out_begin, int64_t out_count, uint32_t *d_lo, uint8_t *d_hi) {
  PSG_REQUIRE(p && d_lo && d_hi && out_begin >= 0 && out_count >= 0 && out_begin + out_count <= p->n, "psg_merge_run_planes: range");
  if (out_count == 0) return 0;
  bool any_hi = false;
  for (const MergeLevel &L : p->levels) any_hi |= L.hi != nullptr;
  EventTimer tm; tm.start();
  const unsigned grid = (unsigned)cdiv(out_count, MT);
  if (merge_cur_enabled(p->H)) {
    if (int rc = any_hi ? launch_merge_cur<true, 2>(p->H, p->d_levels, out_begin, out_count, (u8 *)d_lo, d_hi) : launch_merge_cur<false, 2>(p->H, p->d_levels, out_begin, out_count, (u8 *)d_lo, d_hi)) return rc;
  } else if (any_hi) hipLaunchKernelGGL((merge_kernel<true, 2>), dim3(grid), dim3(PSG_WG), 0, stream(), p->d_levels, p->H, out_begin, out_count, (u8 *)d_lo, d_hi);
  else hipLaunchKernelGGL((merge_kernel<false, 2>), dim3(grid), dim3(PSG_WG), 0, stream(), p->d_levels, p->H, out_begin, out_count, (u8 *)d_lo, d_hi);
  PSG_HIP(hipGetLastError());
  tm.stop();
  PSG_HIP(psg::sync_stream());
  note_kernel_ms(tm.ms());
  return 0;
}

extern "C" int psg_merge_run(const psg_merge_plan_t *p, int64_t out_begin, int64_t out_count, uint8_t *d_out) {
  PSG_REQUIRE(p && d_out && out_begin >= 0 && out_count >= 0 && out_begin + out_count <= p->n, "psg_merge_run: range");
  PSG_REQUIRE(((uintptr_t)d_out & 3) == 0, "psg_merge_run: output must be 4-byte aligned");
  if (out_count == 0) return 0;
  EventTimer tm; tm.start();
  bool any_hi = false;
  for (const MergeLevel &L : p->levels) any_hi |= L.hi != nullptr;
  if (int rc = merge_launch(p->H, p->d_levels, p->levels[0], p->levels[p->H > 1 ? 1 : 0], any_hi, out_begin, out_count, d_out)) return rc;
  tm.stop();
  PSG_HIP(psg::sync_stream());
  note_kernel_ms(tm.ms());
  return 0;
}

// =======================================================================================
// block-per-GPU schedule: rank queries on a merge bitvector, merge plan over slices
// =======================================================================================
// ones[k] = popcount of bits [0, pos[k]): samples per 4096-bit group + partial popcount.  Block k, 128 threads.
__global__ __launch_bounds__(128) void bits_rank1_kernel(const u32 *bits, i64 nbits, const u64 *samp, const i64 *pos, i64 *ones, i64 total) {
  __shared__ u32 red[2];
  const i64 q = pos[blockIdx.x];
  if (q >= nbits) { if (threadIdx.x == 0) ones[blockIdx.x] = total; return; }
  const i64 g = q >> 12, gbase = g << 12;
  u32 part = 0;
  const i64 wb = gbase + (i64)threadIdx.x * 32;
  if (wb < q) { u32 w = gload(bits + (wb >> 5)); const i64 nb = q - wb; if (nb < 32) w &= (1u << nb) - 1u; part = __popc(w); }
  const u32 inc = wave_incl_scan(part);
  if (lane_id() == 63) red[threadIdx.x >> 6] = inc;
  __syncthreads();
  if (threadIdx.x == 0) ones[blockIdx.x] = (i64)samp[g] + red[0] + red[1];
}

extern "C" int psg_bits_rank1(const uint32_t *d_bits, int64_t nbits, const int64_t *h_pos, int64_t count, int64_t *h_ones) {
  PSG_REQUIRE(d_bits && nbits >= 0 && h_pos && h_ones && count >= 0, "psg_bits_rank1");
  if (count == 0) return 0;
  for (i64 k = 0; k < count; ++k) PSG_REQUIRE(h_pos[k] >= 0 && h_pos[k] <= nbits, "psg_bits_rank1: position out of range");
  const i64 ntiles = std::max<i64>(1, cdiv(nbits, TILE_B));
  DevBuf samp, tot, pos, out;
  int rc;
  if ((rc = samp.alloc((ntiles + 1) * 8)) || (rc = tot.alloc(8)) || (rc = pos.alloc(count * 8)) || (rc = out.alloc(count * 8))) return rc;
  hipLaunchKernelGGL((tile_popc_kernel<false>), dim3((unsigned)ntiles), dim3(PSG_WG), 0, stream(), d_bits, nbits, samp.as<u64>());
  if ((rc = scan_u64_inplace(samp.as<u64>(), ntiles, tot.as<u64>()))) return rc;
  u64 total = 0;
  if ((rc = psg::copy_d2h(&total, tot.p, 8))) return rc;
  if ((rc = psg::copy_h2d(pos.p, h_pos, (size_t)count * 8))) return rc;
  hipLaunchKernelGGL(bits_rank1_kernel, dim3((unsigned)count), dim3(128), 0, stream(), d_bits, nbits, samp.as<u64>(), pos.as<i64>(), out.as<i64>(), (i64)total);
  PSG_HIP(hipGetLastError());
  return psg::copy_d2h(h_ones, out.p, (size_t)count * 8);
}

// absolute rank samples of a slice: samp[g] += ones_before for every group of the slice
__global__ __launch_bounds__(PSG_WG) void add_const_u64_kernel(u64 *v, i64 n, u64 c) {
  i64 k = (i64)blockIdx.x * PSG_WG + threadIdx.x;
  if (k < n) v[k] += c;
}

extern "C" int psg_merge_plan_create_sliced(const psg_hb_slice_desc *lv, int H, psg_merge_plan_t **out) {
  PSG_REQUIRE(lv && H >= 1 && out, "psg_merge_plan_create_sliced");
  psg_merge_plan *p = new psg_merge_plan();
  p->H = H;
  int rc = 0;
  i64 n = 0;
  for (int h = 0; h < H; ++h) {
    const psg_hb_slice_desc &D = lv[h];
    if (D.size < 1 || D.psa_first < 0 || D.psa_count < 0 || D.psa_first + D.psa_count > D.size || (D.psa_count > 0 && !D.d_psa_lo) ||
        (h + 1 < H && D.n_words > 0 && (!D.d_mbv_words || (D.first_word & 127) != 0 || D.first_word < 0))) {
      psg_merge_plan_free(p); set_error("psg_merge_plan_create_sliced: bad level " + std::to_string(h)); return PSG_EINVAL;
    }
    n += D.size;
    MergeLevel L{};
    L.beg = D.beg; L.size = D.size;
    L.lo = D.d_psa_lo ? D.d_psa_lo - D.psa_first : nullptr;       // absolute element indices land inside the slice
    L.hi = D.d_psa_hi ? D.d_psa_hi - D.psa_first : nullptr;
    if (h + 1 < H) {
      L.nbits = D.nbits;
      L.mbv = D.d_mbv_words ? D.d_mbv_words - D.first_word : nullptr;
      const i64 slice_bits = D.n_words * 32, ntiles = std::max<i64>(1, cdiv(slice_bits, TILE_B));
      void *samp = nullptr;
      if (psg::pool_alloc(&samp, (size_t)(ntiles + 1) * 8) != hipSuccess) { psg_merge_plan_free(p); set_error("merge plan: hipMalloc failed"); return PSG_ENOMEM; }
      p->owned.push_back(samp);
      if (D.n_words > 0) {
        hipLaunchKernelGGL((tile_popc_kernel<false>), dim3((unsigned)ntiles), dim3(PSG_WG), 0, stream(), D.d_mbv_words, slice_bits, (u64 *)samp);
        if ((rc = scan_u64_inplace((u64 *)samp, ntiles, nullptr))) { psg_merge_plan_free(p); return rc; }
        hipLaunchKernelGGL(add_const_u64_kernel, dim3((unsigned)cdiv(ntiles, PSG_WG)), dim3(PSG_WG), 0, stream(), (u64 *)samp, ntiles, (u64)D.ones_before);
      }
      L.samp = (const u64 *)samp - (D.first_word >> 7);            // one sample per 4096 bits = 128 words
    }
    p->levels.push_back(L);
  }
  p->n = n;
  hipError_t e = psg::pool_alloc((void **)&p->d_levels, sizeof(MergeLevel) * (size_t)H);
  if (e != hipSuccess) { psg_merge_plan_free(p); set_error("merge plan: hipMalloc failed"); return PSG_ENOMEM; }
  if ((rc = psg::copy_h2d(p->d_levels, p->levels.data(), sizeof(MergeLevel) * (size_t)H))) { psg_merge_plan_free(p); return rc; }
  *out = p;
  return 0;
}

// =======================================================================================
// K7 with the partial suffix arrays in host memory (psg_merge_stream)
// =======================================================================================
// cur[b * H + h] = number of own elements of half-block h among the first xs[b] output slots: the walk of
// merge_kernel for ONE position (rank1 of every level from its samples + a partial popcount).  Block b, 128 threads.
__global__ __launch_bounds__(128) void merge_cursor_kernel(const MergeLevel *lv, int H, const i64 *xs, i64 *cur) {
  __shared__ u32 red[2];
  i64 q = xs[blockIdx.x];
  for (int h = 0; h < H; ++h) {
    const MergeLevel L = lv[h];
    if (h == H - 1) { if (threadIdx.x == 0) cur[(i64)blockIdx.x * H + h] = q; break; }
    i64 ones;
    if (q >= L.nbits) ones = L.nbits - L.size;       // everything: all elements of the later half-blocks
    else {
      const i64 g = q >> 12, gbase = g << 12;
      u32 part = 0;
      const i64 wb = gbase + (i64)threadIdx.x * 32;
      if (wb < q) { u32 w = gload(L.mbv + (wb >> 5)); const i64 nb = q - wb; if (nb < 32) w &= (1u << nb) - 1u; part = __popc(w); }
      const u32 inc = wave_incl_scan(part);
      if (lane_id() == 63) red[threadIdx.x >> 6] = inc;
      __syncthreads();
      ones = (i64)gload(L.samp + g) + red[0] + red[1];
      __syncthreads();
    }
    if (threadIdx.x == 0) cur[(i64)blockIdx.x * H + h] = q - ones;
    q = ones;
  }
}

static bool host_ptr_is_pinned(const void *p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeHost;
}

static double wall_ms() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

// copy `pieces` (dst, src, bytes) with a few helper threads (pageable -> pinned runs at 30 GiB/s on one core,
// ~100 GiB/s on eight; the PCIe link takes 50 GiB/s)
struct CopyPiece { char *dst; const char *src; size_t bytes; };
static void parallel_memcpy(const std::vector<CopyPiece> &pieces) {
  size_t total = 0;
  for (auto &c : pieces) total += c.bytes;
  const int nt = total < ((size_t)8 << 20) ? 1 : 8;
  if (nt == 1) { for (auto &c : pieces) memcpy(c.dst, c.src, c.bytes); return; }
  const size_t per = (total + nt - 1) / nt;
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      size_t lo = per * t, hi = std::min(total, lo + per), off = 0;   // this thread copies bytes [lo, hi) of the concatenation
      for (auto &c : pieces) {
        const size_t a = std::max(lo, off), b = std::min(hi, off + c.bytes);
        if (a < b) memcpy(c.dst + (a - off), c.src + (a - off), b - a);
        off += c.bytes;
      }
    });
  for (auto &t : th) t.join();
}

// ---- a merge bitvector leaves HBM (construct_sa --hbm-limit; the reference writes its gap arrays to files,
// gap_array.hpp:156-182, and reads them back during the merge, merge.hpp:80,145)
extern "C" int64_t psg_mbv_spill_words(int64_t nbits) { return ((nbits + 31) / 32 + 3) / 4 * 4; }
extern "C" int psg_mbv_spill(const uint32_t *d_mbv, int64_t nbits, uint32_t *h_words, uint64_t *h_samp) {
  PSG_REQUIRE(d_mbv && nbits >= 1 && h_words && h_samp, "psg_mbv_spill");
  const i64 ntiles = cdiv(nbits, TILE_B), nwords = (nbits + 31) / 32;
  DevBuf samp;
  int rc;
  if ((rc = samp.alloc((ntiles + 1) * 8))) return rc;
  hipLaunchKernelGGL((tile_popc_kernel<false>), dim3((unsigned)ntiles), dim3(PSG_WG), 0, stream(), d_mbv, nbits, samp.as<u64>());
  PSG_HIP(hipGetLastError());
  if ((rc = scan_u64_inplace(samp.as<u64>(), ntiles, samp.as<u64>() + ntiles))) return rc;   // the total lands behind the last sample
  if ((rc = psg::copy_d2h(h_samp, samp.p, (size_t)(ntiles + 1) * 8))) return rc;
  if ((rc = psg::copy_d2h(h_words, d_mbv, (size_t)nwords * 4))) return rc;
  for (i64 w = nwords; w < psg_mbv_spill_words(nbits); ++w) h_words[w] = 0;
  if (nbits & 31) h_words[nwords - 1] &= (1u << (nbits & 31)) - 1u;
  return 0;
}

// the same in the background (psg_d2h_begin): the samples are there on return, the words when psg_copy_wait says so; d_mbv
// (from psg_malloc) is handed over and freed once drained.  psg_mbv_spill_finish then clears what lies behind bit nbits.
extern "C" int psg_mbv_spill_begin(uint32_t *d_mbv, int64_t nbits, uint32_t *h_words, uint64_t *h_samp, psg_copy_t **out) {
  PSG_REQUIRE(d_mbv && nbits >= 1 && h_words && h_samp && out, "psg_mbv_spill_begin");
  const i64 ntiles = cdiv(nbits, TILE_B), nwords = (nbits + 31) / 32;
  DevBuf samp;
  int rc;
  if ((rc = samp.alloc((ntiles + 1) * 8))) return rc;
  hipLaunchKernelGGL((tile_popc_kernel<false>), dim3((unsigned)ntiles), dim3(PSG_WG), 0, stream(), (const u32 *)d_mbv, nbits, samp.as<u64>());
  PSG_HIP(hipGetLastError());
  if ((rc = scan_u64_inplace(samp.as<u64>(), ntiles, samp.as<u64>() + ntiles))) return rc;
  if ((rc = psg::copy_d2h(h_samp, samp.p, (size_t)(ntiles + 1) * 8))) return rc;
  return psg_d2h_begin(h_words, d_mbv, nwords * 4, 1, out);
}
extern "C" int psg_mbv_spill_finish(uint32_t *h_words, int64_t nbits) {
  PSG_REQUIRE(h_words && nbits >= 1, "psg_mbv_spill_finish");
  const i64 nwords = (nbits + 31) / 32;
  for (i64 w = nwords; w < psg_mbv_spill_words(nbits); ++w) h_words[w] = 0;
  if (nbits & 31) h_words[nwords - 1] &= (1u << (nbits & 31)) - 1u;
  return 0;
}

// number of one bits in front of bit q of a merge bitvector in host memory (samples per 4096 bits + popcount)
static i64 host_rank1(const u32 *words, const u64 *samp, i64 nbits, i64 q) {
  const i64 ntiles = cdiv(nbits, TILE_B);
  if (q >= nbits) return (i64)samp[ntiles];
  const i64 g = q >> 12;
  i64 ones = (i64)samp[g];
  for (i64 w = g << 7; w < (q >> 5); ++w) ones += __builtin_popcount(words[w]);
  if (q & 31) ones += __builtin_popcount(words[q >> 5] & ((1u << (q & 31)) - 1u));
  return ones;
}

extern "C" int psg_merge_stream(const psg_hb_host_desc *hbs, int H, int64_t slice_entries, psg_merge_check *check,
                                psg_sink_fn sink, void *sink_ctx, psg_merge_stream_stats *stats) {
  PSG_REQUIRE(hbs && H >= 1 && slice_entries >= 1, "psg_merge_stream");
  PSG_REQUIRE(!check || (check->d_text && check->n > 0), "psg_merge_stream: check needs the text on the device");
  const double w0 = wall_ms();
  slice_entries = cdiv(slice_entries, MT) * MT;        // whole merge tiles: keeps every slice's output 16-byte aligned
  std::vector<psg_hb_desc> dd((size_t)H);
  bool any_hi = false;
  for (int h = 0; h < H; ++h) {
    PSG_REQUIRE(hbs[h].h_psa_lo || hbs[h].d_psa_lo, "psg_merge_stream: partial suffix array missing");
    dd[(size_t)h] = psg_hb_desc{hbs[h].beg, hbs[h].size, nullptr, nullptr, hbs[h].d_mbv};
    any_hi |= hbs[h].d_psa_lo ? hbs[h].d_psa_hi != nullptr : hbs[h].h_psa_hi != nullptr;
  }
  // merge bitvectors in host memory (all of them or none): every slice brings the words it needs along
  bool host_mbv = false;
  for (int h = 0; h + 1 < H; ++h) host_mbv |= !hbs[h].d_mbv && hbs[h].h_mbv;
  if (host_mbv)
    for (int h = 0; h + 1 < H; ++h) PSG_REQUIRE(!hbs[h].d_mbv && hbs[h].h_mbv && hbs[h].h_mbv_samp, "psg_merge_stream: merge bitvectors in host memory: all of them, with their rank samples");
  psg_merge_plan *plan = nullptr;
  if (int rc = plan_build(dd.data(), H, false, &plan, !host_mbv)) return rc;
  struct PlanGuard { psg_merge_plan *p; ~PlanGuard() { psg_merge_plan_free(p); } } plan_guard{plan};
  if (host_mbv) {
    // a slice carries the bits of up to H levels: ~ slice/16 bytes per level on average + up to 1.5 KiB of alignment each,
    // twice (two slots), next to 9 bytes per entry of partial SA pieces and output.  Smaller slices bound the staging --
    // against the 16 Mi entries that keep the copies long, and against the memory the budget leaves (many small blocks
    // under psg_set_memory_limit: H in the tens of thousands).
    if (slice_entries > ((i64)16 << 20)) slice_entries = (i64)16 << 20;
    const double avail = (double)mem_available();
    auto staging = [&](i64 se) { return 2.0 * ((double)H * ((double)se / 16.0 + 1600.0) + 9.0 * (double)se + 64.0 * (double)H); };
    while (slice_entries > MT && staging(slice_entries) > 0.5 * avail) slice_entries = std::max<i64>(MT, slice_entries / 2 / MT * MT);   // (whole merge tiles)
  }
  const i64 n = plan->n, ns = cdiv(n, slice_entries);
  psg_merge_stream_stats st = {};
  st.slices = ns;
  // ---- cursors of every half-block at every slice boundary
  std::vector<i64> xs((size_t)ns + 1);
  for (i64 k = 0; k <= ns; ++k) xs[(size_t)k] = std::min<i64>(n, k * slice_entries);
  int rc;
  std::vector<i64> cur((size_t)(ns + 1) * H), qpos;      // qpos[b * H + h]: where the walk of boundary b stands in level h's bitvector
  if (host_mbv) {
    qpos.resize((size_t)(ns + 1) * H);
    for (i64 b = 0; b <= ns; ++b) {
      i64 q = xs[(size_t)b];
      for (int h = 0; h < H; ++h) {
        qpos[(size_t)b * H + h] = q;
        if (h == H - 1) { cur[(size_t)b * H + h] = q; break; }
        const i64 ones = host_rank1(hbs[h].h_mbv, hbs[h].h_mbv_samp, plan->levels[(size_t)h].nbits, q);
        cur[(size_t)b * H + h] = q - ones;
        q = ones;
      }
    }
  } else {
    DevBuf xs_d, cur_d;
    if ((rc = xs_d.alloc((ns + 1) * 8)) || (rc = cur_d.alloc((ns + 1) * H * 8))) return rc;
    if ((rc = psg::copy_h2d(xs_d.p, xs.data(), (size_t)(ns + 1) * 8))) return rc;
    hipLaunchKernelGGL(merge_cursor_kernel, dim3((unsigned)(ns + 1)), dim3(128), 0, stream(), plan->d_levels, H, xs_d.as<i64>(), cur_d.as<i64>());
    PSG_HIP(hipGetLastError());
    if ((rc = psg::copy_d2h(cur.data(), cur_d.p, cur.size() * 8))) return rc;
  }
  for (int h = 0; h < H; ++h)
    if (cur[(size_t)ns * H + h] != hbs[h].size || cur[(size_t)h] != 0) { set_error("psg_merge_stream: cursor check failed at half-block " + std::to_string(h)); return PSG_ECHECK; }
  // ---- buffers: two slots each
  const i64 lo_cap = slice_entries * 4 + (i64)H * 16, hi_cap = any_hi ? slice_entries + (i64)H * 16 : 0, in_cap = lo_cap + hi_cap;
  const i64 out_cap = 5 * slice_entries + 16;
  // host-resident merge bitvectors: per slice and level the words [w0, w1) (whole groups of 128 words) + their samples
  auto mbv_piece = [&](i64 k, int h, i64 &w0, i64 &w1) {
    const i64 nw = cdiv(cdiv(plan->levels[(size_t)h].nbits, 32), 128) * 128;
    w0 = (qpos[(size_t)k * H + h] >> 5) / 128 * 128;
    w1 = std::min(nw, ((qpos[(size_t)(k + 1) * H + h] + 31) >> 5) / 128 * 128 + 256);
    if (w1 < w0) w1 = w0;
  };
  i64 mbv_cap = 0;                                   // bytes of staging a slice needs for bitvector words + samples
  if (host_mbv)
    for (i64 k = 0; k < ns; ++k) {
      i64 need = 0;
      for (int h = 0; h + 1 < H; ++h) { i64 w0, w1; mbv_piece(k, h, w0, w1); need += (w1 - w0) * 4 + ((w1 - w0) / 128 + 2) * 8; }
      mbv_cap = std::max(mbv_cap, need + 64);
    }
  DevBuf din[2], dout[2], dlv[2], dmbv[2], acc;
  char *pin_mbv[2] = {nullptr, nullptr};
  char *pin_in[2] = {nullptr, nullptr}, *pin_out[2] = {nullptr, nullptr};
  MergeLevel *pin_lv[2] = {nullptr, nullptr};
  const bool direct = [&] {
    for (int h = 0; h < H; ++h)
      if (!hbs[h].d_psa_lo && (!host_ptr_is_pinned(hbs[h].h_psa_lo) || (hbs[h].h_psa_hi && !host_ptr_is_pinned(hbs[h].h_psa_hi)))) return false;
    return true;
  }();
  for (int s = 0; s < 2; ++s) {
    if ((rc = din[s].alloc(in_cap)) || (rc = dout[s].alloc(out_cap)) || (rc = dlv[s].alloc((i64)sizeof(MergeLevel) * H))) return rc;
    if (!direct) pin_in[s] = (char *)pinned_buf(8 + s, (size_t)in_cap);
    if (sink) pin_out[s] = (char *)pinned_buf(10 + s, (size_t)out_cap);
    if ((!direct && !pin_in[s]) || (sink && !pin_out[s])) { set_error("psg_merge_stream: pinned host allocation failed"); return PSG_ENOMEM; }
    if (host_mbv) {
      if ((rc = dmbv[s].alloc(mbv_cap))) return rc;
      if (hipHostMalloc((void **)&pin_mbv[s], (size_t)mbv_cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); set_error("psg_merge_stream: pinned host allocation failed"); return PSG_ENOMEM; }
    }
  }
  pin_lv[0] = (MergeLevel *)pinned_buf(12, 2 * sizeof(MergeLevel) * (size_t)H);
  if (!pin_lv[0]) { set_error("psg_merge_stream: pinned host allocation failed"); return PSG_ENOMEM; }
  pin_lv[1] = pin_lv[0] + H;
  if (check) { if ((rc = acc.alloc(24))) return rc; PSG_HIP(hipMemsetAsync(acc.p, 0, 24, stream())); }
  hipStream_t up = side_stream();
  if (!up) { set_error("psg_merge_stream: cannot create the copy stream"); return PSG_EDEVICE; }
  hipEvent_t ev_up[2] = {event_acquire(), event_acquire()}, ev_dn[2] = {event_acquire(), event_acquire()};
  hipEvent_t ev_k0[2] = {event_acquire(), event_acquire()}, ev_k1[2] = {event_acquire(), event_acquire()};
  struct EvGuard { hipEvent_t *a, *b, *c, *d; ~EvGuard() { for (int s = 0; s < 2; ++s) { event_release(a[s]); event_release(b[s]); event_release(c[s]); event_release(d[s]); } } } ev_guard{ev_up, ev_dn, ev_k0, ev_k1};
  struct PinMbvGuard { char **p; ~PinMbvGuard() { for (int s = 0; s < 2; ++s) if (p[s]) (void)hipHostFree(p[s]); } } pin_mbv_guard{pin_mbv};
  PSG_HIP(psg::sync_stream());   // cursor work and the memset are done before the copy stream starts
  auto wait_event = [](hipEvent_t e) { hipError_t q; while ((q = hipEventQuery(e)) == hipErrorNotReady) { } return q; };
  int result = 0;
  for (i64 k = 0; k < ns + 2 && !result; ++k) {
    if (k >= 2) {                                      // slice k-2 has left the device: hand it to the sink
      const int s = (int)(k & 1);
      if (wait_event(ev_dn[s]) != hipSuccess) { set_error("psg_merge_stream: device error"); result = PSG_EDEVICE; break; }
      float f = 0;
      (void)hipEventElapsedTime(&f, ev_k0[s], ev_k1[s]);
      st.kernel_ms += f;
      if (sink) {
        const i64 x0 = xs[(size_t)k - 2], cnt = xs[(size_t)k - 1] - x0;
        const double t0 = wall_ms();
        if (sink(sink_ctx, (const uint8_t *)pin_out[s], x0, cnt) != 0) { set_error("psg_merge_stream: the sink reported an error"); result = PSG_ECHECK; break; }
        st.sink_ms += wall_ms() - t0;
      }
    }
    if (k >= ns) continue;
    const int s = (int)(k & 1);
    const i64 x0 = xs[(size_t)k], cnt = xs[(size_t)k + 1] - x0;
    // ---- stage the PSA pieces of slice k: device layout = pieces back to back, each 16-byte aligned
    const double t0 = wall_ms();
    std::vector<CopyPiece> pieces;
    i64 off = 0, offh = lo_cap;
    for (int h = 0; h < H; ++h) {
      const i64 c0 = cur[(size_t)k * H + h], c1 = cur[(size_t)(k + 1) * H + h], len = c1 - c0;
      MergeLevel L = plan->levels[(size_t)h];
      if (hbs[h].d_psa_lo) {                                       // resident in HBM: used where it lies
        L.lo = hbs[h].d_psa_lo;
        L.hi = hbs[h].d_psa_hi;
        if (any_hi && !L.hi) {                                     // the kernel variant with high bytes reads them for every level
          L.hi = (const u8 *)(din[s].as<char>() + offh) - c0;
          if (len > 0) {
            if (direct) PSG_HIP(hipMemsetAsync(din[s].as<char>() + offh, 0, (size_t)len, up));
            else memset(pin_in[s] + offh, 0, (size_t)len);
          }
          offh += (len + 15) / 16 * 16;
        }
        pin_lv[s][h] = L;
        continue;
      }
      L.lo = (const u32 *)(din[s].as<char>() + off) - c0;          // absolute indices c0.. land inside the piece
      L.hi = nullptr;
      if (len > 0) {
        if (direct) PSG_HIP(hipMemcpyAsync(din[s].as<char>() + off, hbs[h].h_psa_lo + c0, (size_t)len * 4, hipMemcpyHostToDevice, up));
        else pieces.push_back({pin_in[s] + off, (const char *)(hbs[h].h_psa_lo + c0), (size_t)len * 4});
        st.h2d_bytes += len * 4;
      }
      off += (len * 4 + 15) / 16 * 16;
      if (any_hi) {
        L.hi = (const u8 *)(din[s].as<char>() + offh) - c0;
        if (len > 0) {
          if (hbs[h].h_psa_hi) {
            if (direct) PSG_HIP(hipMemcpyAsync(din[s].as<char>() + offh, hbs[h].h_psa_hi + c0, (size_t)len, hipMemcpyHostToDevice, up));
            else pieces.push_back({pin_in[s] + offh, (const char *)(hbs[h].h_psa_hi + c0), (size_t)len});
          } else if (direct) PSG_HIP(hipMemsetAsync(din[s].as<char>() + offh, 0, (size_t)len, up));
          else memset(pin_in[s] + offh, 0, (size_t)len);
          st.h2d_bytes += len;
        }
        offh += (len + 15) / 16 * 16;
      }
      pin_lv[s][h] = L;
    }
    if (host_mbv) {   // the words of every level this slice touches + the rank samples of their groups
      i64 moff = 0;
      for (int h = 0; h + 1 < H; ++h) {
        i64 w0, w1;
        mbv_piece(k, h, w0, w1);
        const i64 nwords_real = psg_mbv_spill_words(plan->levels[(size_t)h].nbits);
        const i64 wcopy = std::max<i64>(0, std::min(w1, nwords_real) - w0);
        pieces.push_back({pin_mbv[s] + moff, (const char *)(hbs[h].h_mbv + w0), (size_t)wcopy * 4});
        if (w1 - w0 > wcopy) memset(pin_mbv[s] + moff + wcopy * 4, 0, (size_t)(w1 - w0 - wcopy) * 4);
        pin_lv[s][h].mbv = (const u32 *)(dmbv[s].as<char>() + moff) - w0;
        moff += (w1 - w0) * 4;
        const i64 g0 = w0 / 128, ng = (w1 - w0) / 128 + 1, ntiles = cdiv(plan->levels[(size_t)h].nbits, TILE_B);
        u64 *sd = (u64 *)(pin_mbv[s] + moff);
        for (i64 g = 0; g < ng; ++g) sd[g] = hbs[h].h_mbv_samp[std::min(g0 + g, ntiles)];
        pin_lv[s][h].samp = (const u64 *)(dmbv[s].as<char>() + moff) - g0;
        moff += (ng + 1) * 8;
        st.h2d_bytes += (w1 - w0) * 4;
      }
      parallel_memcpy(pieces);                      // (with pageable partial SAs their pieces are in the same list)
      pieces.clear();
      PSG_HIP(hipMemcpyAsync(dmbv[s].p, pin_mbv[s], (size_t)moff, hipMemcpyHostToDevice, up));
    }
    if (!direct) {
      parallel_memcpy(pieces);
      PSG_HIP(hipMemcpyAsync(din[s].p, pin_in[s], (size_t)off, hipMemcpyHostToDevice, up));
      if (any_hi) PSG_HIP(hipMemcpyAsync(din[s].as<char>() + lo_cap, pin_in[s] + lo_cap, (size_t)(offh - lo_cap), hipMemcpyHostToDevice, up));
    }
    PSG_HIP(hipEventRecord(ev_up[s], up));
    st.stage_ms += wall_ms() - t0;
    // ---- merge slice k behind its copy, check it, send it back
    PSG_HIP(hipStreamWaitEvent(stream(), ev_up[s], 0));
    PSG_HIP(hipMemcpyAsync(dlv[s].p, pin_lv[s], sizeof(MergeLevel) * (size_t)H, hipMemcpyHostToDevice, stream()));
    PSG_HIP(hipEventRecord(ev_k0[s], stream()));
    if ((rc = merge_launch(H, dlv[s].as<MergeLevel>(), pin_lv[s][0], pin_lv[s][H > 1 ? 1 : 0], any_hi, x0, cnt, dout[s].as<u8>()))) { result = rc; break; }
    PSG_HIP(hipEventRecord(ev_k1[s], stream()));
    if (check && (rc = psg::check_sa5_accumulate(check->d_text, check->n, dout[s].as<u8>(), cnt, check->samples_per_slice, check->seed + (u64)k, acc.as<unsigned long long>()))) { result = rc; break; }
    if (sink) { PSG_HIP(hipMemcpyAsync(pin_out[s], dout[s].p, (size_t)(5 * cnt), hipMemcpyDeviceToHost, stream())); st.d2h_bytes += 5 * cnt; }
    PSG_HIP(hipEventRecord(ev_dn[s], stream()));
  }
  (void)hipStreamSynchronize(up);
  PSG_HIP(psg::sync_stream());
  if (result) return result;
  if (check) {
    u64 h2[3];
    if ((rc = psg::copy_d2h(h2, acc.p, 24))) return rc;
    check->sum = h2[0]; check->bad_pairs = (i64)h2[1]; check->undecided_pairs = (i64)h2[2];
  }
  st.total_ms = wall_ms() - w0;
  note_kernel_ms(st.kernel_ms);
  if (stats) *stats = st;
  return 0;
}
