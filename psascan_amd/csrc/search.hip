// search.hip -- K8: exact ranks of tail suffixes among the suffixes of a block by string search over the block's
// partial suffix arrays.  Semantics of em_compute_initial_ranks (em_compute_initial_ranks.hpp:222-319): for a text
// position p, |{s in block : text[s..n) < text[p..n)}|; a comparison that reaches the end of the block is decided
// by the gt bit of the position the pattern has reached (lcp_compare, em_compute_initial_ranks.hpp:54-76).
//
// GPU form: one wavefront per position.  The 64 lanes compare 512 symbols per step (8 per lane, unaligned 8-byte
// loads, ballot for the first mismatch); the binary search keeps the common prefix lengths with both bounds
// (Manber-Myers), so the symbols of a long shared prefix are compared about once per position and not once per
// probe.  This replaces the serial hand-over rounds of the chain scheme for every chain whose warm-up interval
// does not close (text with repeats longer than the warm-up).
#include "dev_common.hpp"

#include <algorithm>
#include <vector>

using namespace psg;

struct SearchPart { i64 beg, size; const u32 *lo; const u8 *hi; };
struct SearchParams {
  const u8 *text;
  i64 n, cmp_end;
  const u32 *gt;            // bit (n - j) = [text[j..n) > text[cmp_end..n)], j in (cmp_end, n]; null when cmp_end == n
  SearchPart part[2];
  int nparts;
  const i64 *pos;
  i64 npos;
  i64 *rank;
  i64 text_end;             // text[.. text_end) is readable (= n when the whole text is on the device)
  int *fail;                // set when a comparison would have to read at or behind text_end
  const u8 *text2;          // the pattern side reads here (a second window, addressed by absolute position like `text`); = text with one window
  i64 text2_end;
};

__device__ __forceinline__ u64 load8_unaligned(const u8 *p) {
  u64 v;
  __builtin_memcpy(&v, p, 8);
  return v;
}

// [text[s..n) < text[p..n)] for s < cmp_end <= ... and p > s, comparing from offset k on (the first k symbols are
// known to be equal).  lcp_out: symbols found equal (a lower bound of the true lcp when a gt bit decides).
// All lanes of the wave call this with the same arguments and get the same result.
__device__ __forceinline__ bool suffix_less_wave(const SearchParams &P, i64 s, i64 p, i64 k, i64 &lcp_out) {
  const int lane = (int)lane_id();
  // the known common prefix may reach beyond the end of the block for THIS suffix (it was measured on suffixes that
  // start further left): the decision point is the block end, where the gt bit of position p + (cmp_end - s) applies
  k = std::min(k, P.cmp_end - s);
  for (;;) {
    const i64 rem_p = P.n - (p + k), rem_s = P.cmp_end - (s + k);
    if (rem_p <= 0) { lcp_out = k; return false; }          // the pattern suffix is a proper prefix: it is the smaller one
    if (rem_s <= 0) {                                       // reached the end of the block: text[cmp_end..) against text[p+k..)
      const i64 idx = P.n - (p + k);
      const bool g = P.gt ? ((gload(P.gt + (idx >> 5)) >> (idx & 31)) & 1u) : false;
      lcp_out = k;
      return g;
    }
    i64 chunk = std::min<i64>(512, std::min(rem_p, rem_s));
    const i64 avail = std::min(P.text2_end - (p + k), P.text_end - (s + k));   // either side may reach the end of its window
    if (avail < chunk) {
      if (avail <= 0) { if (lane == 0) *P.fail = 1; lcp_out = k; return false; }
      chunk = avail;
    }
    const i64 off = (i64)lane * 8;
    u64 a = 0, b = 0;
    if (off + 8 <= chunk) { a = load8_unaligned(P.text + s + k + off); b = load8_unaligned(P.text2 + p + k + off); }
    else if (off < chunk) {
      for (int q = 0; q < (int)(chunk - off); ++q) { a |= (u64)P.text[s + k + off + q] << (8 * q); b |= (u64)P.text2[p + k + off + q] << (8 * q); }
    }
    const u64 x = a ^ b;
    const u64 mism = __ballot(x != 0);
    if (mism) {
      const int f = __ffsll((long long)mism) - 1;
      const u64 xa = __shfl(a, f, 64), xb = __shfl(b, f, 64);
      const u64 xx = xa ^ xb;
      const int byte = (__ffsll((long long)xx) - 1) >> 3;
      lcp_out = k + (i64)f * 8 + byte;
      return ((xa >> (8 * byte)) & 255u) < ((xb >> (8 * byte)) & 255u);
    }
    k += chunk;
  }
}

// one wave per position; 4 positions per workgroup
__global__ __launch_bounds__(PSG_WG) void search_kernel(SearchParams P) {
  const i64 w = (i64)blockIdx.x * (PSG_WG / 64) + (threadIdx.x >> 6);
  if (w >= P.npos) return;
  const i64 p = P.pos[w];
  i64 total = 0;
  if (p < P.n) {
    for (int t = 0; t < P.nparts; ++t) {
      const SearchPart S = P.part[t];
      i64 lo = 0, hi = S.size, llcp = 0, rlcp = 0;     // suffixes [0, lo) are smaller than the pattern, [hi, size) are not
      while (lo < hi) {
        const i64 md = lo + ((hi - lo) >> 1);
        const i64 s = S.beg + (i64)gload(S.lo + md) + (S.hi ? (i64)gload(S.hi + md) << 32 : 0);
        i64 l;
        const bool less = suffix_less_wave(P, s, p, std::min(llcp, rlcp), l);
        if (less) { lo = md + 1; llcp = l; } else { hi = md; rlcp = l; }
      }
      total += lo;
    }
  }
  if (lane_id() == 0) P.rank[w] = total;
}

namespace psg {
// one device word, set by a search that ran into the end of the text window
static int *search_fail_flag() {
  static int *flag = nullptr;
  if (!flag && hipMalloc((void **)&flag, 64) != hipSuccess) { (void)hipGetLastError(); flag = nullptr; }
  return flag;
}
// after the search has run (any stream order behind it): did a comparison leave the text window?
int search_window_check() {
  int *flag = search_fail_flag();
  int h = 0;
  if (!flag) return 0;
  if (int rc = copy_d2h(&h, flag, 4)) return rc;
  if (h) { set_error("search: a comparison ran past the end of the text window (repeats longer than the window's look-ahead)"); return PSG_EWINDOW; }
  return 0;
}
// enqueue the search for npos device-resident positions; ranks land in d_rank
int search_ranks_launch(const psg_search_ctx *sc, const i64 *d_pos, i64 npos, i64 *d_rank) {
  PSG_REQUIRE(sc && sc->d_text && sc->n > 0 && sc->cmp_end > 0 && sc->cmp_end <= sc->n && sc->nparts >= 1 && sc->nparts <= 2,
              "search context: text, comparison end and 1-2 parts required");
  PSG_REQUIRE(sc->cmp_end == sc->n || sc->d_gt_cmp_end, "search context: gt bits w.r.t. the comparison end required");
  SearchParams P{};
  P.text = sc->d_text; P.n = sc->n; P.cmp_end = sc->cmp_end; P.gt = sc->cmp_end == sc->n ? nullptr : sc->d_gt_cmp_end;
  P.nparts = sc->nparts;
  for (int t = 0; t < sc->nparts; ++t) {
    PSG_REQUIRE(sc->part[t].d_psa_lo && sc->part[t].size >= 1 && sc->part[t].beg >= 0 && sc->part[t].beg + sc->part[t].size <= sc->cmp_end,
                "search context: bad part");
    P.part[t] = SearchPart{sc->part[t].beg, sc->part[t].size, sc->part[t].d_psa_lo, sc->part[t].d_psa_hi};
  }
  P.pos = d_pos; P.npos = npos; P.rank = d_rank;
  const bool windowed = sc->text_end > 0;
  PSG_REQUIRE(!windowed || (sc->text_begin >= 0 && sc->text_begin <= sc->text_end && sc->text_end <= sc->n), "search context: bad text window");
  for (int t = 0; windowed && t < sc->nparts; ++t)
    PSG_REQUIRE(sc->part[t].beg >= sc->text_begin && sc->part[t].beg + sc->part[t].size <= sc->text_end, "search context: a part lies outside the text window");
  P.text_end = windowed ? sc->text_end : sc->n;
  P.text2 = P.text; P.text2_end = P.text_end;
  if (sc->d_text2) {
    PSG_REQUIRE(windowed && sc->text2_begin >= 0 && sc->text2_begin <= sc->text2_end && sc->text2_end <= sc->n, "search context: bad second text window");
    P.text2 = sc->d_text2; P.text2_end = sc->text2_end;
  }
  P.fail = search_fail_flag();
  if (!P.fail) { set_error("search: flag allocation failed"); return PSG_ENOMEM; }
  if (npos == 0) return 0;
  PSG_HIP(hipMemsetAsync(P.fail, 0, 4, stream()));
  hipLaunchKernelGGL(search_kernel, dim3((unsigned)cdiv(npos, PSG_WG / 64)), dim3(PSG_WG), 0, stream(), P);
  PSG_HIP(hipGetLastError());
  return 0;
}
}  // namespace psg

extern "C" int psg_initial_ranks(const psg_search_ctx *sc, const int64_t *h_positions, int64_t count, int64_t *h_ranks) {
  PSG_REQUIRE(h_positions && h_ranks && count >= 0, "psg_initial_ranks");
  if (count == 0) return 0;
  for (i64 k = 0; k < count; ++k) PSG_REQUIRE(h_positions[k] >= 0 && h_positions[k] <= sc->n, "psg_initial_ranks: position out of range");
  if (sc->d_text2) for (i64 k = 0; k < count; ++k) PSG_REQUIRE(h_positions[k] >= sc->text2_begin, "psg_initial_ranks: a searched position lies in front of the second text window");
  DevBuf pos, rk;
  int rc;
  if ((rc = pos.alloc(count * 8)) || (rc = rk.alloc(count * 8))) return rc;
  if ((rc = psg::copy_h2d(pos.p, h_positions, (size_t)count * 8))) return rc;
  EventTimer tm; tm.start();
  if ((rc = psg::search_ranks_launch(sc, pos.as<i64>(), count, rk.as<i64>()))) return rc;
  tm.stop();
  if ((rc = psg::copy_d2h(h_ranks, rk.p, (size_t)count * 8))) return rc;
  if (sc->text_end > 0 && (rc = psg::search_window_check())) return rc;
  note_kernel_ms(tm.ms());
  return 0;
}

// =======================================================================================
// BWT, i0 and gt_begin of a range from its partial suffix array (inmem_bwt_from_sa.hpp:47-83;
// gt_begin: compute_initial_gt_bitvectors.hpp -- here simply "ranked after the range's first suffix")
// =======================================================================================
// (grid-stride: a launch holds fewer than 2^32 threads, a range may hold more positions)
__global__ __launch_bounds__(PSG_WG) void psa_find_i0_kernel(const u32 *psa, const u8 *psa_hi, i64 size, i64 *i0) {
  for (i64 k = (i64)blockIdx.x * PSG_WG + threadIdx.x; k < size; k += (i64)gridDim.x * PSG_WG)
    if (psa[k] == 0 && (!psa_hi || psa_hi[k] == 0)) *i0 = k;
}
// bwt[k] = text[beg + psa[k] - 1] (dummy 0 at i0); gt bit u = size - s for every suffix s ranked after suffix 0
__global__ __launch_bounds__(PSG_WG) void psa_bwt_gt_kernel(const u8 *text, i64 beg, i64 size, const u32 *psa, const u8 *psa_hi, const i64 *i0p, u8 *bwt, u32 *gt) {
  const i64 i0 = *i0p;
  for (i64 k = (i64)blockIdx.x * PSG_WG + threadIdx.x; k < size; k += (i64)gridDim.x * PSG_WG) {
    const i64 s = (i64)psa[k] + (psa_hi ? (i64)psa_hi[k] << 32 : 0);
    bwt[k] = s ? text[beg + s - 1] : 0;
    if (gt && s && k > i0) { const i64 u = size - s; atomicOr(&gt[u >> 5], 1u << (u & 31)); }
  }
}

extern "C" int psg_halfblock_from_psa(const psg_search_ctx *sc, int64_t beg, int64_t size, const uint32_t *d_psa, uint8_t *d_bwt,
                                      int64_t *i0, uint32_t *d_gt_begin) {
  PSG_REQUIRE(size < 0x100000000ll, "psg_halfblock_from_psa: a range of 2^32 positions or more needs psg_halfblock_from_psa40");
  return psg_halfblock_from_psa40(sc, beg, size, d_psa, nullptr, d_bwt, i0, d_gt_begin);
}

extern "C" int psg_halfblock_from_psa40(const psg_search_ctx *sc, int64_t beg, int64_t size, const uint32_t *d_psa, const uint8_t *d_psa_hi,
                                        uint8_t *d_bwt, int64_t *i0, uint32_t *d_gt_begin) {
  PSG_REQUIRE(sc && sc->d_text && d_psa && d_bwt && i0 && size >= 1 && beg >= 0 && beg + size <= sc->n && (d_psa_hi || size < 0x100000000ll), "psg_halfblock_from_psa40");
  DevBuf misc;
  if (int rc = misc.alloc(16)) return rc;
  PSG_HIP(hipMemsetAsync(misc.p, 0xFF, 16, stream()));
  const unsigned grid = (unsigned)std::min<i64>(cdiv(size, PSG_WG), 1 << 22);
  hipLaunchKernelGGL(psa_find_i0_kernel, dim3(grid), dim3(PSG_WG), 0, stream(), d_psa, d_psa_hi, size, misc.as<i64>());
  if (d_gt_begin) PSG_HIP(hipMemsetAsync(d_gt_begin, 0, (size_t)(((size + 31) >> 5) * 4), stream()));
  hipLaunchKernelGGL(psa_bwt_gt_kernel, dim3(grid), dim3(PSG_WG), 0, stream(), sc->d_text, beg, size, d_psa, d_psa_hi, misc.as<i64>(), d_bwt, d_gt_begin);
  PSG_HIP(hipGetLastError());
  i64 h = -1;
  if (int rc = psg::copy_d2h(&h, misc.p, 8)) return rc;
  if (h < 0 || h >= size) { set_error("psg_halfblock_from_psa: the partial suffix array does not contain the range's first suffix"); return PSG_ECHECK; }
  *i0 = h;
  // bit u = 0 (position beg + size): [text[end..) > text[beg..)] = the first suffix is NOT smaller... decided by search: rank of `end` among {beg}
  if (d_gt_begin && beg + size < sc->n) {
    psg_search_ctx one = *sc;
    const u32 zero = 0;
    DevBuf z;
    if (int rc = z.alloc(16)) return rc;
    PSG_HIP(hipMemsetAsync(z.p, 0, 16, stream()));
    (void)zero;
    one.nparts = 1; one.part[0].beg = beg; one.part[0].size = 1; one.part[0].d_psa_lo = z.as<u32>(); one.part[0].d_psa_hi = nullptr;
    const int64_t pos = beg + size;
    int64_t r = 0;
    if (int rc = psg_initial_ranks(&one, &pos, 1, &r)) return rc;     // r = [text[beg..) < text[end..)]
    if (r == 1) { u32 w; if (int rc = psg::copy_d2h(&w, d_gt_begin, 4)) return rc; w |= 1u; if (int rc = psg::copy_h2d(d_gt_begin, &w, 4)) return rc; }
  }
  return 0;
}
