// host_abi.cpp -- C entry points of the host-side pieces (sorter, start-rank search) so the
// CPU test-suite can exercise them without a GPU.  No HIP dependency.
#include <cstdint>
#include <cstring>

#include "halfblock.hpp"

extern "C" {

int psh_suffix_array(const uint8_t *text, int64_t n, int64_t *sa) {
  psa_host::Sais<int64_t>::run(text, sa, n, (int64_t)256);
  return 0;
}

// gt_tail_bits: bit v (LSB-first bytes), v in [1, end-beg], = [text[end+v..) > text[end..)]
int psh_sort_halfblock(const uint8_t *text, int64_t n, int64_t beg, int64_t end, const uint8_t *gt_tail_bits,
                       uint32_t *psa_lo, uint8_t *bwt, int64_t *i0, uint32_t *gt_begin) {
  try {
    psa_host::HalfBlock hb;
    auto gt = [&](int64_t v) { return (bool)((gt_tail_bits[v >> 3] >> (v & 7)) & 1); };
    psa_host::sort_halfblock(text, n, beg, end, gt, hb);
    if (!hb.psa_hi.empty()) return -2;
    memcpy(psa_lo, hb.psa_lo.data(), 4 * (size_t)hb.size);
    memcpy(bwt, hb.bwt.data(), (size_t)hb.size);
    memcpy(gt_begin, hb.gt_begin.data(), 4 * (size_t)((hb.size + 31) / 32));
    *i0 = hb.i0;
    return 0;
  } catch (const std::exception &) { return -1; }
}

// look-ahead sorters of construct_sa: order by reading on in the text past `end` (no gt bits needed).
// method 0: SA-IS + rename with direct comparisons; 1: prefix-key radix sort.  Returns 1 when the method gives up
// (comparisons longer than cap / large groups of equal prefixes), -1 on bad input.
int psh_sort_halfblock_ahead(const uint8_t *text, int64_t n, int64_t beg, int64_t end, int method, int64_t cap,
                             uint32_t *psa_lo, uint8_t *bwt, int64_t *i0, uint32_t *gt_begin) {
  try {
    psa_host::HalfBlock hb;
    if (method == 1) {
      if (!psa_host::sort_halfblock_radix(text, n, beg, end, hb, cap)) return 1;
    } else {
      try { psa_host::sort_halfblock(text, n, beg, end, psa_host::gt_tail_direct(text, n, end, cap), hb, cap); }
      catch (const psa_host::GtCapExceeded &) { return 1; }
    }
    if (!hb.psa_hi.empty()) return -2;
    memcpy(psa_lo, hb.psa_lo.data(), 4 * (size_t)hb.size);
    memcpy(bwt, hb.bwt.data(), (size_t)hb.size);
    memcpy(gt_begin, hb.gt_begin.data(), 4 * (size_t)((hb.size + 31) / 32));
    *i0 = hb.i0;
    return 0;
  } catch (const std::exception &) { return -1; }
}

int64_t psh_rank_by_search(const uint8_t *text, int64_t n, int64_t beg, int64_t size, const uint32_t *psa_lo, int64_t p) {
  psa_host::HalfBlock hb;
  hb.beg = beg; hb.size = size;
  hb.psa_lo.assign(psa_lo, psa_lo + size);
  return psa_host::rank_by_search(text, n, hb, p);
}
}
