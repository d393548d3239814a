// leafsort.hpp -- suffix sort of one small LEAF [beg,end) of the text on a host core, ordered as suffixes of the WHOLE
// text (comparisons read on past `end`).  This is the host half of the reference's in-memory pSAscan
// (inmem_psascan_src/initial_partial_sufsort.hpp:61-319: max_threads sub-blocks, each sorted by libdivsufsort /
// libsais on a renamed copy); the merging half runs on the device (psg_merge_leaves).  A leaf hands over ONLY its
// partial suffix array, as 16-bit positions relative to `beg` (leaves of at most 65 536 symbols) -- BWT, i0 and gt
// bits are derived on the device.
//
// Method (leaves stay inside the core's L1/L2): every suffix becomes one 64-bit word (its first 48/b symbols as b-bit
// codes of the leaf's alphabet | position), LSD radix sort on the 48 key bits with 12-bit digits, then groups of equal prefixes are refined
// with the NEXT 8 symbols as an integer key (small groups: insertion sort on those keys), deeper and deeper.  A
// comparison that exceeds `cap` symbols (periodic text) makes the sort give up (false): the caller sorts that range
// another way, exactly like the look-ahead sorter of halfblock.hpp.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

namespace psa_host {

struct LeafScratch {                      // per worker thread, reused from leaf to leaf
  std::vector<uint64_t> a, b;             // key words
  std::vector<uint64_t> k2;               // refinement keys of a group
  std::vector<uint32_t> stack;            // (lo, hi, depth) triples
  uint32_t cnt[4][1 << 12];
};

// big-endian integer of the next 8 symbols at absolute position p; a suffix that ends inside them is padded with
// zeros and `ran_off` is set (the group that holds it is finished by plain comparison: the shorter suffix is smaller)
static inline uint64_t leaf_key8(const uint8_t *text, int64_t n, int64_t p, bool &ran_off) {
  uint64_t v = 0;
  if (p + 8 <= n) { std::memcpy(&v, text + p, 8); return __builtin_bswap64(v); }
  ran_off = true;
  for (int t = 0; t < 8; ++t) v = (v << 8) | (p + t < n ? text[p + t] : 0);
  return v;
}

// text[x..n) < text[y..n) for two different positions whose first `known` symbols agree; false + give_up on budget
static inline bool leaf_less(const uint8_t *text, int64_t n, int64_t x, int64_t y, int64_t known, int64_t cap, bool &give_up) {
  int64_t k = known;
  while (x + k + 8 <= n && y + k + 8 <= n) {
    uint64_t p, q;
    std::memcpy(&p, text + x + k, 8); std::memcpy(&q, text + y + k, 8);
    if (p != q) return __builtin_bswap64(p) < __builtin_bswap64(q);
    k += 8;
    if (k - known > cap) { give_up = true; return false; }
  }
  for (;; ++k) {
    if (x + k >= n) return true;          // x ran off the text first (x != y): the shorter suffix is the smaller one
    if (y + k >= n) return false;
    if (text[x + k] != text[y + k]) return text[x + k] < text[y + k];
    if (k - known > cap) { give_up = true; return false; }
  }
}

// out[r] = position (relative to beg) of the r-th smallest suffix of the leaf; size = end - beg <= 65536.
// Returns false when a comparison exceeded `cap` symbols or the refinement work its budget.
static inline bool sort_leaf16(const uint8_t *text, int64_t n, int64_t beg, int64_t end, uint16_t *out, int64_t cap, LeafScratch &S) {
  const int64_t m = end - beg;
  if (m <= 0 || m > 65536) return false;
  if (m == 1) { out[0] = 0; return true; }
  S.a.resize((size_t)m); S.b.resize((size_t)m);
  uint64_t *a = S.a.data(), *b = S.b.data();
#ifdef LEAF_PROFILE
  double tp = now();
#define LEAF_TICK(k) do { const double t_ = now(); g_t[k] += t_ - tp; tp = t_; } while (0)
#else
#define LEAF_TICK(k) do { } while (0)
#endif
  // ---- key words: the first 48 / bits symbols as `bits`-wide codes (order-preserving renaming of the symbols that
  // occur in the leaf and the look-ahead a key can reach) in the top 48 bits, rolled in from the right
  int bits = 8, known0 = 6;
#ifndef LEAF_DB
#define LEAF_DB 12
#endif
  constexpr int DB = LEAF_DB, DM = (1 << DB) - 1, S0 = 64 - 4 * DB;
  static const int npass_env = getenv("LEAF_PASSES") ? atoi(getenv("LEAF_PASSES")) : 0;
  int npass = 4;
  {
    const int64_t reach = std::min<int64_t>(n, end + 48);
    uint32_t hist[256] = {0};
    for (int64_t i = beg; i < reach; ++i) ++hist[text[i]];
    uint8_t code[256];
    int sigma = 0;
    for (int c = 0; c < 256; ++c) { code[c] = (uint8_t)sigma; if (hist[c]) ++sigma; }
    bits = 1;
    while ((1 << bits) < sigma) ++bits;
    // radix passes: a text that fills its alphabet evenly (random bytes, DNA) is told apart by the top 24 key bits and
    // two passes; natural language carries ~2 bits per symbol in context and needs all four (measured: 75 against 53
    // MB/s per core on uniform bytes, 55 against 33 on English-like text)
    {
      uint64_t sq = 0;
      for (int c = 0; c < 256; ++c) sq += (uint64_t)hist[c] * hist[c];
      const double coll = (double)sq / ((double)(reach - beg) * (double)(reach - beg));   // probability that two symbols agree
      npass = coll * (double)(1 << bits) < 1.5 ? 2 : 4;
      if (npass_env >= 1 && npass_env <= 4) npass = npass_env;
    }
    known0 = (DB * npass) / bits;
    const int used = known0 * bits;
    const uint64_t keep = ~((1ull << (64 - used)) - 1ull);
    uint64_t k = 0;
    const int64_t top = std::min<int64_t>(n, end + known0 - 1);
    for (int64_t i = top - 1; i >= end; --i) k = (k >> bits) | ((uint64_t)code[text[i]] << (64 - bits));
    std::memset(S.cnt, 0, sizeof S.cnt);                      // (all four digit histograms in the same sweep)
    for (int64_t i = end - 1; i >= beg; --i) {
      k = (k >> bits) | ((uint64_t)code[text[i]] << (64 - bits));
      const uint64_t v = k & keep;
      a[i - beg] = v | (uint64_t)(i - beg);
      ++S.cnt[0][(v >> S0) & DM]; ++S.cnt[1][(v >> (S0 + DB)) & DM]; ++S.cnt[2][(v >> (S0 + 2 * DB)) & DM]; ++S.cnt[3][(v >> (S0 + 3 * DB)) & DM];
    }
  }
  LEAF_TICK(0);
  // ---- LSD radix on bits 16..63, 12 bits per pass
  LEAF_TICK(1);
  for (int d = 4 - npass; d < 4; ++d) {
    uint32_t *c = S.cnt[d];
    bool one = false;
    uint32_t run = 0;
    for (int k = 0; k <= DM; ++k) { const uint32_t x = c[k]; if (x == (uint32_t)m) one = true; c[k] = run; run += x; }
    if (one) continue;                      // the digit is the same everywhere
    const int sh = S0 + DB * d;
    for (int64_t i = 0; i < m; ++i) { const uint64_t v = a[i]; b[c[(v >> sh) & DM]++] = v; }
    std::swap(a, b);
  }
  LEAF_TICK(2);
  // ---- groups of equal prefixes, refined 8 symbols at a time
  int64_t budget = 64 * m + (1 << 14);      // key fetches + comparisons before the sort gives up (periodic text)
  bool give_up = false;
  const int GS = 64 - known0 * bits;            // groups = equal sorted key bits = known0 whole symbols
  const bool near_end = end + known0 + 8 >= n;   // zero-padded keys may tie with real symbols only this close to the end of the text
  auto pos_of = [&](uint64_t v) { return beg + (int64_t)(v & 0xFFFF); };
  auto finish_by_compare = [&](int64_t lo, int64_t hi, int64_t depth) {
    std::sort(a + lo, a + hi, [&](uint64_t x, uint64_t y) { return leaf_less(text, n, pos_of(x), pos_of(y), depth, cap, give_up); });
  };
  std::vector<uint32_t> &st = S.stack;
  st.clear();
  for (int64_t g0 = 0; g0 < m && !give_up;) {
    int64_t g1 = g0 + 1;
    const uint64_t kk = a[g0] >> GS;
    while (g1 < m && (a[g1] >> GS) == kk) ++g1;
    if (g1 - g0 > 1) {
      if (near_end) finish_by_compare(g0, g1, 0);
      else { st.push_back((uint32_t)g0); st.push_back((uint32_t)g1); st.push_back((uint32_t)known0); }
    }
    while (!st.empty() && !give_up) {
      const int64_t depth = st.back(); st.pop_back();
      const int64_t hi = st.back(); st.pop_back();
      const int64_t lo = st.back(); st.pop_back();
      const int64_t cnt = hi - lo;
      if (depth > cap || (budget -= 4 * cnt) < 0) { give_up = true; break; }
      if (cnt == 2) {                       // the most frequent group size
        if (leaf_less(text, n, pos_of(a[lo + 1]), pos_of(a[lo]), depth, cap, give_up)) std::swap(a[lo], a[lo + 1]);
        continue;
      }
      S.k2.resize((size_t)cnt);
      uint64_t *k2 = S.k2.data();
      bool ran_off = false;
      for (int64_t i = 0; i < cnt; ++i) k2[i] = leaf_key8(text, n, pos_of(a[lo + i]) + depth, ran_off);
      if (ran_off) { finish_by_compare(lo, hi, depth); continue; }
      if (cnt <= 24) {                      // insertion sort on (key, word) pairs
        for (int64_t i = 1; i < cnt; ++i) {
          const uint64_t kv = k2[i], wv = a[lo + i];
          int64_t j = i;
          while (j > 0 && k2[j - 1] > kv) { k2[j] = k2[j - 1]; a[lo + j] = a[lo + j - 1]; --j; }
          k2[j] = kv; a[lo + j] = wv;
        }
      } else {                              // larger group: sort indices by key
        uint64_t *tmp = b;                  // (b is free: the radix passes are done) pairs packed as key-order permutation
        for (int64_t i = 0; i < cnt; ++i) tmp[i] = (uint64_t)i;
        std::sort(tmp, tmp + cnt, [&](uint64_t x, uint64_t y) { return k2[x] < k2[y]; });
        // apply the permutation to a[lo..hi) and k2 (through the other half of b)
        uint64_t *wa = b + cnt, *wk = nullptr;
        if (2 * cnt <= m) {
          for (int64_t i = 0; i < cnt; ++i) wa[i] = a[lo + tmp[i]];
          std::memcpy(a + lo, wa, (size_t)cnt * 8);
          for (int64_t i = 0; i < cnt; ++i) wa[i] = k2[tmp[i]];
          std::memcpy(k2, wa, (size_t)cnt * 8);
        } else {                            // a group larger than half the leaf: separate buffer
          std::vector<uint64_t> w2((size_t)cnt);
          wk = w2.data();
          for (int64_t i = 0; i < cnt; ++i) wk[i] = a[lo + tmp[i]];
          std::memcpy(a + lo, wk, (size_t)cnt * 8);
          for (int64_t i = 0; i < cnt; ++i) wk[i] = k2[tmp[i]];
          std::memcpy(k2, wk, (size_t)cnt * 8);
        }
      }
      for (int64_t r0 = 0; r0 < cnt;) {     // ties go one level deeper
        int64_t r1 = r0 + 1;
        while (r1 < cnt && k2[r1] == k2[r0]) ++r1;
        if (r1 - r0 > 1) { st.push_back((uint32_t)(lo + r0)); st.push_back((uint32_t)(lo + r1)); st.push_back((uint32_t)(depth + 8)); }
        r0 = r1;
      }
    }
    g0 = g1;
  }
  LEAF_TICK(3);
  if (give_up) return false;
  for (int64_t i = 0; i < m; ++i) out[i] = (uint16_t)(a[i] & 0xFFFF);
  LEAF_TICK(4);
  return true;
}

}  // namespace psa_host
