// halfblock.hpp -- host-side suffix sort of one half-block [beg,end) of the text, ordered as
// suffixes of the WHOLE text.  Produces what the reference's in-memory pSAscan hands to
// process_block (inmem_psascan_src/inmem_psascan.hpp:64-304 as used at
// partial_sufsort.hpp:166,286): partial SA, BWT (dummy 0 at i0, inmem_bwt_from_sa.hpp:51-54),
// i0 and the gt_begin bits.  north_star keeps this step on host cores.
//
// Ties that run past `end` are decided the way the reference does it: the block is renamed with
// the gt bits of its positions w.r.t. `end` (idea of initial_partial_sufsort.hpp:61-80: symbols
// above the block's last symbol, and occurrences of the last symbol followed by a suffix greater
// than text[end..), move up by one), after which a plain suffix sorter yields the true order.
// This is why byte 255 is not allowed in the input (same restriction as the reference,
// initial_partial_sufsort.hpp:141-146).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <functional>
#include <stdexcept>
#include <vector>

#include "sais.hpp"

namespace psa_host {

// resize() without zero-filling: a partial SA of 8 GiB is written once, in full, by the sorter or by a download from
// the device -- value-initialising it first touches every page on one thread (1.3 s per 8 GiB)
template <class T> struct default_init_alloc : std::allocator<T> {
  template <class U> struct rebind { using other = default_init_alloc<U>; };
  template <class U, class... A> void construct(U *p, A &&...a) {
    if constexpr (sizeof...(A) == 0) ::new ((void *)p) U; else ::new ((void *)p) U(std::forward<A>(a)...);
  }
};
typedef std::vector<uint32_t, default_init_alloc<uint32_t>> PsaVec;
typedef std::vector<uint8_t, default_init_alloc<uint8_t>> PsaHiVec;

struct HalfBlock {
  int64_t beg = 0, size = 0, i0 = 0;
  PsaVec psa_lo;
  PsaHiVec psa_hi;                 // only when size > 2^32
  std::vector<uint8_t> bwt;
  std::vector<uint32_t> gt_begin;  // bit u <-> position end-u, u in [0,size)
};

// gt_tail(v), v >= 1: [text[end+v..n) > text[end..n)]   (only asked for end+v < n)
typedef std::function<bool(int64_t)> GtTail;

// Thrown by the look-ahead sorter when a comparison runs longer than its cap: that half-block is then sorted in
// the sequential schedule, with the gt bits the streaming passes produce (highly repetitive text).
struct GtCapExceeded {};

// gt_tail by direct comparison in the text (the whole text is in host memory): needs nothing from other blocks,
// so half-blocks can be sorted ahead of the block schedule, on all host cores.  Bounded: throws GtCapExceeded
// after `cap` equal symbols.
static inline GtTail gt_tail_direct(const uint8_t *text, int64_t n, int64_t end, int64_t cap) {
  return [=](int64_t v) {
    const int64_t a = end + v, b = end;     // [text[a..n) > text[b..n)], a > b
    for (int64_t k = 0;; ++k) {
      if (a + k >= n) return false;         // text[a..n) is a proper prefix of text[b..n)
      if (k >= cap) throw GtCapExceeded();
      const uint8_t x = text[a + k], y = text[b + k];
      if (x != y) return x > y;
    }
  };
}

// [text[s..n) > text[e..n)] for beg <= s < e = end, using text up to `end` and gt_tail beyond.
// cap > 0: give up (GtCapExceeded) after that many equal symbols.
static inline bool gt_wrt_end(const uint8_t *text, int64_t n, int64_t s, int64_t e, const GtTail &gt_tail, int64_t cap = 0) {
  int64_t lim = e - s;  // symbols of the s-suffix that lie before e
  for (int64_t k = 0; k < lim; ++k) {
    if (e + k >= n) return true;  // text[e..n) is a proper prefix of text[s..n)
    if (cap > 0 && k >= cap) throw GtCapExceeded();
    uint8_t a = text[s + k], b = text[e + k];
    if (a != b) return a > b;
  }
  // text[s..e) == text[e..e+lim): now text[e..) against text[e+lim..)
  if (e + lim >= n) return true;
  return !gt_tail(lim);
}

// from the sorted positions: i0, BWT, gt_begin
static inline void finish_halfblock(const uint8_t *text, int64_t n, int64_t beg, int64_t end, bool gt_of_beg, HalfBlock &out) {
  const int64_t m = end - beg;
  auto pos = [&](int64_t k) { return (int64_t)out.psa_lo[(size_t)k] | (out.psa_hi.empty() ? 0 : (int64_t)out.psa_hi[(size_t)k] << 32); };
  for (int64_t k = 0; k < m; ++k) if (pos(k) == 0) { out.i0 = k; break; }
  for (int64_t k = 0; k < m; ++k) {
    int64_t s = pos(k);
    out.bwt[(size_t)k] = s ? text[beg + s - 1] : 0;
    if (s && k > out.i0) { int64_t u = m - s; out.gt_begin[(size_t)(u >> 5)] |= 1u << (u & 31); }
  }
  // bit u = 0: position end, [text[end..) > text[beg..)]
  if (end < n && !gt_of_beg) out.gt_begin[0] |= 1u;
}

// gt_wrt_end for EVERY position s of [beg, e) in O(e - beg): the longest common prefix of text[s..e) with
// text[e..) comes from the Z-function of text[e..e+m) (prefix matching), so periodic text costs no more than
// random text.  (The reference gets the same bits from string matching as well: compute_initial_gt_bitvectors.hpp.)
// bit (s - beg) of the result = [text[s..n) > text[e..n)].
static inline std::vector<uint8_t> gt_all_wrt_end(const uint8_t *text, int64_t n, int64_t beg, int64_t e, const GtTail &gt_tail) {
  const int64_t m = e - beg, lp = std::min<int64_t>(m, n - e);   // pattern P = text[e .. e+lp)
  std::vector<uint8_t> bits((size_t)(m + 7) / 8 + 1, 0);
  if (lp <= 0) {                                                   // e == n: every suffix is greater than the empty one
    for (int64_t i = 0; i < m; ++i) bits[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
    return bits;
  }
  const uint8_t *P = text + e, *T = text + beg;
  std::vector<uint32_t> z((size_t)lp, 0);
  z[0] = (uint32_t)lp;
  for (int64_t i = 1, l = 0, r = 0; i < lp; ++i) {
    int64_t k = i < r ? std::min<int64_t>(r - i, z[(size_t)(i - l)]) : 0;
    while (i + k < lp && P[k] == P[i + k]) ++k;
    z[(size_t)i] = (uint32_t)k;
    if (i + k > r) { l = i; r = i + k; }
  }
  for (int64_t i = 0, l = 0, r = 0; i < m; ++i) {                  // k = lcp(T[i..m), P)
    int64_t k = i < r ? std::min<int64_t>(r - i, z[(size_t)(i - l)]) : 0;
    while (i + k < m && k < lp && T[i + k] == P[k]) ++k;
    if (i + k > r) { l = i; r = i + k; }
    const int64_t lim = m - i;                                     // symbols of the suffix that lie before e
    bool g;
    if (k < lim) g = k >= lp ? true : T[i + k] > P[k];             // P exhausted: text[e..n) is a proper prefix
    else g = e + lim >= n ? true : !gt_tail(lim);                  // text[s..e) == text[e..e+lim): the tail decides
    if (g) bits[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
  }
  return bits;
}

static inline void sort_halfblock(const uint8_t *text, int64_t n, int64_t beg, int64_t end, const GtTail &gt_tail,
                                  HalfBlock &out, int64_t cap = 0) {
  const int64_t m = end - beg;
  out.beg = beg; out.size = m;
  std::vector<uint8_t> blk(text + beg, text + end);
  bool renamed = false;
  uint8_t last = blk[(size_t)m - 1];
  bool gt_of_beg = false;  // [text[beg..) > text[end..)]
  if (end < n) {
    // all gt bits at once, in linear time (Z-function), in both schedules.  The tail oracle is only asked for a
    // position whose whole in-block part matches the text behind `end`; in the look-ahead schedule (cap > 0) it is the
    // bounded direct comparison, which gives up (GtCapExceeded) on periodic text instead of asking for tail bits
    // that do not exist yet.  (Deciding every position by its own bounded comparison cost O(m * run length) on
    // text with long runs of varying length -- zero-padded images -- without ever tripping the cap.)
    const std::vector<uint8_t> gtb = gt_all_wrt_end(text, n, beg, end, gt_tail);
    auto gt_at = [&](int64_t i) { return (bool)((gtb[(size_t)(i >> 3)] >> (i & 7)) & 1); };   // [text[beg+i ..) > text[end..)]
    gt_of_beg = gt_at(0);
    renamed = true;
    for (int64_t i = 0; i + 1 < m; ++i) {
      uint8_t c = blk[(size_t)i];
      if (c > last || (c == last && gt_at(i + 1))) {
        if (c == 255) throw std::runtime_error("the input contains byte 255");
        blk[(size_t)i] = c + 1;
      }
    }
    if (last == 255) throw std::runtime_error("the input contains byte 255");
    blk[(size_t)m - 1] = last + 1;
  } else {
    for (int64_t i = 0; i < m; ++i) if (blk[(size_t)i] == 255) throw std::runtime_error("the input contains byte 255");
  }
  (void)renamed;
  out.psa_lo.resize((size_t)m);
  out.bwt.resize((size_t)m);
  out.gt_begin.assign((size_t)((m + 31) / 32 + 1), 0);
  if (m < (int64_t)1 << 31) {
    std::vector<int32_t> sa((size_t)m);
    Sais<int32_t>::run(blk.data(), sa.data(), (int32_t)m, 256);
    for (int64_t k = 0; k < m; ++k) out.psa_lo[(size_t)k] = (uint32_t)sa[(size_t)k];
  } else {
    std::vector<int64_t> sa((size_t)m);
    Sais<int64_t>::run(blk.data(), sa.data(), m, (int64_t)256);
    out.psa_hi.resize((size_t)m);
    for (int64_t k = 0; k < m; ++k) { out.psa_lo[(size_t)k] = (uint32_t)sa[(size_t)k]; out.psa_hi[(size_t)k] = (uint8_t)(sa[(size_t)k] >> 32); }
  }
  std::vector<uint8_t>().swap(blk);
  finish_halfblock(text, n, beg, end, gt_of_beg, out);
}

// Prefix-key sorter for the look-ahead path (order = suffixes of the WHOLE text, decided by reading on past
// `end`): pack the first 64/bits symbols of every suffix into a 64-bit key (bits = width of the block's alphabet
// + an end-of-text code 0), LSD radix sort the (key, position) pairs, then order the suffixes inside every group
// of equal keys by direct comparison.  3-4x faster than SA-IS on text without long repeats; gives up (returns
// false -> the caller runs SA-IS) when a comparison exceeds `cap` symbols or the equal-key groups are large.
static inline bool sort_halfblock_radix(const uint8_t *text, int64_t n, int64_t beg, int64_t end, HalfBlock &out, int64_t cap) {
  const int64_t m = end - beg;
  if (m < 2 || m > ((int64_t)1 << 28)) return false;   // 24 bytes of working memory per symbol and thread
  // alphabet over the block and the symbols a key can reach beyond it
  const int64_t reach = std::min<int64_t>(n, end + 64);
  bool present[256] = {false};
  for (int64_t i = beg; i < reach; ++i) present[text[i]] = true;
  uint8_t code[256];
  int sigma = 0;
  for (int c = 0; c < 256; ++c) { code[c] = 0; if (present[c]) code[c] = (uint8_t)(++sigma); }   // 0 = past the end of the text
  if (present[255]) throw std::runtime_error("the input contains byte 255");
  int bits = 1;
  while ((1 << bits) <= sigma) ++bits;
  const int per_key = 64 / bits, used = per_key * bits;
  const uint64_t keep = used == 64 ? ~0ull : ~((1ull << (64 - used)) - 1ull);
  // only the top `need` key bits are radix-sorted: enough to leave groups of a few suffixes, which a
  // comparison step finishes
  int lg = 0;
  while (((int64_t)1 << lg) < m) ++lg;
  // a key bit carries log2(sigma)/bits bits of information (the codes 1..sigma do not fill the 2^bits values)
  const double info = sigma >= 2 ? std::log2((double)sigma) / bits : 1.0;
  int need = std::min(used, (int)std::ceil((lg + 8) / info));
  if (need > 32 && (int)std::ceil((lg + 5) / info) <= 32) need = 32;   // the packed 32-bit path below; groups a little larger
  PsaVec idx((size_t)m);
  // work allowed in the group phase before SA-IS is the better tool (4 units per suffix and refinement round, 1 per
  // symbol comparison): natural language needs 3-4 rounds, random text none; periodic text runs out and gives up
  int64_t budget = 24 * m + (1 << 16);
  // text[beg+a ..n) < text[beg+b ..n) for two suffixes known to agree on their first `known` symbols
  auto less_from = [&](uint32_t a, uint32_t b, int known) {
    int64_t x = beg + a + known, y = beg + b + known;
    for (int64_t k = 0;; ++k) {
      if (x + k >= n) return y + k < n || a > b;    // ran off the text: the shorter suffix is smaller
      if (y + k >= n) return false;
      if (k >= cap || --budget < 0) throw GtCapExceeded();
      const uint8_t p = text[x + k], q = text[y + k];
      if (p != q) return p < q;
    }
  };
  const int DB = 11;   // LSD digits: 2048 output streams per pass stay within reach of the caches / TLB
  std::vector<uint32_t> cnt((size_t)1 << DB);
  try {
    if (need <= 32) {
      // (top 32 key bits, position) packed into one word: 16 bytes moved per element and pass
      std::vector<uint64_t> kv((size_t)m), kv2((size_t)m);
      {
        uint64_t k = 0;   // symbols i .. i+per_key-1 in the top `used` bits, rolled in from the right
        for (int64_t i = std::min<int64_t>(n, end + per_key) - 1; i >= beg; --i) {
          k = (((uint64_t)code[text[i]] << (64 - bits)) | (k >> bits)) & keep;
          if (i < end) kv[(size_t)(i - beg)] = (k & 0xFFFFFFFF00000000ull) | (uint64_t)(i - beg);
        }
      }
      for (int shift = 64 - need; shift < 64; shift += DB) {
        const uint64_t dmask = ((uint64_t)1 << std::min(DB, 64 - shift)) - 1;
        std::fill(cnt.begin(), cnt.end(), 0u);
        for (int64_t i = 0; i < m; ++i) ++cnt[(size_t)((kv[(size_t)i] >> shift) & dmask)];
        uint32_t run = 0;
        bool one_bucket = false;
        for (size_t b = 0; b < cnt.size(); ++b) { uint32_t c = cnt[b]; if (c == (uint32_t)m) one_bucket = true; cnt[b] = run; run += c; }
        if (one_bucket) continue;   // this digit is the same everywhere
        for (int64_t i = 0; i < m; ++i) { const uint64_t v = kv[(size_t)i]; kv2[(size_t)(cnt[(size_t)((v >> shift) & dmask)]++)] = v; }
        kv.swap(kv2);
      }
      std::vector<uint64_t>().swap(kv2);
      const int gs = 64 - need, known = need / bits;   // a group = equal sorted bits = `known` whole symbols (at least)
      // Groups of equal prefixes are refined with the NEXT per_key symbols as an integer key (read straight from the
      // text), group by group and deeper and deeper -- natural language leaves ~20 suffixes per 12-symbol prefix and
      // needs 3-4 such rounds; comparing suffix pairs symbol by symbol instead made this sorter slower than SA-IS there.
      // The key of a round is the next 8 text bytes as a big-endian integer: one unaligned load + byte swap (packing
      // 12 five-bit codes through a table cost more than the sort it fed).  Equal bytes <=> equal symbols, and byte
      // order is symbol order; a suffix that ends inside the 8 bytes is padded with zeros and flagged, and a sub-group
      // that holds such a suffix is finished by plain comparison (the shorter suffix is the smaller one).
      std::vector<std::pair<uint64_t, uint32_t>> tmp;
      bool ran_off = false;
      auto key_at = [&](int64_t p) -> uint64_t {
        uint64_t v = 0;
        if (p + 8 <= n) { std::memcpy(&v, text + p, 8); return __builtin_bswap64(v); }
        ran_off = true;
        for (int t = 0; t < 8; ++t) v = (v << 8) | (p + t < n ? text[p + t] : 0);
        return v;
      };
      // the same comparison as less_from, 8 bytes at a time while both suffixes have them
      auto less_fast = [&](uint32_t a, uint32_t b, int64_t known) {
        int64_t x = beg + a + known, y = beg + b + known, k = 0;
        while (x + k + 8 <= n && y + k + 8 <= n) {
          uint64_t p, q;
          std::memcpy(&p, text + x + k, 8); std::memcpy(&q, text + y + k, 8);
          if (p != q) return __builtin_bswap64(p) < __builtin_bswap64(q);
          k += 8;
          if (k >= cap || (budget -= 8) < 0) throw GtCapExceeded();
        }
        return less_from(a, b, (int)std::min<int64_t>(known + k, 1 << 30));
      };
      std::vector<std::pair<std::pair<int64_t, int64_t>, int64_t>> stack;   // ((lo, hi), depth): kv[lo..hi) share `depth` symbols
      for (int64_t g0 = 0; g0 < m;) {
        int64_t g1 = g0 + 1;
        while (g1 < m && (kv[(size_t)g1] >> gs) == (kv[(size_t)g0] >> gs)) ++g1;
        if (g1 - g0 > 1) stack.push_back({{g0, g1}, known});
        while (!stack.empty()) {
          const int64_t lo = stack.back().first.first, hi = stack.back().first.second, depth = stack.back().second;
          stack.pop_back();
          const int64_t cnt = hi - lo;
          if (cnt <= 12) {
            std::sort(kv.begin() + lo, kv.begin() + hi, [&](uint64_t a, uint64_t b) { return less_fast((uint32_t)a, (uint32_t)b, depth); });
            continue;
          }
          if (depth >= cap || (budget -= 4 * cnt) < 0) throw GtCapExceeded();
          tmp.resize((size_t)cnt);
          ran_off = false;
          for (int64_t i = 0; i < cnt; ++i) { const uint32_t p = (uint32_t)kv[(size_t)(lo + i)]; tmp[(size_t)i] = {key_at(beg + (int64_t)p + depth), p}; }
          const bool group_ran_off = ran_off;
          std::sort(tmp.begin(), tmp.end());
          for (int64_t i = 0; i < cnt; ++i) kv[(size_t)(lo + i)] = (kv[(size_t)lo] & 0xFFFFFFFF00000000ull) | tmp[(size_t)i].second;
          for (int64_t r0 = 0; r0 < cnt;) {
            int64_t r1 = r0 + 1;
            while (r1 < cnt && tmp[(size_t)r1].first == tmp[(size_t)r0].first) ++r1;
            if (r1 - r0 > 1) {
              bool near_end = false;
              if (group_ran_off) for (int64_t i = r0; i < r1 && !near_end; ++i) near_end = beg + (int64_t)tmp[(size_t)i].second + depth + 8 > n;
              if (near_end) std::sort(kv.begin() + lo + r0, kv.begin() + lo + r1, [&](uint64_t a, uint64_t b) { return less_from((uint32_t)a, (uint32_t)b, (int)std::min<int64_t>(depth, 1 << 30)); });
              else stack.push_back({{lo + r0, lo + r1}, depth + 8});
            }
            r0 = r1;
          }
        }
        g0 = g1;
      }
      for (int64_t i = 0; i < m; ++i) idx[(size_t)i] = (uint32_t)kv[(size_t)i];
    } else {
      std::vector<uint64_t> key((size_t)m), key2((size_t)m);
      PsaVec idx2((size_t)m);
      {
        uint64_t k = 0;
        for (int64_t i = std::min<int64_t>(n, end + per_key) - 1; i >= beg; --i) {
          k = (((uint64_t)code[text[i]] << (64 - bits)) | (k >> bits)) & keep;
          if (i < end) { key[(size_t)(i - beg)] = k; idx[(size_t)(i - beg)] = (uint32_t)(i - beg); }
        }
      }
      for (int shift = 64 - need; shift < 64; shift += DB) {
        const uint64_t dmask = ((uint64_t)1 << std::min(DB, 64 - shift)) - 1;
        std::fill(cnt.begin(), cnt.end(), 0u);
        for (int64_t i = 0; i < m; ++i) ++cnt[(size_t)((key[(size_t)i] >> shift) & dmask)];
        uint32_t run = 0;
        bool one_bucket = false;
        for (size_t b = 0; b < cnt.size(); ++b) { uint32_t c = cnt[b]; if (c == (uint32_t)m) one_bucket = true; cnt[b] = run; run += c; }
        if (one_bucket) continue;
        for (int64_t i = 0; i < m; ++i) {
          const uint64_t kk = key[(size_t)i];
          const uint32_t d = cnt[(size_t)((kk >> shift) & dmask)]++;
          key2[(size_t)d] = kk; idx2[(size_t)d] = idx[(size_t)i];
        }
        key.swap(key2); idx.swap(idx2);
      }
      std::vector<uint64_t>().swap(key2);
      PsaVec().swap(idx2);
      const int gs = 64 - need;
      std::vector<std::pair<uint64_t, uint32_t>> grp;
      for (int64_t g0 = 0; g0 < m;) {
        int64_t g1 = g0 + 1;
        while (g1 < m && (key[(size_t)g1] >> gs) == (key[(size_t)g0] >> gs)) ++g1;
        if (g1 - g0 > 1) {
          if (g1 - g0 > (1 << 16)) return false;
          grp.clear();
          for (int64_t k = g0; k < g1; ++k) grp.emplace_back(key[(size_t)k], idx[(size_t)k]);
          std::sort(grp.begin(), grp.end(), [&](const std::pair<uint64_t, uint32_t> &a, const std::pair<uint64_t, uint32_t> &b) {
            return a.first != b.first ? a.first < b.first : less_from(a.second, b.second, per_key);
          });
          for (int64_t k = g0; k < g1; ++k) idx[(size_t)k] = grp[(size_t)(k - g0)].second;
        }
        g0 = g1;
      }
    }
  } catch (const GtCapExceeded &) { return false; }
  out.beg = beg; out.size = m;
  out.psa_lo.swap(idx);
  out.psa_hi.clear();
  out.bwt.resize((size_t)m);
  out.gt_begin.assign((size_t)((m + 31) / 32 + 1), 0);
  bool gt_of_beg = false;
  if (end < n) {
    try { gt_of_beg = gt_wrt_end(text, n, beg, end, gt_tail_direct(text, n, end, cap), cap); } catch (const GtCapExceeded &) { return false; }
  }
  finish_halfblock(text, n, beg, end, gt_of_beg, out);
  return true;
}

// number of suffixes of the half-block smaller than text[p..n): binary search with direct text
// comparison (what em_compute_initial_ranks.hpp:222-319 computes for one position)
static inline int64_t rank_by_search(const uint8_t *text, int64_t n, const HalfBlock &hb, int64_t p) {
  if (p >= n) return 0;
  auto less = [&](int64_t s) {  // text[s..) < text[p..) ?
    int64_t k = 0;
    while (s + k < n && p + k < n) {
      uint8_t a = text[s + k], b = text[p + k];
      if (a != b) return a < b;
      ++k;
    }
    return s + k >= n;  // the shorter suffix is smaller (s != p)
  };
  int64_t lo = 0, hi = hb.size;
  while (lo < hi) {
    int64_t md = (lo + hi) / 2;
    int64_t s = hb.beg + ((int64_t)hb.psa_lo[(size_t)md] | (hb.psa_hi.empty() ? 0 : (int64_t)hb.psa_hi[(size_t)md] << 32));
    if (less(s)) lo = md + 1; else hi = md;
  }
  return lo;
}

}  // namespace psa_host
