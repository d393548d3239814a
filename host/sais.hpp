// sais.hpp -- clean-room SA-IS (Nong, Zhang, Chan 2009) suffix sorter, host side.
// Stands where the reference calls libdivsufsort / libsais on a (renamed) block
// (inmem_psascan_src/divsufsort_template.hpp:54-62, sais_template.hpp:55-63); the contract is
// only "suffix array of s[0..n)", order = lexicographic with shorter-is-smaller.
#pragma once
#include <cstdint>
#include <vector>

namespace psa_host {

template <class I> class Sais {
 public:
  // s: symbols in [0, K); sa: n entries out
  template <class S> static void run(const S *s, I *sa, I n, I K) {
    if (n <= 0) return;
    if (n == 1) { sa[0] = 0; return; }
    std::vector<bool> t((size_t)n);     // true = S-type
    t[(size_t)n - 1] = false;           // the implicit sentinel after s[n-1] is the smallest symbol
    for (I i = n - 2; i >= 0; --i) t[(size_t)i] = s[i] < s[i + 1] || (s[i] == s[i + 1] && t[(size_t)i + 1]);
    auto is_lms = [&](I i) { return i > 0 && t[(size_t)i] && !t[(size_t)i - 1]; };
    std::vector<I> cnt((size_t)K, 0), bkt((size_t)K);
    for (I i = 0; i < n; ++i) ++cnt[(size_t)s[i]];
    auto heads = [&]() { I sum = 0; for (I c = 0; c < K; ++c) { bkt[(size_t)c] = sum; sum += cnt[(size_t)c]; } };
    auto tails = [&]() { I sum = 0; for (I c = 0; c < K; ++c) { sum += cnt[(size_t)c]; bkt[(size_t)c] = sum; } };
    auto induce = [&]() {
      heads();
      // suffix n-1 is L-type and is induced by the sentinel, which is the smallest suffix
      sa[bkt[(size_t)s[n - 1]]++] = n - 1;
      for (I i = 0; i < n; ++i) {
        I j = sa[i];
        if (j > 0 && !t[(size_t)j - 1]) sa[bkt[(size_t)s[j - 1]]++] = j - 1;
      }
      tails();
      for (I i = n - 1; i >= 0; --i) {
        I j = sa[i];
        if (j > 0 && t[(size_t)j - 1]) sa[--bkt[(size_t)s[j - 1]]] = j - 1;
      }
    };
    // step 1: sort LMS substrings
    for (I i = 0; i < n; ++i) sa[i] = -1;
    tails();
    I n_lms = 0;
    for (I i = n - 1; i > 0; --i) if (is_lms(i)) { sa[--bkt[(size_t)s[i]]] = i; ++n_lms; }
    induce_guard(sa, n);
    induce();
    if (n_lms == 0) return;  // s is non-increasing up to its end: induce() sorted everything
    // step 2: compact sorted LMS positions, name the LMS substrings
    std::vector<I> lms_sorted;
    lms_sorted.reserve((size_t)n_lms);
    for (I i = 0; i < n; ++i) if (sa[i] > 0 && is_lms(sa[i])) lms_sorted.push_back(sa[i]);
    std::vector<I> name_of((size_t)n / 2 + 1, -1);  // indexed by position/2 (LMS positions are >= 2 apart)
    I names = 0, prev = -1;
    for (I k = 0; k < n_lms; ++k) {
      I p = lms_sorted[(size_t)k];
      bool diff = prev < 0;
      if (!diff) {
        for (I d = 0;; ++d) {
          I a = prev + d, b = p + d;
          if (a >= n || b >= n) { diff = true; break; }                  // one ran into the sentinel
          if (s[a] != s[b] || t[(size_t)a] != t[(size_t)b]) { diff = true; break; }
          if (d > 0 && (is_lms(a) || is_lms(b))) { diff = !(is_lms(a) && is_lms(b)); break; }
        }
      }
      if (diff) ++names;
      name_of[(size_t)p / 2] = names - 1;
      prev = p;
    }
    // step 3: order of the LMS suffixes
    std::vector<I> lms_pos;  // LMS positions in text order
    lms_pos.reserve((size_t)n_lms);
    for (I i = 1; i < n; ++i) if (is_lms(i)) lms_pos.push_back(i);
    std::vector<I> sa1((size_t)n_lms);
    if (names < n_lms) {
      std::vector<I> s1((size_t)n_lms);
      for (I k = 0; k < n_lms; ++k) s1[(size_t)k] = name_of[(size_t)lms_pos[(size_t)k] / 2];
      Sais<I>::run(s1.data(), sa1.data(), n_lms, names);
    } else {
      for (I k = 0; k < n_lms; ++k) sa1[(size_t)name_of[(size_t)lms_pos[(size_t)k] / 2]] = k;
    }
    // step 4: place sorted LMS suffixes at bucket tails, induce everything
    for (I i = 0; i < n; ++i) sa[i] = -1;
    tails();
    for (I k = n_lms - 1; k >= 0; --k) {
      I p = lms_pos[(size_t)sa1[(size_t)k]];
      sa[--bkt[(size_t)s[p]]] = p;
    }
    induce();
  }

 private:
  static void induce_guard(I *, I) {}
};

}  // namespace psa_host
