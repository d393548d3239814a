// host leaf sorter (host/leafsort.hpp) throughput and self-check:  g++ -O2 -std=c++17 -o /tmp/leafbench2 tools/leafbench2.cpp -lpthread
//   leafbench2 MiB leaf_KiB threads [kind: 0 = english-like words, 1 = random bytes, 2 = dna] [check]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <thread>
#include <vector>
#include "../host/halfblock.hpp"
#include "../host/leafsort.hpp"
int main(int argc, char **argv) {
  const int64_t n = (int64_t)atol(argv[1]) << 20, leaf = (int64_t)(atof(argv[2]) * 1024);
  const int T = atoi(argv[3]), kind = argc > 4 ? atoi(argv[4]) : 0;
  const bool check = argc > 5;
  std::vector<uint8_t> text((size_t)n);
  std::mt19937_64 rng(5);
  if (kind == 1) for (auto &c : text) c = (uint8_t)(rng() % 255);
  else if (kind == 2) for (auto &c : text) c = "ACGT"[rng() & 3];
  else {
    std::vector<std::string> words;
    for (int w = 0; w < 5000; ++w) { std::string s; int len = 2 + (int)(rng() % 9); for (int k = 0; k < len; ++k) s.push_back((char)('a' + (rng() % 26) * (rng() % 26) / 26)); words.push_back(s); }
    size_t p = 0;
    while (p < (size_t)n) { double u = (double)(rng() >> 11) / 9007199254740992.0; size_t w = (size_t)(words.size() * u * u * u); for (char c : words[w]) if (p < (size_t)n) text[p++] = (uint8_t)c; if (p < (size_t)n) text[p++] = ' '; }
  }
  const int64_t nleaves = (n + leaf - 1) / leaf;
  std::atomic<int64_t> next{0}, ok{0}, bad{0};
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&] {
      psa_host::LeafScratch S;
      std::vector<uint16_t> out((size_t)leaf);
      for (;;) {
        int64_t k = next.fetch_add(1);
        if (k >= nleaves) return;
        const int64_t b = k * leaf, e = std::min(n, b + leaf);
        if (psa_host::sort_leaf16(text.data(), n, b, e, out.data(), 1 << 16, S)) ok++;
        if (check) {
          psa_host::HalfBlock h;
          psa_host::sort_halfblock(text.data(), n, b, e, psa_host::gt_tail_direct(text.data(), n, e, 1 << 16), h, 1 << 16);
          for (int64_t i = 0; i < e - b; ++i) if (h.psa_lo[(size_t)i] != out[(size_t)i]) { bad++; break; }
        }
      }
    });
  for (auto &x : th) x.join();
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("threads=%d leaf=%.0f KiB kind=%d: %.2f s = %.1f MB/s (%.1f per thread), sorted %ld of %ld leaves%s\n", T, leaf / 1024.0, kind, dt, n / 1e6 / dt, n / 1e6 / dt / T, (long)ok.load(), (long)nleaves,
         check ? (bad.load() ? " -- MISMATCH vs SA-IS" : " -- equal to SA-IS") : "");
  return bad.load() ? 1 : 0;
}
