// host leaf sorter throughput by thread count:  g++ -O2 -std=c++17 -o /tmp/leafbench tools/leafbench.cpp -lpthread
//   leafbench MiB leaf_MiB threads [kind: 0 = english-like words, 1 = random bytes]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>
#include "../host/halfblock.hpp"
int main(int argc, char **argv) {
  const int64_t n = (int64_t)atol(argv[1]) << 20, leaf = (int64_t)(atof(argv[2]) * (1 << 20));
  const int T = atoi(argv[3]), kind = argc > 4 ? atoi(argv[4]) : 0;
  std::vector<uint8_t> text((size_t)n);
  std::mt19937_64 rng(5);
  if (kind == 1) for (auto &c : text) c = (uint8_t)(rng() % 255);
  else {
    std::vector<std::string> words;
    for (int w = 0; w < 5000; ++w) { std::string s; int len = 2 + (int)(rng() % 9); for (int k = 0; k < len; ++k) s.push_back((char)('a' + (rng() % 26) * (rng() % 26) / 26)); words.push_back(s); }
    size_t p = 0;
    while (p < (size_t)n) { double u = (double)(rng() >> 11) / 9007199254740992.0; size_t w = (size_t)(words.size() * u * u * u); for (char c : words[w]) if (p < (size_t)n) text[p++] = (uint8_t)c; if (p < (size_t)n) text[p++] = ' '; }
  }
  const int64_t nleaves = n / leaf;
  std::atomic<int64_t> next{0};
  std::atomic<int64_t> radix_ok{0};
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&] {
      for (;;) {
        int64_t k = next.fetch_add(1);
        if (k >= nleaves) return;
        psa_host::HalfBlock h;
        const int64_t b = k * leaf, e = std::min(n, b + leaf);
        if (!getenv("NO_RADIX") && psa_host::sort_halfblock_radix(text.data(), n, b, e, h, 1 << 16)) radix_ok++;
        else { psa_host::HalfBlock h2; psa_host::sort_halfblock(text.data(), n, b, e, psa_host::gt_tail_direct(text.data(), n, e, 1 << 16), h2, 1 << 16); }
      }
    });
  for (auto &x : th) x.join();
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("threads=%d leaf=%.2f MiB kind=%d: %.2f s = %.1f MB/s (%.1f per thread), radix on %ld of %ld leaves\n", T, leaf / 1048576.0, kind, dt, n / 1e6 / dt, n / 1e6 / dt / T, (long)radix_ok.load(), (long)nleaves);
}
