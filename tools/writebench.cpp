// how fast can one process put a large result file into the page cache?  g++ -O2 -std=c++17 -o tools/writebench tools/writebench.cpp -lpthread
//   writebench FILE GiB : (a) fwrite in 320 MiB pieces, (b) pwrite from 8 threads, (c) fallocate + mmap + memcpy from 8 / 16 threads
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const char *fn = argv[1];
  const size_t total = (size_t)(atof(argv[2]) * (1ull << 30)), piece = 320ull << 20;
  std::vector<char> buf(piece);
  for (size_t i = 0; i < piece; ++i) buf[i] = (char)(i * 31);
  {
    double t0 = now();
    FILE *f = fopen(fn, "wb");
    for (size_t off = 0; off < total; off += piece) fwrite(buf.data(), 1, std::min(piece, total - off), f);
    fclose(f);
    printf("fwrite            : %.2f GB/s\n", total / 1e9 / (now() - t0));
    remove(fn);
  }
  for (int nt : {4, 8}) {
    double t0 = now();
    int fd = open(fn, O_RDWR | O_CREAT | O_TRUNC, 0644);
    for (size_t off = 0; off < total; off += piece) {
      const size_t len = std::min(piece, total - off), per = (len + nt - 1) / nt;
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { size_t lo = per * t, hi = std::min(len, lo + per); if (lo < hi) (void)!pwrite(fd, buf.data() + lo, hi - lo, (off_t)(off + lo)); });
      for (auto &x : th) x.join();
    }
    close(fd);
    printf("pwrite x%-2d        : %.2f GB/s\n", nt, total / 1e9 / (now() - t0));
    remove(fn);
  }
  for (int nt : {1, 8}) {   // O_DIRECT: past the page cache, at the speed of the device below the file system
    void *ab = nullptr;
    if (posix_memalign(&ab, 1 << 21, piece) != 0) return 1;
    memcpy(ab, buf.data(), piece);
    double t0 = now();
    int fd = open(fn, O_RDWR | O_CREAT | O_TRUNC | O_DIRECT, 0644);
    if (fd < 0) { perror("open O_DIRECT"); free(ab); break; }
    (void)!fallocate(fd, 0, 0, (off_t)total);
    bool ok = true;
    for (size_t off = 0; off < total && ok; off += piece) {
      const size_t len = std::min(piece, total - off), per = ((len + nt - 1) / nt + (1 << 20) - 1) >> 20 << 20;
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { size_t lo = per * t, hi = std::min(len, lo + per); if (lo < hi && pwrite(fd, (char *)ab + lo, hi - lo, (off_t)(off + lo)) != (ssize_t)(hi - lo)) ok = false; });
      for (auto &x : th) x.join();
    }
    close(fd);
    printf("O_DIRECT pwrite x%d: %.2f GB/s%s\n", nt, total / 1e9 / (now() - t0), ok ? "" : " (FAILED)");
    remove(fn);
    free(ab);
  }
  {   // the file's pages are in the page cache already (written with zeros ahead of time): overwrite them
    std::vector<char> z(piece, 0);
    int fd = open(fn, O_RDWR | O_CREAT | O_TRUNC, 0644);
    double t0 = now();
    for (size_t off = 0; off < total; off += piece) (void)!pwrite(fd, z.data(), std::min(piece, total - off), (off_t)off);
    printf("zero-fill (write)  : %.2f GB/s\n", total / 1e9 / (now() - t0));
    t0 = now();
    for (size_t off = 0; off < total; off += piece) (void)!pwrite(fd, buf.data(), std::min(piece, total - off), (off_t)off);
    printf("overwrite x1       : %.2f GB/s\n", total / 1e9 / (now() - t0));
    for (int nt : {4}) {
      t0 = now();
      for (size_t off = 0; off < total; off += piece) {
        const size_t len = std::min(piece, total - off), per = (len + nt - 1) / nt;
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { size_t lo = per * t, hi = std::min(len, lo + per); if (lo < hi) (void)!pwrite(fd, buf.data() + lo, hi - lo, (off_t)(off + lo)); });
        for (auto &x : th) x.join();
      }
      printf("overwrite x%d       : %.2f GB/s\n", nt, total / 1e9 / (now() - t0));
    }
    for (int nt : {8, 16}) {
      t0 = now();
      char *m = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, 0);
      if (m == MAP_FAILED) { perror("mmap"); break; }
      const double t1 = now();
      for (size_t off = 0; off < total; off += piece) {
        const size_t len = std::min(piece, total - off), per = ((len + nt - 1) / nt + 4095) / 4096 * 4096;
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { size_t lo = per * t, hi = std::min(len, lo + per); if (lo < hi) memcpy(m + off + lo, buf.data() + lo, hi - lo); });
        for (auto &x : th) x.join();
      }
      const double t2 = now();
      munmap(m, total);
      printf("resident mmap x%-2d  : %.2f GB/s (populate %.2fs, copy %.2fs = %.1f GB/s, unmap %.2fs)\n", nt, total / 1e9 / (now() - t0), t1 - t0, t2 - t1, total / 1e9 / (t2 - t1), now() - t2);
    }
    close(fd);
    remove(fn);
  }
  for (int nt : {8, 16}) {
    double t0 = now();
    int fd = open(fn, O_RDWR | O_CREAT | O_TRUNC, 0644);
    int rc = fallocate(fd, 0, 0, (off_t)total);
    double t1 = now();
    if (rc != 0) { perror("fallocate"); if (ftruncate(fd, (off_t)total) != 0) return 1; }
    char *m = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) { perror("mmap"); return 1; }
    for (size_t off = 0; off < total; off += piece) {
      const size_t len = std::min(piece, total - off), per = ((len + nt - 1) / nt + 4095) / 4096 * 4096;
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { size_t lo = per * t, hi = std::min(len, lo + per); if (lo < hi) memcpy(m + off + lo, buf.data() + lo, hi - lo); });
      for (auto &x : th) x.join();
    }
    double t2 = now();
    munmap(m, total);
    close(fd);
    printf("mmap x%-2d (falloc %d): %.2f GB/s (fallocate %.2fs, copy %.2fs, unmap+close %.2fs)\n", nt, rc, total / 1e9 / (now() - t0), t1 - t0, t2 - t1, now() - t2);
    remove(fn);
  }
}
