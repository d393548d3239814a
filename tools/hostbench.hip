// hostbench.hip -- what the host side of a GPU box gives the staging tier of construct_sa:
// pinned allocation rate, PCIe copy rates (one direction, both at once), host memcpy rate, disk write rate.
//   hipcc --offload-arch=gfx950 -O2 -o tools/hostbench tools/hostbench.hip -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <thread>
#include <vector>

static double now() { timeval t; gettimeofday(&t, nullptr); return t.tv_sec + t.tv_usec * 1e-6; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
  size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 4;
  const size_t G = (size_t)1 << 30;
  CK(hipSetDevice(0));
  for (size_t s : {(size_t)1, gib, 4 * gib}) {
    void *p = nullptr;
    double t0 = now();
    CK(hipHostMalloc(&p, s * G, hipHostMallocDefault));
    double t1 = now();
    memset(p, 1, s * G);
    double t2 = now();
    CK(hipHostFree(p));
    double t3 = now();
    printf("hipHostMalloc %zu GiB: %.3f s (%.2f GiB/s), first touch memset %.3f s, free %.3f s\n", s, t1 - t0, s / (t1 - t0), t2 - t1, t3 - t2);
    fflush(stdout);
  }
  {   // register pageable memory
    size_t s = gib;
    char *q = (char *)malloc(s * G);
    double t0 = now();
    memset(q, 2, s * G);
    double t1 = now();
    hipError_t e = hipHostRegister(q, s * G, hipHostRegisterDefault);
    double t2 = now();
    printf("malloc+memset %zu GiB: %.3f s; hipHostRegister: %s %.3f s (%.2f GiB/s)\n", s, t1 - t0, hipGetErrorString(e), t2 - t1, s / (t2 - t1));
    if (e == hipSuccess) (void)hipHostUnregister(q);
    free(q);
  }
  void *h1 = nullptr, *h2 = nullptr, *d1 = nullptr, *d2 = nullptr;
  CK(hipHostMalloc(&h1, gib * G, hipHostMallocDefault));
  CK(hipHostMalloc(&h2, gib * G, hipHostMallocDefault));
  memset(h1, 3, gib * G); memset(h2, 4, gib * G);
  CK(hipMalloc(&d1, gib * G));
  CK(hipMalloc(&d2, gib * G));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now();
    CK(hipMemcpyAsync(d1, h1, gib * G, hipMemcpyHostToDevice, s1));
    CK(hipStreamSynchronize(s1));
    double t1 = now();
    CK(hipMemcpyAsync(h2, d2, gib * G, hipMemcpyDeviceToHost, s2));
    CK(hipStreamSynchronize(s2));
    double t2 = now();
    CK(hipMemcpyAsync(d1, h1, gib * G, hipMemcpyHostToDevice, s1));
    CK(hipMemcpyAsync(h2, d2, gib * G, hipMemcpyDeviceToHost, s2));
    CK(hipStreamSynchronize(s1));
    CK(hipStreamSynchronize(s2));
    double t3 = now();
    printf("pinned %zu GiB: H2D %.2f GiB/s, D2H %.2f GiB/s, both at once %.2f + %.2f GiB/s\n", gib, gib / (t1 - t0), gib / (t2 - t1), gib / (t3 - t2), gib / (t3 - t2));
  }
  {   // chunked copies (32 MiB pieces, back to back on one stream)
    const size_t piece = (size_t)32 << 20;
    double t0 = now();
    for (size_t o = 0; o < gib * G; o += piece) CK(hipMemcpyAsync((char *)d1 + o, (char *)h1 + o, piece, hipMemcpyHostToDevice, s1));
    CK(hipStreamSynchronize(s1));
    double t1 = now();
    printf("H2D in 32 MiB pieces: %.2f GiB/s\n", gib / (t1 - t0));
  }
  for (int nt : {1, 4, 8, 16}) {   // host memcpy pageable -> pinned
    char *src = (char *)malloc(gib * G);
    memset(src, 5, gib * G);
    double t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { size_t a = gib * G * t / nt, b = gib * G * (t + 1) / nt; memcpy((char *)h1 + a, src + a, b - a); });
    for (auto &x : th) x.join();
    double t1 = now();
    printf("host memcpy pageable->pinned, %d threads: %.2f GiB/s\n", nt, gib / (t1 - t0));
    free(src);
  }
  if (argc > 2) {   // disk write rate
    FILE *f = fopen(argv[2], "wb");
    if (f) {
      double t0 = now();
      size_t w = fwrite(h1, 1, gib * G, f);
      fflush(f); fclose(f);
      double t1 = now();
      printf("fwrite %zu GiB to %s: %.2f GiB/s (wrote %zu)\n", gib, argv[2], gib / (t1 - t0), w);
      remove(argv[2]);
    } else printf("cannot open %s\n", argv[2]);
  }
  return 0;
}
