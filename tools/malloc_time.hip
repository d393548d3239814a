// How long does the driver take to hand out device memory?  hipcc --offload-arch=gfx950 -O2 -o malloc_time malloc_time.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipFree(0);
  for (size_t gib : {1, 4, 8, 16, 32}) {
    void *p = nullptr;
    double t0 = now();
    hipError_t e = hipMalloc(&p, gib << 30);
    double t1 = now();
    hipMemset(p, 0, 64); hipDeviceSynchronize();
    double t2 = now();
    hipFree(p);
    double t3 = now();
    printf("hipMalloc %2zu GiB: %8.2f ms (%s)   first touch %6.2f ms   hipFree %8.2f ms\n", gib, t1 - t0, hipGetErrorString(e), t2 - t1, t3 - t2);
  }
  hipStream_t s; hipStreamCreate(&s);
  for (size_t gib : {1, 8, 32}) {
    void *p = nullptr;
    double t0 = now();
    hipError_t e = hipMallocAsync(&p, gib << 30, s);
    hipStreamSynchronize(s);
    double t1 = now();
    hipFreeAsync(p, s); hipStreamSynchronize(s);
    double t2 = now();
    void *q = nullptr;
    hipMallocAsync(&q, gib << 30, s); hipStreamSynchronize(s);
    double t3 = now();
    hipFreeAsync(q, s); hipStreamSynchronize(s);
    printf("hipMallocAsync %2zu GiB: %8.2f ms (%s)   free %8.2f ms   second malloc %8.2f ms\n", gib, t1 - t0, hipGetErrorString(e), t2 - t1, t3 - t2);
  }
  return 0;
}
