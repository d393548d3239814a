// random no-return u32 atomic adds into tables of growing size: where do they execute (L2 / Infinity Cache / HBM)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/atomicbench tools/atomicbench.hip && tools/atomicbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
// every lane adds `steps` times at random places of a window of `win` counters that moves over the table workgroup by workgroup
// (win == n: the whole table is the window)
__global__ __launch_bounds__(256) void atomic_kernel(uint32_t *cnt, size_t n, size_t win, int steps) {
  uint64_t s = mix((uint64_t)blockIdx.x * 256 + threadIdx.x);
  const size_t nwin = n / win, base = ((size_t)blockIdx.x * nwin / gridDim.x) * win;
  for (int t = 0; t < steps; ++t) { s = mix(s); atomicAdd(&cnt[base + s % win], 1u); }
}
int main() {
  const size_t maxb = (size_t)8 << 30;
  uint32_t *cnt; CK(hipMalloc(&cnt, maxb)); CK(hipMemset(cnt, 0, maxb));
  const size_t lanes = 256ull * 32 * 64; const int grid = (int)(lanes / 256), steps = 256;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (size_t bytes : {(size_t)1 << 20, (size_t)4 << 20, (size_t)16 << 20, (size_t)32 << 20, (size_t)64 << 20, (size_t)128 << 20, (size_t)256 << 20, (size_t)1 << 30, (size_t)8 << 30}) {
    const size_t n = bytes / 4;
    hipLaunchKernelGGL(atomic_kernel, dim3(grid), dim3(256), 0, 0, cnt, n, n, steps); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(atomic_kernel, dim3(grid), dim3(256), 0, 0, cnt, n, n, steps);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("table %6zu MiB, all lanes anywhere      : %7.2f G atomics/s\n", bytes >> 20, 3.0 * lanes * steps / ms / 1e6);
  }
  // the partitioned shape: the table is 8 GiB, but at any time the resident workgroups work on neighbouring windows
  for (size_t wb : {(size_t)128 << 10, (size_t)1 << 20, (size_t)8 << 20, (size_t)32 << 20}) {
    const size_t n = maxb / 4, win = wb / 4;
    hipLaunchKernelGGL(atomic_kernel, dim3(grid), dim3(256), 0, 0, cnt, n, win, steps); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(atomic_kernel, dim3(grid), dim3(256), 0, 0, cnt, n, win, steps);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("table 8 GiB, windows of %6zu KiB per workgroup range: %7.2f G atomics/s\n", wb >> 10, 3.0 * lanes * steps / ms / 1e6);
  }
  return 0;
}
