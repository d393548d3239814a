// membench.hip -- what the MI355X memory system gives the access pattern of the stream kernel.
//   hipcc --offload-arch=gfx950 -O3 -o membench membench.hip && ./membench [table_GiB] [atomic_GiB]
// Reports (a) streaming read GB/s, (b) dependent random 16-byte loads/s (1 and 2 per step),
// (c) random no-return u32 atomic adds/s, (d) 2 loads + 1 atomic per step (the stream kernel's
// shape: counter sector + data sector + gap increment).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}

__global__ void fill_kernel(uint4 *p, size_t n16) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += (size_t)gridDim.x * blockDim.x) {
    uint64_t a = mix(k), b = mix(a);
    p[k] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
  }
}

__global__ __launch_bounds__(256) void stream_read_kernel(const uint4 *p, size_t n16, uint32_t *sink) {
  uint32_t acc = 0;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += (size_t)gridDim.x * blockDim.x) {
    uint4 v = p[k]; acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678) *sink = acc;
}

// LOADS dependent random 16-byte loads per step (+ optional atomic), `steps` steps per lane
template <int LOADS, bool ATOMIC, int STRIDE16>
__global__ __launch_bounds__(256) void chase_kernel(const uint4 *tab, size_t nrows, uint32_t *cnt, size_t ncnt, int steps, uint32_t *sink) {
  uint64_t s = mix((uint64_t)blockIdx.x * 256 + threadIdx.x);
  uint32_t acc = 0;
  for (int t = 0; t < steps; ++t) {
    uint64_t r = s % nrows;
    uint4 v = tab[r * STRIDE16];
    uint32_t x = v.x;
    if (LOADS == 2) { uint4 w = tab[r * STRIDE16 + STRIDE16 - 1]; x ^= w.y; }   // far end of the same row: another sector
    acc += x;
    s = mix(s ^ x);                                                               // next address depends on the data
    if (ATOMIC) atomicAdd(&cnt[(s >> 20) % ncnt], 1u);
  }
  if (acc == 0x12345678) *sink = acc;
}

// the real per-step load shape of stream_kernel<256,64>: one dword from the counter sector + NX
// dwordx4 loads that all fall into ONE other 64-byte sector of the same 1088-byte row
template <int NX>
__global__ __launch_bounds__(256) void shape_kernel(const uint4 *tab, size_t nrows, int steps, uint32_t *sink) {
  uint64_t s = mix((uint64_t)blockIdx.x * 256 + threadIdx.x);
  uint32_t acc = 0;
  for (int t = 0; t < steps; ++t) {
    uint64_t r = s % nrows;
    const uint4 *row = tab + r * 68;
    uint32_t x = ((const uint32_t *)row)[(s >> 40) & 255];
#pragma unroll
    for (int q = 0; q < NX; ++q) { uint4 w = row[64 + q]; x ^= w.x + w.y + w.z + w.w; }
    acc += x;
    s = mix(s ^ x);
  }
  if (acc == 0x12345678) *sink = acc;
}

// the stream kernel's full memory shape: one dependent random 16-byte load per step, one coalesced
// 16-byte log store per 4 steps (LOG), one private streaming 16-byte load per 16 steps (TEXT)
template <bool LOG, bool TEXT>
__global__ __launch_bounds__(256) void shape2_kernel(const uint4 *tab, size_t nrows, uint4 *log, const uint4 *text, int steps, uint32_t *sink) {
  const size_t lanes = (size_t)gridDim.x * 256, lane = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint64_t s = mix(lane);
  uint32_t acc = 0;
  for (int t = 0; t < steps; ++t) {
    uint4 v = tab[(s % nrows) * 68];
    uint32_t x = v.x;
    if (TEXT && (t & 15) == 0) { uint4 w = text[lane * (size_t)(steps / 16) + (t >> 4)]; x ^= w.z; }
    acc += x;
    s = mix(s ^ x);
    if (LOG && (t & 3) == 3) log[(size_t)(t >> 2) * lanes + lane] = make_uint4(x, acc, (uint32_t)s, t);
  }
  if (acc == 0x12345678) *sink = acc;
}

template <int U>
__global__ __launch_bounds__(256) void atomic_kernel(uint32_t *cnt, size_t ncnt, int steps) {
  uint64_t s = mix((uint64_t)blockIdx.x * 256 + threadIdx.x);
  for (int t = 0; t < steps; ++t) { s = mix(s); atomicAdd(&cnt[s % ncnt], 1u); }
}

__global__ __launch_bounds__(256) void stream_write_kernel(uint4 *p, size_t n16) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += (size_t)gridDim.x * blockDim.x)
    p[k] = make_uint4((uint32_t)k, 1, 2, 3);
}
__global__ __launch_bounds__(256) void stream_copy_kernel(const uint4 *p, uint4 *q, size_t n16) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += (size_t)gridDim.x * blockDim.x) q[k] = p[k];
}
// the write shape of a radix partition / symbol-major fill: every workgroup owns 512 output streams
// far apart and appends one run of R16 x 16 bytes to each of them per round
template <int R16>
__global__ __launch_bounds__(256) void runs_write_kernel(uint4 *p, size_t n16) {
  const size_t G = gridDim.x, S = 512, chunk16 = n16 / (G * S);
  for (size_t it = 0; it + R16 <= chunk16; it += R16)
    for (size_t sb = 0; sb < S; sb += 256 / R16) {
      size_t st = sb + threadIdx.x / R16;
      p[(st * G + blockIdx.x) * chunk16 + it + threadIdx.x % R16] = make_uint4((uint32_t)it, 1, 2, 3);
    }
}

template <class F> static double time_ms(F f, int reps = 3) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char **argv) {
  double tab_gib = argc > 1 ? atof(argv[1]) : 34.0, cnt_gib = argc > 2 ? atof(argv[2]) : 8.0;
  size_t tab_bytes = (size_t)(tab_gib * (1ull << 30)), cnt_bytes = (size_t)(cnt_gib * (1ull << 30));
  uint4 *tab; uint32_t *cnt, *sink;
  CK(hipMalloc(&tab, tab_bytes)); CK(hipMalloc(&cnt, cnt_bytes)); CK(hipMalloc(&sink, 4));
  size_t n16 = tab_bytes / 16;
  hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, tab, n16);
  CK(hipMemset(cnt, 0, cnt_bytes)); CK(hipDeviceSynchronize());
  printf("table %.1f GiB, counters %.1f GiB\n", tab_gib, cnt_gib);
  double ms = time_ms([&] { hipLaunchKernelGGL(stream_read_kernel, dim3(256 * 16), dim3(256), 0, 0, tab, n16, sink); });
  printf("streaming read            : %8.1f GB/s\n", tab_bytes / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL(stream_write_kernel, dim3(256 * 16), dim3(256), 0, 0, tab, n16); });
  printf("streaming write           : %8.1f GB/s\n", tab_bytes / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL(stream_copy_kernel, dim3(256 * 16), dim3(256), 0, 0, tab, tab + n16 / 2, n16 / 2); });
  printf("streaming copy (r+w bytes): %8.1f GB/s\n", tab_bytes / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((runs_write_kernel<2>), dim3(2048), dim3(256), 0, 0, tab, n16); });
  printf("512-stream runs of   32 B : %8.1f GB/s\n", tab_bytes / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((runs_write_kernel<4>), dim3(2048), dim3(256), 0, 0, tab, n16); });
  printf("512-stream runs of   64 B : %8.1f GB/s\n", tab_bytes / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((runs_write_kernel<8>), dim3(2048), dim3(256), 0, 0, tab, n16); });
  printf("512-stream runs of  128 B : %8.1f GB/s\n", tab_bytes / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((runs_write_kernel<16>), dim3(2048), dim3(256), 0, 0, tab, n16); });
  printf("512-stream runs of  256 B : %8.1f GB/s\n", tab_bytes / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((runs_write_kernel<64>), dim3(2048), dim3(256), 0, 0, tab, n16); });
  printf("512-stream runs of 1024 B : %8.1f GB/s\n", tab_bytes / ms / 1e6);
  hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, tab, n16);
  CK(hipDeviceSynchronize());
  const int steps = 512;
  const size_t lanes = 256ull * 32 * 64;   // 8 waves/SIMD on 256 CUs
  const int grid = (int)(lanes / 256);
  // rows of 1088 B (= 68 x 16 B): counter sector at the start, data sector at the end -- the rank block shape
  size_t nrows = n16 / 68;
  ms = time_ms([&] { hipLaunchKernelGGL((chase_kernel<1, false, 68>), dim3(grid), dim3(256), 0, 0, tab, nrows, cnt, cnt_bytes / 4, steps, sink); });
  printf("dependent random loads x1 : %8.2f G steps/s  (%.1f ns/step/lane)\n", lanes * steps / ms / 1e6, ms * 1e6 / steps);
  ms = time_ms([&] { hipLaunchKernelGGL((chase_kernel<1, false, 1>), dim3(grid), dim3(256), 0, 0, tab, n16, cnt, cnt_bytes / 4, steps, sink); });
  printf("  same, any 16-byte slot  : %8.2f G steps/s\n", lanes * steps / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((chase_kernel<1, false, 4>), dim3(grid), dim3(256), 0, 0, tab, n16 / 4, cnt, cnt_bytes / 4, steps, sink); });
  printf("  same, 64-byte aligned   : %8.2f G steps/s\n", lanes * steps / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((chase_kernel<1, false, 8>), dim3(grid), dim3(256), 0, 0, tab, n16 / 8, cnt, cnt_bytes / 4, steps, sink); });
  printf("  same, 128-byte aligned  : %8.2f G steps/s\n", lanes * steps / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((chase_kernel<2, false, 68>), dim3(grid), dim3(256), 0, 0, tab, nrows, cnt, cnt_bytes / 4, steps, sink); });
  printf("dependent random loads x2 : %8.2f G steps/s  (%.1f ns/step/lane)\n", lanes * steps / ms / 1e6, ms * 1e6 / steps);
  {
    uint4 *logb = (uint4 *)cnt;                              // 8 GiB: lanes * steps/4 * 16 B = 1 GiB used
    const uint4 *text = tab + n16 / 2;                       // lanes * steps/16 * 16 B = 256 MiB used
    ms = time_ms([&] { hipLaunchKernelGGL((shape2_kernel<true, false>), dim3(grid), dim3(256), 0, 0, tab, nrows / 2, logb, text, steps, sink); });
    printf("random load + log store/4 : %8.2f G steps/s\n", lanes * steps / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((shape2_kernel<false, true>), dim3(grid), dim3(256), 0, 0, tab, nrows / 2, logb, text, steps, sink); });
    printf("random load + text/16     : %8.2f G steps/s\n", lanes * steps / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((shape2_kernel<true, true>), dim3(grid), dim3(256), 0, 0, tab, nrows / 2, logb, text, steps, sink); });
    printf("random load + log + text  : %8.2f G steps/s   <- stream kernel (symbol-major, rank log)\n", lanes * steps / ms / 1e6);
    CK(hipMemset(cnt, 0, cnt_bytes));
  }
  ms = time_ms([&] { hipLaunchKernelGGL((atomic_kernel<1>), dim3(grid), dim3(256), 0, 0, cnt, cnt_bytes / 4, steps); });
  printf("random u32 atomics        : %8.2f G atomics/s\n", lanes * steps / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((chase_kernel<1, true, 68>), dim3(grid), dim3(256), 0, 0, tab, nrows, cnt, cnt_bytes / 4, steps, sink); });
  printf("1 load + 1 atomic / step  : %8.2f G steps/s\n", lanes * steps / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((chase_kernel<2, true, 68>), dim3(grid), dim3(256), 0, 0, tab, nrows, cnt, cnt_bytes / 4, steps, sink); });
  printf("2 loads + 1 atomic / step : %8.2f G steps/s   <- stream kernel shape\n", lanes * steps / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((shape_kernel<1>), dim3(grid), dim3(256), 0, 0, tab, nrows, steps, sink); });
  printf("dword + 1 x dwordx4 / step: %8.2f G steps/s\n", lanes * steps / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((shape_kernel<2>), dim3(grid), dim3(256), 0, 0, tab, nrows, steps, sink); });
  printf("dword + 2 x dwordx4 / step: %8.2f G steps/s\n", lanes * steps / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((shape_kernel<4>), dim3(grid), dim3(256), 0, 0, tab, nrows, steps, sink); });
  printf("dword + 4 x dwordx4 / step: %8.2f G steps/s   <- stream_kernel<256,64> loads\n", lanes * steps / ms / 1e6);
  // half the lanes (4 waves/SIMD) to see the latency/occupancy dependence
  ms = time_ms([&] { hipLaunchKernelGGL((chase_kernel<2, true, 68>), dim3(grid / 2), dim3(256), 0, 0, tab, nrows, cnt, cnt_bytes / 4, steps, sink); });
  printf("  same at 4 waves/SIMD    : %8.2f G steps/s\n", lanes / 2 * steps / ms / 1e6);
  return 0;
}
