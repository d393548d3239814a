// gap_array.hip -- the gap array container: counters + in-band excess list (include/psascan_amd.h).
// Mirrors buffered_gap_array (gap_array.hpp:55-383): value(j) = counter[j] + 2^bits * #{excess entries equal to j}
// (get_next, gap_array.hpp:116-124; add_excess :79-88; the producers are update.hpp:88-96).
#include "dev_common.hpp"

#include <rocprim/device/device_radix_sort.hpp>

#include <cstdlib>
#include <cstring>

using namespace psg;

extern "C" int64_t psg_gap_words(int64_t m) { return PSG_GAP_WORDS(m); }

int psg::gap_prepare(u32 *d_gap, i64 m, bool fresh, int *bits) {
  u32 *hdr = gap_hdr(d_gap, m);
  u32 h[4] = {0, 0, 0, 0};
  int want = 32;
  if (const char *e = getenv("PSG_GAP_COUNTER_BITS")) { int v = atoi(e); if (v == 8 || v == 16) want = v; }
  if (!fresh) {
    PSG_HIP(hipMemcpyAsync(pinned_buf(3, 64), hdr, 16, hipMemcpyDeviceToHost, stream()));
    PSG_HIP(psg::sync_stream());
    memcpy(h, pinned_buf(3, 64), 16);
    if (h[1] != 0) { *bits = (int)h[1]; return 0; }          // an array in use keeps its width
    if (h[0] != 0) { set_error("gap array: excess entries without a counter width (array not initialised?)"); return PSG_EINVAL; }
  }
  // header = {0 entries, `want` bits, no flag, default capacity}: two memsets, no host round trip
  PSG_HIP(hipMemsetAsync(hdr, 0, 16, stream()));
  PSG_HIP(hipMemsetAsync((char *)hdr + 4, want, 1, stream()));
  *bits = want;
  return 0;
}

int psg::gap_excess_view(const u32 *d_gap, i64 m, ExcessView *view, void **owned) {
  *owned = nullptr;
  view->sorted = nullptr; view->n = 0; view->bits = 32;
  const u32 *hdr = d_gap + PSG_GAP_HDR_WORD(m);
  u32 h[4];
  PSG_HIP(hipMemcpyAsync(pinned_buf(3, 64), hdr, 16, hipMemcpyDeviceToHost, stream()));
  PSG_HIP(psg::sync_stream());
  memcpy(h, pinned_buf(3, 64), 16);
  if (h[2]) { set_error("gap array: excess list capacity exceeded"); return PSG_ECHECK; }
  view->bits = h[1] ? (int)h[1] : 32;
  if (h[0] == 0) return 0;
  if (h[0] > (u32)PSG_GAP_EXCESS_CAP) { set_error("gap array: corrupt excess header"); return PSG_ECHECK; }
  // sort the entries by slot (the reference sorts its excess the same way before consuming it: gap_array.hpp:457-499)
  void *out = nullptr, *tmp = nullptr;
  if (psg::pool_alloc(&out, (size_t)h[0] * 8) != hipSuccess) { set_error("gap array: allocation failed"); return PSG_ENOMEM; }
  size_t tb = 0;
  const u64 *in = (const u64 *)(hdr + 4);
  hipError_t e = rocprim::radix_sort_keys(nullptr, tb, in, (u64 *)out, (size_t)h[0], 0, 64, stream());
  if (e == hipSuccess && psg::pool_alloc(&tmp, tb < 16 ? 16 : tb) != hipSuccess) e = hipErrorOutOfMemory;
  if (e == hipSuccess) e = rocprim::radix_sort_keys(tmp, tb, in, (u64 *)out, (size_t)h[0], 0, 64, stream());
  if (tmp) psg::pool_free(tmp);
  if (e != hipSuccess) { psg::pool_free(out); set_error(std::string("gap array: sorting the excess failed: ") + hipGetErrorString(e)); return PSG_EDEVICE; }
  view->sorted = (const u64 *)out; view->n = h[0];
  *owned = out;
  return 0;
}

__global__ __launch_bounds__(PSG_WG) void gap_values_kernel(const u32 *gap, i64 m, ExcessView X, u64 *out) {
  const i64 base = ((i64)blockIdx.x * PSG_WG + threadIdx.x) * 8;
  if (base > m) return;
  u64 g[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) g[q] = base + q <= m ? gap[base + q] : 0;
  excess_apply8(X, base, g);
#pragma unroll
  for (int q = 0; q < 8; ++q) if (base + q <= m) out[base + q] = g[q];
}

extern "C" int psg_gap_values(const uint32_t *d_gap, int64_t m, uint64_t *d_out) {
  PSG_REQUIRE(d_gap && d_out && m >= 0, "psg_gap_values");
  ExcessView X;
  void *owned = nullptr;
  if (int rc = gap_excess_view(d_gap, m, &X, &owned)) return rc;
  hipLaunchKernelGGL(gap_values_kernel, dim3((unsigned)cdiv(cdiv(m + 1, 8), PSG_WG)), dim3(PSG_WG), 0, stream(), d_gap, m, X, d_out);
  hipError_t e = hipGetLastError();
  hipError_t e2 = psg::sync_stream();
  if (owned) psg::pool_free(owned);
  PSG_HIP(e); PSG_HIP(e2);
  return 0;
}
