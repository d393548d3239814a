"""Times the rank-log partition + window histogram (psgx_gap_hist) on 2^31 random u32 values."""
import ctypes as C, sys, time
sys.path.insert(0, ".")
from psascan_amd import api, extras
from psascan_amd._lib import lib, check
n = 1 << 31
m = 0xFFFFFFFE
log = api.DeviceBuffer(4 * n + 64)
gap = api.zeros(4 * (m + 1) + 64)
for it in range(4):
    extras.gen_text(4 * n, sigma=255, seed=it + 1, d_text=log)
    api.sync(); t0 = time.perf_counter()
    check(lib().psgx_gap_hist(log.ptr, n, m, gap.ptr))
    api.sync(); dt = time.perf_counter() - t0
    print(f"gap_hist 2^31 entries: {dt * 1e3:.2f} ms", flush=True)
