"""wall vs device time of single C-ABI calls at bench scale (debug aid)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import psascan_amd
from psascan_amd import api, extras
L = psascan_amd.lib()
m = 1 << 31
gap = api.gap_array(m)
L.psg_memset(gap.ptr, 1, 4 * (m + 1))     # every counter = 0x01010101 is too big; use a kernel-free trick: bytes 1 -> value 16843009
L.psg_memset(gap.ptr, 0, 4 * api.gap_words(m))
one = api.upload(np.ones(1 << 20, np.uint32))
for k in range(0, m, 1 << 20):
    L.psg_d2d(gap.ptr + 4 * k, one.ptr, 4 << 20)
api.sync()
bv = api.zeros(4 * ((2 * m + 31) // 32 + 2))
for it in range(6):
    t = time.perf_counter()
    nb = api.gap_to_bitvector(gap, m, bv, 2 * m + 8)
    w = time.perf_counter() - t
    print(f"gap_to_bitvector wall {1e3*w:7.2f} ms  device {api.last_kernel_ms():7.2f} ms  nbits {nb}")
