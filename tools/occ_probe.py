"""chains planned by a pass in the 32-bit-log and the 40-bit-log kernels (resident workgroups per CU = chains / 65536)"""
import os, sys
import numpy as np
sys.path.insert(0, ".")
import psascan_amd
from psascan_amd import api, extras
psascan_amd.lib(0)
n = (1 << 30) + (1 << 24)
m = 1 << 24
d_text = extras.gen_text(n, extras.MODE_ENGLISH, 0, seed=5)
h = extras.sort_halfblock(d_text, n, 0, m, want_gt=False)
for wide in ("", "1"):
    if wide:
        os.environ["PSG_LOG_WIDE"] = "1"
    r = api.rank_build(h["bwt"], m)
    T = n - m
    d_gap = api.gap_array(m, fill=None)
    d_gtout = api.zeros(4 * ((T + 31) // 32 + 4))
    d_gtin = api.zeros(4 * ((T + 31) // 32 + 4))
    last = int(api.download(d_text, np.uint8, 1, m - 1)[0])
    for it in range(2):
        fin, st = api.stream_gap(r, h["i0"], last, d_text.at(m), T, d_gtin, 0, d_gap, d_gtout, 0, fresh_gap=True)
    print("wide" if wide else "u32 ", "chains", st.n_chains, "len", st.chain_len, "blocks/CU", st.n_chains / 65536.0, "kernel ms", round(st.kernel_ms, 3), "G/s", round(T / st.kernel_ms / 1e6, 2))
    r.free()
