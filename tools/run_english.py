"""English-like text (Zipfian words, sigma ~ 28) with the host SA-IS sorter: per-pass statistics on the GPU.
    python tools/run_english.py [text_MiB] [block_MiB]"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_scale_gpu import english_like
from psascan_amd import api, extras, pipeline
from psascan_amd.hostsort import HostSorter
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
blk = int(sys.argv[2]) if len(sys.argv) > 2 else 128
n = mib << 20
t0 = time.time()
text = english_like(n, seed=3)
print(f"text ready in {time.time() - t0:.1f} s", flush=True)
d_text = api.upload(text, pad_to=16)
stats = []
t0 = time.time()
d_out = pipeline.construct_sa5(text, blk << 20, 1 << 40, HostSorter(), stats=stats, d_text=d_text, return_device=True)
api.sync()
print(f"whole schedule {time.time() - t0:.1f} s (host SA-IS sorter, single thread)", flush=True)
bad, s = extras.check_sa5(d_text, n, d_out, n, samples=1 << 20, seed=9)
print("sampled pairs out of order", bad, "permutation sum ok", s == (n * (n - 1) // 2) % (1 << 64))
for kind, b, e, st in stats:
    T = (e - b) - (e - b + 1) // 2 if kind == "A" else n - e
    print(f"  pass {kind} block [{b},{e}): ~{T / 2**20:.0f} Mi suffixes, total {st.total_ms:.1f} ms (kernel {st.kernel_ms:.1f} = {T / max(st.kernel_ms, 1e-9) / 1e6:.1f} G suffixes/s, hist {st.hist_ms:.1f}), "
          f"warm-up {st.warmup_steps}, unresolved {st.unresolved}, rounds {st.rounds}")
