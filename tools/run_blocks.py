"""Multi-block run on one GPU with the device-side sorter (random bytes): per-pass statistics.
    python tools/run_blocks.py [text_MiB] [block_MiB] [bytes|dna]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from psascan_amd import api, extras, pipeline
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
blk = int(sys.argv[2]) if len(sys.argv) > 2 else 512
n = (mib << 20) + 12345
mode = extras.MODE_DNA if len(sys.argv) > 3 and sys.argv[3] == "dna" else extras.MODE_BYTES255
d_text = extras.gen_text(n, mode, 0, seed=7)
text = api.download(d_text, np.uint8, n)
sorter = extras.DeviceSorter(d_text, n)
stats = []
t0 = time.time()
d_out = pipeline.construct_sa5(text, blk << 20, 1 << 40, sorter, stats=stats, d_text=d_text, return_device=True)
api.sync()
wall = time.time() - t0
bad, s = extras.check_sa5(d_text, n, d_out, n, samples=1 << 20, seed=5)
print(f"n={n} blocks of {blk} MiB: wall {wall:.2f} s, sampled pairs out of order {bad}, permutation sum ok {s == (n * (n - 1) // 2) % (1 << 64)}")
tot_stream = 0.0
suff = 0
for kind, b, e, st in stats:
    T = (e - b) - (e - b + 1) // 2 if kind == "A" else n - e      # pass A: right half through left; pass B: whole tail
    print(f"  pass {kind} block [{b},{e}): ~{T / 2**20:.0f} Mi suffixes, total {st.total_ms:.1f} ms (kernel {st.kernel_ms:.1f}, hist {st.hist_ms:.1f}), chains {st.n_chains} x {st.chain_len}, rounds {st.rounds}, unresolved {st.unresolved}")
    tot_stream += st.total_ms
    suff += T
print(f"  streaming passes: {tot_stream:.1f} ms for {suff / 2**30:.2f} Gi suffixes = {suff / tot_stream / 1e6:.2f} G suffixes/s")
