"""Times psg_rank_build (symbol-major layout) on random bytes: python tools/time_rank_build.py [MiB]"""
import sys, time
sys.path.insert(0, ".")
from psascan_amd import api, extras
m = (int(sys.argv[1]) if len(sys.argv) > 1 else 2048) << 20
t = extras.gen_text(m, sigma=255, seed=1)
for it in range(4):
    api.sync(); t0 = time.perf_counter()
    r = api.rank_build(t, m, 1)
    api.sync(); dt = time.perf_counter() - t0
    r.free()
    print(f"rank_build {m >> 20} MiB: {dt * 1e3:.2f} ms", flush=True)
