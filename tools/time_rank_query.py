"""Independent random rank queries/s on the real rank structure: python tools/time_rank_query.py [layout] [MiB]
layout: 0 auto (symbol-major for bytes), 32/64/... block layouts"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from psascan_amd import api, extras
from psascan_amd._lib import lib, check
layout = int(sys.argv[1]) if len(sys.argv) > 1 else 0
m = (int(sys.argv[2]) if len(sys.argv) > 2 else 2048) << 20
t = extras.gen_text(m, sigma=255, seed=1)
r = api.rank_build(t, m, layout)
nq = 1 << 27
rng = np.random.default_rng(1)
di = api.upload(rng.integers(0, m, nq, dtype=np.int64))
dc = api.upload(rng.integers(0, 255, nq, dtype=np.uint8))
do = api.DeviceBuffer(8 * nq)
for it in range(3):
    api.sync(); t0 = time.perf_counter()
    check(lib().psg_rank_query(r.h, di.ptr, dc.ptr, nq, do.ptr))
    api.sync(); dt = time.perf_counter() - t0
    print(f"layout {layout}: {nq / dt / 1e9:.2f} G queries/s ({dt * 1e3:.2f} ms, structure {r.device_bytes() / 2**30:.1f} GiB)", flush=True)
