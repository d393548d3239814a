"""one construct_sa run with its log:  python tools/e2e_one.py MiB kind [construct_sa args...]"""
import os, subprocess, sys, tempfile, time
import numpy as np
sys.path.insert(0, ".")
from psascan_amd import api, extras
import psascan_amd
mib, kind = int(sys.argv[1]), sys.argv[2]
psascan_amd.lib(0)
n = mib << 20
d = os.environ.get("E2E_DIR", "/tmp")
f = os.path.join(d, f"e2e_{kind}_{mib}.bin")
if not os.path.exists(f):
    chunk = 1 << 30
    with open(f, "wb") as fh:
        for off in range(0, n, chunk):       # generated in pieces: the text never sits in host memory at once
            c = min(chunk, n - off)
            d_t = extras.gen_text(c, extras.MODE_ENGLISH if kind == "english" else extras.MODE_BYTES255 if kind == "bytes" else extras.MODE_DNA, 0, seed=11 + off // chunk)
            api.download(d_t, np.uint8, c).tofile(fh)
            d_t.free()
            print(f"generated {off + c >> 20} MiB", flush=True)
api.lib().psg_trim()
print("input file ready", flush=True)
t0 = time.time()
r = subprocess.run(["host/construct_sa"] + sys.argv[3:] + [f], stdout=sys.stdout, stderr=sys.stdout, text=True, env=dict(os.environ, OMP_NUM_THREADS="16"))   # the log streams through
dt = time.time() - t0
print(f"rc={r.returncode} {kind} {mib} MiB: {dt:.2f} s = {n / 1e6 / dt:.1f} MB/s", flush=True)
