"""construct_sa end to end on one file at several block sizes (host sorter vs GPU passes trade-off):
    python tools/e2e_blocksize.py [MiB] [english|bytes] [block MiB ...]"""
import os, subprocess, sys, tempfile, time
import numpy as np
sys.path.insert(0, ".")
from psascan_amd import api, extras
import psascan_amd
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
kind = sys.argv[2] if len(sys.argv) > 2 else "english"
blocks = [int(x) for x in sys.argv[3:]] or [4, 8, 16, 32, 64]
psascan_amd.lib(0)
n = mib << 20
with tempfile.TemporaryDirectory() as d:
    f = os.path.join(d, "t.bin")
    d_t = extras.gen_text(n, extras.MODE_ENGLISH if kind == "english" else extras.MODE_BYTES255, 0, seed=11)
    api.download(d_t, np.uint8, n).tofile(f)
    d_t.free()
    api.lib().psg_trim()
    for b in blocks:
        t0 = time.time()
        r = subprocess.run(["host/construct_sa", "-m", "8G", "--block-size", str(b << 20), "--check=256", "--discard-output", "-v", f], capture_output=True, text=True,
                           env=dict(os.environ, OMP_NUM_THREADS="16"))
        dt = time.time() - t0
        lines = r.stderr.splitlines()
        sort_wait = sum(float(l.split(":")[1].split("s")[0]) for l in lines if "host sufsort" in l)
        stream = sum(float(l.split(":")[1].split("s")[0]) for l in lines if "Stream (" in l)
        merge = [l.strip() for l in lines if "merge + write" in l]
        ok = "permutation sum ok, 0 of" in r.stderr and r.returncode == 0
        print(f"{kind} {mib} MiB, block {b} MiB: {dt:.2f} s = {n / 1e6 / dt:.1f} MB/s  ok={ok}  waited for host sort {sort_wait:.2f}s  stream passes {stream:.2f}s  {merge[:1]}", flush=True)
