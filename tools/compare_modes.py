"""the same input through construct_sa in several modes: the .sa5 files must be identical
    python tools/compare_modes.py MiB kind   (kind: english | dna | bytes)"""
import hashlib, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, ".")
import psascan_amd
from psascan_amd import api, extras
mib, kind = int(sys.argv[1]), sys.argv[2]
psascan_amd.lib(0)
n = mib << 20
f = f"/tmp/cmp_{kind}_{mib}.bin"
d_t = extras.gen_text(n, {"english": extras.MODE_ENGLISH, "dna": extras.MODE_DNA, "bytes": extras.MODE_BYTES255}[kind], 0, seed=77)
api.download(d_t, np.uint8, n).tofile(f)
d_t.free()
api.lib().psg_trim()
env = dict(os.environ, OMP_NUM_THREADS="16")
block = str(n // 5 + 12345)
modes = {"default": [], "text-on-host": ["--text-on-host", "--tail-chunk", str(n // 7)], "device-sort": ["--device-sort"],
         "spill+checkpoint": ["--spill-psa", "--checkpoint", "/tmp/cmp_ck"], "no-device-merge": ["--no-device-merge"],
         # the external-memory schedule: a device budget of ~60 bytes per block symbol -- text, gt bits, partial SAs and merge bitvectors in host memory
         "hbm-limit": ["--hbm-limit", str(max(64 << 20, 60 * (n // 5 + 12345))), "--tail-chunk", str(n // 7)],
         # ... and with the partial SAs and the merge bitvectors in files (mapped for the merge): only the gt bits stay in host memory
         "hbm-limit+files": ["--hbm-limit", str(max(64 << 20, 60 * (n // 5 + 12345))), "--tail-chunk", str(n // 7), "--spill-psa", "-g", f + ".gap"]}
hashes = {}
for name, extra in modes.items():
    out = f + "." + name.replace("+", "_") + ".sa5"
    t0 = time.time()
    r = subprocess.run(["host/construct_sa", "-m", "8G", "--block-size", block, "-o", out, f] + extra, input="y\n", capture_output=True, text=True, env=env)
    h = hashlib.sha256(open(out, "rb").read()).hexdigest() if r.returncode == 0 else "FAILED " + r.stderr[-300:]
    os.remove(out) if os.path.exists(out) else None
    hashes[name] = h
    print(f"{name:18s} {time.time() - t0:6.1f}s {h[:32]}", flush=True)
os.remove(f)
ok = len(set(hashes.values())) == 1 and not any(v.startswith("FAILED") for v in hashes.values())
print("IDENTICAL" if ok else "DIFFERENT")
sys.exit(0 if ok else 1)
