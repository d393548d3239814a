"""construct_sa end to end on English-like text: python tools/run_cli_english.py [MiB] [block_size] [threads]"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from test_scale_gpu import english_like
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
blk = sys.argv[2] if len(sys.argv) > 2 else str(32 << 20)
thr = sys.argv[3] if len(sys.argv) > 3 else "16"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    f = os.path.join(d, "x.txt")
    t0 = time.time()
    english_like(mib << 20, seed=3).tofile(f)
    print(f"text ready in {time.time() - t0:.1f} s", flush=True)
    r = subprocess.run([os.path.join(root, "host", "construct_sa"), "-m", "16G", "-v", "--block-size", blk, f], capture_output=True, text=True,
                       env=dict(os.environ, OMP_NUM_THREADS=thr))
    print("rc", r.returncode)
    print("\n".join(r.stderr.strip().splitlines()[-5:]))
    # spot check of the output order
    import numpy as np
    n = mib << 20
    sa5 = np.fromfile(f + ".sa5", np.uint8, 5 * 200000).reshape(-1, 5).astype(np.int64)
    pos = sa5[:, 0] | (sa5[:, 1] << 8) | (sa5[:, 2] << 16) | (sa5[:, 3] << 24) | (sa5[:, 4] << 32)
    text = np.fromfile(f, np.uint8)
    bad = 0
    for k in range(0, 199999, 997):
        a, b = int(pos[k]), int(pos[k + 1])
        if not bytes(text[a:a + 4096]) <= bytes(text[b:b + 4096]): bad += 1
    print("sampled adjacent pairs out of order (first 200000 entries, 4 KiB prefixes):", bad)
