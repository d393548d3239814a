"""construct_sa on periodic text (worst case for the chain start ranks and for the look-ahead sorter):
    python tools/run_cli_periodic.py [MiB] [block_size]"""
import os, subprocess, sys, tempfile, time
import numpy as np
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 16
blk = sys.argv[2] if len(sys.argv) > 2 else str(4 << 20)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = mib << 20
with tempfile.TemporaryDirectory() as d:
    f = os.path.join(d, "p.bin")
    np.frombuffer((b"abc" * (n // 3 + 1))[:n], np.uint8).tofile(f)
    t0 = time.time()
    r = subprocess.run([os.path.join(root, "host", "construct_sa"), "-m", "8G", "--block-size", blk, f], capture_output=True, text=True,
                       env=dict(os.environ, OMP_NUM_THREADS="16"))
    print("rc", r.returncode, f"wall {time.time() - t0:.1f} s")
    print("\n".join(r.stderr.strip().splitlines()[-3:]))
    sa5 = np.fromfile(f + ".sa5", np.uint8).reshape(-1, 5).astype(np.int64)
    pos = sa5[:, 0] | (sa5[:, 1] << 8) | (sa5[:, 2] << 16) | (sa5[:, 3] << 24) | (sa5[:, 4] << 32)
    # suffix array of (abc)^k: all 'a' suffixes by increasing length... = positions = 0 mod 3 descending, then 1 mod 3, then 2 mod 3
    want = np.concatenate([np.arange(n - 1 - ((n - 1 - r0) % 3), -1, -3) for r0 in (0, 1, 2)])
    print("matches the closed form:", bool(np.array_equal(pos, want)))
