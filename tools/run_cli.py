"""construct_sa end to end on random symbols: python tools/run_cli.py [MiB] [-m value] [block_size] [threads] [sigma]
(prints the tail of the program's own log)"""
import os, subprocess, sys, tempfile
import numpy as np
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mem = sys.argv[2] if len(sys.argv) > 2 else "204682040"
blk = sys.argv[3] if len(sys.argv) > 3 else ""
thr = sys.argv[4] if len(sys.argv) > 4 else "8"
sigma = int(sys.argv[5]) if len(sys.argv) > 5 else 255
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    f = os.path.join(d, "x.bin")
    (np.random.default_rng(7).integers(0, sigma, mib << 20, dtype=np.uint8) + (65 if sigma < 64 else 0)).astype(np.uint8).tofile(f)
    cmd = [os.path.join(root, "host", "construct_sa"), "-m", mem, "-v"] + (["--block-size", blk] if blk else []) + [f]
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS=thr))
    print("rc", r.returncode, " ".join(cmd[1:-1]), "threads", thr)
    lines = r.stderr.strip().splitlines()
    print("\n".join(lines[-6:]))
    # spot check: sampled adjacent entries of the .sa5 must be in suffix order, and the file must have 5n bytes
    n = mib << 20
    text = np.memmap(f, np.uint8, "r")
    assert os.path.getsize(f + ".sa5") == 5 * n
    sa5 = np.memmap(f + ".sa5", np.uint8, "r")
    rng = np.random.default_rng(1)
    bad = 0
    big = 0
    for k in rng.integers(0, n - 1, 20000):
        e = np.array(sa5[5 * k: 5 * k + 10]).astype(np.int64)
        a = int(e[0] | e[1] << 8 | e[2] << 16 | e[3] << 24 | e[4] << 32)
        b = int(e[5] | e[6] << 8 | e[7] << 16 | e[8] << 24 | e[9] << 32)
        big += a >= (1 << 32)
        if a >= n or b >= n or not bytes(text[a:a + 256]) <= bytes(text[b:b + 256]):
            bad += 1
    print(f"sampled adjacent pairs out of order: {bad} of 20000 (positions >= 2^32 among them: {big})")
