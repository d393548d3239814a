"""construct_sa end to end on random bytes: python tools/run_cli.py [MiB] [-m value] [block_size] [threads]
(prints the tail of the program's own log)"""
import os, subprocess, sys, tempfile
import numpy as np
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mem = sys.argv[2] if len(sys.argv) > 2 else "204682040"
blk = sys.argv[3] if len(sys.argv) > 3 else ""
thr = sys.argv[4] if len(sys.argv) > 4 else "8"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    f = os.path.join(d, "x.bin")
    np.random.default_rng(7).integers(0, 255, mib << 20, dtype=np.uint8).tofile(f)
    cmd = [os.path.join(root, "host", "construct_sa"), "-m", mem, "-v"] + (["--block-size", blk] if blk else []) + [f]
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS=thr))
    print("rc", r.returncode, " ".join(cmd[1:-1]), "threads", thr)
    lines = r.stderr.strip().splitlines()
    print("\n".join(lines[-6:]))
