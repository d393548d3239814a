"""Randomised end-to-end check of host/construct_sa against the oracle's suffix array (run on a GPU box, from the
repository root; test infrastructure -- it uses the oracle -- but not collected by pytest):
    python tests/fuzz_cli.py [cases] [seed] [first case] [max symbols]
Random texts (alphabets, runs, repeats, zero bytes), random block / leaf sizes, fan-outs and modes."""
import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import orc
CLI = "host/construct_sa"
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # cases before this one are generated (same random stream) but not run
max_n = int(sys.argv[4]) if len(sys.argv) > 4 else 200_000     # above 2 M symbols the output is verified on the device (--check) instead of against the oracle
env = dict(os.environ, OMP_NUM_THREADS="4")


def text(kind, n):
    if kind == 0:
        return rng.integers(0, rng.integers(1, 255), n, dtype=np.uint8)
    if kind == 1:                                   # runs of varying length
        k = max(1, n // 20)
        return np.repeat(rng.integers(0, 6, k, dtype=np.uint8), rng.integers(1, 60, k))[:n].copy()
    if kind == 2:                                   # words
        w = [bytes(rng.integers(97, 105, rng.integers(1, 9), dtype=np.uint8)) for _ in range(30)]
        return np.frombuffer(b" ".join(w[i] for i in rng.integers(0, 30, n // 3 + 2)), np.uint8)[:n].copy()
    if kind == 3:                                   # a long repeat: X X' Y
        x = rng.integers(0, 200, max(2, n // 3), dtype=np.uint8)
        y = x.copy(); y[rng.integers(0, len(y))] ^= 1
        return np.concatenate([x, y, rng.integers(0, 200, n - 2 * len(x), dtype=np.uint8)])[:n].copy()
    p = int(rng.integers(1, 7))                     # periodic with a defect
    t = np.frombuffer((bytes(rng.integers(0, 4, p, dtype=np.uint8)) * (n // p + 1))[:n], np.uint8).copy()
    if n > 10 and rng.integers(0, 2):
        t[rng.integers(0, n)] = 9
    return t


bad = 0
with tempfile.TemporaryDirectory() as d:
    for c in range(cases):
        n = int(rng.integers(2, max_n))
        t = text(int(rng.integers(0, 5)), n)
        n = len(t)
        f = os.path.join(d, "x.bin")
        t.tofile(f)
        args = ["-m", "1G"]
        if rng.integers(0, 4):
            args += ["--block-size", str(int(rng.integers(2, max(3, n))))]
        mode = int(rng.integers(0, 6))
        if mode == 1:
            args += ["--device-sort"]
        if mode == 2:
            args += ["--text-on-host", "--tail-chunk", str(int(rng.integers(max(64, n // 40), max(100, n))))]   # (a pass costs ~3 ms per chunk)
        if mode == 3:
            args += ["--no-device-merge"]
        if mode == 4:
            args += ["--spill-psa"]
        if mode == 5:                                      # the host tier: text, gt bits, partial SAs and merge bitvectors in host memory
            args += ["--hbm-limit", "256Mi", "--tail-chunk", str(int(rng.integers(max(64, n // 40), max(100, n))))]
            if "--block-size" not in args:                 # (the default block of -m 1G is beyond that budget and refused up front:
                args += ["--block-size", str(max(2, min(n, 4_000_000)))]   # ~50 bytes of device memory per block symbol)
        run_env = dict(env, PSASCAN_MBV_ON_HOST="1") if mode in (0, 4) and rng.integers(0, 3) == 0 else env   # merge bitvectors alone in host memory
        if rng.integers(0, 2):
            args += ["--leaf-size", str(int(rng.integers(500, 60000)))]
            if rng.integers(0, 2):                         # the older tree (one pass at a time, F sub-ranges per step); else: a level per launch sequence
                args += ["--fanout", str(int(rng.integers(2, 9)))]
        if rng.integers(0, 3) == 0:
            args += ["--chains", str(int(rng.integers(1, 5000)))]
        big = n > 2_000_000
        if big:
            args += ["--check=2000"]      # (--text-on-host: the check then runs on the host, over the mapped text)
        out = os.path.join(d, "x.sa5")
        if c < first:
            continue
        print("run", c, n, args, flush=True)
        if os.environ.get("FUZZ_DRY"):
            continue
        try:
            r = subprocess.run([CLI] + args + ["-v", "-o", out, f], input="y\n", capture_output=True, text=True, timeout=300 if big else 120, env=run_env)
        except subprocess.TimeoutExpired as ex:
            bad += 1
            t.tofile(f"gpurun_out/fuzz_timeout_{c}.bin")
            print("TIMEOUT", c, n, args, (ex.stderr or b"")[-600:], flush=True)
            continue
        if big:
            ok = r.returncode == 0 and os.path.getsize(out) == 5 * n and "permutation sum ok, 0 of" in r.stderr
        else:
            ok = r.returncode == 0 and np.array_equal(orc.sa5_to_sa(np.fromfile(out, np.uint8)), orc.suffix_array(t))
        if not ok:
            bad += 1
            keep = f"gpurun_out/fuzz_fail_{c}.bin"
            t.tofile(keep)
            print("FAIL", c, n, args, r.returncode, r.stderr[-400:].replace("\n", " | "), flush=True)
        elif c % 10 == 0:
            print("ok", c, n, args, flush=True)
print("cases", cases, "failures", bad)
sys.exit(1 if bad else 0)
