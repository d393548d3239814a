"""Host side of the drop-in: the suffix sorter / start-rank search (CPU), the construct_sa command
line (flag syntax, error behaviour of src/main.cpp:133-246 -- CPU), and construct_sa end to end
against the reference's .sa5 hashes (GPU)."""
import ctypes as C
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import orc
from golden import inputs as gin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "host")
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
CLI = os.path.join(HOST, "construct_sa")


@pytest.fixture(scope="module")
def H():
    so = os.path.join(HOST, "libpsascan_host.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", HOST, "libpsascan_host.so"], stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    L.psh_suffix_array.argtypes = [orc.u8p, C.c_int64, orc.i64p]
    L.psh_sort_halfblock.argtypes = [orc.u8p, C.c_int64, C.c_int64, C.c_int64, orc.u8p,
                                     np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS"), orc.u8p, C.POINTER(C.c_int64),
                                     np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")]
    L.psh_sort_halfblock_ahead.argtypes = [orc.u8p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int64,
                                           np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS"), orc.u8p, C.POINTER(C.c_int64),
                                           np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")]
    L.psh_rank_by_search.argtypes = [orc.u8p, C.c_int64, C.c_int64, C.c_int64,
                                     np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS"), C.c_int64]
    L.psh_rank_by_search.restype = C.c_int64
    return L


def texts():
    rng = np.random.default_rng(17)
    return {
        "rand254": rng.integers(0, 254, 5000, dtype=np.uint8),
        "sig4z": rng.integers(0, 4, 5000, dtype=np.uint8),
        "sig2": rng.integers(97, 99, 4000, dtype=np.uint8),
        "per3": np.frombuffer((b"abc" * 1400)[:4000], np.uint8).copy(),
        "alla": np.full(3000, 97, np.uint8),
        "fib": gin.fib()[:4181].copy(),
        "zeros": np.zeros(2500, np.uint8),
        "runs": np.repeat(rng.integers(0, 5, 300, dtype=np.uint8), rng.integers(1, 40, 300)),
    }


@pytest.mark.parametrize("name", list(texts().keys()))
def test_sais_vs_oracle(H, name):
    t = texts()[name]
    sa = np.zeros(len(t), np.int64)
    H.psh_suffix_array(t, len(t), sa)
    assert np.array_equal(sa, orc.suffix_array(t))


def test_sais_tiny(H):
    for t in (b"a", b"ab", b"ba", b"aa", b"aab", b"baa", b"abab", b"\x00\x00", b"cab"):
        a = orc.as_u8(t).copy()
        sa = np.zeros(len(a), np.int64)
        H.psh_suffix_array(a, len(a), sa)
        assert np.array_equal(sa, orc.suffix_array(a)), t


@pytest.mark.parametrize("name", list(texts().keys()))
def test_sort_halfblock_vs_definition(H, name):
    """Block in the middle of the text: order and gt_begin must follow the WHOLE-text suffix order."""
    t = texts()[name]
    n = len(t)
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    for (b, e) in [(n // 5, n // 2), (0, n // 3), (n // 2, n), (n - 7, n), (3, 4), (n // 2, n // 2 + 2)]:
        m = e - b
        # gt_tail bit v = [text[e+v..) > text[e..)], v in [1, m]
        gt_tail = orc.packbits([0] + [int(e + v < n and isa[e + v] > isa[e]) if e < n else 0 for v in range(1, m + 1)] + [0] * 8)
        psa = np.zeros(m, np.uint32)
        bwt = np.zeros(m, np.uint8)
        gt = np.zeros((m + 31) // 32 + 1, np.uint32)
        i0 = C.c_int64(-1)
        assert H.psh_sort_halfblock(t, n, b, e, gt_tail, psa, bwt, C.byref(i0), gt) == 0
        wpsa, wbwt, wi0, wgt = orc.partial_sa(t, sa, isa, b, e)
        assert np.array_equal(psa.astype(np.int64), wpsa), (name, b, e)
        assert np.array_equal(bwt, wbwt) and i0.value == wi0
        assert np.array_equal(orc.bits(gt.view(np.uint8), m), orc.bits(wgt, m)), (name, b, e)
        for p in (e, min(n, e + 5), n):
            want = int((isa[b:e] < (isa[p] if p < n else -1)).sum())
            assert H.psh_rank_by_search(t, n, b, m, psa, p) == want


@pytest.mark.parametrize("method", [0, 1])
@pytest.mark.parametrize("name", list(texts().keys()) + ["rand254-long", "english"])
def test_lookahead_sorters_vs_definition(H, name, method):
    """construct_sa's look-ahead sorters (no gt bits: comparisons past the half-block's end read on in the text):
    SA-IS with direct comparisons (0) and the prefix-key radix sorter (1).  Either the definition's result, or an
    explicit 'gave up' (1) on periodic text -- never a wrong order."""
    if name == "rand254-long":
        t = np.random.default_rng(5).integers(0, 254, 200_000, dtype=np.uint8)
    elif name == "english":
        words = [b"the", b"of", b"and", b"suffix", b"array", b"block", b"stream", b"gap", b"merge", b"a"]
        rng = np.random.default_rng(9)
        t = np.frombuffer(b" ".join(words[i] for i in rng.integers(0, len(words), 12000)), np.uint8).copy()
    else:
        t = texts()[name]
    n = len(t)
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    gave_up = 0
    for (b, e) in [(n // 5, n // 2), (0, n // 3), (n // 2, n), (n - 7, n), (n // 2, n // 2 + 2), (0, n)]:
        m = e - b
        psa = np.zeros(m, np.uint32)
        bwt = np.zeros(m, np.uint8)
        gt = np.zeros((m + 31) // 32 + 1, np.uint32)
        i0 = C.c_int64(-1)
        rc = H.psh_sort_halfblock_ahead(t, n, b, e, method, 64, psa, bwt, C.byref(i0), gt)
        assert rc in (0, 1)
        if rc == 1:
            gave_up += 1
            continue
        wpsa, wbwt, wi0, wgt = orc.partial_sa(t, sa, isa, b, e)
        assert np.array_equal(psa.astype(np.int64), wpsa), (name, b, e)
        assert np.array_equal(bwt, wbwt) and i0.value == wi0
        assert np.array_equal(orc.bits(gt.view(np.uint8), m), orc.bits(wgt, m)), (name, b, e)
    if name in ("rand254", "rand254-long", "sig4z", "english") and not (name == "english" and method == 1):
        assert gave_up == 0            # ordinary text is sorted ahead (natural language: the prefix-key sorter hands over to SA-IS)
    if name in ("alla", "zeros", "per3"):
        assert gave_up > 0             # periodic text is left to the sequential schedule


def test_lookahead_radix_zero_bytes_and_text_end(H):
    """the prefix-key sorter refines groups with raw 8-byte keys; a suffix that ends inside a key is padded with zeros:
    texts made of zero bytes / with zero runs at the end must still come out in suffix order (shorter = smaller)."""
    rng = np.random.default_rng(1)
    cases = []
    for n in (50, 300, 5000, 60000):
        cases.append(rng.integers(0, 2, n, dtype=np.uint8))
        t = rng.integers(0, 3, n, dtype=np.uint8); t[-20:] = 0; cases.append(t)
        w = [b"the", b"of", b"and", b"a", b"\x00\x00"]
        cases.append(np.frombuffer(b" ".join(w[i] for i in rng.integers(0, 5, n)), np.uint8)[:n].copy())
        t = np.frombuffer((b"abcdefgh" * (n // 8 + 1))[:n], np.uint8).copy(); t[rng.integers(0, n, n // 40)] = 0; cases.append(t)
    sorted_ok = 0
    for t in cases:
        n = len(t)
        sa = orc.suffix_array(t)
        isa = orc.inverse(sa)
        for (b, e) in ((0, n), (n // 3, n), (n // 4, 3 * n // 4), (n - 9, n)):
            m = e - b
            psa, bwt, gt, i0 = np.zeros(m, np.uint32), np.zeros(m, np.uint8), np.zeros((m + 31) // 32 + 1, np.uint32), C.c_int64(-1)
            rc = H.psh_sort_halfblock_ahead(t, n, b, e, 1, 1 << 16, psa, bwt, C.byref(i0), gt)
            assert rc in (0, 1)
            if rc == 0:
                wpsa, wbwt, wi0, wgt = orc.partial_sa(t, sa, isa, b, e)
                assert np.array_equal(psa.astype(np.int64), wpsa), (n, b, e)
                assert np.array_equal(bwt, wbwt) and i0.value == wi0
                sorted_ok += 1
    assert sorted_ok >= len(cases)


def test_lookahead_sais_linear_on_long_runs(H):
    """Text with long zero runs of varying length (zero-padded images): neither periodic nor capped, but a
    per-position bounded comparison costs O(m * run length) there (10 s for this 2 MiB half-block before the
    rename step used the Z-function in the look-ahead schedule too).  Bound the time and check the order."""
    import time
    rng = np.random.default_rng(23)
    parts = []
    while sum(len(p) for p in parts) < (3 << 20):
        parts.append(np.zeros(int(rng.integers(10_000, 30_000)), np.uint8))
        parts.append(rng.integers(1, 200, int(rng.integers(50, 400)), dtype=np.uint8))
    t = np.concatenate(parts)[: 3 << 20].copy()
    n, b, e = len(t), 1 << 19, (1 << 19) + (2 << 20)
    m = e - b
    psa = np.zeros(m, np.uint32); bwt = np.zeros(m, np.uint8); gt = np.zeros((m + 31) // 32 + 1, np.uint32)
    i0 = C.c_int64(-1)
    t0 = time.time()
    rc = H.psh_sort_halfblock_ahead(t, n, b, e, 0, 1 << 16, psa, bwt, C.byref(i0), gt)
    dt = time.time() - t0
    assert rc == 0 and dt < 5.0, (rc, dt)
    # spot-check the order on adjacent pairs (a full oracle sort of 3 MiB takes too long here)
    tb = t.tobytes()
    for k in rng.integers(0, m - 1, 300):
        a, c = b + int(psa[k]), b + int(psa[k + 1])
        assert tb[a:] < tb[c:]
    assert psa[i0.value] == 0


def test_sort_halfblock_rejects_byte_255(H):
    t = np.array([1, 2, 255, 3, 4, 5], np.uint8)
    z = np.zeros(8, np.uint8)
    out = (np.zeros(4, np.uint32), np.zeros(4, np.uint8), np.zeros(2, np.uint32))
    i0 = C.c_int64(0)
    assert H.psh_sort_halfblock(t, 6, 0, 4, z, out[0], out[1], C.byref(i0), out[2]) == -1


# ------------------------------------------------------------------ command line (no GPU needed for these paths)
def cli(args, stdin=""):
    if not os.path.exists(CLI):
        pytest.skip("host/construct_sa not built")
    return subprocess.run([CLI] + args, input=stdin, capture_output=True, text=True, timeout=600)


def test_cli_help_exits_with_failure():
    r = cli(["-h"])                                  # main.cpp:159-161 -- usage + EXIT_FAILURE
    assert r.returncode == 1 and "Usage:" in r.stdout and "--mem=MEM" in r.stdout


def test_cli_missing_file_and_bad_mem(tmp_path):
    r = cli([])
    assert r.returncode == 1 and "FILE not provided" in r.stderr
    r = cli([str(tmp_path / "nope.txt")])
    assert r.returncode == 1 and "does not exist" in r.stderr
    f = tmp_path / "t.txt"
    f.write_bytes(b"banana")
    for bad in ("12x", "k", "1kib", "5Xi"):
        r = cli(["-m", bad, str(f)])
        assert r.returncode == 1 and "parsing RAM limit" in r.stderr, bad
    r = cli(["-m", "0", str(f)])
    assert r.returncode == 1 and "invalid RAM limit" in r.stderr


def test_cli_memory_error_matches_reference(tmp_path):
    """BASELINE configs[0] as literally written (-m 256Ki) is rejected by the reference with
    'not enough memory to start threads' (psascan.hpp:81-86); same here, and no output is left."""
    f = tmp_path / "t.txt"
    f.write_bytes(bytes(gin.rand1m()[:4096]))
    env = dict(os.environ, OMP_NUM_THREADS="8")
    r = subprocess.run([CLI, "-m", "256Ki", str(f)], capture_output=True, text=True, env=env, timeout=60)
    assert r.returncode == 1 and "not enough memory to start threads. You need at least 113MiB" in r.stderr
    assert not os.path.exists(str(f) + ".sa5")


def test_cli_overwrite_prompt(tmp_path):
    f = tmp_path / "t.txt"
    f.write_bytes(b"banana")
    out = tmp_path / "t.txt.sa5"
    out.write_bytes(b"old")
    r = cli([str(f)], stdin="n\n")                   # main.cpp:216-238
    assert r.returncode == 1 and "Overwrite? [y/n]" in r.stdout and out.read_bytes() == b"old"


# ------------------------------------------------------------------ end to end on the GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", list(GOLD.keys() - {"_comment"}))
def test_cli_end_to_end_vs_reference_hashes(tmp_path, name):
    g = GOLD[name]
    f = tmp_path / (name + ".bin")
    f.write_bytes(bytes(gin.GENERATORS[name]()))
    # -m and thread count that give the reference's block structure for this fixture
    threads = {118803662: "8", 29402727: "2"}[g["ram_use"]]
    env = dict(os.environ, OMP_NUM_THREADS=threads)
    r = subprocess.run([CLI, "-m", str(g["ram_use"]), "-v", str(f)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"Max block size = {g['max_block_size']}" in r.stderr
    assert "speed:" in r.stderr and "elapsed time:" in r.stderr       # psascan.hpp:128-130
    data = (tmp_path / (name + ".bin.sa5")).read_bytes()
    assert len(data) == 5 * g["n"]
    assert hashlib.sha256(data).hexdigest() == g["sa5_sha256"]
    assert sorted(os.listdir(tmp_path)) == [name + ".bin", name + ".bin.sa5"]   # no temp files left


@pytest.mark.gpu
def test_cli_64mib_vs_reference_hash(tmp_path):
    """64 MiB of random bytes in four 16 MiB blocks (pass B tails of 16/32/48 MiB: rank-log mode, chunk-free):
    the .sa5 must hash to what the reference's construct_sa wrote for the same input and -m."""
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_large.json")))["rand64m"]
    f = tmp_path / "rand64m.bin"
    f.write_bytes(bytes(gin.LARGE_GENERATORS["rand64m"]()))
    env = dict(os.environ, OMP_NUM_THREADS=str(g["threads"]))
    r = subprocess.run([CLI, "-m", str(g["ram_use"]), "-v", str(f)], capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"Max block size = {g['max_block_size']}" in r.stderr
    h = hashlib.sha256()
    with open(tmp_path / "rand64m.bin.sa5", "rb") as fh:
        for chunk in iter(lambda: fh.read(1 << 24), b""):
            h.update(chunk)
    assert os.path.getsize(tmp_path / "rand64m.bin.sa5") == 5 * g["n"]
    assert h.hexdigest() == g["sa5_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("n,block", [(1, 4), (2, 4), (5, 2), (1000, 64), (4097, 1000), (30000, 30000), (30000, 7000)])
def test_cli_small_shapes(tmp_path, n, block):
    rng = np.random.default_rng(n + block)
    t = rng.integers(0, 3, n, dtype=np.uint8)
    f = tmp_path / "x.bin"
    f.write_bytes(bytes(t))
    for mem in ("1G", "40"):            # last block with / without a right half (ram/10 rule)
        out = tmp_path / f"x_{mem}.sa5"
        r = subprocess.run([CLI, "-m", mem, "--block-size", str(block), "--chains", "32", "-o", str(out), str(f)],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert np.array_equal(orc.sa5_to_sa(np.frombuffer(out.read_bytes(), np.uint8)), orc.suffix_array(t))


@pytest.mark.gpu
def test_cli_check_and_discard(tmp_path):
    """extensions: --check verifies the merged output on the device (permutation sum + sampled adjacent pairs per
    slice), --discard-output runs everything but writes no file; the device memory peak is reported."""
    rng = np.random.default_rng(77)
    t = rng.integers(0, 4, 3 << 20, dtype=np.uint8)
    f = tmp_path / "y.bin"
    f.write_bytes(bytes(t))
    r = subprocess.run([CLI, "-m", "1G", "--block-size", str(1 << 20), "--check=1000", "-v", str(f)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "check: permutation sum ok, 0 of" in r.stderr and "device memory: peak in use" in r.stderr
    want = hashlib.sha256((tmp_path / "y.bin.sa5").read_bytes()).hexdigest()
    os.remove(tmp_path / "y.bin.sa5")
    r = subprocess.run([CLI, "-m", "1G", "--block-size", str(1 << 20), "--check", "--discard-output", str(f)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert sorted(os.listdir(tmp_path)) == ["y.bin"]
    assert len(want) == 64
    # partial SAs in part files next to GAPFILE (the reference's distributed_file): same bytes, files removed at the end
    r = subprocess.run([CLI, "-m", "1G", "--block-size", str(1 << 20), "--spill-psa", "-g", str(tmp_path / "tmp_gap"), "-v", str(f)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert hashlib.sha256((tmp_path / "y.bin.sa5").read_bytes()).hexdigest() == want
    assert sorted(os.listdir(tmp_path)) == ["y.bin", "y.bin.sa5"]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["per3", "zeros", "padded"])
def test_cli_periodic_text_is_not_a_cliff(tmp_path, kind):
    """Periodic / zero-padded input: no chain start comes out of the warm-up, the look-ahead sorter gives up on
    most half-blocks.  The chain starts are found by string search on the device (one launch per pass; used to be one
    launch per chain, 0.4 MiB/s).  8 MiB in 2 MiB blocks must finish in seconds and be exact (closed forms)."""
    import time
    n = 8 << 20
    if kind == "per3":
        t = np.frombuffer((b"abc" * (n // 3 + 1))[:n], np.uint8)
        want = np.concatenate([np.arange(n - 1 - ((n - 1 - r0) % 3), -1, -3) for r0 in (0, 1, 2)])
    elif kind == "zeros":
        t = np.zeros(n, np.uint8)
        want = np.arange(n - 1, -1, -1)
    else:       # a disk image: random sectors between long zero runs
        rng = np.random.default_rng(3)
        t = np.zeros(n, np.uint8)
        for k in range(0, n, 1 << 20):
            t[k + 300000: k + 300000 + 4096] = rng.integers(1, 250, 4096)
        want = None
    f = tmp_path / "p.bin"
    f.write_bytes(bytes(t))
    t0 = time.time()
    r = subprocess.run([CLI, "-m", "8G", "--block-size", str(2 << 20), "--check=2000", "-v", str(f)], capture_output=True, text=True,
                       env=dict(os.environ, OMP_NUM_THREADS="16"), timeout=900)
    dt = time.time() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    assert "check: permutation sum ok, 0 of" in r.stderr
    assert dt < 60, (dt, r.stderr[-1500:])
    if want is not None:
        sa5 = np.fromfile(str(f) + ".sa5", np.uint8).reshape(-1, 5).astype(np.int64)
        pos = sa5[:, 0] | (sa5[:, 1] << 8) | (sa5[:, 2] << 16) | (sa5[:, 3] << 24) | (sa5[:, 4] << 32)
        assert np.array_equal(pos, want)
    print(kind, f"{dt:.1f} s", [l for l in r.stderr.splitlines() if "string search" in l][:3])


@pytest.mark.gpu
@pytest.mark.parametrize("kind,leaf,fanout", [("rand", 5000, 2), ("rand", 3000, 4), ("sig3", 1000, 3), ("sig3", 40000, 16), ("english", 7777, 4),
                                              ("per3", 5000, 4), ("zeros", 4096, 4), ("runs", 2500, 5)])
def test_cli_leaves_merged_on_the_device(tmp_path, kind, leaf, fanout):
    """half-blocks cut into leaves that are suffix-sorted on the host and merged on the device (the in-memory
    pSAscan of the reference, inmem_psascan.hpp:64-304, with the hot path as the merger): every leaf size / fan-out
    must give the oracle's suffix array; periodic text makes the leaf sorter give up and takes the sequential path."""
    rng = np.random.default_rng(leaf + fanout)
    n = 150_001
    if kind == "rand":
        t = rng.integers(0, 255, n, dtype=np.uint8)
    elif kind == "sig3":
        t = rng.integers(0, 3, n, dtype=np.uint8)
    elif kind == "english":
        words = [b"the", b"of", b"and", b"suffix", b"array", b"block", b"stream", b"gap", b"merge", b"a"]
        t = np.frombuffer(b" ".join(words[i] for i in rng.integers(0, len(words), 40000)), np.uint8)[:n].copy()
        n = len(t)
    elif kind == "per3":
        t = np.frombuffer((b"abc" * (n // 3 + 1))[:n], np.uint8).copy()
    elif kind == "zeros":
        t = np.zeros(n, np.uint8)
    else:
        t = np.repeat(rng.integers(0, 5, 8000, dtype=np.uint8), rng.integers(1, 40, 8000))[:n].copy()
        n = len(t)
    f = tmp_path / "x.bin"
    f.write_bytes(bytes(t))
    want = orc.suffix_array(t)
    for block in (n, 60_000):
        out = tmp_path / f"x_{block}.sa5"
        r = subprocess.run([CLI, "-m", "1G", "--block-size", str(block), "--leaf-size", str(leaf), "--fanout", str(fanout), "--check=500", "-v", "-o", str(out), str(f)],
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="4"))
        assert r.returncode == 0, r.stderr[-3000:]
        assert np.array_equal(orc.sa5_to_sa(np.frombuffer(out.read_bytes(), np.uint8)), want), (kind, block)
        if kind in ("rand", "sig3", "english") and block == n:
            assert "merged on the device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("kind,leaf", [("rand", 5000), ("rand", 700), ("sig3", 1000), ("sig3", 40000), ("english", 7777), ("english", 300), ("per3", 5000), ("zeros", 4096),
                                       ("runs", 2500), ("zeros_mix", 3000)])
def test_cli_leaves_merged_in_batches(tmp_path, kind, leaf):
    """the default placement: leaves (16-bit partial SAs only) sorted on the host, merged on the device pairwise with one
    launch sequence per tree level (psg_merge_leaves).  Every leaf size gives the oracle's suffix array; periodic text makes
    the leaf sorter give up and takes the sequential path."""
    rng = np.random.default_rng(leaf)
    n = 150_001
    if kind == "rand":
        t = rng.integers(0, 255, n, dtype=np.uint8)
    elif kind == "sig3":
        t = rng.integers(0, 3, n, dtype=np.uint8)
    elif kind == "zeros_mix":
        t = np.where(rng.random(n) < 0.4, 0, rng.integers(1, 4, n)).astype(np.uint8)
    elif kind == "english":
        words = [b"the", b"of", b"and", b"suffix", b"array", b"block", b"stream", b"gap", b"merge", b"a"]
        t = np.frombuffer(b" ".join(words[i] for i in rng.integers(0, len(words), 40000)), np.uint8)[:n].copy()
        n = len(t)
    elif kind == "per3":
        t = np.frombuffer((b"abc" * (n // 3 + 1))[:n], np.uint8).copy()
    elif kind == "zeros":
        t = np.zeros(n, np.uint8)
    else:
        t = np.repeat(rng.integers(0, 5, 8000, dtype=np.uint8), rng.integers(1, 40, 8000))[:n].copy()
        n = len(t)
    f = tmp_path / "x.bin"
    f.write_bytes(bytes(t))
    want = orc.suffix_array(t)
    for block in (n, 60_000):
        out = tmp_path / f"x_{block}.sa5"
        r = subprocess.run([CLI, "-m", "1G", "--block-size", str(block), "--leaf-size", str(leaf), "--check=500", "-v", "-o", str(out), str(f)],
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="4"))
        assert r.returncode == 0, r.stderr[-3000:]
        assert np.array_equal(orc.sa5_to_sa(np.frombuffer(out.read_bytes(), np.uint8)), want), (kind, block)
        if kind in ("rand", "sig3", "english", "zeros_mix") and block == n:
            assert "levels," in r.stderr and "merged on the device" in r.stderr


@pytest.mark.gpu
def test_cli_device_sort_extension(tmp_path):
    """--device-sort (NOT the default placement): half-blocks sorted by the bench's device sorter; periodic text makes it
    give up and fall back to the host sorter.  Same bytes as the default path either way."""
    rng = np.random.default_rng(5)
    per2 = np.frombuffer(b"\x00\x03" * 11_860, np.uint8).copy()     # found by tests/fuzz_cli.py: a short periodic range used to keep ONE
    per2[20_000] = 9                                                # device thread comparing for minutes (groups below the size limit)
    for name, t in (("rand", rng.integers(0, 200, 300_000, dtype=np.uint8)), ("per", np.frombuffer((b"abcd" * 50_000), np.uint8).copy()), ("per2", per2)):
        f = tmp_path / f"{name}.bin"
        f.write_bytes(bytes(t))
        outs = []
        for extra in ([], ["--device-sort"]):
            out = tmp_path / f"{name}{len(extra)}.sa5"
            r = subprocess.run([CLI, "-m", "1G", "--block-size", "8499" if name == "per2" else "100000", "--check=300", "-v", "-o", str(out), str(f)] + extra,
                               capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append(out.read_bytes())
            if extra:
                assert ("device sufsort" in r.stderr) and (("gave up" in r.stderr) == (name != "rand"))
        assert outs[0] == outs[1]
        assert np.array_equal(orc.sa5_to_sa(np.frombuffer(outs[0], np.uint8)), orc.suffix_array(t))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["rand", "sig3", "per3"])
def test_cli_text_on_host_chunked_tails(tmp_path, kind):
    """--text-on-host: the text is never resident in HBM; every pass uploads its tail in chunks (here 4096 symbols) and
    streams them one after the other with the exact hand-over rank (stream.hpp:104-106 reads the tail from the file per
    pass).  Same suffix array as the oracle's."""
    rng = np.random.default_rng(3)
    n = 90_001
    t = {"rand": lambda: rng.integers(0, 255, n, dtype=np.uint8), "sig3": lambda: rng.integers(0, 3, n, dtype=np.uint8),
         "per3": lambda: np.frombuffer((b"abc" * (n // 3 + 1))[:n], np.uint8).copy()}[kind]()
    f = tmp_path / "x.bin"
    f.write_bytes(bytes(t))
    out = tmp_path / "x.sa5"
    r = subprocess.run([CLI, "-m", "1G", "--block-size", "20000", "--text-on-host", "--tail-chunk", "4096", "-v", "-o", str(out), str(f)],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="4"))
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Text stays in host memory" in r.stderr
    assert np.array_equal(orc.sa5_to_sa(np.frombuffer(out.read_bytes(), np.uint8)), orc.suffix_array(t))
    r = subprocess.run([CLI, "-m", "1G", "--text-on-host", "--check", "-o", str(out), str(f)], input="y\n", capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, OMP_NUM_THREADS="4"))
    assert r.returncode == 0 and "check: permutation sum ok, 0 of" in r.stderr and "(on the host)" in r.stderr, r.stderr[-2000:]   # the check runs on the host then


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["rand", "sig3", "english", "per3"])
def test_cli_hbm_limit_host_tier(tmp_path, kind):
    """--hbm-limit: the external-memory schedule with HBM in the place of the reference's RAM budget.  The text, the gt bits
    (a chunk's words travel with the chunk), the partial SAs and every finished merge bitvector (psg_mbv_spill) live in
    host memory; the merge streams all of them back slice by slice.  Same bytes as the run without a budget and the
    oracle's suffix array; a block that cannot fit the budget is refused up front; with --checkpoint the spilled state
    survives a stop and a restart."""
    rng = np.random.default_rng(21)
    n = 400_003
    if kind == "english":
        words = [b"the", b"of", b"and", b"suffix", b"array", b"block", b"stream", b"gap", b"merge", b"a"]
        t = np.frombuffer(b" ".join(words[i] for i in rng.integers(0, len(words), 120000)), np.uint8)[:n].copy()
        n = len(t)
    else:
        t = {"rand": lambda: rng.integers(0, 255, n, dtype=np.uint8), "sig3": lambda: rng.integers(0, 3, n, dtype=np.uint8),
             "per3": lambda: np.frombuffer((b"abc" * (n // 3 + 1))[:n], np.uint8).copy()}[kind]()
    f = tmp_path / "x.bin"
    f.write_bytes(bytes(t))
    env = dict(os.environ, OMP_NUM_THREADS="4")
    ref, out = tmp_path / "ref.sa5", tmp_path / "x.sa5"
    base = [CLI, "-m", "1G", "--block-size", "70000", "--leaf-size", "9000", "--tail-chunk", "16384", "-v"]
    r = subprocess.run(base + ["-o", str(ref), str(f)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run(base + ["--hbm-limit", "96Mi", "--check=300", "-o", str(out), str(f)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Device memory budget" in r.stderr and "Text stays in host memory" in r.stderr and "merge bitvectors to host memory" in r.stderr
    assert "check: permutation sum ok, 0 of" in r.stderr and "(on the host" in r.stderr
    assert out.read_bytes() == ref.read_bytes()
    assert np.array_equal(orc.sa5_to_sa(np.frombuffer(out.read_bytes(), np.uint8)), orc.suffix_array(t))
    # the default placement with only the merge bitvectors planned into host memory (what a text in many small blocks gets:
    # their total grows with the square of the number of blocks), the partial SAs staying in HBM
    r = subprocess.run(base + ["--check=300", "-o", str(out), str(f)], input="y\n", capture_output=True, text=True, timeout=600, env=dict(env, PSASCAN_MBV_ON_HOST="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    assert "-> host memory" in r.stderr and "merge bitvectors to host memory" in r.stderr and "Text stays in host memory" not in r.stderr
    assert out.read_bytes() == ref.read_bytes()
    # the same budget with --spill-psa: partial SAs AND merge bitvectors in files next to the gap file prefix, mapped for
    # the merge (the reference's part files and gap files: io/distributed_file.hpp, gap_array.hpp:156-182); gone afterwards
    g = tmp_path / "gapdir"
    g.mkdir()
    r = subprocess.run(base + ["--hbm-limit", "96Mi", "--spill-psa", "-g", str(g / "x"), "--check=300", "-o", str(out), str(f)], input="y\n", capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "merge bitvector files" in r.stderr and "mapped for the merge" in r.stderr
    assert out.read_bytes() == ref.read_bytes() and list(g.iterdir()) == []
    # a block that needs more than the budget is refused before anything runs
    r = subprocess.run([CLI, "-m", "1G", "--block-size", "4000000", "--hbm-limit", "64Mi", "-o", str(out), str(f)], input="y\n", capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 1 and "too large for --hbm-limit" in r.stderr
    if kind == "rand":      # stop and restart under the budget: the spilled merge bitvectors and gt bits come back from the checkpoint
        ck = tmp_path / "ck"
        cmd = base + ["--hbm-limit", "96Mi", "--checkpoint", str(ck), "-o", str(out), str(f)]
        r = subprocess.run(cmd + ["--stop-after", "3"], input="y\n", capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 3, r.stderr[-2000:]
        r = subprocess.run(cmd, input="y\n", capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and "Resuming from the checkpoint" in r.stderr, r.stderr[-2000:]
        assert out.read_bytes() == ref.read_bytes()


@pytest.mark.gpu
def test_cli_text_on_host_leaves_through_a_window(tmp_path):
    """--text-on-host with half-blocks cut into leaves: the device merges them seeing the text through a window (the
    half-block + the look-ahead behind it, psg_search_ctx.text_begin/text_end).  A repeat that spans leaves and runs
    past the window (X X Y with |X| = 150 000 and a half-block that ends inside the second X) makes the search give up
    with PSG_EWINDOW; that half-block falls back to the host sorter.  The oracle's suffix array either way."""
    rng = np.random.default_rng(17)
    env = dict(os.environ, OMP_NUM_THREADS="4")
    t1 = rng.integers(0, 200, 120_001, dtype=np.uint8)
    x = rng.integers(0, 250, 150_000, dtype=np.uint8)
    t2 = np.concatenate([x, x, rng.integers(0, 250, 200_000, dtype=np.uint8)])
    for name, t, block, expect_fallback in (("plain", t1, 40_000, False), ("repeat", t2, 400_000, True)):
        f = tmp_path / f"{name}.bin"
        f.write_bytes(bytes(t))
        out = tmp_path / f"{name}.sa5"
        r = subprocess.run([CLI, "-m", "1G", "--block-size", str(block), "--text-on-host", "--tail-chunk", "8192", "--leaf-size", "30000" if expect_fallback else "3000",
                            "-v", "-o", str(out), str(f)], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        assert "Text stays in host memory" in r.stderr and "merged on the device" in r.stderr
        assert ("ran past the text window" in r.stderr) == expect_fallback, r.stderr[-3000:]
        assert np.array_equal(orc.sa5_to_sa(np.frombuffer(out.read_bytes(), np.uint8)), orc.suffix_array(t)), name


@pytest.mark.gpu
def test_cli_edge_inputs(tmp_path):
    """the reference's input constraints and degenerate sizes: byte 255 is fatal (initial_partial_sufsort.hpp:141-146:
    exit status 1, no output left behind), an empty file gives an empty .sa5, one- and two-symbol files work"""
    env = dict(os.environ, OMP_NUM_THREADS="4")
    f = tmp_path / "bad.bin"
    f.write_bytes(bytes([1, 2, 3, 255, 4, 5, 6, 7] * 1000))
    for extra in ([], ["--device-sort"], ["--no-device-merge"]):
        r = subprocess.run([CLI, "-m", "1G", "--block-size", "3000", str(f)] + extra, capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 1 and "255" in r.stderr, r.stderr[-500:]
        assert not (tmp_path / "bad.bin.sa5").exists()
    e = tmp_path / "empty.bin"
    e.write_bytes(b"")
    r = subprocess.run([CLI, "-m", "1G", str(e)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and (tmp_path / "empty.bin.sa5").read_bytes() == b""
    for data in (b"a", b"ab", b"ba", b"aa", b"\x00", b"\x00\x00\x01"):
        g = tmp_path / "tiny.bin"
        g.write_bytes(data)
        if (tmp_path / "tiny.bin.sa5").exists():
            os.remove(tmp_path / "tiny.bin.sa5")
        r = subprocess.run([CLI, "-m", "1G", str(g)], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-500:]
        got = orc.sa5_to_sa(np.frombuffer((tmp_path / "tiny.bin.sa5").read_bytes(), np.uint8))
        assert np.array_equal(got, orc.suffix_array(np.frombuffer(data, np.uint8)))


@pytest.mark.gpu
@pytest.mark.parametrize("kind,spill", [("rand", False), ("per3", False), ("rand", True)])
def test_cli_checkpoint_resume(tmp_path, kind, spill):
    """--checkpoint DIR (the reference has no restart): the run is stopped after 2 and then after 3 more
    of its 7 blocks (--stop-after, exit status 3, as a crash would leave it), started again twice, and the .sa5 is the
    same bytes as an uninterrupted run's.  A checkpoint of another run is refused; a completed run leaves nothing behind."""
    rng = np.random.default_rng(9)
    n = 130_001
    t = rng.integers(0, 5, n, dtype=np.uint8) if kind == "rand" else np.frombuffer((b"abc" * (n // 3 + 1))[:n], np.uint8).copy()
    f = tmp_path / "x.bin"
    f.write_bytes(bytes(t))
    env = dict(os.environ, OMP_NUM_THREADS="4")
    base = [CLI, "-m", "1G", "--block-size", "20000", "-v"]
    ref = tmp_path / "ref.sa5"
    r = subprocess.run(base + ["-o", str(ref), str(f)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    ck = tmp_path / "ck"
    out = tmp_path / "x.sa5"
    # with --spill-psa the part files are first written next to GAPFILE and become the checkpoint's part files
    cmd = base + (["--spill-psa", "-g", str(tmp_path / "gapx")] if spill else []) + ["--checkpoint", str(ck), "-o", str(out), str(f)]
    r = subprocess.run(cmd + ["--stop-after", "2"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 3 and "Process block 2/7" in r.stderr and "Process block 3/7" not in r.stderr, r.stderr[-2000:]
    assert (ck / "ckpt.manifest").exists()
    # another run's checkpoint is refused
    r = subprocess.run(base + ["--block-size", "30000", "--checkpoint", str(ck), "-o", str(out), str(f)], input="y\n", capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 1 and "belongs to another run" in r.stderr
    r = subprocess.run(cmd + ["--stop-after", "3"], input="y\n", capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 3 and "2 of 7 blocks done" in r.stderr and "Process block 3/7" in r.stderr and "Process block 2/7" not in r.stderr, r.stderr[-2000:]
    r = subprocess.run(cmd, input="y\n", capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "5 of 7 blocks done" in r.stderr and "Process block 6/7" in r.stderr and "Process block 5/7" not in r.stderr, r.stderr[-2000:]
    assert out.read_bytes() == ref.read_bytes()
    assert np.array_equal(orc.sa5_to_sa(np.frombuffer(out.read_bytes(), np.uint8)), orc.suffix_array(t))
    assert sorted(os.listdir(ck)) == []
    assert not [x for x in os.listdir(tmp_path) if x.startswith("gapx")]          # no part file left next to GAPFILE either


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["rand", "sig3"])
def test_cli_wide_nodes_and_device_sorted_pieces(tmp_path, kind):
    """half-blocks of 2^32 symbols or more (the 8 GiB halves of configs[3]'s 16 GiB blocks): the merged node carries its
    partial SA as two planes (psg_merge_run_planes, psg_halfblock_from_psa40) and the half-block keeps a high plane; with
    --device-sort such a half-block is sorted in pieces that are merged the same way.  PSASCAN_TEST_WIDE_NODES forces
    the planes at a small size; --leaf-size forces the pieces.  Same bytes as the plain path and the oracle's order."""
    rng = np.random.default_rng(31)
    n = 140_001
    t = rng.integers(0, 255 if kind == "rand" else 3, n, dtype=np.uint8)
    f = tmp_path / "x.bin"
    f.write_bytes(bytes(t))
    want = orc.suffix_array(t)
    for extra, marker in ((["--leaf-size", "6000"], "merged on the device"), (["--device-sort", "--leaf-size", "9000"], "pieces sorted on the device")):
        for block in (n, 50_000):
            out = tmp_path / "x.sa5"
            r = subprocess.run([CLI, "-m", "1G", "--block-size", str(block), "--check=500", "-v", "-o", str(out), str(f)] + extra, input="y\n",
                               capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="4", PSASCAN_TEST_WIDE_NODES="1"))
            assert r.returncode == 0, r.stderr[-3000:]
            assert marker in r.stderr
            assert np.array_equal(orc.sa5_to_sa(np.frombuffer(out.read_bytes(), np.uint8)), want), (kind, extra, block)
