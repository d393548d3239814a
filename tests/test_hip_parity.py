"""Parity of the HIP path (through the C ABI) against the CPU oracle.  Bit-exact: everything
on this path is integer / byte / index work.  Needs a real MI355X: `pytest -m gpu`."""
import hashlib
import json
import os

import numpy as np
import pytest

import orc
from golden import inputs as gin

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "golden.json")))


@pytest.fixture(scope="module")
def A(gpu_lib):
    import psascan_amd.api as api
    return api


def make_text(kind, n, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "rand255":
        return rng.integers(0, 255, n, dtype=np.uint8)
    if kind == "sig4z":
        return rng.integers(0, 4, n, dtype=np.uint8)
    if kind == "dna":
        return np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)]
    if kind == "sig12":
        return rng.integers(40, 52, n, dtype=np.uint8)
    if kind == "alla":
        return np.full(n, 97, np.uint8)
    if kind == "fib":
        return gin.fib()[:n].copy()
    if kind == "per3":
        return np.frombuffer((b"abc" * (n // 3 + 1))[:n], np.uint8).copy()
    if kind == "zeros":
        return np.zeros(n, np.uint8)
    raise KeyError(kind)


KINDS = ["rand255", "sig4z", "dna", "sig12", "alla", "fib", "per3", "zeros"]


# ------------------------------------------------------------------ rank (a3)
@pytest.mark.parametrize("sigma,layout", [(255, 0), (255, 1), (255, 32), (255, 64), (255, 128), (255, 256), (4, 0), (4, 1), (4, 48), (4, -64), (12, 64),
                                          (12, 0), (12, 1), (2, 0), (1, 0), (1, 1), ("runs", 1), ("skew", 1)])
@pytest.mark.parametrize("m", [1, 47, 48, 49, 64, 4095, 4096, 4097, 100003])
def test_rank_query(A, sigma, layout, m):
    rng = np.random.default_rng(m * 7 + (sigma if isinstance(sigma, int) else 5))
    if sigma == "runs":       # BWT-like: runs of one symbol -> dense buckets of rare symbols (overflow pool of the symbol-major layout)
        bwt = np.repeat(rng.integers(0, 200, m // 20 + 1, dtype=np.uint8), rng.integers(1, 60, m // 20 + 1))[:m]
        if len(bwt) < m:
            bwt = np.concatenate([bwt, np.zeros(m - len(bwt), np.uint8)])
    elif sigma == "skew":     # a few frequent symbols (bitmap mode) among many rare ones (list mode)
        bwt = np.where(rng.random(m) < 0.8, rng.integers(0, 3, m), rng.integers(3, 250, m)).astype(np.uint8)
    else:
        bwt = rng.integers(0, sigma, m, dtype=np.uint8)
    if sigma == 12:
        bwt = bwt * 20 + 3
    r = A.rank_build(A.upload(bwt, pad_to=16), m, layout)
    rk = orc.Rank(bwt)
    assert np.array_equal(r.counts, rk.counts())
    qi = np.concatenate([rng.integers(-3, m + 4, 3000), np.arange(-1, min(m, 200) + 2), [m - 1, m, m + 1]]).astype(np.int64)
    qc = rng.integers(0, 256, len(qi)).astype(np.uint8)
    qc[::2] = bwt[rng.integers(0, m, len(qc[::2]))]
    got = r.query(qi, qc)
    want = np.array([rk.rank(i, c) for i, c in zip(qi, qc)], np.int64)
    assert np.array_equal(got, want)


# ------------------------------------------------------------------ one streaming pass (a1,a2,a4)
def _stream_case(t, b, e, tb, te):
    n = len(t)
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    psa, bwt, i0, _ = orc.partial_sa(t, sa, isa, b, e)
    gt_in = orc.packbits([(isa[te - u] if te - u < n else -1) > isa[e] for u in range(te - tb)])
    init = int((isa[b:e] < (isa[te] if te < n else -1)).sum())
    return bwt, i0, gt_in, init


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("chains", [0, 1, 7])
def test_stream_gap_vs_oracle(A, kind, chains):
    n = 9000
    t = make_text(kind, n, 3)
    b, e = n // 9, n // 9 + n // 4
    bwt, i0, gt_in, init = _stream_case(t, b, e, e, n)
    m = e - b
    want_gap, want_gt, want_fin = orc.stream_pass(orc.Rank(bwt), i0, t[e - 1], t, e, n, gt_in, init)
    r = A.rank_build(A.upload(bwt, pad_to=16), m)
    d_text = A.upload(t, pad_to=16)
    T = n - e
    d_gap = A.gap_array(m)
    d_gtout = A.zeros(4 * ((T + 31) // 32 + 1))
    fin, st = A.stream_gap(r, i0, t[e - 1], d_text.at(e), T, A.upload(gt_in, pad_to=8), init, d_gap, d_gtout, chains)
    assert fin == want_fin
    assert np.array_equal(A.download(d_gap, np.uint32, m + 1).astype(np.uint64), want_gap)
    assert np.array_equal(orc.bits(A.download(d_gtout, np.uint8, (T + 7) // 8), T), orc.bits(want_gt, T))
    if chains == 7:
        assert st.n_chains > 1


def test_stream_gap_mid_tail_and_accumulate(A):
    """tail range that does not end at n (pass A shape) + accumulation into a non-zero gap array."""
    n = 20000
    t = make_text("sig4z", n, 5)
    b, mid, e = 1000, 6000, 11000
    bwt, i0, gt_in, init = _stream_case(t, b, mid, mid, e)
    m = mid - b
    T = e - mid
    rk = orc.Rank(bwt)
    want_gap, want_gt, want_fin = orc.stream_pass(rk, i0, t[mid - 1], t, mid, e, gt_in, init)
    r = A.rank_build(A.upload(bwt, pad_to=16), m)
    d_text = A.upload(t, pad_to=16)
    d_gap = A.gap_array(m, fill=5)
    d_gtout = A.zeros(4 * ((T + 31) // 32 + 1))
    fin, st = A.stream_gap(r, i0, t[mid - 1], d_text.at(mid), T, A.upload(gt_in, pad_to=8), init, d_gap, d_gtout, 64)
    assert fin == want_fin and st.n_chains > 8
    assert np.array_equal(A.download(d_gap, np.uint32, m + 1).astype(np.uint64), want_gap + 5)
    assert np.array_equal(orc.bits(A.download(d_gtout, np.uint8, (T + 7) // 8), T), orc.bits(want_gt, T))


@pytest.mark.parametrize("kind", ["rand255", "sig4z", "alla", "fib"])
def test_stream_gap_log_mode(A, kind, monkeypatch):
    """gap increments through the rank log + sorted-window histogram instead of atomics (gap_hist.hip),
    incl. skewed inputs where a single gap slot receives almost everything."""
    monkeypatch.setenv("PSG_GAP_MODE", "log")
    n = 200000 if kind in ("rand255", "sig4z") else 60000
    t = make_text(kind, n, 21)
    b, e = 1000, 1000 + n // 3
    bwt, i0, gt_in, init = _stream_case(t, b, e, e, n)
    m = e - b
    want_gap, want_gt, want_fin = orc.stream_pass(orc.Rank(bwt), i0, t[e - 1], t, e, n, gt_in, init)
    r = A.rank_build(A.upload(bwt, pad_to=16), m)
    T = n - e
    d_gap = A.gap_array(m, fill=3)
    d_gtout = A.zeros(4 * ((T + 31) // 32 + 1))
    d_text, d_gtin = A.upload(t, pad_to=16), A.upload(gt_in, pad_to=8)
    fin, st = A.stream_gap(r, i0, t[e - 1], d_text.at(e), T, d_gtin, init, d_gap, d_gtout, 300)
    assert fin == want_fin and st.hist_ms > 0
    assert np.array_equal(A.download(d_gap, np.uint32, m + 1).astype(np.uint64), want_gap + 3)
    assert np.array_equal(orc.bits(A.download(d_gtout, np.uint8, (T + 7) // 8), T), orc.bits(want_gt, T))


@pytest.mark.parametrize("mode", ["atomic", "log"])
def test_stream_gap_chunked_pass(A, monkeypatch, mode):
    """long tails are streamed in chunks with exact hand-over ranks (PSG_PASS_CHUNK shrunk for the test)"""
    monkeypatch.setenv("PSG_PASS_CHUNK", "4096")
    monkeypatch.setenv("PSG_GAP_MODE", mode)
    n = 60000
    t = make_text("sig4z", n, 17)
    b, e = 300, 9001
    bwt, i0, gt_in, init = _stream_case(t, b, e, e, n)
    m, T = e - b, n - e
    want_gap, want_gt, want_fin = orc.stream_pass(orc.Rank(bwt), i0, t[e - 1], t, e, n, gt_in, init)
    r = A.rank_build(A.upload(bwt, pad_to=16), m)
    d_text, d_gtin = A.upload(t, pad_to=16), A.upload(gt_in, pad_to=8)
    d_gap = A.gap_array(m)
    d_gtout = A.zeros(4 * ((T + 31) // 32 + 1))
    fin, st = A.stream_gap(r, i0, t[e - 1], d_text.at(e), T, d_gtin, init, d_gap, d_gtout, 16)
    assert fin == want_fin and st.rounds >= (T + 4095) // 4096
    assert np.array_equal(A.download(d_gap, np.uint32, m + 1).astype(np.uint64), want_gap)
    assert np.array_equal(orc.bits(A.download(d_gtout, np.uint8, (T + 7) // 8), T), orc.bits(want_gt, T))


@pytest.mark.parametrize("case", ["uniform", "skewed", "uniform-atomic", "uniform-cpl2", "uniform-fresh", "skewed-fresh", "atomic-fresh"])
def test_stream_gap_large_block(A, monkeypatch, case):
    """a block with more than 512 histogram windows (m > 8 Mi) and a tail long enough for the automatic
    rank-log mode: two-level partition + window histograms at a size where both levels are real.
    "skewed": all tail ranks fall into a handful of windows.  Checked against the CPU oracle."""
    import psascan_amd.extras as X
    fresh = case.endswith("-fresh")        # PSG_GAP_UNINITIALIZED: garbage in the gap array on entry
    if case in ("uniform-atomic", "atomic-fresh"):
        monkeypatch.setenv("PSG_GAP_MODE", "atomic")
    if case == "uniform-cpl2":
        monkeypatch.setenv("PSG_CPL", "2")
    mid, T = 9_500_017, 5_000_003
    n = mid + T
    rng = np.random.default_rng(5)
    t = rng.integers(3, 255, n, dtype=np.uint8)
    if case.startswith("skewed"):
        t[mid:] = rng.integers(1, 3, T, dtype=np.uint8)     # tail over {1,2}: ranks cluster at the low end of the block's order
    t[mid - 1] = 0     # gt_in is only read where a tail symbol equals the block's last symbol: never, here
    d_text = A.upload(t, pad_to=16)
    Lh = X.sort_halfblock(d_text, n, 0, mid, want_gt=False)
    lbwt = A.download(Lh["bwt"], np.uint8, mid)
    gt_in = np.zeros((T + 7) // 8 + 8, np.uint8)
    want_gap, want_gt, want_fin = orc.stream_pass(orc.Rank(lbwt), Lh["i0"], 0, t, mid, n, gt_in, 0)
    r = A.rank_build(Lh["bwt"], mid)
    d_gap = A.gap_array(mid, fill=0xDEADBEEF if fresh else 2)
    d_gtout = A.zeros(4 * ((T + 31) // 32 + 4))
    fin, st = A.stream_gap(r, Lh["i0"], 0, d_text.at(mid), T, A.upload(gt_in, pad_to=16), 0, d_gap, d_gtout, 0, fresh_gap=fresh)
    assert fin == want_fin and (st.hist_ms > 0) == ("atomic" not in case)
    got = A.download(d_gap, np.uint32, mid + 1).astype(np.uint64)
    assert np.array_equal(got, want_gap + (0 if fresh else 2))
    assert np.array_equal(orc.bits(A.download(d_gtout, np.uint8, (T + 7) // 8), T), orc.bits(want_gt, T))


# ------------------------------------------------------------------ blocks of >= 2^32 symbols: the code they run, at oracle sizes
# A block of >= 2^32 - 1 symbols runs (a) several superblocks in the rank structure (counts relative to the
# superblock, bases in the LDS table of the pass) and (b) the two-plane rank log + slab split before the
# histogram.  Both sizes are test parameters (PSG_SM_SB_SHIFT / PSG_BLOCK_SB_SHIFT / PSG_LOG_SLAB_SHIFT /
# PSG_LOG_WIDE) so that the same code runs here against the oracle; tests/test_scale_gpu.py runs a real 2^32+ block.
@pytest.mark.parametrize("sigma,layout", [(255, 0), (255, 1), (4, 0), (12, 0), ("runs", 1), ("skew", 1), (1, 0)])
@pytest.mark.parametrize("m", [4097, 100003, 300007])
def test_rank_query_many_superblocks_symbol_major(A, monkeypatch, sigma, layout, m):
    monkeypatch.setenv("PSG_SM_SB_SHIFT", "12" if m < 200000 else "13")     # 4096 (8192) positions per superblock: up to 37 superblocks
    rng = np.random.default_rng(m + 11)
    if sigma == "runs":
        bwt = np.repeat(rng.integers(0, 200, m // 20 + 1, dtype=np.uint8), rng.integers(1, 60, m // 20 + 1))[:m]
        if len(bwt) < m:
            bwt = np.concatenate([bwt, np.zeros(m - len(bwt), np.uint8)])
    elif sigma == "skew":
        bwt = np.where(rng.random(m) < 0.8, rng.integers(0, 3, m), rng.integers(3, 250, m)).astype(np.uint8)
    else:
        bwt = rng.integers(0, sigma, m, dtype=np.uint8)
    r = A.rank_build(A.upload(bwt, pad_to=16), m, layout)
    assert r.device_bytes() > 0
    rk = orc.Rank(bwt)
    assert np.array_equal(r.counts, rk.counts())
    qi = np.concatenate([rng.integers(-3, m + 4, 6000), np.arange(4090, 4100), np.arange(8186, 8200), [m - 1, m, m + 1]]).astype(np.int64)
    qc = rng.integers(0, 256, len(qi)).astype(np.uint8)
    qc[::2] = bwt[rng.integers(0, m, len(qc[::2]))]
    got = r.query(qi, qc)
    want = np.array([rk.rank(i, c) for i, c in zip(qi, qc)], np.int64)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("layout", [32, 64, 128, 256, -64, 48])
def test_rank_query_many_superblocks_blocks(A, monkeypatch, layout):
    """interleaved-block layouts with several superblocks (T1[sb*256+c], superblock-relative counters)"""
    monkeypatch.setenv("PSG_BLOCK_SB_SHIFT", "14")      # 2^14 blocks per superblock (the smallest the build allows)
    B = abs(layout)
    m = 3 * (1 << 14) * B + 12345                        # four superblocks
    rng = np.random.default_rng(layout + 100)
    bwt = rng.integers(0, 4 if layout == 48 else 255, m, dtype=np.uint8)
    r = A.rank_build(A.upload(bwt, pad_to=16), m, layout)
    rk = orc.Rank(bwt)
    assert np.array_equal(r.counts, rk.counts())
    edges = np.concatenate([np.arange(k * (1 << 14) * B - 3, k * (1 << 14) * B + 4) for k in (1, 2, 3)])
    qi = np.concatenate([rng.integers(-3, m + 4, 6000), edges, [m - 1, m, m + 1]]).astype(np.int64)
    qc = rng.integers(0, 256, len(qi)).astype(np.uint8)
    qc[::2] = bwt[rng.integers(0, m, len(qc[::2]))]
    got = r.query(qi, qc)
    want = np.array([rk.rank(i, c) for i, c in zip(qi, qc)], np.int64)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("kind", ["rand255", "sig4z", "dna", "alla", "fib"])
@pytest.mark.parametrize("fresh", [False, True])
def test_stream_gap_wide_log_and_superblocks(A, monkeypatch, kind, fresh):
    """the whole pass as a block of >= 2^32 symbols runs it: superblock-relative rank entries, 40-bit rank log in
    two planes, slab split, per-slab histograms -- against the oracle."""
    monkeypatch.setenv("PSG_SM_SB_SHIFT", "13")
    monkeypatch.setenv("PSG_LOG_WIDE", "1")
    monkeypatch.setenv("PSG_LOG_SLAB_SHIFT", "13")      # slabs of 8192 counters
    monkeypatch.setenv("PSG_GAP_MODE", "log")
    n = 200000 if kind in ("rand255", "sig4z", "dna") else 60000
    t = make_text(kind, n, 31)
    b, e = 777, 777 + n // 3
    bwt, i0, gt_in, init = _stream_case(t, b, e, e, n)
    m = e - b
    want_gap, want_gt, want_fin = orc.stream_pass(orc.Rank(bwt), i0, t[e - 1], t, e, n, gt_in, init)
    r = A.rank_build(A.upload(bwt, pad_to=16), m)
    T = n - e
    d_gap = A.gap_array(m, fill=0xDEADBEEF if fresh else 3)
    d_gtout = A.zeros(4 * ((T + 31) // 32 + 1))
    d_text, d_gtin = A.upload(t, pad_to=16), A.upload(gt_in, pad_to=8)
    fin, st = A.stream_gap(r, i0, t[e - 1], d_text.at(e), T, d_gtin, init, d_gap, d_gtout, 300, fresh_gap=fresh)
    assert fin == want_fin and st.hist_ms > 0
    assert np.array_equal(A.download(d_gap, np.uint32, m + 1).astype(np.uint64), want_gap + (0 if fresh else 3))
    assert np.array_equal(orc.bits(A.download(d_gtout, np.uint8, (T + 7) // 8), T), orc.bits(want_gt, T))


def test_stream_gap_wide_log_chunked_block_layout(A, monkeypatch):
    """wide log + chunked pass + interleaved-block rank with several superblocks (the fallback layout of a huge block)"""
    monkeypatch.setenv("PSG_BLOCK_SB_SHIFT", "14")
    monkeypatch.setenv("PSG_RANK_LAYOUT", "block")
    monkeypatch.setenv("PSG_LOG_WIDE", "1")
    monkeypatch.setenv("PSG_LOG_SLAB_SHIFT", "16")
    monkeypatch.setenv("PSG_GAP_MODE", "log")
    monkeypatch.setenv("PSG_PASS_CHUNK", "65536")
    import psascan_amd.extras as X
    mid, T = (1 << 14) * 32 * 2 + 999, 150_001           # B = 32 -> three superblocks
    n = mid + T
    rng = np.random.default_rng(77)
    t = rng.integers(3, 255, n, dtype=np.uint8)
    t[mid - 1] = 0
    d_text = A.upload(t, pad_to=16)
    Lh = X.sort_halfblock(d_text, n, 0, mid, want_gt=False)
    lbwt = A.download(Lh["bwt"], np.uint8, mid)
    gt_in = np.zeros((T + 7) // 8 + 8, np.uint8)
    want_gap, want_gt, want_fin = orc.stream_pass(orc.Rank(lbwt), Lh["i0"], 0, t, mid, n, gt_in, 0)
    r = A.rank_build(Lh["bwt"], mid, 32)
    d_gap = A.gap_array(mid)
    d_gtout = A.zeros(4 * ((T + 31) // 32 + 4))
    fin, st = A.stream_gap(r, Lh["i0"], 0, d_text.at(mid), T, A.upload(gt_in, pad_to=16), 0, d_gap, d_gtout, 0)
    assert fin == want_fin and st.hist_ms > 0 and st.rounds >= 3
    assert np.array_equal(A.download(d_gap, np.uint32, mid + 1).astype(np.uint64), want_gap)
    assert np.array_equal(orc.bits(A.download(d_gtout, np.uint8, (T + 7) // 8), T), orc.bits(want_gt, T))


@pytest.mark.parametrize("kind", ["rand255", "sig4z", "alla"])
@pytest.mark.parametrize("fresh", [False, True])
@pytest.mark.parametrize("bits", [32, 8])
def test_stream_gap_histogram_behind_the_next_chunk(A, monkeypatch, kind, fresh, bits):
    """chunked pass with the 40-bit log: the partition + histogram of chunk k runs on the side stream while the main
    stream runs chunk k+1's kernel (PSG_HIST_OVERLAP=1; off by default, see DESIGN 3.2: same result either way, equal to the oracle's;
    narrow counters send carries through the excess list from both)."""
    monkeypatch.setenv("PSG_LOG_WIDE", "1")
    monkeypatch.setenv("PSG_GAP_MODE", "log")
    monkeypatch.setenv("PSG_PASS_CHUNK", "30016")
    if bits != 32:
        monkeypatch.setenv("PSG_GAP_COUNTER_BITS", str(bits))
    n = 200000 if kind != "alla" else 90000
    t = make_text(kind, n, 41)
    b, e = 500, 500 + n // 3
    bwt, i0, gt_in, init = _stream_case(t, b, e, e, n)
    m, T = e - b, n - e
    want_gap, want_gt, want_fin = orc.stream_pass(orc.Rank(bwt), i0, t[e - 1], t, e, n, gt_in, init)
    r = A.rank_build(A.upload(bwt, pad_to=16), m)
    d_text, d_gtin = A.upload(t, pad_to=16), A.upload(gt_in, pad_to=8)
    for overlap in ("1", "0"):
        monkeypatch.setenv("PSG_HIST_OVERLAP", overlap)
        d_gap = A.gap_array(m, fill=None if fresh else 0)
        d_gtout = A.zeros(4 * ((T + 31) // 32 + 1))
        fin, st = A.stream_gap(r, i0, t[e - 1], d_text.at(e), T, d_gtin, init, d_gap, d_gtout, 300, fresh_gap=fresh)
        assert fin == want_fin and st.hist_ms > 0 and st.rounds >= 2
        assert np.array_equal(A.gap_values(d_gap, m), want_gap), (kind, fresh, bits, overlap)
        assert np.array_equal(orc.bits(A.download(d_gtout, np.uint8, (T + 7) // 8), T), orc.bits(want_gt, T))


def test_stream_gap_atomic_chunk_after_deferred_histogram(A, monkeypatch):
    """automatic mode choice: a chunk of 2^22 suffixes goes through the log (its histogram deferred to the side
    stream), the short last chunk updates the gap array with atomics -- it has to wait for the histogram before it"""
    monkeypatch.setenv("PSG_LOG_WIDE", "1")
    monkeypatch.setenv("PSG_HIST_OVERLAP", "1")
    monkeypatch.setenv("PSG_PASS_CHUNK", str(1 << 22))
    rng = np.random.default_rng(12)
    m, T = 70_000, (1 << 22) + 4097
    n = m + T
    t = rng.integers(0, 4, n, dtype=np.uint8)
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    _, bwt, i0, _ = orc.partial_sa(t, sa, isa, 0, m)
    pos = n - np.arange(T)                                     # bit u <-> position n - u
    gt_in = orc.packbits(np.where(pos < n, isa[np.minimum(pos, n - 1)], -1) > isa[m])
    init = 0                                                   # the empty suffix is the smallest
    want_gap, want_gt, want_fin = orc.stream_pass(orc.Rank(bwt), i0, t[m - 1], t, m, n, gt_in, init)
    r = A.rank_build(A.upload(bwt, pad_to=16), m)
    d_gap = A.gap_array(m, fill=None)
    d_gtout = A.zeros(4 * ((T + 31) // 32 + 1))
    d_text, d_gtin = A.upload(t, pad_to=16), A.upload(gt_in, pad_to=8)
    fin, st = A.stream_gap(r, i0, t[m - 1], d_text.at(m), T, d_gtin, init, d_gap, d_gtout, 0, fresh_gap=True)
    assert fin == want_fin and st.hist_ms > 0 and st.rounds == 2
    assert np.array_equal(A.gap_values(d_gap, m), want_gap)
    assert np.array_equal(orc.bits(A.download(d_gtout, np.uint8, (T + 7) // 8), T), orc.bits(want_gt, T))


# ------------------------------------------------------------------ excess list of the gap counters (a5, a7)
@pytest.mark.parametrize("kind", ["alla", "sig4z", "skew", "rand255"])
@pytest.mark.parametrize("mode", ["atomic", "log", "wide"])
@pytest.mark.parametrize("bits", [8, 16])
def test_gap_excess_list_with_narrow_counters(A, monkeypatch, kind, mode, bits):
    """The reference counts in u8 and appends the slot to an excess list whenever a counter wraps (update.hpp:88-96;
    value = count + 256 * #entries, gap_array.hpp:116-124).  Here the counters are 32 bits wide, so the same path
    only runs at > 2^32 suffixes per slot; PSG_GAP_COUNTER_BITS narrows them to 8 / 16 bits so that it runs here:
    atomic and histogram producers, two accumulating passes, then the consumers (values, gap -> bitvector)."""
    monkeypatch.setenv("PSG_GAP_COUNTER_BITS", str(bits))
    monkeypatch.setenv("PSG_GAP_MODE", "atomic" if mode == "atomic" else "log")
    if mode == "wide":
        monkeypatch.setenv("PSG_LOG_WIDE", "1")
        monkeypatch.setenv("PSG_LOG_SLAB_SHIFT", "12")
    n = 120000
    if kind == "skew":
        rng = np.random.default_rng(6)
        t = rng.integers(3, 200, n, dtype=np.uint8)
        t[n // 3:] = rng.integers(1, 3, n - n // 3, dtype=np.uint8)     # the tail's ranks cluster in a few slots
    else:
        t = make_text(kind, n, 9)
    b, e = 300, 300 + n // 4
    bwt, i0, gt_in, init = _stream_case(t, b, e, e, n)
    m, T = e - b, n - e
    want_gap, want_gt, want_fin = orc.stream_pass(orc.Rank(bwt), i0, t[e - 1], t, e, n, gt_in, init)
    assert want_gap.max() >= (1 << bits) or kind != "alla"     # alla: every tail suffix lands in one slot
    r = A.rank_build(A.upload(bwt, pad_to=16), m)
    d_text, d_gtin = A.upload(t, pad_to=16), A.upload(gt_in, pad_to=8)
    d_gtout = A.zeros(4 * ((T + 31) // 32 + 1))
    d_gap = A.gap_array(m, fill=None)
    fin, st = A.stream_gap(r, i0, t[e - 1], d_text.at(e), T, d_gtin, init, d_gap, d_gtout, 300, fresh_gap=True)
    assert fin == want_fin
    cells = A.download(d_gap, np.uint32, m + 1)
    assert cells.max() < (1 << bits)
    assert np.array_equal(A.gap_values(d_gap, m), want_gap)
    d_bv = A.zeros(4 * ((m + T + 31) // 32 + 2))
    assert A.gap_to_bitvector(d_gap, m, d_bv, m + T) == m + T
    bv, nb = orc.gap_to_bitvector(want_gap, m)
    assert np.array_equal(orc.bits(A.download(d_bv, np.uint8, (nb + 7) // 8), nb), orc.bits(bv, nb))
    # a second pass accumulates on top (same tail again): values double
    A.stream_gap(r, i0, t[e - 1], d_text.at(e), T, d_gtin, init, d_gap, d_gtout, 300)
    assert np.array_equal(A.gap_values(d_gap, m), 2 * want_gap)


@pytest.mark.parametrize("bits", [8, 16, 32])
def test_split_gap_with_values_beyond_the_counter_width(A, bits):
    """compute_right_gap / compute_left_gap on a block gap whose values exceed the counters (the reference widens to
    u16 and replays the excess, gap_array.hpp:386-529): 255, 256, 65535, 70000, 131079 and -- with the production
    width -- 2^32 + 5; checked against the oracle, which is pinned against the reference on the same shapes."""
    ml, mr = 4000, 2400
    bs = ml + mr
    rng = np.random.default_rng(8)
    bvb = rng.permutation(np.array([0] * ml + [1] * mr, np.uint8))
    bv = np.concatenate([orc.packbits(bvb), np.zeros(8, np.uint8)])
    gap = rng.integers(0, 3, bs + 1).astype(np.uint64)
    gap[3] = 255; gap[4] = 256; gap[10] = 70000; gap[11] = 65535; gap[64] = 131072 + 7; gap[bs] = 300
    if bits == 32:
        gap[2000] = 2 ** 32 + 5; gap[5000] = 2 ** 33 + 1
    T = int(gap.sum())
    d_gap = A.gap_array_from_values(gap, bits)
    assert np.array_equal(A.gap_values(d_gap, bs), gap)
    d_bv = A.upload(bv, pad_to=8)
    mbvL = A.DeviceBuffer(4 * ((bs + T + 31) // 32 + 1))
    mbvR = A.DeviceBuffer(4 * ((mr + T + 31) // 32 + 1))
    A.split_gap(d_gap, d_bv, ml, mr, T, mbvL, mbvR)
    assert np.array_equal(A.download(A.mbv_to_gap(mbvL, bs + T, ml), np.uint64, ml + 1), orc.left_gap(gap, bv, ml, mr))
    assert np.array_equal(A.download(A.mbv_to_gap(mbvR, mr + T, mr), np.uint64, mr + 1), orc.right_gap(gap, bv, ml, mr))


# ------------------------------------------------------------------ K8: start ranks by string search (a14)
def _gt_cmp_end_bits(isa, n, e):
    """bit (n - j) = [text[j..) > text[e..)] for j in (e, n]; bit 0 (j = n) = 0"""
    bits = np.zeros(n - e + 1, np.uint8)
    for j in range(e + 1, n):
        bits[n - j] = isa[j] > isa[e]
    return orc.packbits(list(bits) + [0] * 64)


@pytest.mark.parametrize("kind", KINDS + ["runs"])
def test_initial_ranks_vs_definition(A, kind):
    """psg_initial_ranks against the definition: one part and two parts, positions all over the tail,
    the end of the text included; periodic texts make every comparison run to the block end (gt bits decide)."""
    n = 6000
    if kind == "runs":
        rng = np.random.default_rng(8)
        t = np.repeat(rng.integers(0, 3, 400, dtype=np.uint8), rng.integers(1, 60, 400))[:n].copy()
        n = len(t)
    else:
        t = make_text(kind, n, 12)
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    b, mid, e = n // 10, n // 10 + n // 5, n // 10 + n // 2
    d_text = A.upload(t, pad_to=16)
    psaL, _, _, _ = orc.partial_sa(t, sa, isa, b, mid)
    psaR, _, _, _ = orc.partial_sa(t, sa, isa, mid, e)
    dL, dR = A.upload(psaL.astype(np.uint32)), A.upload(psaR.astype(np.uint32))
    gt = A.upload(_gt_cmp_end_bits(isa, n, e), pad_to=8)
    rng = np.random.default_rng(1)
    pos = np.concatenate([[e, e + 1, n - 1, n], rng.integers(e, n + 1, 400)]).astype(np.int64)
    rank_of = lambda p, lo, hi: int((isa[lo:hi] < (isa[p] if p < n else -1)).sum())
    # pass B shape: both halves
    sc = A.search_ctx(d_text, n, e, gt, [(b, mid - b, dL, None), (mid, e - mid, dR, None)])
    assert np.array_equal(A.initial_ranks(sc, pos), [rank_of(p, b, e) for p in pos])
    # pass A shape: the left half only, comparisons still run to e; positions from mid on
    posA = np.concatenate([[mid, mid + 1, e], rng.integers(mid, n + 1, 300)]).astype(np.int64)
    scA = A.search_ctx(d_text, n, e, gt, [(b, mid - b, dL, None)])
    gotA = A.initial_ranks(scA, posA)
    assert np.array_equal(gotA, [rank_of(p, b, mid) for p in posA])
    # ... and equal to the oracle's search, which tests/test_oracle.py::test_ref_initial_ranks pins to the reference's own
    # em_compute_initial_ranks (both overloads); here with the comparison boundary of this context (e, gt w.r.t. e)
    gt_n = orc.packbits([0] + [int((isa[n - u] if u > 0 else -1) > isa[e]) for u in range(1, n - e)] + [0] * 64)
    assert np.array_equal(gotA, [orc.initial_rank(t, b, mid, psaL, e, gt_n, int(p)) for p in posA])
    # last block: comparisons run to the end of the text, no gt bits
    psaZ, _, _, _ = orc.partial_sa(t, sa, isa, e, n - 100)
    scZ = A.search_ctx(d_text, n, n, None, [(e, n - 100 - e, A.upload(psaZ.astype(np.uint32)), None)])
    posZ = np.arange(n - 100, n + 1).astype(np.int64)
    assert np.array_equal(A.initial_ranks(scZ, posZ), [rank_of(p, e, n - 100) for p in posZ])


def test_search_with_two_text_windows(A):
    """psg_search_ctx.d_text2: a rank of the block-per-GPU schedule holds its own block (+ a look-ahead) and a piece of a
    far block -- the suffixes of the block are read from the first window, the searched positions from the second one.
    Ranks equal the whole-text search; a comparison that would leave either window raises PSG_EWINDOW."""
    rng = np.random.default_rng(29)
    x = rng.integers(0, 250, 8_000, dtype=np.uint8)
    t = np.concatenate([rng.integers(0, 250, 6_000, dtype=np.uint8), x, rng.integers(0, 250, 30_000, dtype=np.uint8), x, rng.integers(0, 250, 9_000, dtype=np.uint8)])
    n = len(t)                                              # x at 6000..14000 and at 44000..52000
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    b, e = 2_000, 12_000                                    # the block ends inside the first copy of x
    psa, _, _, _ = orc.partial_sa(t, sa, isa, b, e, want_gt=False)
    d_psa = A.upload(psa.astype(np.uint32))
    positions = np.array([44_000, 44_001, 47_000, 50_000, 53_000], np.int64)   # inside / behind the second copy
    want = np.array([int((isa[b:e] < isa[p]).sum()) for p in positions])
    for w1_end, w2, ok in ((24_000, (44_000, 56_000), True), (24_000, (44_000, 51_000), False), (13_000, (44_000, 56_000), False)):
        win1 = A.upload(t[b:w1_end], pad_to=64)
        win2 = A.upload(t[w2[0]:w2[1]], pad_to=64)
        sc = A.search_ctx(win1, n, n, None, [(b, e - b, d_psa, None)], window=(b, w1_end), window2=(win2, w2[0], w2[1]))
        if ok:
            assert np.array_equal(A.initial_ranks(sc, positions), want)
        else:
            with pytest.raises(Exception, match="window"):
                A.initial_ranks(sc, positions)


@pytest.mark.parametrize("kind", ["alla", "per3", "fib", "zeros", "repeats"])
@pytest.mark.parametrize("mode", ["atomic", "log"])
def test_stream_gap_repetitive_text_resolves_in_one_round(A, monkeypatch, kind, mode):
    """Text whose chain starts the warm-up cannot determine (periodic text; English with injected long repeats):
    with a search context every start rank comes from the string search and the pass is ONE kernel launch
    (was: one launch per chain).  Without one, PSG_FAIL_IF_UNRESOLVED reports it instead of serialising."""
    from psascan_amd._lib import PsgError
    monkeypatch.setenv("PSG_GAP_MODE", mode)
    n = 40000
    if kind == "repeats":
        rng = np.random.default_rng(4)
        t = rng.integers(97, 105, n, dtype=np.uint8)
        rep = t[1000:9000].copy()                       # an 8000-symbol repeat: in the block and twice in the tail
        t[22000:30000] = rep
        t[31000:39000] = rep
    else:
        t = make_text(kind, n, 2)
    b, e = 500, 500 + n // 3
    bwt, i0, gt_in, init = _stream_case(t, b, e, e, n)
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    m, T = e - b, n - e
    want_gap, want_gt, want_fin = orc.stream_pass(orc.Rank(bwt), i0, t[e - 1], t, e, n, gt_in, init)
    r = A.rank_build(A.upload(bwt, pad_to=16), m)
    d_text, d_gtin = A.upload(t, pad_to=16), A.upload(gt_in, pad_to=8)
    psa, _, _, _ = orc.partial_sa(t, sa, isa, b, e)
    sc = A.search_ctx(d_text, n, e, A.upload(_gt_cmp_end_bits(isa, n, e), pad_to=8), [(b, m, A.upload(psa.astype(np.uint32)), None)])
    d_gap = A.gap_array(m)
    d_gtout = A.zeros(4 * ((T + 31) // 32 + 1))
    with pytest.raises(PsgError) as ei:
        A.stream_gap(r, i0, t[e - 1], d_text.at(e), T, d_gtin, init, d_gap, d_gtout, 96, fail_if_unresolved=True)
    assert ei.value.code == A.PSG_EUNRESOLVED
    A.lib().psg_memset(d_gap.ptr, 0, d_gap.nbytes)
    fin, st = A.stream_gap(r, i0, t[e - 1], d_text.at(e), T, d_gtin, init, d_gap, d_gtout, 96, search=sc, tail_begin_abs=e)
    assert st.unresolved > 0 and st.rounds == 1 and st.n_chains > 32
    assert fin == want_fin
    assert np.array_equal(A.download(d_gap, np.uint32, m + 1).astype(np.uint64), want_gap)
    assert np.array_equal(orc.bits(A.download(d_gtout, np.uint8, (T + 7) // 8), T), orc.bits(want_gt, T))


def test_search_through_a_text_window(A):
    """psg_search_ctx.text_begin / text_end: only a window of the text is on the device (a text that stays in host
    memory); ranks equal the whole-text search as long as no comparison leaves the window, PSG_EWINDOW otherwise."""
    rng = np.random.default_rng(23)
    x = rng.integers(0, 250, 30_000, dtype=np.uint8)
    t = np.concatenate([rng.integers(0, 250, 5_000, dtype=np.uint8), x, x, rng.integers(0, 250, 20_000, dtype=np.uint8)])
    n = len(t)
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    b, e = 1_000, 40_000                                    # the part ends inside the second copy of x
    psa, _, _, _ = orc.partial_sa(t, sa, isa, b, e, want_gt=False)
    d_psa = A.upload(psa.astype(np.uint32))
    positions = np.array([e, e + 1, e + 777, 50_000], np.int64)
    want = np.array([int((isa[b:e] < isa[p]).sum()) for p in positions])
    whole = A.search_ctx(A.upload(t, pad_to=16), n, n, None, [(b, e - b, d_psa, None)])
    assert np.array_equal(A.initial_ranks(whole, positions), want)
    for w_end, ok in ((n, True), (70_000, True), (52_000, False)):      # lcp of text[5000..) and text[35000..) is 30 000: 35 000 + 30 000 > 52 000
        win = A.upload(t[b:w_end], pad_to=64)
        sc = A.search_ctx(win, n, n, None, [(b, e - b, d_psa, None)], window=(b, w_end))
        if ok:
            assert np.array_equal(A.initial_ranks(sc, positions), want)
        else:
            with pytest.raises(Exception, match="window"):
                A.initial_ranks(sc, positions)


# ------------------------------------------------------------------ in-memory pSAscan pieces (leaves merged on the device)
@pytest.mark.parametrize("kind", ["rand255", "sig4z", "alla", "fib", "per3"])
def test_subranges_merged_into_a_partial_sa(A, kind):
    """inmem_psascan.hpp:64-304 with the hot path as the merger: the partial SAs of consecutive sub-ranges (whole-text
    order) + their gap arrays w.r.t. the sub-ranges to their right inside the range -> psg_merge_run_u32 gives the
    range's partial SA, psg_halfblock_from_psa its BWT / i0 / gt_begin -- all equal to the oracle's for the range."""
    n = 8000
    t = make_text(kind, n, 6)
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    b, e = 700, 6100
    cuts = [b, 1900, 1901, 3300, 4800, e]
    hbs = []
    for h in range(len(cuts) - 1):
        x0, x1 = cuts[h], cuts[h + 1]
        psa, _, _, _ = orc.partial_sa(t, sa, isa, x0, x1, want_gt=False)
        mbv = None
        if x1 < e:
            ranks = np.searchsorted(np.sort(isa[x0:x1]), isa[x1:e])          # gap w.r.t. the rest of the RANGE only
            g = np.bincount(ranks, minlength=x1 - x0 + 1).astype(np.uint64)
            bv, nb = orc.gap_to_bitvector(g, x1 - x0)
            mbv = A.upload(bv[: (nb + 7) // 8])
        hbs.append({"beg": x0 - b, "size": x1 - x0, "psa_lo": A.upload(psa.astype(np.uint32)), "psa_hi": None, "mbv": mbv})
    plan = A.MergePlan(hbs)
    d_psa = A.DeviceBuffer(4 * (e - b) + 16)
    A.merge_run_u32(plan, 0, e - b, d_psa)
    want_psa, want_bwt, want_i0, want_gt = orc.partial_sa(t, sa, isa, b, e)
    assert np.array_equal(A.download(d_psa, np.uint32, e - b).astype(np.int64), want_psa)
    d_text = A.upload(t, pad_to=16)
    sc = A.search_ctx(d_text, n, n, None, [])            # comparisons read on in the text
    d_bwt, i0, d_gt = A.halfblock_from_psa(sc, b, e - b, d_psa)
    assert i0 == want_i0 and np.array_equal(A.download(d_bwt, np.uint8, e - b), want_bwt)
    assert np.array_equal(orc.bits(A.download(d_gt, np.uint8, (e - b + 7) // 8), e - b), orc.bits(want_gt, e - b))
    # the same as two planes (ranges of 2^32 positions or more: configs[3]'s 8 GiB half-blocks)
    d_lo, d_hi = A.DeviceBuffer(4 * (e - b) + 16), A.DeviceBuffer(e - b + 16)
    A.merge_run_planes(plan, 0, e - b, d_lo, d_hi)
    assert np.array_equal(A.download(d_lo, np.uint32, e - b).astype(np.int64), want_psa) and not A.download(d_hi, np.uint8, e - b).any()
    d_bwt, i0, d_gt = A.halfblock_from_psa(sc, b, e - b, d_lo, d_psa_hi=d_hi)
    assert i0 == want_i0 and np.array_equal(A.download(d_bwt, np.uint8, e - b), want_bwt)
    assert np.array_equal(orc.bits(A.download(d_gt, np.uint8, (e - b + 7) // 8), e - b), orc.bits(want_gt, e - b))
    # values beyond 32 bits: the same sub-ranges placed at virtual offsets of 3 * 2^32 + ... inside an enclosing range
    # (the merge never reads the text), a sub-range's own values given with a high plane as well
    big = 3 << 32
    hbs2 = []
    for h, hb in enumerate(hbs):
        d = dict(hb, beg=hb["beg"] + big * h)
        if h == 2:
            k = hb["size"]
            d["psa_hi"] = A.upload(np.full(k, 5, np.uint8), pad_to=16)
        hbs2.append(d)
    plan2 = A.MergePlan(hbs2)
    A.merge_run_planes(plan2, 0, e - b, d_lo, d_hi)
    got = A.download(d_lo, np.uint32, e - b).astype(np.int64) + (A.download(d_hi, np.uint8, e - b).astype(np.int64) << 32)
    owner = np.searchsorted(np.array(cuts[1:]) - b, want_psa, side="right")
    assert np.array_equal(got, want_psa + big * owner + np.where(owner == 2, 5 << 32, 0))
    out5 = A.DeviceBuffer(5 * (e - b) + 16)
    plan2.run(0, e - b, out5)
    assert np.array_equal(orc.sa5_to_sa(A.download(out5, np.uint8, 5 * (e - b))), got)


def test_bits_rank1_and_memory_queries(A):
    rng = np.random.default_rng(3)
    nbits = 300_007
    bits = (rng.random(nbits) < 0.3).astype(np.uint8)
    d = A.upload(np.concatenate([np.packbits(bits, bitorder="little"), np.zeros(8, np.uint8)]), pad_to=8)
    pos = np.concatenate([[0, 1, 31, 32, 4095, 4096, 4097, nbits - 1, nbits], rng.integers(0, nbits + 1, 500)]).astype(np.int64)
    cs = np.concatenate([[0], np.cumsum(bits.astype(np.int64))])
    assert np.array_equal(A.bits_rank1(d, nbits, pos), cs[pos])
    in_use, peak, reserved = A.mem_stats()
    free, total = A.device_memory()
    assert 0 < in_use <= peak <= reserved <= total and free <= total
    pa = A.PinnedArray(1 << 20, np.uint32)
    pa.array[:] = np.arange(1 << 20, dtype=np.uint32)
    buf = A.upload(pa.array)                              # pinned source: one DMA, no staging
    assert np.array_equal(A.download(buf, np.uint32, 1 << 20), pa.array)
    pa.free()


def test_background_download(A):
    """psg_d2h_begin / psg_copy_wait: device buffers drain into pageable host arrays on worker threads while the library
    stream runs kernels; with free_src the buffer goes back to the allocator when it is drained (in use drops)."""
    rng = np.random.default_rng(8)
    srcs = [rng.integers(0, 1 << 32, k, dtype=np.uint32) for k in (1, 1000, (40 << 20) // 4 + 3, (97 << 20) // 4 + 1)]
    bufs = [A.upload(x, pad_to=16) for x in srcs]
    in_use0 = A.mem_stats()[0]
    dls = [A.BackgroundDownload(b, np.uint32, len(x), free_src=(i % 2 == 1)) for i, (b, x) in enumerate(zip(bufs, srcs))]
    t = rng.integers(0, 255, 1 << 20, dtype=np.uint8)       # the library's own stream works meanwhile
    r = A.rank_build(A.upload(t, pad_to=16), len(t))
    r.free()
    for dl, x in zip(dls, srcs):
        assert np.array_equal(dl.wait(), x)
    assert A.mem_stats()[0] <= in_use0 - srcs[3].nbytes
    assert A.BackgroundDownload(None, np.uint8, 0).wait().size == 0
    # and up: psg_h2d_begin
    ups = []
    for x in srcs:
        d = A.DeviceBuffer(x.nbytes + 16)
        ups.append((A.BackgroundUpload(d, x), x))
    r = A.rank_build(A.upload(t, pad_to=16), len(t))
    r.free()
    for up, x in ups:
        assert np.array_equal(A.download(up.wait(), np.uint32, len(x)), x)


def test_device_allocator_arena(A, gpu_lib):
    """psg_malloc/psg_free go through the arena of runtime.hip (best fit, split, coalesce for blocks >= 1 MiB,
    size classes below): live blocks never overlap, whatever the order of frees and the mix of sizes."""
    rng = np.random.default_rng(42)
    live = {}

    def alloc(tag):
        nbytes = int(rng.choice([4096, 70_000, (1 << 20) + 8, 3 << 20, (7 << 20) + 123, 33 << 20, 130 << 20]))
        b = A.DeviceBuffer(nbytes)
        assert gpu_lib.psg_memset(b.ptr, tag & 255, nbytes) == 0
        live[tag] = (b, nbytes)

    tag = 0
    for _ in range(24):
        alloc(tag); tag += 1
    for rnd in range(6):
        for t in list(live)[:: 2 + rnd % 2]:            # free every 2nd / 3rd block
            live.pop(t)[0].free()
        for _ in range(10):
            alloc(tag); tag += 1
        A.sync()
        for t, (b, nbytes) in live.items():             # every live block still holds its own byte everywhere sampled
            for off in (0, nbytes // 2, nbytes - 1):
                assert int(A.download(b, np.uint8, 1, off)[0]) == (t & 255), (t, off)
    spans = sorted((b.ptr, b.ptr + nbytes) for b, nbytes in live.values())
    assert all(spans[k][1] <= spans[k + 1][0] for k in range(len(spans) - 1))
    for b, _ in live.values():
        b.free()


def test_stream_gap_ex_contract(A, gpu_lib):
    """psg_stream_gap_ex: unknown flags are rejected; an empty tail with PSG_GAP_UNINITIALIZED leaves an all-zero gap array"""
    import ctypes as C
    from psascan_amd._lib import StreamStatsC
    t = make_text("sig12", 5000, 4)
    bwt, i0, gt_in, init = _stream_case(t, 100, 2100, 2100, 5000)
    m = 2000
    r = A.rank_build(A.upload(bwt, pad_to=16), m)
    d_text, d_gtin = A.upload(t, pad_to=16), A.upload(gt_in, pad_to=8)
    d_gap = A.gap_array(m, fill=0xABCD1234)
    fin, st = C.c_int64(0), StreamStatsC()
    rc = gpu_lib.psg_stream_gap_ex(r.h, i0, int(t[2099]), d_text.at(2100), 2900, 0, d_gtin.ptr, init, d_gap.ptr, None, 0, 6, C.byref(fin), C.byref(st))
    assert rc != 0 and b"flag" in gpu_lib.psg_last_error()
    rc = gpu_lib.psg_stream_gap_ex(r.h, i0, int(t[2099]), d_text.at(2100), 0, 0, d_gtin.ptr, 7, d_gap.ptr, None, 0, 1, C.byref(fin), C.byref(st))
    assert rc == 0 and fin.value == 7
    assert not A.download(d_gap, np.uint32, m + 1).any()


def test_stream_gap_overflow_checking_mode(A, monkeypatch):
    """the kernel variant used for tails of >= 2^32 suffixes (returning atomics + overflow flag)"""
    monkeypatch.setenv("PSG_GAP_MODE", "ovf")
    n = 30000
    t = make_text("sig12", n, 8)
    b, e = 500, 9000
    bwt, i0, gt_in, init = _stream_case(t, b, e, e, n)
    m, T = e - b, n - e
    want_gap, want_gt, want_fin = orc.stream_pass(orc.Rank(bwt), i0, t[e - 1], t, e, n, gt_in, init)
    r = A.rank_build(A.upload(bwt, pad_to=16), m)
    d_text, d_gtin = A.upload(t, pad_to=16), A.upload(gt_in, pad_to=8)
    d_gap = A.gap_array(m)
    d_gtout = A.zeros(4 * ((T + 31) // 32 + 1))
    fin, st = A.stream_gap(r, i0, t[e - 1], d_text.at(e), T, d_gtin, init, d_gap, d_gtout, 100)
    assert fin == want_fin and st.hist_ms == 0
    assert np.array_equal(A.download(d_gap, np.uint32, m + 1).astype(np.uint64), want_gap)
    # a counter at 2^32-1 keeps counting: it wraps and leaves a carry in the excess list (gap_array.hpp:79-88)
    hot = int(np.argmax(want_gap))
    g = np.zeros(m + 1, np.uint64); g[hot] = 0xFFFFFFFF
    d_g = A.gap_array_from_values(g)
    A.stream_gap(r, i0, t[e - 1], d_text.at(e), T, d_gtin, init, d_g, d_gtout, 100)
    assert np.array_equal(A.gap_values(d_g, m), want_gap + g)
    assert int(A.download(d_g, np.uint32, m + 1)[hot]) == (0xFFFFFFFF + int(want_gap[hot])) & 0xFFFFFFFF


@pytest.mark.parametrize("m,nlog,skew", [(100, 5000, 0), (66666, 133504, 0), (70000, 300, 0), (1 << 20, 1 << 22, 0),
                                         (9_000_000, 3_000_000, 0), (20_000_000, 6_000_000, 1), (5000, 2_000_000, 2)])
def test_gap_hist_from_log(A, gpu_lib, m, nlog, skew):
    """hand-written two-level partition + LDS window histograms vs numpy bincount
    (one level when <= 512 windows, two levels above; skewed logs: hot windows / one hot counter)."""
    from psascan_amd._lib import check
    rng = np.random.default_rng(m + nlog)
    if skew == 0:
        v = rng.integers(0, m + 1, nlog).astype(np.uint32)
    elif skew == 1:
        v = np.where(rng.random(nlog) < 0.7, rng.integers(12_345_000, 12_345_900, nlog), rng.integers(0, m + 1, nlog)).astype(np.uint32)
    else:
        v = np.where(rng.random(nlog) < 0.95, 4321, rng.integers(0, m + 1, nlog)).astype(np.uint32)
    v[rng.integers(0, nlog, nlog // 50)] = 0xFFFFFFFF
    d_log = A.upload(v)
    d_gap = A.upload(np.full(m + 1, 2, np.uint32))
    check(gpu_lib.psgx_gap_hist(d_log.ptr, nlog, m, d_gap.ptr))
    want = np.bincount(v[v != 0xFFFFFFFF], minlength=m + 1).astype(np.uint32) + 2
    assert np.array_equal(A.download(d_gap, np.uint32, m + 1), want)


W = 32768   # counters per histogram window (gap_hist.hip: WSIZE)


@pytest.mark.parametrize("nwin_m1", [511, 512, 513, 1023, 1024, 1025, 1533, 1534, 1535, 2556, 2557, 4000])
@pytest.mark.parametrize("tweak", [-1, 0, 5])
def test_gap_hist_window_geometry(A, gpu_lib, nwin_m1, tweak):
    """window counts around the points where the partition changes shape: one level up to 512 windows; above,
    511 level-1 bins of 2^k windows plus a top bin that takes the remainder (k grows at 511*2^k + 512)."""
    from psascan_amd._lib import check
    m = nwin_m1 * W + tweak
    nlog = 400_000
    rng = np.random.default_rng(m)
    v = rng.integers(0, m + 1, nlog).astype(np.uint32)
    v[:2000] = m                      # the last counter, owned by the top bin
    v[2000:4000] = 0
    v[rng.integers(0, nlog, nlog // 64)] = 0xFFFFFFFF
    d_log = A.upload(v)
    d_gap = A.upload(np.full(m + 1, 1, np.uint32))
    check(gpu_lib.psgx_gap_hist(d_log.ptr, nlog, m, d_gap.ptr))
    want = np.bincount(v[v != 0xFFFFFFFF], minlength=m + 1).astype(np.uint32) + 1
    assert np.array_equal(A.download(d_gap, np.uint32, m + 1), want)


@pytest.mark.parametrize("kind,nparts", [("sig4z", 3), ("rand255", 4), ("alla", 2)])
def test_multi_gpu_building_blocks(A, kind, nparts):
    """stream_gap_log over tail sub-ranges -> log_partition -> (exchange emulated on one device)
    -> gap_hist per slice -> gap_slice_to_bits -> sum -> bits_not  ==  oracle gap array / bitvector."""
    from psascan_amd import distributed as D
    n = 120000 if kind != "alla" else 40000
    t = make_text(kind, n, 33)
    b, e = 2000, 2000 + n // 3
    bwt, i0, gt_all, init = _stream_case(t, b, e, e, n)
    m = e - b
    T = n - e
    want_gap, want_gt, _ = orc.stream_pass(orc.Rank(bwt), i0, t[e - 1], t, e, n, gt_all, init)
    want_bv, nbits = orc.gap_to_bitvector(want_gap, m)
    r = A.rank_build(A.upload(bwt, pad_to=16), m)
    d_text = A.upload(t, pad_to=16)
    d_gt_all = A.upload(gt_all, pad_to=8)
    cuts = D.tail_cuts(e, n, nparts)
    parts, offs_all, vb = [], [], None
    gt_got = np.zeros(T, np.uint8)
    for rk in range(nparts):                         # "rank" rk streams its range
        tb_r, te_r = cuts[rk], cuts[rk + 1]
        ctx = D.context_len(te_r, n) if kind != "alla" else 0
        L = te_r + ctx - tb_r
        d_gt_in = A.zeros(4 * ((L + 31) // 32 + 2))
        A.bitcopy(d_gt_in, 0, d_gt_all, n - (te_r + ctx), L)
        d_gt_out = A.zeros(4 * ((te_r - tb_r + 31) // 32 + 2))
        if te_r + ctx == n:
            start = 0
        elif kind == "alla":                          # repetitive text: no context, exact start rank from the definition
            sa = orc.suffix_array(t); isa = orc.inverse(sa)
            start = int((isa[b:e] < isa[te_r]).sum())
        else:
            start = -1
        log, nlog, fin, st = A.stream_gap_log(r, i0, t[e - 1], d_text.at(tb_r), te_r - tb_r, d_gt_in, start, d_gt_out, 64, ctx)
        gt_got[n - te_r: n - tb_r] = orc.bits(A.download(d_gt_out, np.uint8, (te_r - tb_r + 7) // 8), te_r - tb_r)
        d_part = A.DeviceBuffer(4 * max(nlog, 1))
        offs, vb_r = A.log_partition(log, nlog, m, nparts, d_part)
        assert vb is None or vb == vb_r
        vb = vb_r
        assert offs[-1] == te_r - tb_r
        parts.append(A.download(d_part, np.uint32, offs[-1]))
        offs_all.append(offs)
    assert np.array_equal(gt_got, orc.bits(want_gt, T))
    bits_sum = np.zeros((nbits + 31) // 32 + 2, np.uint32)
    totals = []
    slices = []
    for d in range(nparts):                          # "rank" d receives its parts
        recv = np.concatenate([parts[s][offs_all[s][d]: offs_all[s][d + 1]] for s in range(nparts)])
        base = vb[d]
        count = max(0, min(vb[d + 1], m + 1) - base)
        d_slice = A.zeros(4 * max(count, 1))
        A.gap_hist(A.upload(recv) if len(recv) else None, len(recv), base, count, d_slice)
        got = A.download(d_slice, np.uint32, count)
        assert np.array_equal(got.astype(np.uint64), want_gap[base: base + count])
        totals.append(len(recv)); slices.append((d_slice, base, count))
    assert sum(totals) == T
    for d, (d_slice, base, count) in enumerate(slices):
        d_bits = A.zeros(4 * len(bits_sum))
        A.gap_slice_to_bits(d_slice, base, count, m, sum(totals[:d]), d_bits)
        part_bits = A.download(d_bits, np.uint32, len(bits_sum))
        assert not (bits_sum & part_bits).any()      # disjoint -> sum == or
        bits_sum += part_bits
    d_all = A.upload(bits_sum)
    A.bits_not(d_all, nbits)
    assert np.array_equal(orc.bits(A.download(d_all, np.uint8, (nbits + 7) // 8), nbits), orc.bits(want_bv, nbits))


def test_stream_gap_empty_tail(A):
    bwt = np.array([0, 1, 2, 1], np.uint8)
    r = A.rank_build(A.upload(bwt, pad_to=16), 4)
    d_gap = A.gap_array(4)
    fin, _ = A.stream_gap(r, 0, 1, None, 0, None, 3, d_gap, None)
    assert fin == 3
    assert not A.download(d_gap, np.uint32, 5).any()


# ------------------------------------------------------------------ bitvector / BWT merge / split (a5-a9)
@pytest.mark.parametrize("kind", KINDS)
def test_block_steps_vs_oracle(A, kind):
    n = 30000
    t = make_text(kind, n, 11)
    b, e = n // 7, n // 7 + n // 2 + 1
    mid = b + (e - b) // 2
    ml, mr, bs, T = mid - b, e - mid, e - b, n - e
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    lpsa, lbwt, li0, _ = orc.partial_sa(t, sa, isa, b, mid)
    rpsa, rbwt, ri0, rgt = orc.partial_sa(t, sa, isa, mid, e)
    initA = int((isa[b:mid] < isa[e]).sum())
    gapA, gtA, _ = orc.stream_pass(orc.Rank(lbwt), li0, t[mid - 1], t, mid, e, rgt, initA)
    bv, nb = orc.gap_to_bitvector(gapA, ml)
    bbwt, bi0 = orc.merge_bwt(lbwt, rbwt, li0, ri0, t[mid - 1], bv)
    gt_in = orc.packbits([(isa[n - u] if n - u < n else -1) > isa[e] for u in range(T)])
    gapB, _, _ = orc.stream_pass(orc.Rank(bbwt), bi0, t[e - 1], t, e, n, gt_in, 0)
    rg, lg = orc.right_gap(gapB, bv, ml, mr), orc.left_gap(gapB, bv, ml, mr)
    # device
    d_gapA = A.gap_array_from_values(gapA)
    d_bv = A.zeros(4 * ((bs + 31) // 32 + 2))
    assert A.gap_to_bitvector(d_gapA, ml, d_bv, bs) == bs
    assert np.array_equal(orc.bits(A.download(d_bv, np.uint8, (bs + 7) // 8), bs), orc.bits(bv, bs))
    d_out = A.DeviceBuffer(bs + 16)
    got_i0 = A.merge_bwt(A.upload(lbwt, 16), A.upload(rbwt, 16), ml, mr, li0, ri0, t[mid - 1], d_bv, d_out)
    assert got_i0 == bi0
    assert np.array_equal(A.download(d_out, np.uint8, bs), bbwt)
    d_gapB = A.gap_array_from_values(gapB)
    mbvL = A.DeviceBuffer(4 * ((bs + T + 31) // 32 + 1))
    mbvR = A.DeviceBuffer(4 * ((mr + T + 31) // 32 + 1))
    A.split_gap(d_gapB, d_bv, ml, mr, T, mbvL, mbvR)
    got_lg = A.download(A.mbv_to_gap(mbvL, bs + T, ml), np.uint64, ml + 1)
    got_rg = A.download(A.mbv_to_gap(mbvR, mr + T, mr), np.uint64, mr + 1)
    assert np.array_equal(got_lg, lg)
    assert np.array_equal(got_rg, rg)
    # vbyte of the gap file (a10)
    d_vals = A.upload(lg)
    d_vb, nbytes = A.vbyte_encode(d_vals, len(lg))
    assert np.array_equal(A.download(d_vb, np.uint8, nbytes), orc.vbyte_encode(lg))


def test_vbyte_large_values(A):
    vals = np.array([0, 1, 127, 128, 300, 16383, 16384, 2 ** 32, 2 ** 40 - 1, 2 ** 63] * 300, np.uint64)
    d_vb, nb = A.vbyte_encode(A.upload(vals), len(vals))
    assert np.array_equal(A.download(d_vb, np.uint8, nb), orc.vbyte_encode(vals))


def test_mbv_spill_in_the_background_equals_the_synchronous_one(A):
    """psg_mbv_spill_begin + psg_copy_wait + psg_mbv_spill_finish leave the same words and rank samples in host memory
    as psg_mbv_spill, the bits behind nbits cleared (the device buffer carries garbage there)"""
    rng = np.random.default_rng(5)
    for nbits in (1, 31, 4096, 4097, 1_000_003):
        raw = rng.integers(0, 256, (nbits + 31) // 32 * 4 + 16, dtype=np.uint8)
        w1, s1 = A.mbv_spill(A.upload(raw), nbits)
        w2, s2 = A.MbvSpill(A.upload(raw), nbits).wait()
        assert np.array_equal(w1, w2) and np.array_equal(s1, s2)
        bits = orc.bits(raw, nbits)
        assert int(s2[-1]) == int(bits.sum()) and np.array_equal(orc.bits(w2.view(np.uint8), nbits), bits)
        assert not orc.bits(w2.view(np.uint8), len(w2) * 32)[nbits:].any()


def test_thread_binding_and_arena_stats(A):
    """psgx_bind_threads_near_device (in a child process: it re-binds every thread of its process): the CPUs afterwards are
    a subset of the CPUs before, the reported node is -1 or a node of this host; psgx_arena_stats counts what the arena
    fetched from the driver"""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import ctypes as C, os, sys
        sys.path.insert(0, %r)
        import psascan_amd
        L = psascan_amd.lib(0)
        before = os.sched_getaffinity(0)
        node = C.c_int(-7)
        assert L.psgx_bind_threads_near_device(C.byref(node)) == 0
        after = os.sched_getaffinity(0)
        assert after <= before and len(after) > 0, (before, after)
        assert node.value == -1 or os.path.isdir("/sys/devices/system/node/node%%d" %% node.value), node.value
        if node.value >= 0:
            assert after < before or len(before) == len(after)
        from psascan_amd import api
        d = api.DeviceBuffer(64 << 20)
        sec, segs, nbytes = C.c_double(-1), C.c_int64(-1), C.c_int64(-1)
        assert L.psgx_arena_stats(C.byref(sec), C.byref(segs), C.byref(nbytes)) == 0
        assert segs.value >= 1 and nbytes.value >= 64 << 20 and sec.value >= 0
        print("BIND_OK", node.value, len(before), len(after))
    """ % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "BIND_OK" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]


def test_output_check_reports_pairs_it_could_not_decide(A):
    """the sampled order check of --check follows a pair for 2^24 symbols; what agrees that far (periodic text) is counted
    as undecided -- reported, neither "in order" nor "out of order" -- and a wrong order is still found elsewhere"""
    import psascan_amd.extras as X
    n = 17 << 20
    d_text = A.upload(np.full(n, 97, np.uint8))
    sa = np.arange(n - 1, -1, -1, dtype=np.int64)                 # a^n: shorter suffixes first
    sa5 = np.zeros((n, 5), np.uint8)
    for k in range(5):
        sa5[:, k] = (sa >> (8 * k)) & 255
    d_sa5 = A.upload(sa5.reshape(-1))
    bad, total, und = X.check_sa5_ex(d_text, n, d_sa5, n, samples=2048, seed=3)
    assert bad == 0 and total == n * (n - 1) // 2
    assert 0 < und < 2048 // 4                                      # pairs (k, k+1) share k+1 symbols: the last 1/17 of the slots
    sa5[[5, 6]] = sa5[[6, 5]]                                       # two short suffixes swapped: a decidable pair out of order
    bad2, _, _ = X.check_sa5_ex(d_text, n, A.upload(sa5.reshape(-1)), 16, samples=64, seed=1)
    assert bad2 > 0


def test_bitcopy_and_popcount(A):
    rng = np.random.default_rng(4)
    src = rng.integers(0, 256, 4000, dtype=np.uint8)
    for (db, sb, nb) in [(0, 0, 31999), (5, 0, 1000), (0, 7, 1000), (37, 91, 20001), (64, 64, 64), (31, 1, 1), (100, 3, 31)]:
        dst0 = rng.integers(0, 256, 4100, dtype=np.uint8)
        d_dst, d_src = A.upload(dst0), A.upload(src)
        A.bitcopy(d_dst, db, d_src, sb, nb)
        want = orc.bits(dst0, 4100 * 8).copy()
        want[db:db + nb] = orc.bits(src, 4000 * 8)[sb:sb + nb]
        assert np.array_equal(orc.bits(A.download(d_dst, np.uint8, 4100), 4100 * 8), want)
    assert A.popcount(A.upload(src), 31991) == int(orc.bits(src, 31991).sum())


# ------------------------------------------------------------------ final merge (a11, a12)
@pytest.mark.parametrize("cuts", [[0, 5000], [0, 2500, 5000], [0, 700, 1500, 1501, 2600, 3333, 4100, 5000],
                                  list(range(0, 5001, 250)), list(range(0, 5001, 100))])   # 50 half-blocks: beyond the cursor kernel's 48
@pytest.mark.parametrize("kernel", ["levels", "cursor"])
def test_merge_vs_oracle(A, cuts, kernel, monkeypatch):
    """kernel: the one with the tile-cursor pre-pass (default for 3..48 half-blocks) or the level-by-level one"""
    if kernel == "levels":
        monkeypatch.setenv("PSG_MERGE_CHAIN", "1")
    rng = np.random.default_rng(13)
    t = rng.integers(0, 3, 5000, dtype=np.uint8)
    n = len(t)
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    hbs, begs, sizes, psas, gaps = [], [], [], [], []
    for h in range(len(cuts) - 1):
        b, e = cuts[h], cuts[h + 1]
        psa, _, _, _ = orc.partial_sa(t, sa, isa, b, e, want_gt=False)
        ranks = np.searchsorted(np.sort(isa[b:e]), isa[e:n])      # r_B(p) for every p >= e
        g = np.bincount(ranks, minlength=e - b + 1).astype(np.uint64)
        begs.append(b); sizes.append(e - b); psas.append(psa); gaps.append(g if e < n else None)
        mbv = None
        if e < n:
            bv, nb = orc.gap_to_bitvector(g, e - b)
            assert nb == n - b
            mbv = A.upload(bv[: (nb + 7) // 8])
        hbs.append({"beg": b, "size": e - b, "psa_lo": A.upload(psa.astype(np.uint32)), "psa_hi": None, "mbv": mbv})
    want = orc.merge(begs, sizes, psas, gaps)
    d_out = A.merge_half_blocks(hbs)
    assert np.array_equal(A.download(d_out, np.uint8, 5 * n), want)
    # ranged output (multi-GPU partitioning of the output)
    plan = A.MergePlan(hbs)
    d_part = A.DeviceBuffer(5 * 1234 + 8)
    plan.run(777, 1234, d_part)
    assert np.array_equal(A.download(d_part, np.uint8, 5 * 1234), want[5 * 777: 5 * (777 + 1234)])


@pytest.mark.parametrize("pinned", [False, True])
@pytest.mark.parametrize("cuts", [[0, 30000], [0, 14000, 30000], [0, 700, 1500, 1501, 9600, 13333, 24100, 30000],
                                  list(range(0, 30001, 1500))])
def test_merge_stream_vs_oracle(A, cuts, pinned):
    """merge<T> with the partial SAs in host memory (the reference streams them from part files, merge.hpp:72-81):
    slices of 2048 outputs -> 15 slices, every half-block's PSA is consumed in pieces; bytes equal the oracle's."""
    rng = np.random.default_rng(14)
    n = cuts[-1]
    t = rng.integers(0, 3, n, dtype=np.uint8)
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    hbs, begs, sizes, psas, gaps, keep = [], [], [], [], [], []
    for h in range(len(cuts) - 1):
        b, e = cuts[h], cuts[h + 1]
        psa, _, _, _ = orc.partial_sa(t, sa, isa, b, e, want_gt=False)
        ranks = np.searchsorted(np.sort(isa[b:e]), isa[e:n])
        g = np.bincount(ranks, minlength=e - b + 1).astype(np.uint64)
        begs.append(b); sizes.append(e - b); psas.append(psa); gaps.append(g if e < n else None)
        mbv = None
        if e < n:
            bv, nb = orc.gap_to_bitvector(g, e - b)
            mbv = A.upload(bv[: (nb + 7) // 8])
        lo = psa.astype(np.uint32)
        if pinned:
            pa = A.PinnedArray(len(lo), np.uint32)
            pa.array[:] = lo
            keep.append(pa)
            lo = pa.array
        hbs.append({"beg": b, "size": e - b, "psa_lo": lo, "psa_hi": None, "mbv": mbv})
    want = orc.merge(begs, sizes, psas, gaps)
    got = np.zeros(5 * n, np.uint8)
    calls = []

    def sink(view, first, cnt):
        got[5 * first: 5 * (first + cnt)] = view
        calls.append((first, cnt))

    d_text = A.upload(t, pad_to=16)
    st, chk = A.merge_stream(hbs, 2048, sink, check_text=d_text, n=n, samples_per_slice=512, seed=3)
    assert np.array_equal(got, want)
    assert calls == [(k * 2048, min(2048, n - k * 2048)) for k in range((n + 2047) // 2048)]
    assert chk == ((n * (n - 1) // 2) % (1 << 64), 0)
    assert st.slices == len(calls) and st.h2d_bytes == 4 * n and st.d2h_bytes == 5 * n
    # output dropped on the device (bench mode): only the check comes back
    st2, chk2 = A.merge_stream(hbs, 4096, None, check_text=d_text, n=n, samples_per_slice=512, seed=4)
    assert chk2 == chk and st2.d2h_bytes == 0
    # merge bitvectors spilled to host memory (psg_mbv_spill; the reference reads its gap files back during the merge,
    # merge.hpp:80,145): every slice uploads the words of every level it touches -- same bytes
    nh = [sum(sizes[k:]) for k in range(len(sizes))]
    spilled = []
    for k, hb in enumerate(hbs):
        hb2 = dict(hb)
        if hb["mbv"] is not None:
            words, samp = A.mbv_spill(hb["mbv"], nh[k])
            assert int(samp[-1]) == nh[k] - sizes[k] and samp[0] == 0
            hb2["mbv"], hb2["mbv_host"] = None, (words, samp)
        spilled.append(hb2)
    got3 = np.zeros(5 * n, np.uint8)

    def sink3(view, first, cnt):
        got3[5 * first: 5 * (first + cnt)] = view
    st3, chk3 = A.merge_stream(spilled, 2048, sink3, check_text=d_text, n=n, samples_per_slice=512, seed=3)
    assert np.array_equal(got3, want) and chk3 == chk and (st3.h2d_bytes > 4 * n or len(cuts) == 2)


def test_merge_stream_high_byte(A):
    psa0 = np.array([3, 2 ** 32 + 5, 7], np.uint64)
    psa1 = np.array([1, 0], np.uint64)
    bv = orc.packbits([0, 1, 0, 1, 0])
    hbs = [{"beg": 10, "size": 3, "psa_lo": (psa0 & np.uint64(0xFFFFFFFF)).astype(np.uint32), "psa_hi": (psa0 >> np.uint64(32)).astype(np.uint8), "mbv": A.upload(bv)},
           {"beg": 2 ** 39, "size": 2, "psa_lo": psa1.astype(np.uint32), "psa_hi": None, "mbv": None}]
    out = []
    A.merge_stream(hbs, 2048, lambda v, f, c: out.append(v.copy()))
    assert list(orc.sa5_to_sa(np.concatenate(out))) == [13, 2 ** 39 + 1, 2 ** 32 + 15, 2 ** 39, 17]


def test_merge_high_byte(A):
    """half-block offsets beyond 2^32 (psa_hi) and beg beyond 2^32: uint40 packing."""
    psa0 = np.array([3, 2 ** 32 + 5, 7], np.uint64)
    psa1 = np.array([1, 0], np.uint64)
    bv = orc.packbits([0, 1, 0, 1, 0])             # order: a0 b0 a1 b1 a2
    hbs = [{"beg": 10, "size": 3, "psa_lo": A.upload((psa0 & np.uint64(0xFFFFFFFF)).astype(np.uint32)),
            "psa_hi": A.upload((psa0 >> np.uint64(32)).astype(np.uint8)), "mbv": A.upload(bv)},
           {"beg": 2 ** 39, "size": 2, "psa_lo": A.upload(psa1.astype(np.uint32)), "psa_hi": None, "mbv": None}]
    got = orc.sa5_to_sa(A.download(A.merge_half_blocks(hbs), np.uint8, 25))
    assert list(got) == [13, 2 ** 39 + 1, 2 ** 32 + 15, 2 ** 39, 17]


@pytest.mark.parametrize("kernel", ["levels", "cursor"])
@pytest.mark.parametrize("cuts", [[0, 900, 2500, 3600, 5000], list(range(0, 5001, 200))])
def test_merge_high_planes_many_half_blocks(A, cuts, kernel, monkeypatch):
    """the general merge kernels with high planes in play (HI = true): values of up to 40 bits from the planes and from the
    half-blocks' offsets, 4 and 25 half-blocks, some without a high plane; whole output, a sub-range, and both planes of
    psg_merge_run_planes against the oracle's merge of the same numbers"""
    if kernel == "levels":
        monkeypatch.setenv("PSG_MERGE_CHAIN", "1")
    rng = np.random.default_rng(31)
    n = cuts[-1]
    t = rng.integers(0, 4, n, dtype=np.uint8)
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    hbs, begs, sizes, psas, gaps = [], [], [], [], []
    for h in range(len(cuts) - 1):
        b, e = cuts[h], cuts[h + 1]
        psa, _, _, _ = orc.partial_sa(t, sa, isa, b, e, want_gt=False)
        val = psa.astype(np.uint64) + (np.uint64(rng.integers(0, 200)) << np.uint64(32)) * np.uint64(h % 3 != 1)   # every third half-block: no high plane
        val[:: 7] += np.uint64(1) << np.uint64(32)
        if h % 3 == 1:
            val = psa.astype(np.uint64)
        g = np.bincount(np.searchsorted(np.sort(isa[b:e]), isa[e:n]), minlength=e - b + 1).astype(np.uint64)
        mbv = None
        if e < n:
            bv, nb = orc.gap_to_bitvector(g, e - b)
            mbv = A.upload(bv[: (nb + 7) // 8])
        beg = b + ((2 * h) << 32)                  # (half-block offsets are non-decreasing)
        begs.append(beg); sizes.append(e - b); psas.append(val); gaps.append(g if e < n else None)
        hbs.append({"beg": beg, "size": e - b, "psa_lo": A.upload((val & np.uint64(0xFFFFFFFF)).astype(np.uint32)),
                    "psa_hi": None if h % 3 == 1 else A.upload((val >> np.uint64(32)).astype(np.uint8)), "mbv": mbv})
    want = orc.merge(begs, sizes, psas, gaps)
    assert np.array_equal(A.download(A.merge_half_blocks(hbs), np.uint8, 5 * n), want)
    plan = A.MergePlan(hbs)
    d_part = A.DeviceBuffer(5 * 2049 + 8)
    plan.run(1500, 2049, d_part)
    assert np.array_equal(A.download(d_part, np.uint8, 5 * 2049), want[5 * 1500: 5 * (1500 + 2049)])
    d_lo, d_hi = A.DeviceBuffer(4 * n + 16), A.DeviceBuffer(n + 16)
    A.merge_run_planes(plan, 0, n, d_lo, d_hi)
    w = orc.sa5_to_sa(want)
    assert np.array_equal(A.download(d_lo, np.uint32, n).astype(np.int64), w & 0xFFFFFFFF)
    assert np.array_equal(A.download(d_hi, np.uint8, n).astype(np.int64), w >> 32)


# ------------------------------------------------------------------ whole path vs the reference's hashes
def oracle_sorter(sa, isa):
    def sorter(text, beg, end, gt_tail):
        psa, bwt, i0, gt = orc.partial_sa(text, sa, isa, beg, end)
        return {"psa": psa, "bwt": bwt, "i0": i0, "gt_begin": gt}
    return sorter


@pytest.mark.parametrize("name", list(GOLD.keys() - {"_comment"}))
def test_pipeline_vs_reference_hashes(A, name):
    from psascan_amd import pipeline
    g = GOLD[name]
    t = gin.GENERATORS[name]()
    sa = orc.suffix_array(t)
    st = []
    out = pipeline.construct_sa5(t, g["max_block_size"], g["ram_use"], oracle_sorter(sa, orc.inverse(sa)),
                                 max_chains=256, stats=st)
    assert hashlib.sha256(bytes(out)).hexdigest() == g["sa5_sha256"]
    assert len(st) > 0


@pytest.mark.parametrize("n,mb", [(1, 4), (2, 4), (3, 2), (17, 4), (1000, 64), (1000, 999), (1001, 77), (5000, 5000)])
def test_pipeline_small_shapes(A, n, mb):
    from psascan_amd import pipeline
    rng = np.random.default_rng(n * 31 + mb)
    t = rng.integers(0, 3, n, dtype=np.uint8)
    sa = orc.suffix_array(t)
    for ram in (int(mb * 5.2) + 1, 30, 10 * 1000000):
        out = pipeline.construct_sa5(t, mb, ram, oracle_sorter(sa, orc.inverse(sa)), max_chains=16)
        assert np.array_equal(orc.sa5_to_sa(out), sa)


# ------------------------------------------------------------------ batched leaf merging (in-memory pSAscan, inmem_psascan.hpp:64-304)
def _leaf_case(A, t, b, e, bounds, psa_bytes):
    """leaves of [b, e) cut at `bounds` (absolute), each one's partial SA from the oracle -> psg_merge_leaves"""
    n = len(t)
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    dt = np.uint16 if psa_bytes == 2 else np.uint32
    parts = []
    for lb, le in zip(bounds[:-1], bounds[1:]):
        psa, _, _, _ = orc.partial_sa(t, sa, isa, int(lb), int(le), want_gt=False)
        parts.append(psa.astype(dt))
    d_text = A.upload(t, pad_to=64)
    sc = A.search_ctx(d_text, n, n, None, [])
    d_psa, d_bwt, i0, d_gt, st = A.merge_leaves(sc, b, e - b, bounds, A.upload(np.concatenate(parts)), psa_bytes)
    want_psa, want_bwt, want_i0, want_gt = orc.partial_sa(t, sa, isa, b, e)
    assert i0 == want_i0
    assert np.array_equal(A.download(d_psa, np.uint32, e - b).astype(np.int64), want_psa)
    assert np.array_equal(A.download(d_bwt, np.uint8, e - b), want_bwt)
    assert np.array_equal(orc.bits(A.download(d_gt, np.uint8, (e - b + 7) // 8), e - b), orc.bits(want_gt, e - b))
    return st


@pytest.mark.parametrize("kind", ["rand255", "sig4z", "dna", "sig12", "zeros_mix", "alla", "per3", "fib", "zeros"])
@pytest.mark.parametrize("nleaves,psa_bytes", [(1, 4), (2, 2), (3, 4), (8, 2), (13, 4), (64, 2)])
def test_merge_leaves_vs_oracle(A, kind, nleaves, psa_bytes):
    """psg_merge_leaves == the oracle's partial SA / BWT / i0 / gt bits of the range, for even and odd leaf counts (a node
    without a partner is carried up a level), leaves of unequal sizes, ranges in the middle and at the end of the text."""
    n = 30_011
    rng = np.random.default_rng(nleaves * 3 + psa_bytes)
    t = np.where(rng.random(n) < 0.3, 0, rng.integers(1, 6, n)).astype(np.uint8) if kind == "zeros_mix" else make_text(kind, n, 5)
    for b, e in ((1_003, 21_777), (9_000, n)):
        cuts = np.sort(rng.choice(np.arange(b + 1, e), nleaves - 1, replace=False)) if nleaves > 1 else np.array([], np.int64)
        bounds = np.concatenate([[b], cuts, [e]]).astype(np.int64)
        st = _leaf_case(A, t, b, e, bounds, psa_bytes)
        assert st.passes == nleaves - 1 and st.suffixes > 0 or nleaves == 1


@pytest.mark.parametrize("mode", ["atomic", "log", "ovf"])
def test_merge_leaves_gap_modes_and_chain_lengths(A, monkeypatch, mode):
    """the three gap-update modes of the batched stream kernel and short / long chains give the same result"""
    monkeypatch.setenv("PSG_GAP_MODE", mode)
    if mode == "ovf":
        monkeypatch.setenv("PSG_GAP_COUNTER_BITS", "8")
    t = make_text("sig4z", 200_003, 9)
    t[50_000:50_400] = 1                                   # a run: gap values beyond 8 bits somewhere
    for L in ("32", "128", "1024"):
        monkeypatch.setenv("PSG_BATCH_CHAIN_LEN", L)
        b, e = 10_000, 190_000
        bounds = np.linspace(b, e, 24).astype(np.int64)
        _leaf_case(A, t, b, e, bounds, 2)


def test_merge_leaves_many_superblocks_and_block_layouts(A, monkeypatch):
    """the level-wide rank structure with several superblocks (PSG_SM_SB_SHIFT) and the interleaved-block fallback layouts"""
    t = make_text("sig12", 150_001, 3)
    b, e = 5_000, 150_001
    bounds = np.linspace(b, e, 18).astype(np.int64)
    monkeypatch.setenv("PSG_SM_SB_SHIFT", "12")
    _leaf_case(A, t, b, e, bounds, 4)
    monkeypatch.delenv("PSG_SM_SB_SHIFT")
    monkeypatch.setenv("PSG_RANK_LAYOUT", "block")
    _leaf_case(A, t, b, e, bounds, 4)
    monkeypatch.setenv("PSG_BLOCK_SB_SHIFT", "14")
    _leaf_case(A, make_text("rand255", 150_001, 3), b, e, bounds, 4)


def test_merge_leaves_through_a_text_window(A):
    """a text that stays in host memory: the device sees the range and a look-ahead behind it; a comparison between leaves
    that would leave the window fails the call (PSG_EWINDOW) instead of reading outside"""
    from psascan_amd._lib import PsgError
    rng = np.random.default_rng(2)
    x = rng.integers(0, 250, 3_000, dtype=np.uint8)
    t = np.concatenate([rng.integers(0, 250, 2_000, dtype=np.uint8), x, rng.integers(0, 250, 1_000, dtype=np.uint8), x, rng.integers(0, 250, 9_000, dtype=np.uint8)])
    n = len(t)
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    b, e = 1_000, 8_500                                     # the second copy of x (6000..9000) runs 500 symbols past the range
    bounds = np.array([b, 3_100, 5_000, 6_700, e], np.int64)
    parts = [orc.partial_sa(t, sa, isa, int(lo), int(hi), want_gt=False)[0].astype(np.uint16) for lo, hi in zip(bounds[:-1], bounds[1:])]
    d_leaf = A.upload(np.concatenate(parts))
    want_psa, want_bwt, want_i0, _ = orc.partial_sa(t, sa, isa, b, e)
    for w_end, ok in ((n, True), (e + 1_000, True), (e + 100, False)):
        win = A.upload(t[b:w_end], pad_to=64)
        sc = A.search_ctx(win, n, n, None, [], window=(b, w_end))
        if ok:
            d_psa, d_bwt, i0, _, _ = A.merge_leaves(sc, b, e - b, bounds, d_leaf, 2)
            assert i0 == want_i0 and np.array_equal(A.download(d_psa, np.uint32, e - b).astype(np.int64), want_psa)
        else:
            with pytest.raises(PsgError) as ei:
                A.merge_leaves(sc, b, e - b, bounds, d_leaf, 2)
            assert ei.value.code == -7
