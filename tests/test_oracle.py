"""Pins the CPU oracle (oracle/psascan_oracle.c):
  (1) against brute-force definitions from the full-text ISA,
  (2) against the reference's own .sa5 hashes on seeded inputs (tests/golden),
  (3) stage by stage against the reference's own headers (oracle/_ref), when built here.
CPU only."""
import hashlib
import json
import os
import shutil

import numpy as np
import pytest

import orc
from golden import inputs as gin

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "golden.json")))
REF = orc.ref_lib()
needs_ref = pytest.mark.skipif(REF is None, reason="oracle/_ref not built (reference absent)")


def brute_sa(text):
    b = bytes(text)
    return np.array(sorted(range(len(b)), key=lambda i: b[i:]), np.int64)


def texts():
    rng = np.random.default_rng(99)
    return {
        "rand255": rng.integers(0, 255, 3000, dtype=np.uint8),
        "sig4z": rng.integers(0, 4, 3000, dtype=np.uint8),
        "sig2": rng.integers(97, 99, 2500, dtype=np.uint8),
        "per3": np.frombuffer((b"abc" * 700)[:2000], np.uint8).copy(),
        "alla": np.full(1500, 97, np.uint8),
        "fib": gin.fib()[:2584].copy(),
        "zeros": np.zeros(1200, np.uint8),
    }


@pytest.mark.parametrize("name", list(texts().keys()))
def test_suffix_array_vs_bruteforce(name):
    t = texts()[name]
    assert np.array_equal(orc.suffix_array(t), brute_sa(t))


def test_suffix_array_tiny():
    for t in (b"a", b"ab", b"ba", b"aa", b"\x00", b"\x00\x00\x01"):
        assert np.array_equal(orc.suffix_array(t), brute_sa(t))


def _setup(t, b, mid, e):
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    return sa, isa


def r_B(isa, b, e, p, n):
    """#block suffixes smaller than text[p..n)."""
    rp = isa[p] if p < n else -1
    return int((isa[b:e] < rp).sum())


@pytest.mark.parametrize("name", list(texts().keys()))
def test_stream_pass_vs_definition(name):
    t = texts()[name]
    n = len(t)
    b, e = n // 5, n // 5 + n // 3          # block [b,e), tail [e,n)
    sa, isa = _setup(t, b, 0, e)
    psa, bwt, i0, _ = orc.partial_sa(t, sa, isa, b, e)
    # gt of tail positions w.r.t. block end e: bit u <-> j = n - u
    gt_in = orc.packbits([(isa[n - u] if n - u < n else -1) > isa[e] for u in range(n - e)])
    rk = orc.Rank(bwt)
    gap, gt_out, fin = orc.stream_pass(rk, i0, t[e - 1], t, e, n, gt_in, 0)
    want = np.zeros(e - b + 1, np.uint64)
    for p in range(e, n):
        want[r_B(isa, b, e, p, n)] += 1
    assert np.array_equal(gap, want)
    want_gt = [(isa[n - u] if n - u < n else -1) > isa[b] for u in range(n - e)]
    assert np.array_equal(orc.bits(gt_out, n - e), np.array(want_gt, np.uint8))
    assert fin == r_B(isa, b, e, e, n)
    # rank semantics incl. clamps (rank.hpp:566-568)
    for i in (-3, 0, 1, (e - b) // 2, e - b - 1, e - b, e - b + 5):
        for c in (0, int(t[0]), 200):
            assert rk.rank(i, c) == int((bwt[: max(0, min(i, e - b))] == c).sum())


@pytest.mark.parametrize("name", list(texts().keys()))
def test_block_steps_vs_definition(name):
    """pass A -> bitvector -> merge_bwt -> pass B -> left/right gap, all against A.2."""
    t = texts()[name]
    n = len(t)
    b, e = n // 7, n // 7 + n // 2
    mid = b + (e - b) // 2
    sa, isa = _setup(t, b, mid, e)
    lpsa, lbwt, li0, _ = orc.partial_sa(t, sa, isa, b, mid)
    rpsa, rbwt, ri0, rgt = orc.partial_sa(t, sa, isa, mid, e)
    ml, mr = mid - b, e - mid
    initA = int((isa[b:mid] < isa[e]).sum())
    gapA, gtA, _ = orc.stream_pass(orc.Rank(lbwt), li0, t[mid - 1], t, mid, e, rgt, initA)
    bv, nb = orc.gap_to_bitvector(gapA, ml)
    assert nb == e - b
    order = sorted(range(b, e), key=lambda s: isa[s])
    assert np.array_equal(orc.bits(bv, e - b), np.array([s >= mid for s in order], np.uint8))
    bbwt, bi0 = orc.merge_bwt(lbwt, rbwt, li0, ri0, t[mid - 1], bv)
    _, want_bwt, want_i0, _ = orc.partial_sa(t, sa, isa, b, e, want_gt=False)
    assert np.array_equal(bbwt, want_bwt) and bi0 == want_i0
    gt_in = orc.packbits([(isa[n - u] if n - u < n else -1) > isa[e] for u in range(n - e)])
    gapB, _, _ = orc.stream_pass(orc.Rank(bbwt), bi0, t[e - 1], t, e, n, gt_in, 0)
    rg = orc.right_gap(gapB, bv, ml, mr)
    lg = orc.left_gap(gapB, bv, ml, mr)
    want_r = np.zeros(mr + 1, np.uint64)
    for p in range(e, n):
        want_r[r_B(isa, mid, e, p, n)] += 1
    want_l = np.zeros(ml + 1, np.uint64)
    for p in range(mid, n):
        want_l[r_B(isa, b, mid, p, n)] += 1
    assert np.array_equal(rg, want_r)
    assert np.array_equal(lg, want_l)


def test_vbyte_roundtrip_and_known_bytes():
    vals = np.array([0, 1, 127, 128, 300, 16383, 16384, 2 ** 32, 2 ** 40 - 1, 2 ** 63], np.uint64)
    enc = orc.vbyte_encode(vals)
    assert bytes(enc[:6]) == bytes([0, 1, 127, 0x80, 0x01, 0xAC])      # 128 -> 80 01 ; 300 -> AC 02
    assert np.array_equal(orc.vbyte_decode(enc, len(vals)), vals)


def test_merge_vs_definition():
    rng = np.random.default_rng(5)
    t = rng.integers(0, 3, 1500, dtype=np.uint8)
    n = len(t)
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    cuts = [0, 200, 500, 501, 900, 1499, 1500]
    begs, sizes, psas, gaps = [], [], [], []
    for h in range(len(cuts) - 1):
        b, e = cuts[h], cuts[h + 1]
        psa, _, _, _ = orc.partial_sa(t, sa, isa, b, e, want_gt=False)
        begs.append(b); sizes.append(e - b); psas.append(psa)
        if e < n:
            g = np.zeros(e - b + 1, np.uint64)
            for p in range(e, n):
                g[r_B(isa, b, e, p, n)] += 1
            gaps.append(g)
        else:
            gaps.append(None)
    out = orc.merge(begs, sizes, psas, gaps)
    assert np.array_equal(orc.sa5_to_sa(out), sa)


@pytest.mark.parametrize("name", list(GOLD.keys() - {"_comment"}))
def test_pipeline_vs_reference_hashes(name):
    g = GOLD[name]
    t = gin.GENERATORS[name]()
    assert len(t) == g["n"]
    if "input_sha256_prefix" in g:
        assert hashlib.sha256(bytes(t)).hexdigest().startswith(g["input_sha256_prefix"])
    out = orc.psascan(t, g["max_block_size"], g["ram_use"])
    assert hashlib.sha256(bytes(out)).hexdigest() == g["sa5_sha256"]


@pytest.mark.parametrize("n,mb", [(1, 4), (2, 4), (3, 2), (17, 4), (1000, 64), (1000, 999), (1001, 77)])
def test_pipeline_small_shapes(n, mb):
    rng = np.random.default_rng(n * 31 + mb)
    t = rng.integers(0, 3, n, dtype=np.uint8)
    for ram in (int(mb * 5.2) + 1, 10 * 3, 10 * 1000000):   # last block with / without a right half
        out = orc.psascan(t, mb, ram)
        assert np.array_equal(orc.sa5_to_sa(out), brute_sa(t))


# ---------------------------------------------------------------------------------------
# (3) the reference's own code, stage by stage
# ---------------------------------------------------------------------------------------
@pytest.fixture()
def wd():
    d = orc.workdir()
    yield d.encode()
    shutil.rmtree(d, ignore_errors=True)


def _block_case(name="sig4z", n=6000, seed=1):
    rng = np.random.default_rng(seed)
    t = {"sig4z": lambda: rng.integers(0, 4, n, dtype=np.uint8),
         "rand255": lambda: rng.integers(0, 255, n, dtype=np.uint8),
         "alla": lambda: np.full(n, 97, np.uint8),
         "fib": lambda: gin.fib()[:n].copy()}[name]()
    b, e = n // 7, n // 7 + n // 2
    mid = b + (e - b) // 2
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    return t, n, b, mid, e, sa, isa


@needs_ref
def test_ref_rank():
    rng = np.random.default_rng(2)
    for sigma in (2, 4, 255):
        bwt = rng.integers(0, sigma, 70000, dtype=np.uint8)
        qi = rng.integers(-5, len(bwt) + 5, 4000).astype(np.int64)
        qc = rng.integers(0, 256, 4000).astype(np.uint8)
        out = np.zeros(4000, np.int64)
        cnt = np.zeros(256, np.int64)
        REF.ref_rank(bwt, len(bwt), qi, qc, 4000, out, cnt)
        rk = orc.Rank(bwt)
        assert np.array_equal(cnt, rk.counts())
        assert np.array_equal(out, np.array([rk.rank(i, c) for i, c in zip(qi, qc)], np.int64))


@pytest.fixture(params=["int", "uint40"])
def ref_T(request):
    """The reference instantiates compute_gap<T> / merge<T> with int below 2^31 symbols and uint40 above
    (psascan.hpp:117-125): every BASELINE config from configs[1] on runs the uint40 code.  Pin both."""
    if REF is not None:
        REF.ref_set_uint40(1 if request.param == "uint40" else 0)
    yield request.param
    if REF is not None:
        REF.ref_set_uint40(0)


@needs_ref
@pytest.mark.parametrize("name", ["sig4z", "rand255", "alla", "fib"])
def test_ref_block_stages(name, wd, ref_T):
    t, n, b, mid, e, sa, isa = _block_case(name)
    ml, mr = mid - b, e - mid
    lpsa, lbwt, li0, _ = orc.partial_sa(t, sa, isa, b, mid)
    rpsa, rbwt, ri0, rgt = orc.partial_sa(t, sa, isa, mid, e)
    # pass A through the reference's compute_gap, 3 stream threads
    S = (mr + 2) // 3
    ends = [min(mid + (k + 1) * S, e) for k in range((mr + S - 1) // S)]
    ir = np.array([r_B(isa, b, mid, p, n) for p in ends], np.int64)
    gap_ref = np.zeros(ml + 1, np.uint64)
    gt_ref = np.zeros(mr // 8 + 2, np.uint8)
    REF.ref_compute_gap(lbwt, ml, li0, int(t[mid - 1]), t, n, mid, e, rgt.ctypes.data, ir, len(ir), wd, gap_ref, gt_ref)
    gapA, gtA, _ = orc.stream_pass(orc.Rank(lbwt), li0, t[mid - 1], t, mid, e, rgt, int(ir[-1]))
    assert np.array_equal(gapA, gap_ref)
    assert np.array_equal(orc.bits(gtA, mr), orc.bits(gt_ref, mr))
    # bitvector
    bv, nb = orc.gap_to_bitvector(gapA, ml)
    bv_ref = np.zeros(len(bv), np.uint8)
    REF.ref_gap_to_bitvector(gapA, ml, wd, bv_ref, len(bv_ref))
    assert np.array_equal(orc.bits(bv, nb), orc.bits(bv_ref, nb))
    # merge_bwt
    bbwt, bi0 = orc.merge_bwt(lbwt, rbwt, li0, ri0, t[mid - 1], bv)
    out_ref = np.zeros(ml + mr, np.uint8)
    bi0_ref = REF.ref_merge_bwt(lbwt, rbwt, ml, mr, li0, ri0, int(t[mid - 1]), bv, out_ref)
    assert bi0 == bi0_ref and np.array_equal(bbwt, out_ref)
    # pass B (single thread) + split
    gt_in = orc.packbits([(isa[n - u] if n - u < n else -1) > isa[e] for u in range(n - e)])
    gapB, gtB, _ = orc.stream_pass(orc.Rank(bbwt), bi0, t[e - 1], t, e, n, gt_in, 0)
    gapB_ref = np.zeros(ml + mr + 1, np.uint64)
    gtB_ref = np.zeros((n - e) // 8 + 2, np.uint8)
    REF.ref_compute_gap(bbwt, ml + mr, bi0, int(t[e - 1]), t, n, e, n, gt_in.ctypes.data, np.zeros(1, np.int64), 1, wd,
                        gapB_ref, gtB_ref)
    assert np.array_equal(gapB, gapB_ref)
    assert np.array_equal(orc.bits(gtB, n - e), orc.bits(gtB_ref, n - e))
    import ctypes as C
    for fn_ref, fn_orc, cnt in ((REF.ref_right_gap, orc.right_gap, mr + 1), (REF.ref_left_gap, orc.left_gap, ml + 1)):
        vals = np.zeros(cnt, np.uint64)
        vb = np.zeros(10 * cnt, np.uint8)
        nbytes = C.c_long(0)
        fn_ref(gapB, bv, ml, mr, wd, vals, vb, C.byref(nbytes))
        mine = fn_orc(gapB, bv, ml, mr)
        assert np.array_equal(mine, vals)
        assert np.array_equal(orc.vbyte_encode(mine), vb[: nbytes.value])


@needs_ref
def test_ref_big_gap_values_and_vbyte(wd):
    """gap values beyond u8 / u16 (excess paths of gap_array.hpp:79-88,405-434)."""
    ml, mr = 40, 24
    rng = np.random.default_rng(8)
    bvb = rng.permutation(np.array([0] * ml + [1] * mr, np.uint8))
    bv = np.concatenate([orc.packbits(bvb), np.zeros(2, np.uint8)])
    gap = rng.integers(0, 3, ml + mr + 1).astype(np.uint64)
    gap[3] = 255; gap[4] = 256; gap[10] = 70000; gap[11] = 65535; gap[64] = 131072 + 7
    import ctypes as C
    for fn_ref, fn_orc, cnt in ((REF.ref_right_gap, orc.right_gap, mr + 1), (REF.ref_left_gap, orc.left_gap, ml + 1)):
        vals = np.zeros(cnt, np.uint64); vb = np.zeros(10 * cnt, np.uint8); nbytes = C.c_long(0)
        fn_ref(gap, bv, ml, mr, wd, vals, vb, C.byref(nbytes))
        assert np.array_equal(fn_orc(gap, bv, ml, mr), vals)
    out = np.zeros(10 * len(gap), np.uint8)
    nb = REF.ref_gap_save_vbyte(gap, len(gap), wd, out)
    assert np.array_equal(orc.vbyte_encode(gap), out[:nb])


@needs_ref
def test_ref_merge(wd, ref_T):
    rng = np.random.default_rng(11)
    t = rng.integers(0, 3, 5000, dtype=np.uint8)
    n = len(t)
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    cuts = [0, 700, 1500, 1501, 2600, 3333, 4100, 5000]
    begs, sizes, psas, gaps = [], [], [], []
    for h in range(len(cuts) - 1):
        b, e = cuts[h], cuts[h + 1]
        psa, _, _, _ = orc.partial_sa(t, sa, isa, b, e, want_gt=False)
        begs.append(b); sizes.append(e - b); psas.append(psa)
        g = np.zeros(e - b + 1, np.uint64)
        for p in range(e, n):
            g[r_B(isa, b, e, p, n)] += 1
        gaps.append(g if e < n else None)
    mine = orc.merge(begs, sizes, psas, gaps)
    import ctypes as C
    H = len(begs)
    p32 = [np.ascontiguousarray(p, np.int32) for p in psas]
    pp = (C.c_void_p * H)(*[p.ctypes.data for p in p32])
    gp = (C.c_void_p * H)(*[None if g is None else g.ctypes.data for g in gaps])
    out = np.zeros(5 * n, np.uint8)
    rc = REF.ref_merge(H, np.array(begs, np.int64), np.array(sizes, np.int64), pp, gp, 1 << 20, wd, out)
    assert rc == 0
    assert np.array_equal(mine, out)
    assert np.array_equal(orc.sa5_to_sa(out), sa)


def _gt_wrt(isa, n, ref_pos):
    """bit u <-> position n - u: [text[n-u..) > text[ref_pos..)] (the empty suffix is the smallest)."""
    return orc.packbits([(isa[n - u] if u > 0 else -1) > (isa[ref_pos] if ref_pos < n else -1) for u in range(n + 1)])


def _start_rank_texts():
    rng = np.random.default_rng(5)
    return {"rand255": rng.integers(0, 255, 6000, dtype=np.uint8), "sig4z": rng.integers(0, 4, 6000, dtype=np.uint8),
            "per3": np.frombuffer((b"abc" * 2000)[:6000], np.uint8).copy(), "alla": np.full(4000, 97, np.uint8), "fib": gin.fib()[:6765].copy()}


@pytest.mark.parametrize("name", list(_start_rank_texts().keys()))
def test_initial_rank_restatement_vs_definition(name):
    """The oracle's start-rank search (K8's checker) against the A.2 definition, with the comparison boundary at the block
    end (gt w.r.t. the block end: the first overload's rule), behind a mid block (second overload) and at the text end."""
    t = _start_rank_texts()[name]
    n = len(t)
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    b, e = n // 7, n // 7 + n // 3
    psa, _, _, _ = orc.partial_sa(t, sa, isa, b, e, want_gt=False)
    tb = e + n // 5
    pos = sorted(set([e, e + 1, tb, tb + 1, n - 1, n] + list(np.random.default_rng(1).integers(e, n, 40))))
    for cmp_end in (e, tb, n):
        gt = None if cmp_end == n else _gt_wrt(isa, n, cmp_end)
        for p in pos:
            if p < cmp_end:
                continue
            assert orc.initial_rank(t, b, e, psa, cmp_end, gt, int(p)) == r_B(isa, b, e, int(p), n), (name, cmp_end, p)


@needs_ref
@pytest.mark.parametrize("name", list(_start_rank_texts().keys()))
def test_ref_initial_ranks(name, wd, ref_T):
    """K8's pin to the reference's own code: em_compute_initial_ranks (both overloads, em_compute_initial_ranks.hpp:222-319 and
    :513-561, with approx_rank.hpp / sparse_isa.hpp behind the first) == the definition == the oracle's search."""
    if not hasattr(REF, "ref_initial_ranks"):
        pytest.skip("oracle/_ref was built before the start-rank wrappers existed")
    t = _start_rank_texts()[name]
    n = len(t)
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    b, e = n // 7, n // 7 + n // 3
    psa, bwt, i0, _ = orc.partial_sa(t, sa, isa, b, e, want_gt=False)
    psa32 = np.ascontiguousarray(psa, np.int32)
    for threads, tail_end in ((1, n), (3, n), (7, n), (4, e + (n - e) // 2)):
        # first overload: chunk starts e + k * ceil(tail / threads); gt w.r.t. e for positions (e, tail_end], bit u <-> tail_end - u
        gt_full = orc.bits(_gt_wrt(isa, n, e), n + 1)
        gt = orc.packbits([gt_full[n - (tail_end - u)] for u in range(tail_end - e)] + [0] * 8)
        out = np.zeros(64, np.int64)
        after = r_B(isa, b, e, tail_end, n)
        cnt = REF.ref_initial_ranks(t, n, b, e, psa32, bwt, i0, tail_end, gt.ctypes.data, after, threads, wd, out)
        S = (tail_end - e + threads - 1) // threads
        starts = [e + k * S for k in range(cnt)]
        assert starts[-1] < tail_end
        want = [r_B(isa, b, e, p, n) for p in starts]
        assert list(out[:cnt]) == want, (name, threads, tail_end)
        assert [orc.initial_rank(t, b, e, psa, e, _gt_wrt(isa, n, e), p) for p in starts] == want
    for threads, tb in ((1, e), (3, e + n // 9), (5, e + n // 4)):
        # second overload: the text between e and tail_begin is read, gt w.r.t. tail_begin (bit u <-> n - u)
        gt = _gt_wrt(isa, n, tb)
        out = np.zeros(64, np.int64)
        cnt = REF.ref_initial_ranks2(t, n, b, e, psa32, tb, gt.ctypes.data, threads, wd, out)
        S = (n - tb + threads - 1) // threads
        starts = [tb + k * S for k in range(cnt)]
        want = [r_B(isa, b, e, p, n) for p in starts]
        assert list(out[:cnt]) == want, (name, threads, tb)
        assert [orc.initial_rank(t, b, e, psa, tb, gt, p) for p in starts] == want
