"""Synthetic-input preparation on the device (include/psascan_amd_extras.h): seeded text,
prefix-key half-block sorter, .sa5 property check.  Used by bench.py and the full-size tests;
not part of the drop-in boundary."""
import ctypes as C

from ._lib import check, lib
from .api import DeviceBuffer, _ptr, zeros

MODE_BYTES255, MODE_DNA, MODE_LETTERS, MODE_ENGLISH = 0, 1, 2, 3


def gen_text(n, mode=MODE_BYTES255, sigma=0, seed=1, d_text=None):
    if d_text is None:
        d_text = zeros((n + 15) // 16 * 16 + 16)
    check(lib().psgx_gen_text(_ptr(d_text), n, mode, sigma, seed))
    return d_text


def sort_halfblock(d_text, n, beg, end, want_gt=True, text_begin=0):
    """-> dict(psa_lo=DeviceBuffer u32, bwt=DeviceBuffer, i0, gt_begin=DeviceBuffer|None, tie_groups).
    text_begin > 0: only text[text_begin .. n) is on the device (d_text = pointer of position 0 all the same)."""
    size = end - beg
    psa = DeviceBuffer(4 * size + 16)
    bwt = DeviceBuffer(size + 16)
    gt = zeros(4 * ((size + 31) // 32 + 1)) if want_gt else None
    i0, ties = C.c_int64(-1), C.c_int64(0)
    if text_begin:
        check(lib().psgx_sort_halfblock_window(_ptr(d_text), text_begin, n, beg, end, psa.ptr, bwt.ptr, C.byref(i0), _ptr(gt), C.byref(ties)))
    else:
        check(lib().psgx_sort_halfblock(_ptr(d_text), n, beg, end, psa.ptr, bwt.ptr, C.byref(i0), _ptr(gt), C.byref(ties)))
    return {"beg": beg, "size": size, "psa_lo": psa, "psa_hi": None, "bwt": bwt, "i0": i0.value, "gt_begin": gt,
            "tie_groups": ties.value, "mbv": None}


def sort_halfblock_pieces(text_ptr, n_eff, window, beg, end, piece_max=1 << 31):
    """A half-block of any size on the device: sorted in pieces of at most piece_max symbols by the prefix-key sorter, the
    pieces merged with the hot path itself (piece i streams the pieces to its right through its rank structure, the gap
    arrays become merge bitvectors, one merge gives the 40-bit partial SA in two planes) -- what construct_sa
    --device-sort does for the 8 GiB half-blocks of BASELINE configs[3] (host/construct_sa.cpp: merge_nodes).
    text_ptr: device pointer of text position 0 (a window pointer minus its begin), window = (lo, hi) readable range or
    None, n_eff: end of the text as the comparisons see it.  -> dict like sort_halfblock (+ psa_hi when >= 2^32)."""
    from . import api
    import numpy as np
    size = end - beg
    np_ = max(1, -(-size // piece_max))
    tb = window[0] if window else 0
    if np_ == 1 and size < (1 << 32):
        return sort_halfblock(text_ptr, n_eff, beg, end, text_begin=tb)
    cuts = [beg + size * k // np_ for k in range(np_ + 1)]
    pieces = [sort_halfblock(text_ptr, n_eff, cuts[k], cuts[k + 1], text_begin=tb) for k in range(np_)]
    sym = lambda pos: int(api.download(text_ptr + pos, np.uint8, 1)[0])
    gtw = 4 * ((size + 31) // 32 + 4)
    gt_c, mbvs = api.zeros(gtw), [None] * np_
    api.bitcopy(gt_c, 0, pieces[-1]["gt_begin"], 0, pieces[-1]["size"])
    for i in range(np_ - 2, -1, -1):
        c = pieces[i]
        x1 = c["beg"] + c["size"]
        T = end - x1
        sc = api.search_ctx(text_ptr + (window[0] if window else 0), n_eff, n_eff, None, [(c["beg"], c["size"], c["psa_lo"], None)], window=window)
        r_end = int(api.initial_ranks(sc, [end])[0]) if end < n_eff else 0
        rk = api.rank_build(c["bwt"], c["size"])
        gap = api.gap_array(c["size"], fill=None)
        gt_n = api.zeros(gtw)
        api.stream_gap(rk, c["i0"], sym(x1 - 1), text_ptr + x1, T, gt_c, r_end, gap, gt_n, 0, fresh_gap=True, search=sc, tail_begin_abs=x1)
        rk.free()
        mbvs[i] = api.zeros(4 * ((c["size"] + T + 31) // 32 + 2))
        assert api.gap_to_bitvector(gap, c["size"], mbvs[i], c["size"] + T) == c["size"] + T
        gap.free()
        api.bitcopy(gt_n, T, c["gt_begin"], 0, c["size"])
        gt_c.free()
        gt_c = gt_n
    plan = api.MergePlan([{"beg": p["beg"] - beg, "size": p["size"], "psa_lo": p["psa_lo"], "psa_hi": None, "mbv": mbvs[k]} for k, p in enumerate(pieces)])
    lo, hi = DeviceBuffer(4 * size + 16), DeviceBuffer(size + 16)
    api.merge_run_planes(plan, 0, size, lo, hi)
    plan.free()
    for p in pieces:
        for key in ("psa_lo", "bwt", "gt_begin"):
            p[key].free()
    for m in mbvs:
        if m is not None:
            m.free()
    gt_c.free()
    sc = api.search_ctx(text_ptr + (window[0] if window else 0), n_eff, n_eff, None, [], window=window)
    bwt, i0, gt = api.halfblock_from_psa(sc, beg, size, lo, d_psa_hi=hi)
    return {"beg": beg, "size": size, "psa_lo": lo, "psa_hi": hi, "bwt": bwt, "i0": i0, "gt_begin": gt, "tie_groups": sum(p["tie_groups"] for p in pieces), "mbv": None}


def check_sa5(d_text, n, d_sa5, count, samples=1 << 20, seed=7):
    """-> (bad_pairs, sum of entries mod 2^64)."""
    bad, s = C.c_int64(0), C.c_uint64(0)
    check(lib().psgx_check_sa5(_ptr(d_text), n, _ptr(d_sa5), count, samples, seed, C.byref(bad), C.byref(s)))
    return bad.value, s.value


def check_sa5_ex(d_text, n, d_sa5, count, samples=1 << 20, seed=7):
    """-> (bad_pairs, sum of entries mod 2^64, pairs left undecided after 2^24 equal symbols)."""
    bad, s, und = C.c_int64(0), C.c_uint64(0), C.c_int64(0)
    check(lib().psgx_check_sa5_ex(_ptr(d_text), n, _ptr(d_sa5), count, samples, seed, C.byref(bad), C.byref(s), C.byref(und)))
    return bad.value, s.value, und.value


class DeviceSorter:
    """Sorter for psascan_amd.pipeline.construct_sa5 that keeps everything in HBM (prefix-key radix
    sort: texts with short repeats only).  Stands where the host sorter stands; used by the
    full-size property tests and the bench, never by construct_sa."""
    device = True

    def __init__(self, d_text, n):
        self.d_text, self.n = d_text, n
        self.tie_groups = 0

    def __call__(self, text, beg, end, gt_tail):
        r = sort_halfblock(self.d_text, self.n, beg, end)
        self.tie_groups += r["tie_groups"]
        return {"device": True, "psa_lo": r["psa_lo"], "bwt": r["bwt"], "gt_begin": r["gt_begin"], "i0": r["i0"], "size": end - beg}
