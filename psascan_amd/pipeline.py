"""Device-side block schedule of the hot path: the six steps of the reference's process_block
(partial_sufsort.hpp:67-551) and the final merge (psascan.hpp:117-125), with every hot-path
step running on the GPU through the C ABI.  The per-half-block suffix sort stays on the host
(north_star) and is injected as `sorter`.

    sorter(text: np.uint8[n], beg, end, gt_tail) -> dict(psa=int array rel. to beg, bwt=u8 array,
                                                          i0=int, gt_begin=packed bits, u = end - j)
    gt_tail(v) for v >= 1 answers [text[end+v..n) > text[end..n)] (only v <= end-beg is asked).

This module is used by the tests, bench.py and the multi-GPU driver; the C++ `construct_sa`
implements the same schedule natively.
"""
import numpy as np

from . import api


def block_plan(n, max_block_size, ram_use):
    """[(beg, mid, end)] right-to-left, as partial_sufsort.hpp:564-580 + :86-93."""
    n_blocks = (n + max_block_size - 1) // max_block_size
    plan = []
    for bid in range(n_blocks - 1, -1, -1):
        b = max_block_size * bid
        e = min(b + max_block_size, n)
        bs = e - b
        if e == n:
            ls = min(bs, max(1, ram_use // 10))
        else:
            ls = max(1, bs // 2)
        plan.append((b, b + ls, e))
    return plan


def rank_by_search(text_bytes, beg, psa, p):
    """#suffixes of the block (given by its partial SA) smaller than text[p..n)
    (what em_compute_initial_ranks.hpp:222-319 computes for one position)."""
    n = len(text_bytes)
    if p >= n:
        return 0
    lo, hi = 0, len(psa)

    def less(s):            # text[s..) < text[p..) ?
        k = 64
        while True:
            a, b_ = text_bytes[s:s + k], text_bytes[p:p + k]
            if a != b_ or s + k >= n or p + k >= n:
                return text_bytes[s:s + k] < text_bytes[p:p + k] if a != b_ else (n - s) < (n - p)
            k *= 4

    while lo < hi:
        md = (lo + hi) // 2
        if less(beg + int(psa[md])):
            lo = md + 1
        else:
            hi = md
    return lo


class _GtReader:
    """gt_tail(v) view over a downloaded reversed bit slice."""

    def __init__(self, bits_fn):
        self.fn = bits_fn

    def __call__(self, v):
        return self.fn(v)


def construct_sa5(text, max_block_size, ram_use, sorter, max_chains=0, stats=None, d_text=None, return_device=False,
                  n=None, merge="device", slice_entries=64 << 20, sink=None, check_samples=None, timings=None):
    """Whole run on one GPU.  Returns the .sa5 bytes (np.uint8, 5n), or the device buffer holding them.
    A sorter may return device-resident results ({"device": True, "psa_lo", "bwt", "gt_begin": DeviceBuffers,
    "i0", "size"}), e.g. psascan_amd.extras.sort_halfblock for the full-size property tests; with "psa_host" (a
    uint32 numpy array, ideally pinned) instead of "psa_lo" the partial SA stays in host memory.
    text=None: device-only flow (d_text and n given; no host copy of the text is made).
    merge="stream": the final merge streams the partial SAs from host memory (psg_merge_stream) and hands the
    output to `sink` (None = dropped); returns (merge stats, (sum, bad_pairs) or None)."""
    if text is not None:
        text = np.ascontiguousarray(text, np.uint8)
        n = len(text)
    if n == 0:
        return np.zeros(0, np.uint8)
    tb = None
    if d_text is None:
        d_text = api.upload(text, pad_to=16)

    def sym(pos):                      # text[pos]
        return int(text[pos]) if text is not None else int(api.download(d_text, np.uint8, 1, pos)[0])

    host_sorter = not getattr(sorter, "device", False)
    gt_words = (n + 31) // 32 + 2
    gt_cur = api.zeros(4 * gt_words)   # gt w.r.t. current block begin; bit idx = n - j
    gt_new = api.zeros(4 * gt_words)
    half_blocks = []
    keep = []
    import time as _time

    def dev(res, key, pad):
        """device buffer of a sorter result field (uploads host arrays)"""
        if res.get("device"):
            return res[key]
        a = np.asarray(res[key], np.uint8)
        return api.upload(a[: (res["_bits"] + 7) // 8] if key == "gt_begin" else a, pad_to=pad)

    def host_psa(res):
        if res.get("psa_host") is not None:
            return res["psa_host"]
        if res.get("device"):
            return api.download(res["psa_lo"], np.uint32, res["size"])
        return res["psa"]

    def up_hb(beg, res):
        if res.get("psa_host") is not None:
            return {"beg": beg, "size": res["size"], "psa_lo": res["psa_host"], "psa_hi": None, "mbv": None, "dev_psa": None}
        if res.get("device"):        # resident in HBM (the streamed merge uses it where it lies)
            return {"beg": beg, "size": res["size"], "psa_lo": res["psa_lo"], "psa_hi": None, "mbv": None, "dev_psa": res["psa_lo"]}
        psa = np.asarray(res["psa"], np.uint64)
        if merge == "stream":
            hi = (psa >> np.uint64(32)).astype(np.uint8) if len(psa) and int(psa.max()) >> 32 else None
            return {"beg": beg, "size": len(psa), "psa_lo": (psa & np.uint64(0xFFFFFFFF)).astype(np.uint32), "psa_hi": hi, "mbv": None, "dev_psa": None}
        d_lo = api.upload((psa & np.uint64(0xFFFFFFFF)).astype(np.uint32))
        d_hi = api.upload((psa >> np.uint64(32)).astype(np.uint8)) if len(psa) and int(psa.max()) >> 32 else None
        return {"beg": beg, "size": len(psa), "psa_lo": d_lo, "psa_hi": d_hi, "mbv": None, "dev_psa": d_lo if d_hi is None else None}

    def search_for(e, parts):
        """search context of a pass when every part's partial SA is on the device (chain starts on repetitive text)"""
        if any(p["dev_psa"] is None for p in parts):
            return None
        return api.search_ctx(d_text, n, e, gt_cur if e < n else None, [(p["beg"], p["size"], p["dev_psa"], None) for p in parts])

    for (b, mid, e) in block_plan(n, max_block_size, ram_use):
        ls, rs, bs = mid - b, e - mid, e - b
        last_block = e == n
        api.lib().psg_memset(gt_new.ptr, 0, gt_new.nbytes)
        # host copy of the tail gt (w.r.t. e) for the sorter: positions j in (e, e+rs]
        cur_host = api.download(gt_cur, np.uint8, 4 * gt_words) if (not last_block and host_sorter) else None

        def gt_tail_e(v, cur_host=cur_host, e=e):
            j = e + v
            if j >= n:
                return 0
            idx = n - j
            return (int(cur_host[idx >> 3]) >> (idx & 7)) & 1

        def gt_tail_e_bits(count, cur_host=cur_host, e=e):
            """packed bits (LSB-first), bit v = gt_tail_e(v) for v in [0, count]"""
            out = np.zeros(count + 1, np.uint8)
            if cur_host is not None:
                vmax = min(count, n - e - 1)                 # positions e+v < n
                if vmax >= 1:
                    allbits = np.unpackbits(cur_host, bitorder="little")
                    idx = n - (e + np.arange(1, vmax + 1))
                    out[1: vmax + 1] = allbits[idx]
            return np.packbits(out, bitorder="little")
        gt_tail_e.bits = gt_tail_e_bits

        R = None
        if rs > 0:
            R = sorter(text, mid, e, gt_tail_e)
            R["_bits"] = rs
            d_rbwt = dev(R, "bwt", 16)
            d_rgt = dev(R, "gt_begin", 8)
            rgt_host = np.asarray(R["gt_begin"], np.uint8) if host_sorter else None

            def gt_tail_mid(v, rgt_host=rgt_host, e=e, mid=mid):
                j = mid + v            # j in (mid, e]: right half's gt_begin, u = e - j
                u = e - j
                return (int(rgt_host[u >> 3]) >> (u & 7)) & 1

            def gt_tail_mid_bits(count, rgt_host=rgt_host, e=e, mid=mid):
                out = np.zeros(count + 1, np.uint8)
                vmax = min(count, e - mid)
                if rgt_host is not None and vmax >= 1:
                    allbits = np.unpackbits(rgt_host, bitorder="little")
                    out[1: vmax + 1] = allbits[e - (mid + np.arange(1, vmax + 1))]
                return np.packbits(out, bitorder="little")
            gt_tail_mid.bits = gt_tail_mid_bits
        else:
            gt_tail_mid = gt_tail_e
        L = sorter(text, b, mid, gt_tail_mid)
        L["_bits"] = ls
        d_lgt = dev(L, "gt_begin", 8)
        hbL = up_hb(b, L)
        if rs == 0:
            api.bitcopy(gt_new, n - mid, d_lgt, 0, ls)
            half_blocks.append(hbL)
            gt_cur, gt_new = gt_new, gt_cur
            continue
        hbR = up_hb(mid, R)
        d_lbwt = dev(L, "bwt", 16)
        last_left, last_block_sym = sym(mid - 1), sym(e - 1)
        # ---- step 3: pass A, right half streamed through rank(left BWT) (:403-414)
        rankL = api.rank_build(d_lbwt, ls)
        gapA = api.gap_array(ls, fill=None)       # fresh gap array: the pass zero-fills / overwrites it
        gtA = api.zeros(4 * ((rs + 31) // 32 + 1))
        scA = search_for(e, [hbL])
        if L.get("initA") is not None:
            initA = L["initA"]
        elif scA is not None:
            initA = int(api.initial_ranks(scA, [e])[0])
        else:
            if tb is None:
                tb = text.tobytes()
            initA = rank_by_search(tb, b, host_psa(L), e)
        _, stA = api.stream_gap(rankL, L["i0"], last_left, d_text.at(mid), rs, d_rgt, initA, gapA, gtA, max_chains, fresh_gap=True,
                                search=scA, tail_begin_abs=mid)
        rankL.free()
        if stats is not None:
            stats.append(("A", b, e, stA))
        bvA = api.zeros(4 * ((bs + 31) // 32 + 2))
        nb = api.gap_to_bitvector(gapA, ls, bvA, bs)
        assert nb == bs, (nb, bs)
        gapA.free()
        if last_block:                           # :418-429 -- the left gap IS its merge bitvector
            hbL["mbv"] = bvA
            half_blocks += [hbL, hbR]
            api.bitcopy(gt_new, n - e, gtA, 0, rs)
            api.bitcopy(gt_new, n - mid, d_lgt, 0, ls)
            gt_cur, gt_new = gt_new, gt_cur
            continue
        # ---- step 4: BWT merge (:468-471)
        d_bbwt = api.DeviceBuffer(bs + 16)
        block_i0 = api.merge_bwt(d_lbwt, d_rbwt, ls, rs, L["i0"], R["i0"], last_left, bvA, d_bbwt)
        if not L.get("keep_inputs"):
            d_lbwt.free(); d_rbwt.free()
        # ---- step 5: pass B, the tail streamed through rank(block BWT) (:500-514)
        rankB = api.rank_build(d_bbwt, bs)
        d_bbwt.free()
        gapB = api.gap_array(bs, fill=None)
        T = n - e
        _, stB = api.stream_gap(rankB, block_i0, last_block_sym, d_text.at(e), T, gt_cur, 0, gapB, gt_new, max_chains, fresh_gap=True,
                                search=search_for(e, [hbL, hbR]), tail_begin_abs=e)
        if stats is not None:
            stB.rank_bytes = rankB.device_bytes()
            stats.append(("B", b, e, stB))
        rankB.free()
        api.bitcopy(gt_new, n - e, gtA, 0, rs)
        api.bitcopy(gt_new, n - mid, d_lgt, 0, ls)
        # ---- step 6: split into the half-block merge bitvectors (:536-542)
        mbvL = api.DeviceBuffer(4 * ((bs + T + 31) // 32 + 1))
        mbvR = api.DeviceBuffer(4 * ((rs + T + 31) // 32 + 1))
        api.split_gap(gapB, bvA, ls, rs, T, mbvL, mbvR)
        gapB.free(); bvA.free()
        hbL["mbv"], hbR["mbv"] = mbvL, mbvR
        half_blocks += [hbL, hbR]
        gt_cur, gt_new = gt_new, gt_cur
    gt_cur.free(); gt_new.free()
    half_blocks.sort(key=lambda h: h["beg"])       # merge.hpp:59
    half_blocks[-1]["mbv"] = None
    if timings is not None:
        api.sync()
        timings["passes_done"] = _time.perf_counter()
    if merge == "stream":
        res = api.merge_stream(half_blocks, slice_entries, sink, check_text=d_text if check_samples is not None else None, n=n,
                               samples_per_slice=check_samples or 0)
        for h in half_blocks:
            if h.get("mbv") is not None:
                h["mbv"].free()
        return res
    d_out = api.merge_half_blocks(half_blocks)
    if return_device:
        return d_out
    out = api.download(d_out, np.uint8, 5 * n)
    keep.clear()
    return out
