"""Python mirror of the C ABI (include/psascan_amd.h).  Names follow the reference's call sites
(partial_sufsort.hpp:403-542, psascan.hpp:120): rank_build ~ `new rank4n<>`, stream_gap ~
`compute_gap<T>`, gap_to_bitvector ~ `convert_to_bitvector`, merge_bwt, split_gap ~
`compute_right_gap`+`compute_left_gap`, merge_half_blocks ~ `merge<T>`.  Everything runs on the
device; numpy arrays are only the host ends of explicit upload()/download()."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (HbDescC, HbHostDescC, HbSliceDescC, MergeCheckC, MergeStreamStatsC, SINK_FN, SearchCtxC, StreamArgsC, StreamStatsC, check, lib)


class DeviceBuffer:
    """Owning handle of a device allocation (HBM)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().psg_malloc(C.byref(p), self.nbytes))
        self.ptr = p.value

    def free(self):
        if getattr(self, "ptr", None):
            lib().psg_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def at(self, byte_offset):
        return self.ptr + int(byte_offset)


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, DeviceBuffer):
        return x.ptr
    return int(x)


def sync():
    check(lib().psg_sync())


def zeros(nbytes):
    b = DeviceBuffer(nbytes)
    check(lib().psg_memset(b.ptr, 0, b.nbytes))
    return b


def upload(arr, pad_to=4):
    """numpy -> new device buffer (size rounded up to `pad_to` bytes, padding zeroed)."""
    a = np.ascontiguousarray(arr)
    nb = a.nbytes
    cap = (nb + pad_to - 1) // pad_to * pad_to
    b = zeros(max(cap, pad_to))
    if nb:
        check(lib().psg_h2d(b.ptr, a.ctypes.data, nb))
    return b


def download(buf, dtype, count, byte_offset=0):
    out = np.empty(int(count), dtype)
    if out.nbytes:
        check(lib().psg_d2h(out.ctypes.data, _ptr(buf) + byte_offset, out.nbytes))
    return out


class RankStructure:
    """Device-resident rank over a block BWT (reference: rank4n<>, rank.hpp:74-722)."""

    def __init__(self, handle, m):
        self.h = handle
        self.m = m
        cnt = (C.c_int64 * 256)()
        check(lib().psg_rank_counts(self.h, cnt))
        self.counts = np.array(cnt[:], np.int64)

    def device_bytes(self):
        return lib().psg_rank_device_bytes(self.h)

    def query(self, i, c):
        i = np.ascontiguousarray(i, np.int64)
        c = np.ascontiguousarray(c, np.uint8)
        di, dc, do = upload(i), upload(c), DeviceBuffer(max(8, i.nbytes))
        check(lib().psg_rank_query(self.h, di.ptr, dc.ptr, len(i), do.ptr))
        return download(do, np.int64, len(i))

    def free(self):
        if self.h:
            lib().psg_rank_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def rank_build(d_bwt, m, data_bytes_per_block=0):
    h = C.c_void_p()
    check(lib().psg_rank_build(_ptr(d_bwt), m, data_bytes_per_block, C.byref(h)))
    return RankStructure(h.value, m)


class StreamStats:
    def __init__(self, s):
        for k, _ in StreamStatsC._fields_:
            setattr(self, k, getattr(s, k))

    def __repr__(self):
        return "StreamStats(" + ", ".join(f"{k}={getattr(self, k)}" for k, _ in StreamStatsC._fields_) + ")"


PSG_GAP_UNINITIALIZED, PSG_FAIL_IF_UNRESOLVED, PSG_EUNRESOLVED = 1, 2, -6


def search_ctx(d_text, n, cmp_end, d_gt_cmp_end, parts, window=None, window2=None):
    """psg_search_ctx: parts = [(beg, size, d_psa_lo, d_psa_hi or None)] (1 or 2 half-blocks below cmp_end);
    d_gt_cmp_end: bit (n - j) = [text[j..) > text[cmp_end..)].  window = (begin, end): d_text holds text[begin..end)
    only (d_text = pointer to the window's first byte); a comparison that leaves it raises PSG_EWINDOW.
    window2 = (d_text2, begin, end): the searched positions are read from a second window (psg_initial_ranks)."""
    sc = SearchCtxC()
    sc.d_text, sc.n, sc.cmp_end, sc.d_gt_cmp_end, sc.nparts = _ptr(d_text), n, cmp_end, _ptr(d_gt_cmp_end), len(parts)
    if window is not None:
        sc.text_begin, sc.text_end = window
        sc.d_text = _ptr(d_text) - window[0]
    if window2 is not None:
        sc.d_text2, sc.text2_begin, sc.text2_end = _ptr(window2[0]) - window2[1], window2[1], window2[2]
        sc._keep2 = window2[0]
    for k, (beg, size, lo, hi) in enumerate(parts):
        sc.part[k].beg, sc.part[k].size, sc.part[k].d_psa_lo, sc.part[k].d_psa_hi = beg, size, _ptr(lo), _ptr(hi)
    sc._keep = (d_text, d_gt_cmp_end, parts)
    return sc


def initial_ranks(sc, positions):
    """em_compute_initial_ranks: #block suffixes smaller than text[p..n) for every p (string search on the device)."""
    pos = np.ascontiguousarray(positions, np.int64)
    out = np.zeros(len(pos), np.int64)
    check(lib().psg_initial_ranks(C.byref(sc), pos.ctypes.data_as(C.POINTER(C.c_int64)), len(pos), out.ctypes.data_as(C.POINTER(C.c_int64))))
    return out


def stream_gap(rank, block_i0, block_last_symbol, d_tail, tail_len, d_gt_in, rank_at_tail_end, d_gap, d_gt_out,
               max_chains=0, right_context=0, fresh_gap=False, search=None, tail_begin_abs=0, fail_if_unresolved=False, search_all=False):
    """One streaming pass (compute_gap<T>), optionally over a sub-range of the tail with
    `right_context` bytes/bits of valid text/gt to its right.  fresh_gap: d_gap is uninitialised
    (the reference's freshly constructed gap array) -- the pass zero-fills / overwrites it.
    search: psg_search_ctx for the chain starts the warm-up leaves open (tail_begin_abs = position of d_tail[0]).
    Returns (final_rank, StreamStats)."""
    fin = C.c_int64(0)
    st = StreamStatsC()
    if search is not None or fail_if_unresolved:
        a = StreamArgsC(rank.h, block_i0, int(block_last_symbol), _ptr(d_tail), tail_len, right_context, _ptr(d_gt_in), rank_at_tail_end,
                        _ptr(d_gap), _ptr(d_gt_out), max_chains, (1 if fresh_gap else 0) | (2 if fail_if_unresolved else 0) | (4 if search_all else 0),
                        C.pointer(search) if search is not None else None, tail_begin_abs)
        check(lib().psg_stream_gap_args(C.byref(a), C.byref(fin), C.byref(st)))
        return fin.value, StreamStats(st)
    if fresh_gap:
        check(lib().psg_stream_gap_ex(rank.h, block_i0, int(block_last_symbol), _ptr(d_tail), tail_len, right_context,
                                      _ptr(d_gt_in), rank_at_tail_end, _ptr(d_gap), _ptr(d_gt_out), max_chains, 1,
                                      C.byref(fin), C.byref(st)))
    else:
        check(lib().psg_stream_gap_ctx(rank.h, block_i0, int(block_last_symbol), _ptr(d_tail), tail_len, right_context,
                                       _ptr(d_gt_in), rank_at_tail_end, _ptr(d_gap), _ptr(d_gt_out), max_chains,
                                       C.byref(fin), C.byref(st)))
    return fin.value, StreamStats(st)


def gap_words(m):
    """uint32 words of a gap array over m + 1 slots (counters + in-band excess list, psg_gap_words)"""
    return ((m + 1 + 3) & ~3) + 4 + 2 * 65536


def gap_array(m, fill=0):
    """A gap array over m + 1 slots.  fill=0: empty (all zero); fill=None: uninitialised (for PSG_GAP_UNINITIALIZED
    passes); another value: every counter starts at it (tests of accumulating passes)."""
    if fill is None:
        return DeviceBuffer(4 * gap_words(m))
    b = zeros(4 * gap_words(m))
    if fill:
        a = np.full(m + 1, fill, np.uint32)
        check(lib().psg_h2d(b.ptr, a.ctypes.data, a.nbytes))
    return b


def gap_array_from_values(values, bits=32):
    """A gap array holding the given 64-bit values: counters = low `bits` bits, one excess entry per 2^bits above."""
    v = np.ascontiguousarray(values, np.uint64)
    m = len(v) - 1
    b = zeros(4 * gap_words(m))
    cells = (v & np.uint64((1 << bits) - 1)).astype(np.uint32)
    check(lib().psg_h2d(b.ptr, cells.ctypes.data, cells.nbytes))
    carries = (v >> np.uint64(bits)).astype(np.int64)
    ent = np.repeat(np.arange(m + 1, dtype=np.uint64), carries)
    assert len(ent) <= 65536
    hdr = np.array([len(ent), bits, 0, 0], np.uint32)
    hw = ((m + 1 + 3) & ~3)
    check(lib().psg_h2d(b.ptr + 4 * hw, hdr.ctypes.data, 16))
    if len(ent):
        ent = np.ascontiguousarray(ent[np.random.default_rng(1).permutation(len(ent))])     # the list is unordered
        check(lib().psg_h2d(b.ptr + 4 * hw + 16, ent.ctypes.data, ent.nbytes))
    return b


def gap_values(d_gap, m):
    """value(j) = counter[j] + 2^bits * #{excess entries equal to j}, as np.uint64[m + 1]"""
    out = DeviceBuffer(8 * (m + 1))
    check(lib().psg_gap_values(_ptr(d_gap), m, out.ptr))
    return download(out, np.uint64, m + 1)


def gap_to_bitvector(d_gap, m, d_bv, capacity_bits):
    nbits = C.c_int64(0)
    check(lib().psg_gap_to_bitvector(_ptr(d_gap), m, _ptr(d_bv), capacity_bits, C.byref(nbits)))
    return nbits.value


def merge_bwt(d_left_bwt, d_right_bwt, ml, mr, left_i0, right_i0, left_last, d_bv, d_out):
    bi0 = C.c_int64(-1)
    check(lib().psg_merge_bwt(_ptr(d_left_bwt), _ptr(d_right_bwt), ml, mr, left_i0, right_i0, int(left_last), _ptr(d_bv),
                              _ptr(d_out), C.byref(bi0)))
    return bi0.value


def split_gap(d_block_gap, d_bv, ml, mr, tail_len, d_mbv_left, d_mbv_right):
    check(lib().psg_split_gap(_ptr(d_block_gap), _ptr(d_bv), ml, mr, tail_len, _ptr(d_mbv_left), _ptr(d_mbv_right)))


def mbv_to_gap(d_mbv, nbits, size):
    out = DeviceBuffer(8 * (size + 1))
    check(lib().psg_mbv_to_gap(_ptr(d_mbv), nbits, size, out.ptr))
    return out


def vbyte_encode(d_vals, count):
    out = DeviceBuffer(10 * count + 16)
    nb = C.c_int64(0)
    check(lib().psg_vbyte_encode(_ptr(d_vals), count, out.ptr, out.nbytes, C.byref(nb)))
    return out, nb.value


class MergePlan:
    """half_blocks: list of dicts {beg, size, psa_lo, psa_hi (or None), mbv (or None for the last)}."""

    def __init__(self, half_blocks):
        H = len(half_blocks)
        arr = (HbDescC * H)()
        for k, hb in enumerate(half_blocks):
            arr[k].beg = hb["beg"]
            arr[k].size = hb["size"]
            arr[k].d_psa_lo = _ptr(hb["psa_lo"])
            arr[k].d_psa_hi = _ptr(hb.get("psa_hi"))
            arr[k].d_mbv = _ptr(hb.get("mbv"))
        self._keep = half_blocks
        self.n = sum(hb["size"] for hb in half_blocks)
        h = C.c_void_p()
        check(lib().psg_merge_plan_create(arr, H, C.byref(h)))
        self.h = h.value

    def run(self, out_begin, out_count, d_out):
        check(lib().psg_merge_run(self.h, out_begin, out_count, _ptr(d_out)))

    def free(self):
        if getattr(self, "h", None):
            lib().psg_merge_plan_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def merge_run_u32(plan, out_begin, out_count, d_out):
    """merged order as u32 values (plan levels' beg relative to the enclosing range)"""
    check(lib().psg_merge_run_u32(plan.h, out_begin, out_count, _ptr(d_out)))


class BackgroundDownload:
    """psg_d2h_begin: a device buffer drains into a host array on a worker thread with its own stream while the
    library's stream goes on; wait() returns the array.  free_src hands the device buffer to the library."""

    def __init__(self, d_src, dtype, count, free_src=False):
        self.array = np.empty(count, dtype)
        h = C.c_void_p()
        check(lib().psg_d2h_begin(self.array.ctypes.data_as(C.c_void_p), _ptr(d_src), self.array.nbytes, 1 if free_src else 0, C.byref(h)))
        self.h = h.value
        self._keep = None if free_src else d_src
        if free_src and hasattr(d_src, "ptr"):
            d_src.ptr = None                          # no longer ours

    def wait(self):
        if self.h:
            h, self.h = self.h, None
            check(lib().psg_copy_wait(h))
        return self.array


class BackgroundUpload:
    """psg_h2d_begin: a host array goes up into a device buffer on a worker thread with its own stream; wait() before
    the library's stream touches the buffer."""

    def __init__(self, d_dst, array):
        self._keep = (d_dst, np.ascontiguousarray(array))
        h = C.c_void_p()
        check(lib().psg_h2d_begin(_ptr(d_dst), self._keep[1].ctypes.data_as(C.c_void_p), self._keep[1].nbytes, C.byref(h)))
        self.h = h.value

    def wait(self):
        if self.h:
            h, self.h = self.h, None
            check(lib().psg_copy_wait(h))
        return self._keep[0]


def merge_run_planes(plan, out_begin, out_count, d_lo, d_hi):
    """merged order as values of up to 40 bits in two planes (u32 low words, u8 bits 32..39)"""
    check(lib().psg_merge_run_planes(plan.h, out_begin, out_count, _ptr(d_lo), _ptr(d_hi)))


def halfblock_from_psa(sc, beg, size, d_psa, want_gt=True, d_psa_hi=None):
    """-> (d_bwt, i0, d_gt_begin): BWT (dummy 0 at i0), i0 and gt_begin of text[beg..beg+size) from its partial SA
    (d_psa_hi: the plane of bits 32..39, for ranges of 2^32 positions or more)"""
    d_bwt = DeviceBuffer(size + 16)
    d_gt = zeros(4 * ((size + 31) // 32 + 2)) if want_gt else None
    i0 = C.c_int64(-1)
    if d_psa_hi is None:
        check(lib().psg_halfblock_from_psa(C.byref(sc), beg, size, _ptr(d_psa), d_bwt.ptr, C.byref(i0), _ptr(d_gt)))
    else:
        check(lib().psg_halfblock_from_psa40(C.byref(sc), beg, size, _ptr(d_psa), _ptr(d_psa_hi), d_bwt.ptr, C.byref(i0), _ptr(d_gt)))
    return d_bwt, i0.value, d_gt


def merge_leaves(sc, beg, size, leaf_bounds, d_leaf_psa, psa_bytes):
    """psg_merge_leaves: the batched in-memory pSAscan merging (inmem_psascan.hpp:64-304).  leaf_bounds: absolute text
    positions, leaf l = [leaf_bounds[l], leaf_bounds[l+1]); d_leaf_psa: the leaves' partial SAs back to back (positions
    relative to the leaf, 2 or 4 bytes).  -> (d_psa u32 relative to beg, d_bwt, i0, d_gt_begin, stats)"""
    from ._lib import LeafMergeStatsC
    lb = np.ascontiguousarray(leaf_bounds, np.int64)
    d_psa, d_bwt, d_gt = DeviceBuffer(4 * size + 16), DeviceBuffer(size + 16), zeros(4 * ((size + 31) // 32 + 2))
    i0 = C.c_int64(-1)
    st = LeafMergeStatsC()
    check(lib().psg_merge_leaves(C.byref(sc), beg, size, lb.ctypes.data_as(C.POINTER(C.c_int64)), len(lb) - 1, _ptr(d_leaf_psa), psa_bytes,
                                 d_psa.ptr, d_bwt.ptr, C.byref(i0), d_gt.ptr, C.byref(st)))
    return d_psa, d_bwt, i0.value, d_gt, st


def bits_rank1(d_bits, nbits, positions):
    """ones in d_bits[0 .. pos) for every pos (rank1 of ranksel_support.hpp:45-187)"""
    pos = np.ascontiguousarray(positions, np.int64)
    out = np.zeros(len(pos), np.int64)
    check(lib().psg_bits_rank1(_ptr(d_bits), nbits, pos.ctypes.data_as(C.POINTER(C.c_int64)), len(pos), out.ctypes.data_as(C.POINTER(C.c_int64))))
    return out


class SlicedMergePlan(MergePlan):
    """merge plan over slices of the levels (block-per-GPU schedule): levels = dicts with beg, size, nbits, d_mbv
    (device pointer or None), first_word, n_words, ones_before, d_psa (device pointer or None), psa_first, psa_count."""

    def __init__(self, levels):
        H = len(levels)
        arr = (HbSliceDescC * H)()
        for k, lv in enumerate(levels):
            arr[k].beg, arr[k].size, arr[k].nbits = lv["beg"], lv["size"], lv["nbits"]
            arr[k].d_mbv_words, arr[k].first_word, arr[k].n_words, arr[k].ones_before = _ptr(lv["d_mbv"]), lv["first_word"], lv["n_words"], lv["ones_before"]
            arr[k].d_psa_lo, arr[k].d_psa_hi, arr[k].psa_first, arr[k].psa_count = _ptr(lv["d_psa"]), _ptr(lv.get("d_psa_hi")), lv["psa_first"], lv["psa_count"]
        self._keep = levels
        self.n = sum(lv["size"] for lv in levels)
        h = C.c_void_p()
        check(lib().psg_merge_plan_create_sliced(arr, H, C.byref(h)))
        self.h = h.value


def merge_half_blocks(half_blocks, d_out=None):
    """merge<T>: whole suffix array as uint40 LE into a device buffer (returned)."""
    plan = MergePlan(half_blocks)
    if d_out is None:
        d_out = DeviceBuffer(5 * plan.n + 8)
    plan.run(0, plan.n, d_out)
    plan.free()
    return d_out


class PinnedArray:
    """numpy view of page-locked host memory (psg_host_alloc): DMA source / target without a staging copy."""

    def __init__(self, count, dtype):
        self.dtype = np.dtype(dtype)
        self.count = int(count)
        p = C.c_void_p()
        check(lib().psg_host_alloc(C.byref(p), max(16, self.count * self.dtype.itemsize)))
        self.ptr = p.value
        buf = (C.c_char * (self.count * self.dtype.itemsize)).from_address(self.ptr)
        self.array = np.frombuffer(buf, self.dtype, self.count)

    def free(self):
        if getattr(self, "ptr", None):
            self.array = None
            lib().psg_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def mem_stats():
    """(bytes in use, peak bytes in use, bytes reserved from the driver) of the library's device allocator."""
    a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    check(lib().psg_mem_stats(C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def device_memory():
    """(free, total) bytes of the device as the driver sees it"""
    a, b = C.c_int64(0), C.c_int64(0)
    check(lib().psg_device_memory(C.byref(a), C.byref(b)))
    return a.value, b.value


def merge_stream(half_blocks, slice_entries, sink=None, check_text=None, n=0, samples_per_slice=0, seed=1):
    """merge<T> with the partial SAs in host memory (numpy arrays / PinnedArray.array under "psa_lo", optional
    "psa_hi"), merge bitvectors on the device.  sink(bytes_view: np.uint8, first_entry, n_entries) is called in
    output order.  Returns (stats, check) with check = (sum, bad_pairs) or None."""
    H = len(half_blocks)
    arr = (HbHostDescC * H)()
    keep = []
    for k, hb in enumerate(half_blocks):
        arr[k].beg, arr[k].size = hb["beg"], hb["size"]
        arr[k].d_mbv = _ptr(hb.get("mbv"))
        if hb.get("mbv_host") is not None:                  # merge bitvector spilled to host memory (mbv_spill)
            words, samp = hb["mbv_host"]
            keep += [words, samp]
            arr[k].d_mbv = None
            arr[k].h_mbv, arr[k].h_mbv_samp = words.ctypes.data, samp.ctypes.data
        if isinstance(hb["psa_lo"], DeviceBuffer):          # resident in HBM: used where it lies
            arr[k].d_psa_lo = hb["psa_lo"].ptr
            arr[k].d_psa_hi = _ptr(hb.get("psa_hi"))
            continue
        lo = np.ascontiguousarray(hb["psa_lo"], np.uint32) if not isinstance(hb["psa_lo"], np.ndarray) or hb["psa_lo"].dtype != np.uint32 else hb["psa_lo"]
        keep.append(lo)
        arr[k].h_psa_lo = lo.ctypes.data
        hi = hb.get("psa_hi")
        if hi is not None:
            hi = np.ascontiguousarray(hi, np.uint8)
            keep.append(hi)
            arr[k].h_psa_hi = hi.ctypes.data
    err = []

    def _sink(ctx, ptr, first, cnt):
        try:
            view = np.frombuffer((C.c_char * (5 * cnt)).from_address(ptr), np.uint8, 5 * cnt)
            sink(view, first, cnt)
            return 0
        except Exception as e:   # never let an exception cross the C boundary
            err.append(e)
            return 1

    cb = SINK_FN(_sink) if sink is not None else C.cast(None, SINK_FN)
    chk = None
    if check_text is not None:
        chk = MergeCheckC(_ptr(check_text), n, samples_per_slice, seed, 0, 0, 0)
    st = MergeStreamStatsC()
    rc = lib().psg_merge_stream(arr, H, slice_entries, C.byref(chk) if chk is not None else None, cb, None, C.byref(st))
    if err:
        raise err[0]
    check(rc)
    return st, ((chk.sum, chk.bad_pairs) if chk is not None else None)


def mbv_spill(d_mbv, nbits):
    """psg_mbv_spill: a merge bitvector leaves HBM -> (words: np.uint32, rank samples: np.uint64) in host memory"""
    words = np.zeros(lib().psg_mbv_spill_words(nbits), np.uint32)
    samp = np.zeros((nbits + 4095) // 4096 + 1, np.uint64)
    check(lib().psg_mbv_spill(_ptr(d_mbv), nbits, words.ctypes.data, samp.ctypes.data))
    return words, samp


class MbvSpill:
    """psg_mbv_spill_begin / psg_mbv_spill_finish: the rank samples are there at once, the words drain in the background
    on the library's copy worker; the device buffer is handed to the library.  wait() -> (words, samples)."""

    def __init__(self, d_mbv, nbits):
        self.nbits = int(nbits)
        self.words = np.empty(lib().psg_mbv_spill_words(nbits), np.uint32)
        self.samp = np.zeros((nbits + 4095) // 4096 + 1, np.uint64)
        h = C.c_void_p()
        check(lib().psg_mbv_spill_begin(_ptr(d_mbv), self.nbits, self.words.ctypes.data, self.samp.ctypes.data, C.byref(h)))
        self.h = h.value
        if hasattr(d_mbv, "ptr"):
            d_mbv.ptr = None                          # no longer ours

    def wait(self):
        if self.h:
            h, self.h = self.h, None
            check(lib().psg_copy_wait(h))
            check(lib().psg_mbv_spill_finish(self.words.ctypes.data, self.nbits))
        return self.words, self.samp


def bitcopy(d_dst, dst_bit, d_src, src_bit, nbits):
    check(lib().psg_bitcopy(_ptr(d_dst), dst_bit, _ptr(d_src), src_bit, nbits))


def popcount(d_bits, nbits):
    v = C.c_int64(0)
    check(lib().psg_popcount(_ptr(d_bits), nbits, C.byref(v)))
    return v.value


def last_kernel_ms():
    return lib().psg_last_kernel_ms()


# ---- multi-GPU building blocks (include/psascan_amd.h, "multi-GPU building blocks") ----
class BorrowedBuffer(DeviceBuffer):
    """A device buffer allocated inside the library and handed to the caller (released with psg_free)."""

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, int(nbytes)


def stream_gap_log(rank, block_i0, block_last_symbol, d_tail, tail_len, d_gt_in, rank_at_context_end, d_gt_out,
                   max_chains=0, right_context=0):
    """Stream a tail range and return its rank log: (log buffer of nlog u32, nlog, final_rank, stats)."""
    fin, st, p, n = C.c_int64(0), StreamStatsC(), C.c_void_p(), C.c_int64(0)
    check(lib().psg_stream_gap_log(rank.h, block_i0, int(block_last_symbol), _ptr(d_tail), tail_len, right_context, _ptr(d_gt_in),
                                   rank_at_context_end, _ptr(d_gt_out), max_chains, C.byref(fin), C.byref(st), C.byref(p), C.byref(n)))
    return BorrowedBuffer(p.value, 4 * n.value), n.value, fin.value, StreamStats(st)


def log_partition(d_log, nlog, m, nparts, d_out):
    """-> (offsets[nparts+1], value_bounds[nparts+1])"""
    offs = (C.c_int64 * (nparts + 1))()
    vb = (C.c_int64 * (nparts + 1))()
    check(lib().psg_log_partition(_ptr(d_log), nlog, m, nparts, _ptr(d_out), offs, vb))
    return list(offs), list(vb)


def gap_hist(d_log, nlog, value_base, count, d_gap_slice):
    check(lib().psg_gap_hist(_ptr(d_log), nlog, value_base, count, _ptr(d_gap_slice)))


def gap_slice_to_bits(d_gap_slice, j0, count, m, ps_before, d_bits):
    check(lib().psg_gap_slice_to_bits(_ptr(d_gap_slice), j0, count, m, ps_before, _ptr(d_bits)))


def bits_not(d_bits, nbits):
    check(lib().psg_bits_not(_ptr(d_bits), nbits))
