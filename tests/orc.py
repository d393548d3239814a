"""ctypes bindings of the parity checkers (oracle/liborc.so and, when built, oracle/_ref).

Test infrastructure only: imported from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the product package.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")

i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def _build():
    so = os.path.join(ORC_DIR, "liborc.so")
    src = os.path.join(ORC_DIR, "psascan_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORC_DIR, "liborc.so"], stdout=subprocess.DEVNULL)
    return so


_lib = C.CDLL(_build())
_lib.orc_suffix_array.argtypes = [u8p, C.c_int64, i64p]
_lib.orc_partial_sa.argtypes = [u8p, C.c_int64, i64p, i64p, C.c_int64, C.c_int64, i64p, u8p, C.POINTER(C.c_int64), C.c_void_p]
_lib.orc_rank_build.argtypes = [u8p, C.c_int64]
_lib.orc_rank_build.restype = C.c_void_p
_lib.orc_rank_free.argtypes = [C.c_void_p]
_lib.orc_rank.argtypes = [C.c_void_p, C.c_int64, C.c_int]
_lib.orc_rank.restype = C.c_int64
_lib.orc_rank_counts.argtypes = [C.c_void_p]
_lib.orc_rank_counts.restype = C.POINTER(C.c_int64)
_lib.orc_stream_pass.argtypes = [C.c_void_p, C.c_int64, C.c_int, u8p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, u64p, C.c_void_p]
_lib.orc_stream_pass.restype = C.c_int64
_lib.orc_gap_to_bitvector.argtypes = [u64p, C.c_int64, u8p]
_lib.orc_gap_to_bitvector.restype = C.c_int64
_lib.orc_merge_bwt.argtypes = [u8p, u8p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, u8p, u8p]
_lib.orc_merge_bwt.restype = C.c_int64
_lib.orc_right_gap.argtypes = [u64p, u8p, C.c_int64, C.c_int64, u64p]
_lib.orc_left_gap.argtypes = [u64p, u8p, C.c_int64, C.c_int64, u64p]
_lib.orc_vbyte_encode.argtypes = [u64p, C.c_int64, u8p]
_lib.orc_vbyte_encode.restype = C.c_int64
_lib.orc_vbyte_decode.argtypes = [u8p, C.c_int64, u64p]
_lib.orc_vbyte_decode.restype = C.c_int64
_lib.orc_merge.argtypes = [C.c_int, i64p, i64p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), u8p]
_lib.orc_psascan.argtypes = [u8p, C.c_int64, C.c_int64, C.c_int64, u8p]
_lib.orc_psascan.restype = C.c_int
_lib.orc_initial_rank.argtypes = [u8p, C.c_int64, C.c_int64, C.c_int64, i64p, C.c_int64, C.c_void_p, C.c_int64]
_lib.orc_initial_rank.restype = C.c_int64


def as_u8(x):
    return np.ascontiguousarray(np.frombuffer(x, dtype=np.uint8) if isinstance(x, (bytes, bytearray)) else x, dtype=np.uint8)


def suffix_array(text):
    text = as_u8(text)
    sa = np.zeros(len(text), np.int64)
    assert _lib.orc_suffix_array(text, len(text), sa) == 0
    return sa


def inverse(sa):
    isa = np.empty_like(sa)
    isa[sa] = np.arange(len(sa), dtype=np.int64)
    return isa


def partial_sa(text, sa, isa, beg, end, want_gt=True):
    """-> psa (rel), bwt, i0, gt_begin bits (u = end - j)."""
    text = as_u8(text)
    m = end - beg
    psa = np.zeros(m, np.int64)
    bwt = np.zeros(m, np.uint8)
    i0 = C.c_int64(0)
    gt = np.zeros(m // 8 + 2, np.uint8)
    _lib.orc_partial_sa(text, len(text), sa, isa, beg, end, psa, bwt, C.byref(i0), gt.ctypes.data if want_gt else None)
    return psa, bwt, i0.value, gt


class Rank:
    def __init__(self, bwt):
        self.bwt = as_u8(bwt)
        self.h = _lib.orc_rank_build(self.bwt, len(self.bwt))

    def rank(self, i, c):
        return _lib.orc_rank(self.h, int(i), int(c))

    def counts(self):
        p = _lib.orc_rank_counts(self.h)
        return np.array([p[c] for c in range(256)], np.int64)

    def __del__(self):
        if getattr(self, "h", None):
            _lib.orc_rank_free(self.h)
            self.h = None


def stream_pass(rank, i0, last, text, tb, te, gt_in, init_rank, gap=None):
    """-> gap (u64[m+1]), gt_out bits (u = te - j), final rank."""
    text = as_u8(text)
    m = len(rank.bwt)
    if gap is None:
        gap = np.zeros(m + 1, np.uint64)
    gt_out = np.zeros((te - tb) // 8 + 2, np.uint8)
    gin = None if gt_in is None else as_u8(gt_in)
    fin = _lib.orc_stream_pass(rank.h, i0, int(last), text, tb, te, None if gin is None else gin.ctypes.data, init_rank,
                               gap, gt_out.ctypes.data)
    return gap, gt_out, fin


def gap_to_bitvector(gap, m):
    gap = np.ascontiguousarray(gap, np.uint64)
    total = m + int(gap.sum())
    bv = np.zeros(total // 8 + 2, np.uint8)
    n = _lib.orc_gap_to_bitvector(gap, m, bv)
    assert n == total
    return bv, total


def merge_bwt(lbwt, rbwt, li0, ri0, left_last, bv):
    lbwt, rbwt = as_u8(lbwt), as_u8(rbwt)
    out = np.zeros(len(lbwt) + len(rbwt), np.uint8)
    bi0 = _lib.orc_merge_bwt(lbwt, rbwt, len(lbwt), len(rbwt), li0, ri0, int(left_last), as_u8(bv), out)
    return out, bi0


def right_gap(block_gap, bv, ml, mr):
    out = np.zeros(mr + 1, np.uint64)
    _lib.orc_right_gap(np.ascontiguousarray(block_gap, np.uint64), as_u8(bv), ml, mr, out)
    return out


def left_gap(block_gap, bv, ml, mr):
    out = np.zeros(ml + 1, np.uint64)
    _lib.orc_left_gap(np.ascontiguousarray(block_gap, np.uint64), as_u8(bv), ml, mr, out)
    return out


def vbyte_encode(vals):
    vals = np.ascontiguousarray(vals, np.uint64)
    out = np.zeros(10 * len(vals) + 1, np.uint8)
    nb = _lib.orc_vbyte_encode(vals, len(vals), out)
    return out[:nb].copy()


def vbyte_decode(buf, cnt):
    out = np.zeros(cnt, np.uint64)
    k = _lib.orc_vbyte_decode(as_u8(buf), len(buf), out)
    assert k == cnt
    return out


def merge(begs, sizes, psas, gaps):
    H = len(begs)
    begs = np.ascontiguousarray(begs, np.int64)
    sizes = np.ascontiguousarray(sizes, np.int64)
    psas = [np.ascontiguousarray(p, np.int64) for p in psas]
    gaps = [None if g is None else np.ascontiguousarray(g, np.uint64) for g in gaps]
    pp = (C.c_void_p * H)(*[p.ctypes.data for p in psas])
    gp = (C.c_void_p * H)(*[None if g is None else g.ctypes.data for g in gaps])
    out = np.zeros(5 * int(sizes.sum()), np.uint8)
    _lib.orc_merge(H, begs, sizes, pp, gp, out)
    return out


def initial_rank(text, bb, be, psa, cmp_end, gt_cmp_end, p):
    """#suffixes of [bb,be) smaller than text[p..): binary search over the partial SA, comparisons that reach cmp_end are
    decided by gt_cmp_end (bit u <-> position n - u, w.r.t. cmp_end); em_compute_initial_ranks.hpp:54-76, 321-363."""
    text = as_u8(text)
    g = None if gt_cmp_end is None else as_u8(gt_cmp_end)
    return _lib.orc_initial_rank(text, len(text), bb, be, np.ascontiguousarray(psa, np.int64), cmp_end,
                                 None if g is None else g.ctypes.data, p)


def psascan(text, max_block_size, ram_use=None):
    text = as_u8(text)
    out = np.zeros(5 * len(text), np.uint8)
    if ram_use is None:
        ram_use = int(max_block_size * 5.2) + 1
    assert _lib.orc_psascan(text, len(text), max_block_size, ram_use, out) == 0
    return out


def sa_to_sa5(sa):
    sa = np.asarray(sa, np.uint64)
    out = np.zeros((len(sa), 5), np.uint8)
    for b in range(5):
        out[:, b] = (sa >> np.uint64(8 * b)) & np.uint64(255)
    return out.reshape(-1)


def sa5_to_sa(buf):
    a = as_u8(buf).reshape(-1, 5).astype(np.uint64)
    v = np.zeros(len(a), np.uint64)
    for b in range(5):
        v |= a[:, b] << np.uint64(8 * b)
    return v.astype(np.int64)


def bits(bv, nbits):
    return np.unpackbits(as_u8(bv), bitorder="little")[:nbits]


def packbits(b):
    return np.packbits(np.asarray(b, np.uint8), bitorder="little")


# --------------------------------------------------------------------------------------
# the real reference (oracle/_ref), present only where it was built from /root/reference
# --------------------------------------------------------------------------------------
def ref_lib():
    so = os.path.join(ORC_DIR, "_ref", "libpsascan_ref.so")
    if not os.path.exists(so):
        return None
    L = C.CDLL(so)
    L.ref_rank.argtypes = [u8p, C.c_long, i64p, u8p, C.c_long, i64p, i64p]
    L.ref_compute_gap.argtypes = [u8p, C.c_long, C.c_long, C.c_int, u8p, C.c_long, C.c_long, C.c_long, C.c_void_p, i64p,
                                  C.c_long, C.c_char_p, u64p, u8p]
    L.ref_gap_to_bitvector.argtypes = [u64p, C.c_long, C.c_char_p, u8p, C.c_long]
    L.ref_merge_bwt.argtypes = [u8p, u8p, C.c_long, C.c_long, C.c_long, C.c_long, C.c_int, u8p, u8p]
    L.ref_merge_bwt.restype = C.c_long
    L.ref_set_uint40.argtypes = [C.c_int]
    for f in (L.ref_right_gap, L.ref_left_gap):
        f.argtypes = [u64p, u8p, C.c_long, C.c_long, C.c_char_p, u64p, u8p, C.POINTER(C.c_long)]
    L.ref_gap_save_vbyte.argtypes = [u64p, C.c_long, C.c_char_p, u8p]
    L.ref_gap_save_vbyte.restype = C.c_long
    L.ref_merge.argtypes = [C.c_int, i64p, i64p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_long, C.c_char_p, u8p]
    if hasattr(L, "ref_initial_ranks"):
        L.ref_initial_ranks.argtypes = [u8p, C.c_long, C.c_long, C.c_long, i32p, u8p, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_long,
                                        C.c_char_p, i64p]
        L.ref_initial_ranks.restype = C.c_long
        L.ref_initial_ranks2.argtypes = [u8p, C.c_long, C.c_long, C.c_long, i32p, C.c_long, C.c_void_p, C.c_long, C.c_char_p, i64p]
        L.ref_initial_ranks2.restype = C.c_long
    return L


def workdir():
    return tempfile.mkdtemp(prefix="psascan_ref_")
