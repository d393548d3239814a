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


def sort_halfblock(d_text, n, beg, end, want_gt=True):
    """-> dict(psa_lo=DeviceBuffer u32, bwt=DeviceBuffer, i0, gt_begin=DeviceBuffer|None, tie_groups)."""
    size = end - beg
    psa = DeviceBuffer(4 * size + 16)
    bwt = DeviceBuffer(size + 16)
    gt = zeros(4 * ((size + 31) // 32 + 1)) if want_gt else None
    i0, ties = C.c_int64(-1), C.c_int64(0)
    check(lib().psgx_sort_halfblock(_ptr(d_text), n, beg, end, psa.ptr, bwt.ptr, C.byref(i0), _ptr(gt), C.byref(ties)))
    return {"beg": beg, "size": size, "psa_lo": psa, "psa_hi": None, "bwt": bwt, "i0": i0.value, "gt_begin": gt,
            "tie_groups": ties.value, "mbv": None}


def check_sa5(d_text, n, d_sa5, count, samples=1 << 20, seed=7):
    """-> (bad_pairs, sum of entries mod 2^64)."""
    bad, s = C.c_int64(0), C.c_uint64(0)
    check(lib().psgx_check_sa5(_ptr(d_text), n, _ptr(d_sa5), count, samples, seed, C.byref(bad), C.byref(s)))
    return bad.value, s.value


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
