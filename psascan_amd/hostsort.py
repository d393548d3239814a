"""ctypes wrapper of the host-side half-block sorter (host/libpsascan_host.so: clean-room SA-IS +
gt renaming, the same code construct_sa uses).  north_star keeps the per-half-block suffix sort on
host cores; this makes it usable as the `sorter` of psascan_amd.pipeline.construct_sa5."""
import ctypes as C
import os
import subprocess

import numpy as np

HOST_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "host")
_u8 = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_u32 = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")


def _load():
    so = os.path.join(HOST_DIR, "libpsascan_host.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", HOST_DIR, "libpsascan_host.so"], stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    L.psh_sort_halfblock.argtypes = [_u8, C.c_int64, C.c_int64, C.c_int64, _u8, _u32, _u8, C.POINTER(C.c_int64), _u32]
    return L


class HostSorter:
    device = False

    def __init__(self):
        self.L = _load()

    def __call__(self, text, beg, end, gt_tail):
        m = end - beg
        n = len(text)
        bits = gt_tail.bits(m) if hasattr(gt_tail, "bits") else np.packbits(
            np.array([0] + [gt_tail(v) for v in range(1, m + 1)], np.uint8), bitorder="little")
        bits = np.ascontiguousarray(np.concatenate([bits, np.zeros(8, np.uint8)]))
        psa = np.zeros(m, np.uint32)
        bwt = np.zeros(m, np.uint8)
        gt = np.zeros((m + 31) // 32 + 1, np.uint32)
        i0 = C.c_int64(-1)
        rc = self.L.psh_sort_halfblock(np.ascontiguousarray(text, np.uint8), n, beg, end, bits, psa, bwt, C.byref(i0), gt)
        if rc != 0:
            raise RuntimeError("host sorter failed (byte 255 in the input?)" if rc == -1 else f"host sorter error {rc}")
        return {"psa": psa, "bwt": bwt, "i0": i0.value, "gt_begin": gt.view(np.uint8)}
