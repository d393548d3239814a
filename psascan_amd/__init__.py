"""psascan_amd -- MI355X-native streaming-gap + merge path of pSAscan.

The product is the HIP library `libpsascan_hip.so` (C ABI in include/psascan_amd.h) and the
C++ `construct_sa` host program.  This package is a thin ctypes mirror of the C ABI used by
the tests, the bench and the multi-GPU driver.  There is NO CPU fallback: `lib()` raises if
the library is missing or no HIP device is present.
"""
from ._lib import lib, load_library, PsgError, LIB_PATH  # noqa: F401
from .api import (  # noqa: F401
    DeviceBuffer, RankStructure, StreamStats, rank_build, stream_gap, gap_to_bitvector, merge_bwt, split_gap,
    mbv_to_gap, vbyte_encode, MergePlan, merge_half_blocks, bitcopy, popcount, upload, download, zeros, sync,
)
