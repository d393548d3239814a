"""ctypes loader of libpsascan_hip.so (C ABI: include/psascan_amd.h)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PSASCAN_AMD_LIB") or os.path.join(HERE, "libpsascan_hip.so")


class PsgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"psascan_amd error {code}: {msg}")
        self.code = code


class StreamStatsC(C.Structure):
    _fields_ = [("n_chains", C.c_int64), ("chain_len", C.c_int64), ("warmup_steps", C.c_int64), ("unresolved", C.c_int64),
                ("rounds", C.c_int64), ("kernel_ms", C.c_double), ("total_ms", C.c_double), ("hist_ms", C.c_double)]


class HbDescC(C.Structure):
    _fields_ = [("beg", C.c_int64), ("size", C.c_int64), ("d_psa_lo", C.c_void_p), ("d_psa_hi", C.c_void_p), ("d_mbv", C.c_void_p)]


class HbHostDescC(C.Structure):
    _fields_ = [("beg", C.c_int64), ("size", C.c_int64), ("h_psa_lo", C.c_void_p), ("h_psa_hi", C.c_void_p), ("d_mbv", C.c_void_p),
                ("d_psa_lo", C.c_void_p), ("d_psa_hi", C.c_void_p), ("h_mbv", C.c_void_p), ("h_mbv_samp", C.c_void_p)]


class MergeCheckC(C.Structure):
    _fields_ = [("d_text", C.c_void_p), ("n", C.c_int64), ("samples_per_slice", C.c_int64), ("seed", C.c_uint64),
                ("sum", C.c_uint64), ("bad_pairs", C.c_int64), ("undecided_pairs", C.c_int64)]


class MergeStreamStatsC(C.Structure):
    _fields_ = [("slices", C.c_int64), ("total_ms", C.c_double), ("kernel_ms", C.c_double), ("stage_ms", C.c_double),
                ("sink_ms", C.c_double), ("h2d_bytes", C.c_int64), ("d2h_bytes", C.c_int64)]


class SearchPartC(C.Structure):
    _fields_ = [("beg", C.c_int64), ("size", C.c_int64), ("d_psa_lo", C.c_void_p), ("d_psa_hi", C.c_void_p)]


class SearchCtxC(C.Structure):
    _fields_ = [("d_text", C.c_void_p), ("n", C.c_int64), ("cmp_end", C.c_int64), ("d_gt_cmp_end", C.c_void_p), ("nparts", C.c_int),
                ("part", SearchPartC * 2), ("text_begin", C.c_int64), ("text_end", C.c_int64),
                ("d_text2", C.c_void_p), ("text2_begin", C.c_int64), ("text2_end", C.c_int64)]


class StreamArgsC(C.Structure):
    _fields_ = [("rank", C.c_void_p), ("block_i0", C.c_int64), ("block_last_symbol", C.c_int), ("d_tail", C.c_void_p),
                ("tail_len", C.c_int64), ("right_context", C.c_int64), ("d_gt_in", C.c_void_p), ("rank_at_context_end", C.c_int64),
                ("d_gap", C.c_void_p), ("d_gt_out", C.c_void_p), ("max_chains", C.c_int64), ("flags", C.c_int),
                ("search", C.POINTER(SearchCtxC)), ("tail_begin_abs", C.c_int64)]


class HbSliceDescC(C.Structure):
    _fields_ = [("beg", C.c_int64), ("size", C.c_int64), ("nbits", C.c_int64), ("d_mbv_words", C.c_void_p), ("first_word", C.c_int64),
                ("n_words", C.c_int64), ("ones_before", C.c_int64), ("d_psa_lo", C.c_void_p), ("d_psa_hi", C.c_void_p),
                ("psa_first", C.c_int64), ("psa_count", C.c_int64)]


class LeafMergeStatsC(C.Structure):
    _fields_ = [("levels", C.c_int64), ("passes", C.c_int64), ("suffixes", C.c_int64), ("prepare_ms", C.c_double), ("rank_ms", C.c_double),
                ("search_ms", C.c_double), ("stream_ms", C.c_double), ("hist_ms", C.c_double), ("bitvector_ms", C.c_double), ("merge_ms", C.c_double),
                ("total_ms", C.c_double)]


SINK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64)

# every symbol include/psascan_amd.h declares: name -> (restype, argtypes)
_vp, _i64, _int = C.c_void_p, C.c_int64, C.c_int
SIGNATURES = {
    "psg_init": (_int, [_int]),
    "psg_last_error": (C.c_char_p, []),
    "psg_device_name": (_int, [C.c_char_p, _int]),
    "psg_malloc": (_int, [C.POINTER(_vp), _i64]),
    "psg_free": (_int, [_vp]),
    "psg_trim": (_int, []),
    "psg_memset": (_int, [_vp, _int, _i64]),
    "psg_h2d": (_int, [_vp, _vp, _i64]),
    "psg_d2h": (_int, [_vp, _vp, _i64]),
    "psg_d2d": (_int, [_vp, _vp, _i64]),
    "psg_sync": (_int, []),
    "psg_host_alloc": (_int, [C.POINTER(_vp), _i64]),
    "psg_host_free": (_int, [_vp]),
    "psg_mem_stats": (_int, [C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "psg_device_memory": (_int, [C.POINTER(_i64), C.POINTER(_i64)]),
    "psg_set_stream": (_int, [_vp]),
    "psg_set_memory_limit": (_int, [_i64]),
    "psg_gap_words": (_i64, [_i64]),
    "psg_gap_values": (_int, [_vp, _i64, _vp]),
    "psg_rank_build": (_int, [_vp, _i64, _int, C.POINTER(_vp)]),
    "psg_rank_counts": (_int, [_vp, C.POINTER(_i64)]),
    "psg_rank_device_bytes": (_i64, [_vp]),
    "psg_rank_query": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "psg_rank_free": (None, [_vp]),
    "psg_stream_gap": (_int, [_vp, _i64, _int, _vp, _i64, _vp, _i64, _vp, _vp, _i64, C.POINTER(_i64), C.POINTER(StreamStatsC)]),
    "psg_stream_gap_ctx": (_int, [_vp, _i64, _int, _vp, _i64, _i64, _vp, _i64, _vp, _vp, _i64, C.POINTER(_i64), C.POINTER(StreamStatsC)]),
    "psg_stream_gap_ex": (_int, [_vp, _i64, _int, _vp, _i64, _i64, _vp, _i64, _vp, _vp, _i64, _int, C.POINTER(_i64), C.POINTER(StreamStatsC)]),
    "psg_stream_gap_args": (_int, [C.POINTER(StreamArgsC), C.POINTER(_i64), C.POINTER(StreamStatsC)]),
    "psg_initial_ranks": (_int, [C.POINTER(SearchCtxC), C.POINTER(_i64), _i64, C.POINTER(_i64)]),
    "psg_stream_gap_log": (_int, [_vp, _i64, _int, _vp, _i64, _i64, _vp, _i64, _vp, _i64, C.POINTER(_i64), C.POINTER(StreamStatsC),
                                  C.POINTER(_vp), C.POINTER(_i64)]),
    "psg_log_partition": (_int, [_vp, _i64, _i64, _int, _vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "psg_gap_hist": (_int, [_vp, _i64, _i64, _i64, _vp]),
    "psg_gap_slice_to_bits": (_int, [_vp, _i64, _i64, _i64, C.c_uint64, _vp]),
    "psg_bits_not": (_int, [_vp, _i64]),
    "psg_gap_to_bitvector": (_int, [_vp, _i64, _vp, _i64, C.POINTER(_i64)]),
    "psg_merge_bwt": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, C.POINTER(_i64)]),
    "psg_split_gap": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "psg_mbv_to_gap": (_int, [_vp, _i64, _i64, _vp]),
    "psg_vbyte_encode": (_int, [_vp, _i64, _vp, _i64, C.POINTER(_i64)]),
    "psg_merge_plan_create": (_int, [C.POINTER(HbDescC), _int, C.POINTER(_vp)]),
    "psg_merge_run": (_int, [_vp, _i64, _i64, _vp]),
    "psg_merge_plan_free": (None, [_vp]),
    "psg_merge_run_u32": (_int, [_vp, _i64, _i64, _vp]),
    "psg_halfblock_from_psa": (_int, [C.POINTER(SearchCtxC), _i64, _i64, _vp, _vp, C.POINTER(_i64), _vp]),
    "psg_merge_run_planes": (_int, [_vp, _i64, _i64, _vp, _vp]),
    "psg_d2h_begin": (_int, [_vp, _vp, _i64, _int, C.POINTER(_vp)]),
    "psg_h2d_begin": (_int, [_vp, _vp, _i64, C.POINTER(_vp)]),
    "psg_copy_wait": (_int, [_vp]),
    "psg_halfblock_from_psa40": (_int, [C.POINTER(SearchCtxC), _i64, _i64, _vp, _vp, _vp, C.POINTER(_i64), _vp]),
    "psg_bits_rank1": (_int, [_vp, _i64, C.POINTER(_i64), _i64, C.POINTER(_i64)]),
    "psg_merge_plan_create_sliced": (_int, [C.POINTER(HbSliceDescC), _int, C.POINTER(_vp)]),
    "psg_merge_stream": (_int, [C.POINTER(HbHostDescC), _int, _i64, C.POINTER(MergeCheckC), SINK_FN, _vp, C.POINTER(MergeStreamStatsC)]),
    "psg_merge_leaves": (_int, [C.POINTER(SearchCtxC), _i64, _i64, C.POINTER(_i64), _i64, _vp, _int, _vp, _vp, C.POINTER(_i64), _vp, C.POINTER(LeafMergeStatsC)]),
    "psg_mbv_spill_words": (_i64, [_i64]),
    "psg_mbv_spill": (_int, [_vp, _i64, _vp, _vp]),
    "psg_mbv_spill_begin": (_int, [_vp, _i64, _vp, _vp, C.POINTER(_vp)]),
    "psg_mbv_spill_finish": (_int, [_vp, _i64]),
    "psg_bitcopy": (_int, [_vp, _i64, _vp, _i64, _i64]),
    "psg_popcount": (_int, [_vp, _i64, C.POINTER(_i64)]),
    "psg_last_kernel_ms": (C.c_double, []),
}


# include/psascan_amd_extras.h (bench / property-test input preparation, not the boundary)
EXTRA_SIGNATURES = {
    "psgx_gen_text": (_int, [_vp, _i64, _int, _int, C.c_uint64]),
    "psgx_sort_halfblock": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, C.POINTER(_i64), _vp, C.POINTER(_i64)]),
    "psgx_bind_threads_near_device": (_int, [C.POINTER(_int)]),
    "psgx_arena_stats": (_int, [C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(_i64)]),
    "psgx_sort_halfblock_window": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, C.POINTER(_i64), _vp, C.POINTER(_i64)]),
    "psgx_gap_hist": (_int, [_vp, _i64, _i64, _vp]),
    "psgx_check_sa5": (_int, [_vp, _i64, _vp, _i64, _i64, C.c_uint64, C.POINTER(_i64), C.POINTER(C.c_uint64)]),
    "psgx_check_sa5_ex": (_int, [_vp, _i64, _vp, _i64, _i64, C.c_uint64, C.POINTER(_i64), C.POINTER(C.c_uint64), C.POINTER(_i64)]),
}


def load_library(path=LIB_PATH):
    """dlopen the HIP library and bind every declared symbol (no device needed)."""
    if not os.path.exists(path):
        raise PsgError(-2, f"{path} not found -- run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950); "
                           "psascan_amd has no CPU fallback")
    L = C.CDLL(path)
    for name, (res, args) in list(SIGNATURES.items()) + list(EXTRA_SIGNATURES.items()):
        f = getattr(L, name)   # AttributeError if the symbol is missing
        f.restype = res
        f.argtypes = args
    return L


_LIB = None


def lib(device=None):
    """The initialised library; raises PsgError when no HIP device is available."""
    global _LIB
    if _LIB is None:
        L = load_library()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        rc = L.psg_init(device)
        if rc != 0:
            raise PsgError(rc, L.psg_last_error().decode())
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        raise PsgError(rc, _LIB.psg_last_error().decode() if _LIB else "library not initialised")
