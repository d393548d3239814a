"""Fixed cost of the calls a small device pass is made of (the leaf merging of construct_sa issues thousands):
    python tools/time_small_pass.py [leaf MiB] [tail MiB]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from psascan_amd import api, extras
leaf = (int(sys.argv[1]) if len(sys.argv) > 1 else 2) << 20
T = (int(sys.argv[2]) if len(sys.argv) > 2 else 6) << 20
n = leaf + T + 4096
d_text = extras.gen_text(n, extras.MODE_ENGLISH, 0, seed=5)
r = extras.sort_halfblock(d_text, n, 0, leaf)
last = int(api.download(d_text, np.uint8, 1, leaf - 1)[0])
sc = api.search_ctx(d_text, n, n, None, [(0, leaf, r["psa_lo"], None)])
gt_in = extras.sort_halfblock(d_text, n, leaf, leaf + T)["gt_begin"]      # [text[j..) > text[leaf..)] for the tail positions
gt_out = api.zeros(4 * (T // 32 + 4))
def timeit(name, f, reps=30):
    f(); api.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    api.sync()
    print(f"{name:34s} {1e3 * (time.perf_counter() - t0) / reps:7.3f} ms")
init = int(api.initial_ranks(sc, [leaf + T])[0])
timeit("psg_initial_ranks (1 position)", lambda: api.initial_ranks(sc, [leaf + T]))
rk = [None]
def build():
    if rk[0] is not None: rk[0].free()
    rk[0] = api.rank_build(r["bwt"], leaf)
timeit("psg_rank_build", build)
gap = api.gap_array(leaf, fill=None)
st = [None]
def stream():
    _, st[0] = api.stream_gap(rk[0], r["i0"], last, d_text.at(leaf), T, gt_in, init, gap, gt_out, 0, fresh_gap=True, search=sc, tail_begin_abs=leaf, search_all=True)
timeit(f"psg_stream_gap_args ({T >> 20} Mi suffixes)", stream)
print("   stats:", st[0])
bv = api.zeros(4 * ((leaf + T) // 32 + 4))
timeit("psg_gap_to_bitvector", lambda: api.gap_to_bitvector(gap, leaf, bv, leaf + T))
timeit("psg_bitcopy", lambda: api.bitcopy(gt_out, T, r["gt_begin"], 0, leaf))
timeit("psg_halfblock_from_psa", lambda: api.halfblock_from_psa(sc, 0, leaf, r["psa_lo"]))
hbs = [{"beg": 0, "size": leaf, "psa_lo": r["psa_lo"], "psa_hi": None, "mbv": bv}, {"beg": leaf, "size": T, "psa_lo": api.zeros(4 * T), "psa_hi": None, "mbv": None}]
def plan():
    p = api.MergePlan(hbs); p.free()
timeit("psg_merge_plan_create (2 levels)", plan)
a = np.zeros(leaf, np.uint32)
timeit("upload 8 MiB pageable", lambda: api.upload(a).free())
timeit("psg_malloc+free 8 MiB", lambda: api.DeviceBuffer(8 << 20).free())
