"""Times the block steps that the 4 GiB bench does not exercise (K4 BWT merge, K5 gap split, K6 export) on a
two-block text: python tools/time_block_steps.py [half_MiB]   (block = 2 halves; tail = another block)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from psascan_amd import api, extras
half = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
n = 4 * half
mid, e = half, 2 * half                      # block [0, e), tail [e, n)
d_text = extras.gen_text(n, 0, 0, seed=3)
L = extras.sort_halfblock(d_text, n, 0, mid)
R = extras.sort_halfblock(d_text, n, mid, e)
text_mid = int(api.download(d_text, np.uint8, 1, mid - 1)[0])
text_e = int(api.download(d_text, np.uint8, 1, e - 1)[0])

def timed(name, fn, units):
    api.sync(); t0 = time.perf_counter(); r = fn(); api.sync(); dt = time.perf_counter() - t0
    print(f"{name:34s} {dt * 1e3:8.2f} ms   {units / dt / 1e9:7.2f} G units/s", flush=True)
    return r

for rep in range(2):     # the first round pays for hipMalloc (the pool is empty), the second is steady state
    print(f'--- round {rep}', flush=True)
    # pass A: right half through rank(left BWT); rank at the tail end via a full stream of [e, n) is not needed here:
    # use the device path with a right context on [mid, e) (tail of pass A ends at e < n)
    rkL = api.rank_build(L["bwt"], half)
    gapA = api.gap_array(half, fill=None); gtA = api.zeros(4 * (half // 32 + 4))
    from psascan_amd import distributed as D
    ctx = D.context_len(e, n)
    gt_in = api.zeros(4 * ((half + ctx) // 32 + 4))          # gt of the right half w.r.t. e: from the sorter, shifted by ctx
    api.bitcopy(gt_in, ctx, R["gt_begin"], 0, half)
    finA, stA = timed("pass A stream (half through half)", lambda: api.stream_gap(rkL, L["i0"], text_mid, d_text.at(mid), half, gt_in, -1, gapA, gtA, 0, right_context=ctx, fresh_gap=True), half)
    rkL.free()
    bv = api.zeros(4 * (2 * half // 32 + 4))
    timed("K3 gap -> bitvector", lambda: api.gap_to_bitvector(gapA, half, bv, 2 * half), half)
    bbwt = api.DeviceBuffer(2 * half + 16)
    bi0 = timed("K4 merge_bwt", lambda: api.merge_bwt(L["bwt"], R["bwt"], half, half, L["i0"], R["i0"], text_mid, bv, bbwt), 2 * half)
    rkB = timed("K1 rank build (block BWT)", lambda: api.rank_build(bbwt, 2 * half), 2 * half)
    gapB = api.gap_array(2 * half, fill=None); gtB = api.zeros(4 * ((n - e) // 32 + 4))
    gt0 = api.zeros(4 * ((n - e) // 32 + 4))
    finB, stB = timed("pass B stream (tail through block)", lambda: api.stream_gap(rkB, bi0, text_e, d_text.at(e), n - e, gt0, 0, gapB, gtB, 0, fresh_gap=True), n - e)
    rkB.free()
    T = n - e
    mbvL = api.zeros(4 * ((half + half + T) // 32 + 4)); mbvR = api.zeros(4 * ((half + T) // 32 + 4))
    timed("K5 split_gap", lambda: api.split_gap(gapB, bv, half, half, T, mbvL, mbvR), 2 * half)
    g = timed("K6 mbv_to_gap (right half)", lambda: api.mbv_to_gap(mbvR, half + T, half), half)
    timed("K6 vbyte_encode", lambda: api.vbyte_encode(g, half + 1), half)
    print("pass A:", stA, "\npass B:", stB)
    for b in (gapA, gtA, gt_in, bv, bbwt, gapB, gtB, gt0, mbvL, mbvR, g):
        b.free()
