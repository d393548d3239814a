"""Block-per-GPU schedule (north_star's multi-GPU split).

One process per GPU (torch.distributed: "nccl" = RCCL over xGMI on the node, "gloo" in the CPU tests and the
shared-GPU rehearsal).  The text is cut into `world` blocks, rank g owns block g = [b_g, e_g):

  local phase   sort the two halves, pass A (right half through the left half's rank), gap -> bitvector, BWT
                merge, rank over the block BWT -- process_block's steps 1-4 (partial_sufsort.hpp:166-500), no
                communication.  Product: the rank structure of the block, and the block's OWN gt slice: bits
                [text[j..) > text[b_g..)] for j in (b_g, e_g].
  rounds        round r = 1 .. world-1: ONE exchange of the gt slice every rank produced in round r-1 (point to point: to the
                left neighbour and, when that one is helped, its helper -- exchange_gt), then
                rank g streams chunk q = g + r (the text of block q) through its rank structure into its gap array
                (compute_gap, partial_sufsort.hpp:512-514).  The gt bits it needs -- [text[j..) > text[e_g..)] for
                j in chunk q -- are exactly what rank g+1 produced one round earlier while streaming the same chunk
                (stream.hpp:150, handed over at partial_sufsort.hpp:573-579); in round 1 they are rank g+1's own
                slice.  The pass writes the next slice.  Chunks are streamed near-to-far, so the start rank of a
                chunk (rank of the suffix at the chunk's end among the block's suffixes) does not come from the
                previous pass: it is found by string search over the block's partial SAs before the rounds
                (em_compute_initial_ranks.hpp:222-319).
  split         the block gap is split into the two halves' merge bitvectors (partial_sufsort.hpp:536-542).
  merge         output-range partition (merge.hpp:55-180 in closed form): rank d produces the entries
                [X_d, X_{d+1}) of the suffix array.  The positions every level's walk reaches at the range
                boundaries are computed level by level by the owner of the level (one small broadcast each); then
                ONE all-to-all brings every rank the slices of the merge bitvectors and partial SAs its range
                touches, and the merge runs locally on a plan over slices (psg_merge_plan_create_sliced).

`ops` supplies the compute: HipBlockOps = the C ABI on this rank's GPU; the CPU tests inject oracle-backed
stand-ins to check the schedule, the indexing of the exchanged bits and the slice arithmetic without a GPU.

BASELINE configs[3] (128 GiB of DNA, 8 blocks of 16 GiB, half-blocks of 2^33 symbols):
  * partial SAs are 40-bit values in two planes (u32 low words + bytes 32..39): the high plane travels through the
    search contexts, the merge pieces and the sliced plan like the low one (`wide`);
  * a rank never holds the whole text: its own block (+ a look-ahead) is resident, the chunk of the current round comes
    from a TextSource (generated / uploaded one round ahead), and the start ranks of the far chunks are searched through
    two text windows (psg_search_ctx.d_text2);
  * the merge runs in `rounds` sub-ranges of every rank's output range, one all-to-all each, so that the exchange
    buffers are a fraction of the 5 bytes per symbol a rank receives in total.
"""
import os as _os

import numpy as np


def block_bounds(n, world):
    """[b_0 = 0, b_1, ..., b_world = n]: equal blocks, every block at least two symbols"""
    bs = -(-n // world)
    bounds = [min(n, g * bs) for g in range(world)] + [n]
    if any(bounds[g + 1] - bounds[g] < 2 for g in range(world)):
        raise ValueError(f"text of {n} symbols is too short for {world} blocks")
    return bounds


def output_cuts(n, world):
    return [min(n, (n * r // world + 4095) // 4096 * 4096) for r in range(world)] + [n]


def slice_words(bounds):
    """int32 words of a gt slice (one bit per position of the largest block, padded)"""
    return (max(bounds[g + 1] - bounds[g] for g in range(len(bounds) - 1)) + 31) // 32 + 4


def helper_of(world):
    """Load balancing of the systolic schedule (SURVEY 8e caveat 1, 8f row 4): rank g streams world-1-g chunks, so the
    high ranks run out of work first.  Rank h = world-1-g HELPS rank g (g < world/2): once h has streamed its own last
    chunk (round world-h = g+1) it holds a replica of block g's rank structure and streams the right half of rank g's
    chunk of every later round into a gap array of its own, which rank g adds to its array at the end (the reference's
    streamers share one gap array among threads, compute_gap.hpp:114-124; across devices that is one reduce per pair).
    Every rank then streams (world-1)/2 chunks in total; a round still lasts as long as its slowest rank, which gives
    rounds of 1,1,1,.5,.5,.5,.5 chunk times at world = 8: 5 instead of 7 (utilisation 70 % instead of 50 %)."""
    return {g: world - 1 - g for g in range(world // 2) if world - 1 - g > g}


def helper_active(pairs, g, r, world):
    """does rank g's helper stream half of rank g's chunk in round r?"""
    return g in pairs and r >= g + 1 and g + r < world


def split_point(cb, ce):
    """the helper takes [cmid, ce): a multiple of 64 positions, so that gt words do not straddle the two parts"""
    return ce - ((ce - cb) // 2 + 63) // 64 * 64


def streams_for(pairs, d, r, world):
    """the block whose chunk rank d streams (a part of) in round r: its own, the one it helps, or None"""
    if d + r < world:
        return d
    g = next((g for g, h in pairs.items() if h == d), None)
    return g if g is not None and helper_active(pairs, g, r, world) else None


def gt_sources(pairs, x, r, world):
    """who holds the gt bits block x's chunk of round r is streamed against: rank x+1 produced them one round earlier
    (round 1: its own slice), and its helper the right half's share if it was at work then"""
    return [x + 1] + ([pairs[x + 1]] if helper_active(pairs, x + 1, r - 1, world) else [])


def exchange_gt(dist, ops, world, rank, pairs, r, prev, words):
    """The ONE exchange of round r: every slice goes to the ranks that stream against it -- its left neighbour and,
    when that one is helped, the helper: at most two point-to-point transfers per rank over xGMI (PSASCAN_GT_EXCHANGE=
    allgather: the older all-gather, G slices to everybody of which one or two are read).  -> {source rank: slice}"""
    if world == 1:
        return {}
    if _os.environ.get("PSASCAN_GT_EXCHANGE", "p2p") == "allgather":
        gathered = [ops.new_i32(words) for _ in range(world)]
        ops.before_collective()
        dist.all_gather(gathered, prev)
        ops.after_collective()
        return dict(enumerate(gathered))
    mine = streams_for(pairs, rank, r, world)
    need = gt_sources(pairs, mine, r, world) if mine is not None else []
    got = {src: (prev if src == rank else ops.new_i32(words)) for src in need}
    p2p = []
    for d in range(world):
        x = streams_for(pairs, d, r, world)
        if d != rank and x is not None and rank in gt_sources(pairs, x, r, world):
            p2p.append(dist.P2POp(dist.isend, prev, d))
    for src in need:
        if src != rank:
            p2p.append(dist.P2POp(dist.irecv, got[src], src))
    if p2p:
        ops.before_collective()
        for req in dist.batch_isend_irecv(p2p):
            req.wait()
        ops.after_collective()
    return got


def run(dist, ops, world, rank, n, stats=None):
    """The whole schedule on this rank.  Returns (x0, x1, sa5 bytes of the output entries [x0, x1))."""
    bounds = block_bounds(n, world)
    b, e = bounds[rank], bounds[rank + 1]
    mid = b + (e - b) // 2
    words = slice_words(bounds)
    st = ops.local_block(b, mid, e, words)                      # -> state with .own_gt (int32 tensor of `words`)
    # start ranks of the chunks this rank will stream: rank of text[e_q..) among the block's suffixes
    chunk_ends = [bounds[q + 1] for q in range(rank + 1, world)]
    start = ops.start_ranks(st, chunk_ends) if chunk_ends else []
    pairs = helper_of(world) if getattr(ops, "helpers", False) and world >= 3 else {}
    helping = next((g for g, h in pairs.items() if h == rank), None)
    hp = None
    if pairs:
        # ---- every owner hands its helper the block BWT (one all-to-all: only the pairs carry data) and block_i0, the last
        # symbol and the start ranks of its chunks (one small all-gather)
        meta = ops.i64_from(ops.block_meta(st) + [int(x) for x in start] + [0] * (world - len(start)))
        metas = [ops.new_i64(3 + world) for _ in range(world)]
        dist.all_gather(metas, meta)
        send_sizes, recv_sizes = [0] * world, [0] * world
        bwt_t = ops.new_i32(1)
        if rank in pairs:
            bwt_t = ops.export_bwt(st)
            send_sizes[pairs[rank]] = int(bwt_t.numel())
        if helping is not None:
            recv_sizes[helping] = (bounds[helping + 1] - bounds[helping] + 3) // 4
        recv_t = ops.new_i32(max(1, sum(recv_sizes)))
        ops.before_collective()
        dist.all_to_all_single(recv_t[: sum(recv_sizes)], bwt_t[: sum(send_sizes)], recv_sizes, send_sizes)
        ops.after_collective()
        if helping is not None:
            m = ops.to_numpy_i64(metas[helping])
            hp = ops.import_block(recv_t, bounds[helping], bounds[helping + 1], int(m[0]), int(m[1]), [int(x) for x in m[3:]])
        del recv_t, bwt_t
    prev = st.own_gt
    for r in range(1, world):
        gathered = exchange_gt(dist, ops, world, rank, pairs, r, prev, words)   # the ONE exchange of the round

        def gt_in_for(x):
            """gt bits of chunk x + r w.r.t. the end of block x: what rank x+1 (and its helper) wrote one round earlier"""
            src = gt_sources(pairs, x, r, world)
            t = gathered[src[0]]
            for y in src[1:]:
                t = ops.bits_or(t, gathered[y])
            return t
        q = rank + r
        if q < world:
            cb, ce = bounds[q], bounds[q + 1]
            lo_end = split_point(cb, ce) if helper_active(pairs, rank, r, world) else ce
            prev = ops.stream(st, cb, ce, gt_in_for(rank), int(start[q - rank - 1]), words, first=(r == 1), part=(cb, lo_end))
            if stats is not None:
                stats.append((rank, q, getattr(st, "last_stats", None)))
        elif helping is not None and helper_active(pairs, helping, r, world):
            qh = helping + r
            cb, ce = bounds[qh], bounds[qh + 1]
            prev = ops.stream(hp, cb, ce, gt_in_for(helping), hp.start[qh - helping - 1], words, first=not getattr(hp, "streamed", False), part=(split_point(cb, ce), ce))
            hp.streamed = True
            if stats is not None:
                stats.append((rank, qh, getattr(hp, "last_stats", None)))
        else:
            prev = ops.new_i32(words)
    if pairs:
        # ---- the helpers' gap arrays go home: one all-to-all (pairs only), added slot by slot
        send_sizes, recv_sizes = [0] * world, [0] * world
        gap_t = ops.new_i32(1)
        if helping is not None and getattr(hp, "streamed", False):
            gap_t = ops.export_gap(hp)
            send_sizes[helping] = int(gap_t.numel())
        if rank in pairs and any(helper_active(pairs, rank, r, world) for r in range(1, world)):
            recv_sizes[pairs[rank]] = e - b + 1
        recv_t = ops.new_i32(max(1, sum(recv_sizes)))
        ops.before_collective()
        dist.all_to_all_single(recv_t[: sum(recv_sizes)], gap_t[: sum(send_sizes)], recv_sizes, send_sizes)
        ops.after_collective()
        if sum(recv_sizes):
            ops.add_gap(st, recv_t)
        del recv_t, gap_t
        if hp is not None:
            ops.free_block(hp)
    hbs = ops.finish(st, n - e)                                 # [{beg, size, mbv, nbits, psa, psa_hi}] left half, right half
    return merge_ranges(dist, ops, world, rank, n, bounds, hbs)


def merge_ranges(dist, ops, world, rank, n, bounds, my_hbs):
    """Output-range partitioned merge.  my_hbs: this rank's two half-blocks (mbv = None for the very last one)."""
    S = max(1, int(getattr(ops, "merge_rounds", 1)))                 # sub-ranges per rank: one all-to-all each
    wide = bool(getattr(ops, "force_wide", False)) or max(bounds[g + 1] - bounds[g] for g in range(world)) // 2 + 1 >= (1 << 32)
    H = 2 * world
    sizes = []
    for g in range(world):
        b, e = bounds[g], bounds[g + 1]
        mid = b + (e - b) // 2
        sizes += [(b, mid - b), (mid, e - mid)]
    later = [0] * (H + 1)
    for h in range(H - 1, -1, -1):
        later[h] = later[h + 1] + sizes[h][1]
    nbits = [later[h] if h + 1 < H else 0 for h in range(H)]     # level h: own elements + everything behind it
    Xr = output_cuts(n, world)                                        # rank d produces [Xr[d], Xr[d+1])
    X = []                                                            # ... in S sub-ranges: boundary d * S + s
    for d in range(world):
        for k in range(S):
            X.append(min(Xr[d + 1], Xr[d] + ((Xr[d + 1] - Xr[d]) * k // S + 4095) // 4096 * 4096) if k else Xr[d])
    X.append(n)
    nb = len(X)
    # ---- positions of the range boundaries on every level: q_0 = X, q_{h+1} = rank1(mbv_h, q_h)
    q = np.array(X, np.int64)
    qs, ones_al, cur = [], [], []
    for h in range(H):
        qs.append(q.copy())
        if h == H - 1:
            cur.append(q.copy())
            ones_al.append(np.zeros(nb, np.int64))
            break
        t = ops.new_i64(2 * nb)
        if h // 2 == rank:
            hb = my_hbs[h % 2]
            qa = (q // 4096) * 4096
            r1 = ops.rank1(hb["mbv"], nbits[h], np.concatenate([q, qa]))
            t = ops.i64_from(r1)
        if world > 1:
            ops.before_collective()
            dist.broadcast(t, src=h // 2)
            ops.after_collective()
        r1 = ops.to_numpy_i64(t)
        ones, oa = r1[:nb], r1[nb:]
        cur.append(q - ones)
        ones_al.append(oa)
        q = ones
    for h in range(H):
        assert cur[h][0] == 0 and cur[h][-1] == sizes[h][1], (h, cur[h], sizes[h])
    # ---- what dest d gets of level h in round s: mbv words [fw, lw), PSA elements [c0, c1) (+ their high bytes, packed 4 per word)
    def piece(h, d, s):
        j = d * S + s
        c0, c1 = int(cur[h][j]), int(cur[h][j + 1])
        hw = (c1 - c0 + 3) // 4 if wide else 0
        if h == H - 1:
            return 0, 0, c0, c1, hw
        q0, q1 = int(qs[h][j]), int(qs[h][j + 1])
        total_words = (nbits[h] + 31) // 32
        fw = min((q0 // 4096) * 128, total_words)
        lw = min(total_words, (q1 + 31) // 32 + 1)          # + 1: a 32-bit fetch at the range end may look one word ahead
        return fw, max(fw, lw), c0, c1, hw

    outs = []
    for s in range(S):
        send_parts, send_sizes = [], []
        for d in range(world):
            tot = 0
            for k in (0, 1):
                h = 2 * rank + k
                fw, lw, c0, c1, hw = piece(h, d, s)
                hb = my_hbs[k]
                if lw > fw:
                    send_parts.append(ops.mbv_words(hb["mbv"], fw, lw - fw))
                if c1 > c0:
                    send_parts.append(ops.psa_words(hb["psa"], c0, c1 - c0))
                    if wide:
                        send_parts.append(ops.psa_hi_words(hb.get("psa_hi"), c0, c1 - c0))
                tot += (lw - fw) + (c1 - c0) + hw
            send_sizes.append(tot)
        recv_sizes = []
        for src in range(world):
            tot = 0
            for k in (0, 1):
                fw, lw, c0, c1, hw = piece(2 * src + k, rank, s)
                tot += (lw - fw) + (c1 - c0) + hw
            recv_sizes.append(tot)
        send_t = ops.cat_i32(send_parts, sum(send_sizes))
        recv_t = ops.new_i32(max(1, sum(recv_sizes)))
        if world > 1:
            ops.before_collective()
            dist.all_to_all_single(recv_t[: sum(recv_sizes)], send_t[: sum(send_sizes)], recv_sizes, send_sizes)
            ops.after_collective()
        else:
            recv_t = send_t
        # ---- local merge over the slices
        levels, off = [], 0
        j = rank * S + s
        for h in range(H):
            fw, lw, c0, c1, hw = piece(h, rank, s)
            lv = {"beg": sizes[h][0], "size": sizes[h][1], "nbits": nbits[h], "first_word": fw, "n_words": lw - fw,
                  "ones_before": int(ones_al[h][j]), "psa_first": c0, "psa_count": c1 - c0, "mbv_off": off, "psa_off": off + (lw - fw),
                  "psa_hi_off": (off + (lw - fw) + (c1 - c0)) if wide else None}
            off += (lw - fw) + (c1 - c0) + hw
            levels.append(lv)
        outs.append(ops.merge_slices(levels, recv_t, X[j], X[j + 1]))
        del send_t, recv_t, send_parts
    x0, x1 = Xr[rank], Xr[rank + 1]
    if any(o is None for o in outs):
        return x0, x1, None
    return x0, x1, (np.concatenate(outs) if len(outs) > 1 else outs[0])


class WholeText:
    """TextSource for a text that is resident in HBM as a whole (small texts, tests)."""

    def __init__(self, d_text, n):
        self.d_text, self.n = d_text, n

    def window(self, lo, hi):
        """-> (pointer to text position 0, first, last): text[first .. last) is readable there, first <= lo, last >= hi"""
        return self.d_text.ptr, 0, self.n

    def prefetch(self, lo, hi):
        pass


class ChunkedText:
    """TextSource for a text that no rank holds as a whole: `load(lo, hi) -> DeviceBuffer` with text[lo .. hi) produces
    any range (bench: the seeded generator on the device; a real run: an upload from the memory-mapped file).  The rank's
    own block (+ look-ahead) is loaded once and kept; other ranges live in two rotating buffers, the next one loaded
    while the current one is in use (prefetch)."""

    def __init__(self, load, n, own_lo, own_hi):
        self.load, self.n = load, n
        self.own = (load(own_lo, own_hi), own_lo, own_hi)
        self.slots = []                                     # [(buffer, lo, hi)], at most two

    def _find(self, lo, hi):
        for buf, a, b in [self.own] + self.slots:
            if a <= lo and hi <= b:
                return buf, a, b
        return None

    def prefetch(self, lo, hi):
        if self._find(lo, hi) is None:
            if len(self.slots) >= 2:
                self.slots.pop(0)[0].free()
            self.slots.append((self.load(lo, hi), lo, hi))

    def window(self, lo, hi):
        self.prefetch(lo, hi)
        buf, a, b = self._find(lo, hi)
        return buf.ptr - a, a, b


class HipBlockOps:
    """The schedule's compute on this rank's GPU through the C ABI.  comm = "cuda": tensors handed to the
    collectives are torch CUDA tensors (NCCL/RCCL; the library is put on torch's stream by the caller, so kernels and
    collectives are ordered on that one stream without device-wide syncs); comm = "cpu": CPU tensors (gloo; data is
    staged through the library's copies -- shared-GPU rehearsal).  text: a DeviceBuffer with the whole text or a
    TextSource (WholeText / ChunkedText)."""
    LOOKAHEAD = 1 << 16                                         # text kept behind a block / a searched position for comparisons that read on

    def __init__(self, torch, api, text, n, sorter, comm="cuda", max_chains=0, keep_output_on_device=False, merge_rounds=1, force_wide=False,
                 check_text=None, helpers=False):
        self.torch, self.api, self.n, self.sorter, self.comm, self.max_chains = torch, api, n, sorter, comm, max_chains
        self.helpers = helpers                                 # idle ranks stream half of the busy ranks' chunks (helper_of)
        self.text = text if hasattr(text, "window") else WholeText(text, n)
        self.keep_output_on_device = keep_output_on_device
        self.merge_rounds, self.force_wide = merge_rounds, force_wide
        self.check_text = check_text                           # (d_text of the whole text, samples): every round's output is checked on the device
        self.check_acc = [0, 0]                                # bad pairs, sum of entries mod 2^64
        self.d_out = None

    # ---- tensors
    def new_i32(self, k):
        return self.torch.zeros(int(k), dtype=self.torch.int32, device=self.comm)

    def new_i64(self, k):
        return self.torch.zeros(int(k), dtype=self.torch.int64, device=self.comm)

    def i64_from(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, np.int64)).to(self.comm)

    def to_numpy_i64(self, t):
        return t.cpu().numpy().astype(np.int64)

    def cat_i32(self, parts, total):
        return self.torch.cat(parts) if parts else self.new_i32(1)

    def before_collective(self):
        # every library call has completed when it returns and, with comm = "cuda", library and collectives share one
        # stream: nothing to wait for (round 2 drained the whole device on both sides of every collective)
        pass

    def after_collective(self):
        pass

    def _to_dev(self, t, nwords):
        """device pointer of an int32 tensor's first nwords (uploads CPU tensors)"""
        if self.comm == "cuda":
            return t, t.data_ptr()
        buf = self.api.upload(t[:nwords].numpy(), pad_to=16)
        return buf, buf.ptr

    def _from_dev(self, buf_or_ptr, nwords):
        """int32 tensor of nwords from a device buffer"""
        a = self.api.download(buf_or_ptr, np.int32, nwords)
        return self.torch.from_numpy(a).to(self.comm)

    def _slice_out(self, words):
        """(tensor, device pointer, finish()) of a zeroed gt slice the kernels write: with comm = "cuda" the tensor IS
        the device buffer (no copy); with comm = "cpu" a device buffer that finish() downloads into the tensor"""
        if self.comm == "cuda":
            t = self.new_i32(words)
            return t, t.data_ptr(), (lambda: t)
        buf = self.api.zeros(4 * words)
        return None, buf.ptr, (lambda: self._from_dev(buf, words))

    def _text_ptr(self, lo, hi):
        """device pointer of text position `lo` inside a window that holds text[lo .. hi)"""
        base, a, b = self.text.window(lo, min(self.n, hi))
        return base + lo

    def _search_ctx(self, lo, hi, parts, pattern=None):
        """search context over the window that holds text[lo .. hi) (comparisons read on to the end of the text);
        pattern = (pos, len): the searched position lies in a second window"""
        api, n = self.api, self.n
        base, a, b = self.text.window(lo, min(n, hi))
        win = None if (a == 0 and b == n) else (a, b)
        w2 = None
        if pattern is not None and win is not None and not (a <= pattern[0] and pattern[0] + pattern[1] <= b):
            p0, p1 = pattern[0], min(n, pattern[0] + pattern[1])
            base2, a2, b2 = self.text.window(p0, p1)
            w2 = (base2 + a2, a2, b2)
        return api.search_ctx(base + (a if win else 0), n, n, None, parts, window=win, window2=w2)

    def sym(self, pos):
        return int(self.api.download(self._text_ptr(pos, pos + 1), np.uint8, 1)[0])

    # ---- local phase
    def local_block(self, b, mid, e, words):
        api, n = self.api, self.n
        ls, rs, bs = mid - b, e - mid, e - b
        own_hi = min(n, e + self.LOOKAHEAD)

        class State:
            pass
        st = State()
        st.b, st.mid, st.e = b, mid, e
        R = self.sorter(None, mid, e, None)
        L = self.sorter(None, b, mid, None)
        st.L, st.R = L, R
        st.last_left, st.last = self.sym(mid - 1), self.sym(e - 1)
        rankL = api.rank_build(L["bwt"], ls)
        gapA = api.gap_array(ls, fill=None)
        gtA = api.zeros(4 * ((rs + 31) // 32 + 4))
        if L.get("initA") is not None:                        # found when the half-block was sorted (its partial SA has left the device since)
            initA = int(L["initA"])
        else:
            scA = self._search_ctx(b, own_hi, [(b, ls, L["psa_lo"], L.get("psa_hi"))])     # direct comparison, reading on behind the block
            initA = int(api.initial_ranks(scA, [e])[0])
        api.stream_gap(rankL, L["i0"], st.last_left, self._text_ptr(mid, e), rs, R["gt_begin"], initA, gapA, gtA, self.max_chains, fresh_gap=True)
        rankL.free()
        st.bvA = api.zeros(4 * ((bs + 31) // 32 + 2))
        assert api.gap_to_bitvector(gapA, ls, st.bvA, bs) == bs
        gapA.free()
        _, own, done = self._slice_out(words)
        api.bitcopy(own, 0, gtA, 0, rs)                       # positions (mid, e]: u = e - j
        api.bitcopy(own, rs, L["gt_begin"], 0, ls)            # positions (b, mid]
        api.sync()
        st.own_gt = done()
        st.rank = None
        if e < n:                                             # not the last block: it will be streamed through
            d_bbwt = api.DeviceBuffer(bs + 16)
            st.block_i0 = api.merge_bwt(L["bwt"], R["bwt"], ls, rs, L["i0"], R["i0"], st.last_left, st.bvA, d_bbwt)
            st.rank = api.rank_build(d_bbwt, bs)
            st.bbwt = d_bbwt if self.helpers else None        # a helper rank builds its replica of the rank structure from it
            if not self.helpers:
                d_bbwt.free()
            st.gap = api.gap_array(bs, fill=None)
        for hb in (L, R):                                     # BWT and gt bits of the halves are not needed any more
            for key in ("bwt", "gt_begin"):
                if hb.get(key) is not None and hb.get("keep_inputs") is not True:
                    hb[key].free()
                    hb[key] = None
        return st

    def start_ranks(self, st, positions):
        """rank of text[p..) among the block's suffixes for the ends p of the far chunks: string search through two text
        windows -- the block (+ look-ahead) and a piece behind p (em_compute_initial_ranks.hpp:222-319)"""
        cached = st.L.get("start_ranks")                      # {position: rank}: searched when the block was sorted
        if cached is not None:
            return [int(cached[int(p)]) for p in positions]
        parts = [(st.b, st.mid - st.b, st.L["psa_lo"], st.L.get("psa_hi")), (st.mid, st.e - st.mid, st.R["psa_lo"], st.R.get("psa_hi"))]
        out = []
        for p in positions:
            if p >= self.n:
                out.append(0)                                 # the empty suffix is the smallest
                continue
            sc = self._search_ctx(st.b, min(self.n, st.e + self.LOOKAHEAD), parts, pattern=(p, self.LOOKAHEAD))
            out.append(int(self.api.initial_ranks(sc, [p])[0]))
        return out

    def stream(self, st, cb, ce, gt_in_t, start_rank, words, first, part=None):
        """stream the positions part = [lo, hi) of chunk [cb, ce) through st's rank structure (default: the whole chunk).
        gt_in_t / the returned slice: bit u <-> position ce - u.  hi == ce: start_rank is the exact rank at ce; hi < ce
        (the left part, its right half is streamed by a helper rank): the start rank is found inside a right context of
        up to 64 Ki positions (psg_stream_gap_ctx)."""
        api = self.api
        lo, hi = part if part is not None else (cb, ce)
        T = hi - lo
        ctx = 0 if hi == ce else min(self.LOOKAHEAD, ce - hi) // 64 * 64
        assert hi == ce or (ctx >= 64 and (ce - hi) % 64 == 0), "a split chunk needs 64-aligned parts"
        keep, gin = self._to_dev(gt_in_t, (ce - cb + 31) // 32 + 1)
        _, gout, done = self._slice_out(words)
        base, _, _ = self.text.window(cb, ce)
        _, s = api.stream_gap(st.rank, st.block_i0, st.last, base + lo, T, gin + (ce - hi - ctx) // 8, start_rank if hi == ce else -1, st.gap, gout + (ce - hi) // 8,
                              self.max_chains, right_context=ctx, fresh_gap=first)
        if ce < self.n:
            self.text.prefetch(ce, min(self.n, ce + (ce - cb)))   # the next round's chunk (a TextSource that loads in the background overlaps it with the collective)
        st.last_stats = s
        return done()

    # ---- helper ranks (helper_of)
    def bits_or(self, a, b):
        return self.torch.bitwise_or(a, b)

    def block_meta(self, st):
        return [int(getattr(st, "block_i0", 0)), int(st.last), 0]

    def export_bwt(self, st):
        """the block BWT as an int32 tensor (padded to whole words)"""
        bs = st.e - st.b
        return self._words_from(st.bbwt, 0, (bs + 3) // 4, bs)

    def import_block(self, bwt_t, gb, ge, block_i0, last, start):
        """replica of block [gb, ge)'s rank structure on this (helper) rank + a gap array of its own"""
        api = self.api

        class Replica:
            pass
        hp = Replica()
        hp.b, hp.e, hp.block_i0, hp.last, hp.start = gb, ge, block_i0, last, start
        keep, ptr = self._to_dev(bwt_t, (ge - gb + 3) // 4)
        hp.rank = api.rank_build(ptr, ge - gb)
        hp.gap = api.gap_array(ge - gb, fill=None)
        return hp

    def export_gap(self, hp):
        """the m + 1 counters of the helper's gap array (32-bit counters: a helper streams fewer than 2^32 suffixes)"""
        m = hp.e - hp.b
        assert self.api.download(hp.gap, np.uint32, 4, 4 * ((m + 1 + 3) // 4 * 4))[0] == 0, "excess entries in a helper's gap array"
        return self._words_from(hp.gap, 0, m + 1)

    def add_gap(self, st, t):
        """gap[j] += t[j]: the helper's counts join the owner's (no counter comes near 2^32 here: asserted on the helper)"""
        api = self.api
        m = st.e - st.b
        if self.comm == "cuda":
            mine = self._words_from(st.gap, 0, m + 1)
            mine += t[: m + 1]
            api.check(api.lib().psg_d2d(st.gap.ptr, mine.data_ptr(), 4 * (m + 1)))
            api.sync()
        else:
            mine = api.download(st.gap, np.uint32, m + 1)
            tot = mine.astype(np.uint64) + t[: m + 1].numpy().view(np.uint32)
            assert tot.max() < (1 << 32)
            tot32 = np.ascontiguousarray(tot.astype(np.uint32))           # (kept alive across the call)
            api.check(api.lib().psg_h2d(st.gap.ptr, tot32.ctypes.data, 4 * (m + 1)))

    def free_block(self, hp):
        hp.rank.free()
        hp.gap.free()

    def finish(self, st, T):
        api = self.api
        ls, rs, bs = st.mid - st.b, st.e - st.mid, st.e - st.b
        L = {"beg": st.b, "size": ls, "psa": st.L["psa_lo"], "psa_hi": st.L.get("psa_hi"), "mbv": None, "nbits": 0}
        R = {"beg": st.mid, "size": rs, "psa": st.R["psa_lo"], "psa_hi": st.R.get("psa_hi"), "mbv": None, "nbits": 0}
        if st.rank is None:                                   # last block: the left half's gap is its merge bitvector
            L["mbv"], L["nbits"] = st.bvA, bs
            return [L, R]
        st.rank.free()
        if getattr(st, "bbwt", None) is not None:
            st.bbwt.free()
        mbvL = api.DeviceBuffer(4 * ((bs + T + 31) // 32 + 2))
        mbvR = api.DeviceBuffer(4 * ((rs + T + 31) // 32 + 2))
        api.split_gap(st.gap, st.bvA, ls, rs, T, mbvL, mbvR)
        st.gap.free()
        L["mbv"], L["nbits"], R["mbv"], R["nbits"] = mbvL, bs + T, mbvR, rs + T
        return [L, R]

    # ---- merge
    def rank1(self, mbv, nbits, positions):
        return self.api.bits_rank1(mbv, nbits, positions)

    def _words_from(self, src, byte_off, n_words, nbytes=None):
        """int32 tensor with n_words words copied from a device buffer or a host array (partial SAs that live in pinned
        host memory) at byte_off; nbytes < 4 * n_words: the rest is zero"""
        nbytes = 4 * n_words if nbytes is None else nbytes
        if isinstance(src, np.ndarray):                       # (pinned host memory: one DMA into the tensor)
            if self.comm == "cuda":
                t = self.torch.zeros(n_words, dtype=self.torch.int32, device="cuda")
                if nbytes:
                    self.api.check(self.api.lib().psg_h2d(t.data_ptr(), src.ctypes.data + byte_off, nbytes))
                return t
            raw = np.zeros(4 * n_words, np.uint8)
            raw[:nbytes] = src.view(np.uint8)[byte_off: byte_off + nbytes]
            return self.torch.from_numpy(raw.view(np.int32))
        if self.comm == "cuda":
            t = self.torch.zeros(n_words, dtype=self.torch.int32, device="cuda")
            self.api.check(self.api.lib().psg_d2d(t.data_ptr(), src.ptr + byte_off, nbytes))
            return t
        raw = np.zeros(4 * n_words, np.uint8)
        raw[:nbytes] = self.api.download(src, np.uint8, nbytes, byte_off)
        return self.torch.from_numpy(raw.view(np.int32))

    def mbv_words(self, mbv, first_word, n_words):
        return self._words_from(mbv, 4 * first_word, n_words)

    def psa_words(self, psa, first, count):
        return self._words_from(psa, 4 * first, count)

    def psa_hi_words(self, psa_hi, first, count):
        """bits 32..39 of `count` entries, four per word (zeros for a half-block below 2^32 symbols)"""
        nw = (count + 3) // 4
        if psa_hi is None:
            return self.new_i32(nw)
        return self._words_from(psa_hi, first, nw, count)

    def merge_slices(self, levels, recv_t, x0, x1):
        api = self.api
        keep, base = self._to_dev(recv_t, recv_t.numel())
        descs = []
        for lv in levels:
            descs.append(dict(lv, d_mbv=(base + 4 * lv["mbv_off"]) if lv["n_words"] else None, d_psa=(base + 4 * lv["psa_off"]) if lv["psa_count"] else None,
                              d_psa_hi=(base + 4 * lv["psa_hi_off"]) if (lv.get("psa_hi_off") is not None and lv["psa_count"]) else None))
        plan = api.SlicedMergePlan(descs)
        d_out = api.DeviceBuffer(5 * (x1 - x0) + 16)
        plan.run(x0, x1 - x0, d_out)
        plan.free()
        if self.check_text is not None and x1 > x0:
            from . import extras
            bad, sm = extras.check_sa5(self.check_text[0], self.n, d_out, x1 - x0, samples=self.check_text[1], seed=7 + x0 % 1000)
            self.check_acc[0] += bad
            self.check_acc[1] = (self.check_acc[1] + sm) % (1 << 64)
        self.d_out = d_out
        if self.keep_output_on_device:
            return None
        return api.download(d_out, np.uint8, 5 * (x1 - x0))
