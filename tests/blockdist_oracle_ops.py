"""Oracle-backed stand-ins for psascan_amd.blockdist's `ops` (test infrastructure): the block-per-GPU schedule on
CPU tensors, every compute step done by the CPU oracle.  Checks the schedule itself -- which slice goes where, the
indexing of the exchanged gt bits, the start ranks of near-to-far chunks, the slice arithmetic of the merge."""
import numpy as np
import torch

import orc


def _words(bits_u8_packed, words):
    raw = np.zeros(words * 4, np.uint8)
    k = min(len(raw), len(bits_u8_packed))
    raw[:k] = bits_u8_packed[:k]
    return torch.from_numpy(raw.view(np.int32).copy())


class OracleBlockOps:
    def __init__(self, text, merge_rounds=1, force_wide=False, helpers=False):
        self.merge_rounds, self.force_wide, self.helpers = merge_rounds, force_wide, helpers
        self.t = np.ascontiguousarray(text, np.uint8)
        self.n = len(self.t)
        self.sa = orc.suffix_array(self.t)
        self.isa = orc.inverse(self.sa)

    # ---- tensors
    def new_i32(self, k): return torch.zeros(int(k), dtype=torch.int32)
    def new_i64(self, k): return torch.zeros(int(k), dtype=torch.int64)
    def i64_from(self, a): return torch.from_numpy(np.ascontiguousarray(a, np.int64))
    def to_numpy_i64(self, t): return t.numpy().astype(np.int64)
    def cat_i32(self, parts, total): return torch.cat(parts) if parts else self.new_i32(1)
    def before_collective(self): pass
    def after_collective(self): pass

    def _rank_of(self, lo, hi, p):
        return int((self.isa[lo:hi] < (self.isa[p] if p < self.n else -1)).sum())

    # ---- local phase
    def local_block(self, b, mid, e, words):
        t, n = self.t, self.n
        ls, rs, bs = mid - b, e - mid, e - b

        class State:
            pass
        st = State()
        st.b, st.mid, st.e = b, mid, e
        st.lpsa, lbwt, li0, lgt = orc.partial_sa(t, self.sa, self.isa, b, mid)
        st.rpsa, rbwt, ri0, rgt = orc.partial_sa(t, self.sa, self.isa, mid, e)
        gapA, gtA, _ = orc.stream_pass(orc.Rank(lbwt), li0, int(t[mid - 1]), t, mid, e, rgt, self._rank_of(b, mid, e))
        st.bvA, nb = orc.gap_to_bitvector(gapA, ls)
        assert nb == bs
        own = np.concatenate([orc.bits(gtA, rs), orc.bits(lgt, ls)])      # u = e - j: right half's positions first
        st.own_gt = _words(orc.packbits(list(own) + [0] * 64), words)
        st.rank = None
        if e < n:
            bbwt, st.bi0 = orc.merge_bwt(lbwt, rbwt, li0, ri0, int(t[mid - 1]), st.bvA)
            st.bbwt = bbwt
            st.rank = orc.Rank(bbwt)
            st.gap = np.zeros(bs + 1, np.uint64)
        return st

    def start_ranks(self, st, positions):
        return [self._rank_of(st.b, st.e, p) for p in positions]

    def stream(self, st, cb, ce, gt_in_t, start_rank, words, first, part=None):
        lo, hi = part if part is not None else (cb, ce)
        assert (ce - hi) % 64 == 0
        gt_in = np.ascontiguousarray(gt_in_t.numpy()).view(np.uint8)[(ce - hi) // 8:]          # bit u' = hi - j = u - (ce - hi)
        init = start_rank if hi == ce else self._rank_of(st.b, st.e, hi)                          # (the device finds it inside a right context)
        g, gto, _ = orc.stream_pass(st.rank, st.bi0, int(self.t[st.e - 1]), self.t, lo, hi, np.ascontiguousarray(gt_in), init)
        st.gap += g
        out = np.zeros(words * 4, np.uint8)
        nb = (hi - lo + 7) // 8
        out[(ce - hi) // 8: (ce - hi) // 8 + nb] = gto[:nb]
        return torch.from_numpy(out.view(np.int32).copy())

    # ---- helper ranks
    def bits_or(self, a, b): return torch.bitwise_or(a, b)
    def block_meta(self, st): return [int(getattr(st, "bi0", 0)), int(self.t[st.e - 1]), 0]

    def export_bwt(self, st):
        raw = np.zeros((len(st.bbwt) + 3) // 4 * 4, np.uint8)
        raw[: len(st.bbwt)] = st.bbwt
        return torch.from_numpy(raw.view(np.int32).copy())

    def import_block(self, bwt_t, gb, ge, block_i0, last, start):
        class Replica:
            pass
        hp = Replica()
        hp.b, hp.e, hp.bi0, hp.start = gb, ge, block_i0, start
        hp.bbwt = np.ascontiguousarray(bwt_t.numpy()).view(np.uint8)[: ge - gb].copy()
        hp.rank = orc.Rank(hp.bbwt)
        hp.gap = np.zeros(ge - gb + 1, np.uint64)
        assert last == int(self.t[ge - 1])
        return hp

    def export_gap(self, hp):
        assert hp.gap.max() < (1 << 31)
        return torch.from_numpy(hp.gap.astype(np.uint32).view(np.int32).copy())

    def add_gap(self, st, t):
        st.gap += t.numpy().view(np.uint32)[: len(st.gap)].astype(np.uint64)

    def free_block(self, hp): pass

    def finish(self, st, T):
        ls, rs, bs = st.mid - st.b, st.e - st.mid, st.e - st.b
        L = {"beg": st.b, "size": ls, "psa": st.lpsa, "mbv": None, "nbits": 0}
        R = {"beg": st.mid, "size": rs, "psa": st.rpsa, "mbv": None, "nbits": 0}
        if st.rank is None:
            L["mbv"], L["nbits"] = st.bvA, bs
            return [L, R]
        assert int(st.gap.sum()) == T
        lg, rg = orc.left_gap(st.gap, st.bvA, ls, rs), orc.right_gap(st.gap, st.bvA, ls, rs)
        L["mbv"], L["nbits"] = orc.gap_to_bitvector(lg, ls)
        R["mbv"], R["nbits"] = orc.gap_to_bitvector(rg, rs)
        assert L["nbits"] == bs + T and R["nbits"] == rs + T
        return [L, R]

    # ---- merge
    def rank1(self, mbv, nbits, positions):
        cs = np.concatenate([[0], np.cumsum(orc.bits(mbv, nbits).astype(np.int64))])
        return cs[np.asarray(positions, np.int64)]

    def mbv_words(self, mbv, first_word, n_words):
        raw = np.zeros((first_word + n_words) * 4 + 8, np.uint8)
        raw[: len(mbv)] = mbv[: len(raw)]
        return torch.from_numpy(raw.view(np.int32)[first_word: first_word + n_words].copy())

    def psa_words(self, psa, first, count):
        return torch.from_numpy(np.ascontiguousarray(psa[first: first + count]).astype(np.uint32).view(np.int32).copy())

    def psa_hi_words(self, psa_hi, first, count):
        """bits 32..39 of the entries, four per word: all zero at test sizes -- a marker byte 0 is what must come back"""
        return torch.zeros((count + 3) // 4, dtype=torch.int32)

    def merge_slices(self, levels, recv_t, x0, x1):
        """the level walk of merge.hpp:123-158 in closed form, on the slices only"""
        buf = recv_t.numpy()
        out = np.zeros(x1 - x0, np.int64)
        slots = np.arange(x1 - x0)                 # output slots still to be filled, in order
        q0 = x0                                     # their positions on the current level: [q0, q0 + len(slots))
        H = len(levels)
        for h, lv in enumerate(levels):
            psa = buf[lv["psa_off"]: lv["psa_off"] + lv["psa_count"]].view(np.uint32).astype(np.int64)
            if lv.get("psa_hi_off") is not None:           # wide: the high plane travels behind the low one
                hi = buf[lv["psa_hi_off"]: lv["psa_hi_off"] + (lv["psa_count"] + 3) // 4].view(np.uint8)[: lv["psa_count"]].astype(np.int64)
                psa = psa + (hi << 32)
            if h == H - 1:
                idx = q0 + np.arange(len(slots)) - lv["psa_first"]
                assert len(slots) == 0 or (idx.min() >= 0 and idx.max() < lv["psa_count"])
                out[slots] = lv["beg"] + psa[idx]
                break
            words = buf[lv["mbv_off"]: lv["mbv_off"] + lv["n_words"]].view(np.uint8)
            bits = np.unpackbits(words, bitorder="little")
            first_bit = lv["first_word"] * 32
            assert len(slots) == 0 or (q0 >= first_bit and q0 + len(slots) <= first_bit + len(bits)), (h, q0, len(slots), first_bit, len(bits))
            ones_before_q0 = lv["ones_before"] + int(bits[: q0 - first_bit].sum())
            mine = bits[q0 - first_bit: q0 - first_bit + len(slots)]
            zeros_before_q0 = q0 - ones_before_q0
            own = np.flatnonzero(mine == 0)
            idx = zeros_before_q0 + np.arange(len(own)) - lv["psa_first"]
            assert len(own) == 0 or (idx.min() >= 0 and idx.max() < lv["psa_count"]), (h, idx.min() if len(idx) else 0, lv)
            out[slots[own]] = lv["beg"] + psa[idx]
            slots = slots[mine == 1]
            q0 = ones_before_q0
        sa5 = np.zeros((x1 - x0, 5), np.uint8)
        for k in range(5):
            sa5[:, k] = (out >> (8 * k)) & 255
        return sa5.reshape(-1)
