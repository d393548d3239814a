"""Tail-sharded streaming pass + range-partitioned merge over N ranks (one process per GPU,
torch.distributed; backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests).

The reference parallelises one pass by cutting the tail into ranges that all update one gap
array (compute_gap.hpp:68-69,114-124; update.hpp:86-96).  Same axis here:
    rank r streams tail range [cuts[r], cuts[r+1])           -> local gap array, local gt bits
    ONE all-reduce(sum) of the gap arrays                     (the exchange step of the pass)
    ONE all-gather of the gt bits each range produced         (north_star's per-round gt exchange)
    every rank converts the summed gap array and merges 1/N of the output range.

`ops` supplies the compute (HipOps = the C ABI on this rank's GPU; the CPU tests inject an
oracle-backed object to check the sharding logic without a GPU).  Tensors handed to the
collectives are torch tensors on ops.device.
"""
import numpy as np


def tail_cuts(tb, te, world):
    """Range boundaries, 64-aligned relative to te so gt words never straddle two ranks."""
    T = te - tb
    # counted from the tail END (u = te - j): rank 0 takes the leftmost positions
    cuts = [te - ((T * (world - r) // world + 63) // 64) * 64 for r in range(world)]
    cuts[0] = tb
    for r in range(1, world):
        cuts[r] = min(max(cuts[r], cuts[r - 1]), te)
    return cuts + [te]


def output_cuts(n, world):
    return [min(n, (n * r // world + 4095) // 4096 * 4096) for r in range(world)] + [n]


def context_len(te_r, te, max_ctx=1 << 16):
    """Right context of a range: up to max_ctx positions right of its end, multiple of 64, inside the tail."""
    return min(te - te_r, max_ctx) // 64 * 64 if te - te_r >= 64 else 0


class HipOps:
    """Compute on this rank's GPU through the C ABI; buffers that take part in collectives are torch tensors."""

    def __init__(self, torch, api, device):
        self.torch, self.api, self.device = torch, api, device

    def new_i32(self, nwords):
        return self.torch.zeros(int(nwords), dtype=self.torch.int32, device=self.device)

    def ptr(self, t):
        return t.data_ptr()

    def stream_range(self, rank_struct, i0, last, d_text, tb, te_r, ctx, gt_in_ptr, start_rank, gap_t, gt_out_t, max_chains=0):
        fin, st = self.api.stream_gap(rank_struct, i0, last, d_text.at(tb), te_r - tb, gt_in_ptr, start_rank, self.ptr(gap_t),
                                      self.ptr(gt_out_t), max_chains, right_context=ctx)
        return fin, st


def sharded_pass(dist, ops, world, rank, tb, te, stream_fn, gap_t, gt_words_per_rank):
    """Runs stream_fn(tb_r, te_r, ctx) -> gt_out tensor on this rank's range, then the two collectives.
    Returns (cuts, list of gathered gt tensors (one per rank; bit u <-> position cuts[r+1]-u))."""
    cuts = tail_cuts(tb, te, world)
    tb_r, te_r = cuts[rank], cuts[rank + 1]
    ctx = context_len(te_r, te)
    gt_mine = stream_fn(tb_r, te_r, ctx)
    assert gt_mine.numel() == gt_words_per_rank
    if world > 1:
        dist.all_reduce(gap_t)
        parts = [ops.new_i32(gt_words_per_rank) for _ in range(world)]
        dist.all_gather(parts, gt_mine)
    else:
        parts = [gt_mine]
    return cuts, parts


def a2a_pass(dist, ops, world, rank, m, tb, te, stream_log_fn, gt_words_per_rank):
    """Tail-sharded pass whose gap array is sharded too (no full-size all-reduce):
        stream own tail range -> rank log                     (psg_stream_gap_log)
        split the log by owner of the gap slice               (psg_log_partition)
        all-to-all the parts                                  (4 B per streamed suffix, once)
        count the received entries into the own gap slice     (psg_gap_hist)
        mark the slice's zero bits in a global-size array     (psg_gap_slice_to_bits)
        all-reduce(sum) of the bit arrays (disjoint bits), invert, all-gather the gt bits.
    Returns a dict with the slice geometry, the full bitvector (every rank) and the gt parts."""
    sync = getattr(ops, "sync", lambda: None)   # device-wide sync around collectives (needed for gloo on GPU tensors)
    cuts = tail_cuts(tb, te, world)
    tb_r, te_r = cuts[rank], cuts[rank + 1]
    ctx = context_len(te_r, te)
    log, nlog, gt_mine = stream_log_fn(tb_r, te_r, ctx)
    part, offs, vb = ops.partition(log, nlog, m, world)
    send = [int(offs[d + 1] - offs[d]) for d in range(world)]
    if world > 1:
        sc = ops.i64_from(send)
        rc = ops.i64_from([0] * world)
        sync()
        dist.all_to_all_single(rc, sc)
        recv = [int(x) for x in rc.tolist()]
        nrecv = sum(recv)
        recv_t = ops.new_i32(max(nrecv, 1))
        sync()
        dist.all_to_all_single(recv_t[:nrecv], part[: offs[world]], recv, send)
        sync()
    else:
        recv, nrecv, recv_t = send, sum(send), part
    base = int(vb[rank])
    count = max(0, min(int(vb[rank + 1]), m + 1) - base)
    gap_slice = ops.hist_slice(recv_t, nrecv, base, count)
    if world > 1:
        tot = ops.i64_from([nrecv])
        tots = [ops.i64_from([0]) for _ in range(world)]
        dist.all_gather(tots, tot)
        totals = [int(t.tolist()[0]) for t in tots]
    else:
        totals = [nrecv]
    ps_before = sum(totals[:rank])
    T_all = sum(totals)
    nbits = m + T_all
    bits = ops.slice_to_bits(gap_slice, base, count, m, ps_before, nbits)
    if world > 1:
        sync()
        dist.all_reduce(bits)
        sync()
    ops.bits_not(bits, nbits)
    if world > 1:
        gt_parts = [ops.new_i32(gt_words_per_rank) for _ in range(world)]
        sync()
        dist.all_gather(gt_parts, gt_mine)
        sync()
    else:
        gt_parts = [gt_mine]
    return {"cuts": cuts, "value_bounds": vb, "base": base, "count": count, "gap_slice": gap_slice, "bits": bits, "nbits": nbits,
            "gt_parts": gt_parts, "streamed": T_all, "send": send, "recv": recv}


class HipA2AOps:
    """ops for a2a_pass on this rank's GPU (C ABI); torch tensors for everything the collectives touch."""

    def __init__(self, torch, api, device, full_sync=False):
        self.torch, self.api, self.device = torch, api, device
        self.full_sync = full_sync     # True for the gloo rehearsal: gloo does not order itself with our stream

    def new_i32(self, n):
        return self.torch.zeros(int(n), dtype=self.torch.int32, device=self.device)

    def i64_from(self, vals):
        return self.torch.tensor(vals, dtype=self.torch.int64, device=self.device)

    def partition(self, log, nlog, m, world):
        out = self.torch.empty(max(int(nlog), 1), dtype=self.torch.int32, device=self.device)
        offs, vb = self.api.log_partition(log.data_ptr() if hasattr(log, "data_ptr") else log, nlog, m, world, out.data_ptr())
        return out, offs, vb

    def hist_slice(self, recv_t, nrecv, base, count):
        g = self.new_i32(self.api.gap_words(max(count, 1) - 1))   # a gap array: counters + in-band excess area
        self.api.gap_hist(recv_t.data_ptr(), nrecv, base, count, g.data_ptr())
        return g

    def slice_to_bits(self, gap_slice, j0, count, m, ps_before, nbits):
        bits = self.new_i32((nbits + 31) // 32 + 2)
        self.api.gap_slice_to_bits(gap_slice.data_ptr(), j0, count, m, ps_before, bits.data_ptr())
        return bits

    def bits_not(self, bits, nbits):
        self.api.bits_not(bits.data_ptr(), nbits)

    def sync(self):
        if self.full_sync:
            self.torch.cuda.synchronize()


def assemble_gt(parts_bits, cuts, te):
    """numpy helper (tests / small sizes): per-rank bit arrays (u = cuts[r+1] - j) -> one array with u = te - j."""
    T = te - cuts[0]
    out = np.zeros(T, np.uint8)
    for r, bits in enumerate(parts_bits):
        tb_r, te_r = cuts[r], cuts[r + 1]
        out[te - te_r: te - tb_r] = bits[: te_r - tb_r]
    return out
