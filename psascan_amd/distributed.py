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


def assemble_gt(parts_bits, cuts, te):
    """numpy helper (tests / small sizes): per-rank bit arrays (u = cuts[r+1] - j) -> one array with u = te - j."""
    T = te - cuts[0]
    out = np.zeros(T, np.uint8)
    for r, bits in enumerate(parts_bits):
        tb_r, te_r = cuts[r], cuts[r + 1]
        out[te - te_r: te - tb_r] = bits[: te_r - tb_r]
    return out
