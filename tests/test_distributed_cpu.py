"""N > 1 path on CPU: world_size-2 (and 3) gloo runs of psascan_amd.distributed.sharded_pass with the
oracle standing in for the kernels (test-only injection), checking the sharding logic: range cuts,
right contexts, the gap all-reduce, placement of the gathered gt bits and the output partition."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import orc
    from psascan_amd import distributed as D
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    rng = np.random.default_rng(5)
    for kind in ("sig4", "alla"):
        n = 6000
        t = rng.integers(0, 4, n, dtype=np.uint8) if kind == "sig4" else np.full(n, 97, np.uint8)
        b, mid, e = 500, 2500, 5777                      # block [b,mid), tail [mid,e): pass A shape
        sa = orc.suffix_array(t); isa = orc.inverse(sa)
        psa, bwt, i0, _ = orc.partial_sa(t, sa, isa, b, mid)
        m = mid - b
        rk = orc.Rank(bwt)
        def gt_bits(hi, lo):                              # gt wrt mid of positions (lo, hi], u = hi - j
            return orc.packbits([(isa[hi - u] if hi - u < n else -1) > isa[mid] for u in range(hi - lo)] + [0] * 64)
        def rank_at(p):
            return int((isa[b:mid] < (isa[p] if p < n else -1)).sum())

        class Ops:
            device = "cpu"
            def new_i32(self, k): return torch.zeros(int(k), dtype=torch.int32)
        ops = Ops()
        gap_t = torch.zeros(m + 2, dtype=torch.int32)
        T = e - mid
        words = (T // world + 64) // 32 + 4
        def stream_fn(tb_r, te_r, ctx):
            # the oracle is handed the exact start rank; the HIP path finds it inside the context
            assert ctx % 64 == 0 and te_r + ctx <= e
            g, gto, fin = orc.stream_pass(rk, i0, int(t[mid - 1]), t, tb_r, te_r, gt_bits(te_r, tb_r), rank_at(te_r))
            gap_t[: m + 1] += torch.from_numpy(g.astype(np.int32))
            out = torch.zeros(words, dtype=torch.int32)
            raw = np.zeros(words * 4, np.uint8); raw[: len(gto)] = gto[: words * 4]
            out[:] = torch.from_numpy(raw.view(np.int32))
            return out
        cuts, parts = D.sharded_pass(dist, ops, world, rank, mid, e, stream_fn, gap_t, words)
        assert cuts[0] == mid and cuts[-1] == e and all(cuts[k] <= cuts[k + 1] for k in range(world))
        want_gap, want_gt, _ = orc.stream_pass(rk, i0, int(t[mid - 1]), t, mid, e, gt_bits(e, mid), rank_at(e))
        assert np.array_equal(gap_t[: m + 1].numpy().astype(np.uint64), want_gap), kind
        bits = [orc.bits(p.numpy().view(np.uint8), cuts[r + 1] - cuts[r]) for r, p in enumerate(parts)]
        got = D.assemble_gt(bits, cuts, e)
        assert np.array_equal(got, orc.bits(want_gt, T)), kind
        oc = D.output_cuts(n, world)
        assert oc[0] == 0 and oc[-1] == n and all(oc[k] <= oc[k + 1] for k in range(world))
    dist.barrier()
    dist.destroy_process_group()
    print("WORKER_OK", rank)
""")


WORKER_A2A = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import orc
    from psascan_amd import distributed as D
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    rng = np.random.default_rng(9)
    n = 400000
    t = rng.integers(0, 4, n, dtype=np.uint8)
    b, mid, e = 1000, 150000, 390001                  # block [b,mid), tail [mid,e)
    sa = orc.suffix_array(t); isa = orc.inverse(sa)
    psa, bwt, i0, _ = orc.partial_sa(t, sa, isa, b, mid)
    m = mid - b
    rk = orc.Rank(bwt)
    def gt_bits(hi, lo):
        return orc.packbits((np.concatenate([isa, [-1]])[np.arange(hi, lo, -1)] > isa[mid]).astype(np.uint8).tolist() + [0] * 64)
    def rank_at(p):
        return int((isa[b:mid] < (isa[p] if p < n else -1)).sum())
    SH, PB = 14, 512                                   # window bits / level-1 bins of the HIP partition

    class Ops:                                         # numpy stand-ins with the semantics of the C ABI entry points
        def new_i32(self, k): return torch.zeros(int(k), dtype=torch.int32)
        def i64_from(self, v): return torch.tensor(v, dtype=torch.int64)
        def partition(self, log, nlog, m, world):
            v = log.numpy().view(np.uint32)[:nlog]; v = v[v != 0xFFFFFFFF]
            nwin = (m + 1 + (1 << SH) - 1) >> SH
            bits = int(np.ceil(np.log2(nwin))) if nwin > 1 else 0
            shift1 = SH + max(0, bits - 9)
            nb = ((m + 1) + (1 << shift1) - 1) >> shift1
            vb = [((nb * p // world) << shift1) for p in range(world)] + [max((nb << shift1), m + 1)]
            order = np.argsort(v >> shift1, kind="stable")
            v = v[order]
            offs = [int(np.searchsorted(v, vb[p])) for p in range(world)] + [len(v)]
            out = torch.zeros(max(len(v), 1), dtype=torch.int32); out[: len(v)] = torch.from_numpy(v.view(np.int32))
            return out, offs, vb
        def hist_slice(self, recv_t, nrecv, base, count):
            v = recv_t.numpy().view(np.uint32)[:nrecv].astype(np.int64) - base
            assert ((v >= 0) & (v < max(count, 1))).all()
            return torch.from_numpy(np.bincount(v, minlength=max(count, 1)).astype(np.int32))
        def slice_to_bits(self, gap_slice, j0, count, m, ps_before, nbits):
            g = gap_slice.numpy()[:count].astype(np.int64)
            pos = j0 + np.arange(count) + ps_before + np.cumsum(g)
            pos = pos[(j0 + np.arange(count)) < m]
            bits = np.zeros(((nbits + 31) // 32 + 2) * 32, np.uint8); bits[pos] = 1
            return torch.from_numpy(np.packbits(bits, bitorder="little").view(np.int32).copy())
        def bits_not(self, bits, nbits):
            x = np.unpackbits(bits.numpy().view(np.uint8), bitorder="little"); x[:nbits] ^= 1; x[nbits:] = 0
            bits[:] = torch.from_numpy(np.packbits(x, bitorder="little").view(np.int32).copy())
    ops = Ops()
    T = e - mid
    words = (T // world + 64) // 32 + 4
    def stream_log_fn(tb_r, te_r, ctx):
        g, gto, fin = orc.stream_pass(rk, i0, int(t[mid - 1]), t, tb_r, te_r, gt_bits(te_r, tb_r), rank_at(te_r))
        log = np.repeat(np.arange(m + 1, dtype=np.uint32), g.astype(np.int64))     # the multiset of ranks = the log
        log = np.concatenate([log, np.full(37, 0xFFFFFFFF, np.uint32)]); rng.shuffle(log)
        out = torch.zeros(words, dtype=torch.int32)
        raw = np.zeros(words * 4, np.uint8); raw[: len(gto)] = gto[: words * 4]
        out[:] = torch.from_numpy(raw.view(np.int32))
        return torch.from_numpy(log.view(np.int32).copy()), len(log), out
    res = D.a2a_pass(dist, ops, world, rank, m, mid, e, stream_log_fn, words)
    want_gap, want_gt, _ = orc.stream_pass(rk, i0, int(t[mid - 1]), t, mid, e, gt_bits(e, mid), rank_at(e))
    base, count = res["base"], res["count"]
    assert np.array_equal(res["gap_slice"].numpy()[:count].astype(np.uint64), want_gap[base: base + count])
    assert res["streamed"] == T and res["nbits"] == m + T
    want_bv, nb = orc.gap_to_bitvector(want_gap, m)
    assert nb == res["nbits"]
    assert np.array_equal(orc.bits(res["bits"].numpy().view(np.uint8), nb), orc.bits(want_bv, nb))
    bits = [orc.bits(p.numpy().view(np.uint8), res["cuts"][r + 1] - res["cuts"][r]) for r, p in enumerate(res["gt_parts"])]
    assert np.array_equal(D.assemble_gt(bits, res["cuts"], e), orc.bits(want_gt, T))
    assert sum(res["send"]) == res["cuts"][rank + 1] - res["cuts"][rank]
    dist.barrier()
    dist.destroy_process_group()
    print("WORKER_OK", rank)
""")


WORKER_BLOCKS = textwrap.dedent("""
    import os, sys, hashlib, json
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import orc
    from golden import inputs as gin
    from psascan_amd import blockdist as BD
    from blockdist_oracle_ops import OracleBlockOps
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    GOLD = json.load(open(os.path.join({root!r}, "tests", "golden", "golden.json")))
    for name in ("sig4_with_zero", "alla", "per3", "fib", "rand1m"):
        t = gin.GENERATORS[name]()
        n = len(t)
        # (the short texts also run the merge in three sub-ranges per rank with the high plane of 40-bit partial SAs on the
        #  wire: configs[3]'s form of the exchange)
        # ... and with helper ranks (blockdist.helper_of): the high ranks stream half of the low ranks' chunks into gap arrays
        # of their own, which go home at the end
        ops = OracleBlockOps(t, merge_rounds=1 if name == "rand1m" else 3, force_wide=name != "rand1m", helpers=name in ("sig4_with_zero", "fib", "rand1m"))
        x0, x1, sa5 = BD.run(dist, ops, world, rank, n)
        assert len(sa5) == 5 * (x1 - x0)
        # every rank contributes its output range; rank 0 assembles the file and compares with the reference's hash
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([len(sa5)], dtype=torch.int64))
        mx = max(int(s) for s in sizes)
        buf = torch.zeros(mx, dtype=torch.uint8); buf[: len(sa5)] = torch.from_numpy(sa5)
        parts = [torch.zeros(mx, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(parts, buf)
        if rank == 0:
            whole = np.concatenate([p.numpy()[: int(s)] for p, s in zip(parts, sizes)])
            assert len(whole) == 5 * n
            assert hashlib.sha256(whole.tobytes()).hexdigest() == GOLD[name]["sa5_sha256"], name
    dist.barrier()
    dist.destroy_process_group()
    print("WORKER_OK", rank)
""")


@pytest.mark.parametrize("world,exchange", [(1, "p2p"), (2, "p2p"), (3, "p2p"), (4, "p2p"), (5, "p2p"), (4, "allgather"), (5, "allgather")])
def test_block_per_gpu_schedule_gloo(tmp_path, world, exchange):
    """north_star's multi-GPU split (psascan_amd/blockdist.py): one block per rank, one exchange of gt slices per round
    (point to point: a slice goes to the left neighbour and, when that one is helped, to its helper; or the older
    all-gather), near-to-far chunks with searched start ranks, output-range partitioned merge over slices -- with the
    oracle standing in for the kernels, the assembled output must hash to the reference's .sa5 for the five seeded
    inputs of tests/golden/golden.json (random bytes, periodic texts, a 4-letter text with zero bytes)."""
    _run_workers(tmp_path, WORKER_BLOCKS, world, {"PSASCAN_GT_EXCHANGE": exchange})


def _run_workers(tmp_path, src, world, extra_env=None):
    script = tmp_path / "worker.py"
    script.write_text(src.format(root=ROOT))
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_RANK=str(r), OMP_NUM_THREADS="1", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for r, p in enumerate(procs):
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        assert p.returncode == 0 and f"WORKER_OK {r}" in o, o[-3000:]


@pytest.mark.parametrize("world", [1, 2, 4])
def test_a2a_pass_gloo(tmp_path, world):
    """gap array sharded by index range: rank-log all-to-all + slice histograms + bit all-reduce."""
    _run_workers(tmp_path, WORKER_A2A, world)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_pass_gloo(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_RANK=str(r), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"WORKER_OK {r}" in o, o[-3000:]


def test_cuts_properties():
    from psascan_amd import distributed as D
    for (tb, te, w) in [(0, 1000, 2), (5, 70, 4), (0, 63, 8), (100, 100, 3), (0, 1 << 31, 8), (7, 4097, 5)]:
        c = D.tail_cuts(tb, te, w)
        assert c[0] == tb and c[-1] == te and len(c) == w + 1
        assert all(c[k] <= c[k + 1] for k in range(w))
        assert all((te - c[k]) % 64 == 0 for k in range(1, w) if c[k] not in (tb, te))
        for k in range(w):
            ctx = D.context_len(c[k + 1], te)
            assert ctx % 64 == 0 and c[k + 1] + ctx <= te
    for n, w in [(5, 2), (4096, 2), (100000, 8), (1 << 32, 8)]:
        oc = D.output_cuts(n, w)
        assert oc[0] == 0 and oc[-1] == n and all(oc[k] <= oc[k + 1] for k in range(w))
        assert all(x % 4096 == 0 or x == n for x in oc[:-1])
