"""Full block schedule at sizes the oracle cannot reach (GPU only): multi-block runs with pass A,
BWT merge, pass B (rank-log mode kicks in above 4 Mi streamed suffixes), gap split and the
multi-level merge -- checked through size-independent properties of the output: it is a permutation
(sum of entries) and a large random sample of adjacent entries is in suffix order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_device_sorter_english_like_vs_oracle(gpu_lib):
    """the bench's input preparation on configs[2]-like text: device generator (mode 3) + prefix-key sorter with
    refinement rounds, against the oracle's partial SA / BWT / gt bits of the same half-block"""
    import orc
    from psascan_amd import api, extras
    n = 300_000
    d_text = extras.gen_text(n, extras.MODE_ENGLISH, 0, seed=5)
    t = api.download(d_text, np.uint8, n)
    assert set(np.unique(t)) <= set(b"etaoinshrdlcumwfgypbvkjxqz .") and len(np.unique(t)) >= 20
    sa = orc.suffix_array(t)
    isa = orc.inverse(sa)
    for b, e in [(0, n), (n // 3, 2 * n // 3), (n // 2, n)]:
        r = extras.sort_halfblock(d_text, n, b, e)
        psa, bwt, i0, gt = orc.partial_sa(t, sa, isa, b, e)
        m = e - b
        assert np.array_equal(api.download(r["psa_lo"], np.uint32, m).astype(np.int64), psa), (b, e)
        assert np.array_equal(api.download(r["bwt"], np.uint8, m), bwt) and r["i0"] == i0
        assert np.array_equal(orc.bits(api.download(r["gt_begin"], np.uint8, (m + 7) // 8), m), orc.bits(gt, m))


@pytest.mark.parametrize("mode,sigma,mib,block_mib", [(0, 0, 192, 48), (1, 0, 160, 64), (2, 12, 96, 20), (3, 0, 128, 40)])
def test_multiblock_properties(gpu_lib, mode, sigma, mib, block_mib):
    from psascan_amd import api, extras, pipeline
    n = (mib << 20) + 12345
    d_text = extras.gen_text(n, mode, sigma, seed=77 + mode)
    text = api.download(d_text, np.uint8, n)
    sorter = extras.DeviceSorter(d_text, n)
    stats = []
    d_out = pipeline.construct_sa5(text, block_mib << 20, 1 << 40, sorter, stats=stats, d_text=d_text, return_device=True)
    bad, s = extras.check_sa5(d_text, n, d_out, n, samples=1 << 20, seed=5)
    assert bad == 0
    assert s == (n * (n - 1) // 2) % (1 << 64)
    passes = [p[0] for p in stats]
    assert "A" in passes and "B" in passes
    assert any(p[3].hist_ms > 0 for p in stats)          # the atomics-free path was exercised
    # first and last entries: smallest / largest suffix by direct comparison on a sample
    sa_head = np.frombuffer(api.download(d_out, np.uint8, 50).tobytes(), np.uint8).reshape(-1, 5)
    first = int(sum(int(sa_head[0, b]) << (8 * b) for b in range(5)))
    assert text[first] == text.min()


def test_block_of_more_than_2_pow_32_symbols(gpu_lib):
    """A real block of >= 2^32 symbols on the fast path (reference: psascan.hpp:117-125 has no block size limit,
    rank.hpp:566-568): 6 GiB of random bytes, block 0 = [0, 4.25 GiB) in two halves of 2.125 GiB, so that pass B
    streams a 1.75 Gi tail through a rank structure over m = 4.25 Gi symbols (three superblocks of 2^31
    positions, 40-bit rank log in two planes, three slabs of gap counters).  Checked by the library's own
    invariants on the way (sum(gap) == tail length, chain hand-over ranks) and the .sa5 properties."""
    from psascan_amd import api, extras, pipeline
    n = (6 << 30) + 12345
    blk = (17 << 28)                                   # 4.25 GiB
    d_text = extras.gen_text(n, 0, 0, seed=41)
    text = api.download(d_text, np.uint8, n)
    sorter = extras.DeviceSorter(d_text, n)
    stats = []
    d_out = pipeline.construct_sa5(text, blk, 1 << 40, sorter, stats=stats, d_text=d_text, return_device=True)
    bad, s = extras.check_sa5(d_text, n, d_out, n, samples=1 << 20, seed=5)
    assert bad == 0
    assert s == (n * (n - 1) // 2) % (1 << 64)
    pb = [p for p in stats if p[0] == "B"]
    assert len(pb) == 1 and pb[0][2] - pb[0][1] == blk and pb[0][3].hist_ms > 0
    print("pass B over a 4.25 Gi block:", pb[0][3].total_ms, "ms, kernel", pb[0][3].kernel_ms, "hist", pb[0][3].hist_ms)


def english_like(n, seed=1, nwords=4096):
    """Zipfian words over a skewed 26-letter alphabet: repeats of tens of bytes, sigma = 28."""
    rng = np.random.default_rng(seed)
    letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", np.uint8)
    lp = 1.0 / np.arange(1, 27) ** 0.9
    lp /= lp.sum()
    words = [bytes(rng.choice(letters, l, p=lp)) for l in rng.integers(2, 10, nwords)]
    wp = 1.0 / np.arange(1, nwords + 1)
    wp /= wp.sum()
    out = bytearray()
    while len(out) < n:
        out += b" ".join(words[i] for i in rng.choice(nwords, 200000, p=wp)) + b". "
    return np.frombuffer(bytes(out[:n]), np.uint8).copy()


def test_english_like_with_host_sorter(gpu_lib):
    """configs[2]-shaped input at test scale: skewed alphabet (frequent symbols in BITMAP mode, rare in
    LIST mode, dense buckets in the overflow pool), longer repeats, multi-block, host SA-IS sorter."""
    from psascan_amd import api, extras, pipeline
    from psascan_amd.hostsort import HostSorter
    n = 40 << 20
    text = english_like(n, seed=3)
    d_text = api.upload(text, pad_to=16)
    stats = []
    d_out = pipeline.construct_sa5(text, 10 << 20, 1 << 40, HostSorter(), stats=stats, d_text=d_text, return_device=True)
    bad, s = extras.check_sa5(d_text, n, d_out, n, samples=1 << 20, seed=9)
    assert bad == 0 and s == (n * (n - 1) // 2) % (1 << 64)
    assert sum(p[3].unresolved for p in stats) >= 0
    print("english-like passes:", [(p[0], p[3].n_chains, p[3].warmup_steps, p[3].unresolved, p[3].rounds) for p in stats])


WORKER_BLOCKS_GPU = """
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import orc
import psascan_amd
from psascan_amd import api, extras, blockdist as BD
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
psascan_amd.lib(0)                                   # every rank on the one GPU of this box (rehearsal)
for case, (mode, n) in enumerate(((extras.MODE_BYTES255, 3_000_017), (extras.MODE_ENGLISH, 2_500_000), (extras.MODE_DNA, 1_999_999))):
    d_text = extras.gen_text(n, mode, 0, seed=21)
    if case == 0:
        ops = BD.HipBlockOps(torch, api, d_text, n, extras.DeviceSorter(d_text, n), comm="cpu", max_chains=4096)
    else:
        # BASELINE configs[3]'s form of the schedule at a small size: no rank holds the whole text (own block + look-ahead
        # resident, the chunk of a round loaded into rotating buffers, far start ranks searched through two text windows),
        # partial SAs carry a high plane (forced: all zero here), the merge runs in three sub-ranges per rank; the DNA case
        # also forces the 40-bit rank log and many superblocks in the rank structure
        host_text = api.download(d_text, np.uint8, n)
        bounds = BD.block_bounds(n, world)
        b, e = bounds[rank], bounds[rank + 1]
        src = BD.ChunkedText(lambda lo, hi: api.upload(host_text[lo:hi], pad_to=64), n, b, min(n, e + BD.HipBlockOps.LOOKAHEAD))
        base = extras.DeviceSorter(d_text, n)
        def sorter(text, hb, he, gt_tail, base=base):
            r = base(text, hb, he, gt_tail)
            r["psa_hi"] = api.zeros(he - hb + 16)
            return r
        if case == 2:
            os.environ["PSG_LOG_WIDE"] = "1"; os.environ["PSG_SM_SB_SHIFT"] = "14"
        ops = BD.HipBlockOps(torch, api, src, n, sorter, comm="cpu", max_chains=4096, merge_rounds=3, force_wide=True, helpers=True)   # (+ helper ranks from world = 3 on)
    stats = []
    x0, x1, sa5 = BD.run(dist, ops, world, rank, n, stats)
    os.environ.pop("PSG_LOG_WIDE", None); os.environ.pop("PSG_SM_SB_SHIFT", None)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([len(sa5)], dtype=torch.int64))
    mx = max(int(s) for s in sizes)
    buf = torch.zeros(mx, dtype=torch.uint8); buf[: len(sa5)] = torch.from_numpy(sa5)
    parts = [torch.zeros(mx, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(parts, buf)
    if rank == 0:
        whole = np.concatenate([p.numpy()[: int(s)] for p, s in zip(parts, sizes)])
        t = api.download(d_text, np.uint8, n)
        assert np.array_equal(orc.sa5_to_sa(whole), orc.suffix_array(t)), mode
    assert len(stats) >= world - 1 - rank                 # (helper ranks stream for their partners as well)
dist.barrier()
dist.destroy_process_group()
print("WORKER_OK", rank)
"""


@pytest.mark.parametrize("world", [2, 3, 4])
def test_block_per_gpu_schedule_real_kernels(gpu_lib, tmp_path, world):
    """psascan_amd.blockdist with the real kernels: `world` processes share this box's one GPU and talk over gloo
    (the same schedule runs over RCCL on a multi-GPU node).  Three texts: the plain form (whole text resident), and twice
    the form BASELINE configs[3] needs -- chunked text with two-window start-rank search, 40-bit partial SAs in two planes
    through the exchange, merge in sub-ranges, wide rank log.  Output compared with the oracle's suffix array."""
    import os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(WORKER_BLOCKS_GPU.format(root=root))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0", OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for r, p in enumerate(procs):
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        assert p.returncode == 0 and f"WORKER_OK {r}" in o, o[-3000:]


def test_configs1_full_size_properties(gpu_lib):
    """BASELINE configs[1] at full size, as a test: 4 GiB of uniform bytes, one block of two 2 GiB half-blocks --
    rank build (LIST8 symbol-major), pass A over 2^31 suffixes (rank log + two-level partition + window histograms,
    fresh gap array), gap -> bitvector, two-way merge to 4 Gi uint40 entries.  Size-independent properties: the
    library's own invariants (sum(gap) == tail length, chain hand-over ranks), the output is a permutation and 2^20
    sampled adjacent entries are in suffix order."""
    from psascan_amd import api, extras
    n = 4 << 30
    mid = n // 2
    d_text = extras.gen_text(n, extras.MODE_BYTES255, 0, seed=2)
    Rh = extras.sort_halfblock(d_text, n, mid, n)
    Lh = extras.sort_halfblock(d_text, n, 0, mid)
    last_left = int(api.download(d_text, np.uint8, 1, mid - 1)[0])
    rk = api.rank_build(Lh["bwt"], mid)
    gap = api.gap_array(mid, fill=None)
    gt_out = api.zeros(4 * ((n - mid + 31) // 32 + 4))
    fin, st = api.stream_gap(rk, Lh["i0"], last_left, d_text.at(mid), n - mid, Rh["gt_begin"], 0, gap, gt_out, 0, fresh_gap=True)
    assert st.hist_ms > 0 and st.rounds == 1 and st.unresolved == 0
    rk.free()
    mbv = api.zeros(4 * ((n + 31) // 32 + 2))
    assert api.gap_to_bitvector(gap, mid, mbv, n) == n
    assert api.popcount(mbv, n) == n - mid
    Lh["mbv"] = mbv
    d_out = api.merge_half_blocks([Lh, Rh])
    bad, s = extras.check_sa5(d_text, n, d_out, n, samples=1 << 20, seed=3)
    assert bad == 0 and s == (n * (n - 1) // 2) % (1 << 64)


def test_bench_configs2_shape_reduced(gpu_lib):
    """bench.py's configs[2] step at a reduced size (4 GiB English-like text, 8 blocks = 16 half-blocks, partial SAs
    in pinned host memory, streamed merge): the step verifies its own output on the device and raises otherwise."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gib", "4", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary",
                        "--with-output-d2h", "--psa-hbm-gib", "6"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["config"]["blocks"] == 8 and d["config"]["half_blocks"] == 16 and d["value"] > 0
    # 6 GiB of the 16 GiB of partial SAs are resident in HBM (5 half-blocks of 1 GiB + margin), the rest streams in
    assert 0 < d["pcie"]["h2d_bytes_per_step"] < 4 * (4 << 30) and d["with_output_d2h"]["entries_received_on_host"] == 4 << 30
    assert d["roofline"]["frac"] > 0.05 and d["streamed_suffixes_per_step"] > 15 * (1 << 30)


def test_bench_configs2_full_size(gpu_lib):
    """BASELINE configs[2] at its full size -- 32 GiB of English-like text, 8 blocks of 4 GiB, 137 G suffixes streamed --
    once through bench.py's step (about a minute with the device-side preparation): every output slice passes the
    property check on the device (permutation sum, sampled adjacent pairs against the text) or the step raises."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import gc
    gc.collect()
    gpu_lib.psg_trim()                       # the child needs ~250 GiB of the device: give back what this process has cached
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary", "--no-output-d2h"],
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["config"]["text_bytes"] == 32 << 30 and d["config"]["blocks"] == 8 and d["config"]["half_blocks"] == 16
    assert d["streamed_suffixes_per_step"] > 120 * (1 << 30) and d["value"] > 0 and "every_step" in d["property_check"]


def test_halfblocks_of_more_than_2_pow_32_symbols(gpu_lib, tmp_path):
    """BASELINE configs[3]'s shape in small: a 9 GiB block of DNA, two half-blocks of 4.5 GiB = more than 2^32 suffixes
    each.  construct_sa --device-sort sorts each in three pieces, merges them with the hot path into a partial SA of
    40-bit values in two planes (psg_merge_run_planes, psg_halfblock_from_psa40), hands it to host memory in the
    background (psg_d2h_begin), runs pass A with a 4.5 Gi-symbol rank structure and merges 9 Gi entries; the output is
    verified on the device (permutation sum, sampled adjacent pairs in suffix order)."""
    import os, subprocess
    from psascan_amd import api, extras
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = 9 << 30
    f = tmp_path / "dna.bin"
    with open(f, "wb") as fh:
        for k in range(9):
            d_t = extras.gen_text(1 << 30, extras.MODE_DNA, 0, seed=70 + k)
            api.download(d_t, np.uint8, 1 << 30).tofile(fh)
            d_t.free()
    api.lib().psg_trim()
    try:
        r = subprocess.run([os.path.join(root, "host", "construct_sa"), "-m", str(5 * n), "--block-size", str(n), "--device-sort", "--check=1024", "--discard-output", "-v", str(f)],
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="16"))
    finally:
        os.remove(f)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stderr.count("3 pieces sorted on the device") == 2 and "permutation sum ok, 0 of" in r.stderr


def test_all_modes_write_the_same_sa5(gpu_lib):
    """256 MiB of English-like text in 6 blocks through construct_sa in seven modes -- default (host leaves merged on the
    device in batches), --text-on-host (tails in chunks, leaves through a text window), --device-sort, --spill-psa +
    --checkpoint, --no-device-merge (half-blocks sorted whole on host threads), --hbm-limit (a device budget of ~60 bytes per
    block symbol: text, gt bits, partial SAs and merge bitvectors in host memory), and that budget with --spill-psa (partial
    SAs and merge bitvectors in files, mapped for the merge): the .sa5 files are byte-identical (the
    first of them is what the reference's hashes pin at smaller sizes, tests/test_host.py)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "compare_modes.py"), "256", "english"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "IDENTICAL" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


@pytest.mark.parametrize("pieces", [False, True])
def test_bench_block_per_gpu_rehearsal(gpu_lib, pieces):
    """bench.py --gpus 3 as the driver launches it (torch.distributed.run, one process per rank), rehearsed on this box's one
    GPU over gloo: BASELINE configs[3]'s form of the schedule -- DNA, chunked text from the seeded generator, start ranks
    searched through two text windows, partial SAs in pinned host memory, merge in sub-ranges -- at 48 MiB per block.  The
    property check of the line (sampled adjacent pairs in suffix order + permutation sum over every rank's output) must hold.
    pieces: every half-block is sorted in pieces merged with the hot path and carries a high plane (the 2^33-symbol
    half-blocks of 16 GiB blocks, forced at this size)."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, PSASCAN_DIST_BACKEND="gloo", PSASCAN_SHARE_GPU="1", PSASCAN_MERGE_ROUNDS="3", OMP_NUM_THREADS="2")
    if pieces:
        env["PSASCAN_TEST_PIECE_MAX"] = str(9 << 20)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0", "--block-gib", "0.046875"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 3 and d["config"]["blocks"] == 3 and "configs[3] shape" in d["config"]["workload"]
    assert d["property_check"] == {"sampled_adjacent_pairs_out_of_order": 0, "sum_matches_permutation": True}, d["property_check"]
    assert ("40-bit" in d["config"]["workload"]) == pieces


def test_python_schedule_and_construct_sa_run_the_same_passes(gpu_lib, tmp_path):
    """Two orchestrations of ONE schedule: bench.py times psascan_amd/pipeline.py (Python over the C ABI), the drop-in is
    host/construct_sa (C++ over the same ABI).  On the same text, block size and sorter placement (the device sorter on
    both sides) they must run the same streaming passes -- per pass the same tail, chain count, chain length and kernel
    launches -- and write the same bytes."""
    import os, re, subprocess
    from psascan_amd import api, extras, pipeline
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n, block = 96 << 20, 20_000_003
    d_text = extras.gen_text(n, extras.MODE_ENGLISH, 0, seed=5)
    t = api.download(d_text, np.uint8, n)
    f = tmp_path / "x.bin"
    t.tofile(f)
    ram_use = 1_000_000_000
    stats = []
    sa5 = pipeline.construct_sa5(None, block, ram_use, extras.DeviceSorter(d_text, n), 0, stats, d_text=d_text, n=n)
    out = tmp_path / "x.sa5"
    r = subprocess.run([os.path.join(root, "host", "construct_sa"), "-m", str(ram_use), "--block-size", str(block), "--device-sort", "-v", "-o", str(out), str(f)],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="4", PSASCAN_PSA_ON_HOST="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(np.frombuffer(out.read_bytes(), np.uint8), sa5)
    cli = [tuple(int(x) for x in m) for m in re.findall(r"chains=(\d+) len=(\d+) warmup=\d+ unresolved=\d+ rounds=(\d+)", r.stderr)]
    mine = [(st.n_chains, st.chain_len, st.rounds) for (_, _, _, st) in stats]
    assert len(cli) == len(mine) >= 2 * (len(pipeline.block_plan(n, block, ram_use)) - 1), (len(cli), len(mine))
    assert cli == mine, (cli, mine)
