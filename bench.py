#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X-native pSAscan hot path.

N = 1 (default): BASELINE.json configs[2], the largest configuration one GPU holds -- 32 GiB of English-like text
(seeded synthetic: Zipfian words over a skewed alphabet, psgx_gen_text mode 3), 8 blocks of 4 GiB = 16 half-blocks.
One "step" = the whole block schedule of the reference's process_block / pSAscan on the GPU, right to left
(partial_sufsort.hpp:67-551, psascan.hpp:117-125):
    per block: rank over the left half's BWT, stream the right half (pass A), gap -> bitvector, merge the BWTs,
               rank over the block BWT (m = 2^32), stream the whole tail (pass B, 4..28 Gi suffixes in chunks of
               2^31), split the block gap into the half-blocks' merge bitvectors
    final merge of the 16 partial suffix arrays into 32 Gi uint40 entries.
Resident in HBM when the timed region starts: the text, every half-block's BWT, gt bits and i0.  The partial SAs
(128 GiB) do NOT fit next to the pass temporaries: they wait in pinned host memory (where the host sorter of
construct_sa leaves them) and are streamed through the device during the merge (psg_merge_stream); the .sa5
slices are verified on the device (permutation sum + sampled adjacent pairs per slice) and dropped, so the
timed step carries 128 GiB of H2D traffic for the PSAs and no D2H (`pcie` in the JSON line says how much; the
rate with the 160 GiB of output copied back to the host is reported as `with_output_d2h`).  The host suffix sort of
the half-blocks is NOT part of the hot path (north_star keeps it on host cores); the bench prepares the partial
SAs on the device before the timed region.  `end_to_end_cli` times the whole construct_sa program on a bounded
sample; `configs1_step` keeps last round's single-block step (4 GiB uniform bytes) as a secondary figure.

N > 1: north_star's multi-GPU split -- one 4 GiB block per GPU (text = N blocks), systolic rounds with ONE RCCL
point-to-point exchange of the gt slices per round, output-range partitioned merge (psascan_amd/blockdist.py, config_blocks()).
`--config 1` keeps last round's tail-sharded single-block job (rank-log all-to-all, config1()).

Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

A_STREAM = 39.25      # algorithmic bytes per streamed suffix (b = 64)
A_MERGE = 11.0        # algorithmic bytes per merged output suffix (merge.hpp:161)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
# HBM bytes per streamed suffix of the stream kernel, from the PMC passes committed in
# profiles/r01_pmc_summary.csv: (FETCH_SIZE + WRITE_SIZE) * 1024 / 2^31, keyed by rank layout
# (bytes of rank structure per BWT symbol):
#   8.3  = symbol-major layout with 8-byte entries, rank-log mode: (1.411e8 + 8.92e6) KB -> 71.5 B/suffix
#          (64 rank sector + 4 log + ~1 text + gt words; 16-byte entries: 71.4; before the 64-byte text blocks /
#          16-byte gt_out stores: 75.0; the first version -- interleaved blocks B=64 with atomics -- 170.6)
PMC_TRAFFIC_B_PER_SUFFIX = {8.3: 71.5, 16.0: 71.4}


def end_to_end_cli(sample_mib, log, api=None, extras=None, english=True, device_sort=False):
    """The whole construct_sa program with its defaults (-m 3584Mi, block size from the reference's formula) on a
    bounded sample, as a child process: map the file, suffix-sort the half-blocks (host cores: leaves sorted by SA-IS /
    prefix-key sorter on the host threads and merged on the device; device_sort=True: the --device-sort extension),
    the device passes, streamed merge, .sa5 to a file, the program's own device-side check of the output (--check)."""
    import subprocess
    import tempfile
    import numpy as np
    cli = os.path.join(ROOT, "host", "construct_sa")
    if not os.path.exists(cli):
        return None
    threads = min(16, os.cpu_count() or 1)
    n = sample_mib << 20
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "sample.bin")
        if english and api is not None:
            with open(f, "wb") as fh:
                for off in range(0, n, 1 << 30):
                    c = min(1 << 30, n - off)
                    d_t = extras.gen_text(c, extras.MODE_ENGLISH, 0, seed=11 + (off >> 30))
                    api.download(d_t, np.uint8, c).tofile(fh)
                    d_t.free()
        else:
            np.random.default_rng(11).integers(0, 255, n, dtype=np.uint8).tofile(f)
        t0 = time.time()
        r = subprocess.run([cli, "--check=1024", f] + (["--device-sort"] if device_sort else []), capture_output=True, text=True,
                           env=dict(os.environ, OMP_NUM_THREADS=str(threads)), timeout=900)
        wall = time.time() - t0
        ok = r.returncode == 0 and os.path.getsize(f + ".sa5") == 5 * n and "permutation sum ok, 0 of" in r.stderr
    if not ok:
        log("construct_sa failed:", r.stderr[-300:])
        return {"value": None, "unit": "MB/s", "sample": "failed"}
    peak = [l.strip() for l in r.stderr.splitlines() if "device memory" in l]
    inner = [l.strip() for l in r.stderr.splitlines() if "In-HBM merging" in l]
    prog = [l.strip() for l in r.stderr.splitlines() if l.strip().startswith("elapsed time:")]
    prog_s = float(prog[0].split()[2].rstrip("s")) if prog else None
    return {"value": n / 1e6 / wall, "unit": "MB/s", "host_threads": threads, "host_cpu_count": os.cpu_count(), "seconds": round(wall, 2),
            "seconds_in_program": prog_s, "value_in_program": (n / 1e6 / prog_s) if prog_s else None,
            "output_check": "permutation sum ok, 0 sampled pairs out of order",
            "device_memory": peak[0] if peak else None, "leaf_merging": inner[0] if inner else None,
            "sample": f"{sample_mib} MiB {'English-like text' if english else 'uniform bytes 0..254'} from a file, default -m (646 MiB blocks), "
                      + ("half-blocks sorted on the device (--device-sort, not the reference's placement)" if device_sort else
                         f"the program's default placement: half-blocks suffix-sorted on {threads} host threads as 32 KiB leaves (16-bit partial SAs), merged on the device one tree level per batch of passes")
                      + f", .sa5 ({5 * n >> 20} MiB) written to a file next to the input; wall time of the child process (process start to exit)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=0, help="2 = configs[2] block schedule (default at 1 GPU), 3 = block-per-GPU schedule (default at N > 1), 1 = configs[1] single-block step (at N > 1: tail-sharded)")
    ap.add_argument("--gib", type=float, default=0.0, help="text size in GiB (default: 32 for configs[2], 4 for configs[1])")
    ap.add_argument("--block-gib", type=float, default=0.0, help="configs[2]: block size in GiB (default text/8)")
    ap.add_argument("--text", choices=["bytes", "dna", "english"], default=None, help="alphabet of the synthetic text (default: english for configs[2], bytes for configs[1])")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[1] secondary figure and the CLI sample")
    ap.add_argument("--bwt-hbm", action="store_true", help="configs[2]: keep the BWT and gt_begin of every half-block resident in HBM (default: pinned host memory, uploaded one half-block ahead)")
    ap.add_argument("--psa-hbm-gib", type=float, default=-1.0, help="configs[2]: GiB of partial SAs kept resident in HBM next to the pass temporaries (default: as many half-blocks as fit, sized by an untimed step)")
    ap.add_argument("--with-output-d2h", action="store_true", help="(default now) configs[2]: also time one step with all partial SAs coming from the host and the .sa5 slices copied back")
    ap.add_argument("--no-output-d2h", action="store_true", help="configs[2]: skip that extra step")
    ap.add_argument("--e2e-mib", type=int, default=8192, help="size of the end_to_end_cli sample (default placement of construct_sa), MiB")
    ap.add_argument("--max-chains", type=int, default=0)
    ap.add_argument("--rank-block", type=int, default=0, help="data bytes per rank block (0=auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=int, default=512)
    ap.add_argument("--no-check", action="store_true")
    return ap.parse_args()


def cpu_baseline(api, extras, sample_mib, log, mode=None):
    """The reference's own compute_gap / convert_to_bitvector / merge (oracle/_ref, built from
    /root/reference where it lies) timed on this box's host cores on a bounded sample of the
    same workload.  Falls back to the C restatement (kind "port") when oracle/_ref is absent."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc                                      # checker / baseline only
    import shutil
    n = sample_mib << 20
    mid = n // 2
    mode = extras.MODE_BYTES255 if mode is None else mode
    d_text = extras.gen_text(n, mode, 0, seed=99)
    Lh = extras.sort_halfblock(d_text, n, 0, mid)
    Rh = extras.sort_halfblock(d_text, n, mid, n)
    text = api.download(d_text, np.uint8, n)
    lbwt = api.download(Lh["bwt"], np.uint8, mid)
    rgt = api.download(Rh["gt_begin"], np.uint8, (n - mid + 7) // 8)
    lpsa = api.download(Lh["psa_lo"], np.uint32, mid)
    rpsa = api.download(Rh["psa_lo"], np.uint32, n - mid)
    REF = orc.ref_lib()
    cores = min(16, os.cpu_count() or 1)
    T = n - mid
    if REF is not None:
        REF.ref_last_seconds.restype = C.c_double
        REF.ref_last_rank_build_seconds.restype = C.c_double
        REF.ref_set_threads.argtypes = [C.c_long]
        REF.ref_set_threads(cores)
        # per-thread start ranks (the reference gets them from em_compute_initial_ranks): here
        # from the device path -- rank at the end of thread chunk t = final rank of streaming [end_t, n)
        S = (T + cores - 1) // cores
        nthr = (T + S - 1) // S
        rk = api.rank_build(Lh["bwt"], mid)
        ir = np.zeros(nthr, np.int64)
        d_rgt = Rh["gt_begin"]
        for t in range(nthr):
            end_t = min(mid + (t + 1) * S, n)
            if end_t < n:
                scratch = api.gap_array(mid)
                ir[t], _ = api.stream_gap(rk, Lh["i0"], int(text[mid - 1]), d_text.at(end_t), n - end_t, d_rgt, 0, scratch, None)
                scratch.free()
        rk.free()
        wd = orc.workdir().encode()
        gap = np.zeros(mid + 1, np.uint64)
        gto = np.zeros(T // 8 + 2, np.uint8)
        REF.ref_compute_gap(lbwt, mid, Lh["i0"], int(text[mid - 1]), text, n, mid, n, rgt.ctypes.data, ir, nthr, wd, gap, gto)
        t_rank, t_stream = REF.ref_last_rank_build_seconds(), REF.ref_last_seconds()
        bv = np.zeros(n // 8 + 2, np.uint8)
        REF.ref_gap_to_bitvector(gap, mid, wd, bv, len(bv))
        t_bv = REF.ref_last_seconds()
        H = 2
        p32 = [lpsa.astype(np.int32), rpsa.astype(np.int32)]
        pp = (C.c_void_p * H)(*[p.ctypes.data for p in p32])
        gp = (C.c_void_p * H)(gap.ctypes.data, None)
        out = np.zeros(5 * n, np.uint8)
        REF.ref_merge(H, np.array([0, mid], np.int64), np.array([mid, n - mid], np.int64), pp, gp, 1 << 30, wd, out)
        t_merge = REF.ref_last_seconds()
        shutil.rmtree(wd.decode(), ignore_errors=True)
        total = t_rank + t_stream + t_bv + t_merge
        kind = "reference"
        detail = {"rank_build_s": round(t_rank, 3), "stream_s": round(t_stream, 3), "to_bitvector_s": round(t_bv, 3), "merge_s": round(t_merge, 3),
                  "stream_suffixes_per_s": T / t_stream}
    else:
        cores = 1
        t0 = time.time()
        rk = orc.Rank(lbwt)
        gap, gto, _ = orc.stream_pass(rk, Lh["i0"], int(text[mid - 1]), text, mid, n, rgt, 0)
        t_stream = time.time() - t0
        bv, _ = orc.gap_to_bitvector(gap, mid)
        out = orc.merge([0, mid], [mid, n - mid], [lpsa.astype(np.int64), rpsa.astype(np.int64)], [gap, None])
        total = time.time() - t0
        kind = "port"
        detail = {"stream_s": round(t_stream, 3), "stream_suffixes_per_s": T / t_stream}
    # the CPU result doubles as a parity check of the device path on the sample
    bad, _ = extras.check_sa5(d_text, n, api.upload(out), n, samples=1 << 16)
    detail["sample_sa_bad_pairs"] = bad
    for b in (d_text, Lh["bwt"], Lh["psa_lo"], Lh["gt_begin"], Rh["bwt"], Rh["psa_lo"], Rh["gt_begin"]):
        b.free()
    return {"value": n / 1e6 / total, "unit": "MB/s", "cores": cores, "host_cpu_count": os.cpu_count(), "kind": kind,
            "sample": f"{sample_mib} MiB {'English-like text' if mode == extras.MODE_ENGLISH else 'uniform bytes 0..254'}, two {sample_mib // 2} MiB half-blocks: rank build + stream + gap->bitvector + merge, seconds={total:.2f}",
            **detail}


def setup(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    if world > 1 or args.config == 3:                                # --config 3 on one GPU: the block schedule with a world of one (its full-size block)
        import torch
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("PSASCAN_DIST_BACKEND", "nccl")     # "gloo" + PSASCAN_SHARE_GPU=1: rehearsal on one GPU
        if os.environ.get("PSASCAN_SHARE_GPU") == "1":
            local = 0
        torch.cuda.set_device(local)
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            except TypeError:      # older torch: no device_id argument
                dist.init_process_group("nccl")
        else:
            dist.init_process_group(backend)
    import numpy as np
    import psascan_amd
    from psascan_amd import api, extras
    L = psascan_amd.lib(local)
    numa_node = C.c_int(-1)
    L.psgx_bind_threads_near_device(C.byref(numa_node))      # pinned buffers and copies on the device's socket (a hint)
    if dist is not None:
        # everything on ONE explicit stream so RCCL collectives and our kernels are ordered.  (torch's default
        # stream has handle 0, which psg_set_stream takes as "create your own": make a real stream current.)
        shared_stream = torch.cuda.Stream()
        torch.cuda.set_stream(shared_stream)
        assert shared_stream.cuda_stream != 0
        L.psg_set_stream(C.c_void_p(shared_stream.cuda_stream))

    def log(*a):
        if rank == 0:
            print("[bench]", *a, file=sys.stderr, flush=True)

    return rank, world, local, dist, torch, np, api, extras, L, log


def config1(args, ctx, gib, steps, warmup, text_mode, with_baselines):
    """configs[1]-shaped step: one block of two half-blocks (rank build + pass A + gap -> bitvector + merge); at
    N > 1 the pass is sharded over the tail (rank-log all-to-all).  Returns the result dict on rank 0."""
    rank, world, local, dist, torch, np, api, extras, L, log = ctx
    from psascan_amd import distributed as D
    n = int(gib * (1 << 30)) // 128 * 128
    mid = n // 2
    ls, rs = mid, n - mid
    t0 = time.time()
    d_text = extras.gen_text(n, {"dna": extras.MODE_DNA, "english": extras.MODE_ENGLISH}.get(text_mode, extras.MODE_BYTES255), 0, seed=2)
    Rh = extras.sort_halfblock(d_text, n, mid, n)
    Lh = extras.sort_halfblock(d_text, n, 0, mid)
    api.sync()
    log(f"prepared {n / 2 ** 30:.2f} GiB text + 2 half-block SAs in {time.time() - t0:.1f}s (ties L/R {Lh['tie_groups']}/{Rh['tie_groups']})")
    last_left = int(api.download(d_text, np.uint8, 1, mid - 1)[0])
    Lh["psa_hi"] = Rh["psa_hi"] = None

    # tail range of this rank (64-aligned cut points so gt words do not straddle ranks)
    cuts = D.tail_cuts(mid, n, world)
    tb, te = cuts[rank], cuts[rank + 1]
    out_cuts = D.output_cuts(n, world)
    ob, oe = out_cuts[rank], out_cuts[rank + 1]
    ctx = D.context_len(te, n)           # right context for the start rank of this range
    gt_words = max(cuts[r + 1] - cuts[r] for r in range(world)) // 32 + 4
    gap_words = api.gap_words(ls)
    if world > 1:
        a2a_ops = D.HipA2AOps(torch, api, "cuda", full_sync=True)   # device-wide sync around every collective: cheap next to the collectives, and independent of stream identity
        gt_mine = torch.zeros(gt_words, dtype=torch.int32, device="cuda")
    else:
        gap_buf = api.zeros(4 * gap_words)
        gap_ptr = gap_buf.ptr
    gt_out = api.zeros(4 * ((te - tb + 31) // 32 + 2))
    mbv = api.zeros(4 * ((n + 31) // 32 + 2))
    d_out = api.DeviceBuffer(5 * (oe - ob) + 16)
    # gt_in slice for this range: bits of Rh.gt_begin (u = n - j) for j in (tb, te+ctx]
    gt_in = api.zeros(4 * ((te + ctx - tb + 31) // 32 + 2))
    api.bitcopy(gt_in, 0, Rh["gt_begin"], n - (te + ctx), te + ctx - tb)
    api.sync()

    times = {"rank_build": 0.0, "stream": 0.0, "stream_kernel": 0.0, "comm": 0.0, "to_bv": 0.0, "merge": 0.0}
    stats_last = None

    def step(timed):
        nonlocal stats_last
        t = time.perf_counter()
        rk = api.rank_build(Lh["bwt"], ls, args.rank_block)
        t1 = time.perf_counter()
        start_rank = 0 if te + ctx == n else -1   # exact only at n (rank of the empty suffix = 0)
        if world == 1:
            # like the reference, the pass starts from a fresh gap array (partial_sufsort.hpp:405): no memset here,
            # the library zero-fills or overwrites it (PSG_GAP_UNINITIALIZED)
            fin, st = api.stream_gap(rk, Lh["i0"], last_left, d_text.at(tb), te - tb, gt_in, start_rank, gap_ptr, gt_out,
                                     args.max_chains, right_context=ctx, fresh_gap=True)
            t2 = t3 = time.perf_counter()
            if os.environ.get("PSG_TIMING"):
                ta = time.perf_counter(); api.sync(); tb_ = time.perf_counter(); api.sync(); tc = time.perf_counter()
                print(f"[bench] after stream_gap: first sync {1e3*(tb_-ta):.2f} ms, second sync {1e3*(tc-tb_):.2f} ms", file=sys.stderr)
                t2 = t3 = time.perf_counter()
            nb = api.gap_to_bitvector(gap_ptr, ls, mbv, n)
            assert nb == n, (nb, n)
            Lh["mbv"] = mbv
            t4 = time.perf_counter()
            if timed:
                times["to_bv_device"] = times.get("to_bv_device", 0.0) + api.last_kernel_ms() / 1e3
                times["stream_hist_device"] = times.get("stream_hist_device", 0.0) + st.hist_ms / 1e3
        else:
            # gap array sharded by index range: rank-log all-to-all, slice histograms, bit all-reduce
            # (psascan_amd/distributed.py: a2a_pass; gloo-tested in tests/test_distributed_cpu.py)
            box = {}

            def stream_log_fn(tb_r, te_r, ctx_r):
                log, nlog, fin, st_ = api.stream_gap_log(rk, Lh["i0"], last_left, d_text.at(tb_r), te_r - tb_r, gt_in, start_rank,
                                                         gt_mine.data_ptr(), args.max_chains, ctx_r)
                box["st"], box["log"] = st_, log
                box["t2"] = time.perf_counter()
                return log.ptr, nlog, gt_mine

            res = D.a2a_pass(dist, a2a_ops, world, rank, ls, mid, n, stream_log_fn, gt_words)
            torch.cuda.current_stream().synchronize()
            box["log"].free()
            st = box["st"]
            assert res["nbits"] == n, (res["nbits"], n)
            Lh["mbv"] = res["bits"].data_ptr()
            box["keep"] = res
            t2 = box["t2"]
            t3 = t4 = time.perf_counter()
        plan = api.MergePlan([Lh, Rh])
        plan.run(ob, oe - ob, d_out)
        plan.free()
        rk_bytes = rk.device_bytes()
        rk.free()
        t5 = time.perf_counter()
        if timed:
            times["rank_build"] += t1 - t; times["stream"] += t2 - t1; times["stream_kernel"] += st.kernel_ms / 1e3
            times["comm"] += t3 - t2; times["to_bv"] += t4 - t3; times["merge"] += t5 - t4
        stats_last = (st, rk_bytes)

    def barrier():
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        api.sync()

    for _ in range(warmup):
        step(False)
    barrier()
    t_start = time.perf_counter()
    for _ in range(steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- property checks at full size (untimed): every output slice is sorted and the whole is a permutation
    check = None
    if not args.no_check:
        bad, s = extras.check_sa5(d_text, n, d_out, oe - ob, samples=1 << 20)
        if world > 1:
            v = torch.tensor([bad, s & 0x7FFFFFFFFFFFFFFF, s >> 63], dtype=torch.int64, device="cuda")
            parts = [torch.empty_like(v) for _ in range(world)]
            dist.all_gather(parts, v)
            bad = sum(int(p[0]) for p in parts)
            s = sum(int(p[1]) + (int(p[2]) << 63) for p in parts) % (1 << 64)
        want = (n * (n - 1) // 2) % (1 << 64)
        check = {"sampled_adjacent_pairs_out_of_order": bad, "sum_matches_permutation": s == want}
        if bad or s != want:
            log("PROPERTY CHECK FAILED", check)

    if rank == 0:
        st, rk_bytes = stats_last
        K = steps
        per = {k: v / K for k, v in times.items()}
        stream_suffixes = te - tb
        kernel_s = per["stream_kernel"]
        achieved = A_STREAM * stream_suffixes / kernel_s / 1e9 if kernel_s > 0 else 0.0
        res = {
            "metric": "input MB/s, hot path (rank build + gap-stream + gap->bitvector + merge to .sa5), inputs resident in HBM",
            "value": n * K / 1e6 / elapsed, "unit": "MB/s", "n_gpus": world, "steps": K, "warmup": warmup,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8/u32/u40 integer", "data": "synthetic",
            "config": {"workload": (f"configs[1]: {n / 2 ** 30:.2f} GiB uniform random bytes 0..254 (sigma=255; byte 255 is reserved by the reference)" if text_mode == "bytes" else f"{n / 2 ** 30:.2f} GiB {text_mode} text, not configs[1]'s alphabet") + f", one block = two {mid / 2 ** 30:.2f} GiB half-blocks, single pass A + merge",
                       "text_bytes": n, "half_blocks": 2, "tail_sharding": f"dp{world}" if world > 1 else "none",
                       "chains": st.n_chains, "chain_len": st.chain_len, "rank_bytes_per_symbol": rk_bytes / ls},
            "gap_stream_suffixes_per_s": rs * 1.0 / (per["stream"] + per["comm"]) if world > 1 else stream_suffixes / per["stream"],
            "gap_stream_kernel_suffixes_per_s": stream_suffixes / kernel_s if kernel_s else None,
            "phase_ms": {k: round(1e3 * v, 3) for k, v in per.items()},
            "roofline": {"bound": "hbm", "kernel": "stream_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (PMC_TRAFFIC_B_PER_SUFFIX.get(round(rk_bytes / ls, 1)) or 0) * stream_suffixes or None,
                         "traffic_source": "PMC FETCH_SIZE+WRITE_SIZE, profiles/r01_pmc_summary.csv (bytes per launch)",
                         "random_access_ceiling": "profiles/r01_membench.txt: dependent random 16-byte loads (one 64 B sector each) top out at 48-51 G/s on this chip (46.7 G/s with this kernel's log stores and text loads mixed in); one sector per suffix is this kernel's floor, i.e. >= 42 ms per 2^31 suffixes",
                         "algorithmic_bytes_per_suffix": A_STREAM, "suffixes_per_launch": stream_suffixes,
                         "avg_launch_ms": 1e3 * kernel_s},
            "merge_roofline": {"achieved": A_MERGE * (oe - ob) / per["merge"] / 1e9, "unit": "GB/s", "note": "includes plan build (rank samples over the merge bitvector)"},
            "property_check": check,
        }
        if world == 1 and with_baselines and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(api, extras, args.cpu_sample_mib, log)
            except Exception as e:  # the baseline must not take the bench down
                res["cpu_baseline"] = {"value": None, "unit": "MB/s", "cores": 0, "kind": "reference", "sample": f"failed: {e!r}"}
            try:
                res["end_to_end_cli"] = end_to_end_cli(1024, log, english=False)
            except Exception as e:
                res["end_to_end_cli"] = {"value": None, "unit": "MB/s", "sample": f"failed: {e!r}"}
        for b in (d_text, Lh["bwt"], Lh["psa_lo"], Lh["gt_begin"], Rh["bwt"], Rh["psa_lo"], Rh["gt_begin"], gt_out, mbv, d_out, gt_in):
            b.free()
        if world == 1:
            gap_buf.free()
        return res
    return None


def reference_ram_use(block, threads=16):
    """-m that gives the reference this max_block_size with `threads` threads (psascan.hpp:73-91): the last block's
    left half is min(block, ram_use / 10) (partial_sufsort.hpp:86-93), so the split depends on it."""
    g = 1 << 21
    rft = 2 * threads * g + int(0.8 * threads) * g + threads * g + threads * (6 << 20)
    return int(block * 5.2) + rft + 8


def config2(args, ctx, n, block):
    """configs[2]: the whole block schedule on one GPU (see the module docstring)."""
    rank, world, local, dist, torch, np, api, extras, L, log = ctx
    from psascan_amd import pipeline
    ram_use = reference_ram_use(block)
    plan = pipeline.block_plan(n, block, ram_use)
    mode = {"bytes": extras.MODE_BYTES255, "dna": extras.MODE_DNA, "english": extras.MODE_ENGLISH}[args.text]
    t0 = time.time()
    d_text = extras.gen_text(n, mode, 0, seed=3)
    prepared, pins = {}, []
    tsort = tpin = 0.0
    for (b, mid, e) in plan:
        for hb, he in ((mid, e), (b, mid)):
            if he <= hb:
                continue
            t1 = time.time()
            r = extras.sort_halfblock(d_text, n, hb, he)
            api.sync()
            tsort += time.time() - t1
            init = None
            if hb == b and e > mid:      # left half: start rank of pass A = #{s in L : text[s..) < text[e..)}, by string search
                sc = api.search_ctx(d_text, n, n, None, [(hb, he - hb, r["psa_lo"], None)])
                init = int(api.initial_ranks(sc, [e])[0])
            t1 = time.time()
            pa = api.PinnedArray(he - hb, np.uint32)            # the partial SA waits in pinned host memory
            api.check(L.psg_d2h(pa.ptr, r["psa_lo"].ptr, 4 * (he - hb)))
            r["psa_lo"].free()
            pins.append(pa)
            ent = {"device": True, "psa_host": pa.array, "psa_lo": None, "bwt": r["bwt"], "gt_begin": r["gt_begin"], "i0": r["i0"], "size": he - hb,
                   "initA": init, "keep_inputs": True}
            if not args.bwt_hbm:
                # the sorter's other products (BWT, gt_begin) wait in pinned host memory too, as they would behind a host
                # sorter; the step uploads them one half-block ahead of the schedule, in the background
                gw = (he - hb + 31) // 32 + 2
                pb, pg = api.PinnedArray(he - hb + 16, np.uint8), api.PinnedArray(gw, np.uint32)
                api.check(L.psg_d2h(pb.ptr, r["bwt"].ptr, he - hb))
                api.check(L.psg_d2h(pg.ptr, r["gt_begin"].ptr, min(4 * gw, r["gt_begin"].nbytes)))
                r["bwt"].free(); r["gt_begin"].free()
                pins += [pb, pg]
                ent.update({"bwt": None, "gt_begin": None, "bwt_host": pb.array, "gt_host": pg.array, "keep_inputs": False})
            tpin += time.time() - t1
            prepared[(hb, he)] = ent
    L.psg_trim()        # the sorter's temporaries go back to the driver: the step starts from the resident inputs only
    log(f"prepared {n / 2 ** 30:.2f} GiB {args.text} text, {len(prepared)} half-blocks in {time.time() - t0:.1f}s (device sort {tsort:.1f}s, pinned alloc + D2H {tpin:.1f}s)")

    order = [k for (b, mid, e) in plan for k in ((mid, e), (b, mid)) if k[1] > k[0]]     # the schedule's order of half-blocks

    class Replay:
        device = True
        inflight = {}

        def prefetch(self, key):
            if args.bwt_hbm or key in self.inflight:
                return
            v = prepared[key]
            d_b, d_g = api.DeviceBuffer(v["bwt_host"].nbytes), api.DeviceBuffer(v["gt_host"].nbytes)
            self.inflight[key] = (d_b, d_g, api.BackgroundUpload(d_b, v["bwt_host"]), api.BackgroundUpload(d_g, v["gt_host"]))

        def __call__(self, text, beg, end, gt_tail):
            r = dict(prepared[(beg, end)])
            if not args.bwt_hbm:
                self.prefetch((beg, end))
                d_b, d_g, u1, u2 = self.inflight.pop((beg, end))
                u1.wait(); u2.wait()
                r["bwt"], r["gt_begin"] = d_b, d_g
                k = order.index((beg, end))
                for d in (1, 2):                                  # both halves of the next block go up during this block's passes
                    self.prefetch(order[(k + d) % len(order)])
            return r
    replay = Replay()

    want_sum = (n * (n - 1) // 2) % (1 << 64)
    agg = {"stream_ms": 0.0, "kernel_ms": 0.0, "hist_ms": 0.0, "suffixes": 0, "launches": 0, "passes_s": 0.0, "merge_s": 0.0, "merge_kernel_ms": 0.0,
           "h2d": 0, "d2h": 0, "stage_ms": 0.0}
    last = {}

    def step(timed, sink=None):
        stats, tm = [], {}
        ts = time.perf_counter()
        ms, chk = pipeline.construct_sa5(None, block, ram_use, replay, args.max_chains, stats, d_text=d_text, n=n, merge="stream",
                                         sink=sink, check_samples=4096, timings=tm)
        api.sync()
        te = time.perf_counter()
        if chk != (want_sum, 0):
            raise RuntimeError(f"output check failed: sum ok {chk[0] == want_sum}, {chk[1]} sampled adjacent pairs out of order")
        if timed:
            for kind, b, e, st in stats:
                T = (e - b) - (e - b + 1) // 2 if kind == "A" else n - e
                for (pb, pm, pe) in plan:
                    if pb == b and kind == "A":
                        T = pe - pm
                agg["stream_ms"] += st.total_ms; agg["kernel_ms"] += st.kernel_ms; agg["hist_ms"] += st.hist_ms
                agg["suffixes"] += T; agg["launches"] += st.rounds
            agg["passes_s"] += tm["passes_done"] - ts; agg["merge_s"] += te - tm["passes_done"]
            agg["merge_kernel_ms"] += ms.kernel_ms; agg["h2d"] += ms.h2d_bytes; agg["d2h"] += ms.d2h_bytes; agg["stage_ms"] += ms.stage_ms
        last["stats"], last["ms"] = stats, ms
        return te - ts

    # ---- partial SAs that fit in HBM next to the pass temporaries are resident inputs (no PCIe for them); the first
    # untimed step measures the peak of everything else
    warm = args.warmup
    resident = 0
    if args.psa_hbm_gib != 0:
        step(False)
        warm = max(0, warm - 1)
        _, peak0, _ = api.mem_stats()
        total = api.device_memory()[1]
        budget = int(args.psa_hbm_gib * 2 ** 30) if args.psa_hbm_gib > 0 else total - peak0 - (36 << 30)   # headroom: arena granules, fragmentation
        L.psg_trim()
        placed = []
        for key in sorted(prepared, key=lambda k: -k[0]):           # rightmost half-blocks first
            v = prepared[key]
            if 4 * v["size"] + (1 << 20) > budget:
                continue
            try:
                d = api.DeviceBuffer(4 * v["size"] + 16)
            except Exception:
                break                                               # no room after all: the rest stays in host memory
            api.check(L.psg_h2d(d.ptr, v["psa_host"].ctypes.data, 4 * v["size"]))
            v["psa_lo"], v["psa_pinned"], v["psa_host"] = d, v["psa_host"], None
            budget -= 4 * v["size"] + (1 << 20)
            placed.append(key)
        # the device arena was trimmed to place them: one untimed step lets it grow back.  The steps allocate in the same
        # order every time, so a settling step that fits means the timed steps fit; if it does not, the last partial SA
        # placed goes back to host memory and the step is tried again.
        while True:
            failed = None
            try:
                dt = step(False)
            except Exception as ex:
                failed = str(ex)
            if failed is None:
                break
            if not placed or "memory" not in failed.lower():
                raise RuntimeError(failed)
            import gc
            gc.collect()                                             # the failed step's buffers
            v = prepared[placed.pop()]
            v["psa_lo"].free()
            v["psa_lo"], v["psa_host"] = None, v["psa_pinned"]
            for (d_b, d_g, u1, u2) in replay.inflight.values():
                u1.wait(); u2.wait(); d_b.free(); d_g.free()
            replay.inflight.clear()
            L.psg_trim()
            log(f"settling step ran out of device memory: one partial SA back to host memory ({len(placed)} resident)")
        resident = len(placed)
        log(f"{resident} of {len(prepared)} partial SAs resident in HBM ({sum(4 * v['size'] for v in prepared.values() if v['psa_host'] is None) / 2 ** 30:.1f} GiB), the rest in pinned host memory")
        warm = max(0, warm - 1)
        log(f"settling step: {dt:.2f}s")
    for k in range(warm):
        dt = step(False)
        log(f"warm-up step {k + 1}/{warm}: {dt:.2f}s")
    api.sync()
    t_start = time.perf_counter()
    for k in range(args.steps):
        dt = step(True)
        log(f"step {k + 1}/{args.steps}: {dt:.2f}s")          # a progress line per step (a step takes several seconds)
    api.sync()
    elapsed = time.perf_counter() - t_start
    K = args.steps
    in_use, peak, reserved = api.mem_stats()
    with_d2h = None
    if not args.no_output_d2h:
        # one more step the way a run behind a host sorter sees it: NO partial SA resident in HBM (all 4 bytes per symbol
        # come in from pinned host memory during the merge) and every .sa5 slice copied back to the host
        for v in prepared.values():
            if v.get("psa_host") is None and v.get("psa_pinned") is not None:
                v["psa_lo"].free()
                v["psa_lo"], v["psa_host"] = None, v["psa_pinned"]
        got = [0]

        def sink(view, first, cnt):
            got[0] += cnt
        agg_keep = dict(agg)
        try:
            dt = step(True, sink)
            ms = last["ms"]
            with_d2h = {"value": n / 1e6 / dt, "unit": "MB/s", "seconds": round(dt, 3), "entries_received_on_host": got[0],
                        "h2d_bytes": int(ms.h2d_bytes), "d2h_bytes": int(ms.d2h_bytes), "partial_sas_resident_in_hbm": 0,
                        "note": "one step with ALL partial SAs streamed in from pinned host memory (4 bytes per suffix H2D) and every .sa5 slice copied to pinned host "
                                "memory (5 bytes per suffix D2H) and handed to a sink that drops it: what the schedule costs behind a host sorter, without the file write"}
        except Exception as ex:
            with_d2h = {"value": None, "note": f"failed: {ex!r}"}
        agg.clear(); agg.update(agg_keep)
    suff, kern_s = agg["suffixes"] / K, agg["kernel_ms"] / K / 1e3
    achieved = A_STREAM * suff / kern_s / 1e9 if kern_s > 0 else 0.0
    rankB = [getattr(p[3], "rank_bytes", 0) / (p[2] - p[1]) for p in last["stats"] if p[0] == "B"]
    traffic_tab = {}
    try:
        traffic_tab = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        pass
    tr = traffic_tab.get(f"configs2_{args.text}")
    launches = agg["launches"] / K
    res = {
        "metric": "input MB/s, hot path of the whole block schedule (rank builds + all gap-stream passes + gap->bitvector + BWT merge + gap split + final merge to .sa5)",
        "value": n * K / 1e6 / elapsed, "unit": "MB/s", "n_gpus": 1, "steps": K, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / K,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32/u40 integer", "data": "synthetic",
        "config": {"workload": f"configs[2]: {n / 2 ** 30:.2f} GiB {'English-like text (seeded Zipfian words, sigma=28)' if args.text == 'english' else args.text + ' text'}, "
                               f"{len(plan)} blocks of {block / 2 ** 30:.2f} GiB = {len(prepared)} half-blocks on 1 GPU, whole schedule (passes A and B of every block + final merge)",
                   "text_bytes": n, "blocks": len(plan), "half_blocks": len(prepared), "block_bytes": block,
                   "resident_in_hbm": f"text, {'BWT + gt bits of every half-block, ' if args.bwt_hbm else ''}merge bitvectors, {resident} of {len(prepared)} partial suffix arrays",
                   "in_pinned_host_memory": f"{len(prepared) - resident} of {len(prepared)} partial suffix arrays (4 B/symbol), streamed during the merge"
                                            + ("" if args.bwt_hbm else "; BWT + gt_begin of every half-block (the host sorter's products), uploaded one half-block ahead of the schedule in the background"),
                   "rank_bytes_per_symbol_pass_B": round(sum(rankB) / max(1, len(rankB)), 3)},
        "gap_stream_suffixes_per_s": suff / (agg["stream_ms"] / K / 1e3),
        "gap_stream_kernel_suffixes_per_s": suff / kern_s if kern_s else None,
        "streamed_suffixes_per_step": suff,
        "phase_ms": {"passes_total": round(1e3 * agg["passes_s"] / K, 1), "stream_passes": round(agg["stream_ms"] / K, 1), "stream_kernel": round(agg["kernel_ms"] / K, 1),
                     "stream_partition_hist": round(agg["hist_ms"] / K, 1), "merge_total": round(1e3 * agg["merge_s"] / K, 1),
                     "merge_kernels": round(agg["merge_kernel_ms"] / K, 1), "merge_host_staging": round(agg["stage_ms"] / K, 1)},
        "pcie": {"h2d_bytes_per_step": agg["h2d"] // K, "d2h_bytes_per_step": agg["d2h"] // K,
                 "h2d_halfblock_inputs_bytes_per_step": 0 if args.bwt_hbm else sum(v["bwt_host"].nbytes + v["gt_host"].nbytes for v in prepared.values()),
                 "note": "partial SAs stream in from pinned host memory during the merge (h2d_bytes_per_step); BWT + gt_begin of the half-blocks go up in the background during the passes; the output is checked on the device and dropped"},
        "with_output_d2h": with_d2h,
        "device_memory": {"peak_in_use_gib": round(peak / 2 ** 30, 2), "reserved_gib": round(reserved / 2 ** 30, 2)},
        "roofline": {"bound": "hbm", "kernel": "stream_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": (tr["bytes_per_suffix"] * suff / launches) if tr else None,
                     "traffic_source": (tr["source"] if tr else "no PMC pass recorded for this workload in profiles/pmc_traffic.json"),
                     "algorithmic_bytes_per_suffix": A_STREAM, "suffixes_per_launch": suff / launches, "launches_per_step": launches,
                     "avg_launch_ms": 1e3 * kern_s / launches,
                     "random_access_ceiling": "profiles/r01_membench.txt: dependent random 16-byte loads (one 64 B sector each) top out at 48-51 G/s on this chip; one sector per suffix is this kernel's floor"},
        "merge_roofline": {"achieved": A_MERGE * n / (agg["merge_s"] / K) / 1e9, "unit": "GB/s",
                           "note": "PCIe-bound: 4 bytes per suffix of partial SA cross the link (H2D) inside the timed merge"},
        "property_check": {"every_step": "permutation sum == n(n-1)/2 and 4096 sampled adjacent pairs per 64 Mi-entry slice in suffix order, on the device"},
    }
    for (d_b, d_g, u1, u2) in replay.inflight.values():
        u1.wait(); u2.wait(); d_b.free(); d_g.free()
    replay.inflight.clear()
    for v in prepared.values():
        if v.get("bwt") is not None:
            v["bwt"].free(); v["gt_begin"].free()
        if v.get("psa_lo") is not None:
            v["psa_lo"].free()
    for pa in pins:
        pa.free()
    d_text.free()
    L.psg_trim()
    return res


def config_blocks(args, ctx, block):
    """N > 1: north_star's multi-GPU split (psascan_amd/blockdist.py) in BASELINE configs[3]'s shape -- DNA, one block of
    `block` symbols per GPU (16 GiB: half-blocks of 2^33 symbols, 40-bit partial SAs), one exchange of the gt slices per
    round over RCCL, output-range partitioned merge in sub-ranges.  Weak scaling in the data a GPU holds (text = N blocks).
    No rank holds the whole text: block q is the seeded generator's output for seed 1000 + q, a rank keeps its own block
    (+ look-ahead) and produces the chunk of a round on its device when the round needs it (a real run uploads it from
    the memory-mapped file).  Untimed preparation: the suffix sort of the rank's two half-blocks (pieces of 2^31 merged
    with the hot path), the start ranks found by string search while the partial SAs are on the device (pass A's and
    the far chunks': a handful of searches), the partial SAs moved to pinned host memory -- where the host sorter of
    construct_sa leaves them; the merge pieces go up from there inside the timed step."""
    rank, world, local, dist, torch, np, api, extras, L, log = ctx
    from psascan_amd import blockdist as BD
    n = block * world
    mode = {"bytes": extras.MODE_BYTES255, "dna": extras.MODE_DNA, "english": extras.MODE_ENGLISH}[args.text]
    t0 = time.time()
    bounds = BD.block_bounds(n, world)
    b, e = bounds[rank], bounds[rank + 1]
    mid = b + (e - b) // 2
    LOOK = BD.HipBlockOps.LOOKAHEAD

    def load(lo, hi):
        """text[lo .. hi) on the device: every block is the generator's output for its own seed (prefix-consistent)"""
        d = api.zeros((hi - lo + 63) // 64 * 64 + 64)
        for q in range(world):
            a, z = max(lo, bounds[q]), min(hi, bounds[q + 1])
            if a < z:
                assert a == bounds[q], "ranges start on block boundaries"
                extras.gen_text(z - a, mode, 0, seed=1000 + q, d_text=d.ptr + (a - lo))
        return d
    own_hi = min(n, e + LOOK)
    src = BD.ChunkedText(load, n, b, own_hi)
    base, wlo, whi = src.window(b, own_hi)
    comm = os.environ.get("PSASCAN_COMM") or ("cuda" if os.environ.get("PSASCAN_DIST_BACKEND", "nccl") == "nccl" else "cpu")
    wide = (e - b) // 2 + 1 >= (1 << 32)
    rounds = max(1, int(os.environ.get("PSASCAN_MERGE_ROUNDS", "0")) or -(-5 * (e - b) // (8 << 30)))   # ~8 GiB of pieces per all-to-all
    prepared, pins = {}, []
    for hb, he in ((mid, e), (b, mid)):
        r = extras.sort_halfblock_pieces(base, own_hi, (b, own_hi), hb, he, piece_max=int(os.environ.get("PSASCAN_TEST_PIECE_MAX", str(1 << 31))))
        ent = {"device": True, "bwt": r["bwt"], "gt_begin": r["gt_begin"], "i0": r["i0"], "size": he - hb, "keep_inputs": True}
        prepared[(hb, he)] = (r, ent)
    # start ranks while the partial SAs are on the device: pass A's (rank of text[e..) among the left half's suffixes) and the
    # far chunks' (rank of text[e_q..) among the block's suffixes)
    tmp_ops = BD.HipBlockOps(torch, api, src, n, None, comm=comm)
    (rR, entR), (rL, entL) = prepared[(mid, e)], prepared[(b, mid)]
    scA = tmp_ops._search_ctx(b, own_hi, [(b, mid - b, rL["psa_lo"], rL.get("psa_hi"))])
    entL["initA"] = int(api.initial_ranks(scA, [e])[0])

    class _St:
        pass
    st0 = _St()
    st0.b, st0.mid, st0.e = b, mid, e
    st0.L, st0.R = {"psa_lo": rL["psa_lo"], "psa_hi": rL.get("psa_hi")}, {"psa_lo": rR["psa_lo"], "psa_hi": rR.get("psa_hi")}
    ends = [bounds[q + 1] for q in range(rank + 1, world)]
    entL["start_ranks"] = dict(zip(ends, tmp_ops.start_ranks(st0, ends))) if ends else {}
    for (r, ent) in (prepared[(mid, e)], prepared[(b, mid)]):     # partial SAs -> pinned host memory
        sz = ent["size"]
        pa = api.PinnedArray(sz, np.uint32)
        api.check(L.psg_d2h(pa.ptr, r["psa_lo"].ptr, 4 * sz))
        r["psa_lo"].free()
        ent["psa_lo"] = pa.array
        pins.append(pa)
        if r.get("psa_hi") is not None:
            ph = api.PinnedArray(sz, np.uint8)
            api.check(L.psg_d2h(ph.ptr, r["psa_hi"].ptr, sz))
            r["psa_hi"].free()
            ent["psa_hi"] = ph.array
            pins.append(ph)
    L.psg_trim()
    api.sync()
    log(f"rank 0 prepared its block of {n / 2 ** 30:.2f} GiB {args.text} text in {time.time() - t0:.1f}s ({'40-bit' if wide else '32-bit'} partial SAs, merge in {rounds} rounds)")

    # helper ranks (blockdist.helper_of) cost the helper a replica of its partner's rank structure and BWT, a gap array of
    # its own (4 bytes per block symbol) and the two tensors of the gap reduce: on when that fits next to the rank's own
    # ~12 bytes per block symbol of resident state (PSASCAN_HELPERS=1 / 0 forces it).  At configs[3]'s 16 GiB blocks it does
    # not (191 + 221 GiB): the run is then the plain systolic schedule.
    henv = os.environ.get("PSASCAN_HELPERS", "auto")
    if henv == "auto":
        rank_bps = {"dna": 0.8, "english": 4.0}.get(args.text, 8.5)
        own = block * (1 + 2 + 1 + 0.25 + rank_bps + 4 + 1.7 + 1.2)
        extra = block * (rank_bps + 1 + 4 + 4 + 4)
        helpers = own + extra <= 0.85 * api.device_memory()[1]
    else:
        helpers = henv != "0"

    def replay(text, hb, he, gt_tail):
        return dict(prepared[(hb, he)][1])
    # tensors handed to the collectives: CUDA tensors with RCCL; CPU tensors in the gloo rehearsal (PSASCAN_COMM=cuda
    # rehearses the CUDA-tensor code path over gloo)
    ops = BD.HipBlockOps(torch, api, src, n, replay, comm=comm, max_chains=args.max_chains, keep_output_on_device=True, merge_rounds=rounds,
                         force_wide=any(ent.get("psa_hi") is not None for (_, ent) in prepared.values()),   # (rehearsals force pieces: PSASCAN_TEST_PIECE_MAX)
                         helpers=helpers)
    wide = wide or ops.force_wide
    agg = {"suffixes": 0, "kernel_ms": 0.0, "stream_ms": 0.0, "launches": 0}

    def step(timed):
        stats = []
        x0, x1, _ = BD.run(dist, ops, world, rank, n, stats)
        if timed:
            for (_, q, st) in stats:
                agg["suffixes"] += bounds[q + 1] - bounds[q]; agg["kernel_ms"] += st.kernel_ms; agg["stream_ms"] += st.total_ms; agg["launches"] += st.rounds
        return x0, x1

    def barrier():
        dist.barrier()
        if comm == "cuda":
            torch.cuda.synchronize()
        api.sync()
    for _ in range(args.warmup):
        step(False)
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        x0, x1 = step(True)
    barrier()
    elapsed = time.perf_counter() - t_start
    tt = torch.tensor([elapsed], dtype=torch.float64, device=comm)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    # ---- property check (untimed extra step): possible where the whole text fits next to everything else (rehearsals and
    # small blocks): every round's output is checked on the device against the whole text -- sampled adjacent pairs in
    # suffix order, sum of all entries
    chk = None
    if n <= (int(os.environ.get("PSASCAN_CHECK_MAX_GIB", "6")) << 30) and not args.no_check:   # (a rehearsal with HBM to spare raises the bound)
        d_whole = load(0, n)
        ops.check_text, ops.check_acc = (d_whole, 1 << 16), [0, 0]
        step(False)
        cv = torch.tensor([ops.check_acc[0], ops.check_acc[1] & 0x7FFFFFFFFFFFFFFF, ops.check_acc[1] >> 63], dtype=torch.int64, device=comm)
        cparts = [torch.empty_like(cv) for _ in range(world)]
        dist.all_gather(cparts, cv)
        chk = (sum(int(p[0]) for p in cparts), sum(int(p[1]) + (int(p[2]) << 63) for p in cparts) % (1 << 64))
        d_whole.free()
    v = torch.tensor([agg["suffixes"], int(agg["kernel_ms"] * 1000)], dtype=torch.int64, device=comm)
    parts = [torch.empty_like(v) for _ in range(world)]
    dist.all_gather(parts, v)
    if rank != 0:
        return None
    K = args.steps
    suff = sum(int(p[0]) for p in parts) / K
    kern0 = agg["kernel_ms"] / K / 1e3                       # rank 0 streams the most chunks: its kernel time bounds the rounds
    suff0 = agg["suffixes"] / K
    achieved = A_STREAM * suff0 / kern0 / 1e9 if kern0 else 0.0
    return {
        "metric": "input MB/s, hot path of the block-per-GPU schedule (local pass A + BWT merge + rank, systolic gap-stream rounds with one point-to-point gt exchange each, gap split, output-range partitioned merge)",
        "value": n * K / 1e6 / elapsed, "unit": "MB/s", "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / K,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32/u40 integer", "data": "synthetic",
        "config": {"workload": f"configs[3] shape: {n / 2 ** 30:.2f} GiB {args.text} text (seeded, block q = generator seed 1000 + q), {world} blocks of {block / 2 ** 30:.2f} GiB sharded one per GPU "
                               f"({'40-bit partial SAs in two planes' if wide else '32-bit partial SAs'}), {world - 1} rounds, one exchange of the gt slices per round (a slice goes to the left neighbour and, if that one is helped, its helper: RCCL send/recv)"
                               + (f", helper ranks (rank N-1-g streams half of rank g's chunks once its own are done; one BWT hand-over and one gap reduce per pair)" if ops.helpers and world >= 3 else "")
                               + f", merge partitioned by output range in {rounds} sub-ranges per rank",
                   "text_bytes": n, "blocks": world, "block_bytes": block, "helper_ranks": bool(ops.helpers and world >= 3),
                   "resident_per_rank": "own block of text + look-ahead, two chunk buffers, BWT + gt bits of the halves, rank structure, gap array; partial SAs in pinned host memory",
                   "untimed_preparation": "half-block suffix sorts (device, pieces of 2^31 merged with the hot path), the start ranks found by string search while the partial SAs are on the device",
                   "collectives_per_step": f"{world - 1} rounds of at most two sends + two receives of {BD.slice_words(bounds) * 4} B per rank, {2 * world - 1} broadcasts, {rounds} all-to-alls"},
        "gap_stream_suffixes_per_s": suff / (elapsed / K), "streamed_suffixes_per_step": suff,
        "roofline": {"bound": "hbm", "kernel": "stream_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "traffic_source": "rank 0's launches; no PMC pass for the multi-GPU run", "algorithmic_bytes_per_suffix": A_STREAM,
                     "suffixes_per_launch": suff0 / max(1, agg["launches"] / K), "avg_launch_ms": 1e3 * kern0 / max(1, agg["launches"] / K)},
        "property_check": ({"sampled_adjacent_pairs_out_of_order": chk[0], "sum_matches_permutation": chk[1] == (n * (n - 1) // 2) % (1 << 64)} if chk is not None else
                           {"note": "no rank holds the whole text: the schedule's own invariants (gap sums, chain hand-over ranks, ones of every merge bitvector, cursor totals) are checked in every step; "
                                    "output parity of this code path against the oracle's suffix array: tests/test_scale_gpu.py::test_block_per_gpu_schedule_real_kernels and the gloo tests"}),
    }


def main():
    args = parse()
    # stdout carries ONE line, the result: whatever the libraries below print there (RCCL announces its version on stdout)
    # goes to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(res):
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(res) + "\n").encode())
    ctx = setup(args)
    rank, world, local, dist, torch, np, api, extras, L, log = ctx
    cfg = args.config or (2 if world == 1 else 3)     # N > 1: the block-per-GPU schedule; --config 1: the tail-sharded single block
    if cfg == 2 and world == 1:
        args.text = args.text or "english"
        n = int((args.gib or 32.0) * (1 << 30)) // 4096 * 4096
        block = int((args.block_gib * (1 << 30)) if args.block_gib else n // 8)
        res = config2(args, ctx, n, block)
        if not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(api, extras, args.cpu_sample_mib, log, {"english": extras.MODE_ENGLISH, "dna": extras.MODE_DNA}.get(args.text))
            except Exception as e:  # the baseline must not take the bench down
                res["cpu_baseline"] = {"value": None, "unit": "MB/s", "cores": 0, "kind": "reference", "sample": f"failed: {e!r}"}
        if not args.no_secondary:
            try:
                res["end_to_end_cli"] = end_to_end_cli(args.e2e_mib, log, api, extras, english=args.text == "english")
                res["end_to_end_cli_device_sort"] = end_to_end_cli(2048, log, api, extras, english=args.text == "english", device_sort=True)
            except Exception as e:
                res["end_to_end_cli"] = {"value": None, "unit": "MB/s", "sample": f"failed: {e!r}"}
            try:
                c1 = config1(args, ctx, 4.0, 3, 1, "bytes", False)
                res["configs1_step"] = {k: c1[k] for k in ("value", "unit", "ms_per_step", "phase_ms", "gap_stream_kernel_suffixes_per_s", "roofline", "property_check")}
                res["configs1_step"]["workload"] = c1["config"]["workload"]
            except Exception as e:
                res["configs1_step"] = {"value": None, "note": f"failed: {e!r}"}
        emit(res)
    elif cfg == 3 or (world > 1 and cfg != 1):
        args.text = args.text or "dna"                       # BASELINE configs[3]: DNA, 16 GiB blocks, one per GPU
        block = int((args.block_gib or 16.0) * (1 << 30)) // 4096 * 4096
        res = config_blocks(args, ctx, block)
        if rank == 0:
            emit(res)
    else:
        args.text = args.text or "bytes"
        res = config1(args, ctx, args.gib or 4.0, args.steps, args.warmup, args.text, True)
        if rank == 0:
            emit(res)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
