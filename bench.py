#!/usr/bin/env python3
"""bench.py -- hot-path throughput of the MI355X-native pSAscan path.

Workload (BASELINE.json configs[1], scaled by --gib): one text block of uniform random bytes
0..254 cut into two half-blocks (last-block schedule of the reference, partial_sufsort.hpp:
86-93,418-429).  One "step" = one pass of the hot path over it with all inputs resident in HBM:
    rank build over the left half's BWT            (new rank4n<>,          partial_sufsort.hpp:403)
    stream the right half through it -> gap array  (compute_gap<T>,        :412-414)
    gap array -> merge bitvector                   (convert_to_bitvector / save_to_file, :422,441)
    merge the two partial SAs -> 5n bytes of .sa5  (merge<T>,              psascan.hpp:120-124)
The host suffix sort of the half-blocks is NOT part of the hot path (north_star: stays on host
cores); the bench prepares the partial SAs on the device before the timed region.

N > 1 (strong scaling of the same job): every rank holds the inputs; the tail is cut into N
ranges (the reference's own parallel axis, compute_gap.hpp:68-69), each rank streams its range
into a rank log; the logs are split by owner of the gap slice and exchanged with ONE RCCL
all-to-all (4 B per streamed suffix), every rank counts its slice and marks its bits, the bit
arrays are summed with one all-reduce (n/8 bytes), the gt bits are all-gathered, and each rank
merges 1/N of the output.

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


def end_to_end_cli(sample_mib, log):
    """The whole construct_sa program (host/construct_sa: read the file, host SA-IS of the half-blocks on the host
    threads, the device passes, merge, write the .sa5) on a bounded sample, as a child process.  Reported next to
    the hot-path metric; it is bound by the host sorter, not by the GPU."""
    import subprocess
    import tempfile
    import numpy as np
    cli = os.path.join(ROOT, "host", "construct_sa")
    if not os.path.exists(cli):
        return None
    threads = min(16, os.cpu_count() or 1)
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "sample.bin")
        np.random.default_rng(11).integers(0, 255, sample_mib << 20, dtype=np.uint8).tofile(f)
        t0 = time.time()
        r = subprocess.run([cli, "-m", "8G", "--block-size", str(32 << 20), f], capture_output=True, text=True,
                           env=dict(os.environ, OMP_NUM_THREADS=str(threads)), timeout=600)
        wall = time.time() - t0
        ok = r.returncode == 0 and os.path.getsize(f + ".sa5") == 5 * (sample_mib << 20)
    if not ok:
        log("construct_sa failed:", r.stderr[-300:])
        return {"value": None, "unit": "MB/s", "sample": "failed"}
    return {"value": (sample_mib << 20) / 1e6 / wall, "unit": "MB/s", "host_threads": threads, "seconds": round(wall, 2),
            "sample": f"{sample_mib} MiB uniform bytes 0..254 from a file, 32 MiB blocks, .sa5 written to a file; wall time of the child process"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gib", type=float, default=4.0, help="text size in GiB (configs[1] = 4)")
    ap.add_argument("--max-chains", type=int, default=0)
    ap.add_argument("--rank-block", type=int, default=0, help="data bytes per rank block (0=auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=int, default=512)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--text", choices=["bytes", "dna"], default="bytes", help="bytes = configs[1] (the metric's workload); dna = sigma 4 (configs[3]'s alphabet) at the same size, for DESIGN.md")
    return ap.parse_args()


def cpu_baseline(api, extras, sample_mib, log):
    """The reference's own compute_gap / convert_to_bitvector / merge (oracle/_ref, built from
    /root/reference where it lies) timed on this box's host cores on a bounded sample of the
    same workload.  Falls back to the C restatement (kind "port") when oracle/_ref is absent."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc                                      # checker / baseline only
    import shutil
    n = sample_mib << 20
    mid = n // 2
    d_text = extras.gen_text(n, extras.MODE_BYTES255, 0, seed=99)
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
                scratch = api.zeros(4 * (mid + 2))
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
    return {"value": n / 1e6 / total, "unit": "MB/s", "cores": cores, "kind": kind,
            "sample": f"{sample_mib} MiB uniform bytes 0..254, two {sample_mib // 2} MiB half-blocks: rank build + stream + gap->bitvector + merge, seconds={total:.2f}",
            **detail}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
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
    if world > 1:
        # everything on ONE explicit stream so RCCL collectives and our kernels are ordered.  (torch's default
        # stream has handle 0, which psg_set_stream takes as "create your own": make a real stream current.)
        shared_stream = torch.cuda.Stream()
        torch.cuda.set_stream(shared_stream)
        assert shared_stream.cuda_stream != 0
        L.psg_set_stream(C.c_void_p(shared_stream.cuda_stream))

    def log(*a):
        if rank == 0:
            print("[bench]", *a, file=sys.stderr, flush=True)

    n = int(args.gib * (1 << 30)) // 128 * 128
    mid = n // 2
    ls, rs = mid, n - mid
    t0 = time.time()
    d_text = extras.gen_text(n, extras.MODE_DNA if args.text == "dna" else extras.MODE_BYTES255, 0, seed=2)
    Rh = extras.sort_halfblock(d_text, n, mid, n)
    Lh = extras.sort_halfblock(d_text, n, 0, mid)
    api.sync()
    log(f"prepared {n / 2 ** 30:.2f} GiB text + 2 half-block SAs in {time.time() - t0:.1f}s (ties L/R {Lh['tie_groups']}/{Rh['tie_groups']})")
    last_left = int(api.download(d_text, np.uint8, 1, mid - 1)[0])
    Lh["psa_hi"] = Rh["psa_hi"] = None

    # tail range of this rank (64-aligned cut points so gt words do not straddle ranks)
    from psascan_amd import distributed as D      # cut logic shared with the gloo tests
    cuts = D.tail_cuts(mid, n, world)
    tb, te = cuts[rank], cuts[rank + 1]
    out_cuts = D.output_cuts(n, world)
    ob, oe = out_cuts[rank], out_cuts[rank + 1]
    ctx = D.context_len(te, n)           # right context for the start rank of this range
    gt_words = max(cuts[r + 1] - cuts[r] for r in range(world)) // 32 + 4
    gap_words = ls + 2
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

    for _ in range(args.warmup):
        step(False)
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
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
        K = args.steps
        per = {k: v / K for k, v in times.items()}
        stream_suffixes = te - tb
        kernel_s = per["stream_kernel"]
        achieved = A_STREAM * stream_suffixes / kernel_s / 1e9 if kernel_s > 0 else 0.0
        res = {
            "metric": "input MB/s, hot path (rank build + gap-stream + gap->bitvector + merge to .sa5), inputs resident in HBM",
            "value": n * K / 1e6 / elapsed, "unit": "MB/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8/u32/u40 integer", "data": "synthetic",
            "config": {"workload": (f"configs[1]: {n / 2 ** 30:.2f} GiB uniform random bytes 0..254 (sigma=255; byte 255 is reserved by the reference)" if args.text == "bytes" else f"{n / 2 ** 30:.2f} GiB uniform random DNA (sigma=4), not the metric's workload") + f", one block = two {mid / 2 ** 30:.2f} GiB half-blocks, single pass A + merge",
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
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(api, extras, args.cpu_sample_mib, log)
            except Exception as e:  # the baseline must not take the bench down
                res["cpu_baseline"] = {"value": None, "unit": "MB/s", "cores": 0, "kind": "reference", "sample": f"failed: {e!r}"}
            try:
                res["end_to_end_cli"] = end_to_end_cli(1024, log)
            except Exception as e:
                res["end_to_end_cli"] = {"value": None, "unit": "MB/s", "sample": f"failed: {e!r}"}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
