#!/bin/bash
# counters of the general merge kernel on a reduced configs[2] step (8 GiB, 16 half-blocks); run from the repo root through gpurun
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/pmc_merge; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" ; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $O/$tag -- python3 $ROOT/bench.py --gib 8 --no-cpu-baseline --no-secondary --no-output-d2h --steps 1 --warmup 0 --psa-hbm-gib 64 > $O/$tag.json 2> $O/$tag.err || { tail -3 $O/$tag.err; }
done
cd $ROOT
python3 - <<P
import csv, glob, collections
for f in sorted(glob.glob("$O/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        if "merge_kernel" in k or "merge_tile_cursor" in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
    for k in acc:
        print(k, {c: (v, calls[(k, c)]) for c, v in acc[k].items()})
P
