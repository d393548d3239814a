#!/bin/bash
# counters of the two general merge kernels on a reduced configs[2] step (8 GiB, 16 half-blocks); run from the repo root
# through gpurun.  Writes gpurun_out/r03/merge_kernel_pmc.txt (sums over all dispatches of the step, and the dispatch count).
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/pmc_merge; mkdir -p $O $ROOT/gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
for K in levels cursor; do
  if [ $K = levels ]; then export PSG_MERGE_CHAIN=1; else unset PSG_MERGE_CHAIN; fi
  for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" ; do
    tag=${K}_$(echo $C | tr ' ' '_' | cut -c1-40)
    timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $O/$tag -- python3 $ROOT/bench.py --gib 8 --no-cpu-baseline --no-secondary --no-output-d2h --steps 1 --warmup 0 --psa-hbm-gib 64 > $O/$tag.json 2> $O/$tag.err || { tail -3 $O/$tag.err; }
  done
done
cd $ROOT
python3 - > $ROOT/gpurun_out/r03/merge_kernel_pmc.txt <<P
import csv, glob, collections, os
print("# rocprofv3 --pmc, one pass per counter group, bench.py --gib 8 --steps 1 --warmup 0 (16 half-blocks, 8 Gi outputs per step,")
print("# two untimed + one timed step => 3 x 128 slices of 64 Mi outputs); sums over all dispatches, (sum, dispatches)")
for f in sorted(glob.glob("$O/*/*/*counter_collection.csv")):
    which = f[len("$O/"):].split("_")[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        if "merge_kernel" in k or "merge_tile_cursor" in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
    for k in acc:
        print(which, k, {c: (v, calls[(k, c)]) for c, v in acc[k].items()})
P
rm -rf $O
cat $ROOT/gpurun_out/r03/merge_kernel_pmc.txt
