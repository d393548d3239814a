#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run from the repo root through gpurun):
#   bash tools/profile.sh r01
# kernel-trace stats and the two PMC passes are separate rocprofv3 runs (counters and traces are never combined).
set -o pipefail
R=${1:-r01}
ROOT=$(pwd)
O=$ROOT/gpurun_out/prof_$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/trace_bench.json 2> $O/trace.err || exit 1
echo "[profile] kernel trace done"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $ROOT/bench.py --no-cpu-baseline --no-check --steps 1 --warmup 0 > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 1
echo "[profile] FETCH_SIZE pass done"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $ROOT/bench.py --no-cpu-baseline --no-check --steps 1 --warmup 0 > $O/pmc_write.json 2> $O/pmc_write.err || exit 1
echo "[profile] WRITE_SIZE pass done"
cd $ROOT
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/${R}_bench_4gib_kernel_stats.csv
python3 tools/pmc_summary.py bench_4gib $O/pmc_fetch $O/pmc_write > $O/${R}_pmc_summary.csv
timeout -k 10 300 ./tools/membench 32 8 > $O/${R}_membench.txt 2>&1
timeout -k 10 600 python3 bench.py > $O/${R}_bench_4gib.json 2> $O/bench.err
tail -c 600 $O/${R}_bench_4gib.json
