#!/bin/bash
# Kernel trace of the configs[2] bench step on a GPU box (run from the repo root through gpurun):
#   bash tools/profile2.sh r02 [extra bench args]
# Counters (PMC) are separate runs: tools/profile2_pmc.sh.  Traces and counters are never combined.
set -o pipefail
R=${1:-r02}; shift
ROOT=$(pwd)
O=$ROOT/gpurun_out/prof_$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 1000 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-secondary --no-output-d2h --steps 1 --warmup 1 "$@" > $O/trace_bench.json 2> $O/trace.err || { tail -5 $O/trace.err; exit 1; }
cd $ROOT; find $O -name "*kernel_trace.csv" -delete
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/${R}_configs2_kernel_stats.csv
head -30 $O/${R}_configs2_kernel_stats.csv
