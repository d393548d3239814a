#!/bin/bash
# HBM traffic of the configs[2] bench step from the PMC counters (separate runs per counter, never combined with
# traces), plus the calibration of FETCH_SIZE on membench's random 16-byte loads:
#   bash tools/profile2_pmc.sh r02          (run from the repo root through gpurun)
set -o pipefail
R=${1:-r02}
ROOT=$(pwd)
O=$ROOT/gpurun_out/pmc_$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/membench_fetch -- $ROOT/tools/membench 16 4 > $O/membench.txt 2> $O/membench.err || { tail -3 $O/membench.err; exit 1; }
echo "[pmc] membench FETCH_SIZE done"
timeout -k 10 1000 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $ROOT/bench.py --no-cpu-baseline --no-secondary --no-output-d2h --steps 1 --warmup 0 > $O/fetch.json 2> $O/fetch.err || { tail -3 $O/fetch.err; exit 1; }
echo "[pmc] FETCH_SIZE pass done"
timeout -k 10 1000 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $ROOT/bench.py --no-cpu-baseline --no-secondary --no-output-d2h --steps 1 --warmup 0 > $O/write.json 2> $O/write.err || { tail -3 $O/write.err; exit 1; }
echo "[pmc] WRITE_SIZE pass done"
cd $ROOT
python3 tools/pmc_summary.py membench $O/membench_fetch > $O/${R}_pmc_membench.csv
python3 tools/pmc_summary.py configs2 $O/fetch $O/write > $O/${R}_pmc_configs2.csv
rm -rf $O/fetch $O/write $O/membench_fetch      # raw per-dispatch rows: too large to travel back
head -12 $O/${R}_pmc_membench.csv; grep -E "stream_kernel|p2_|hist_items|merge_kernel|sm_fill" $O/${R}_pmc_configs2.csv | head -30
