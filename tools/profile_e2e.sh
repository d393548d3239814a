#!/bin/bash
# kernel trace of one default-placement construct_sa run (run from the repo root through gpurun):
#   bash tools/profile_e2e.sh r03 4096        (round tag, MiB of English-like text)
set -o pipefail
R=${1:-r03}; MIB=${2:-4096}
ROOT=$(pwd)
O=$ROOT/gpurun_out/prof_e2e_$R
mkdir -p $O
python3 tools/e2e_one.py $MIB english -v --check=1024 > $O/plain.log 2>&1 || { tail -5 $O/plain.log; exit 1; }
grep -E "since start|sufsort|Stream \(|Summary|elapsed|speed|rc=|In-HBM|batched|merge \+ write|slices|device memory|check:|allocator" $O/plain.log > $O/${R}_construct_sa_${MIB}mib_default.txt
rm -f /tmp/e2e_english_${MIB}.bin.sa5
cd /tmp && export TMPDIR=/tmp
OMP_NUM_THREADS=16 PSASCAN_NORMAL_EXIT=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $ROOT/host/construct_sa -v --discard-output /tmp/e2e_english_${MIB}.bin > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
cd $ROOT; find $O -name "*kernel_trace.csv" -delete
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/${R}_construct_sa_${MIB}mib_kernel_stats.csv
head -40 $O/${R}_construct_sa_${MIB}mib_kernel_stats.csv
rm -f /tmp/e2e_english_${MIB}.bin
