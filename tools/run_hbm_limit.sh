#!/bin/bash
# the external-memory schedule under a device budget, run from the repo root through gpurun:
#   bash tools/run_hbm_limit.sh r03 16384 32Gi 4G      (round tag, MiB of English-like text, --hbm-limit, -m)
# 1. tools/compare_modes.py 512 english: six modes of construct_sa, byte-identical .sa5 (one of them --hbm-limit)
# 2. one big run whose working set would not fit the budget: text, gt bits, partial SAs and merge bitvectors in host
#    memory, output verified on the host (--check), .sa5 discarded (the box's disk holds 79 GB)
set -o pipefail
R=${1:-r03}; MIB=${2:-16384}; LIM=${3:-32Gi}; MEM=${4:-4G}
ROOT=$(pwd)
O=$ROOT/gpurun_out/hbm_limit_$R
mkdir -p $O
python3 tools/compare_modes.py 512 english > $O/${R}_compare_modes_512mib.txt 2>&1 || { tail -5 $O/${R}_compare_modes_512mib.txt; exit 1; }
tail -8 $O/${R}_compare_modes_512mib.txt
python3 tools/e2e_one.py $MIB english -v -m $MEM --hbm-limit $LIM --check=256 --discard-output > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
grep -vE "^generated|sufsort|Stream \(|Construct rank|Merge BWTs|Compute gaps|chains=|merge bitvectors to host|since start\]$|^Process block" $O/run.log > $O/${R}_construct_sa_${MIB}mib_hbm_limit_${LIM}.txt
grep -c "merge bitvectors to host" $O/run.log >> $O/${R}_construct_sa_${MIB}mib_hbm_limit_${LIM}.txt
tail -25 $O/${R}_construct_sa_${MIB}mib_hbm_limit_${LIM}.txt
rm -f /tmp/e2e_english_${MIB}.bin
