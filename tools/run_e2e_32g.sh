#!/bin/bash
# one default-placement construct_sa run on 32 GiB of English-like text (configs[2]'s size), log kept:
#   bash tools/run_e2e_32g.sh r03        (run from the repo root through gpurun)
R=${1:-r03}; ROOT=$(pwd); O=$ROOT/gpurun_out/$R; mkdir -p $O
# (the box's /tmp holds 79 GB: the 160 GiB .sa5 goes through the sink -- merge, D2H -- and is dropped instead of written)
python3 tools/e2e_one.py 32768 english -v --check=4096 --discard-output "${@:2}" > $O/e2e_32g.log 2>&1
grep -E "^Input|^RAM|^Max block|Device|bound to|Text on the device|Process block|since start|sufsort|Stream \(|Summary|elapsed|speed|rc=|In-HBM|batched|merge \+ write|slices|device memory|check:|allocator" $O/e2e_32g.log > $O/${R}_construct_sa_32768mib_default.txt
rm -f /tmp/e2e_english_32768.bin /tmp/e2e_english_32768.bin.sa5
tail -30 $O/${R}_construct_sa_32768mib_default.txt
