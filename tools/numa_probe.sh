#!/bin/bash
# does the placement of construct_sa's threads relative to the GPU's NUMA node matter?  run from the repo root through gpurun
ROOT=$(pwd); O=$ROOT/gpurun_out/r03; mkdir -p $O
{
lscpu | grep -E "Model name|Socket|NUMA|Thread|Core"
for d in /sys/class/drm/card*/device; do echo "$d numa_node=$(cat $d/numa_node 2>/dev/null) $(cat $d/vendor 2>/dev/null)"; done
python3 tools/e2e_one.py 4096 english -v 2>&1 | grep -E "Text on the device|merge \+ write|sink=|elapsed|rc=|\[3\.|waited" | tail -12
N0=$(lscpu | grep "NUMA node0 CPU" | awk '{print $NF}'); N1=$(lscpu | grep "NUMA node1 CPU" | awk '{print $NF}')
for cpus in "$N0" "$N1"; do
  [ -z "$cpus" ] && continue
  rm -f /tmp/e2e_english_4096.bin.sa5
  echo "== taskset -c $cpus"
  OMP_NUM_THREADS=16 taskset -c $cpus host/construct_sa -v /tmp/e2e_english_4096.bin 2>&1 | grep -E "Text on the device|merge \+ write|sink=|elapsed|waited 0\.[1-9]" | tail -8
done
rm -f /tmp/e2e_english_4096.bin /tmp/e2e_english_4096.bin.sa5
} > $O/numa_probe.txt 2>&1
cat $O/numa_probe.txt
