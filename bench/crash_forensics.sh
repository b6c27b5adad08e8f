#!/bin/bash
# GPU-side crash forensics (needs the GPU box): round 3 saw ONE unexplained process death in a
# multi-process fuzz run and has no backtrace of it.  This script tries to get one: N concurrent tests/fuzz_gpu.py campaigns on the
# one GPU with core dumps allowed for these processes (ulimit only - no system setting is touched: a core lands wherever the box's
# own core_pattern puts it, normally the working directory), Python's faulthandler on, and rocgdb's backtrace of every core found.
#   bash bench/crash_forensics.sh [seconds per campaign = 240] [processes = 4]      -> gpurun_out/forensics/
set -x
cd "$(dirname "$0")/.."
O=gpurun_out/forensics
mkdir -p $O
SECS=${1:-240}; NP=${2:-4}
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/dgsparse-lib_amd PYTHONFAULTHANDLER=1
export AMD_LOG_LEVEL=1 DGS_CANARY=1
ulimit -c unlimited
cat /proc/sys/kernel/core_pattern > $O/core_pattern_of_the_box.txt 2>/dev/null
pids=()
for i in $(seq 1 $NP); do
  ( cd $O && timeout $((SECS + 120)) python ../../tests/fuzz_gpu.py $SECS $((900 + i)) > fuzz_$i.txt 2>&1; echo "rc=$?" >> fuzz_$i.txt ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
tail -n 3 $O/fuzz_*.txt
for c in $O/core*; do
  [ -f "$c" ] || continue
  case "$c" in *.txt) continue;; esac
  timeout 300 /opt/rocm/bin/rocgdb -batch -ex "thread apply all bt 25" -ex "info sharedlibrary" "$(which python)" "$c" > "$c.bt.txt" 2>&1
  head -n 80 "$c.bt.txt"
  rm -f "$c"  # cores of a torch process are GBs: keep the backtrace only
done
grep -L "rc=0" $O/fuzz_*.txt > $O/failed_campaigns.txt; cat $O/failed_campaigns.txt
ls -la $O
