#!/bin/bash
# One rocprofv3 --pmc pass over an arbitrary command, summarised per dgs:: kernel.
#   bash bench/pmc_one.sh OUTDIR "COUNTER COUNTER ..." command args...
set -u
OUT=$1; CTRS=$2; shift 2
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
(cd "$REPO" && rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/run" -- "$@" > "$OUT/run.log" 2>&1)
python "$REPO/bench/pmc_summary.py" "$OUT" | tee "$OUT/summary.txt"
if [ "${PMC_KEEP_RAW:-0}" != 1 ]; then rm -rf "$OUT/run"; fi
