#!/bin/bash
# PMC passes for the headline bench (separate passes: TCC has 4 slots, FETCH_SIZE=3, WRITE_SIZE=2; never mixed
# with --kernel-trace/--stats).  Usage on the GPU box: bash bench/prof_pmc.sh OUTDIR [bench.py args...]
# PMC_CMD="python bench/panel_probe.py" bash bench/prof_pmc.sh OUTDIR   profiles another command instead of bench.py
set -u
OUT=$1; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD=${PMC_CMD:-"python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline"}
run() { tag=$1; shift; (cd "$REPO" && rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$tag" -- $CMD $ARGS > "$OUT/$tag.log" 2>&1); }
ARGS="$*"
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run dram TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT
python "$REPO/bench/pmc_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
if [ "${PMC_KEEP_RAW:-0}" != 1 ]; then for t in fetch write tcc dram sq; do rm -rf "$OUT/$t"; done; fi  # raw csvs are ~10 MB per pass
cat "$OUT/summary.txt"
