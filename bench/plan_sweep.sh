#!/bin/bash
# Sweep of the plan's cut parameters on the headline workload (GPU box): bash bench/plan_sweep.sh OUTDIR
OUT=$1; mkdir -p "$OUT"
for u in 16 32 64 128; do for t in 128 256 512; do
  DGS_PLAN_UNIT=$u DGS_PLAN_TSLICE=$t python bench.py --no-dense --no-cpu-baseline --no-protocol --plan 1 --steps 50 > "$OUT/u${u}_t$t.json" 2>/dev/null
  python - "$OUT/u${u}_t$t.json" $u $t <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); print('unit',sys.argv[2],'tslice',sys.argv[3],'ms',j['ms_per_step'],'frac',j['roofline']['frac'])
PY
done; done
