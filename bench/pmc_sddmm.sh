#!/bin/bash
# FETCH / WRITE / TCC passes for one SDDMM configuration (GPU box): bash bench/pmc_sddmm.sh OUTDIR "C4"|"synth-1M SDDMM"
OUT=$1; CFG=$2; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  bash bench/pmc_one.sh $OUT/$tag "$c" python bench/bench_configs.py --quick --only "$CFG" > /dev/null 2>&1
done
cat $OUT/*/summary.txt | grep -v "^rocprim\|at::" 
