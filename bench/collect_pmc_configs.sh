#!/bin/bash
# FETCH / WRITE / TCC / SQ passes (separate rocprofv3 --pmc runs, never combined with tracing) for the BASELINE.json
# configurations other than the headline, one summary file per configuration.  GPU box:
#   bash bench/collect_pmc_configs.sh OUTDIR [TAG ...]      TAG in: c3sum c3max c3sddmm c4sddmm c2sum ns_sddmm
set -u
OUT=$1; shift
TAGS=${*:-"c3sum c3max c3sddmm c4sddmm c2sum"}
mkdir -p $OUT
declare -A CFG=( [c3sum]="C3 reddit-shaped SpMM-sum" [c3max]="C3 reddit-shaped SpMM-max" [c3sddmm]="C3 reddit-shaped SDDMM"
                 [c4sddmm]="C4 products-shaped SDDMM" [c2sum]="C2 arxiv-shaped SpMM-sum" [ns_sddmm]="NS synth-1M SDDMM" )
for tag in $TAGS; do
  cfg=${CFG[$tag]}
  : > $OUT/pmc_summary_$tag.txt
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VALU"; do
    p=$(echo $c | cut -d' ' -f1)
    echo "## pass: $c   (python bench/bench_configs.py --quick --only \"$cfg\")" >> $OUT/pmc_summary_$tag.txt
    bash bench/pmc_one.sh $OUT/_$tag.$p "$c" python bench/bench_configs.py --quick --only "$cfg" 2>/dev/null \
      | grep -v "^rocprim\|at::" >> $OUT/pmc_summary_$tag.txt
    rm -rf $OUT/_$tag.$p
  done
done
