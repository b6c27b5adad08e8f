#!/bin/bash
# Re-collects, on the final code of round 3, the evidence the late kernel changes touch (combine, unit path, panel
# shuffles): the bench line, the per-config table, kernel stats and the counter passes of the changed kernels.
#   bash bench/collect_r03_late.sh        (GPU box; results under gpurun_out/r03late/)
set -u
OUT=gpurun_out/r03late; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py --sweep > $OUT/bench_line.json 2> $OUT/bench_line.err
timeout 300 python bench/bench_configs.py > $OUT/configs.jsonl 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_cfg -- python bench/bench_configs.py --quick > /dev/null 2>&1
cp $(ls $OUT/kstats_cfg/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_configs.csv; rm -rf $OUT/kstats_cfg
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_bench -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-dense --no-protocol > /dev/null 2>&1
cp $(ls $OUT/kstats_bench/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_bench_feat64_sum_plan.csv; rm -rf $OUT/kstats_bench
timeout 400 bash bench/prof_pmc.sh $OUT/pmc_plan --no-dense --no-protocol > /dev/null 2>&1
timeout 900 bash bench/collect_pmc_configs.sh $OUT c3max c3sddmm > /dev/null 2>&1
python bench/pmc_table.py $OUT/pmc_plan/summary.txt $OUT/pmc_summary_*.txt > $OUT/pmc_table.txt
ls -la $OUT
