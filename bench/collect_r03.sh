#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/r03/ (copy what should be judged into profiles/).
#   bash bench/collect_r03.sh
set -u
OUT=gpurun_out/r03; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --sweep > $OUT/bench_line.json 2> $OUT/bench_line.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_line_steps20_warmup5.json 2>/dev/null
python bench.py --strict fma --no-dense > $OUT/bench_line_strict_fma.json 2>/dev/null
python bench.py --strict nofma --no-dense --no-protocol > $OUT/bench_line_strict_nofma.json 2>/dev/null
python bench.py --plan 0 --no-dense --no-cpu-baseline --no-protocol > $OUT/bench_line_noplan.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_bench -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-dense --no-protocol > /dev/null 2>&1
cp $(ls $OUT/kstats_bench/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_bench_feat64_sum_plan.csv; rm -rf $OUT/kstats_bench
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_strict -- python bench.py --strict fma --steps 100 --warmup 10 --no-cpu-baseline --no-dense --no-protocol > /dev/null 2>&1
cp $(ls $OUT/kstats_strict/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_bench_feat64_sum_strict.csv; rm -rf $OUT/kstats_strict
bash bench/prof_pmc.sh $OUT/pmc_plan --no-dense --no-protocol > /dev/null 2>&1
bash bench/prof_pmc.sh $OUT/pmc_strict --no-dense --no-protocol --strict fma > /dev/null 2>&1
python bench/bench_configs.py > $OUT/configs.jsonl 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_cfg -- python bench/bench_configs.py --quick > /dev/null 2>&1
cp $(ls $OUT/kstats_cfg/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_configs.csv; rm -rf $OUT/kstats_cfg
bash bench/collect_pmc_configs.sh $OUT c3sum c3max c3sddmm c4sddmm c2sum ns_sddmm > /dev/null 2>&1
python bench/pmc_table.py $OUT/pmc_summary_*.txt > $OUT/pmc_table.txt
python bench/plan_lifecycle.py > $OUT/plan_lifecycle.json 2>/dev/null
for g in "synth1m 64" "synth1m 128" "arxiv 64" "reddit 128"; do python bench/strict_time.py $g 2>/dev/null | tail -6; done > $OUT/strict_time.txt
python bench/strict_parts.py > $OUT/strict_parts.txt 2>/dev/null
python experiments/home_slice_permute.py > $OUT/home_slice_permute.txt 2>/dev/null
python bench/bench_spmm_time.py --datasets cora citeseer pubmed ppi0 --feats 32 64 128 --json $OUT/spmm_time_grid_small.json > /dev/null 2>&1
python bench/bench_spmm_time.py --datasets reddit --feats 32 64 128 --json $OUT/spmm_time_grid_reddit.json > /dev/null 2>&1
ls -la $OUT
