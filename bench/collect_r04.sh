#!/bin/bash
# Round-4 evidence on the final code: bench line (+ driver's invocation, plan-free, strict), kernel stats and counter passes of
# the headline with the hub chains on and off, strict parts / times, per-config table, the reference's mtx benchmark.
#   bash bench/collect_r04.sh        (GPU box; results under gpurun_out/r04/)
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04; mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/dgsparse-lib_amd
timeout 900 python bench.py --sweep > $OUT/bench_line.json 2> $OUT/bench_line.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-dense > $OUT/bench_line_steps20_warmup5.json 2>/dev/null
DGS_HUB_CHAIN=0 timeout 600 python bench.py --no-dense > $OUT/bench_line_nohub.json 2>/dev/null
timeout 600 python bench.py --plan 0 --no-dense --no-protocol > $OUT/bench_line_noplan.json 2>/dev/null
timeout 600 python bench.py --strict fma --no-dense --no-protocol > $OUT/bench_line_strict_fma.json 2>/dev/null
timeout 600 python bench.py --strict nofma --no-dense --no-protocol > $OUT/bench_line_strict_nofma.json 2>/dev/null
timeout 600 python bench/strict_parts.py 64 > $OUT/strict_parts.txt 2>&1
timeout 600 python bench/strict_time.py > $OUT/strict_time.txt 2>&1
timeout 600 python bench/nocut_probe.py 64 > $OUT/nocut_probe.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_bench -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-dense --no-protocol > /dev/null 2>&1
cp $(ls $OUT/kstats_bench/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_bench_feat64_sum_plan.csv; rm -rf $OUT/kstats_bench
DGS_HUB_CHAIN=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_nohub -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-dense --no-protocol > /dev/null 2>&1
cp $(ls $OUT/kstats_nohub/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_bench_feat64_sum_plan_nohub.csv; rm -rf $OUT/kstats_nohub
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_strict -- python bench.py --strict fma --steps 100 --warmup 10 --no-cpu-baseline --no-dense --no-protocol > /dev/null 2>&1
cp $(ls $OUT/kstats_strict/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_bench_feat64_sum_strict.csv; rm -rf $OUT/kstats_strict
timeout 400 bash bench/prof_pmc.sh $OUT/pmc_plan --no-dense --no-protocol > /dev/null 2>&1
DGS_HUB_CHAIN=0 timeout 400 bash bench/prof_pmc.sh $OUT/pmc_plan_nohub --no-dense --no-protocol > /dev/null 2>&1
timeout 400 bash bench/prof_pmc.sh $OUT/pmc_strict --no-dense --no-protocol --strict fma > /dev/null 2>&1
timeout 300 python bench/bench_configs.py > $OUT/configs.jsonl 2>/dev/null
timeout 900 python bench/mtx_bench.py --out $OUT/r04_mtx > $OUT/mtx_bench.txt 2>&1
ls -la $OUT
# the locality machinery on structured graphs (community-like columns) next to the random ones
timeout 600 python bench/bench_configs.py --cols local > $OUT/configs_local.jsonl 2>/dev/null
timeout 600 python bench.py --cols local --no-dense --no-protocol > $OUT/bench_line_cols_local.json 2>/dev/null
DGS_SDDMM_FUSED=1 timeout 300 python bench/bench_configs.py --cols local --only SDDMM > $OUT/configs_local_sddmm_fused.jsonl 2>/dev/null
DGS_SDDMM_FUSED=0 timeout 300 python bench/bench_configs.py --cols local --only SDDMM > $OUT/configs_local_sddmm_nnzbal.jsonl 2>/dev/null
ls -la $OUT
