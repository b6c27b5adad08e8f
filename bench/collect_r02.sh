#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/r02/ (copy what should be judged into profiles/).
#   bash bench/collect_r02.sh
set -u
OUT=gpurun_out/r02; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --sweep > $OUT/bench_line.json 2> $OUT/bench_line.err
python bench.py --plan 0 --no-dense --no-cpu-baseline --no-protocol > $OUT/bench_line_noplan.json 2>/dev/null
python bench.py --cols uniform --no-dense --no-cpu-baseline --no-protocol > $OUT/bench_line_uniform.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_bench -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-dense --no-protocol > /dev/null 2>&1
python bench/kstats.py $OUT/kstats_bench 12 > $OUT/kernel_stats_bench_feat64_sum_plan.txt
cp $(ls $OUT/kstats_bench/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_bench_feat64_sum_plan.csv; rm -rf $OUT/kstats_bench
bash bench/prof_pmc.sh $OUT/pmc_plan --no-dense --no-protocol > /dev/null 2>&1
bash bench/prof_pmc.sh $OUT/pmc_noplan --no-dense --no-protocol --plan 0 > /dev/null 2>&1
python bench/bench_configs.py > $OUT/configs.jsonl 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_cfg -- python bench/bench_configs.py --quick > /dev/null 2>&1
python bench/kstats.py $OUT/kstats_cfg 40 > $OUT/kernel_stats_configs.txt
cp $(ls $OUT/kstats_cfg/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_configs.csv; rm -rf $OUT/kstats_cfg
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  bash bench/pmc_one.sh $OUT/pmc_sddmm_$tag "$c" python bench/bench_configs.py --quick --only SDDMM > /dev/null 2>&1
done
cat $OUT/pmc_sddmm_*/summary.txt > $OUT/pmc_summary_sddmm.txt
python bench/bench_spmm_time.py --datasets cora citeseer pubmed ppi0 --feats 32 64 128 --json $OUT/spmm_time_grid_small.json > /dev/null 2>&1
python bench/bench_spmm_time.py --datasets reddit --feats 32 64 128 --json $OUT/spmm_time_grid_reddit.json > /dev/null 2>&1
hipcc --offload-arch=gfx950 -O3 experiments/mfma_hub.cpp -o /tmp/mh 2>/dev/null && /tmp/mh > $OUT/mfma_hub.txt 2>&1
hipcc --offload-arch=gfx950 -O3 experiments/gather_sizes.cpp -o /tmp/gs 2>/dev/null && /tmp/gs > $OUT/gather_sizes.txt 2>&1
ls -la $OUT
