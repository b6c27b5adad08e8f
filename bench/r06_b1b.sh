#!/bin/bash
# round 6, GPU batch 1b: kernel stats + PMC passes of the default (hub chains, combine launch) with the hub chains off and the fold
# on beside it, the strict / no-cut probes, the reference's mtx benchmark, every BASELINE config, the multi-GPU min's parts.
set -x
cd "$(dirname "$0")/.."
O=gpurun_out/r06b1
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/dgsparse-lib_amd
for cfg in "hub_combine:" "hub_fold:DGS_FOLD=1" "nohub_combine:DGS_HUB_CHAIN=0" "nohub_fold:DGS_HUB_CHAIN=0 DGS_FOLD=1"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$tag -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-dense --no-protocol > /dev/null 2>&1
  cp $(ls $O/ks_$tag/*/*kernel_stats.csv | head -1) $O/kernel_stats_bench_feat64_sum_plan_$tag.csv; rm -rf $O/ks_$tag
done
timeout 400 bash bench/prof_pmc.sh $O/pmc_plan_hub_combine --no-dense --no-protocol > /dev/null 2>&1
DGS_HUB_CHAIN=0 timeout 400 bash bench/prof_pmc.sh $O/pmc_plan_nohub_combine --no-dense --no-protocol > /dev/null 2>&1
DGS_FOLD=1 timeout 400 bash bench/prof_pmc.sh $O/pmc_plan_hub_fold --no-dense --no-protocol > /dev/null 2>&1
# VERDICT r5 #4c: what a lower hub threshold costs (headline; C2 / C3 through bench_configs below)
for th in 1024 2048 4096; do
  DGS_HUB_CHAIN=$th timeout 300 python bench.py --no-dense --no-protocol --no-cpu-baseline > $O/bench_line_hub$th.json 2>/dev/null
  DGS_HUB_CHAIN=$th timeout 300 python bench/bench_configs.py > $O/configs_hub$th.jsonl 2>/dev/null
done
timeout 600 python bench/strict_parts.py 64 > $O/strict_parts.txt 2>&1
timeout 600 python bench/strict_time.py > $O/strict_time.txt 2>&1
timeout 900 python bench/nocut_probe.py 64 > $O/nocut_probe.txt 2>&1
make -C examples > /dev/null 2>&1
timeout 900 python bench/mtx_bench.py --out $O/r06_mtx > $O/mtx_bench.txt 2>&1
timeout 300 python bench/bench_configs.py > $O/configs.jsonl 2>/dev/null
tail -n 12 $O/strict_parts.txt $O/nocut_probe.txt
timeout 300 python bench/dist_min_parts.py > $O/dist_min_parts.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 experiments/lds_dma_gather.cpp -o /tmp/ldg > /dev/null 2>&1 && timeout 400 /tmp/ldg part1 > $O/lds_dma_gather_part1.txt 2>&1
tail -n 12 $O/lds_dma_gather_part1.txt
ls -la $O; tail -n 6 $O/dist_min_parts.txt
