#!/bin/bash
# round 5, GPU batch 1: first hardware contact of the round-4 hub-chain default, in the order asked:
# hub smoke (stop on hang) -> all -m gpu tests (slice_by_slice last) -> bench line -> kernel stats + PMC, hub chains on / off.
set -x
cd "$(dirname "$0")/.."
O=gpurun_out/r05b1
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/dgsparse-lib_amd
timeout 300 python bench/hub_smoke.py > $O/hub_smoke.txt 2>&1; echo "hub_smoke rc=$?" >> $O/hub_smoke.txt
tail -n 30 $O/hub_smoke.txt
if grep -q "rc=124" $O/hub_smoke.txt; then echo "hub smoke hung: stopping"; exit 1; fi
timeout 1500 python -m pytest tests -q -m gpu -k "not slice_by_slice" > $O/pytest_all.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_all.txt
tail -n 25 $O/pytest_all.txt
timeout 900 python bench.py --sweep > $O/bench_line.json 2> $O/bench_err.txt; echo "bench rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --no-dense > $O/bench_line_steps20_warmup5.json 2>/dev/null
DGS_HUB_CHAIN=0 timeout 600 python bench.py --no-dense > $O/bench_line_nohub.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_bench -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-dense --no-protocol > /dev/null 2>&1
cp $(ls $O/kstats_bench/*/*kernel_stats.csv | head -1) $O/kernel_stats_bench_feat64_sum_plan_hub.csv; rm -rf $O/kstats_bench
DGS_HUB_CHAIN=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_nohub -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-dense --no-protocol > /dev/null 2>&1
cp $(ls $O/kstats_nohub/*/*kernel_stats.csv | head -1) $O/kernel_stats_bench_feat64_sum_plan_nohub.csv; rm -rf $O/kstats_nohub
timeout 400 bash bench/prof_pmc.sh $O/pmc_plan_hub --no-dense --no-protocol > /dev/null 2>&1
DGS_HUB_CHAIN=0 timeout 400 bash bench/prof_pmc.sh $O/pmc_plan_nohub --no-dense --no-protocol > /dev/null 2>&1
timeout 600 python bench/strict_parts.py 64 > $O/strict_parts.txt 2>&1
timeout 600 python bench/strict_time.py > $O/strict_time.txt 2>&1
# the experimental DGS_HUB_XCD modes wait across workgroups: after everything that matters
timeout 300 python -m pytest tests/test_gpu_strict.py -x -q -m gpu -k "slice_by_slice" > $O/pytest_hub_xcd.txt 2>&1; echo "hub_xcd rc=$?" >> $O/pytest_hub_xcd.txt
tail -n 5 $O/pytest_hub_xcd.txt
if ! grep -q "rc=124" $O/pytest_hub_xcd.txt; then
  timeout 900 python bench/nocut_probe.py 64 > $O/nocut_probe.txt 2>&1
  DGS_HUB_XCD=1 timeout 600 python bench.py --no-dense --no-cpu-baseline > $O/bench_line_hub_xcd1.json 2> $O/bench_err_hub_xcd1.txt
  DGS_HUB_XCD=2 timeout 600 python bench.py --no-dense --no-cpu-baseline > $O/bench_line_hub_xcd2.json 2> $O/bench_err_hub_xcd2.txt
fi
timeout 900 python bench/mtx_bench.py --out $O/r05_mtx > $O/mtx_bench.txt 2>&1
timeout 300 python bench/bench_configs.py > $O/configs.jsonl 2>/dev/null
ls -la $O
tail -n 12 $O/strict_parts.txt $O/nocut_probe.txt
head -c 3000 $O/bench_line.json
