#!/bin/bash
# round 4, GPU batch 2: hub chain first contact, then tests, then timings
set -x
cd "$(dirname "$0")/.."
O=gpurun_out/r04b2
mkdir -p $O
export PYTHONPATH=$PWD:$PWD/dgsparse-lib_amd
timeout 300 python bench/hub_smoke.py > $O/hub_smoke.txt 2>&1; echo "hub_smoke rc=$?" >> $O/hub_smoke.txt
tail -n 30 $O/hub_smoke.txt
if grep -q "rc=124" $O/hub_smoke.txt; then echo "hub smoke hung: stopping"; exit 1; fi
# (the workgroups of the experimental DGS_HUB_XCD mode wait for one another: its first contact comes after the numbers that matter)
timeout 900 python -m pytest tests/test_gpu_strict.py tests/test_gpu_plan.py -x -q -m gpu -k "not slice_by_slice" > $O/pytest_strict_plan.txt 2>&1
tail -n 5 $O/pytest_strict_plan.txt
timeout 900 python bench.py --no-dense > $O/bench_line.json 2> $O/bench_err.txt
timeout 600 python bench/strict_parts.py 64 > $O/strict_parts.txt 2>&1
timeout 600 python bench/strict_time.py > $O/strict_time.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_strict.py -x -q -m gpu -k "slice_by_slice" > $O/pytest_hub_xcd.txt 2>&1
tail -n 5 $O/pytest_hub_xcd.txt
timeout 900 python bench/nocut_probe.py 64 > $O/nocut_probe.txt 2>&1
DGS_HUB_XCD=1 timeout 900 python bench.py --no-dense > $O/bench_line_hub_xcd.json 2> $O/bench_err_hub_xcd.txt
DGS_HUB_XCD=2 timeout 900 python bench.py --no-dense > $O/bench_line_hub_xcd2.json 2> $O/bench_err_hub_xcd2.txt
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
tail -n 12 $O/strict_parts.txt $O/nocut_probe.txt $O/pytest_all.txt
