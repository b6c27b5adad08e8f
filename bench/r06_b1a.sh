#!/bin/bash
# round 6, GPU batch 1 (copy of r05_b1a.sh): first hardware contact of the round-4 hub-chain default and of round 5's gates / in-kernel fold, in the
# order VERDICT r4 #1 asks for: device self-test + hub smoke (stop on hang) -> all -m gpu tests -> bench line -> kernel stats +
# PMC with the hub chains on / off and the fold on / off -> the reference's mtx benchmark on HEAD.
set -x
cd "$(dirname "$0")/.."
O=gpurun_out/r06b1
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/dgsparse-lib_amd
timeout 300 python -c "
import torch
from dgsparse import _capi
_capi.ensure_hub_selftest(torch.device('cuda', 0))
print('hub gate', _capi.hub_gate(), 'fold gate', _capi.fold_gate(), 'threshold', _capi.hub_threshold())
" > $O/selftest.txt 2>&1; echo "selftest rc=$?" >> $O/selftest.txt
cat $O/selftest.txt
if grep -q "rc=124" $O/selftest.txt; then echo "self-test hung: stopping"; exit 1; fi
timeout 300 python bench/hub_smoke.py > $O/hub_smoke.txt 2>&1; echo "hub_smoke rc=$?" >> $O/hub_smoke.txt
tail -n 30 $O/hub_smoke.txt
if grep -q "rc=124" $O/hub_smoke.txt; then echo "hub smoke hung: stopping"; exit 1; fi
# the two pieces of evidence the round cannot do without come first: a bench line of the default schedule, and the kernel stats of the same command
timeout 600 python bench.py --no-dense > $O/bench_line_default.json 2> $O/bench_default_err.txt; echo "bench (default) rc=$?"
head -c 2500 $O/bench_line_default.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_first -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-dense --no-protocol > /dev/null 2>&1
cp $(ls $O/ks_first/*/*kernel_stats.csv | head -1) $O/kernel_stats_bench_feat64_sum_plan_default.csv; rm -rf $O/ks_first
head -n 6 $O/kernel_stats_bench_feat64_sum_plan_default.csv
timeout 1800 python -m pytest tests -q -m gpu --durations=40 > $O/pytest_all.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_all.txt
tail -n 25 $O/pytest_all.txt
timeout 900 python bench.py --sweep > $O/bench_line.json 2> $O/bench_err.txt; echo "bench rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --no-dense > $O/bench_line_steps20_warmup5.json 2>/dev/null
DGS_HUB_CHAIN=0 timeout 600 python bench.py --no-dense --no-protocol > $O/bench_line_nohub.json 2>/dev/null
DGS_FOLD=0 timeout 600 python bench.py --no-dense --no-protocol > $O/bench_line_nofold.json 2>/dev/null
ls -la $O
head -c 3000 $O/bench_line.json
tail -n 5 $O/bench_err.txt
