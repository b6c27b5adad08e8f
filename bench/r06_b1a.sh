#!/bin/bash
# round 6, GPU batch 1a: first hardware contact of everything written since round 3, in the order VERDICT r5 #1 asks for:
# device self-tests (hub: 14 shapes; fold: 9 families x 3 rounds, loaded) -> hub smoke (stop on hang) -> the default bench line and the
# kernel stats of the same command -> all -m gpu tests (--durations) -> sweep line, steps-20 line, hub off, fold on.
set -x
cd "$(dirname "$0")/.."
O=gpurun_out/r06b1
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/dgsparse-lib_amd
timeout 300 python -c "
import torch, time
from dgsparse import _capi
t = time.time()
_capi.ensure_hub_selftest(torch.device('cuda', 0))
print('hub gate', _capi.hub_gate(), 'threshold', _capi.hub_threshold(), 'per shape', _capi.selftest_detail()[16:30], round(time.time() - t, 2), 's')
for rounds, load in ((1, False), (3, True), (10, True)):
    t = time.time()
    print('fold selftest rounds', rounds, 'load', load, _capi.fold_selftest(rounds=rounds, load=load), 'gate', _capi.fold_gate(), round(time.time() - t, 2), 's')
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
tail -n 60 $O/pytest_all.txt
timeout 900 python bench.py --sweep > $O/bench_line.json 2> $O/bench_err.txt; echo "bench rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --no-dense > $O/bench_line_steps20_warmup5.json 2>/dev/null
DGS_HUB_CHAIN=0 timeout 600 python bench.py --no-dense --no-protocol > $O/bench_line_nohub.json 2>/dev/null
DGS_FOLD=1 timeout 600 python bench.py --no-dense --no-protocol > $O/bench_line_fold.json 2>/dev/null
ls -la $O
head -c 3000 $O/bench_line.json
tail -n 5 $O/bench_err.txt
