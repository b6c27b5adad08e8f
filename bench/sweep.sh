#!/bin/bash
# On the GPU box: rebuild spmm.hip with -D variants and time the headline workload.  bash bench/sweep.sh "VAR1" "VAR2" ...
cd "$(dirname "$0")/.."
for v in "$@"; do
  (cd dgsparse-lib_amd/csrc && touch spmm_impl.h && make -s -j8 FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include $v" >/dev/null 2>&1)
  for cfg in "" "--feat 128" "--feat 32"; do
    python bench.py --steps 30 --warmup 3 --no-cpu-baseline $cfg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v | $cfg |', d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done
done
