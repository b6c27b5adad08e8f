#!/bin/bash
# Compile-time parameter sweep on the GPU box: rebuild the SpMM instantiation units with extra -D flags and time a
# list of bench.py workloads.  Flags understood by csrc/spmm_impl.h: DGS_T1 (stream/units threshold), DGS_T2, DGS_CAP
# (LDS tile), DGS_KU1 (gather window), DGS_NBU (unit blocks), DGS_NT (non-temporal streaming), DGS_B_NT (non-temporal
# gathers), DGS_XCD_REMAP.
#   bash bench/sweep.sh "-DDGS_T1=32" "-DDGS_T1=64 -DDGS_KU1=8"
#   CFGS=$'--feat 128\n--cols local' bash bench/sweep.sh "-DDGS_XCD_REMAP=0" "-DDGS_XCD_REMAP=1"
cd "$(dirname "$0")/.."
CFGS=${CFGS:-$'\n--feat 128\n--feat 32'}
for v in "$@"; do
  (cd dgsparse-lib_amd/csrc && touch spmm_impl.h && make -s -j8 FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include $v" ../dgsparse/libdgsparse_hip.so >/dev/null 2>&1)
  while IFS= read -r cfg; do
    python bench.py --steps 30 --warmup 3 --no-cpu-baseline $cfg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v | $cfg |', d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done <<< "$CFGS"
done
