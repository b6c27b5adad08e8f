#!/bin/bash
# round 4, GPU batch 2: new hub chain (two gather sets, role-specialised waves): bit-exactness + timing; cost of leaving the slice order
set -x
cd "$(dirname "$0")/.."
O=gpurun_out/r04b2
mkdir -p $O
export PYTHONPATH=$PWD:$PWD/dgsparse-lib_amd
timeout 900 python -m pytest tests/test_gpu_strict.py -x -q -m gpu > $O/pytest_strict.txt 2>&1
timeout 600 python bench/strict_parts.py 64 > $O/strict_parts.txt 2>&1
timeout 600 python bench/strict_time.py > $O/strict_time.txt 2>&1
timeout 600 python bench/nocut_probe.py 64 > $O/nocut_probe.txt 2>&1
tail -n 12 $O/pytest_strict.txt $O/strict_parts.txt $O/nocut_probe.txt
