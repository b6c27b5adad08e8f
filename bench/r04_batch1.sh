#!/bin/bash
# round 4, GPU batch 1: design experiments (LDS-DMA gather, hub-chain feeding, cost of leaving the slice order, strict parts)
set -x
cd "$(dirname "$0")/.."
O=gpurun_out/r04b1
mkdir -p $O
export PYTHONPATH=$PWD:$PWD/dgsparse-lib_amd
hipcc --offload-arch=gfx950 -O3 experiments/lds_dma_gather.cpp -o /tmp/ldg 2>$O/ldg_build.log && timeout 600 /tmp/ldg > $O/lds_dma_gather.txt 2>&1
timeout 600 python bench/nocut_probe.py 64 > $O/nocut_probe.txt 2>&1
L=$PWD/dgsparse-lib_amd/csrc/build/dbg
timeout 600 python bench/strict_parts.py 64 > $O/strict_parts_default.txt 2>&1
DGS_LIB_PATH=$L/libdgs_dbg1.so timeout 600 python bench/strict_parts.py 64 > $O/strict_parts_nochain.txt 2>&1
DGS_LIB_PATH=$L/libdgs_dbg2.so timeout 600 python bench/strict_parts.py 64 > $O/strict_parts_nogather.txt 2>&1
timeout 900 python bench/mtx_bench.py --out $O/r04_mtx > $O/mtx_bench.txt 2>&1
tail -5 $O/*.txt
