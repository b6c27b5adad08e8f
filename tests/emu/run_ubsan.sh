#!/bin/bash
# Runs a Python command over the UndefinedBehaviorSanitizer build of the emulated kernels (make UBSAN=1):
#   tests/emu/run_ubsan.sh python -m pytest tests/test_emu_cpu.py -x -q
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.ubsan_standalone-x86_64.so)
export DGS_EMU_UBSAN=1 LD_PRELOAD="$RT" UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:${UBSAN_OPTIONS:-}"
exec "$@"
