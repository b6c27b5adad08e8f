#!/bin/bash
# Runs a Python command over the AddressSanitizer build of the emulated kernels:  tests/emu/run_asan.sh python tests/fuzz_emu.py 600 7
# (the interpreter is not instrumented, so the sanitizer's runtime is preloaded; leak checking is off - CPython's arenas)
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
export DGS_EMU_ASAN=1 LD_PRELOAD="$RT" ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:detect_stack_use_after_return=0:${ASAN_OPTIONS:-}"
exec "$@"
