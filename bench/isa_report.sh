#!/bin/bash
# What hipcc makes of the headline kernels for gfx950, without a GPU: registers / LDS / scratch of every spmm_fused instantiation of
# spmm_v4a.hip (code-object notes) and the wait / barrier skeleton of (a) the hub workgroup's gather-wave phase loop, (b) the
# in-kernel fold's hand-over, in spmm_fused<16,4,SUM,values,!ACC,HUB> - the kernel bench.py times.
#   bash bench/isa_report.sh > profiles/r05_isa_headline.txt        (~1 min)
set -e
cd "$(dirname "$0")/../dgsparse-lib_amd/csrc"
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include --cuda-device-only -c spmm_v4a.hip -o $T/v4a.co 2>/dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/v4a.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/v4a.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/v4a.elf > $T/notes
echo "== spmm_fused<G, V, OP, HAS_VAL, ACC, HUB> (spmm_v4a.hip, gfx950, hipcc $(/opt/rocm/bin/hipcc --version | grep -o 'HIP version: [0-9.-]*'))"
python3 - $T/notes <<'PY'
import re, subprocess, sys
t = open(sys.argv[1]).read()
rows = []
for k in re.split(r'\n\s+- \.agpr_count', t)[1:]:
    name = re.search(r'\.name:\s+(\S+)', k).group(1)
    if 'spmm_fused' not in name and 'spmm_small' not in name and 'spmm_combine' not in name:
        continue
    g = lambda f: int(re.search(r'\.' + f + r':\s+(\d+)', k).group(1))
    d = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    rows.append((re.sub(r'\(.*', '', d).replace('void dgs::', ''), g('vgpr_count'), g('vgpr_spill_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size')))
for r in sorted(rows):
    print(f'{r[0]:58s} vgpr {r[1]:3d}  spilled {r[2]:3d}  lds {r[3]:6d} B  scratch {r[4]:4d} B')
PY
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $T/v4a.elf | awk '/^[0-9a-f]+ <_ZN3dgs10spmm_fusedILi16ELi4ELi0ELb1ELb0ELb1EE/{p=1} p&&/^$/{if(n++>0)p=0} p' | sed 's#//.*##' > $T/f16.dis
echo
echo "== spmm_fused<16,4,0,true,false,true>: every s_barrier / s_waitcnt vmcnt / global load-store class / atomic, in program order"
echo "   (hub gather waves: the phase loop is the run of [vmcnt(18) .. vmcnt(8) -> s_barrier -> 4 x dword nt + 8 x dwordx4 -> s_barrier] pairs: the"
echo "    gathers of two register sets stay in flight across the barriers, which wait for lgkmcnt only;"
echo "    in-kernel fold: 'buffer_store_dwordx4 sc1' (one 16-byte write-through store per lane), later 's_waitcnt vmcnt(0)' + 'global_atomic_add', then '8 x buffer_load_dwordx4 sc1' back to back and counted waits in the fold)"
grep -nE "s_barrier|s_waitcnt vmcnt|global_atomic|sc1|s_setprio|buffer_wbl2|buffer_inv" $T/f16.dis | awk '{ $1=$1; print }' | sed -E 's/v\[[0-9:]+\]|v[0-9]+|s\[[0-9:]+\]//g' | awk '{k=$0; sub(/^[0-9]+: */,"",k); if (k==last) {c++} else { if (last!="") print (c>1? c" x ":"") last; last=k; c=1 } } END { print (c>1? c" x ":"") last }' | head -150
rm -rf $T
