#!/bin/bash
# What hipcc makes of the headline kernels for gfx950, without a GPU: registers / LDS / scratch of every spmm_fused / spmm_small /
# spmm_combine instantiation of spmm_v4a.hip (sum, mean, masked sum) and spmm_v4b.hip (max, min) (code-object notes) and the wait /
# barrier skeleton of (a) the hub workgroup's gather-wave phase loop in spmm_fused<16,4,SUM,values,!ACC,HUB,!FOLD> - the kernel
# bench.py times - and (b) the in-kernel fold's hand-over in its FOLD twin (DGS_FOLD=1).  tests/test_isa_cpu.py pins the same facts
# on the code objects inside the shipped libdgsparse_hip.so.
#   bash bench/isa_report.sh > profiles/r06_isa_headline.txt        (~2.5 min)
set -e
cd "$(dirname "$0")/../dgsparse-lib_amd/csrc"
T=$(mktemp -d)
for tu in v4a v4b; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include --cuda-device-only -c spmm_$tu.hip -o $T/$tu.co 2>/dev/null &
done
wait
: > $T/notes
for tu in v4a v4b; do
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/$tu.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/$tu.elf
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/$tu.elf >> $T/notes
done
echo "== spmm_fused<G, V, OP, HAS_VAL, ACC, HUB, FOLD> (spmm_v4a.hip + spmm_v4b.hip, gfx950; OP: 0 sum 1 max 2 min 3 mean 4 masked sum; hipcc $(/opt/rocm/bin/hipcc --version | grep -o 'HIP version: [0-9.-]*'))"
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
skeleton() {  # $1 = mangled-name prefix
  /opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $T/v4a.elf | awk -v pat="^[0-9a-f]+ <$1" '$0 ~ pat {p=1} p&&/^$/{if(n++>0)p=0} p' | sed 's#//.*##' > $T/k.dis
  grep -nE "s_barrier|s_waitcnt vmcnt|global_atomic|sc1|s_setprio|buffer_wbl2|buffer_inv" $T/k.dis | awk '{ $1=$1; print }' | sed -E 's/v\[[0-9:]+\]|v[0-9]+|s\[[0-9:]+\]//g' | awk '{k=$0; sub(/^[0-9]+: */,"",k); if (k==last) {c++} else { if (last!="") print (c>1? c" x ":"") last; last=k; c=1 } } END { print (c>1? c" x ":"") last }' | head -150
}
echo
echo "== spmm_fused<16,4,0,true,false,true,false> (the default: fold off): every s_barrier / s_waitcnt vmcnt / sc1 access / atomic, in program order"
echo "   (hub gather waves: the phase loop is the run of [vmcnt(18) .. vmcnt(8) -> s_barrier -> 4 x dword nt + 8 x dwordx4 -> s_barrier] pairs: the"
echo "    gathers of two register sets stay in flight across the barriers, which wait for lgkmcnt only; NO atomic, NO sc1 access: the fold is a compile-time twin)"
skeleton _ZN3dgs10spmm_fusedILi16ELi4ELi0ELb1ELb0ELb1ELb0EE
echo
echo "== spmm_fused<16,4,0,true,false,true,true> (DGS_FOLD=1): the same, with the in-kernel fold's hand-over:"
echo "   'buffer_store_dwordx4 sc1' (one 16-byte write-through store per lane), a unit later 's_waitcnt vmcnt(0)' + 'global_atomic_add', then"
echo "   '8 x buffer_load_dwordx4 sc1' back to back and counted waits in the fold"
skeleton _ZN3dgs10spmm_fusedILi16ELi4ELi0ELb1ELb0ELb1ELb1EE
rm -rf $T
