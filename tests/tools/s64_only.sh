#!/bin/bash
# Developer tool: the S64 instantiation of the plan kernel alone (-DDLSCQP_ONLY_S64), as assembly with the kernel's resources, or as a
# library for tests/tools/ab.py (run those with DLSCQP_NO_SELFTEST=1: the self-test wants every instantiation).
#   tests/tools/s64_only.sh asm  /tmp/x.s   [-D...]     resources of every kernel in the file are printed
#   tests/tools/s64_only.sh lib  /tmp/x.so  [-D...]
mode=$1; out=$2; shift 2
cd "$(dirname "$0")/../../dlsc_gc_planner_amd/csrc" || exit 1
F="-O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -Wno-pass-failed -I../../include -DDLSCQP_ONLY_S64"
if [ "$mode" = asm ]; then
  /opt/rocm/bin/hipcc $F --cuda-device-only -save-temps=obj -c "$@" -o "${out%.s}.o" dlscqp.hip 2>&1 | grep -v "hip-link\|^ "; mv "$(dirname "$out")"/dlscqp-hip-amdgcn-amd-amdhsa-gfx950.s "$out"
  python3 - "$out" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for e in re.split(r'\n  - ', txt[txt.find('amdhsa.kernels'):])[1:]:
    d = dict(re.findall(r'\.(\w+):\s+(\S+)', e))
    if 'plan_kernel' in d.get('name', ''):
        print(d['name'][22:70], 'vgpr', d.get('vgpr_count'), 'agpr', d.get('agpr_count'), 'sgpr', d.get('sgpr_count'), 'scratch', d.get('private_segment_fixed_size'),
              'vspill', d.get('vgpr_spill_count'), 'sspill', d.get('sgpr_spill_count'))
PY
else
  /opt/rocm/bin/hipcc $F -fPIC -shared "$@" -o "$out" dlscqp.hip qp_tables.cpp sfc_world.cpp grid_mapf.cpp 2>&1 | grep -v "hip-link"
fi
