#!/bin/bash
# Resource usage (compiler remarks) of every kernel of the library: name, VGPRs, AGPRs, scratch, occupancy, SGPR spills.
cd "$(dirname "$0")/../../dlsc_gc_planner_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off \
  -I../../include -Rpass-analysis=kernel-resource-usage "$@" -o /tmp/kres.so dlscqp.hip qp_tables.cpp sfc_world.cpp grid_mapf.cpp 2>&1 |
  grep -A10 "Function Name: " | grep -E "Function Name| VGPRs:|AGPRs|Scratch|Occup|SGPRs Spill" |
  sed 's/.*remark: *//; s/\[-R.*//; s/_ZN12_GLOBAL__N_1[0-9]*//; s/EEEvN6dlscqp.*//; s/Function Name: //; s/ScratchSize \[bytes\/lane\]/scratch/; s/Occupancy \[waves\/SIMD\]/occ/' |
  paste - - - - - - | tr -s ' \t' ' '
