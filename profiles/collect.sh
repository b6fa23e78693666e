#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run from the repo root through gpurun):
#   bash profiles/collect.sh r01_weak_duality
# One rocprofv3 run per counter group (PMC passes carry --kernel-trace only, never a runtime/sys trace), the
# program itself after "--".  Summaries are written by profiles/summarize.py into profiles/<name>.{md,json}.
set -u
NAME=${1:-r05_final}
OUT=gpurun_out/prof_$NAME
STEPS=30
SWARMS=1024 # bench.py's default (round 6)
GROUPS_=4   # bench.py's default on one GPU: four swarm groups, four launches of 16 384 agents side by side per step
export TMPDIR=/tmp
B="python3 bench.py --steps $STEPS --warmup 4 --swarms $SWARMS --groups $GROUPS_ --headline-only"
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/sq1 -- $B > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/sq2 -- $B > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq3 -- $B > $OUT/sq3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq4 -- $B > $OUT/sq4.log 2>&1
# the FP64 share of the VALU instructions (4 cycles of a SIMD's vector pipe each, the others 2): what valu_issue_frac is priced with
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/sq5 -- $B > $OUT/sq5.log 2>&1
# matrix-core instructions: ZERO by design (active-set method: matrix-vector products with <= 16 columns) -- said with the counter
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/sq6 -- $B > $OUT/sq6.log 2>&1
python3 profiles/summarize.py $OUT $NAME --steps $((STEPS * GROUPS_)) --agents $((SWARMS * 64 / GROUPS_)) --concurrent $GROUPS_ > $OUT/summary.txt 2>&1
cp profiles/$NAME.md profiles/$NAME.json profiles/hbm_traffic_latest.json $OUT/ 2>/dev/null
# (the diagnostic builds are compiled on first use: their compiler output stays out of the files)
CLEAN='warning|^ +[0-9]+ \||^ +\||amdgpu.ids|warnings? generated|^In file included|note:'
python3 tests/tools/phase_cost.py 2>&1 | grep -Ev "$CLEAN" > $OUT/phase_cost.txt
python3 profiles/timeline.py 2>&1 | grep -Ev "$CLEAN" > $OUT/timeline.txt
python3 profiles/stamps.py --swarms 512 2>&1 | grep -Ev "$CLEAN" > $OUT/stamps.txt
python3 tests/tools/isa_phases.py 2>&1 | grep -Ev "$CLEAN" > $OUT/isa_phases.txt
cp $OUT/stamps.txt profiles/${NAME}_stamps.txt
cp $OUT/timeline.txt profiles/${NAME}_timeline.txt
cp $OUT/phase_cost.txt profiles/${NAME}_phase_cost.txt
cp $OUT/isa_phases.txt profiles/${NAME}_isa_phases.txt
# every kernel of the path in one trace: the full bench (closed-loop missions: half-space, corridor and advance kernels; stress batch)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/all -- python3 bench.py --steps 40 --no-cpu-baseline > $OUT/all.log 2>&1
python3 profiles/kernels.py $OUT/all $NAME > $OUT/kernels.txt 2>&1
# what is to be committed under profiles/ travels back with gpurun_out/
mkdir -p $OUT/for_profiles
cp profiles/${NAME}* profiles/hbm_traffic_latest.json $OUT/for_profiles/ 2>/dev/null
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/sq1 $OUT/sq2 $OUT/sq3 $OUT/sq4 $OUT/sq5 $OUT/sq6 $OUT/all
tail -5 $OUT/summary.txt
