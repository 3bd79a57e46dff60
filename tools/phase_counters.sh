#!/bin/bash
# Per-phase instruction counts of the small-scene kernel on the C2 frame: one rocprofv3 --pmc pass per phase-cut
# build (make -C raytracing_amd/csrc variant NAME=cutN DEFS=-DMRT_CUT=N; N = 8..14, plus abl1 = every tile sky,
# abl4 = prologue only, and the product library).  Run through gpurun; tools/phase_counters.py sums it up.
# usage: tools/phase_counters.sh <tag>
set -e
tag=$1
V=$GRAFT_REPO_ROOT/raytracing_amd/lib/variants
for v in full abl4 abl1 cut8 cut9 cut10 cut11 cut12 cut13 cut14; do
  if [ $v = full ]; then unset MRT_LIB; else export MRT_LIB=$V/libmrt_$v.so; fi
  bash $GRAFT_REPO_ROOT/tools/pmc.sh ${tag}_$v C2 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_BUSY_CYCLES > $GRAFT_REPO_ROOT/gpurun_out/phase_${tag}_$v.txt 2>&1
  echo "[phase] $v done"
done
