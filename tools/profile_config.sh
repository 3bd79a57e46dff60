#!/bin/bash
# rocprofv3 record of one bench config on the GPU box (run through gpurun): a kernel trace with stats, then
# counter passes (two of SQ activity, one of vector instructions by kind, FETCH_SIZE, WRITE_SIZE), each in a run of its own (MI355X_MICROARCH.md: at most 8 SQ / 4 TCC / 2 GRBM counters per pass,
# FETCH_SIZE and WRITE_SIZE never together, no tracing domains beside --pmc).  The program follows "--" directly.
# usage: tools/profile_config.sh <round-tag> <config> [bench args...]    e.g. tools/profile_config.sh r02 C3
# Output: gpurun_out/prof_<tag>_<config>/...; tools/summarize_profile.py turns it into profiles/<tag>_<config>_*.
set -e
tag=$1; cfg=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_${cfg}
mkdir -p $out
# provenance: the hash of the kernel sources this box runs (bench.py compares it with the tree it is run from later
# and marks the counters stale when they differ); tools/summarize_profile.py adds the git revision
python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import bench, json; print(json.dumps({'source_hash': bench.source_hash()}))" > $out/provenance.json
cd /tmp && export TMPDIR=/tmp
bench="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --config ${cfg%_brute} $@"
echo "[profile] $cfg: kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace -- $bench > $out/bench_under_rocprof.json 2> $out/trace.err
pass() { # name counters...
  local name=$1; shift
  echo "[profile] $cfg: pmc pass $name"
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $out -o pmc_$name -- $bench > $out/pmc_$name.log 2>&1
}
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_BUSY_CYCLES
pass sq2 SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VALU GRBM_GUI_ACTIVE
pass sq3 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT
pass fetch FETCH_SIZE
pass write WRITE_SIZE
echo "[profile] $cfg: done"
