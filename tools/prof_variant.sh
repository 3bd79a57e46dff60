#!/bin/bash
# kernel durations (rocprofv3 kernel trace) of the exp_variants child under the current environment
# usage: VAR=... tools/prof_variant.sh <tag> <config>
set -e
tag=$1; cfg=$2
out=$GRAFT_REPO_ROOT/gpurun_out/pv_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 $GRAFT_REPO_ROOT/tools/exp_variants.py child $cfg > $out/run.log 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("$out/${tag}_kernel_stats.csv")):
    if "render" in r["Name"]: print("$tag", r["Name"][:40], "calls", r["Calls"], "avg_ns", r["AverageNs"], "min_ns", r["MinNs"])
PY
