#!/bin/bash
# One rocprofv3 --pmc pass (counters in their own run, no tracing domains) over a short render loop.
# usage: tools/pmc.sh <tag> <config> <counter> [<counter>...]
set -e
tag=$1; cfg=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out -o $tag -- python3 $GRAFT_REPO_ROOT/tools/exp_variants.py child $cfg > $out/run.log 2>&1
python3 - <<PY
import csv, collections, glob
f = glob.glob("$out/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    n[(k, r["Counter_Name"])] += 1
for k in agg:
    if "render" not in k: continue
    print(k)
    for c, v in agg[k].items(): print("   %-28s per-dispatch mean %.4g  (n=%d)" % (c, v / n[(k, c)], n[(k, c)]))
PY
