#!/bin/bash
# One rocprofv3 --pmc pass (counters in their own run, no tracing domains) over tools/ref_child.py.
# usage: tools/pmc_ref.sh <tag> <counter> [<counter>...]
set -e
tag=$1; shift 1
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out -o $tag -- python3 $GRAFT_REPO_ROOT/tools/ref_child.py 3 > $out/run.log 2>&1
python3 - <<PY
import csv, collections, glob
f = glob.glob("$out/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:48]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]);
    n[(k, r["Counter_Name"])] += 1
for k in agg:
    if "render" not in k: continue
    print(k)
    for c, v in agg[k].items(): print("   %-28s per-dispatch mean %.5g  (n=%d)" % (c, v / n[(k, c)], n[(k, c)]))
PY
