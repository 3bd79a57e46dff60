#!/bin/bash
# HBM traffic of the bench command: FETCH_SIZE and WRITE_SIZE in separate --pmc passes
# (MI355X_MICROARCH.md, HBM section).  usage: tools/pmc_bench.sh <tag> [bench args]
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/traffic_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 180 rocprofv3 --pmc $c --output-format csv -d $out -o ${tag}_$c -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 100 --warmup 10 "$@" > $out/$c.log 2>&1
done
python3 - <<PY
import csv, collections, glob, json
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$out/${tag}_%s*counter_collection.csv" % c)[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if "render_" in r["Kernel_Name"] and r["Counter_Name"] == c:
            tot += float(r["Counter_Value"]); n += 1
    res[c] = {"mean_per_dispatch_KB": tot / max(n, 1), "dispatches": n}
print(json.dumps(res))
open("$out/summary.json", "w").write(json.dumps(res, indent=1))
PY
