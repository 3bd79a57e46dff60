#!/bin/bash
# one --pmc pass over a short program; prints the per-dispatch mean of every counter for kernels matching "render_"
# usage: tools/pmc_quick.sh <name> "<counters>" <program args...>     (run through gpurun)
name=$1; ctrs=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/pmcq_$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $out -o p -- python3 "$@" > $out/log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
tot, n = {}, {}
for f in glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_" not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
        tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"]); n[k] = n.get(k, 0) + 1
for k in sorted(tot): print(k[0], k[1], "%.4g" % (tot[k] / n[k]), "x", n[k])
PY
