#!/bin/bash
# rocprofv3 kernel trace of the bench command (run on the GPU box through gpurun).
# usage: tools/prof_bench.sh <tag> [bench args...]
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $out/bench.json 2> $out/bench.err
