#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command (run on the GPU box via gpurun).
set -u
TAG=${1:-r02}
EXTRA=${2:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o "$TAG" -- \
    python $GRAFT_REPO_ROOT/bench.py --cpu-pairs 0 --skip-secondary $EXTRA > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.log"
ls "$OUT"
