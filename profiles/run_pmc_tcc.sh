#!/bin/bash
# HBM-traffic passes only (FETCH_SIZE, WRITE_SIZE) of run_pmc.sh -- refreshes traffic_per_pair.json cheaply.
set -u
OUT=${1:-$GRAFT_REPO_ROOT/gpurun_out/pmc}
EXTRA=${2:-}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --pairs ${PMC_PAIRS:-512} --steps 1 --warmup 1 --cpu-pairs 0 --no-profile --skip-secondary --pairs-in-flight ${PMC_PAIRS:-512} $EXTRA"
run() { name=$1; shift; rm -rf "$OUT/$name"; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- $CMD > "$OUT/$name.log" 2>&1; }
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
