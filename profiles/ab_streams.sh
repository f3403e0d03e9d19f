#!/bin/bash
# headline pairs through BatchAligner(streams=k): parts of a call on their own plans and HIP streams, so that one part's list
# extraction (memory-bound) can run beside another part's correlation (LDS / ALU-bound).   bash profiles/ab_streams.sh "default mix1" "1 2 4"
cd "$GRAFT_REPO_ROOT"
for m in $1; do
  lib=ffsubsync_amd/libffsalign_$m.so; [ $m = default ] && lib=ffsubsync_amd/libffsalign.so
  for st in $2; do
    echo -n "build=$m streams=$st "
    FFS_LIBRARY_PATH=$PWD/$lib timeout 300 python profiles/runs_quick.py ${PAIRS:-8192} auto 6000 512 $st 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['solves_per_s']), 'us/pair', round(d['us_per_pair'],4), {k:round(v,4) for k,v in d['kernels_us_per_pair'].items()}, d['ground_truth'])"
  done
done
