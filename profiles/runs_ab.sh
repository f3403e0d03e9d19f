#!/bin/bash
# A/B of compile-time variants of the run-boundary kernels: libffsalign_<name>.so built with
#   make -C ffsubsync_amd/csrc variant NAME=<name> DEFS=-D...
# (`default` = libffsalign.so).  4096 headline pairs through profiles/runs_quick.py: per-kernel HIP-event times + ground truth.
#   bash profiles/runs_ab.sh default tpw1 ...
cd "$GRAFT_REPO_ROOT"
PAIRS=${PAIRS:-4096}
for m in "$@"; do
  lib=ffsubsync_amd/libffsalign_$m.so; [ $m = default ] && lib=ffsubsync_amd/libffsalign.so
  [ -f $lib ] || { echo "build=$m missing"; continue; }
  echo -n "build=$m "
  FFS_LIBRARY_PATH=$PWD/$lib timeout 300 python profiles/runs_quick.py $PAIRS auto 6000 512 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['solves_per_s']), {k:round(v,4) for k,v in d['kernels_us_per_pair'].items()}, d['ground_truth'], 'host_us/pair', round(d['host_us_per_pair'],3))"
done
