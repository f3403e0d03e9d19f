#!/bin/bash
# Round-6 A/B of k_runs_corr variants: library builds (libffsalign_<name>.so, `default` = libffsalign.so) x FFS_RUNS_SPLIT values.
#   bash profiles/runs_ab6.sh "default tpw2" "0 7"        (split 0 = the library's own rule)
cd "$GRAFT_REPO_ROOT"
PAIRS=${PAIRS:-4096}
for m in $1; do
  lib=ffsubsync_amd/libffsalign_$m.so; [ $m = default ] && lib=ffsubsync_amd/libffsalign.so
  [ -f $lib ] || { echo "build=$m missing"; continue; }
  for sp in $2; do
    echo -n "build=$m split=$sp "
    FFS_RUNS_SPLIT=$sp FFS_LIBRARY_PATH=$PWD/$lib timeout 300 python profiles/runs_quick.py $PAIRS auto 6000 512 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['solves_per_s']), {k:round(v,4) for k,v in d['kernels_us_per_pair'].items()}, d['ground_truth'], 'host_us/pair', round(d['host_us_per_pair'],3))"
  done
done
