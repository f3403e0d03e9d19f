#!/bin/bash
# Section timing of k_vad_tokenize_scan: library variants built with -DFFS_TOK_STOP=k return early (WRONG RESULTS) after
# 1 the coalesced load, 2 scan 1 (last valid frame), 3 scan 2 (island start), 4 scan 3 (island end), 5 the marker pass;
# `default` = the whole kernel.  Kernel time from rocprofv3's kernel trace (one 90-minute file, 54 chunks of 100 s).
#   for k in 1 2 3 4 5; do make -C ffsubsync_amd/csrc variant NAME=tok$k DEFS=-DFFS_TOK_STOP=$k; done
#   bash profiles/tok_sections.sh "tok1 tok2 tok3 tok4 tok5 default"
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for m in $1; do
  lib=ffsubsync_amd/libffsalign_$m.so; [ $m = default ] && lib=ffsubsync_amd/libffsalign.so
  [ -f $lib ] || { echo "build=$m missing"; continue; }
  d=/tmp/tok_$m; rm -rf $d
  FFS_LIBRARY_PATH=$PWD/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python profiles/vad_tokenize_rate.py > /dev/null 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo -n "build=$m "; python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'tokenize_scan' in r['Name']:
        print('calls',r['Calls'],'avg_us',round(float(r['AverageNs'])/1e3,2),'min_us',round(float(r['MinNs'])/1e3,2),'max_us',round(float(r['MaxNs'])/1e3,2))
PY
done
