#!/bin/bash
# A/B of streaming (`nt`) stores / loads on the intermediates: libffsalign_nt<mask>.so are builds with -DFFS_NT=<mask>
# (1 = first-pass stores, 2 = mid-pass loads, 4 = mid-pass stores, 8 = last-pass loads; the product library is built
# with 15); full bench line, every leg.  `default` = libffsalign.so.
#   make -C ffsubsync_amd/csrc variant NAME=nt0 DEFS=-DFFS_NT=0;  bash profiles/nt_ab.sh nt0 default nt0 default
# (run3l was taken when the product default was still 0: `0 15 0 15` with a -DFFS_NT=15 variant)
cd "$GRAFT_REPO_ROOT"
for m in "$@"; do
  lib=ffsubsync_amd/libffsalign_$m.so; [ $m = default ] && lib=ffsubsync_amd/libffsalign.so
  [ -f $lib ] || continue
  echo "build=$m"
  FFS_LIBRARY_PATH=$PWD/$lib timeout 300 python bench.py --steps 6 --warmup 2 --cpu-pairs 0 --no-vad --e2e-files 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
us=lambda k:{n:round(v['us_per_pair'],2) for n,v in k.items()}
print(' headline', round(d['value']), us(d['kernels']), d['offset_match']['pairs_matching_reference_golden'])
print(' reflen', round(d['reference_length']['value']), us(d['reference_length']['kernels']), d['reference_length']['identical_pair_results'])
print(' windowless', round(d['windowless']['value']), us(d['windowless']['kernels']))
for k,v in d['single_ratio'].items(): print(' single', k, round(v['solves_per_s']), us(v['kernels']), v['pairs_matching_reference_golden'])
print(' bytes', round(d['byte_inputs']['value']), d['byte_inputs']['identical_pair_results'])"
done
