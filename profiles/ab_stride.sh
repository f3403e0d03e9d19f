# plan-owned list stride (entries of 8 bytes per slot) by environment: bash profiles/ab_stride.sh "4096 4160 4096 4448 5120 3104"
cd "$GRAFT_REPO_ROOT"
for st in ${1:-4096 32768 4096 2048}; do
  echo -n "stride=$st "
  FFS_RUNS_STRIDE=$st timeout 300 python profiles/runs_quick.py 8192 auto 6000 512 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['solves_per_s']), {k:round(v,4) for k,v in d['kernels_us_per_pair'].items()}, d['ground_truth'])"
done
