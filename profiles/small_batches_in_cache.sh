for cfg in "4 1" "8 1" "8 2" "12 3" "16 4" "32 4" "512 1"; do set -- $cfg; echo "pif=$1 streams=$2"; timeout 120 python bench.py --steps 5 --warmup 2 --pairs 2048 --cpu-pairs 0 --skip-secondary --pairs-in-flight $1 --streams $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), {k:round(v['us_per_pair'],2) for k,v in d.get('kernels',{}).items()})"; done
