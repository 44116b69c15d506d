# whole default (fast) steps, two builds alternating on one box: usage ab_step.sh <other lib> [bench args]
LIB=$1; shift
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f G  %.4f ms/step' % (d['value']/1e9, d['ms_per_step']), [ '%.4f' % k['avg_pass_ms'] for k in d['roofline']['kernels']])"; }
for i in 1 2 3; do
  echo -n "in-tree  "; timeout -k 10 200 env -u CFD_LIB bash -c "$(declare -f run); run $*" || exit 1
  echo -n "other    "; CFD_LIB=$PWD/$LIB timeout -k 10 200 bash -c "$(declare -f run); run $*" || exit 1
done
