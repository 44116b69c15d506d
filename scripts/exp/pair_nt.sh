set -e
run() { timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1  %.2f G  %.4f ms/step  ' % (d['value']/1e9, d['ms_per_step']), ['%s %.4f ms' % (k['kernel'][:14], k['avg_pass_ms']) for k in r['kernels']])"; }
run base
CFD_PAIR_NT=128 CFD_PAIR_CHUNK=152 run nt128_152
CFD_PAIR_NT=128 CFD_PAIR_CHUNK=76 run nt128_76
CFD_PAIR_NT=128 CFD_PAIR_CHUNK=103 run nt128_103
run base
