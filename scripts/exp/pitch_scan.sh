# does the power-of-two field stride (ny * 8 B = 32 KiB at ny = 4096) cost anything?  same nx, neighbouring ny
set -e
for ny in 4080 4096 4112 4320 4096; do
  timeout -k 10 200 python bench.py --nx 4096 --ny $ny --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ny $ny  %.2f G  %.4f ms/step  ' % (d['value']/1e9, d['ms_per_step']), ['%s %.4f ms %.3f' % (k['kernel'][:14], k['avg_pass_ms'], k['frac']) for k in r['kernels']])"
done
