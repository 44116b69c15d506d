# usage: bash scripts/bench_variants.sh "v:chunk v:chunk ..." [bench flags]  -- whole-step bench for several launch shapes
SPECS=$1; shift
for s in $SPECS; do
  v=${s%%:*}; c=${s##*:}
  CFD_VARIANT=$v CFD_CHUNK=$c timeout -k 10 120 python bench.py --no-cpu-baseline --steps 40 --warmup 5 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $v chunk $c: %.4f ms/step  %.2f G/s  %s' % (d['ms_per_step'], d['value']/1e9, '  '.join('%s %.4f ms' % (k['kernel'][:28], k['avg_pass_ms']) for k in d['roofline']['kernels'])))"
done
