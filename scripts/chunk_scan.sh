# usage: bash scripts/chunk_scan.sh <chunk> ...  -- the default bench with several values of CFD_CHUNK (x-planes per block of
# the plain sweep), same box, two rounds
for i in 1 2; do
  for c in "$@"; do
    CFD_CHUNK=$c timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunk $c: %.4f ms/step  %.2f G/s  %s' % (d['ms_per_step'], d['value']/1e9, '  '.join('%s %.4f ms' % (k['kernel'][:24], k['avg_pass_ms']) for k in d['roofline']['kernels'])))" || exit 1
  done
done
