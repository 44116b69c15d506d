# usage: bash scripts/pair_chunk_scan.sh <chunk> ...  -- the default bench with several values of CFD_PAIR_CHUNK (x-planes per block
# of the two-sweeps kernel; "-" = the library's own choice), same box, two rounds
for i in 1 2; do
  for c in "$@"; do
    if [ "$c" = "-" ]; then unset CFD_PAIR_CHUNK; else export CFD_PAIR_CHUNK=$c; fi
    timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pair chunk $c: %.4f ms/step  %s' % (d['ms_per_step'], '  '.join('%s %.4f ms' % (k['kernel'][:24], k['avg_pass_ms']) for k in d['roofline']['kernels'])))" || exit 1
  done
done
