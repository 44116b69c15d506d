# usage: bash scripts/variant_scan.sh "<variant>:<chunk>" ...  -- the default bench with other launch shapes of the plain 2-D sweep
# (CFD_VARIANT; "-" = the default shape with its own chunk rule), same box, two rounds.  A shape other than the default one
# runs a step as three launches, so only the plain sweep's time is comparable.
for i in 1 2; do
  for vc in "$@"; do
    v=${vc%%:*}; c=${vc##*:}
    if [ "$v" = "-" ]; then unset CFD_VARIANT CFD_CHUNK; else export CFD_VARIANT=$v CFD_CHUNK=$c; fi
    timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $vc: %.4f ms/step  %s' % (d['ms_per_step'], '  '.join('%s %.4f ms' % (k['kernel'][:24], k['avg_pass_ms']) for k in d['roofline']['kernels'])))" || exit 1
  done
done
