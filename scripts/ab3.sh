# usage: bash scripts/ab3.sh "lib1.so lib2.so ..." [bench flags] -- the default bench with the in-tree library and other builds, alternating, same box
LIBS=$1; shift
for i in 1 2; do
  for lib in "" $LIBS; do
    CFD_LIB=$lib timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${lib:-in-tree}: %.4f ms/step  %.2f G/s  %s' % (d['ms_per_step'], d['value']/1e9, '  '.join('%s %.4f ms' % (k['kernel'][:24], k['avg_pass_ms']) for k in d['roofline']['kernels'])))" || exit 1
  done
done
