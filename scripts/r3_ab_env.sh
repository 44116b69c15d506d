# usage: bash scripts/r3_ab_env.sh "<label>=<ENV=VAL ...>" ... -- the default bench under several environments, same box, two rounds
# e.g.   bash scripts/r3_ab_env.sh "trio256=CFD_TRIO_NTS=256" "pair=CFD_TRIO=0"
mkdir -p gpurun_out/ab
for round in 1 2; do
  for spec in "$@"; do
    label=${spec%%=*}; envs=${spec#*=}
    env $envs timeout -k 10 200 python bench.py --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/ab/$label.$round.json 2> gpurun_out/ab/$label.$round.err || { echo "$label failed"; tail -5 gpurun_out/ab/$label.$round.err; exit 1; }
    python - "$label" gpurun_out/ab/$label.$round.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print('%-12s %.4f ms/step  %.2f G/s  sclk %s  %s' % (sys.argv[1], d['ms_per_step'], d['value'] / 1e9, d['sclk_mhz']['while_queued_steps_run'],
      '  '.join('%s %.4f ms' % (k['kernel'][:28], k['avg_pass_ms']) for k in d['roofline']['kernels'])))
PY
  done
done
