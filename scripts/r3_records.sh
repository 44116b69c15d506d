# usage: bash scripts/r3_records.sh  -- the bench lines kept under profiles/r03/ besides the default one: every BASELINE
# configuration's one-GPU share, the strict kernels, the shipped dt, two virtual slabs
set -e
O=gpurun_out/r03_records
mkdir -p $O
run() { name=$1; shift; timeout -k 10 500 python -u bench.py "$@" > $O/$name.json 2> $O/$name.err || { tail -20 $O/$name.err; exit 1; }; python3 - $O/$name.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], "%.2f G" % (d["value"] / 1e9), "%.4f ms/step" % d["ms_per_step"], d["config"]["workload"],
      "finite" if d["config"]["fields_finite_at_end"] else "NOT FINITE", "frac %.3f" % d["roofline"]["frac"],
      "strict %.2f G" % ((d.get("strict") or {}).get("value", 0) / 1e9))
PY
}
run bench_4096sq_1000steps --steps 1000 --warmup 20 --no-cpu-baseline
run bench_4096sq_driver_command --steps 20 --warmup 5 --no-cpu-baseline
run bench_4096sq_dt1e-5_20steps --dt 1e-5 --steps 20 --warmup 0 --warmup-seconds 0 --no-cpu-baseline
run bench_4096sq_dt1e-7_20steps --dt 1e-7 --steps 20 --warmup 0 --warmup-seconds 0 --no-cpu-baseline
run bench_4096sq_strict --arith strict --steps 40 --warmup 5 --no-cpu-baseline
run bench_4096sq_two_virtual_slabs --virtual-slabs 2 --steps 100 --warmup 10 --no-cpu-baseline
run bench_8192sq --nx 8192 --ny 8192 --steps 50 --warmup 5 --no-cpu-baseline
run bench_cavity_2048x16384_slab --case cavity --nx 2048 --ny 16384 --steps 100 --warmup 10 --no-cpu-baseline
run bench_16384sq_one_gpu --nx 16384 --ny 16384 --steps 20 --warmup 3 --no-cpu-baseline
run bench_512cube_strict --nx 512 --ny 512 --nz 512 --arith strict --steps 8 --warmup 2 --no-cpu-baseline
run bench_256cube --nx 256 --ny 256 --nz 256 --steps 100 --warmup 10 --no-cpu-baseline
run bench_256cube_two_virtual_slabs --nx 256 --ny 256 --nz 256 --virtual-slabs 2 --steps 100 --warmup 10 --no-cpu-baseline
run bench_768cube_one_gpu --nx 768 --ny 768 --nz 768 --steps 10 --warmup 2 --no-cpu-baseline
