# the other BASELINE configurations on one GPU, short runs (records for profiles/)
set -e
O=gpurun_out/r02_cfg; mkdir -p $O
timeout -k 10 300 python -u bench.py --case cavity --nx 2048 --ny 16384 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_cavity_2048x16384_slab.json 2> $O/err.log
echo cavity done
timeout -k 10 300 python -u bench.py --nx 8192 --ny 8192 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_8192sq.json 2>> $O/err.log
echo 8192 done
timeout -k 10 400 python -u bench.py --nx 16384 --ny 16384 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_16384sq_one_gpu.json 2>> $O/err.log
echo 16384 done
timeout -k 10 400 python -u bench.py --nx 768 --ny 768 --nz 768 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_768cube_one_gpu.json 2>> $O/err.log
echo 768 done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "%.2f G" % (d["value"]/1e9), "%.3f ms/step" % d["ms_per_step"], d["config"]["workload"], d["config"]["fields_finite_at_end"], "strict", (d.get("strict") or {}).get("value"))
PY
