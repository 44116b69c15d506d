set -e
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b20.json 2>gpurun_out/b20.err
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b20b.json 2>>gpurun_out/b20.err
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/b200.json 2>>gpurun_out/b20.err
timeout -k 10 300 python bench.py --virtual-slabs 2 --no-cpu-baseline > gpurun_out/bv2.json 2>>gpurun_out/b20.err
python3 - <<PY
import json
for f in ("b20","b20b","b200","bv2"):
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, "%.2f G" % (d["value"]/1e9), "%.4f ms/step" % d["ms_per_step"], "extra warmup", d["warmup_extra_steps"], d["sclk_mhz"]["while_queued_steps_run"], [ "%.4f"%k["avg_pass_ms"] for k in d["roofline"]["kernels"]])
PY
