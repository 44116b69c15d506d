# usage: bash scripts/gpu_bench.sh <tag> [bench flags...]   (run from the repo root on the GPU box)
set -e
TAG=${1:-run}; shift || true
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
timeout -k 10 600 python -u bench.py --steps 20 --warmup 3 "$@" > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err || { tail -20 gpurun_out/$TAG/bench.err; exit 1; }
cat gpurun_out/$TAG/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/$TAG/prof.log 2>&1 || { tail -20 gpurun_out/$TAG/prof.log; exit 1; }
find gpurun_out/$TAG/prof -name '*kernel_stats.csv' | head -1 | xargs cat
