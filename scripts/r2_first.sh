# first GPU pass of the round: all GPU tests, the no-alloc test, default bench, 3-D bench (baseline for this box)
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_first
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -x -q --timeout 300 --timeout-method thread -m gpu > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -5 $O/gpu_tests.log
timeout -k 10 200 python -u -m pytest tests/wip_no_alloc.py -x -q --timeout 120 --timeout-method thread > $O/no_alloc.log 2>&1 || { tail -40 $O/no_alloc.log; exit 1; }
tail -3 $O/no_alloc.log
timeout -k 10 300 python -u -m pytest tests/test_gpu_steps.py -x -q -s --timeout 300 --timeout-method thread > $O/steps.log 2>&1 || { tail -40 $O/steps.log; exit 1; }
grep -E "fast step|step [0-9]+:|worst|passed" $O/steps.log
timeout -k 10 600 python -u bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cut -c1-400 $O/bench.json
timeout -k 10 600 python -u bench.py --nx 512 --ny 512 --nz 512 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_512cube.json 2> $O/bench_512cube.err || { tail -20 $O/bench_512cube.err; exit 1; }
cut -c1-400 $O/bench_512cube.json
